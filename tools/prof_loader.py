"""Where does a loader-fed step spend its wall time?  python3 tools/prof_loader.py [workers] [lazy 0|1]
Times, per step: the main thread blocked in the DataLoader (`wait`), the pin thread's RaggedBatch.pin_memory (`pin`), the upload call, train_loop
(host side of the launch sequence) -- next to the steady-state step time."""
import importlib, os, sys, time, tempfile, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
PKG = "3d-semantic-segmentation-amp-net_amd"
sub = lambda n: importlib.import_module(PKG + "." + n)
synth, D, C, A, M, T, U = sub("synthetic"), sub("pointNet.datasets"), sub("pointNet.collate_fns"), sub("pointNet.amp_train"), sub("pointNet.model.pointnetAtt"), sub("trainer"), sub("utils.utils")
S, P = sub("pointNet.amp_step"), sub("pointNet.prefetch")
nw = int(sys.argv[1]) if len(sys.argv) > 1 else 4
lazy = (sys.argv[2] != "0") if len(sys.argv) > 2 else True
sub("_lib").set_matrix_precision("f32x3")
dev = torch.device("cuda:0")
root = tempfile.mkdtemp(prefix="ampnet_pl_")
try:
    paths = synth.write_dataset(root, n_train=64, n_val=0, n_test=0, n_points=2048, seed=7000, max_w=9)
    names = sorted(f[len("kmeans_"):] for f in os.listdir(paths["data"]) if f.startswith("kmeans_")) * 48
    ds = D.LidarKmeansDataset(paths["data"], task="segmentation", number_of_points=2048, files=names, lazy=lazy)
    enc = M.BasePointNet(point_dimension=3, return_local_features=True, global_feat_dim=256, device=dev)
    att = M.SegmentationWithAttention(256, 8, num_classes=5, local_dim=64, device=dev)
    ce = torch.nn.CrossEntropyLoss(weight=torch.FloatTensor([1, 2, 2, 1, 1]).to(dev), reduction="mean", ignore_index=-1)
    opt_p, opt_a = T.FusedAdam(enc.parameters(), lr=1e-3), T.FusedAdam(att.parameters(), lr=1e-3)
    U.limit_host_threads(reserve=nw)
    tm = dict(pin=[], wait=[], upload=[], loop=[])
    orig_pin = C.RaggedBatch.pin_memory
    def timed_pin(self):
        t0 = time.perf_counter(); r = orig_pin(self); tm["pin"].append(time.perf_counter() - t0); return r
    C.RaggedBatch.pin_memory = timed_pin
    orig_up = P.DevicePrefetcher._upload
    def timed_up(self, batch):
        t0 = time.perf_counter(); r = orig_up(self, batch); tm["upload"].append(time.perf_counter() - t0); return r
    P.DevicePrefetcher._upload = timed_up
    loader = torch.utils.data.DataLoader(ds, batch_size=64, shuffle=True, num_workers=nw, drop_last=True, collate_fn=C.collate_seq_ragged, pin_memory=True,
                                         persistent_workers=os.environ.get('PERSISTENT') == '1')
    class Waited:
        def __len__(self): return len(loader)
        def __iter__(self):
            it = iter(loader)
            while True:
                t0 = time.perf_counter()
                try: b = next(it)
                except StopIteration: return
                tm["wait"].append(time.perf_counter() - t0)
                yield b
    orig_loop = A.train_loop
    def timed_loop(*a, **k):
        t0 = time.perf_counter(); r = orig_loop(*a, **k); tm["loop"].append(time.perf_counter() - t0); return r
    A.train_loop = timed_loop
    epochs = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    for ep in range(epochs):
        for v in tm.values():
            v.clear()
        evs = []
        orig_loop2 = A.train_loop
        def ev_loop(*a, **k):
            e = torch.cuda.Event(enable_timing=True); e.record(); evs.append((time.perf_counter(), e)); return orig_loop2(*a, **k)
        A.train_loop = ev_loop
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        A._epoch(Waited(), True, enc, att, opt_p, opt_a, ce, 0)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        A.train_loop = orig_loop2
        n = len(tm["loop"]); h = n // 2
        print(f"epoch {ep}: workers {nw} lazy {lazy}: {n} steps, epoch {t1-t0:.2f} s")
        for k, v in tm.items():
            v = np.asarray(v[len(v) // 2:]) * 1e3
            print(f"  {k:7s} mean {v.mean():7.2f} ms  median {np.median(v):7.2f}  max {v.max():7.2f}   (second half, {len(v)} calls)")
        host = [round((evs[i][0] - evs[i - 1][0]) * 1e3, 1) for i in range(1, len(evs))]
        gpu = [round(evs[i - 1][1].elapsed_time(evs[i][1]), 1) for i in range(1, len(evs))]
        print("  host ms between step starts:", host)
        print("  gpu  ms between step starts:", gpu)
finally:
    shutil.rmtree(root, ignore_errors=True)
