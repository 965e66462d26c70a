"""Times the REFERENCE's own train_loop next to the oracle port on the same inputs (the ratio that
lets bench.py's `cpu_baseline` (kind "port": the oracle, which travels to the GPU box) stand for the reference's CPU speed,
which cannot travel).

Run only in the build container (it reads /root/reference):   python tests/golden/time_reference.py [B] [steps]
Writes tests/golden/reference_vs_port.json = {"ratio_train": port points/s / reference points/s, ...}; BASELINE.md and
bench.py (`cpu_baseline.reference_ratio`) quote it.  Both sides: torch CPU fp32, the same thread count, warm.
"""
import json
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as G                                   # noqa: E402  (stubs + loaders, nothing is regenerated)
from oracle import ampnet_oracle as O                     # noqa: E402

synth, P = G.synth, G.P


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    threads = len(os.sched_getaffinity(0))
    torch.set_num_threads(threads)
    N, W = 2048, 9
    G.install_stubs()
    tr = G.load_script(os.path.join(G.REF, "pointNet/self-attention/train_pointnet-attention.py"), "ref_train_att")
    pc, tg, cent, _ = synth.sample_batch(7, B, N, max_w=W)
    names = [f"f{i}" for i in range(B)]
    ce = torch.nn.CrossEntropyLoss(weight=torch.FloatTensor([1, 2, 2, 1, 1]), reduction="mean", ignore_index=-1)
    out = {"B": B, "W": W, "N": N, "threads": threads, "steps": steps, "torch": torch.__version__}
    pts = B * W * N

    # ---- the reference: train_loop(train=True) and train_loop(train=False), its own modules and torch.optim.Adam ----
    enc, att = G.ref_models(3, 4)
    opt_p, opt_a = torch.optim.Adam(enc.parameters(), lr=1e-3), torch.optim.Adam(att.parameters(), lr=1e-3)

    def ref_step(train):
        data = (torch.from_numpy(pc.copy()), torch.from_numpy(tg.copy()), names, torch.from_numpy(cent))
        if train:
            tr.train_loop(data, opt_p, opt_a, ce, enc, att, None, "segmentation", True, 0, 0)
        else:
            with torch.no_grad():
                tr.train_loop(data, opt_p, opt_a, ce, enc, att, None, "segmentation", False, 0, 0)

    # ---- the port: what bench.py's cpu_baseline leg runs (forward_windows + loss + backward + Adam / eval forward) ----
    ep = {k: torch.from_numpy(v).requires_grad_(True) for k, v in synth.make_params(3, P.ENC_PARAMS).items()}
    hp = {k: torch.from_numpy(v).requires_grad_(True) for k, v in synth.make_params(4, P.HEAD_PARAMS).items()}
    eb = {k: torch.from_numpy(v) for k, v in synth.make_buffers(3, P.ENC_BUFFERS).items()}
    hb = {k: torch.from_numpy(v) for k, v in synth.make_buffers(4, P.HEAD_BUFFERS).items()}
    state = {id(v): (torch.zeros_like(v), torch.zeros_like(v)) for d in (ep, hp) for v in d.values()}
    pct, tgt, cet = torch.from_numpy(pc), torch.from_numpy(tg), torch.from_numpy(cent)
    n_port = [0]

    def port_step(train):
        if not train:
            with torch.no_grad():
                lg, _, _, _ = O.forward_windows(ep, eb, hp, hb, pct, tgt, cet, False, False)
                O.predictions(lg)
            return
        lg, tpc, ft, _ = O.forward_windows(ep, eb, hp, hb, pct, tgt, cet, True, True)
        c, r = O.loss_terms(lg, tpc, ft)
        for d in (ep, hp):
            for v in d.values():
                v.grad = None
        (c + 0.001 * r).backward()
        n_port[0] += 1
        with torch.no_grad():
            for d in (ep, hp):
                for v in d.values():
                    m, s = state[id(v)]
                    O.adam_step(v, v.grad, m, s, n_port[0])

    for mode, train in (("train", True), ("eval", False)):
        for name, fn in (("reference", ref_step), ("port", port_step)):
            np.random.seed(0)
            fn(train)                                        # warm-up
            ts = []
            for _ in range(steps):
                t0 = time.perf_counter()
                fn(train)
                ts.append(time.perf_counter() - t0)
            out[f"{name}_{mode}_s_per_step"] = round(float(np.mean(ts)), 4)
            out[f"{name}_{mode}_points_per_s"] = round(pts / float(np.mean(ts)), 1)
            print(name, mode, out[f"{name}_{mode}_s_per_step"], "s/step", out[f"{name}_{mode}_points_per_s"], "points/s", flush=True)
        out[f"ratio_{mode}"] = round(out[f"port_{mode}_points_per_s"] / out[f"reference_{mode}_points_per_s"], 3)
    with open(os.path.join(HERE, "reference_vs_port.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
