#!/usr/bin/env python3
"""bench.py -- AMP-Net hot path on MI355X: synthetic ALS windows of 2048 points x 9 features.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--mode fwd|train] [--batch B]

One "step" = one pass of the hot path over one batch resident in HBM:
  fwd   (BASELINE.json configs[1]): AMP-Net forward, eval mode, fp32, B=32 samples x W=9 windows x N=2048 points
        -> logits + argmax (train_pointnet-attention.py train_loop(train=False) without the host legs).
  train (BASELINE.json configs[2], when the backward path is built): forward + loss + backward + 2x Adam, B=64.
For N > 1 the driver launches one rank per GPU (torch.distributed, backend nccl = RCCL); every rank processes
its own batch (data parallel, weak scaling); in train mode gradients are all-reduced.
Rank 0 prints ONE JSON line.  `value` = points/s over all ranks, inputs resident in HBM before the timed region.
The roofline object comes from a SECOND pass of the same K steps with HIP events around every kernel launch
(ampnet_profile_*), so that event recording does not sit inside the throughput number; both times are printed.
"""
import argparse
import ctypes
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
PKG = "3d-semantic-segmentation-amp-net_amd"

PEAK_MFMA_F32_TFLOPS = 157.3      # MI355X_MICROARCH.md: fp32 matrix peak
PEAK_HBM_GBPS = 8000.0            # HBM3E spec
PEAK_MFMA_BF16_TFLOPS = 2500.0    # dense bf16 matrix peak (no sparsity)
RIDGE = PEAK_MFMA_F32_TFLOPS * 1e12 / (PEAK_HBM_GBPS * 1e9)
N_POINTS, N_WIN = 2048, 9


def sub(name=""):
    return importlib.import_module(PKG + ("." + name if name else ""))


def build_models(device, train):
    synth, P = sub("synthetic"), sub("params")
    M = sub("pointNet.model.pointnetAtt")
    enc = M.BasePointNet(point_dimension=3, return_local_features=True, global_feat_dim=256, device=device)
    att = M.SegmentationWithAttention(256, 8, num_classes=5, local_dim=64, device=device)
    sd = {k: torch.from_numpy(v) for k, v in synth.make_params(3, P.ENC_PARAMS).items()}
    sd.update({k: torch.from_numpy(v) for k, v in synth.make_buffers(3, P.ENC_BUFFERS).items()})
    enc.load_state_dict(sd, strict=False)
    sd = {k: torch.from_numpy(v) for k, v in synth.make_params(4, P.HEAD_PARAMS).items()}
    sd.update({k: torch.from_numpy(v) for k, v in synth.make_buffers(4, P.HEAD_BUFFERS).items()})
    att.load_state_dict(sd, strict=False)
    enc.train(train)
    att.train(train)
    return enc, att


def profile_read():
    L = sub("_lib").lib()
    n_max = 64
    names = ctypes.create_string_buffer(64 * n_max)
    ms = (ctypes.c_double * n_max)()
    calls = (ctypes.c_longlong * n_max)()
    flops = (ctypes.c_double * n_max)()
    nbytes = (ctypes.c_double * n_max)()
    n = L.ampnet_profile_read(n_max, names, ms, calls, flops, nbytes)
    out = []
    for i in range(max(n, 0)):
        nm = names.raw[64 * i:64 * (i + 1)].split(b"\0")[0].decode()
        out.append(dict(name=nm, ms=ms[i], calls=calls[i], flops=flops[i], bytes=nbytes[i]))
    return out


def pmc_traffic_for(name, rows=None, table="pmc_traffic.json"):
    """HBM bytes per launch (read + write) of the kernel bench.py calls `name`, from the rocprofv3 PMC passes summarised
    in profiles/pmc_traffic.json (profiles/pmc_traffic.py; FETCH_SIZE x 2 on gfx950, WRITE_SIZE exact; the bf16_store step has its own
    table).  None when the file or an unambiguous match is missing."""
    path = os.path.join(ROOT, "profiles", table)
    if not os.path.exists(path):
        return None
    table = json.load(open(path))
    v = table.get("events", {}).get(name)
    if not v or (rows is not None and v.get("workload_rows", table.get("workload_rows")) != rows):
        return None                                     # the counters were collected on a different batch
    return round(v["read_bytes"] + (v["write_bytes"] or 0.0))


def pmc_counters_for(name, rows=None, tables=("pmc_traffic.json", "r03_pmc_counters.json")):
    """(matrix-pipe busy fraction, shader clock in GHz) of the kernel bench.py calls `name`, from the rocprofv3 SQ passes summarised in
    profiles/r02_pmc_counters.json (profiles/pmc_counters.py): SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8), i.e. at the
    clock the chip actually held -- roofline.frac prices the same kernel against the 2.4 GHz peak.  None when not collected."""
    tpath = os.path.join(ROOT, "profiles", tables[0])
    cpath = os.path.join(ROOT, "profiles", tables[1])
    if not (os.path.exists(tpath) and os.path.exists(cpath)):
        return None
    table = json.load(open(tpath))
    ev = table.get("events", {}).get(name)
    if not ev or "grid" not in ev or (rows is not None and ev.get("workload_rows", table.get("workload_rows")) != rows):
        return None
    c = json.load(open(cpath)).get(f"{ev['symbol']}|grid={ev['grid']}")
    if not c or "mfma_busy" not in c:
        return None
    return {"mfma_busy": round(c["mfma_busy"], 4), "sclk_ghz": round(c.get("sclk_ghz", 0.0), 3), "launch_us_under_pmc": round(c.get("duration_us", 0.0), 1)}


def roofline_from(rows, work_rows=None, tables=("pmc_traffic.json", "r03_pmc_counters.json"), hbm_from_pmc=False):
    """The roofline object of the kernel with the largest share of the profiled steps.  hbm_from_pmc (the bf16 legs, which are HBM-bound):
    achieved = PMC HBM bytes per launch / launch time against the 8 TB/s peak, algorithmic bytes next to it."""
    if not rows:
        return None
    top = max(rows, key=lambda r: r["ms"])
    if hbm_from_pmc and not top["name"].endswith(" bf16"):
        hbm_from_pmc = False                 # mode 1 keeps the fp32 fused backward: its dominant kernel is priced like the fp32 step's (MFMA or HBM by intensity)
    if hbm_from_pmc:
        per_ms = top["ms"] / top["calls"]
        traffic = pmc_traffic_for(top["name"], work_rows, tables[0])
        alg = top["bytes"] / top["calls"]
        used = traffic if traffic else alg
        ach = used / (per_ms * 1e-3) / 1e9
        return dict(bound="hbm", kernel=top["name"], achieved=round(ach, 1), peak=PEAK_HBM_GBPS, unit="GB/s", frac=round(ach / PEAK_HBM_GBPS, 4),
                    traffic=traffic, bytes_source="pmc" if traffic else "algorithmic (no PMC table for this mode)", algorithmic_bytes=round(alg),
                    mfma_busy=pmc_counters_for(top["name"], work_rows, tables), tflops=round(top["flops"] / top["calls"] / (per_ms * 1e-3) / 1e12, 1),
                    launch_ms=round(per_ms, 4), launches=int(top["calls"]), share_of_step=round(top["ms"] / sum(r["ms"] for r in rows), 3))
    per_ms = top["ms"] / top["calls"]
    intensity = top["flops"] / max(top["bytes"], 1.0)
    is_bf16 = top["name"].endswith(" bf16")
    if top["name"].endswith(" x3"):
        # split kernels (precision mode f32x3): every algorithmic product is SIX v_mfma_f32_32x32x16_bf16 partial products, so the matrix pipe
        # executes 6 x the algorithmic flops; priced on those against the dense bf16 peak, the algorithmic rate stated beside it
        alg = top["flops"] / top["calls"] / (per_ms * 1e-3) / 1e12
        ach = 6.0 * alg
        return dict(bound="mfma", kernel=top["name"], achieved=round(ach, 2), peak=PEAK_MFMA_BF16_TFLOPS, unit="TFLOP/s", frac=round(ach / PEAK_MFMA_BF16_TFLOPS, 4),
                    flops_counted="executed bf16 MFMA flops = 6 x algorithmic (three-term split of both operands, six exact partial products)",
                    algorithmic_tflops=round(alg, 2), algorithmic_frac_of_f32_mfma_peak=round(alg / PEAK_MFMA_F32_TFLOPS, 4),
                    traffic=pmc_traffic_for(top["name"], work_rows, tables[0]), mfma_busy=pmc_counters_for(top["name"], work_rows, tables),
                    algorithmic_bytes=round(top["bytes"] / top["calls"]), launch_ms=round(per_ms, 4),
                    launches=int(top["calls"]), share_of_step=round(top["ms"] / sum(r["ms"] for r in rows), 3))
    peak_tf = PEAK_MFMA_BF16_TFLOPS if is_bf16 else PEAK_MFMA_F32_TFLOPS
    if intensity >= peak_tf * 1e3 / PEAK_HBM_GBPS:
        ach = top["flops"] / top["calls"] / (per_ms * 1e-3) / 1e12
        return dict(bound="mfma", kernel=top["name"], achieved=round(ach, 2), peak=peak_tf, unit="TFLOP/s",
                    frac=round(ach / peak_tf, 4), traffic=pmc_traffic_for(top["name"], work_rows, tables[0]), mfma_busy=pmc_counters_for(top["name"], work_rows, tables),
                    algorithmic_bytes=round(top["bytes"] / top["calls"]), launch_ms=round(per_ms, 4),
                    launches=int(top["calls"]), share_of_step=round(top["ms"] / sum(r["ms"] for r in rows), 3))
    ach = top["bytes"] / top["calls"] / (per_ms * 1e-3) / 1e9
    return dict(bound="hbm", kernel=top["name"], achieved=round(ach, 1), peak=PEAK_HBM_GBPS, unit="GB/s",
                frac=round(ach / PEAK_HBM_GBPS, 4), traffic=pmc_traffic_for(top["name"], work_rows, tables[0]), launch_ms=round(per_ms, 4), launches=int(top["calls"]),
                share_of_step=round(top["ms"] / sum(r["ms"] for r in rows), 3))


def cpu_baseline(mode, budget_s=20.0):
    """The oracle (CPU restatement of the reference path, torch fp32 on the host cores) on a bounded sample of the
    same workload: B_cpu samples x 9 windows x 2048 points, eval forward (fwd) or forward+backward+Adam (train)."""
    from oracle import ampnet_oracle as O
    synth, P = sub("synthetic"), sub("params")
    # the threads this process may actually run on (the GPU box gives a 16-CPU share of a 256-thread host)
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, int(os.environ.get("AMPNET_CPU_THREADS", "16"))))
    torch.set_num_threads(cores)
    Bc = 2 if mode == "fwd" else 4
    pc, tg, cent, _ = synth.sample_batch(7, Bc, N_POINTS, max_w=N_WIN)
    ep = {k: torch.from_numpy(v) for k, v in synth.make_params(3, P.ENC_PARAMS).items()}
    eb = {k: torch.from_numpy(v) for k, v in synth.make_buffers(3, P.ENC_BUFFERS).items()}
    hp = {k: torch.from_numpy(v) for k, v in synth.make_params(4, P.HEAD_PARAMS).items()}
    hb = {k: torch.from_numpy(v) for k, v in synth.make_buffers(4, P.HEAD_BUFFERS).items()}
    pct, tgt, cet = torch.from_numpy(pc), torch.from_numpy(tg), torch.from_numpy(cent)
    if mode == "train":
        for d in (ep, hp):
            for v in d.values():
                v.requires_grad_(True)
        state = {id(v): (torch.zeros_like(v), torch.zeros_like(v)) for d in (ep, hp) for v in d.values()}

    def one(step):
        if mode == "fwd":
            with torch.no_grad():
                lg, tpc, ft, _ = O.forward_windows(ep, eb, hp, hb, pct, tgt, cet, False, False)
                O.predictions(lg)
        else:
            lg, tpc, ft, _ = O.forward_windows(ep, eb, hp, hb, pct, tgt, cet, True, True)
            ce, reg = O.loss_terms(lg, tpc, ft)
            for d in (ep, hp):
                for v in d.values():
                    v.grad = None
            (ce + 0.001 * reg).backward()
            with torch.no_grad():
                for d in (ep, hp):
                    for v in d.values():
                        m, s = state[id(v)]
                        O.adam_step(v, v.grad, m, s, step)

    one(1)                                                # warm-up
    t0, n = time.perf_counter(), 0
    while True:
        one(n + 2)
        n += 1
        if time.perf_counter() - t0 > budget_s or n >= 20:
            break
    dt = time.perf_counter() - t0
    pts = n * Bc * N_WIN * N_POINTS
    out = dict(value=round(pts / dt, 1), unit="points/s", cores=cores, kind="port",
               sample=f"{n} x ({Bc} samples x {N_WIN} windows x {N_POINTS} pts) {mode}, oracle torch-CPU fp32, {dt:.1f} s")
    # the port timed next to the reference's own train_loop in the build container (tests/golden/time_reference.py): port / reference
    ratio_file = os.path.join(ROOT, "tests", "golden", "reference_vs_port.json")
    if os.path.exists(ratio_file):
        r = json.load(open(ratio_file))
        key = "ratio_train" if mode == "train" else "ratio_eval"
        out["reference_ratio"] = {"port_over_reference": r[key], "measured": f"build container, {r['threads']} threads, B={r['B']}, tests/golden/time_reference.py"}
    return out


def fps_leg(dev, B, steps, warmup, seed, cpu_baseline_s=0.0):
    """BASELINE.json configs[4] on one GPU: farthest-point sampling, B clouds x 8192 points -> 4096 samples
    (data_proc/sample_fps.py:23-31 sizes) and the build-defined k-NN grouping (k = 32) of those centres.
    Roofline accounting: 16 B per (candidate, round) = 12 B xyz + 4 B running minimum; the cloud is
    register-resident, so the true HBM traffic is the floor B*(N*12 + S*4) bytes (PMC: profiles/)."""
    synth = sub("synthetic")
    U = sub("utils.utils")
    N, S, K = 8192, 4096, 32
    xyz = torch.from_numpy(synth.clouds(seed, B, N)).to(dev)
    for _ in range(max(warmup, 1)):
        idx = U.fps_indices(xyz, S)
    torch.cuda.synchronize(dev)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(steps):
        idx = U.fps_indices(xyz, S)
    ev1.record()
    torch.cuda.synchronize(dev)
    fps_ms = ev0.elapsed_time(ev1) / steps                     # the kernel runs on torch's current stream
    for _ in range(2):
        U.knn_indices(xyz, idx, K)
    torch.cuda.synchronize(dev)
    ev2, ev3 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev2.record()
    for _ in range(steps):
        U.knn_indices(xyz, idx, K)
    ev3.record()
    torch.cuda.synchronize(dev)
    knn_ms = ev2.elapsed_time(ev3) / steps
    ach = float(B) * S * N * 16 / (fps_ms * 1e-3) / 1e9
    # What binds a resident cloud is the DEPENDENT ROUND on the one CU it lives on, not HBM (PMC: ~2 MB of traffic per launch against
    # 8.59 GB "algorithmic"): the roof is the VALU issue floor of a round's update -- 512 threads x 16 points, 8 packed instructions per PAIR
    # of points = 64 instructions per wave, two waves per SIMD, 4 cycles each = 512 cycles per round -- and the measured round is that
    # update (584 cycles) + the slot write and barrier (978) + the cross-wave fold (282) + the winner's coordinates (244), which are
    # latencies of a chain, not throughput (ampnet_fps_round_stamps).
    round_cyc = fps_ms * 1e-3 / (S - 1) * 2.4e9
    valu_floor = 2.0 * 8 * 8 * 4                         # waves per SIMD x pairs per thread x instructions per pair x issue cycles
    out = {"workload": f"farthest-point sampling, {B} clouds x {N} points -> {S} samples", "ms": round(fps_ms, 4),
           "selections_per_s": round(B * S / (fps_ms * 1e-3), 1), "us_per_round": round(fps_ms * 1e3 / (S - 1), 4),
           "roofline": {"bound": "valu-issue (one CU per cloud, dependent rounds)", "kernel": "fps_kernel", "achieved": round(2.4e9 / round_cyc, 1),
                        "peak": round(2.4e9 / valu_floor, 1), "unit": "rounds/s per cloud", "frac": round(valu_floor / round_cyc, 4),
                        "cycles_per_round_at_2.4GHz": round(round_cyc, 1), "valu_issue_floor_cycles": valu_floor,
                        "round_breakdown_cycles": {"update": 584, "slot_and_barrier": 978, "fold": 282, "winner_coordinates": 244,
                                                   "source": "ampnet_fps_round_stamps, thread 0, N = 8192 (DESIGN.md section 5)"},
                        "traffic": pmc_traffic_for("fps_kernel", B * N), "launch_ms": round(fps_ms, 4),
                        "hbm_accounting_8d": {"achieved": round(ach, 1), "peak": PEAK_HBM_GBPS, "unit": "GB/s", "frac": round(ach / PEAK_HBM_GBPS, 4),
                                              "note": "SURVEY 8(d): 16 B per (candidate, round); the cloud is register / LDS resident, so these bytes never "
                                                      "reach HBM (true traffic = B*(N*12+S*4)): kept for continuity, not the binding roof"},
                        "note": "throughput comes from many clouds at once (fps.many_clouds): 256 clouds fill the chip at the same time per round"},
           "knn": {"k": K, "ms": round(knn_ms, 4), "centres_per_s": round(B * S / (knn_ms * 1e-3), 1),
                   "achieved_GBps": round(float(B) * S * N * 12 / (knn_ms * 1e-3) / 1e9, 1), "traffic": pmc_traffic_for("knn_kernel", B * N),
                   "note": "build-defined exact k-NN (the reference has none); algorithmic 12 B per (candidate, centre), served from LDS"}}
    if cpu_baseline_s > 0:
        from oracle import fps_oracle
        pc = synth.clouds(seed, 1, N)[0]
        t1 = time.perf_counter()
        n = 0
        while time.perf_counter() - t1 < cpu_baseline_s and n < 8:
            fps_oracle.fps_indices_c(pc, S)
            n += 1
        dtc = time.perf_counter() - t1
        out["cpu_baseline"] = {"value": round(n * S / dtc, 1), "unit": "selections/s", "cores": 1, "kind": "port",
                               "sample": f"{n} clouds x {N} -> {S}, oracle/fps_oracle.c (scalar C), {dtc:.1f} s"}
    return out


def kmeans_leg(dev, steps):
    """The only inference-time stage of the GRU variant's test() without a number so far (test_pointnet_gru_segmen.py:135: in-situ constrained
    k-means of every file): ampnet_kmeans_balanced_f32 on one 40 000-point file, k = 18 clusters of >= 2048 points, n_init 5, max_iter 10 --
    the arguments utils.kmeans_clustering uses.  Build-defined algorithm (parity unpinned by construction: KMeansConstrained is third party)."""
    synth, U = sub("synthetic"), sub("utils.utils")
    n, k = 40000, 18
    feat = torch.from_numpy(np.ascontiguousarray(synth.uniform(8100, (n, 3), 0.0, 1.0).astype(np.float32))).to(dev)
    U.kmeans_balanced(feat, k, 2048, n, n_init=5, max_iter=10, tol=0.01, seed=0)
    torch.cuda.synchronize(dev)
    reps = max(1, min(steps, 3))
    t0 = time.perf_counter()
    for _ in range(reps):
        labels, _, inertia = U.kmeans_balanced(feat, k, 2048, n, n_init=5, max_iter=10, tol=0.01, seed=0)      # (returns the inertia: one sync per call)
    dt = (time.perf_counter() - t0) / reps
    sizes = torch.bincount(labels.long(), minlength=k)
    return {"workload": f"size-constrained k-means, {n} points x 3 features, k = {k}, size_min 2048, n_init 5, max_iter 10 (one test file of the GRU variant)",
            "ms_per_file": round(dt * 1e3, 3), "points_per_s": round(n / dt, 1), "min_cluster": int(sizes.min().item()), "max_cluster": int(sizes.max().item()),
            "inertia": round(float(inertia), 4), "note": "build-defined algorithm (DESIGN.md section 3), untuned: a bitonic sort of n k keys per iteration"}


def fps_many_leg(dev, steps, seed):
    """FPS where its throughput is (the data_proc/sample_fps.py cascade over many files, package data_proc/sample_fps.py): a chip-filling
    batch for each stage and one streaming case.  Same 16 B per (candidate, round) accounting as fps_leg."""
    synth, U = sub("synthetic"), sub("utils.utils")
    out = {}
    for tag, B, N, S in (("stage2_256x8192_to_4096", 256, 8192, 4096), ("stage1_16x16384_to_8192", 16, 16384, 8192),
                         ("stage1_256x16384_to_8192", 256, 16384, 8192), ("stream_8x32768_to_8192", 8, 32768, 8192)):
        xyz = torch.from_numpy(synth.clouds(seed, B, N)).to(dev)
        U.fps_indices(xyz, S)
        torch.cuda.synchronize(dev)
        n = max(1, min(steps, 3))
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for _ in range(n):
            U.fps_indices(xyz, S)
        ev1.record()
        torch.cuda.synchronize(dev)
        ms = ev0.elapsed_time(ev1) / n
        kern = "fps_stream_kernel" if N > 16384 else "fps_kernel"
        out[tag] = {"clouds": B, "points": N, "samples": S, "ms": round(ms, 3), "selections_per_s": round(B * S / (ms * 1e-3), 1),
                    "us_per_round": round(ms * 1e3 / (S - 1), 4), "achieved_GBps_algorithmic": round(float(B) * S * N * 16 / (ms * 1e-3) / 1e9, 1),
                    "kernel": kern, "traffic": pmc_traffic_for(f"{kern}:{tag}", B * N)}
        del xyz
    return out


def inference_leg(enc, att, dev, steps):
    """The inference path the reference runs (test_pointnet_att_segmen.py:127-181): batch 1, one file of 18 ragged clusters (>= 2048 points
    each, unequal) per step -- amp_test.segment_file, host side included (cluster concatenation, upload, prediction download: what a call
    costs) -- and its several-files-per-launch form (amp_test.segment_files; per-file predictions identical, tests/test_inference_gpu.py)."""
    synth, A = sub("synthetic"), sub("pointNet.amp_test")
    was = enc.training, att.training
    enc.eval(); att.eval()
    files = [synth.test_file_clusters(6000 + 50 * i, 18) for i in range(16)]
    L = sub("_lib").lib()
    out = {"workload": "synthetic test files of 18 ragged clusters (2048 .. 2447 points each), eval forward + argmax, in the headline precision mode; "
                       "*_device_metrics = as amp_test.test() runs it (labels and confusion counts on the device, one download per run)"}
    try:
        G = sub("utils.get_metrics")
        for tag, group, on_dev in (("batch1", 1, False), ("files_per_launch_4", 4, False), ("files_per_launch_16", 16, False),
                                   ("batch1_device_metrics", 1, True), ("files_per_launch_16_device_metrics", 16, True)):
            def run():
                # on_dev: what amp_test.test() does -- predictions and labels stay on the device, one confusion-count kernel per file,
                # ONE download per run; otherwise the reference's contract: every call returns its predictions and labels on the host
                counts = []
                if group == 1:
                    for cl, ce in files:
                        r = A.segment_file(enc, att, cl, ce, dev, device_outputs=on_dev)
                        if on_dev:
                            counts.append(G.confusion_device(r[0], r[1], 5))
                else:
                    for g0 in range(0, len(files), group):
                        for r in A.segment_files(enc, att, files[g0:g0 + group], dev, device_outputs=on_dev):
                            if on_dev:
                                counts.append(G.confusion_device(r[0], r[1], 5))
                if on_dev:
                    return torch.stack(counts).cpu()
            for _ in range(2):                 # both page-locked staging buffers of the call exist (and are large enough) before the clock starts
                run()
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(max(1, min(steps, 4))):
                run()
            torch.cuda.synchronize(dev)
            dt = (time.perf_counter() - t0) / max(1, min(steps, 4)) / len(files)
            pts = sum(int(c.shape[0]) for cl, _ in files for c in cl) / len(files)
            # launches per file: every ampnet kernel is counted by the event profiler's table when it is on (one pass, untimed)
            L.ampnet_profile_enable(1)
            run()
            rows = profile_read()
            L.ampnet_profile_enable(0)
            out[tag] = {"ms_per_file": round(dt * 1e3, 4), "points_per_s": round(pts / dt, 1), "points_per_file": round(pts, 1),
                        "instrumented_launches_per_file": round(sum(r["calls"] for r in rows) / len(files), 1)}
    finally:
        enc.train(was[0]); att.train(was[1])
    return out


def bench_fps(args, dev, rank, world, dist):
    """--mode fps: BASELINE.json configs[4] as the headline line.  Clouds are independent: replicas only, no collective."""
    B, S = args.batch or 16, 4096
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    leg = fps_leg(dev, B, args.steps, args.warmup, 200 + rank, 10.0 if (world == 1 and not args.no_cpu_baseline) else 0.0)
    if dist is not None:
        dist.barrier()
    ms = leg["ms"]
    if dist is not None:
        tt = torch.tensor([ms], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        ms = float(tt.item())
    if rank == 0:
        out = {"metric": "FPS selections/sec (N=8192 -> 4096)", "value": round(world * B * S / (ms * 1e-3), 1), "unit": "selections/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 4),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": leg["workload"] + " per GPU", "parallelism": f"replicas{world}"},
               "us_per_round": leg["us_per_round"], "roofline": leg["roofline"], "knn": leg["knn"]}
        if "cpu_baseline" in leg:
            out["cpu_baseline"] = leg["cpu_baseline"]
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def train_loop_inclusive(enc, att, trainer_mod, B, dev, steps):
    """PCIe-inclusive figure (never `value`): one drop-in train_loop call per step on a HOST batch exactly as collate_seq_padd
    returns it -- upload (pageable memory), device augmentation kernel, the fused step, download of predictions and targets
    (train_pointnet-attention.py:337-475 end to end)."""
    synth, S = sub("synthetic"), sub("pointNet.amp_step")
    pc, tg, cent, _ = synth.sample_batch(300, B, N_POINTS, max_w=N_WIN)
    opt_p, opt_a = trainer_mod.FusedAdam(enc.parameters(), lr=1e-3), trainer_mod.FusedAdam(att.parameters(), lr=1e-3)
    ce = torch.nn.CrossEntropyLoss(weight=torch.FloatTensor([1, 2, 2, 1, 1]), reduction="mean", ignore_index=-1)
    out = {"note": "host batch -> upload + ampnet_augment_f32 + fused step + preds/targets download, per train_loop call; informational"}
    for tag, pin in (("pageable", False), ("pinned", True)):     # pinned = what DataLoader(pin_memory=True) hands to train_loop (amp_train.py)
        data = (torch.from_numpy(pc), torch.from_numpy(tg), ["f"] * B, torch.from_numpy(cent))
        if pin:
            data = (data[0].pin_memory(), data[1].pin_memory(), data[2], data[3])
        np.random.seed(0)
        for _ in range(2):
            S.train_loop(data, opt_p, opt_a, ce, enc, att, None, "segmentation", True, 0, 0)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(steps):
            S.train_loop(data, opt_p, opt_a, ce, enc, att, None, "segmentation", True, 0, 0)
        torch.cuda.synchronize(dev)
        dt = (time.perf_counter() - t0) / steps
        out[tag] = {"ms_per_step": round(dt * 1e3, 4), "points_per_s": round(B * N_WIN * N_POINTS / dt, 1)}
    # what the training driver does (amp_train._epoch): the next batch uploads on a copy stream while the current step computes
    P = sub("pointNet.prefetch")
    pinned = (torch.from_numpy(pc).pin_memory(), torch.from_numpy(tg).pin_memory(), ["f"] * B, torch.from_numpy(cent))
    np.random.seed(0)
    for data in P.DevicePrefetcher([pinned] * 3, dev):
        S.train_loop(data, opt_p, opt_a, ce, enc, att, None, "segmentation", True, 0, 0)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for data in P.DevicePrefetcher([pinned] * steps, dev):
        S.train_loop(data, opt_p, opt_a, ce, enc, att, None, "segmentation", True, 0, 0)
    torch.cuda.synchronize(dev)
    dt = (time.perf_counter() - t0) / steps
    out["prefetched"] = {"ms_per_step": round(dt * 1e3, 4), "points_per_s": round(B * N_WIN * N_POINTS / dt, 1)}
    # the package's epoch loop (amp_train._epoch): prefetched upload, predictions stay on the device, one confusion-count kernel per
    # step, ONE download per epoch -- no host synchronisation inside the loop
    G = sub("utils.get_metrics")
    for rep in range(2):
        counts = []
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for data in P.DevicePrefetcher([pinned] * steps, dev):
            m, tgt, prd, _ = S.train_loop(data, opt_p, opt_a, ce, enc, att, None, "segmentation", True, 0, 0, device_outputs=True)
            counts.append(G.confusion_device(prd, tgt, 5))
        host = torch.stack(counts).cpu()
        dt = (time.perf_counter() - t0) / steps
    out["device_metrics"] = {"ms_per_step": round(dt * 1e3, 4), "points_per_s": round(B * N_WIN * N_POINTS / dt, 1),
                             "accuracy_last_step": round(G.metrics_from_confusion(host[-1].numpy(), 5)[0], 4)}
    return out


def train_att_epoch_leg(enc, att, trainer_mod, B, dev, resident_ms, n_distinct=64, n_files=3072, workers=(4, 8)):
    """Training throughput THROUGH THE REAL LOADER (never `value`): one epoch of the package's epoch loop (amp_train._epoch, what
    train_att runs: train_pointnet-attention.py:95-106, 216-217) over a synthetic dataset in the reference's on-disk format --
    DataLoader workers running LidarKmeansDataset.__getitem__ (torch.load of kmeans_<name>.pt, noise-row removal, label mapping,
    centroids) + collate_seq_padd (resampling to 2048 points, cluster padding to 9), pin_memory, prefetched upload, device-side
    augmentation, the fused step.  n_files names point at n_distinct files of 2048 points x 1..9 clusters (a warm page cache, as in
    every epoch after the first); steady-state ms per step = time between the first and the last batch handed to the step."""
    import shutil
    import tempfile
    synth, A = sub("synthetic"), sub("pointNet.amp_train")
    D, C = sub("pointNet.datasets"), sub("pointNet.collate_fns")
    root = tempfile.mkdtemp(prefix="ampnet_epoch_")
    out = {"note": "amp_train._epoch over DataLoader(LidarKmeansDataset, collate_seq_ragged | collate_seq_padd, pin_memory) + DevicePrefetcher; files on local disk, warm cache",
           "files": n_files, "distinct_files": n_distinct, "batch": B, "steps_per_epoch": n_files // B, "resident_input_ms_per_step": round(resident_ms, 4)}
    try:
        paths = synth.write_dataset(root, n_train=n_distinct, n_val=0, n_test=0, n_points=N_POINTS, seed=7000, max_w=N_WIN)
        base = open(os.path.join(paths["lists"], "train_seg_files.txt")).read().split()
        names = []
        for i in range(n_files):
            src = base[i % n_distinct]
            if i < n_distinct:
                names.append(src)
                continue
            dst = f"rep{i}_{src}"
            os.symlink(os.path.join(paths["data"], "kmeans_" + src), os.path.join(paths["data"], "kmeans_" + dst))
            names.append(dst)
        ce = torch.nn.CrossEntropyLoss(weight=torch.FloatTensor([1, 2, 2, 1, 1]).to(dev), reduction="mean", ignore_index=-1)
        opt_p, opt_a = trainer_mod.FusedAdam(enc.parameters(), lr=1e-3), trainer_mod.FusedAdam(att.parameters(), lr=1e-3)
        ds_eager = D.LidarKmeansDataset(paths["data"], task="segmentation", number_of_points=N_POINTS, files=names)
        ds_lazy = D.LidarKmeansDataset(paths["data"], task="segmentation", number_of_points=N_POINTS, files=names, lazy=True)      # what train_att builds
        out["host_loader"] = "libampnet_host.so" if sub("_hostlib").lib() is not None else "numpy (library not built)"
        U = sub("utils.utils")
        cpus = U.host_cpu_budget()                     # affinity capped by the cgroup quota (the GPU box: a 16-CPU share of a 256-thread host)
        out["host_cpus"] = cpus
        threads_before = torch.get_num_threads()

        class Stamped:                       # the loader, with the time each batch was handed out
            def __init__(self, loader):
                self.loader, self.t = loader, []

            def __len__(self):
                return len(self.loader)

            def __iter__(self):
                half = len(self.loader) // 2
                trace = os.environ.get("AMPNET_BENCH_STALL_TRACE") == "1"       # where is the host when a step takes over a second?
                if trace:
                    import faulthandler
                for i, b in enumerate(self.loader):
                    if i == half:
                        torch.cuda.synchronize(dev)          # the host runs ahead of the GPU: drain it once so that the second half is timed exactly
                    self.t.append(time.perf_counter())
                    if trace:
                        faulthandler.dump_traceback_later(1.0, exit=False)
                    yield b
                    if trace:
                        faulthandler.cancel_dump_traceback_later()

        # ragged = the package's loader (collate_seq_ragged: the reference's random draws, resampling / padding inside the augmentation
        # kernel, 20 MB per batch); padded = the reference's collate_seq_padd in the workers (51 MB per batch), one worker count for comparison
        # "ragged" is what amp_train.train_att builds: lazy samples, read / filtered / relabelled by libampnet_host.so inside the collate;
        # "ragged_eager_samples" = the same collate on samples __getitem__ returned as arrays (the reference's Dataset contract)
        for nw, kind in [(w, "ragged") for w in workers] + [(4, "ragged_eager_samples"), (4, "padded")]:
            if nw > cpus:
                continue
            # persistent workers, as train_att builds them; this process already holds page-locked memory, so the fork stalls the GPU queues
            # for seconds (amp_train.start_workers) -- the warm-up epoch takes that, the timed epoch is the second one
            loader = Stamped(torch.utils.data.DataLoader(ds_lazy if kind == "ragged" else ds_eager, batch_size=B, shuffle=True, num_workers=nw, drop_last=True,
                                                         collate_fn=C.collate_seq_padd if kind == "padded" else C.collate_seq_ragged, pin_memory=True,
                                                         persistent_workers=True))
            np.random.seed(0)
            torch.set_num_threads(max(1, min(threads_before, cpus - nw)))      # as train_att does (limit_host_threads): the pool next to nw workers
            pin_s = []                                   # the loader's pin thread: seconds per RaggedBatch.pin_memory call (first-use page-locking shows here)
            orig_pin = C.RaggedBatch.pin_memory

            def timed_pin(self, _o=orig_pin, _l=pin_s):
                t_ = time.perf_counter()
                r = _o(self)
                _l.append(time.perf_counter() - t_)
                return r
            C.RaggedBatch.pin_memory = timed_pin
            torch.cuda.synchronize(dev)
            tw = time.perf_counter()
            A._epoch(loader, True, enc, att, opt_p, opt_a, ce, 0)             # warm-up epoch: workers fork, first-use allocations
            torch.cuda.synchronize(dev)
            warm_s = time.perf_counter() - tw
            loader.t.clear()
            pin_s.clear()
            ms0 = torch.cuda.memory_stats(dev)
            t0 = time.perf_counter()
            m = A._epoch(loader, True, enc, att, opt_p, opt_a, ce, 1)
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            C.RaggedBatch.pin_memory = orig_pin
            n = len(loader.t)
            # steady state: the second half of the epoch, from a drained GPU at its first batch to a drained GPU after its last (the first half
            # holds the workers' start-up and the first-use allocations of the loop)
            h = n // 2
            steady = (t1 - loader.t[h]) / (n - h) if n > 1 else float("nan")
            whole = (t1 - t0) / n if n else float("nan")             # the timed epoch from a drained GPU to a drained GPU, worker restart included
            stamps = [round(loader.t[i] - t0, 3) for i in (0, n // 4, h, (3 * n) // 4)] + [round(t1 - t0, 3)] if n >= 4 else None
            out[f"workers_{nw}" + {"ragged": "", "padded": "_padded_collate", "ragged_eager_samples": "_eager_samples"}[kind]] = {"epoch_s": round(t1 - t0, 3), "warmup_epoch_s": round(warm_s, 3), "steps": n, "ms_per_step": round(steady * 1e3, 3),
                                    "ms_per_step_whole_epoch": round(whole * 1e3, 3),
                                    "points_per_s": round(B * N_WIN * N_POINTS / steady, 1) if n else None,
                                    "step_share_of_wall": round(resident_ms * 1e-3 / steady, 3) if n else None,
                                    "first_batch_after_s": round(loader.t[0] - t0, 3) if n else None, "handout_s_at_0_25_50_75_100_pct": stamps,
                                    "handout_gaps_over_50ms": [[i, round(loader.t[i] - loader.t[i - 1], 3)] for i in range(1, n) if loader.t[i] - loader.t[i - 1] > 0.05],
                                    "device_allocator": {k: torch.cuda.memory_stats(dev).get(k, 0) - ms0.get(k, 0) for k in ("num_device_alloc", "num_device_free", "num_alloc_retries")},
                                    "pin_thread_s": {"first_half": round(sum(pin_s[:h]), 3), "second_half": round(sum(pin_s[h:]), 3), "max_call": round(max(pin_s), 3)} if pin_s else None,
                                    "train_loss": round(float(m["loss"]), 4)}
    finally:
        shutil.rmtree(root, ignore_errors=True)
        try:
            torch.set_num_threads(threads_before)
        except Exception:
            pass
    return out


def self_check(mode, B, args, rank, world, losses, n_warm, first_terms):
    """Loss terms of the run against tests/golden/bench_pin.json (see main()); returns the `check` object of the JSON line or exits."""
    host = torch.stack(losses).double().cpu().numpy() if losses else np.zeros((0, 2))
    timed = host[n_warm:]
    if timed.size and not np.isfinite(timed).all():
        print(f"bench.py: non-finite loss in the timed region (rank {rank}): {timed.tolist()}", file=sys.stderr, flush=True)
        raise SystemExit(3)
    out = {"timed_losses_finite": True, "pinned": None}
    if mode == "train" and len(host):
        out["first_step"] = {"ce": float(host[0][0]), "reg": float(host[0][1])}
        out["last_timed_step"] = {"ce": float(host[-1][0]), "reg": float(host[-1][1])}
    pin_file = os.path.join(ROOT, "tests", "golden", "bench_pin.json")
    key = {"train": "train_B64", "fwd": "fwd_B32"}[mode]
    want_B = {"train": 64, "fwd": 32}[mode]
    # the pin is rank 0's batch on fresh modules in fp32 with per-rank BatchNorm statistics
    # (f32x3 is an fp32 path: held to the same pin with the same tolerance)
    if rank == 0 and B == want_B and args.precision in ("fp32", "f32x3") and not (world > 1 and args.sync_bn) and os.path.exists(pin_file):
        pin = json.load(open(pin_file))
        tol = float(pin.get("rel_tol", 1e-4))
        got = {"ce": float(first_terms[0][0].item())} if mode == "fwd" else dict(out["first_step"])
        bad = {k: (got[k], pin[key][k]) for k in got if k in pin[key] and not abs(got[k] - pin[key][k]) <= tol * abs(pin[key][k])}
        if bad:
            print(f"bench.py: first-step loss terms differ from the oracle's pin (tests/golden/bench_pin.json, rel tol {tol}): {bad}",
                  file=sys.stderr, flush=True)
            raise SystemExit(4)
        out["pinned"] = {"against": "tests/golden/bench_pin.json (oracle float32, same inputs)", "rel_tol": tol,
                         "rel_err": {k: abs(got[k] - pin[key][k]) / abs(pin[key][k]) for k in got if k in pin[key]}}
    return out


def syncbn_probe_child(steps):
    """Runs in a child process (`bench.py --probe syncbn`): ONE rank on the `nccl` backend (RCCL) with AMPNET_FORCE_COLLECTIVES=1, the
    BASELINE configs[2] step with and without global-batch BatchNorm.  With one rank every collective is the identity, so the difference is
    the HOST cost of the exchange path: 36 C -> Python callbacks -> torch.distributed calls per step + the loss all-reduce (no wire time)."""
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", device_id=dev)
    synth, T = sub("synthetic"), sub("trainer")
    B = 64
    pc, tg, cent, _ = synth.sample_batch(100, B, N_POINTS, max_w=N_WIN)
    x = torch.from_numpy(np.ascontiguousarray(pc.transpose(0, 3, 1, 2))).to(dev)
    t = torch.from_numpy(np.ascontiguousarray(tg.transpose(0, 2, 1))).to(dev)
    centd = torch.from_numpy(cent).to(dev)
    cw = torch.tensor([1.0, 2.0, 2.0, 1.0, 1.0], device=dev)
    out = {}
    for tag, sync in (("gradient_allreduce_only", False), ("global_batch_batchnorm", True)):
        enc, att = build_models(dev, train=True)
        tr = T.Trainer(enc, att, lr=1e-3, class_w=cw)
        if sync:
            assert T.enable_sync_batchnorm()
        for _ in range(2):
            tr.step(x, t, centd)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(steps):
            tr.step(x, t, centd)
        torch.cuda.synchronize(dev)
        out[tag] = round((time.perf_counter() - t0) / steps * 1e3, 4)
        if sync:
            T.disable_sync_batchnorm()
    dist.barrier()
    dist.destroy_process_group()
    print(json.dumps({"probe": "syncbn", "ms_per_step": out}), flush=True)


def syncbn_probe(steps):
    """Host cost of the data-parallel exchanges on RCCL without a second GPU (the child process above; never an exec of this process)."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", AMPNET_FORCE_COLLECTIVES="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0",
               WORLD_SIZE="1", LOCAL_RANK="0")
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--probe", "syncbn", "--steps", str(steps)], env=env, capture_output=True, text=True, timeout=600)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if r.returncode != 0 or not line:
            return {"error": (r.stderr or r.stdout)[-400:]}
        ms = json.loads(line[-1])["ms_per_step"]
        return {"backend": "rccl (nccl), one rank, AMPNET_FORCE_COLLECTIVES=1: every collective is the identity, the difference is host cost",
                "ms_per_step_plain_exchange": ms["gradient_allreduce_only"], "ms_per_step_global_batch_batchnorm": ms["global_batch_batchnorm"],
                "host_ms_of_36_batchnorm_exchanges_and_loss_allreduce": round(ms["global_batch_batchnorm"] - ms["gradient_allreduce_only"], 4)}
    except Exception as e:                          # the probe is informational: never fail the bench line over it
        return {"error": repr(e)[:300]}


def self_launch(n_ranks):
    """`python bench.py --gpus N` from a plain shell: run this script as N ranks under torch.distributed.run (one process per
    GPU, rendezvous on 127.0.0.1), pass the ranks' output through (rank 0 prints the JSON line) and return the launcher's
    exit code, non-zero when any rank failed.  Returns instead of exec'ing: a process that may have touched the GPU must never
    be replaced."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC: what RCCL needs on this driver
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_ranks}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def dry_run(rank, world):
    """No GPU: every rank joins a gloo group, all-reduces (rank + 1), rank 0 prints one JSON line.  AMPNET_BENCH_FAIL_RANK=r
    makes rank r exit non-zero first (the launcher must report that)."""
    if os.environ.get("AMPNET_BENCH_FAIL_RANK") == str(rank):
        raise SystemExit(3)
    total = float(rank + 1)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("gloo")
        t = torch.tensor([total], dtype=torch.float64)
        dist.all_reduce(t)
        total = float(t.item())
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"dry_run": True, "ranks": world, "backend": "gloo", "rank_sum": total}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--mode", choices=["fwd", "train", "fps", "auto"], default="auto")
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip the informational legs (train_loop_inclusive, fps)")
    ap.add_argument("--sync-bn", action="store_true", help="N > 1: BatchNorm statistics and loss normalisation over the global batch "
                                                           "(36 more latency-bound collectives per step; default: per-rank statistics)")
    ap.add_argument("--kernels", type=int, default=8, help="how many kernels to list in the JSON line")
    ap.add_argument("--dry-run", action="store_true", help="launcher / rendezvous check only: no GPU work (tests/test_dp_cpu.py)")
    ap.add_argument("--probe", choices=["syncbn"], default=None, help="internal: child-process probes of the default run")
    ap.add_argument("--precision", choices=["fp32", "bf16", "bf16_train", "bf16_store", "f32x3"], default=os.environ.get("AMPNET_PRECISION", "f32x3"),
                    help="matrix-pipe arithmetic: f32x3 (headline since round 4: fp32 results from three-term bf16 split operands on the MFMA-bound "
                         "layers, every fp32 parity test passes in it with unchanged bars), fp32 = exact fp32 MFMA everywhere (a leg of the default line), "
                         "bf16 = forward per-point layers, bf16_train = forward + fused backward (fp32 accumulate), bf16_store = + bf16-stored activations")
    args = ap.parse_args()

    if args.probe == "syncbn":
        return syncbn_probe_child(max(args.steps, 3))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: start the N ranks ourselves.  Nothing in this process has touched the GPU yet
        # (importing torch does not initialise HIP), and the ranks are fresh children, never an exec of this process.
        raise SystemExit(self_launch(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU (or unset WORLD_SIZE to let bench.py start them)")
    if args.dry_run:
        return dry_run(rank, world)
    # one rank per GPU; AMPNET_BENCH_SHARE_GPU=1 (rehearsal on a one-GPU box, with AMPNET_DIST_BACKEND=gloo) folds the ranks onto
    # the devices that exist
    ndev = torch.cuda.device_count()
    dev_index = local_rank % ndev if os.environ.get("AMPNET_BENCH_SHARE_GPU") == "1" else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    backend = None
    if world > 1:
        import torch.distributed as dist
        backend = os.environ.get("AMPNET_DIST_BACKEND", "nccl")          # "nccl" is RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    if args.mode == "fps":
        return bench_fps(args, dev, rank, world, dist)
    if world > 1 and args.sync_bn:
        sub("trainer").enable_sync_batchnorm()

    trainer_mod = None
    try:
        trainer_mod = sub("trainer")
    except ModuleNotFoundError:
        pass
    mode = args.mode
    if mode == "auto":
        mode = "train" if trainer_mod is not None and getattr(trainer_mod, "READY", False) else "fwd"
    B = args.batch or (64 if mode == "train" else 32)

    synth = sub("synthetic")
    S = sub("pointNet.amp_step")
    sub("_lib").set_matrix_precision(args.precision)
    enc, att = build_models(dev, train=(mode == "train"))
    pc, tg, cent, _ = synth.sample_batch(100 + rank, B, N_POINTS, max_w=N_WIN)
    x = torch.from_numpy(np.ascontiguousarray(pc.transpose(0, 3, 1, 2))).to(dev)          # [B, W, N, 9] resident in HBM
    t = torch.from_numpy(np.ascontiguousarray(tg.transpose(0, 2, 1))).to(dev)
    centd = torch.from_numpy(cent).to(dev)
    cw = torch.tensor([1.0, 2.0, 2.0, 1.0, 1.0], device=dev)

    if mode == "fwd":
        def step():
            with torch.no_grad():
                return S.forward_batch(enc, att, x, t, centd, cw, want_loss=False, want_preds=True)
    else:
        tr = trainer_mod.Trainer(enc, att, lr=1e-3, class_w=cw, world_size=world)

        def step():
            return tr.step(x, t, centd)

    def sync():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize(dev)

    # Self-check (never inside the timed region's clock: device tensors are kept, read after the final synchronisation):
    #  * the FIRST step of the fresh modules must reproduce the oracle's loss terms on the same inputs (tests/golden/bench_pin.json,
    #    made by tests/golden/make_bench_pin.py and re-derived on the GPU box by tests/test_fullsize_gpu.py) within 1e-4 relative;
    #  * every timed step's ce / reg must be finite.
    # A failed check prints to stderr and exits non-zero WITHOUT a JSON line.
    losses = []
    if mode == "fwd":
        with torch.no_grad():
            first = S.forward_batch(enc, att, x, t, centd, cw, want_loss=True, want_preds=False)
        first_terms = (first["ce"][:1].clone(), None)
    else:
        first_terms = None

    def checked_step():
        out = step()
        if mode == "train":
            losses.append(torch.stack([out["ce"][0], out["reg"].reshape(-1)[0]]))
        return out

    for _ in range(args.warmup):
        checked_step()
    n_warm = len(losses)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        checked_step()
    sync()
    dt = time.perf_counter() - t0
    check = self_check(mode, B, args, rank, world, losses, n_warm, first_terms)
    if dist is not None:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    # second pass with HIP events around every instrumented kernel (rank 0 only needs it, all ranks run it)
    L = sub("_lib").lib()
    L.ampnet_profile_enable(1)
    torch.cuda.synchronize(dev)
    t1 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize(dev)
    dt_prof = time.perf_counter() - t1
    rows = profile_read()
    L.ampnet_profile_enable(0)

    fwd_ms = None
    if mode == "train":        # the metric also asks for forward ms/window: eval forward of the same batch, same modules
        enc.eval(); att.eval()
        for _ in range(2):
            with torch.no_grad():
                S.forward_batch(enc, att, x, t, centd, cw, want_loss=False, want_preds=True)
        torch.cuda.synchronize(dev)
        t2 = time.perf_counter()
        for _ in range(args.steps):
            with torch.no_grad():
                S.forward_batch(enc, att, x, t, centd, cw, want_loss=False, want_preds=True)
        torch.cuda.synchronize(dev)
        fwd_ms = (time.perf_counter() - t2) / args.steps * 1e3
        enc.train(); att.train()

    # informational legs (never the headline value): the same train step with bf16 MFMA operands in the forward per-point layers,
    # and in forward + fused backward (BASELINE.json configs[2] names bf16 MFMA; tests/test_bf16_gpu.py states what holds there)
    def precision_leg(prec, note):
        sub("_lib").set_matrix_precision(prec)
        leg_losses = []
        try:
            for _ in range(2):
                step()
            sync()
            t3 = time.perf_counter()
            for _ in range(args.steps):
                o = step()
                leg_losses.append(torch.stack([o["ce"][0], o["reg"].reshape(-1)[0]]))      # device tensors: read after the clock stops
            sync()
            dt_bf = time.perf_counter() - t3
            ll = torch.stack(leg_losses).cpu().numpy()
            if not np.isfinite(ll).all():
                print(f"bench.py: non-finite loss in the {prec} leg: {ll.tolist()}", file=sys.stderr, flush=True)
                raise SystemExit(3)
            # the leg's own roofline: HIP events around every launch for a few steps in this mode
            L.ampnet_profile_enable(1)
            for _ in range(min(args.steps, 5)):
                step()
            torch.cuda.synchronize(dev)
            rows_bf = profile_read()
            L.ampnet_profile_enable(0)
        finally:
            sub("_lib").set_matrix_precision(args.precision)
        if dist is not None:
            tt = torch.tensor([dt_bf], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt_bf = float(tt.item())
        tables = ("pmc_traffic.json", "r03_pmc_counters.json") if prec == "fp32" else (f"r03_pmc_traffic_{prec}.json", f"r03_pmc_counters_{prec}.json")
        return {"ms_per_step": round(dt_bf / args.steps * 1e3, 4), "points_per_s": round(world * B * N_WIN * N_POINTS * args.steps / dt_bf, 1), "note": note,
                "last_step": {"ce": float(ll[-1][0]), "reg": float(ll[-1][1])},
                "roofline": roofline_from(rows_bf, B * N_WIN * N_POINTS, tables, hbm_from_pmc=prec != "fp32"),
                "kernels": sorted([dict(name=r["name"], ms_per_step=round(r["ms"] / min(args.steps, 5), 4), tflops=round(r["flops"] / max(r["ms"], 1e-9) / 1e9, 1),
                                        gbps=round(r["bytes"] / max(r["ms"], 1e-9) / 1e6, 1)) for r in rows_bf], key=lambda r: -r["ms_per_step"])[:5]}

    bf16_leg = bf16_train_leg = bf16_store_leg = fp32_leg = None
    if mode == "train" and args.precision == "f32x3":
        fp32_leg = precision_leg("fp32", "the same step on exact fp32 MFMA (v_mfma_f32_32x32x2_f32) in every layer: the round-1 .. 3 headline mode")
    if mode == "train" and args.precision in ("fp32", "f32x3"):
        bf16_store_leg = precision_leg("bf16_store", "bf16_train + the activations kept for the backward (nine encoder z tensors, z2 / z3 of the head) "
                                                     "stored as bf16; inputs, outputs, gradients, parameters, statistics f32; bar: tests/test_bf16_gpu.py")
        bf16_leg = precision_leg("bf16", "forward per-point layers on v_mfma_f32_32x32x16_bf16 (bf16 operands, f32 accumulate); backward f32")
        bf16_train_leg = precision_leg("bf16_train", "forward AND fused backward of the per-point layers on bf16 MFMA operands (f32 accumulate, "
                                                     "f32 tensors in HBM, f32 BatchNorm statistics / sums); gradient bar: tests/test_bf16_gpu.py")

    # N > 1: the SAME step with BatchNorm statistics and loss normalisation over the GLOBAL batch (the reference's single-device semantics at
    # batch_per_gpu x N; 36 latency-bound exchanges per step) next to the headline's per-rank statistics, so that one multi-GPU run yields both
    sync_bn_leg = None
    if world > 1 and mode == "train" and not args.sync_bn:
        if trainer_mod.enable_sync_batchnorm():
            try:
                for _ in range(2):
                    step()
                sync()
                t5 = time.perf_counter()
                sl = []
                for _ in range(args.steps):
                    o = step()
                    sl.append(torch.stack([o["ce"][0], o["reg"].reshape(-1)[0]]))
                sync()
                dt_s = time.perf_counter() - t5
            finally:
                trainer_mod.disable_sync_batchnorm()
            tt = torch.tensor([dt_s], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt_s = float(tt.item())
            sl = torch.stack(sl).cpu().numpy()
            if not np.isfinite(sl).all():
                print(f"bench.py: non-finite loss in the global-batch BatchNorm leg (rank {rank}): {sl.tolist()}", file=sys.stderr, flush=True)
                raise SystemExit(3)
            sync_bn_leg = {"batchnorm": "global batch (statistics all-gathered / all-reduced per BatchNorm, loss over the global batch)",
                           "ms_per_step": round(dt_s / args.steps * 1e3, 4), "points_per_s": round(world * B * N_WIN * N_POINTS * args.steps / dt_s, 1),
                           "last_step": {"ce": float(sl[-1][0]), "reg": float(sl[-1][1])}}

    # the data-parallel exchange on its own: one SUM all-reduce per network over its flat gradient buffer (4.8 MB in all)
    ar_ms = None
    if dist is not None and mode == "train":
        bufs = [torch.zeros_like(b) for b in step()["grad_bufs"]]
        sync()
        t4 = time.perf_counter()
        for _ in range(args.steps):
            for b in bufs:
                dist.all_reduce(b, op=dist.ReduceOp.SUM)
        torch.cuda.synchronize(dev)
        ar_ms = (time.perf_counter() - t4) / args.steps * 1e3
    incl = fps = infer = dp_probe = epoch_leg = kmeans = None
    if mode == "train" and world == 1 and not args.no_extra_legs:
        incl = train_loop_inclusive(enc, att, trainer_mod, B, dev, max(args.steps // 2, 3))
        epoch_leg = train_att_epoch_leg(enc, att, trainer_mod, B, dev, dt / args.steps * 1e3)
        fps = fps_leg(dev, 16, max(args.steps // 2, 3), 1, 200, 0.0 if args.no_cpu_baseline else 5.0)
        fps["many_clouds"] = fps_many_leg(dev, args.steps, 210)
        infer = inference_leg(enc, att, dev, args.steps)
        kmeans = kmeans_leg(dev, args.steps)
        dp_probe = syncbn_probe(min(args.steps, 8))

    if rank == 0:
        pts_step = B * N_WIN * N_POINTS
        value = world * pts_step * args.steps / dt
        flop_pt = 413148 if mode == "fwd" else 1239444
        out = {
            "metric": "train points/sec + forward ms/window (N=2048)" if mode == "train" else "forward points/sec + forward ms/window (N=2048)",
            "value": round(value, 1), "unit": "points/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": {"fp32": "f32", "bf16": "bf16 forward MFMA operands, f32 accumulate / statistics / backward",
                                          "bf16_train": "bf16 MFMA operands (forward + fused backward), f32 accumulate / statistics / tensors",
                                          "bf16_store": "bf16 MFMA operands + bf16 stored activations, f32 accumulate / statistics / gradients",
                                          "f32x3": "f32 via 3xbf16 split operands (six exact bf16 partial products per product on the MFMA-bound layers), f32 accumulate, f32 tensors"}[args.precision], "data": "synthetic",
            "config": {"workload": ("AMP-Net full train step (fwd+loss+bwd+2xAdam)" if mode == "train" else "AMP-Net forward only (eval, logits+argmax)")
                       + f", {B} samples x {N_WIN} windows x {N_POINTS} pts x 9 feats per GPU", "batch_per_gpu": B,
                       "global_batch": B * world, "parallelism": f"dp{world}",
                       "batchnorm": ("global batch" if (world > 1 and args.sync_bn) else ("per rank" if world > 1 else "single device"))},
            "ms_per_window": round(dt / args.steps * 1e3 / (B * N_WIN), 5),
            "forward_ms_per_window": round((fwd_ms if fwd_ms is not None else dt / args.steps * 1e3) / (B * N_WIN), 5),
            "model_tflops": round(value * flop_pt / 1e12, 2),
            "ms_per_step_with_events": round(dt_prof / args.steps * 1e3, 4),
            "fp32_mfma_mode": fp32_leg, "bf16_forward_mode": bf16_leg, "bf16_train_mode": bf16_train_leg, "bf16_store_mode": bf16_store_leg,
            "ranks": world, "backend": ("rccl" if backend == "nccl" else backend), "allreduce_ms_per_step": None if ar_ms is None else round(ar_ms, 4),
            "global_batch_batchnorm_mode": sync_bn_leg,
            "train_loop_inclusive": incl, "train_att_epoch": epoch_leg, "fps": fps, "inference": infer, "kmeans": kmeans, "data_parallel_host_cost": dp_probe,
            "check": check,
            "roofline": roofline_from(rows, B * N_WIN * N_POINTS, ("r04_pmc_traffic_f32x3.json", "r04_pmc_counters_f32x3.json") if args.precision == "f32x3"
                                      else ("pmc_traffic.json", "r03_pmc_counters.json")),
            "kernels": sorted([dict(name=r["name"], ms_per_step=round(r["ms"] / args.steps, 4), launches_per_step=r["calls"] / args.steps,
                                    tflops=round(r["flops"] / max(r["ms"], 1e-9) / 1e9, 2), gbps=round(r["bytes"] / max(r["ms"], 1e-9) / 1e6, 1))
                               for r in rows], key=lambda r: -r["ms_per_step"])[:args.kernels],
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(mode)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
