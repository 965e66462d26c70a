#!/usr/bin/env python3
"""Turns rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs, --kernel-trace only, as MI355X_MICROARCH.md
prescribes) into HBM bytes per launch for the kernels bench.py names in its `roofline` object.

  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc/FETCH_SIZE -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc/WRITE_SIZE -- python3 bench.py ...
  python profiles/pmc_traffic.py gpurun_out/pmc/train profiles/pmc_traffic.json [rows] [--fps gpurun_out/pmc/fps]   (tools/run_pmc.sh runs the passes)

gfx950 corrections: FETCH_SIZE is in KiB and reports HALF of the bytes of wide coalesced reads (x 2); WRITE_SIZE in KiB
reads exact.  A bench.py event name such as `pw_gemm<128,128>+pool` is a subset of the launches of one kernel symbol
(`pw_gemm_kernel<128, 4>`); the subset is identified by its grid size (the pooled 128->256 launches have two column
blocks, i.e. the largest grid of that symbol)."""
import collections
import csv
import glob
import json
import sys


def per_dispatch(dirname, counter):
    f = glob.glob(f"{dirname}/{counter}/*/*counter_collection.csv")
    out = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] == counter:
            out[r["Kernel_Name"]].append((int(r["Grid_Size"]), float(r["Counter_Value"])))
    return out


FPS_CASES = {"stage2_256x8192_to_4096": (256, 8192, 4096), "stage1_16x16384_to_8192": (16, 16384, 8192),
             "stage1_256x16384_to_8192": (256, 16384, 8192), "stream_8x32768_to_8192": (8, 32768, 8192)}


def main(src, dst, rows_per_launch=64 * 9 * 2048, fps_src=None, fps_cases=None):
    fe, wr = per_dispatch(src, "FETCH_SIZE"), per_dispatch(src, "WRITE_SIZE")
    res = {}
    for sym, rows in fe.items():
        grids = sorted({g for g, _ in rows})
        for g in grids:
            rd = [v for gg, v in rows if gg == g]
            ww = [v for gg, v in wr.get(sym, []) if gg == g]
            key = f"{sym.split('(')[0].replace('void ', '')}|grid={g}"
            res[key] = {"launches": len(rd), "read_bytes": sum(rd) / len(rd) * 1024 * 2,
                        "write_bytes": (sum(ww) / len(ww) * 1024) if ww else None}
    # bench.py's event names -> the matching (symbol, grid) entry; where several grids share an event name (point layers vs
    # the tiny T-Net FC launches of the same instantiation) the largest grid is the point-layer launch the roofline is about
    import re
    events = {}
    for key, v in res.items():
        sym, grid = key.split("|grid=")
        name = None
        m = re.match(r"ampnet::pw_gemm_kernel<(\d+), (\d+), (\d+), (true|false), (true|false)[,>]", sym)   # <CIN, NT, PRO, POOL, BF, ABF, ZBF, PIPE, ARG, X3, ...>
        if m:
            x3 = re.match(r"ampnet::pw_gemm_kernel<(?:[^,]+, ){9}true", sym) is not None                    # the three-term split kernels (mode f32x3)
            name = f"pw_gemm<{m.group(1)},{32 * int(m.group(2))}>" + ("+pool" if m.group(4) == "true" else "+store") + (" x3" if x3 else (" bf16" if m.group(5) == "true" else ""))
        m = re.match(r"ampnet::pw_bwd_x3n_kernel<(\d+), (\d+), (true|false), (true|false), (true|false)>", sym)
        if m:
            name = (f"pw_bwd<{m.group(1)},{m.group(2)}>" + ("" if m.group(3) == "true" else " lin") + ("+add" if m.group(5) == "true" else "")
                    + ("+drop" if m.group(4) == "true" else "") + " x3")
        m = re.match(r"ampnet::pw_bwd_x3_kernel<(true|false)>", sym)
        if m:
            name = "pw_bwd<128,128>" + ("+gram" if m.group(1) == "true" else "") + " x3"
        m = re.match(r"ampnet::pw_bwd_kernel<(\d+), (\d+), (\d+), (true|false), (true|false), (true|false), (true|false)", sym)   # <CX, CY, ROWS, GRAM, YACT, ADD, DROP>
        if m:
            name = (f"pw_bwd<{m.group(1)},{m.group(2)}>" + ("+gram" if m.group(4) == "true" else "") + ("" if m.group(5) == "true" else " lin")
                    + ("+add" if m.group(6) == "true" else "") + ("+drop" if m.group(7) == "true" else ""))
        m = re.match(r"ampnet::pw_bwd_bf16_kernel<(\d+), (\d+), (\d+), (true|false)", sym)
        if m:
            name = f"pw_bwd<{m.group(1)},{m.group(2)}>" + ("+gram" if m.group(4) == "true" else "") + " bf16"
        if name and (name not in events or int(grid) > events[name]["grid"]):
            events[name] = dict(v, grid=int(grid), symbol=sym)
    # configs[4] kernels (tools/prof_fps.py: 16 clouds x 8192 points -> 4096 samples, k = 32): their own passes, their own workload size.
    # Their loads are 4-byte strided, a width the FETCH_SIZE x 2 correction is not calibrated for: the raw counter is kept next to it.
    if fps_src:
        ff, fw = per_dispatch(fps_src, "FETCH_SIZE"), per_dispatch(fps_src, "WRITE_SIZE")
        for sym, rows in ff.items():
            for name in ("fps_kernel", "knn_kernel"):
                if name in sym:
                    rd = [v for _, v in rows]
                    ww = [v for _, v in fw.get(sym, [])]
                    events[name] = {"launches": len(rd), "read_bytes": sum(rd) / len(rd) * 1024 * 2, "read_bytes_uncorrected": sum(rd) / len(rd) * 1024,
                                    "write_bytes": (sum(ww) / len(ww) * 1024) if ww else None, "symbol": sym.split("(")[0].replace("void ", ""),
                                    "workload_rows": 16 * 8192, "floor_bytes": {"fps_kernel": 16 * (8192 * 12 + 4096 * 4), "knn_kernel": 16 * (8192 * 12 + 4096 * 4 + 4096 * 32 * 4)}[name]}
    # bench.py fps.many_clouds: one pass directory per case, event name "<kernel>:<case>"
    for tag, d in (fps_cases or {}).items():
        ff, fw = per_dispatch(d, "FETCH_SIZE"), per_dispatch(d, "WRITE_SIZE")
        B, N, S = FPS_CASES[tag]
        for sym, rows in ff.items():
            for name in ("fps_stream_kernel", "fps_kernel"):
                if name in sym and not (name == "fps_kernel" and "fps_stream_kernel" in sym):
                    rd = [v for _, v in rows]
                    ww = [v for _, v in fw.get(sym, [])]
                    events[f"{name}:{tag}"] = {"launches": len(rd), "read_bytes": sum(rd) / len(rd) * 1024 * 2, "read_bytes_uncorrected": sum(rd) / len(rd) * 1024,
                                               "write_bytes": (sum(ww) / len(ww) * 1024) if ww else None, "symbol": sym.split("(")[0].replace("void ", ""),
                                               "workload_rows": B * N, "floor_bytes": B * (N * 12 + S * 4), "algorithmic_bytes": float(B) * S * N * 16}
    res["events"] = events
    res["workload_rows"] = rows_per_launch          # rows (points) every point-layer launch of the profiled step processes
    json.dump(res, open(dst, "w"), indent=1, sort_keys=True)
    print(f"wrote {dst}: {len(res) - 2} (kernel, grid) entries, {len(events)} bench event names")


if __name__ == "__main__":
    fps, cases = None, {}
    argv = list(sys.argv[1:])
    if "--fps" in argv:
        i = argv.index("--fps")
        fps = argv[i + 1]
        del argv[i:i + 2]
    while "--fps-case" in argv:                     # --fps-case <tag>=<dir>
        i = argv.index("--fps-case")
        tag, d = argv[i + 1].split("=", 1)
        cases[tag] = d
        del argv[i:i + 2]
    main(argv[0], argv[1], *(int(a) for a in argv[2:3]), fps_src=fps, fps_cases=cases)
