"""Compile the MFMA kernels' sources to ISA and report, per kernel, scratch (spill) reloads that sit INSIDE the span of its MFMAs, conservative
`s_waitcnt vmcnt(0)` waits there and the hazard no-ops -- a reload in a k loop waits for every global load in flight (scratch counts on vmcnt),
which is how the split forward's store variant lost 20 % before round 4 found it.
    python3 tools/scan_spills.py [file.hip ...]      (default: pw_gemm, pw_bwd_fused, pw_bwd_bf16, pw_bwd_x3; ~5 minutes of hipcc)"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "3d-semantic-segmentation-amp-net_amd", "csrc")
files = sys.argv[1:] or ["pw_gemm.hip", "pw_bwd_fused.hip", "pw_bwd_bf16.hip", "pw_bwd_x3.hip"]
tmp = tempfile.mkdtemp(prefix="ampnet_isa_")
procs = []
for f in files:
    out = os.path.join(tmp, os.path.basename(f)[:-4] + ".s")
    procs.append((f, out, subprocess.Popen(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", out,
                                            os.path.join(CSRC, f)], stderr=subprocess.DEVNULL)))
bad = 0
for f, out, p in procs:
    if p.wait() != 0:
        print(f"{f}: hipcc failed")
        continue
    L = open(out).read().split("\n")
    starts = [i for i, l in enumerate(L) if re.match(r"^_ZN6ampnet\S*:\s", l + " ")]
    n = 0
    for s0 in starts:
        e = next((j for j in range(s0, len(L)) if L[j].startswith(".Lfunc_end")), len(L))      # (a kernel may hold several s_endpgm)
        body = L[s0:e]
        idx = [j for j, l in enumerate(body) if "v_mfma" in l]
        if not idx:
            continue
        n += 1
        a, b = idx[0], idx[-1]
        span = body[a:b + 1]
        reloads = sum(1 for l in span if "scratch_load" in l)
        total = sum(1 for l in body if "scratch_" in l)
        w0 = sum(1 for l in span if "vmcnt(0)" in l)
        loads = sum(1 for l in span if re.search(r"\b(global_load|flat_load|buffer_load)", l))
        nops = sum(1 for l in span if "s_nop" in l)
        flat = sum(1 for l in span if "flat_load" in l)
        if reloads or flat or (total and "-v" in sys.argv):
            bad += 1
            print(f"{f}: {L[s0].split(':')[0][14:110]}: {len(idx)} MFMAs, {reloads} scratch reloads in their span ({total} scratch ops in all), "
                  f"{loads} loads / {w0} vmcnt(0) waits, {flat} FLAT loads, {nops} s_nop")
    print(f"{f}: {n} MFMA kernels scanned")
print("kernels with reloads or flat loads inside the MFMA span:", bad)
