"""Builds libampnet_hip.so (all HIP kernels + the C ABI) for gfx950 with hipcc, in-tree.

hipcc cross-compiles without a GPU.  Each csrc/*.hip is compiled to an object (only fps.hip needs
-ffp-contract=off for bit parity) and the objects are linked into <package>/libampnet_hip.so,
which travels to the GPU box with the snapshot.
"""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "csrc", "_obj")
LIB = os.path.join(HERE, "libampnet_hip.so")
ARCH = "gfx950"
COMMON = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
PER_FILE = {"fps.hip": ["-ffp-contract=off"], "knn.hip": ["-ffp-contract=off"], "augment.hip": ["-ffp-contract=off"], "kmeans.hip": ["-ffp-contract=off"]}


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _stamp(src, flags):
    h = hashlib.sha1()
    h.update(" ".join(flags).encode())
    for f in [src] + sorted(os.path.join(CSRC, x) for x in os.listdir(CSRC) if x.endswith(".h")) + \
            [os.path.join(HERE, "..", "include", "ampnet_hip.h")]:
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


HOST_LIB = os.path.join(HERE, "libampnet_host.so")
HOST_SRC = [os.path.join(CSRC, "host", "sample_loader.cpp")]


def build_host(verbose=True, force=False):
    """libampnet_host.so (include/ampnet_host.h): the host side of the input pipeline, plain g++ -- no HIP, loadable in DataLoader workers.
    -ffp-contract=off: its float32 roundings are numpy's."""
    hdr = os.path.join(HERE, "..", "include", "ampnet_host.h")
    newest = max(os.path.getmtime(f) for f in HOST_SRC + [hdr])
    if not force and os.path.exists(HOST_LIB) and os.path.getmtime(HOST_LIB) >= newest:
        return HOST_LIB
    cmd = [os.environ.get("CXX", "g++"), "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-Wall", "-o", HOST_LIB] + HOST_SRC
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return HOST_LIB


def build(verbose=True, force=False):
    os.makedirs(OBJ, exist_ok=True)
    build_host(verbose, force)
    hipcc = _hipcc()
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    objs, rebuilt = [], False
    procs = []
    for f in srcs:
        src = os.path.join(CSRC, f)
        obj = os.path.join(OBJ, f[:-4] + ".o")
        flags = COMMON + PER_FILE.get(f, [])
        stamp_file = obj + ".stamp"
        stamp = _stamp(src, flags)
        objs.append(obj)
        if not force and os.path.exists(obj) and os.path.exists(stamp_file) and open(stamp_file).read() == stamp:
            continue
        cmd = [hipcc] + flags + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True), f, stamp_file, stamp))
        rebuilt = True
    for p, f, stamp_file, stamp in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {f}:\n{out}")
        if verbose and out.strip():
            print(out)
        with open(stamp_file, "w") as fh:
            fh.write(stamp)
    if rebuilt or not os.path.exists(LIB):
        cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
