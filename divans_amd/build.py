"""Builds divans_amd/libdivans_hip.so (HIP kernels + C ABI) in-tree with hipcc for gfx950."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libdivans_hip.so")
SOURCES = ["lit_kernels.hip", "lit_kernels_p8.hip", "lit_bucket.hip", "lit_bucket_mix.hip", "capi.cpp", "host_stream.cpp", "ffi.cpp", "ir.cpp", "batch.cpp"]


def hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the divans HIP extension cannot be built")


def is_stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", h) for h in ("divans_gpu.h", "divans_ffi.h", "divans_ir.h", "divans_batch.h")]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force=False, verbose=False):
    if not force and not is_stale():
        return LIB
    cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall",
           "-Wno-unused-function", "-o", LIB] + os.environ.get("DIVANS_EXTRA_HIPCC_FLAGS", "").split() + [os.path.join(CSRC, s) for s in SOURCES]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        sys.stderr.write(res.stdout + res.stderr)
        raise RuntimeError("hipcc failed building libdivans_hip.so")
    if verbose:
        sys.stderr.write(res.stderr)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
