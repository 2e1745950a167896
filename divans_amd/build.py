"""Builds divans_amd/libdivans_hip.so (HIP kernels + C ABI) in-tree with hipcc for gfx950.

Every source is compiled to its own object (in parallel, only when it or a header changed) under divans_amd/build/, then
linked; the objects stay out of history and off the GPU box's critical path (only the .so is loaded)."""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "libdivans_hip.so")
SOURCES = ["lit_kernels.hip", "lit_decode2.hip", "lit_bucket.hip", "lit_bucket_mix.hip", "capi.cpp", "host_stream.cpp",
           "ffi.cpp", "ir.cpp", "batch.cpp"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
# The decoders that lost their measurements (generation 4 = lit_decode_t.hip, generation 1 with unified / split caches; lit_kernels.h)
# are compiled only on request: DIVANS_WITH_EXPERIMENTAL_DECODERS=1 python divans_amd/build.py --force
EXPERIMENTAL = os.environ.get("DIVANS_WITH_EXPERIMENTAL_DECODERS", "0") not in ("", "0")
if EXPERIMENTAL:
    SOURCES.insert(2, "lit_decode_t.hip")
    FLAGS.append("-DDIVANS_WITH_EXPERIMENTAL_DECODERS=1")


def hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the divans HIP extension cannot be built")


def _headers():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hs += [os.path.join(HERE, "..", "include", h) for h in ("divans_gpu.h", "divans_ffi.h", "divans_ir.h", "divans_batch.h")]
    return [h for h in hs if os.path.exists(h)]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _flags_now():
    return " ".join(FLAGS + os.environ.get("DIVANS_EXTRA_HIPCC_FLAGS", "").split())


def is_stale():
    """sources or headers newer than the library -- or the library was built with other flags (e.g. the other value of
    DIVANS_WITH_EXPERIMENTAL_DECODERS): the flavour on disk must be the one asked for"""
    tag = os.path.join(OBJ, "flags.txt")
    if not os.path.exists(tag) or open(tag).read() != _flags_now():
        return True
    return _stale(LIB, [os.path.join(CSRC, s) for s in SOURCES] + _headers())


def build(force=False, verbose=False):
    if not force and not is_stale():
        return LIB
    os.makedirs(OBJ, exist_ok=True)
    cc = hipcc()
    extra = os.environ.get("DIVANS_EXTRA_HIPCC_FLAGS", "").split()
    headers = _headers()
    flags_tag = os.path.join(OBJ, "flags.txt")
    flags_now = _flags_now()
    if not os.path.exists(flags_tag) or open(flags_tag).read() != flags_now:
        force = True

    def compile_one(src):
        path = os.path.join(CSRC, src)
        obj = os.path.join(OBJ, src + ".o")
        if not force and not _stale(obj, [path] + headers):
            return obj, None
        res = subprocess.run([cc] + FLAGS + extra + ["-x", "hip", "-c", path, "-o", obj], capture_output=True, text=True)
        if res.returncode != 0:
            return obj, res.stdout + res.stderr
        return obj, res.stderr if verbose else None

    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 4)) as ex:
        results = list(ex.map(compile_one, SOURCES))
    objs = []
    for obj, msg in results:
        if msg:
            sys.stderr.write(msg)
        if not os.path.exists(obj) or (msg and "error" in msg):
            raise RuntimeError("hipcc failed building libdivans_hip.so")
        objs.append(obj)
    res = subprocess.run([cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs, capture_output=True, text=True)
    if res.returncode != 0:
        sys.stderr.write(res.stdout + res.stderr)
        raise RuntimeError("hipcc failed linking libdivans_hip.so")
    with open(flags_tag, "w") as f:
        f.write(flags_now)
    return LIB


def check_experimental():
    """Compile check of the sources only experiment builds link (lit_decode_t.hip): object only, kept under build/."""
    if EXPERIMENTAL:
        return
    os.makedirs(OBJ, exist_ok=True)
    path = os.path.join(CSRC, "lit_decode_t.hip")
    obj = os.path.join(OBJ, "lit_decode_t.hip.check.o")
    if not _stale(obj, [path] + _headers()):
        return
    res = subprocess.run([hipcc()] + FLAGS + ["-DDIVANS_WITH_EXPERIMENTAL_DECODERS=1", "-x", "hip", "-c", path, "-o", obj], capture_output=True, text=True)
    if res.returncode != 0:
        sys.stderr.write(res.stdout + res.stderr)
        raise RuntimeError("hipcc failed compiling lit_decode_t.hip (experiment-only source)")


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
