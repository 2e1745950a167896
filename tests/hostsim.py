"""Builds the host-logic test harness of tests/c: the product's per-stream host code (divans_amd/csrc/host_stream.cpp, ffi.cpp)
linked against tests/c/hostsim_device_stub.cpp, which answers the nine GPU entry points that code uses with the CPU oracle.
TEST INFRASTRUCTURE: nothing in divans_amd/ knows of it; it exists so that the call-by-call container logic and the parser of
untrusted input run in the "not gpu" tier, and under AddressSanitizer / UBSan."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "c", "_build")
# HOSTSIM_CXX / HOSTSIM_CC select the compilers (default g++ / gcc; /opt/rocm/lib/llvm/bin/clang++ is the host compiler hipcc uses for the product)
CXX = os.environ.get("HOSTSIM_CXX", "g++")
CC = os.environ.get("HOSTSIM_CC", "gcc")

CXX_SOURCES = ["divans_amd/csrc/host_stream.cpp", "divans_amd/csrc/ffi.cpp", "divans_amd/csrc/ir.cpp", "tests/c/hostsim_device_stub.cpp"]
C_SOURCES = ["oracle/cdf.c", "oracle/ans.c", "oracle/literal.c", "oracle/crc32c.c", "oracle/stream.c"]


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    deps = list(sources) + [os.path.join(ROOT, "divans_amd", "csrc", "host_stream.h"), os.path.join(ROOT, "oracle", "divans_oracle.h"),
                            os.path.join(ROOT, "include", "divans_gpu.h"), os.path.join(ROOT, "include", "divans_ffi.h"), os.path.join(ROOT, "include", "divans_ir.h"), os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def _objects(tag, flags):
    os.makedirs(OUT, exist_ok=True)
    objs = []
    for src in CXX_SOURCES + C_SOURCES:
        path = os.path.join(ROOT, src)
        obj = os.path.join(OUT, tag + "_" + os.path.basename(src) + ".o")
        if _newer(obj, [path]):
            cc = [CXX, "-std=c++17"] if src.endswith(".cpp") else [CC, "-std=c11"]
            subprocess.run(cc + flags + ["-fPIC", "-c", path, "-o", obj, "-I" + os.path.join(ROOT, "include")], check=True)
        objs.append(obj)
    return objs


def build_library():
    """tests/c/_build/libdivans_hostsim.so: the per-stream C ABI over the oracle-backed device stub, for ctypes."""
    lib = os.path.join(OUT, "libdivans_hostsim.so")
    objs = _objects("so", ["-O2", "-g", "-msse4.2"])
    if _newer(lib, objs):
        subprocess.run([CXX, "-shared", "-Wl,-Bsymbolic", "-o", lib] + objs + ["-lpthread"], check=True)   # its own divans_* symbols, whatever else the process has loaded
    return lib


def build_fuzzer():
    """tests/c/_build/hostsim_fuzz: tests/c/hostsim_fuzz.cpp with the same objects, everything under -fsanitize=address,undefined."""
    exe = os.path.join(OUT, "hostsim_fuzz")
    san = ["-O1", "-g", "-msse4.2", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer"]
    objs = _objects("san", san)
    driver = os.path.join(ROOT, "tests", "c", "hostsim_fuzz.cpp")
    if _newer(exe, objs + [driver]):
        subprocess.run([CXX, "-std=c++17"] + san + ["-o", exe, driver] + objs + ["-I" + os.path.join(ROOT, "include"), "-lpthread"], check=True)
    return exe


def build_program(name, source, lang="c++", sanitize_main=True):
    """tests/c/_build/<name>: `source` (a caller of the per-stream C ABI) linked with the sanitized harness objects;
    sanitize_main=False leaves the caller itself uninstrumented (the reference's c/example.c trips UBSan in its own vec_u8.h)."""
    exe = os.path.join(OUT, name)
    san = ["-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer"]
    objs = _objects("san", san + ["-msse4.2"])
    if _newer(exe, objs + [source]):
        obj = os.path.join(OUT, name + ".main.o")
        cc = [CXX, "-std=c++14"] if lang == "c++" else [CC, "-Wno-unused-result"]
        subprocess.run(cc + (san if sanitize_main else ["-O1", "-g"]) + ["-c", source, "-o", obj, "-I" + os.path.join(ROOT, "include"), "-I" + os.path.dirname(source)], check=True)
        subprocess.run([CXX] + san + ["-o", exe, obj] + objs + ["-lpthread"], check=True)
    return exe


def build_thread_test():
    """tests/c/_build/hostsim_threads: tests/c/hostsim_threads.cpp and the harness objects under -fsanitize=thread."""
    exe = os.path.join(OUT, "hostsim_threads")
    san = ["-O1", "-g", "-msse4.2", "-fsanitize=thread", "-fno-omit-frame-pointer"]
    objs = _objects("tsan", san)
    driver = os.path.join(ROOT, "tests", "c", "hostsim_threads.cpp")
    if _newer(exe, objs + [driver]):
        subprocess.run([CXX, "-std=c++17"] + san + ["-o", exe, driver] + objs + ["-I" + os.path.join(ROOT, "include"), "-lpthread"], check=True)
    return exe


BATCH_SOURCE = "divans_amd/csrc/batch.cpp"


def build_batch_test(sanitizer="address,undefined"):
    """tests/c/_build/hostsim_batch_<tag>: tests/c/hostsim_batch.cpp + divans_amd/csrc/batch.cpp (compiled by g++ against
    tests/c/fakehip's stand-in for <hip/hip_runtime.h>) + the harness objects, under the given sanitizer."""
    tag = "tsan" if sanitizer == "thread" else "san"
    exe = os.path.join(OUT, "hostsim_batch_" + tag)
    san = ["-O1", "-g", "-msse4.2", "-fsanitize=" + sanitizer, "-fno-omit-frame-pointer"] + (["-fno-sanitize-recover=undefined"] if tag == "san" else [])
    objs = _objects(tag, san)
    batch_src = os.path.join(ROOT, BATCH_SOURCE)
    fake = os.path.join(ROOT, "tests", "c", "fakehip")
    batch_obj = os.path.join(OUT, tag + "_batch.cpp.o")
    if _newer(batch_obj, [batch_src, os.path.join(fake, "hip", "hip_runtime.h"), os.path.join(ROOT, "include", "divans_batch.h")]):
        subprocess.run([CXX, "-std=c++17"] + san + ["-fPIC", "-c", batch_src, "-o", batch_obj, "-I" + fake, "-I" + os.path.join(ROOT, "include")], check=True)
    driver = os.path.join(ROOT, "tests", "c", "hostsim_batch.cpp")
    if _newer(exe, objs + [batch_obj, driver]):
        subprocess.run([CXX, "-std=c++17"] + san + ["-o", exe, driver, batch_obj] + objs + ["-I" + os.path.join(ROOT, "include"), "-lpthread"], check=True)
    return exe
