"""Driver of tests/tools/sim_hierarchy.c (design aid, not product): which rows of a stream's CDF table share a 128-byte line,
replayed against LDS row caches + one XCD's L2 + its share of the Infinity Cache, for the decoders' real access streams.

    python tests/tools/sim_hierarchy.py [--streams 3584] [--config plain|mixing|both] [--data text|binary] [--layouts 0,1,...]

Writes the blocks (tests/workload.py: the benchmark's streams, or random_then_unicode for --data binary) and -- for the mixing
configuration -- the fused context table (oracle luts, TestContextMixing: cm[i] = i & 63, UTF8) into /tmp, builds the simulator with
gcc and runs the layouts in parallel.  VERDICT r05 item 1."""
import argparse
import ctypes
import lzma
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import workload  # noqa: E402


def ctxf_file(path):
    import pyoracle as po
    L = po.lib()
    l0 = (ctypes.c_uint8 * 256)(); l1 = (ctypes.c_uint8 * 256)()
    L.orc_get_lut0(2, l0); L.orc_get_lut1(2, l1)
    l0 = np.array(l0, dtype=np.int64); l1 = np.array(l1, dtype=np.int64)
    vals = []
    cls = np.zeros(256, dtype=np.uint8)
    for b in range(256):
        if l1[b] not in vals:
            vals.append(int(l1[b]))
        cls[b] = vals.index(int(l1[b]))
    assert len(vals) <= 4
    ctxf = np.zeros((256, 4), dtype=np.uint16)
    for prev in range(256):
        used = 0
        ctxs = []
        slots = []
        for k in range(len(vals)):
            c = int((l0[prev] | vals[k]) & 63)     # cm[i] = i & 63
            if c in ctxs:
                slot = slots[ctxs.index(c)]
            else:
                slot = used; used += 1
            ctxs.append(c); slots.append(slot)
            ctxf[prev, k] = c | (slot << 8)
    with open(path, "wb") as f:
        f.write(ctxf.tobytes()); f.write(cls.tobytes())


def blocks_file(path, n, kind):
    if kind == "text":
        corpus = workload.load_corpus()
        blocks = workload.make_blocks(corpus, 0, n)
    else:
        with lzma.open(os.path.join(workload.GOLDEN, "random_then_unicode.xz")) as f:
            raw = np.frombuffer(f.read(), dtype=np.uint8)
        whole = raw[: (raw.size // 65536) * 65536].reshape(-1, 65536)
        blocks = whole[np.arange(n) % whole.shape[0]]
    np.ascontiguousarray(blocks).tofile(path)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=3584)
    ap.add_argument("--config", default="plain")
    ap.add_argument("--data", default="text")
    ap.add_argument("--layouts", default="0,1,2,3,4,5,6,7,8,9")
    ap.add_argument("--extra", default="", help="space-separated key=value passed to every run")
    ap.add_argument("--jobs", type=int, default=6)
    a = ap.parse_args()
    exe = "/tmp/sim_hierarchy"
    subprocess.check_call(["gcc", "-O2", "-o", exe, os.path.join(ROOT, "tests/tools/sim_hierarchy.c")])
    bpath = f"/tmp/sim_blocks_{a.data}_{a.streams}.bin"
    if not os.path.exists(bpath):
        blocks_file(bpath, a.streams, a.data)
    cpath = "/tmp/sim_ctxf.bin"
    ctxf_file(cpath)
    runs = []
    for cfg in (["plain", "mixing"] if a.config == "both" else [a.config]):
        for lay in a.layouts.split(","):
            cmd = [exe, bpath, str(a.streams), "65536", "1" if cfg == "mixing" else "0", lay, f"ctxf={cpath}"] + a.extra.split()
            runs.append(cmd)
    with ThreadPoolExecutor(a.jobs) as ex:
        for out in ex.map(lambda c: subprocess.run(c, capture_output=True, text=True).stdout, runs):
            print(out, flush=True)


if __name__ == "__main__":
    main()
