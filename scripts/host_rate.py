"""PCIe-inclusive rate of the host-buffer entry points (divans_gpu_lit_encode_host / _decode_host) called through the
C ABI with caller buffers that already exist (touched), pageable vs pinned.  Reported in DESIGN.md section 5; not the
bench metric."""
import ctypes, sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch, divans_amd as da, workload
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
L = 65536
corpus = workload.load_corpus()
blocks = workload.make_blocks(corpus, 0, N, block_len=L)
codec = da.LiteralCodec(da.config_simple(), L)
lib = da.load_library()
cap = da.encode_bound(L) * N // 2 + 64          # text codes to < half the worst case


def buffers(pinned):
    mk = (lambda n: torch.empty(n, dtype=torch.uint8, pin_memory=True).numpy()) if pinned else (lambda n: np.zeros(n, dtype=np.uint8))
    src = mk(N * L); src[:] = blocks.reshape(-1)
    coded = mk(cap); coded[:] = 0
    back = mk(N * L); back[:] = 0
    return src, coded, back


for pinned in (False, True):
    src, coded, back = buffers(pinned)
    offs = np.zeros(N, dtype=np.uint64); sizes = np.zeros(N, dtype=np.uint32); total = ctypes.c_size_t(0)
    for rep in range(3):
        t0 = time.perf_counter()
        rc = lib.divans_gpu_lit_encode_host(codec._h, src.ctypes.data, L, N, coded.ctypes.data, cap, offs.ctypes.data, sizes.ctypes.data, ctypes.byref(total))
        t1 = time.perf_counter()
        assert rc == 0, lib.divans_gpu_last_error()
        rc = lib.divans_gpu_lit_decode_host(codec._h, coded.ctypes.data, offs.ctypes.data, sizes.ctypes.data, N, back.ctypes.data, L)
        t2 = time.perf_counter()
        assert rc == 0 and (back == src).all()
        mb = N * L / 1e6
        print(f"{'pinned  ' if pinned else 'pageable'} rep {rep}: {N} x {L} B: encode {mb / (t1 - t0):.0f} MB/s, decode {mb / (t2 - t1):.0f} MB/s, "
              f"encode+decode {mb / (t2 - t0):.0f} MB/s (coded {total.value / 1e6:.0f} MB)")
