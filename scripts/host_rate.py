"""PCIe-inclusive rate of the host-buffer entry points (divans_gpu_lit_encode_host / _decode_host): pageable numpy
buffers in, packed coded streams out, and back.  Reported in DESIGN.md section 5; not the bench metric."""
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, divans_amd as da, workload
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
L = 65536
corpus = workload.load_corpus()
blocks = workload.make_blocks(corpus, 0, N, block_len=L)
codec = da.LiteralCodec(da.config_simple(), L)
for rep in range(3):
    t0 = time.perf_counter(); packed, offs, sizes = codec.encode_host(blocks, L); t1 = time.perf_counter()
    back = codec.decode_host(packed, offs, sizes, L); t2 = time.perf_counter()
    assert (back == blocks).all()
    mb = N * L / 1e6
    print(f"rep {rep}: {N} x {L} B host buffers: encode {mb / (t1 - t0):.0f} MB/s, decode {mb / (t2 - t1):.0f} MB/s, "
          f"encode+decode {mb / (t2 - t0):.0f} MB/s (coded {int(sizes.sum()) / 1e6:.0f} MB)")
