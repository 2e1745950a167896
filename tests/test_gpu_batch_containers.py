"""many complete streams per call with the CMD coders / framing on host threads overlapped with the
LIT coders on the GPU (include/divans_batch.h).  Every container must be byte-identical to the per-stream ABI's and the oracle's."""
import numpy as np
import pytest

import pyoracle as po

pytestmark = pytest.mark.gpu


def test_batch_containers_equal_per_stream_abi_and_oracle(corpus):
    import divans_amd as da
    import workload
    from test_gpu_ffi import ffi_compress, ffi_decompress
    rng = np.random.default_rng(5)
    sizes = [0, 1, 15, 16, 4097, 65536, 70001, 200000] + [int(x) for x in rng.integers(1, 90000, size=24)]
    blocks = workload.make_blocks(corpus, 100, len(sizes), block_len=65536)
    inputs = [np.resize(blocks[i], s) if s <= 65536 else np.concatenate([blocks[i], corpus[:s - 65536]]) for i, s in enumerate(sizes)]
    for opts, ffi_opts, orc in ((dict(), [(5, 0)], dict()),
                                (dict(dynamic_context_mixing=2, force_stride=0, window_size=16), [(5, 0), (4, 2), (9, 0), (2, 16)],
                                 dict(dynamic_context_mixing=2, force_stride=0, window_size=16))):
        containers, timing = da.batch_compress(inputs, da.batch_options(host_threads=4, **opts))
        assert timing["total_ms"] > 0 and timing["host_overlapped_ms"] >= 0
        for i in (0, 1, 3, 4, 5, 6, 7, 11, 20, 31):
            ref = po.stream_compress_raw(inputs[i], po.stream_options(call_buffer_size=65536, **orc), call_inputs=[inputs[i].size])
            assert containers[i].size == ref.size and (containers[i] == ref).all(), (i, sizes[i])
        for i in (2, 5, 6, 9):
            assert (containers[i] == ffi_compress(inputs[i], ffi_opts)).all()
        back, t2 = da.batch_decompress(containers, sum(sizes), da.batch_options(host_threads=4))
        for i, x in enumerate(inputs):
            assert back[i].size == x.size and (back[i] == x).all(), i
        assert (ffi_decompress(containers[7], sizes[7]) == inputs[7]).all()


def test_batch_decompress_groups_configurations_and_rejects_damage(corpus):
    import divans_amd as da
    a, _ = da.batch_compress([corpus[:30000], corpus[40000:45000]], da.batch_options())
    b, _ = da.batch_compress([corpus[100:20100]], da.batch_options(dynamic_context_mixing=2, force_stride=0))
    mixed = [a[0], b[0], a[1]]
    back, _ = da.batch_decompress(mixed, 30000 + 20000 + 5000, da.batch_options())
    assert (back[0] == corpus[:30000]).all() and (back[1] == corpus[100:20100]).all() and (back[2] == corpus[40000:45000]).all()
    bad = a[0].copy(); bad[bad.size // 2] ^= 0x10
    with pytest.raises(da.DivansGpuError):
        da.batch_decompress([a[1], bad], 40000, da.batch_options())
    with pytest.raises(da.DivansGpuError):
        da.batch_decompress([a[1], bad], 40000, da.batch_options(skip_crc=1))


def test_batch_one_long_stream_among_many_short_ones(corpus):
    # length classes + slices: a 3 MB stream next to 1500 short ones neither sizes the whole batch by its length nor switches
    # the bucketed encoder off for the others; several slices per class exercise the lane pipeline
    import divans_amd as da
    import workload
    rng = np.random.default_rng(17)
    blocks = workload.make_blocks(corpus, 7, 1500, block_len=8192)
    lens = rng.integers(1, 8193, size=1500)
    inputs = [blocks[i, :lens[i]] for i in range(1500)]
    big = np.resize(corpus, 3 * 1024 * 1024 + 17)
    inputs.insert(700, big)
    inputs.insert(3, np.resize(corpus[50000:], 150000))
    containers, timing = da.batch_compress(inputs, da.batch_options(host_threads=4))
    assert timing["host_overlapped_ms"] > 0
    for i in (0, 3, 4, 699, 700, 701, 1501):
        ref = po.stream_compress_raw(inputs[i], po.stream_options(call_buffer_size=65536), call_inputs=[inputs[i].size])
        assert containers[i].size == ref.size and (containers[i] == ref).all(), i
    back, _ = da.batch_decompress(containers, sum(x.size for x in inputs), da.batch_options(host_threads=4))
    for i, x in enumerate(inputs):
        assert back[i].size == x.size and (back[i] == x).all(), i


def test_batch_decompress_names_the_damaged_stream(corpus):
    # a LIT stream whose damage passes framing (skip_crc) is caught by the decoder's integrity check, and the call says which one
    import divans_amd as da
    inputs = [corpus[k * 5000:k * 5000 + 4000 + k] for k in range(40)]
    containers, _ = da.batch_compress(inputs, da.batch_options())
    bad = [c.copy() for c in containers]
    bad[23][bad[23].size // 2] ^= 0x04
    with pytest.raises(da.DivansGpuError) as ei:
        da.batch_decompress(bad, sum(x.size for x in inputs), da.batch_options(skip_crc=1))
    assert "23" in str(ei.value)
