"""whole containers of the literal-only internal compressor from a second, independent restatement of the
Rust (tests/ref_container.py: ring emission, PredictionMode coding incl. the prior-aliasing quirk, literal-length nibbles, the
Mux, flush / EOF / CRC trailer, the caller's buffer sizes) against the C oracle, byte for byte.  CPU only.  This narrows the
risk that the oracle misreads the container; it does not pin a compressed byte to the reference build (parity stays "partial")."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE); sys.path.insert(0, os.path.join(os.path.dirname(HERE), "oracle"))
import pyoracle as po          # noqa: E402
import ref_container as rc     # noqa: E402
import workload                # noqa: E402

# the option sets tests/test_gpu_ffi.py drives the C FFI with: defaults of the FFI, no context map / no mixing, everything on,
# explicit speeds
OPTION_SETS = [
    dict(window_size=22, dynamic_context_mixing=1, use_context_map=1, force_stride=9),
    dict(window_size=10, dynamic_context_mixing=0, use_context_map=0, force_stride=0),
    dict(window_size=16, dynamic_context_mixing=2, use_context_map=1, force_stride=1, prior_depth=3),
    dict(window_size=16, dynamic_context_mixing=0, use_context_map=1, force_stride=9,
         literal_adaptation=[(64, 16384), (128, 16384), (1, 16384), (4, 1024)]),
]


def _both(data, buf, calls=None, **kw):
    pykw = dict(kw); pykw["use_context_map"] = bool(kw.get("use_context_map", 0))
    got = rc.compress(data, call_inputs=calls, call_buffer_size=buf, **pykw)
    want = bytes(po.stream_compress_raw(data, po.stream_options(call_buffer_size=buf, **kw), call_inputs=calls))
    return got, want


@pytest.fixture(scope="module")
def corpus():
    return workload.load_corpus()


@pytest.mark.parametrize("opt", range(len(OPTION_SETS)))
@pytest.mark.parametrize("buf", [65536, 4096, 17])
def test_containers_match_the_oracle(opt, buf, corpus):
    rng = np.random.default_rng(100 * opt + buf)
    for data in (corpus[2000:2000 + 7000], rng.integers(0, 256, 5000, dtype=np.uint8), np.zeros(3000, np.uint8),
                 np.frombuffer(b"x", dtype=np.uint8), np.zeros(0, np.uint8), corpus[:14], corpus[:15], corpus[:16], corpus[:17]):
        got, want = _both(data, buf, **OPTION_SETS[opt])
        assert got == want, (opt, buf, len(data))
        assert (po.stream_decompress(np.frombuffer(got, dtype=np.uint8), len(data) + 16) == data).all()


def test_more_input_than_the_window(corpus):
    # window 10 = a 1 KiB ring: the first lap emits the whole ring, later laps size-1 and split where the ring wraps
    # (raw_to_cmd/mod.rs:55-181); literal lengths of 1, 1022, 1023, 1024 go through every branch of the length nibbles
    rng = np.random.default_rng(3)
    for n in (1023, 1024, 1025, 2047, 2048, 5000, 20011):
        data = rng.integers(0, 256, n, dtype=np.uint8) if n % 2 else corpus[300:300 + n]
        for buf in (65536, 1000):
            got, want = _both(data, buf, **OPTION_SETS[1])
            assert got == want, (n, buf)


def test_call_patterns(corpus):
    # what the ring holds when a call ends decides what the next one emits; tiny output buffers cut headers, slices and the trailer
    rng = np.random.default_rng(9)
    data = rng.integers(0, 256, 12000, dtype=np.uint8)
    for calls, buf in (([1, 4095, 1, 5000, 2903], 5), ([12000], 1), ([4096, 4096, 3808], 4096), ([1] * 40 + [11960], 300)):
        got, want = _both(data, buf, calls=calls, window_size=12, dynamic_context_mixing=2, use_context_map=1, force_stride=9)
        assert got == want, (calls[:3], buf)


def test_long_streams_reach_every_mux_slice_size(corpus):
    # > 32 KiB of literals: the LIT coder flushes 65 536-symbol chunks mid-command; the Mux moves from 4 KiB to 16 KiB to 64 KiB
    # slices as last_flush grows (mux.rs:36-47), and past 128 KiB of LIT bytes the CMD stream counts as lagging (:456-459)
    rng = np.random.default_rng(21)
    data = np.concatenate([rng.integers(0, 256, 200000, dtype=np.uint8), corpus[:100000]])
    for buf, opts in ((65536, OPTION_SETS[0]), (4096, OPTION_SETS[2]), (1 << 20, OPTION_SETS[1])):
        got, want = _both(data, buf, **opts)
        assert got == want, buf
