"""Oracle container layer vs the reference's own tests: mux KAT, round trips, compressed-size bounds."""
import ctypes
import os

import numpy as np
import pytest

import workload

import pyoracle as po
from conftest import GOLDEN

L = po.lib()
MED, GLACIAL = (0x30, 0x4000), (0x4, 0xa00)   # probability/interface.rs:322,325


def test_mux_known_answer():
    # src/test_mux.rs:1192-1207 unit_test_decode_mux
    kat = np.fromfile(os.path.join(GOLDEN, "mux_kat.bin"), dtype=np.uint8)
    s0 = np.zeros(64, np.uint8); s1 = np.zeros(64, np.uint8)
    n0 = ctypes.c_size_t(64); n1 = ctypes.c_size_t(64); used = ctypes.c_size_t(0)
    rc = L.orc_mux_demux(kat.ctypes.data, kat.size, s0.ctypes.data, ctypes.byref(n0), s1.ctypes.data, ctypes.byref(n1), ctypes.byref(used))
    assert rc == 0 and used.value == 41
    assert s0[:n0.value].tolist() == kat[3:19].tolist()
    assert s1[:n1.value].tolist() == kat[22:38].tolist()


def _roundtrip(data, opts):
    c = po.stream_compress_raw(data, opts)
    assert c[:4].tolist() == [0xff, 0xe5, 0x8c, 0x9f] and c[-4:].tobytes() == b"ans~"
    back = po.stream_decompress(c, data.size)
    assert back.size == data.size and (back == data).all()
    return c


@pytest.mark.parametrize("use_context_map", [True, False])
@pytest.mark.parametrize("buffer_size", [1, 15, 65536])
def test_e2e_alice_literal_only(corpus, use_context_map, buffer_size):
    # integration_test.rs:111-139,225-236 e2e_no_ir(use_brotli=false): alice29 <= 0.46 with window 16,
    # literal_adaptation [MED, MED, GLACIAL, GLACIAL], dynamic_context_mixing 0, prior_depth 0
    alice = corpus[:152089]
    o = po.stream_options(window_size=16, dynamic_context_mixing=0, prior_depth=0, use_context_map=int(use_context_map),
                          force_stride=9, literal_adaptation=[MED, MED, GLACIAL, GLACIAL], call_buffer_size=buffer_size)
    c = _roundtrip(alice, o)
    assert c.size / alice.size <= 0.46


def test_e2e_small_inputs():
    # integration_test.rs: empty input, 64 x 'X', 262145 x '@' (test_e2e_empty / _64x / _262145_at)
    for raw in (b"", b"X" * 64, b"@" * 262145):
        a = np.frombuffer(raw, dtype=np.uint8)
        for bs in (1, 65536):
            _roundtrip(a, po.stream_options(call_buffer_size=bs))


def test_window_split_literals(corpus):
    # a 2^10 ring buffer turns the input into many Literal commands (raw_to_cmd/mod.rs:144-179)
    _roundtrip(corpus[:5000], po.stream_options(window_size=10))


def _bench_no_ir(shuffle384, prediction_mode, mixing, size=104857):
    # src/bin/benchmark.rs:292-343 bench_no_ir: [PredictionMode, BlockSwitchLiteral(1,2), Literal(shuffled-384 pattern)]
    data = np.resize(shuffle384, size)
    cm = (np.arange(256) & 63).astype(np.uint8)
    dm = (np.arange(256) & 63).astype(np.uint8)
    mix = np.full(8192, 4, np.uint8)
    pm = po.PredictionMode()
    pm.prediction_mode = prediction_mode
    pm.literal_context_map = cm.ctypes.data; pm.n_literal_context_map = 256
    pm.distance_context_map = dm.ctypes.data; pm.n_distance_context_map = 256
    pm.mixing_values = mix.ctypes.data; pm.has_context_speeds = 1
    c0 = po.StreamCommand(); c0.kind = 7; c0.pm = pm
    c1 = po.StreamCommand(); c1.kind = 4; c1.btype = 1; c1.stride = 2
    c2 = po.StreamCommand(); c2.kind = 3; c2.data = data.ctypes.data; c2.len = data.size
    o = po.stream_options(window_size=22, dynamic_context_mixing=mixing, prior_depth=0, use_context_map=1, force_stride=0)
    coded = po.stream_compress_commands([c0, c1, c2], o, keepalive=(data, cm, dm, mix))
    back = po.stream_decompress(coded, size)
    assert (back == data).all()
    return coded.size / size


def test_raw_literal_stream_bound(shuffle384):
    # benchmark.rs:409-417 test_raw_literal_stream: TestContextMixing (utf8, mixing 2) <= 2.5 %
    assert _bench_no_ir(shuffle384, 2, 2) <= 0.025


def test_raw_adaptive_literal_stream_bound(shuffle384):
    # benchmark.rs:419-427 test_raw_adaptive_literal_stream: TestAdapt (lsb6, no mixing) <= 29 %
    assert _bench_no_ir(shuffle384, 0, 0) <= 0.29


def test_stream_lit_bytes_equal_literal_coder(corpus):
    # the LIT-coder bytes inside the container are exactly the stand-alone literal stream (what the GPU produces)
    blk = corpus[1000:1000 + 65536]
    o = po.stream_options(use_context_map=0, force_stride=1, dynamic_context_mixing=0)   # TestSimple options
    c = po.stream_compress_raw(blk, o)
    s0 = np.zeros(c.size, np.uint8); s1 = np.zeros(c.size, np.uint8)
    n0 = ctypes.c_size_t(c.size); n1 = ctypes.c_size_t(c.size); used = ctypes.c_size_t(0)
    body = c[16:]
    assert L.orc_mux_demux(body.ctypes.data, body.size, s0.ctypes.data, ctypes.byref(n0), s1.ctypes.data, ctypes.byref(n1), ctypes.byref(used)) == 0
    ref = po.lit_encode(po.config_simple(), blk)
    assert n1.value == ref.size and (s1[:n1.value] == ref).all()


def test_brotli_derived_prediction_mode_round_trips(corpus):
    # the PredictionMode of reference testdata/alice29-priors.ir (golden fixture) through the oracle's literal coder:
    # round trip; dynamic mixing of the stride prior with the context-map prior must beat either map alone.  (The map
    # was clustered for the literals brotli leaves after its copies, so on the whole text it is no match for order 1.)
    data = corpus[:152089]
    sizes = {}
    for btype in (0, 1):
        for mixing in (0, 2):
            cfg = workload.brotli_derived_config(po.LitConfig(), btype, mixing)
            coded = po.lit_encode(cfg, data)
            assert (po.lit_decode(cfg, coded, data.size) == data).all()
            sizes[(btype, mixing)] = coded.size
    for btype in (0, 1):
        assert sizes[(btype, 2)] < sizes[(btype, 0)] < 0.55 * data.size
