"""f3 on the host: the product's IR parser + command expansion (divans_amd/csrc/ir.cpp) against what the reference pins
(src/bin/integration_test.rs:76-108: recode(testdata/X.ir) reproduces testdata/X byte for byte), against an independent
test-side reader of the same grammar, and its PredictionMode handling against the oracle's."""
import ctypes
import lzma
import os

import numpy as np
import pytest

import irtext
import pyoracle as po

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = ["alice29", "alice29-q11", "alice29-priors", "asyoulik", "random_then_unicode", "ends_with_truncated_dictionary"]


def golden_raw(name, corpus, random_then_unicode):
    if name.startswith("alice29"):
        return corpus[:152089]
    if name == "asyoulik":
        assert corpus.size - 152089 == 125179    # the length integration_test.rs pins is the IR's (541 890); the raw file is this long
        return corpus[152089:]
    if name == "random_then_unicode":
        return random_then_unicode
    return np.fromfile(os.path.join(ROOT, "tests", "golden", "ends_with_truncated_dictionary.bin"), dtype=np.uint8)


def oracle_pm(pm):
    """oracle-side PredictionMode struct from the test-side reader's dict (speeds as f8, as PredictionModeContextMap stores them)"""
    keep = dict(l=np.array(pm["lcontextmap"], dtype=np.uint8), d=np.array(pm["dcontextmap"], dtype=np.uint8),
                m=np.array((pm["mixingvalues"] + [0] * 8192)[:8192], dtype=np.uint8))
    s = po.PredictionMode()
    s.prediction_mode = pm["mode"]; s.is_adv_context_map = 0
    s.literal_context_map = keep["l"].ctypes.data; s.n_literal_context_map = keep["l"].size
    s.distance_context_map = keep["d"].ctypes.data; s.n_distance_context_map = keep["d"].size
    s.mixing_values = keep["m"].ctypes.data; s.has_context_speeds = 1
    L = po.lib()
    for field, inc, mx in (("context_map_speed_f8", "cmspeedinc", "cmspeedmax"), ("stride_speed_f8", "stspeedinc", "stspeedmax"),
                           ("combined_stride_speed_f8", "mxspeedinc", "mxspeedmax")):
        for i in range(2):
            a = pm["speeds"][inc] + [0, 0]; b = pm["speeds"][mx] + [0, 0]
            getattr(s, field)[i][0] = L.orc_speed_to_u8(a[i]); getattr(s, field)[i][1] = L.orc_speed_to_u8(b[i])
    return s, keep


@pytest.mark.parametrize("name", NAMES)
def test_ir_expands_to_the_reference_files(name, corpus, random_then_unicode):
    import divans_amd as da
    raw = golden_raw(name, corpus, random_then_unicode)
    text = irtext.load_ir_text(name)
    if name == "asyoulik":
        assert len(text) == 541890            # integration_test.rs:99
    ir = da.CommandIR(text)
    out = ir.expand()
    assert out.size == raw.size and (out == raw).all()
    # the literal coder's view agrees with the independent reader: bytes, command boundaries, block types, reloaded contexts
    cmds = irtext.parse(text)
    praw, plit, psegs = irtext.expand(cmds)
    lit, segs = ir.literal_segments()
    assert (praw == raw).all() and lit.size == plit.size and (lit == plit).all()
    assert [(int(s["len"]), int(s["btype"]), int(s["last8"])) for s in segs] == psegs
    assert ir.count("literal") == len(psegs) and ir.count("copy") == sum(c[0] == "copy" for c in cmds)
    assert ir.count("dict") == sum(c[0] == "dict" for c in cmds) and ir.count("ltype") == sum(c[0] == "ltype" for c in cmds)
    assert ir.num_block_types == 1 + max([s[1] for s in psegs] + [0])
    ir.close()


@pytest.mark.parametrize("name", ["alice29-q11", "alice29-priors", "random_then_unicode", "alice29"])
@pytest.mark.parametrize("opts", [dict(), dict(dynamic_context_mixing=2), dict(dynamic_context_mixing=0, use_context_map=0),
                                  dict(dynamic_context_mixing=2, literal_adaptation=[(64, 16384), (128, 16384), (1, 16384), (4, 1024)])])
def test_ir_prediction_mode_matches_oracle(name, opts):
    """LiteralBookKeeping after the IR's PredictionMode command: product host code (CommandModel) == oracle (code_prediction_mode)."""
    import divans_amd as da
    text = irtext.load_ir_text(name)
    ir = da.CommandIR(text)
    got = ir.lit_config(**opts)
    pms = [c[1] for c in irtext.parse(text) if c[0] == "prediction"]
    so = po.stream_options(**opts)
    if pms:
        spm, keep = oracle_pm(pms[0])
        ref = po.lit_config_from_prediction_mode(so, spm)
    else:
        ref = po.lit_config_from_prediction_mode(so, None)
    assert bytes(got.literal_context_map) == bytes(ref.literal_context_map)
    assert bytes(got.mixing_mask) == bytes(ref.mixing_mask)
    assert (got.prediction_mode, got.context_mixing) == (ref.prediction_mode, ref.context_mixing)
    assert [(s.inc, s.lim) for s in got.literal_adaptation] == [(s.inc, s.lim) for s in ref.literal_adaptation]
    ir.close()


def test_ir_errors():
    import divans_amd as da
    for bad in (b"bogus 1 2\n", b"insert 3 0d0a\n", b"copy 4 from 9\n", b"dict 5 word 5,100 6f6674656e func 0\n", b"ltype 1 9\n", b"prediction klingon\n"):
        with pytest.raises(da.DivansGpuError):
            da.CommandIR(bad)
    ir = da.CommandIR(b"window 22 len 3\ninsert 3 414243\ncopy 5 from 2\ninsert 0 \n")
    assert ir.expand().tobytes() == b"ABCBCBCB" and ir.count("literal") == 1
    ir.close()
