#!/usr/bin/env python3
"""Generate the committed fixtures under tests/golden/ from the reference tree.

Run in the build container only (needs /root/reference, which is absent on the GPU box):
    python tests/golden/make_golden.py
Everything written here is DATA the reference's own tests use (fixture arrays, lookup tables,
test corpora) -- no reference source code is copied.  Each output names its origin.
"""
import lzma
import os
import re
import sys

import numpy as np

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def ints_of(text):
    return [int(x) for x in re.findall(r"-?\d+", text)]


def array_after(src, marker, count, open_ch="["):
    """ints of the first `[...]` literal following `marker`."""
    i = src.index(marker)
    i = src.index("=", i)
    i = src.index(open_ch, i)
    j = src.index("];", i)
    vals = ints_of(src[i:j])
    assert len(vals) == count, (marker, len(vals), count)
    return vals


def main():
    if not os.path.isdir(REF):
        sys.exit("reference tree not mounted; fixtures can only be regenerated in the build container")

    # 1. context lookup tables: src/constants.rs (RFC 7932 tables) -> 512 + 256 bytes
    consts = open(f"{REF}/src/constants.rs").read()
    utf8 = array_after(consts, "UTF8_CONTEXT_LOOKUP", 512)
    signed = array_after(consts, "SIGNED_3_BIT_CONTEXT_LOOKUP", 256)
    np.array(utf8 + signed, dtype=np.uint8).tofile(f"{OUT}/context_luts.bin")

    # 2. reciprocal tables: src/probability/div_lut.rs
    lut = open(f"{REF}/src/probability/div_lut.rs").read()
    r8 = array_after(lut, "RECIPROCAL8", 256)
    np.array(r8, dtype=np.int64).tofile(f"{OUT}/reciprocal8.i64")
    i = lut.index("pub static RECIPROCAL:")
    pairs = re.findall(r"\((\d+),(\d+)\)", lut[i:])
    assert len(pairs) == 65536
    rec = np.array([(int(a), int(b)) for a, b in pairs], dtype=np.int64)
    # the whole table is 65536 x (i64,u8); keep it xz-compressed (~150 KiB)
    with lzma.open(f"{OUT}/reciprocal16.i64x2.xz", "wb", preset=9) as f:
        f.write(rec.tobytes())

    # 3. fixture byte patterns: src/test_ans.rs:38-67 (init_shuffle_384 / init_src seed)
    tans = open(f"{REF}/src/test_ans.rs").read()
    k = tans.index("fn init_shuffle_384")
    sh = tans[k:]
    sh = sh[sh.index("let shuffled = [") + 15:]
    sh = sh[:sh.index("];")]
    shuffle = ints_of(sh)
    assert len(shuffle) == 384 and sorted(shuffle[:256]) == list(range(256)) and shuffle[256:] == shuffle[:128]
    np.array(shuffle, dtype=np.uint8).tofile(f"{OUT}/shuffle384.bin")
    k = tans.index("fn init_src")
    sd = tans[k:]
    sd = sd[sd.index("= [") + 2:]
    seed = [int(x, 16) for x in re.findall(r"0x([0-9a-fA-F]{2})", sd[:sd.index("]")])]
    assert len(seed) == 16
    np.array(seed, dtype=np.uint8).tofile(f"{OUT}/init_src_seed.bin")

    # 4. mux known-answer vector: src/test_mux.rs:1192-1207
    tmux = open(f"{REF}/src/test_mux.rs").read()
    k = tmux.index("fn unit_test_decode_mux")
    body = tmux[k:]
    body = body[body.index("= [") + 2:]
    kat = [int(x, 16) for x in re.findall(r"0x([0-9a-fA-F]{1,2})", body[:body.index("];")])]
    assert len(kat) == 41 and kat[-3:] == [0xff, 0xfe, 0xff]
    np.array(kat, dtype=np.uint8).tofile(f"{OUT}/mux_kat.bin")  # stream0 = [3:19], stream1 = [22:38]

    # 5. test corpora (Canterbury corpus texts shipped in testdata/): benchmark + parity inputs
    corpus = open(f"{REF}/testdata/alice29", "rb").read() + open(f"{REF}/testdata/asyoulik", "rb").read()
    assert len(corpus) == 277268
    with lzma.open(f"{OUT}/corpus_alice29_asyoulik.xz", "wb", preset=9) as f:
        f.write(corpus)
    rtu = open(f"{REF}/testdata/random_then_unicode", "rb").read()
    with lzma.open(f"{OUT}/random_then_unicode.xz", "wb", preset=9) as f:
        f.write(rtu)
    # 6. a brotli-derived PredictionMode: the `prediction` line of testdata/alice29-priors.ir (text IR of a brotli -q11 run
    #    with context-map clustering and per-context mixing values, parsed like src/bin/divans.rs:205-318):
    #    128 literal-context-map entries (2 block types x 64) then 8192 mixing values; mode utf8, all speeds 0 (= default)
    pred = [l for l in open(f"{REF}/testdata/alice29-priors.ir") if l.startswith("prediction")][0].split()
    assert pred[1] == "utf8"
    i, j, k, m = (pred.index(t) for t in ("lcontextmap", "dcontextmap", "mixingvalues", "cmspeedinc"))
    lmap = [int(x) for x in pred[i + 1:j]]
    mix = [int(x) for x in pred[k + 1:m]]
    assert len(lmap) == 128 and len(mix) == 8192 and all(x == "0" for x in pred[m:] if x.isdigit())
    np.array(lmap + mix, dtype=np.uint8).tofile(f"{OUT}/alice29_priors_prediction.bin")
    # 7. the textual command IRs of testdata/ (captured brotli runs; what src/bin/integration_test.rs:76-108 recodes back to the
    #    raw files, and what BASELINE configs[0](ii) feeds the codec) and the one raw file not already in the corpora
    for name in ("alice29", "alice29-q11", "alice29-priors", "asyoulik", "random_then_unicode", "ends_with_truncated_dictionary"):
        with lzma.open(f"{OUT}/ir_{name}.ir.xz", "wb", preset=9 | lzma.PRESET_EXTREME) as f:
            f.write(open(f"{REF}/testdata/{name}.ir", "rb").read())
    np.frombuffer(open(f"{REF}/testdata/ends_with_truncated_dictionary", "rb").read(), dtype=np.uint8).tofile(f"{OUT}/ends_with_truncated_dictionary.bin")
    # 8. the ONE compressed vector the reference tree holds: `_example_dv_file` in wasm/wasm.html:98-107, a 113-byte .divans
    #    container written by a real Rust build (brotli front end; the sentence "It snowed, rained, and hailed ..." seven times).
    #    tests/test_reference_vectors.py decodes it; DESIGN.md section 4 says what it pins and where its wire format is older
    #    than the tree's HEAD.
    html = open(f"{REF}/wasm/wasm.html").read()
    k = html.index("var _example_dv_file = [")
    ex = [int(x, 16) for x in re.findall(r"0x([0-9a-fA-F]{2})", html[k:html.index("];", k)])]
    assert len(ex) == 113 and ex[:4] == [0xff, 0xe5, 0x8c, 0x9f] and ex[-4:] == list(b"ans~")
    np.array(ex, dtype=np.uint8).tofile(f"{OUT}/ref_wasm_example.divans")
    print("golden fixtures written to", OUT)


if __name__ == "__main__":
    main()
