"""Synthetic workload of BASELINE.md / SURVEY.md section 8d: 64 KiB blocks cut from the committed
alice29 || asyoulik corpus at stride 4099 with a seeded 1 % xorshift64* perturbation."""
import lzma
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MASK = (1 << 64) - 1


def load_corpus():
    with lzma.open(os.path.join(GOLDEN, "corpus_alice29_asyoulik.xz")) as f:
        return np.frombuffer(f.read(), dtype=np.uint8).copy()


def _xorshift64star(state, n):
    """n outputs of xorshift64* from `state` (vectorised over an array of independent states)."""
    outs = []
    x = state.copy()
    for _ in range(n):
        x ^= x >> np.uint64(12)
        x ^= (x << np.uint64(25))
        x ^= x >> np.uint64(27)
        outs.append(x * np.uint64(2685821237909765))  # wraps mod 2^64
    return outs, x


def make_blocks(corpus, first, count, block_len=65536, perturb_per_block=None):
    """Blocks first..first+count-1 as a (count, block_len) uint8 array.

    Block i = corpus[o_i : o_i + L], o_i = (i * 4099) mod (len(corpus) - L); then ~1 % of its bytes are
    XORed with a non-zero value, positions and values drawn from xorshift64* seeded 0x9E3779B97F4A7C15 ^ i."""
    L = int(block_len)
    span = corpus.size - L
    idx = np.arange(first, first + count, dtype=np.int64)
    starts = (idx * 4099) % span
    blocks = corpus[starts[:, None] + np.arange(L)[None, :]].copy()
    k = L // 100 if perturb_per_block is None else perturb_per_block
    if k:
        with np.errstate(over="ignore"):
            state = np.uint64(0x9E3779B97F4A7C15) ^ idx.astype(np.uint64)
            state[state == 0] = np.uint64(1)
            outs, _ = _xorshift64star(state, k)
        rows = np.arange(count)
        for o in outs:
            pos = (o >> np.uint64(20)) % np.uint64(L)
            val = ((o >> np.uint64(8)) & np.uint64(0xFF)).astype(np.uint8)
            val[val == 0] = 1
            blocks[rows, pos.astype(np.int64)] ^= val
    return blocks


def brotli_derived_config(cfg, btype, context_mixing):
    """Fill a LitConfig (oracle or product layout: same fields) with the PredictionMode brotli -q11 produced for
    alice29 (tests/golden/alice29_priors_prediction.bin <- reference testdata/alice29-priors.ir): clustered literal
    context map for 2 block types, per-context mixing values (0 context-map only, 1 half-byte + context, 2 no prior,
    3 context only without the nibble), utf8 context lookups, default speeds (the IR carries none)."""
    import ctypes
    raw = np.fromfile(os.path.join(GOLDEN, "alice29_priors_prediction.bin"), dtype=np.uint8)
    cmap = np.zeros(256 * 64, dtype=np.uint8); cmap[:128] = raw[:128]
    mix = np.ascontiguousarray(raw[128:128 + 8192])
    ctypes.memmove(cfg.literal_context_map, cmap.ctypes.data, cmap.size)
    ctypes.memmove(cfg.mixing_mask, mix.ctypes.data, mix.size)
    cfg.prediction_mode = 2
    cfg.btype = btype
    cfg.context_mixing = context_mixing
    for i in range(4):
        cfg.literal_adaptation[i].inc = 0x10; cfg.literal_adaptation[i].lim = 0x2000   # Speed::MUD, probability/interface.rs:323
    return cfg
