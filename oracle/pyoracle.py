"""ctypes binding for the CPU oracle (test infrastructure only -- never imported by divans_amd)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

MAX_CMAP = 256 * 64
NUM_MIXING = 8192


class Speed(ctypes.Structure):
    _fields_ = [("inc", ctypes.c_int16), ("lim", ctypes.c_int16)]


class LitConfig(ctypes.Structure):
    _fields_ = [
        ("literal_context_map", ctypes.c_uint8 * MAX_CMAP),
        ("mixing_mask", ctypes.c_uint8 * NUM_MIXING),
        ("prediction_mode", ctypes.c_uint8),
        ("btype", ctypes.c_uint8),
        ("context_mixing", ctypes.c_uint8),
        ("reserved", ctypes.c_uint8),
        ("literal_adaptation", Speed * 4),
    ]


class StreamOptions(ctypes.Structure):
    _fields_ = [("window_size", ctypes.c_int), ("dynamic_context_mixing", ctypes.c_uint8), ("prior_depth", ctypes.c_uint8),
                ("use_context_map", ctypes.c_int), ("force_stride", ctypes.c_uint8), ("has_literal_adaptation", ctypes.c_int),
                ("literal_adaptation", Speed * 4), ("call_buffer_size", ctypes.c_size_t),
                ("call_inputs", ctypes.c_void_p), ("n_call_inputs", ctypes.c_size_t)]


class PredictionMode(ctypes.Structure):
    _fields_ = [("prediction_mode", ctypes.c_uint8), ("is_adv_context_map", ctypes.c_uint8),
                ("literal_context_map", ctypes.c_void_p), ("n_literal_context_map", ctypes.c_size_t),
                ("distance_context_map", ctypes.c_void_p), ("n_distance_context_map", ctypes.c_size_t),
                ("mixing_values", ctypes.c_void_p), ("has_context_speeds", ctypes.c_int),
                ("context_map_speed_f8", (ctypes.c_uint8 * 2) * 2), ("stride_speed_f8", (ctypes.c_uint8 * 2) * 2),
                ("combined_stride_speed_f8", (ctypes.c_uint8 * 2) * 2)]


class StreamCommand(ctypes.Structure):
    _fields_ = [("kind", ctypes.c_int), ("pm", PredictionMode), ("btype", ctypes.c_uint8), ("stride", ctypes.c_uint8),
                ("data", ctypes.c_void_p), ("len", ctypes.c_size_t)]


class PredictionModeResult(ctypes.Structure):
    _fields_ = [("prediction_mode", ctypes.c_uint8), ("mixing_math", ctypes.c_uint8), ("literal_adaptation", Speed * 4),
                ("literal_context_map", ctypes.c_void_p), ("mixing_values", ctypes.c_void_p)]


class WalkCommand(ctypes.Structure):
    _fields_ = [("kind", ctypes.c_uint8), ("a", ctypes.c_uint8), ("b", ctypes.c_uint8), ("x", ctypes.c_uint32), ("y", ctypes.c_uint32)]


class CmdWalk(ctypes.Structure):
    _fields_ = [("cmds", WalkCommand * 64), ("n_cmds", ctypes.c_uint32), ("nibbles", ctypes.c_uint32),
                ("state_a", ctypes.c_uint64), ("state_b", ctypes.c_uint64), ("consumed", ctypes.c_size_t), ("starved", ctypes.c_int),
                ("pm", PredictionModeResult), ("literal_context_map_nonzero", ctypes.c_uint8),
                ("mixing_value_min", ctypes.c_uint8), ("mixing_value_max", ctypes.c_uint8)]


WIRE_HEAD, WIRE_WASM_EXAMPLE = 0, 1


class Cdf16(ctypes.Structure):
    _fields_ = [("cdf", ctypes.c_int16 * 16)]


class SymStartFreq(ctypes.Structure):
    _fields_ = [("start", ctypes.c_int16), ("freq", ctypes.c_int16), ("sym", ctypes.c_uint8)]


class Weights(ctypes.Structure):
    _fields_ = [("model_weights", ctypes.c_int32 * 2), ("mixing_param", ctypes.c_uint8),
                ("normalized_weight", ctypes.c_int16)]


def build(native=False, force=False):
    out = "liboracle_native.so" if native else "liboracle.so"
    path = os.path.join(_HERE, out)
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".c", ".h"))]
    stale = (not os.path.exists(path)) or any(os.path.getmtime(s) > os.path.getmtime(path) for s in srcs)
    if force or stale:
        cmd = ["make", "-C", _HERE, "-B", "OUT=" + out]
        if native:
            cmd.append("EXTRA=-march=native")
        subprocess.run(cmd, check=True, capture_output=True)
    return path


def lib(native=False):
    global _LIB
    if _LIB is not None and not native:
        return _LIB
    L = ctypes.CDLL(build(native=native))
    u8p = ctypes.POINTER(ctypes.c_uint8)
    L.orc_lit_stream_encode.restype = ctypes.c_size_t
    L.orc_lit_stream_encode.argtypes = [ctypes.POINTER(LitConfig), ctypes.c_void_p, ctypes.c_size_t,
                                        ctypes.c_void_p, ctypes.c_size_t]
    L.orc_lit_stream_encode_trace.restype = ctypes.c_size_t
    L.orc_lit_stream_encode_trace.argtypes = [ctypes.POINTER(LitConfig), ctypes.c_void_p, ctypes.c_size_t,
                                              ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    L.orc_lit_stream_decode.restype = ctypes.c_int
    L.orc_lit_stream_decode.argtypes = [ctypes.POINTER(LitConfig), ctypes.c_void_p, ctypes.c_size_t,
                                        ctypes.c_void_p, ctypes.c_size_t]
    L.orc_lit_batch_roundtrip.restype = ctypes.c_int
    L.orc_lit_batch_roundtrip.argtypes = [ctypes.POINTER(LitConfig), ctypes.c_void_p, ctypes.c_size_t,
                                          ctypes.c_size_t, ctypes.c_int, ctypes.POINTER(ctypes.c_double),
                                          ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_uint64)]
    L.orc_lit_segments_encode.restype = ctypes.c_size_t
    L.orc_lit_segments_encode.argtypes = [ctypes.POINTER(LitConfig), ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p,
                                          ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t]
    L.orc_lit_segments_decode.restype = ctypes.c_int
    L.orc_lit_segments_decode.argtypes = [ctypes.POINTER(LitConfig), ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p,
                                          ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t]
    L.orc_lit_config_from_prediction_mode.restype = ctypes.c_int
    L.orc_lit_config_from_prediction_mode.argtypes = [ctypes.POINTER(StreamOptions), ctypes.c_void_p, ctypes.c_uint8, ctypes.POINTER(LitConfig)]
    L.orc_lit_batch_bench.restype = ctypes.c_int
    L.orc_lit_batch_bench.argtypes = L.orc_lit_batch_roundtrip.argtypes
    L.orc_lit_batch_check.restype = ctypes.c_long
    L.orc_lit_batch_check.argtypes = [ctypes.POINTER(LitConfig), ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int,
                                      ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_size_t)]
    L.orc_cdf_default.argtypes = [ctypes.POINTER(Cdf16)]
    L.orc_cdf_blend.argtypes = [ctypes.POINTER(Cdf16), ctypes.c_uint8, Speed]
    L.orc_cdf_average.argtypes = [ctypes.POINTER(Cdf16), ctypes.POINTER(Cdf16), ctypes.c_int32, ctypes.POINTER(Cdf16)]
    L.orc_cdf_sym_to_start_and_freq.argtypes = [ctypes.POINTER(Cdf16), ctypes.c_uint8, ctypes.POINTER(SymStartFreq)]
    L.orc_cdf_offset_to_sym_start_and_freq.argtypes = [ctypes.POINTER(Cdf16), ctypes.c_int16, ctypes.POINTER(SymStartFreq)]
    L.orc_fast_divide_30bit_by_16bit.restype = ctypes.c_int32
    L.orc_fast_divide_30bit_by_16bit.argtypes = [ctypes.c_int32, ctypes.c_int16]
    L.orc_fast_divide_16bit_by_8bit.restype = ctypes.c_int16
    L.orc_fast_divide_16bit_by_8bit.argtypes = [ctypes.c_uint16, ctypes.c_uint8]
    L.orc_speed_to_u8.restype = ctypes.c_uint8
    L.orc_speed_to_u8.argtypes = [ctypes.c_int16]
    L.orc_u8_to_speed.restype = ctypes.c_int16
    L.orc_u8_to_speed.argtypes = [ctypes.c_uint8]
    L.orc_speed_palette.restype = Speed
    L.orc_speed_palette.argtypes = [ctypes.c_int]
    L.orc_weights_init.argtypes = [ctypes.POINTER(Weights)]
    L.orc_weights_update.argtypes = [ctypes.POINTER(Weights), ctypes.POINTER(ctypes.c_int16 * 2), ctypes.c_int16]
    L.orc_crc32c_update.restype = ctypes.c_uint32
    L.orc_crc32c_update.argtypes = [ctypes.c_uint32, ctypes.c_void_p, ctypes.c_size_t]
    L.orc_get_lut0.argtypes = [ctypes.c_uint8, u8p]
    L.orc_get_lut1.argtypes = [ctypes.c_uint8, u8p]
    L.orc_lit_config_simple.argtypes = [ctypes.POINTER(LitConfig)]
    L.orc_lit_config_context_mixing.argtypes = [ctypes.POINTER(LitConfig)]
    L.orc_stream_options_default.argtypes = [ctypes.POINTER(StreamOptions)]
    L.orc_stream_compress.restype = ctypes.c_size_t
    L.orc_stream_compress.argtypes = [ctypes.POINTER(StreamOptions), ctypes.POINTER(StreamCommand), ctypes.c_size_t,
                                      ctypes.c_void_p, ctypes.c_size_t]
    L.orc_stream_compress_raw.restype = ctypes.c_size_t
    L.orc_stream_compress_raw.argtypes = [ctypes.POINTER(StreamOptions), ctypes.c_void_p, ctypes.c_size_t,
                                          ctypes.c_void_p, ctypes.c_size_t]
    L.orc_stream_decompress.restype = ctypes.c_int
    L.orc_stream_decompress.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t,
                                        ctypes.POINTER(ctypes.c_size_t)]
    L.orc_mux_demux.restype = ctypes.c_int
    L.orc_mux_demux.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.POINTER(ctypes.c_size_t),
                                ctypes.c_void_p, ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_size_t)]
    L.orc_cmd_stream_walk.restype = ctypes.c_int
    L.orc_cmd_stream_walk.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.POINTER(CmdWalk)]
    if not native:
        _LIB = L
    return L


def config_simple():
    c = LitConfig()
    lib().orc_lit_config_simple(ctypes.byref(c))
    return c


def config_context_mixing():
    c = LitConfig()
    lib().orc_lit_config_context_mixing(ctypes.byref(c))
    return c


def lit_encode(cfg, data, trace=False):
    data = np.ascontiguousarray(np.frombuffer(bytes(data), dtype=np.uint8)) if not isinstance(data, np.ndarray) else np.ascontiguousarray(data)
    n = data.size
    cap = 2 * n + 64
    out = np.empty(cap, dtype=np.uint8)
    if trace:
        tr = np.zeros((2 * n, 3), dtype=np.int16)
        r = lib().orc_lit_stream_encode_trace(ctypes.byref(cfg), data.ctypes.data, n, out.ctypes.data, cap, tr.ctypes.data)
    else:
        r = lib().orc_lit_stream_encode(ctypes.byref(cfg), data.ctypes.data, n, out.ctypes.data, cap)
    if r == ctypes.c_size_t(-1).value:
        raise RuntimeError("oracle encode failed")
    coded = out[:r].copy()
    return (coded, tr) if trace else coded


def lit_batch_check(cfg, blocks, coded, offsets, sizes, threads=1):
    """Encode every row of `blocks` (n x L uint8) on `threads` workers and compare with coded[offsets[i] : offsets[i] + sizes[i]].
    Returns (number of differing streams, index of the first one or n)."""
    blocks = np.ascontiguousarray(blocks, dtype=np.uint8)
    coded = np.ascontiguousarray(coded, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    sizes = np.ascontiguousarray(sizes, dtype=np.uint32)
    n, L = blocks.shape
    first = ctypes.c_size_t(n)
    bad = lib().orc_lit_batch_check(ctypes.byref(cfg), blocks.ctypes.data, n, L, int(threads), coded.ctypes.data, offsets.ctypes.data,
                                    sizes.ctypes.data, ctypes.byref(first))
    return int(bad), int(first.value)


def lit_decode(cfg, coded, n):
    coded = np.ascontiguousarray(coded, dtype=np.uint8)
    out = np.empty(max(n, 1), dtype=np.uint8)
    r = lib().orc_lit_stream_decode(ctypes.byref(cfg), coded.ctypes.data, coded.size, out.ctypes.data, n)
    if r != 0:
        raise RuntimeError("oracle decode starved")
    return out[:n]


def stream_options(**kw):
    o = StreamOptions()
    lib().orc_stream_options_default(ctypes.byref(o))
    for k, v in kw.items():
        if k == "literal_adaptation":
            o.has_literal_adaptation = 1
            for i, (inc, lim) in enumerate(v):
                o.literal_adaptation[i] = Speed(inc, lim)
        else:
            setattr(o, k, v)
    return o


def stream_compress_raw(data, opts=None, call_inputs=None):
    """the literal-only internal compressor (use_brotli = UseInternalCommandSelection); `call_inputs` = bytes handed to each
    divans_encode call (default: all in one call)"""
    opts = opts or stream_options()
    data = np.ascontiguousarray(data, dtype=np.uint8)
    if call_inputs is not None:
        ci = np.ascontiguousarray(call_inputs, dtype=np.uint64)
        opts.call_inputs = ci.ctypes.data; opts.n_call_inputs = ci.size
    cap = 2 * data.size + 65536
    out = np.empty(cap, dtype=np.uint8)
    r = lib().orc_stream_compress_raw(ctypes.byref(opts), data.ctypes.data, data.size, out.ctypes.data, cap)
    if r == ctypes.c_size_t(-1).value:
        raise RuntimeError("oracle stream compress failed")
    return out[:r].copy()


def stream_compress_commands(cmds, opts, keepalive):
    arr = (StreamCommand * len(cmds))(*cmds)
    total = sum(c.len for c in cmds)
    cap = 2 * total + 65536
    out = np.empty(cap, dtype=np.uint8)
    r = lib().orc_stream_compress(ctypes.byref(opts), arr, len(cmds), out.ctypes.data, cap)
    if r == ctypes.c_size_t(-1).value:
        raise RuntimeError("oracle stream compress failed")
    return out[:r].copy()


def stream_decompress(coded, max_out):
    coded = np.ascontiguousarray(coded, dtype=np.uint8)
    out = np.empty(max(max_out, 1), dtype=np.uint8)
    n = ctypes.c_size_t(0)
    rc = lib().orc_stream_decompress(coded.ctypes.data, coded.size, out.ctypes.data, max_out, ctypes.byref(n))
    if rc != 0:
        raise RuntimeError(f"oracle stream decompress failed rc={rc}")
    return out[:n.value].copy()


def mux_demux(body):
    """(CMD stream, LIT stream, bytes of `body` up to and including the ff fe ff end marker) of a container body (after the 16-byte header)."""
    body = np.ascontiguousarray(body, dtype=np.uint8)
    s0 = np.empty(body.size + 1, np.uint8); s1 = np.empty(body.size + 1, np.uint8)
    n0 = ctypes.c_size_t(s0.size); n1 = ctypes.c_size_t(s1.size); used = ctypes.c_size_t(0)
    if lib().orc_mux_demux(body.ctypes.data, body.size, s0.ctypes.data, ctypes.byref(n0), s1.ctypes.data, ctypes.byref(n1), ctypes.byref(used)) != 0:
        raise RuntimeError("oracle demux failed")
    return s0[:n0.value].copy(), s1[:n1.value].copy(), used.value


def cmd_stream_walk(cmd, wire=WIRE_HEAD):
    """(rc, CmdWalk) of orc_cmd_stream_walk: every command of a CMD-coder stream, final rANS states, bytes consumed."""
    cmd = np.ascontiguousarray(cmd, dtype=np.uint8)
    w = CmdWalk()
    rc = lib().orc_cmd_stream_walk(cmd.ctypes.data, cmd.size, wire, ctypes.byref(w))
    return rc, w


def lit_segments_encode(cfg, lit, seg_len, seg_btype, seg_last8):
    """LIT-coder bytes of a general stream: literal bytes `lit` split into Literal commands (lengths, block types, reloaded last_8_literals)."""
    lit = np.ascontiguousarray(lit, dtype=np.uint8)
    sl = np.ascontiguousarray(seg_len, dtype=np.uint32); sb = np.ascontiguousarray(seg_btype, dtype=np.uint32)
    s8 = np.ascontiguousarray(seg_last8, dtype=np.uint64)
    cap = 2 * lit.size + 64
    out = np.empty(cap, dtype=np.uint8)
    r = lib().orc_lit_segments_encode(ctypes.byref(cfg), lit.ctypes.data, lit.size, sl.ctypes.data, sb.ctypes.data, s8.ctypes.data, sl.size,
                                      out.ctypes.data, cap)
    if r == ctypes.c_size_t(-1).value:
        raise RuntimeError("oracle segment encode failed")
    return out[:r].copy()


def lit_segments_decode(cfg, coded, n, seg_len, seg_btype, seg_last8):
    coded = np.ascontiguousarray(coded, dtype=np.uint8)
    sl = np.ascontiguousarray(seg_len, dtype=np.uint32); sb = np.ascontiguousarray(seg_btype, dtype=np.uint32)
    s8 = np.ascontiguousarray(seg_last8, dtype=np.uint64)
    out = np.empty(max(n, 1), dtype=np.uint8)
    r = lib().orc_lit_segments_decode(ctypes.byref(cfg), coded.ctypes.data, coded.size, sl.ctypes.data, sb.ctypes.data, s8.ctypes.data, sl.size,
                                      out.ctypes.data, n)
    if r != 0:
        raise RuntimeError("oracle segment decode failed")
    return out[:n]


def lit_config_from_prediction_mode(options, pm, btype=0):
    """LiteralBookKeeping after the encoder coded PredictionMode `pm` (a PredictionMode struct or None) under `options`."""
    cfg = LitConfig()
    r = lib().orc_lit_config_from_prediction_mode(ctypes.byref(options), ctypes.byref(pm) if pm is not None else None, btype, ctypes.byref(cfg))
    if r != 0:
        raise RuntimeError("oracle could not code the PredictionMode command")
    return cfg
