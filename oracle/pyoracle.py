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


def lit_decode(cfg, coded, n):
    coded = np.ascontiguousarray(coded, dtype=np.uint8)
    out = np.empty(max(n, 1), dtype=np.uint8)
    r = lib().orc_lit_stream_decode(ctypes.byref(cfg), coded.ctypes.data, coded.size, out.ctypes.data, n)
    if r != 0:
        raise RuntimeError("oracle decode starved")
    return out[:n]
