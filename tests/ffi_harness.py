"""The calling patterns of the reference's c/example.c over the per-stream C ABI (include/divans_ffi.h), parameterised by the loaded
library: the product (divans_amd/libdivans_hip.so, GPU tests) or the host-logic harness of tests/c (CPU tests, test_host_logic_cpu.py)."""
import ctypes

import numpy as np


class CAllocator(ctypes.Structure):
    _fields_ = [("alloc_func", ctypes.c_void_p), ("free_func", ctypes.c_void_p), ("opaque", ctypes.c_void_p)]


def bind(L):
    vp = ctypes.c_void_p
    L.divans_new_compressor.restype = vp
    L.divans_new_decompressor.restype = vp
    L.divans_set_option.argtypes = [vp, ctypes.c_uint8, ctypes.c_uint32]
    L.divans_set_option.restype = ctypes.c_uint8
    szp = ctypes.POINTER(ctypes.c_size_t)
    L.divans_encode.argtypes = [vp, vp, ctypes.c_size_t, szp, vp, ctypes.c_size_t, szp]
    L.divans_encode.restype = ctypes.c_uint8
    L.divans_encode_flush.argtypes = [vp, vp, ctypes.c_size_t, szp]
    L.divans_encode_flush.restype = ctypes.c_uint8
    L.divans_decode.argtypes = [vp, vp, ctypes.c_size_t, szp, vp, ctypes.c_size_t, szp]
    L.divans_decode.restype = ctypes.c_uint8
    L.divans_free_compressor.argtypes = [vp]
    L.divans_free_decompressor.argtypes = [vp]
    return L


def ffi_compress(L, data, options, buf_size=65536, feed=None, trace=None):
    """c/example.c's loop, once per piece of `feed` bytes of input (default: one piece = everything): divans_encode is called with
    what is left of the piece and an empty buffer until the piece is taken.  `trace` collects the bytes every encode call returned."""
    st = L.divans_new_compressor()
    for sel, val in options:
        assert L.divans_set_option(st, sel, val) == 0
    data = np.ascontiguousarray(data, dtype=np.uint8)
    out = bytearray()
    buf = np.empty(buf_size, np.uint8)
    off = 0
    piece_end = 0
    while off < data.size:
        if off == piece_end:
            piece_end = data.size if feed is None else min(off + feed, data.size)
        ro = ctypes.c_size_t(0); wo = ctypes.c_size_t(0)
        r = L.divans_encode(st, data.ctypes.data + off, piece_end - off, ctypes.byref(ro), buf.ctypes.data, buf_size, ctypes.byref(wo))
        assert r != 3
        off += ro.value; out += buf[:wo.value].tobytes()
        if trace is not None:
            trace.append(wo.value)
    while True:
        wo = ctypes.c_size_t(0)
        r = L.divans_encode_flush(st, buf.ctypes.data, buf_size, ctypes.byref(wo))
        assert r != 3
        out += buf[:wo.value].tobytes()
        if r == 0:
            break
    L.divans_free_compressor(st)
    return np.frombuffer(bytes(out), dtype=np.uint8)


def ffi_decompress(L, coded, expect_len, buf_size=65536, feed=100000):
    st = L.divans_new_decompressor()
    coded = np.ascontiguousarray(coded, dtype=np.uint8)
    out = bytearray(); buf = np.empty(buf_size, np.uint8); off = 0
    while True:
        ro = ctypes.c_size_t(0); wo = ctypes.c_size_t(0)
        n = min(feed, coded.size - off)
        r = L.divans_decode(st, coded.ctypes.data + off, n, ctypes.byref(ro), buf.ctypes.data, buf_size, ctypes.byref(wo))
        assert r != 3 and not (r == 1 and off + ro.value >= coded.size and n == 0)
        off += ro.value; out += buf[:wo.value].tobytes()
        if r == 0:
            break
    L.divans_free_decompressor(st)
    assert len(out) == expect_len
    return np.frombuffer(bytes(out), dtype=np.uint8)


OPTION_SETS = [
    # (ffi options, oracle options): the literal-only internal compressor = DIVANS_OPTION_USE_BROTLI_COMMAND_SELECTION 0
    ([(5, 0)], dict()),
    ([(5, 0), (7, 0), (9, 1), (4, 0)], dict(use_context_map=0, force_stride=1, dynamic_context_mixing=0)),       # TestSimple
    ([(5, 0), (4, 2), (9, 0), (11, 0)], dict(dynamic_context_mixing=2, force_stride=0)),                          # mixing on
    ([(5, 0), (2, 16), (4, 0), (12, 12), (8, 5), (14, 10), (13, 3)],
     dict(window_size=16, dynamic_context_mixing=0, literal_adaptation=[(64, 16384), (128, 16384), (1, 16384), (4, 1024)])),
]
