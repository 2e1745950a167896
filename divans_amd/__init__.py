"""divans_amd -- MI355X-native literal coder of dropbox/divans behind a C ABI.

Host-side mirror (Python) of the reference interface for this path.  The product path is the HIP
extension ``libdivans_hip.so`` (hand-written gfx950 kernels + C ABI, see include/divans_gpu.h);
this module only moves pointers: torch supplies device memory and streams, ctypes calls the ABI.
There is no CPU fallback -- importing works anywhere, but every codec operation raises
``DivansGpuError`` unless the extension is built and a HIP device is present.
"""
import ctypes
import os

import numpy as np

from . import build as _build

_HERE = os.path.dirname(os.path.abspath(__file__))
MAX_LITERAL_CONTEXT_MAP_SIZE = 256 * 64
NUM_MIXING_VALUES = 8192

# LiteralPredictionModeNibble values (brotli numbering)
LITERAL_PREDICTION_MODE_LSB6 = 0
LITERAL_PREDICTION_MODE_MSB6 = 1
LITERAL_PREDICTION_MODE_UTF8 = 2
LITERAL_PREDICTION_MODE_SIGN = 3


class DivansGpuError(RuntimeError):
    pass


class Speed(ctypes.Structure):
    """Speed(inc, lim), src/probability/interface.rs:298-375"""
    _fields_ = [("inc", ctypes.c_int16), ("lim", ctypes.c_int16)]


class LitConfig(ctypes.Structure):
    """divans_lit_config (include/divans_gpu.h): LiteralBookKeeping after PredictionMode + BlockSwitchLiteral."""
    _fields_ = [
        ("literal_context_map", ctypes.c_uint8 * MAX_LITERAL_CONTEXT_MAP_SIZE),
        ("mixing_mask", ctypes.c_uint8 * NUM_MIXING_VALUES),
        ("prediction_mode", ctypes.c_uint8),
        ("btype", ctypes.c_uint8),
        ("context_mixing", ctypes.c_uint8),
        ("reserved", ctypes.c_uint8),
        ("literal_adaptation", Speed * 4),
    ]


class LitSegment(ctypes.Structure):
    """divans_lit_segment (include/divans_gpu.h): one Literal command of a general stream."""
    _fields_ = [("len", ctypes.c_uint32), ("btype", ctypes.c_uint32), ("last8", ctypes.c_uint64)]


class IrOptions(ctypes.Structure):
    """divans_ir_options (include/divans_ir.h)."""
    _fields_ = [("dynamic_context_mixing", ctypes.c_uint8), ("use_context_map", ctypes.c_uint8), ("force_stride", ctypes.c_uint8),
                ("has_prior_depth", ctypes.c_uint8), ("prior_depth", ctypes.c_uint8), ("has_literal_adaptation", ctypes.c_uint8),
                ("literal_adaptation", Speed * 4)]


class BatchOptions(ctypes.Structure):
    """divans_batch_options (include/divans_batch.h)."""
    _fields_ = [("window_size", ctypes.c_int32), ("dynamic_context_mixing", ctypes.c_uint8), ("use_context_map", ctypes.c_uint8),
                ("force_stride", ctypes.c_uint8), ("has_prior_depth", ctypes.c_uint8), ("prior_depth", ctypes.c_uint8),
                ("has_literal_adaptation", ctypes.c_uint8), ("literal_adaptation", Speed * 4), ("call_buffer_size", ctypes.c_uint32),
                ("device", ctypes.c_int32), ("host_threads", ctypes.c_int32), ("skip_crc", ctypes.c_uint8)]


class ContainerProbe(ctypes.Structure):
    """divans_container_probe (include/divans_batch.h)."""
    _fields_ = [("status", ctypes.c_int32), ("window", ctypes.c_uint8), ("crc_ok", ctypes.c_uint8), ("have_prediction_mode", ctypes.c_uint8),
                ("stopped_at_command", ctypes.c_uint8), ("cmd_bytes", ctypes.c_uint32), ("lit_bytes", ctypes.c_uint32), ("commands", ctypes.c_uint32),
                ("cmd_nibbles", ctypes.c_uint32), ("first_literal_length", ctypes.c_uint32), ("literal_bytes", ctypes.c_uint64), ("cfg", LitConfig)]


WIRE_HEAD, WIRE_WASM_EXAMPLE = 0, 1


class BatchTiming(ctypes.Structure):
    _fields_ = [("total_ms", ctypes.c_double), ("gpu_ms", ctypes.c_double), ("host_overlapped_ms", ctypes.c_double), ("host_serial_ms", ctypes.c_double)]


class TablePlacement(ctypes.Structure):
    _fields_ = [("policy_candidates", ctypes.c_uint32), ("tried", ctypes.c_uint32), ("first_ms", ctypes.c_float), ("best_ms", ctypes.c_float),
                ("worst_ms", ctypes.c_float), ("kept_chunks", ctypes.c_uint32), ("searching", ctypes.c_uint32)]


class TableMemoryInfo(ctypes.Structure):
    _fields_ = [("va_reserved_bytes", ctypes.c_uint64), ("va_cap_bytes", ctypes.c_uint64), ("idle_bytes", ctypes.c_uint64), ("idle_ranges", ctypes.c_uint32)]


def table_memory():
    """divans_gpu_table_memory: address space reserved for tables and never returned, its cap, idle mapped ranges"""
    t = TableMemoryInfo()
    _check(load_library().divans_gpu_table_memory(ctypes.byref(t)), "table_memory")
    return {"va_reserved_bytes": t.va_reserved_bytes, "va_cap_bytes": t.va_cap_bytes, "idle_bytes": t.idle_bytes, "idle_ranges": t.idle_ranges}


class GpuInfo(ctypes.Structure):
    _fields_ = [
        ("rows_per_stream", ctypes.c_uint32), ("resident_groups", ctypes.c_uint32),
        ("blocks", ctypes.c_uint32), ("threads", ctypes.c_uint32),
        ("table_bytes", ctypes.c_uint64), ("scratch_bytes", ctypes.c_uint64),
        ("last_model_ms", ctypes.c_float), ("last_rans_ms", ctypes.c_float), ("last_decode_ms", ctypes.c_float),
        ("last_pack_ms", ctypes.c_float),
    ]


_LIB = None


def library_path():
    # DIVANS_HIP_LIBRARY: a differently built libdivans_hip.so (kernel experiments, scripts/build_variants.sh); never a fallback
    return os.environ.get("DIVANS_HIP_LIBRARY") or os.path.join(_HERE, "libdivans_hip.so")


def load_library():
    """Loads libdivans_hip.so (in-tree).  Raises DivansGpuError when it has not been built."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if not os.path.exists(path):
        raise DivansGpuError(
            f"{path} is missing: run `python -m divans_amd.build` (hipcc, gfx950). There is no CPU fallback.")
    # torch ships its own libamdhip64.so.7 / libhsa-runtime64; it must be the first HIP runtime mapped
    # into the process, and the extension (same SONAME) then binds to that copy.  Loading ours first
    # maps /opt/rocm's runtime and torch can no longer see the GPU.
    import torch  # noqa: F401  (plumbing: device memory + streams)
    L = ctypes.CDLL(path)
    vp, u32, u64 = ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint64
    L.divans_gpu_last_error.restype = ctypes.c_char_p
    L.divans_lit_config_simple.argtypes = [ctypes.POINTER(LitConfig)]
    L.divans_lit_config_context_mixing.argtypes = [ctypes.POINTER(LitConfig)]
    L.divans_gpu_codec_create.argtypes = [ctypes.POINTER(vp), ctypes.POINTER(LitConfig), ctypes.c_int, vp, u32]
    L.divans_gpu_codec_destroy.argtypes = [vp]
    L.divans_gpu_codec_destroy.restype = None
    L.divans_gpu_lit_encode_bound.restype = ctypes.c_size_t
    L.divans_gpu_lit_encode_bound.argtypes = [ctypes.c_size_t]
    L.divans_gpu_lit_encode_batch.argtypes = [vp, vp, vp, vp, u32, u32, vp, u64, vp, vp]
    L.divans_gpu_lit_decode_batch.argtypes = [vp, vp, vp, vp, u32, vp, vp, vp, u32]
    L.divans_gpu_pack_streams.argtypes = [vp, vp, vp, vp, u32, vp, vp, vp]
    L.divans_gpu_lit_encode_host.argtypes = [vp, vp, u32, u32, vp, ctypes.c_size_t, vp, vp, ctypes.POINTER(ctypes.c_size_t)]
    L.divans_gpu_lit_decode_host.argtypes = [vp, vp, vp, vp, u32, vp, u32]
    L.divans_gpu_codec_info.argtypes = [vp, ctypes.POINTER(GpuInfo)]
    L.divans_gpu_codec_set_geometry.argtypes = [vp, u32, u32]
    L.divans_gpu_codec_set_split_cache.argtypes = [vp, u32, u32]
    L.divans_gpu_codec_tune_tables.argtypes = [vp, u32]
    L.divans_gpu_codec_search_tables.argtypes = [vp, u32]
    L.divans_gpu_trim.argtypes = []; L.divans_gpu_trim.restype = None
    L.divans_gpu_codec_table_placement.argtypes = [vp, ctypes.POINTER(TablePlacement)]
    L.divans_gpu_table_memory.argtypes = [ctypes.POINTER(TableMemoryInfo)]
    L.divans_gpu_set_table_va_cap.argtypes = [u64]; L.divans_gpu_set_table_va_cap.restype = None
    L.divans_gpu_codec_set_decoder.argtypes = [vp, u32, ctypes.POINTER(u32), ctypes.POINTER(u32), u32]
    L.divans_gpu_codec_set_decoder.restype = ctypes.c_int
    L.divans_gpu_codec_set_byte_order.argtypes = [vp, u32]
    L.divans_gpu_codec_byte_order.argtypes = [vp, ctypes.POINTER(u32), ctypes.POINTER(u32), vp]
    L.divans_gpu_codec_row_replay.argtypes = [vp, vp, vp, vp, u32, u32, ctypes.POINTER(ctypes.c_float)]
    L.divans_gpu_codec_set_rans_split.argtypes = [vp, u32]
    L.divans_gpu_experimental_decoders.argtypes = []; L.divans_gpu_experimental_decoders.restype = ctypes.c_int
    L.divans_gpu_codec_set_encode_path.argtypes = [vp, u32]
    L.divans_gpu_codec_set_bucket_batch.argtypes = [vp, u32]
    L.divans_gpu_lit_encode_host_pipelined.argtypes = [vp, vp, u32, u32, vp, ctypes.c_size_t, vp, vp, ctypes.POINTER(ctypes.c_size_t), u32]
    L.divans_gpu_lit_decode_host_pipelined.argtypes = [vp, vp, vp, vp, u32, vp, u32, u32]
    L.divans_gpu_host_alloc.argtypes = [ctypes.c_size_t]
    L.divans_gpu_host_alloc.restype = vp
    L.divans_gpu_host_free.argtypes = [vp]
    L.divans_gpu_host_free.restype = None
    L.divans_gpu_lit_model_batch.argtypes = [vp, vp, vp, vp, u32, u32, vp]
    L.divans_gpu_lit_encode_batch_chunks.argtypes = [vp, vp, vp, vp, u32, u32, vp, u64, vp, vp, vp, u32]
    L.divans_gpu_lit_encode_batch_chunks.restype = ctypes.c_int
    L.divans_gpu_selftest_division.argtypes = [vp, ctypes.POINTER(u64)]
    L.divans_gpu_lit_stream_begin.argtypes = [vp]
    L.divans_gpu_lit_stream_encode.argtypes = [vp, vp, u32, u64, vp, ctypes.c_size_t, vp, u32, ctypes.POINTER(u32), ctypes.POINTER(ctypes.c_size_t)]
    L.divans_gpu_lit_stream_finish.argtypes = [vp, vp, ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t)]
    L.divans_gpu_lit_stream_decode_begin.argtypes = [vp]
    L.divans_gpu_lit_stream_decode.argtypes = [vp, vp, ctypes.c_size_t, u32, u64, vp, ctypes.POINTER(ctypes.c_size_t)]
    L.divans_gpu_speed_supported.argtypes = [ctypes.c_int32, ctypes.c_int32]
    L.divans_gpu_speed_supported.restype = ctypes.c_int
    L.divans_gpu_speed_accepted.argtypes = [ctypes.c_int32, ctypes.c_int32]
    L.divans_gpu_speed_accepted.restype = ctypes.c_int
    L.divans_gpu_codec_status.argtypes = [vp, ctypes.POINTER(u32)]
    L.divans_gpu_lit_encode_packed.argtypes = [vp, vp, vp, vp, u32, u32, vp, u64, vp, vp, vp, u32]
    L.divans_gpu_selftest_cdf_ops.argtypes = [vp, vp, u32, vp]
    L.divans_gpu_selftest_rans_pairs.argtypes = [vp, vp, u32, vp, ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t)]
    L.divans_gpu_codec_set_block_types.argtypes = [vp, u32]
    L.divans_gpu_lit_encode_segments_batch.argtypes = [vp, vp, vp, vp, u32, u32, vp, vp, vp, u64, vp, vp]
    L.divans_gpu_lit_decode_segments_batch.argtypes = [vp, vp, vp, vp, u32, vp, vp, vp, vp, vp, u32]
    sz = ctypes.c_size_t
    L.divans_batch_options_default.argtypes = [ctypes.POINTER(BatchOptions)]
    L.divans_batch_options_default.restype = None
    L.divans_batch_compress_bound.argtypes = [sz]; L.divans_batch_compress_bound.restype = sz
    L.divans_batch_compress.argtypes = [ctypes.POINTER(BatchOptions), vp, vp, sz, vp, sz, vp, vp, ctypes.POINTER(BatchTiming)]
    L.divans_batch_decompress.argtypes = [ctypes.POINTER(BatchOptions), vp, vp, sz, vp, sz, vp, vp, ctypes.POINTER(BatchTiming)]
    L.divans_probe_container.argtypes = [vp, sz, ctypes.c_int, ctypes.POINTER(ContainerProbe)]
    L.divans_batch_last_phases.argtypes = [ctypes.POINTER(ctypes.c_double), ctypes.c_int]; L.divans_batch_last_phases.restype = None
    L.divans_ir_parse.argtypes = [ctypes.c_char_p, sz, ctypes.POINTER(vp)]
    L.divans_ir_free.argtypes = [vp]
    L.divans_ir_free.restype = None
    for name in ("divans_ir_num_commands", "divans_ir_raw_size", "divans_ir_literal_size", "divans_ir_num_segments"):
        getattr(L, name).argtypes = [vp]; getattr(L, name).restype = sz
    L.divans_ir_count.argtypes = [vp, ctypes.c_int]; L.divans_ir_count.restype = sz
    L.divans_ir_num_block_types.argtypes = [vp]; L.divans_ir_num_block_types.restype = u32
    L.divans_ir_expand.argtypes = [vp, vp, sz]
    L.divans_ir_literal_segments.argtypes = [vp, vp, sz, vp, sz]
    L.divans_ir_options_default.argtypes = [ctypes.POINTER(IrOptions)]
    L.divans_ir_options_default.restype = None
    L.divans_ir_lit_config.argtypes = [vp, ctypes.POINTER(IrOptions), ctypes.POINTER(LitConfig)]
    _LIB = L
    return L


def experimental_decoders():
    """True if the loaded library was built with the decoders that lost their measurements (generations 1-with-caches and 4)."""
    return bool(load_library().divans_gpu_experimental_decoders())


def decoder_generations():
    """The decoder generations divans_gpu_codec_set_decoder accepts in this build."""
    return (1, 2, 3, 4) if experimental_decoders() else (2, 3)


def exported_symbols():
    """Entry points include/divans_gpu.h declares (used by the CPU-side ABI test)."""
    return [
        "divans_lit_config_simple", "divans_lit_config_context_mixing", "divans_gpu_codec_create",
        "divans_gpu_codec_destroy", "divans_gpu_last_error", "divans_gpu_lit_encode_bound",
        "divans_gpu_lit_encode_batch", "divans_gpu_lit_encode_packed", "divans_gpu_lit_decode_batch", "divans_gpu_pack_streams",
        "divans_gpu_lit_encode_host", "divans_gpu_lit_encode_host_chunks", "divans_gpu_lit_decode_host", "divans_gpu_codec_info",
        "divans_gpu_codec_set_geometry", "divans_gpu_codec_set_split_cache", "divans_gpu_codec_tune_tables", "divans_gpu_codec_search_tables", "divans_gpu_codec_table_placement", "divans_gpu_table_memory", "divans_gpu_set_table_va_cap", "divans_gpu_trim", "divans_gpu_codec_set_decoder", "divans_gpu_experimental_decoders", "divans_gpu_codec_set_byte_order", "divans_gpu_codec_byte_order", "divans_gpu_codec_row_replay", "divans_gpu_codec_set_rans_split", "divans_gpu_codec_set_encode_path", "divans_gpu_codec_set_bucket_batch", "divans_gpu_lit_model_batch",
        "divans_gpu_selftest_division", "divans_gpu_speed_supported", "divans_gpu_speed_accepted", "divans_gpu_codec_status", "divans_gpu_codec_clear_status", "divans_gpu_codec_status_async", "divans_gpu_codec_last_decode_kernel", "divans_gpu_codec_set_stream_flags", "divans_gpu_codec_set_block_types",
        "divans_gpu_lit_encode_segments_batch", "divans_gpu_lit_decode_segments_batch",
        "divans_gpu_selftest_cdf_ops", "divans_gpu_selftest_rans_pairs", "divans_gpu_lit_encode_batch_chunks",
        "divans_gpu_lit_encode_host_pipelined", "divans_gpu_lit_decode_host_pipelined", "divans_gpu_host_alloc", "divans_gpu_host_free",
        "divans_gpu_lit_stream_begin", "divans_gpu_lit_stream_encode", "divans_gpu_lit_stream_finish",
        "divans_gpu_lit_stream_decode_begin", "divans_gpu_lit_stream_decode",
    ]


def exported_batch_symbols():
    """Entry points include/divans_batch.h declares."""
    return ["divans_batch_options_default", "divans_batch_compress_bound", "divans_batch_compress", "divans_batch_decompress", "divans_batch_release", "divans_batch_release_device",
            "divans_probe_container", "divans_batch_last_phases"]


def exported_ir_symbols():
    """Entry points include/divans_ir.h declares."""
    return ["divans_ir_parse", "divans_ir_free", "divans_ir_num_commands", "divans_ir_count", "divans_ir_raw_size", "divans_ir_expand",
            "divans_ir_literal_size", "divans_ir_num_segments", "divans_ir_num_block_types", "divans_ir_literal_segments",
            "divans_ir_options_default", "divans_ir_lit_config"]


class CommandIR:
    """The reference's textual command IR (src/bin/divans.rs:191-483) parsed by the product's host code: expansion to the
    original bytes (cmd_to_raw) and the literal coder's view of a general stream (literal bytes + one segment per Literal
    command).  Host-only: works without a GPU."""
    KINDS = {"copy": 0, "dict": 1, "literal": 3, "ltype": 4, "ctype": 5, "dtype": 6, "prediction": 7}

    def __init__(self, text):
        self._lib = load_library()
        if isinstance(text, str):
            text = text.encode()
        h = ctypes.c_void_p()
        _check(self._lib.divans_ir_parse(text, len(text), ctypes.byref(h)), "divans_ir_parse")
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._lib.divans_ir_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def count(self, kind):
        return int(self._lib.divans_ir_count(self._h, self.KINDS[kind]))

    @property
    def num_block_types(self):
        return int(self._lib.divans_ir_num_block_types(self._h))

    def expand(self):
        n = int(self._lib.divans_ir_raw_size(self._h))
        out = np.empty(max(n, 1), dtype=np.uint8)
        _check(self._lib.divans_ir_expand(self._h, out.ctypes.data, n), "divans_ir_expand")
        return out[:n]

    def literal_segments(self):
        """(literal bytes uint8[n], segments structured array with fields len / btype / last8)"""
        n = int(self._lib.divans_ir_literal_size(self._h)); k = int(self._lib.divans_ir_num_segments(self._h))
        lit = np.empty(max(n, 1), dtype=np.uint8)
        segs = np.zeros(max(k, 1), dtype=np.dtype([("len", "<u4"), ("btype", "<u4"), ("last8", "<u8")]))
        _check(self._lib.divans_ir_literal_segments(self._h, lit.ctypes.data, n, segs.ctypes.data, k), "divans_ir_literal_segments")
        return lit[:n], segs[:k]

    def lit_config(self, **options):
        o = IrOptions()
        self._lib.divans_ir_options_default(ctypes.byref(o))
        for key, val in options.items():
            if key == "literal_adaptation":
                o.has_literal_adaptation = 1
                for i, (inc, lim) in enumerate(val):
                    o.literal_adaptation[i].inc = inc; o.literal_adaptation[i].lim = lim
            elif key == "prior_depth":
                o.has_prior_depth = 1; o.prior_depth = val
            else:
                setattr(o, key, val)
        cfg = LitConfig()
        _check(self._lib.divans_ir_lit_config(self._h, ctypes.byref(o), ctypes.byref(cfg)), "divans_ir_lit_config")
        return cfg


def config_simple():
    """BASELINE.json config 2: reference TestSimple (src/bin/benchmark.rs:195-206)."""
    c = LitConfig()
    load_library().divans_lit_config_simple(ctypes.byref(c))
    return c


def config_context_mixing():
    """BASELINE.json config 3: reference TestContextMixing via bench_no_ir (src/bin/benchmark.rs:156-167,305-343)."""
    c = LitConfig()
    load_library().divans_lit_config_context_mixing(ctypes.byref(c))
    return c


def speed_supported(inc, lim):
    """True when (inc, lim) is a literal_adaptation speed the GPU coder accepts (no row count ever leaves i16 under it)."""
    return bool(load_library().divans_gpu_speed_supported(int(inc), int(lim)))


def trim():
    """give the memory of the idle table ranges back (divans_gpu_trim)"""
    load_library().divans_gpu_trim()


def speed_accepted(inc, lim):
    """divans_gpu_speed_accepted: the codec takes the speed (possibly on the wrap-checked streaming kernels)"""
    return bool(load_library().divans_gpu_speed_accepted(int(inc), int(lim)))


def encode_bound(n):
    return int(load_library().divans_gpu_lit_encode_bound(int(n)))


def _check(rc, what):
    if rc != 0:
        msg = load_library().divans_gpu_last_error()
        raise DivansGpuError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")


class LiteralCodec:
    """Batch literal-stream codec (one instance per GPU / configuration).

    Mirrors what the reference does per stream with ``LiteralState::encode_or_decode_content_bytes``
    (src/codec/literal.rs:404-494) and ``ANSEncoder`` / ``ANSDecoder`` (src/ans.rs), but over
    thousands of independent streams per launch.  Tensors are torch uint8 CUDA(=HIP) tensors.
    """

    def __init__(self, config, max_stream_len, device=0, stream=None):
        import torch  # device memory + streams only
        if not torch.cuda.is_available():
            raise DivansGpuError("no HIP device visible to torch: the literal coder has no CPU fallback")
        self._torch = torch
        self._lib = load_library()
        self.device = int(device)
        self.max_stream_len = int(max_stream_len)
        self.config = config
        if stream is None:
            stream = torch.cuda.current_stream(self.device)
        self.stream = stream
        h = ctypes.c_void_p()
        _check(self._lib.divans_gpu_codec_create(ctypes.byref(h), ctypes.byref(config), self.device,
                                                 ctypes.c_void_p(stream.cuda_stream), self.max_stream_len),
               "divans_gpu_codec_create")
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._lib.divans_gpu_codec_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_geometry(self, blocks=0, cache_rows=None):
        cr = 0xFFFFFFFF if cache_rows is None else int(cache_rows)
        _check(self._lib.divans_gpu_codec_set_geometry(self._h, int(blocks), cr), "set_geometry")

    def set_encode_path(self, path):
        """0 automatic, 1 streaming model kernel, 2 bucketed model pass (mixing value 4 everywhere, streams <= 64 KiB)."""
        _check(self._lib.divans_gpu_codec_set_encode_path(self._h, int(path)), "set_encode_path")

    def set_bucket_batch(self, streams):
        """streams per launch sequence of the bucketed two-model pass (tuning / test knob)"""
        _check(self._lib.divans_gpu_codec_set_bucket_batch(self._h, int(streams)), "set_bucket_batch")

    def set_decoder(self, generation=2, rows=None, shifts=None, blocks=0):
        """Decode kernel generation (2 / 3 = lit_decode2.hip direct mapped / 2-way; 1 = lit_kernels.hip and 4 = lit_decode_t.hip only in a
        library built with DIVANS_WITH_EXPERIMENTAL_DECODERS=1) and, for generation 2 / 3, the rows / hash shifts of its
        four caches (high stride, high context-map, low stride, low context-map rows) and its persistent grid."""
        u32x4 = ctypes.c_uint32 * 4
        r = u32x4(*[int(x) for x in rows]) if rows is not None else None
        sh = u32x4(*[int(x) for x in (shifts if shifts is not None else (5, 5, 5, 5))]) if rows is not None else None
        _check(self._lib.divans_gpu_codec_set_decoder(self._h, int(generation), r, sh, int(blocks)), "set_decoder")

    def set_rans_split(self, mode):
        """rANS pass: 0 automatic, 1 one lane per 65 536-symbol chunk, 2 two lanes per chunk (one per rANS state)"""
        _check(self._lib.divans_gpu_codec_set_rans_split(self._h, int(mode)), "set_rans_split")

    def set_byte_order(self, order):
        """order of the previous byte's rows in the stride-1 decoder's tables: 0 = learned from the codec's own data (default), 1 = numeric,
        2 = the fixed English-text rank (a hint)"""
        _check(self._lib.divans_gpu_codec_set_byte_order(self._h, int(order)), "set_byte_order")

    def byte_order(self, with_rank=False):
        """dict(mode, ready[, rank]): divans_gpu_codec_byte_order"""
        mode = ctypes.c_uint32(); ready = ctypes.c_uint32()
        rank = (ctypes.c_uint8 * 256)() if with_rank else None
        _check(self._lib.divans_gpu_codec_byte_order(self._h, ctypes.byref(mode), ctypes.byref(ready), rank), "byte_order")
        d = {"mode": int(mode.value), "ready": bool(ready.value)}
        if with_rank:
            d["rank"] = list(rank)
        return d

    def row_replay(self, d_literals, n_streams, stream_len, offsets=None, sizes=None):
        """divans_gpu_codec_row_replay: ms the batch's CDF-row traffic takes by itself (device tensor of literal bytes)"""
        ms = ctypes.c_float()
        _check(self._lib.divans_gpu_codec_row_replay(self._h, d_literals.data_ptr(), offsets.data_ptr() if offsets is not None else None,
                                                     sizes.data_ptr() if sizes is not None else None, int(n_streams), int(stream_len), ctypes.byref(ms)), "row_replay")
        return float(ms.value)

    def tune_tables(self, candidates=3):
        """The next decode_batch call that fills the persistent grid runs on up to `candidates` differently placed copies of the CDF
        tables and keeps the fastest placement (divans_gpu_codec_tune_tables; 0 = the library's policy, 1 = off)."""
        _check(self._lib.divans_gpu_codec_tune_tables(self._h, int(candidates)), "tune_tables")

    def search_tables(self, candidates=0):
        """The call-by-call placement search with `candidates` placements, one per qualifying decode_batch call (divans_gpu_codec_search_tables)"""
        _check(self._lib.divans_gpu_codec_search_tables(self._h, int(candidates)), "search_tables")

    def table_placement(self):
        """what the placement tuning saw: dict(policy_candidates, tried, first_ms, best_ms, worst_ms, kept_chunks)"""
        t = TablePlacement()
        _check(self._lib.divans_gpu_codec_table_placement(self._h, ctypes.byref(t)), "table_placement")
        return {"policy_candidates": t.policy_candidates, "tried": t.tried, "first_ms": round(t.first_ms, 3), "best_ms": round(t.best_ms, 3),
                "worst_ms": round(t.worst_ms, 3), "kept": "chunks" if t.kept_chunks else "one block", "searching": bool(t.searching)}

    def set_split_cache(self, high_rows, low_rows):
        _check(self._lib.divans_gpu_codec_set_split_cache(self._h, int(high_rows), int(low_rows)), "set_split_cache")

    def info(self):
        i = GpuInfo()
        _check(self._lib.divans_gpu_codec_info(self._h, ctypes.byref(i)), "divans_gpu_codec_info")
        return i

    def last_decode_kernel(self):
        """the decode kernel instance the last decode call launched, as rocprofv3 names it"""
        buf = ctypes.create_string_buffer(160)
        _check(self._lib.divans_gpu_codec_last_decode_kernel(self._h, buf, 160), "divans_gpu_codec_last_decode_kernel")
        return buf.value.decode()

    def status(self):
        """Synchronises and returns (then clears) the sticky device status word: 1 = invalid (start,freq) in an encode
        pass, 2 = a decoded stream failed its integrity check (truncated / corrupt / wrong configuration)."""
        st = ctypes.c_uint32(0)
        _check(self._lib.divans_gpu_codec_status(self._h, ctypes.byref(st)), "divans_gpu_codec_status")
        return int(st.value)

    def selftest_cdf_ops(self, ops):
        """ops: (n, 4) uint32 script (include/divans_gpu.h) -> (n, 16) int32 records"""
        ops = np.ascontiguousarray(ops, dtype=np.uint32).reshape(-1, 4)
        out = np.zeros((ops.shape[0], 16), dtype=np.int32)
        _check(self._lib.divans_gpu_selftest_cdf_ops(self._h, ops.ctypes.data, ops.shape[0], out.ctypes.data), "divans_gpu_selftest_cdf_ops")
        return out

    def selftest_rans_pairs(self, pairs):
        pairs = np.ascontiguousarray(pairs, dtype=np.uint32)
        cap = encode_bound(pairs.size // 2) + 64
        out = np.empty(cap, dtype=np.uint8); n = ctypes.c_size_t(0)
        _check(self._lib.divans_gpu_selftest_rans_pairs(self._h, pairs.ctypes.data, pairs.size, out.ctypes.data, cap, ctypes.byref(n)),
               "divans_gpu_selftest_rans_pairs")
        return out[:n.value].copy()

    def selftest_division(self):
        m = ctypes.c_uint64(0)
        _check(self._lib.divans_gpu_selftest_division(self._h, ctypes.byref(m)), "selftest_division")
        return int(m.value)

    # ---- device-resident batch API -----------------------------------------------------------
    def alloc_encode_outputs(self, n_streams, stream_len=None):
        t = self._torch
        slot = encode_bound(stream_len or self.max_stream_len)
        dev = t.device("cuda", self.device)
        return dict(slot=slot,
                    out=t.empty(n_streams * slot + 64, dtype=t.uint8, device=dev),
                    offsets=t.empty(n_streams, dtype=t.int64, device=dev),
                    sizes=t.empty(n_streams, dtype=t.int32, device=dev))

    def encode_batch(self, d_in, n_streams, stream_len, outputs, in_offsets=None, in_sizes=None, chunk_bytes=None):
        """d_in: uint8 tensor of n_streams*stream_len bytes (stream i = rows i), or ragged streams located by the
        int64 `in_offsets` / int32 `in_sizes` device tensors (then stream_len = the longest).  Fills outputs in place.
        chunk_bytes: optional int32 device tensor [n_streams, max_chunks] that receives the coded size of every 65 536-symbol chunk."""
        if chunk_bytes is not None:
            _check(self._lib.divans_gpu_lit_encode_batch_chunks(
                self._h, d_in.data_ptr(), in_offsets.data_ptr() if in_offsets is not None else None,
                in_sizes.data_ptr() if in_sizes is not None else None, int(stream_len), int(n_streams),
                outputs["out"].data_ptr(), int(outputs["slot"]), outputs["offsets"].data_ptr(),
                outputs["sizes"].data_ptr(), chunk_bytes.data_ptr(), int(chunk_bytes.shape[1])), "divans_gpu_lit_encode_batch_chunks")
            return
        _check(self._lib.divans_gpu_lit_encode_batch(
            self._h, d_in.data_ptr(), in_offsets.data_ptr() if in_offsets is not None else None,
            in_sizes.data_ptr() if in_sizes is not None else None, int(stream_len), int(n_streams),
            outputs["out"].data_ptr(), int(outputs["slot"]), outputs["offsets"].data_ptr(),
            outputs["sizes"].data_ptr()), "divans_gpu_lit_encode_batch")

    def encode_packed(self, d_in, n_streams, stream_len, packed, packed_offsets, sizes, total, in_offsets=None, in_sizes=None, sub_batch=0):
        """divans_gpu_lit_encode_packed: coded streams contiguous in the uint8 tensor `packed` (int64 `packed_offsets`, int32 `sizes`,
        int64[1] `total`), coded in sub-batches through slots the codec owns.  status() & 8: they did not fit."""
        _check(self._lib.divans_gpu_lit_encode_packed(
            self._h, d_in.data_ptr(), in_offsets.data_ptr() if in_offsets is not None else None,
            in_sizes.data_ptr() if in_sizes is not None else None, int(stream_len), int(n_streams),
            packed.data_ptr(), int(packed.numel()), packed_offsets.data_ptr(), sizes.data_ptr(), total.data_ptr(), int(sub_batch)),
            "divans_gpu_lit_encode_packed")

    def set_block_types(self, n_btypes):
        """Context tables for literal block types 0 .. n_btypes-1 (general streams with BlockSwitchLiteral commands)."""
        _check(self._lib.divans_gpu_codec_set_block_types(self._h, int(n_btypes)), "divans_gpu_codec_set_block_types")

    def encode_segments_batch(self, d_in, in_offsets, in_sizes, n_streams, stream_len, seg_begin, segs, outputs):
        """General streams: stream i = in_sizes[i] literal bytes at in_offsets[i], split by the divans_lit_segment records
        segs[seg_begin[i] : seg_begin[i+1]] (device tensors: seg_begin int32[n+1], segs uint8 view of the 16-byte records)."""
        _check(self._lib.divans_gpu_lit_encode_segments_batch(
            self._h, d_in.data_ptr(), in_offsets.data_ptr(), in_sizes.data_ptr(), int(stream_len), int(n_streams),
            seg_begin.data_ptr(), segs.data_ptr(), outputs["out"].data_ptr(), int(outputs["slot"]), outputs["offsets"].data_ptr(),
            outputs["sizes"].data_ptr()), "divans_gpu_lit_encode_segments_batch")

    def decode_segments_batch(self, d_coded, d_offsets, d_sizes, n_streams, stream_len, seg_begin, segs, d_out, out_offsets, out_sizes):
        _check(self._lib.divans_gpu_lit_decode_segments_batch(
            self._h, d_coded.data_ptr(), d_offsets.data_ptr(), d_sizes.data_ptr(), int(n_streams), seg_begin.data_ptr(), segs.data_ptr(),
            d_out.data_ptr(), out_offsets.data_ptr(), out_sizes.data_ptr(), int(stream_len)), "divans_gpu_lit_decode_segments_batch")

    def stream_encode_pieces(self, pieces):
        """One stream handed over piece by piece (divans_gpu_lit_stream_*): returns (coded bytes, chunk sizes)."""
        import numpy as np
        _check(self._lib.divans_gpu_lit_stream_begin(self._h), "divans_gpu_lit_stream_begin")
        out = bytearray(); sizes = []
        last8 = 0
        for piece in pieces:
            piece = np.ascontiguousarray(piece, dtype=np.uint8)
            buf = np.empty(encode_bound(piece.size) + 65536, dtype=np.uint8)
            max_chunks = piece.size // 32768 + 2
            cs = np.zeros(max_chunks, dtype=np.uint32); n = ctypes.c_uint32(0); got = ctypes.c_size_t(0)
            _check(self._lib.divans_gpu_lit_stream_encode(self._h, piece.ctypes.data, piece.size, last8, buf.ctypes.data, buf.size,
                                                          cs.ctypes.data, max_chunks, ctypes.byref(n), ctypes.byref(got)), "divans_gpu_lit_stream_encode")
            out += buf[:got.value].tobytes(); sizes += [int(x) for x in cs[:n.value]]
            for b in piece[-8:]:
                last8 = (last8 >> 8) | (int(b) << 56)
        buf = np.empty(encode_bound(32768) + 64, dtype=np.uint8); got = ctypes.c_size_t(0)
        _check(self._lib.divans_gpu_lit_stream_finish(self._h, buf.ctypes.data, buf.size, ctypes.byref(got)), "divans_gpu_lit_stream_finish")
        out += buf[:got.value].tobytes()
        if got.value:
            sizes.append(got.value)
        return np.frombuffer(bytes(out), dtype=np.uint8), sizes

    def stream_decode_chunks(self, coded, n, chunks_per_call=1, slack=None):
        """One stream decoded chunk by chunk (divans_gpu_lit_stream_decode): every call sees at most `slack` coded bytes past the
        current position (default: the bound of the chunks it asks for) -- a chunk must not need more."""
        import numpy as np
        coded = np.ascontiguousarray(coded, dtype=np.uint8)
        _check(self._lib.divans_gpu_lit_stream_decode_begin(self._h), "divans_gpu_lit_stream_decode_begin")
        out = np.empty(n, dtype=np.uint8)
        pos = 0; done = 0; last8 = 0
        while done < n:
            want = min(32768 * chunks_per_call, n - done)
            window = min(coded.size - pos, slack if slack is not None else encode_bound(32768) * chunks_per_call)
            got = ctypes.c_size_t(0)
            piece = np.ascontiguousarray(coded[pos:pos + window])
            _check(self._lib.divans_gpu_lit_stream_decode(self._h, piece.ctypes.data, piece.size, want, last8,
                                                          out[done:].ctypes.data, ctypes.byref(got)), "divans_gpu_lit_stream_decode")
            for b in out[max(done, done + want - 8):done + want]:
                last8 = (last8 >> 8) | (int(b) << 56)
            pos += got.value; done += want
        return out, pos

    def model_batch(self, d_in, n_streams, stream_len, in_offsets=None, in_sizes=None):
        """Model pass only: int32 device tensor [n_streams, 2 * M] of start | freq << 16 per nibble (M = max_stream_len, even)."""
        t = self._torch
        m = (self.max_stream_len + 1) & ~1
        pairs = t.zeros((n_streams, 2 * m), dtype=t.int32, device=d_in.device)
        _check(self._lib.divans_gpu_lit_model_batch(
            self._h, d_in.data_ptr(), in_offsets.data_ptr() if in_offsets is not None else None,
            in_sizes.data_ptr() if in_sizes is not None else None, int(stream_len), int(n_streams), pairs.data_ptr()),
            "divans_gpu_lit_model_batch")
        return pairs

    def decode_batch(self, d_coded, d_offsets, d_sizes, n_streams, stream_len, d_out, out_offsets=None, out_sizes=None):
        _check(self._lib.divans_gpu_lit_decode_batch(
            self._h, d_coded.data_ptr(), d_offsets.data_ptr(), d_sizes.data_ptr(), int(n_streams),
            d_out.data_ptr(), out_offsets.data_ptr() if out_offsets is not None else None,
            out_sizes.data_ptr() if out_sizes is not None else None, int(stream_len)), "divans_gpu_lit_decode_batch")

    def pack_into(self, outputs, n_streams, packed, packed_offsets, total):
        """divans_gpu_pack_streams into caller-owned tensors (packed: uint8, >= sum of 4-byte-rounded sizes; offsets int64[n]; total int64[1])."""
        _check(self._lib.divans_gpu_pack_streams(self._h, outputs["out"].data_ptr(), outputs["offsets"].data_ptr(),
                                                 outputs["sizes"].data_ptr(), int(n_streams), packed.data_ptr(),
                                                 packed_offsets.data_ptr(), total.data_ptr()), "divans_gpu_pack_streams")

    def pack(self, outputs, n_streams):
        t = self._torch
        dev = outputs["out"].device
        packed = t.empty_like(outputs["out"])
        poff = t.empty(n_streams, dtype=t.int64, device=dev)
        total = t.zeros(1, dtype=t.int64, device=dev)
        _check(self._lib.divans_gpu_pack_streams(self._h, outputs["out"].data_ptr(), outputs["offsets"].data_ptr(),
                                                 outputs["sizes"].data_ptr(), int(n_streams), packed.data_ptr(),
                                                 poff.data_ptr(), total.data_ptr()), "divans_gpu_pack_streams")
        return packed, poff, total

    # ---- host convenience (numpy in / out) ---------------------------------------------------
    def encode_host(self, data, stream_len):
        """data: numpy uint8 of n*stream_len bytes -> (packed bytes, offsets, sizes) numpy arrays."""
        data = np.ascontiguousarray(data, dtype=np.uint8).reshape(-1)
        n = data.size // stream_len
        assert n * stream_len == data.size
        cap = encode_bound(stream_len) * n + 64
        out = np.empty(cap, dtype=np.uint8)
        offs = np.empty(n, dtype=np.uint64)
        sizes = np.empty(n, dtype=np.uint32)
        total = ctypes.c_size_t(0)
        _check(self._lib.divans_gpu_lit_encode_host(self._h, data.ctypes.data, int(stream_len), n, out.ctypes.data, cap,
                                                    offs.ctypes.data, sizes.ctypes.data, ctypes.byref(total)),
               "divans_gpu_lit_encode_host")
        return out[:total.value].copy(), offs, sizes

    def encode_host_pipelined(self, data, stream_len, slice_streams=0, out=None):
        """encode_host through the copy / code / copy pipeline; `data` and `out` should be page-locked (pinned_empty)."""
        data = np.ascontiguousarray(data, dtype=np.uint8).reshape(-1)
        n = data.size // stream_len
        assert n * stream_len == data.size
        cap = encode_bound(stream_len) * n + 64
        if out is None:
            out = np.empty(cap, dtype=np.uint8)   # a caller's `out` may be smaller: DIVANS_GPU_ECAP if the packed streams do not fit
        offs = np.empty(n, dtype=np.uint64)
        sizes = np.empty(n, dtype=np.uint32)
        total = ctypes.c_size_t(0)
        _check(self._lib.divans_gpu_lit_encode_host_pipelined(self._h, data.ctypes.data, int(stream_len), n, out.ctypes.data, out.size,
                                                              offs.ctypes.data, sizes.ctypes.data, ctypes.byref(total), int(slice_streams)),
               "divans_gpu_lit_encode_host_pipelined")
        return out[:total.value], offs, sizes

    def decode_host_pipelined(self, packed, offsets, sizes, stream_len, slice_streams=0, out=None):
        packed = np.ascontiguousarray(packed, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        sizes = np.ascontiguousarray(sizes, dtype=np.uint32)
        n = offsets.size
        if out is None:
            out = np.empty(max(n * stream_len, 1), dtype=np.uint8)
        _check(self._lib.divans_gpu_lit_decode_host_pipelined(self._h, packed.ctypes.data, offsets.ctypes.data, sizes.ctypes.data,
                                                              n, out.ctypes.data, int(stream_len), int(slice_streams)),
               "divans_gpu_lit_decode_host_pipelined")
        return out[:n * stream_len].reshape(n, stream_len) if stream_len else out[:0].reshape(n, 0)

    def decode_host(self, packed, offsets, sizes, stream_len):
        packed = np.ascontiguousarray(packed, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        sizes = np.ascontiguousarray(sizes, dtype=np.uint32)
        n = offsets.size
        out = np.empty(max(n * stream_len, 1), dtype=np.uint8)
        _check(self._lib.divans_gpu_lit_decode_host(self._h, packed.ctypes.data, offsets.ctypes.data, sizes.ctypes.data,
                                                    n, out.ctypes.data, int(stream_len)), "divans_gpu_lit_decode_host")
        return out[:n * stream_len].reshape(n, stream_len) if stream_len else out[:0].reshape(n, 0)


class PinnedBuffer:
    """Page-locked host memory from divans_gpu_host_alloc as a numpy uint8 array (`.array`); freed by close() / GC."""
    def __init__(self, nbytes):
        self._lib = load_library()
        self._p = self._lib.divans_gpu_host_alloc(int(nbytes))
        if not self._p:
            raise DivansGpuError("divans_gpu_host_alloc failed")
        self.array = np.ctypeslib.as_array((ctypes.c_uint8 * int(nbytes)).from_address(self._p))

    def close(self):
        if self._p:
            self.array = None
            self._lib.divans_gpu_host_free(self._p)
            self._p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def build_extension(force=False):
    return _build.build(force=force)


def batch_options(**kw):
    o = BatchOptions()
    load_library().divans_batch_options_default(ctypes.byref(o))
    for key, val in kw.items():
        if key == "literal_adaptation":
            o.has_literal_adaptation = 1
            for i, (inc, lim) in enumerate(val):
                o.literal_adaptation[i].inc = inc; o.literal_adaptation[i].lim = lim
        elif key == "prior_depth":
            o.has_prior_depth = 1; o.prior_depth = val
        else:
            setattr(o, key, val)
    return o


def _batch_call(fn, what, options, items, cap):
    L = load_library()
    items = [np.ascontiguousarray(x, dtype=np.uint8) for x in items]
    n = len(items)
    ptrs = (ctypes.c_void_p * max(n, 1))(*[x.ctypes.data for x in items])
    sizes = (ctypes.c_size_t * max(n, 1))(*[x.size for x in items])
    out = np.empty(max(cap, 1), dtype=np.uint8)
    offs = (ctypes.c_size_t * max(n, 1))(); osz = (ctypes.c_size_t * max(n, 1))()
    t = BatchTiming()
    _check(fn(ctypes.byref(options), ptrs, sizes, n, out.ctypes.data, cap, offs, osz, ctypes.byref(t)), what)
    res = [out[offs[i]:offs[i] + osz[i]].copy() for i in range(n)]
    return res, dict(total_ms=t.total_ms, gpu_ms=t.gpu_ms, host_overlapped_ms=t.host_overlapped_ms, host_serial_ms=t.host_serial_ms)


def batch_last_phases():
    """divans_batch_last_phases: dict of the last batch call's host phases in milliseconds"""
    out = (ctypes.c_double * 8)()
    load_library().divans_batch_last_phases(out, 8)
    return dict(zip(("cmd_coders_ms", "stage_ms", "wait_gpu_ms", "finish_ms", "gather_ms", "stage_reserve_ms", "stage_copy_ms", "stage_codec_calls_ms"), (round(float(x), 2) for x in out[:8])))


def probe_container(container, wire=WIRE_HEAD):
    """divans_probe_container (include/divans_batch.h): host-only report on one container -> ContainerProbe."""
    c = np.ascontiguousarray(container, dtype=np.uint8)
    pr = ContainerProbe()
    _check(load_library().divans_probe_container(c.ctypes.data, c.size, wire, ctypes.byref(pr)), "divans_probe_container")
    return pr


def batch_compress(inputs, options=None):
    """inputs: list of uint8 arrays -> (list of complete .divans containers, timing dict).  LIT coders on the GPU, CMD coders and framing
    on host threads, overlapped (include/divans_batch.h)."""
    L = load_library()
    options = options or batch_options()
    cap = sum(int(L.divans_batch_compress_bound(int(np.asarray(x).size))) for x in inputs) + 64
    return _batch_call(L.divans_batch_compress, "divans_batch_compress", options, inputs, cap)


def batch_decompress(containers, total_out, options=None):
    """list of complete containers -> (list of payloads, timing dict)"""
    L = load_library()
    options = options or batch_options()
    return _batch_call(L.divans_batch_decompress, "divans_batch_decompress", options, containers, int(total_out) + 64)
