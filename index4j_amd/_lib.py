"""ctypes binding of libfmx.so (include/fmx.h).  The library is built in-tree by
``__graft_entry__.build()`` / ``make -C index4j_amd/csrc``; there is no Python or CPU fallback: if the
shared object is missing, importing the package fails loudly."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# FMX_LIBRARY: alternative build of the same library (tuning experiments only, e.g. another WALK_WAVES)
LIB_PATH = os.environ.get("FMX_LIBRARY") or os.path.join(_HERE, "libfmx.so")

OK = 0
E_ARG, E_ALPHABET, E_FORMAT, E_VERSION, E_NO_DEVICE, E_HIP, E_NOMEM, E_UNSUPPORTED = -1, -2, -3, -4, -5, -6, -7, -8

ST_OK = 0
ST_NOT_ENABLED = 1
ST_POS_NEGATIVE = 2
ST_STOP_TOO_LONG = 3
ST_DEST_TOO_SMALL = 4
ST_POS_TOO_LONG = 5
ST_DEST_SIZE_ZERO = 6
ST_NO_BOUNDARY = 7
ST_DOES_NOT_FIT = 8
ST_JAVA_AIOOBE = 9

# every symbol include/fmx.h declares (tests/test_abi.py checks the .so exports all of them)
SYMBOLS = [
    "fmx_build", "fmx_build_on_device", "fmx_build_wavelet_seconds", "fmx_suffix_table_info", "fmx_window_cells_info", "fmx_load", "fmx_save", "fmx_save_key_order_modelled", "fmx_free_buffer", "fmx_free",
    "fmx_input_length", "fmx_alphabet_length", "fmx_sample_rate", "fmx_extract_enabled",
    "fmx_blob", "fmx_to_device", "fmx_attach_device_blob", "fmx_device_blob", "fmx_host_register", "fmx_host_unregister",
    "fmx_count_batch", "fmx_locate_batch", "fmx_extract_batch", "fmx_extract_boundary_batch",
    "fmx_count_batch_dev", "fmx_count_plan_dev", "fmx_count_ordered_dev", "fmx_count_batch_is_planned", "fmx_batch_policy", "fmx_locate_batch_dev", "fmx_extract_batch_dev", "fmx_extract_boundary_batch_dev",
    "fmx_locate_extract_batch", "fmx_locate_lines_batch", "fmx_locate_extract_batch_dev", "fmx_locate_lines_batch_dev",
    "fmx_count_segments", "fmx_locate_segments", "fmx_count_segments_dev", "fmx_locate_segments_dev", "fmx_count_locate_segments_dev",
    "fmx_count_locate_segments", "fmx_resident_bytes",
    "fmx_replicate", "fmx_device_of", "fmx_shard_range", "fmx_count_batch_multi", "fmx_locate_batch_multi", "fmx_extract_batch_multi",
    "fmx_extract_boundary_batch_multi", "fmx_count_locate_segments_multi", "fmx_count_batch_multi_dev",
    "fmx_count_locate_segments_multi_dev", "fmx_multi_synchronize",
    "fmx_wavelet_build", "fmx_wavelet_rank_batch", "fmx_wavelet_inverse_select_batch",
    "fmx_rrr_build", "fmx_rrr_rank_ones_batch", "fmx_rrr_access_batch", "fmx_rrr_rank_ones_batch_dev", "fmx_rrr_access_batch_dev",
    "fmx_convert_byte_pattern", "fmx_status_message", "fmx_status_kind", "fmx_last_error", "fmx_release_scratch", "fmx_device_count", "fmx_set_option",
    "fmx_synth_log", "fmx_synth_log_multichar", "fmx_synth_patterns",
]


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "index4j_amd: %s is missing - build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C index4j_amd/csrc` (hipcc, gfx950). There is no CPU fallback." % LIB_PATH
        )
    L = C.CDLL(LIB_PATH)
    vp, i32, u16, u64 = C.c_void_p, C.c_int32, C.c_uint16, C.c_uint64
    P = C.POINTER
    sz = C.c_size_t
    L.fmx_build.argtypes = [vp, i32, i32, C.c_int, P(vp)]
    L.fmx_build_on_device.argtypes = [vp, i32, i32, C.c_int, C.c_int, P(vp), P(i32), P(C.c_int64), P(C.c_double)]
    L.fmx_build_wavelet_seconds.argtypes = [vp]
    L.fmx_build_wavelet_seconds.restype = C.c_double
    L.fmx_suffix_table_info.argtypes = [vp, P(i32), P(C.c_int64)]
    L.fmx_window_cells_info.argtypes = [vp, P(C.c_int64)]
    L.fmx_load.argtypes = [vp, sz, P(vp)]
    L.fmx_save.argtypes = [vp, C.c_int, P(vp), P(sz)]
    L.fmx_save_key_order_modelled.argtypes = [vp]
    L.fmx_batch_policy.argtypes = [vp, C.c_int, C.c_int64]
    L.fmx_free_buffer.argtypes = [vp]
    L.fmx_free_buffer.restype = None
    L.fmx_free.argtypes = [vp]
    L.fmx_free.restype = None
    for name in ("fmx_input_length", "fmx_alphabet_length", "fmx_sample_rate", "fmx_extract_enabled"):
        getattr(L, name).argtypes = [vp]
        getattr(L, name).restype = i32
    L.fmx_blob.argtypes = [vp, P(vp), P(sz)]
    L.fmx_to_device.argtypes = [vp, C.c_int]
    L.fmx_attach_device_blob.argtypes = [vp, sz, C.c_int, P(vp)]
    L.fmx_device_blob.argtypes = [vp, P(sz)]
    L.fmx_device_blob.restype = vp
    L.fmx_host_register.argtypes = [vp, sz]
    L.fmx_host_unregister.argtypes = [vp]
    L.fmx_count_batch.argtypes = [vp, vp, vp, i32, vp, vp, vp]
    L.fmx_locate_batch.argtypes = [vp, vp, vp, i32, i32, vp, i32, vp, vp, vp]
    L.fmx_extract_batch.argtypes = [vp, vp, vp, i32, vp, i32, i32, vp, vp, vp]
    L.fmx_extract_boundary_batch.argtypes = [vp, vp, i32, u16, C.c_int, vp, i32, i32, vp, vp, vp, vp]
    L.fmx_count_batch_dev.argtypes = [vp, vp, vp, i32, vp, vp, vp, vp]
    L.fmx_count_plan_dev.argtypes = [vp, vp, vp, i32, P(vp), vp]
    L.fmx_count_ordered_dev.argtypes = [vp, vp, vp, vp, i32, vp, vp, vp, vp]
    L.fmx_locate_batch_dev.argtypes = [vp, vp, vp, i32, i32, vp, i32, vp, vp, vp, vp, vp]
    L.fmx_extract_batch_dev.argtypes = [vp, vp, vp, i32, vp, i32, i32, vp, vp, vp, vp]
    L.fmx_extract_boundary_batch_dev.argtypes = [vp, vp, i32, u16, C.c_int, vp, i32, i32, vp, vp, vp, vp, vp]
    L.fmx_locate_extract_batch.argtypes = [vp, vp, vp, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp]
    L.fmx_locate_lines_batch.argtypes = [vp, vp, vp, i32, i32, u16, C.c_int, i32, vp, vp, vp, vp, vp, vp, vp, vp]
    L.fmx_locate_extract_batch_dev.argtypes = [vp, vp, vp, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.fmx_locate_lines_batch_dev.argtypes = [vp, vp, vp, i32, i32, u16, C.c_int, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.fmx_count_segments.argtypes = [vp, i32, vp, vp, i32, vp, vp, vp]
    L.fmx_locate_segments.argtypes = [vp, i32, vp, vp, vp, i32, i32, vp, vp, vp]
    L.fmx_count_segments_dev.argtypes = [vp, i32, vp, vp, i32, vp, vp, vp, vp, vp]
    L.fmx_locate_segments_dev.argtypes = [vp, i32, vp, vp, vp, i32, i32, vp, vp, vp, vp, vp]
    L.fmx_count_locate_segments_dev.argtypes = [vp, i32, vp, vp, vp, i32, i32, vp, vp, vp, vp, vp, vp, vp]
    L.fmx_count_locate_segments.argtypes = [vp, i32, vp, vp, vp, i32, i32, vp, vp, vp, vp, vp]
    L.fmx_resident_bytes.argtypes = [vp, P(C.c_int64), P(C.c_int64), P(C.c_int64)]
    L.fmx_replicate.argtypes = [vp, vp, i32, vp]
    L.fmx_device_of.argtypes = [vp]
    L.fmx_shard_range.argtypes = [C.c_int64, i32, i32, P(C.c_int64), P(C.c_int64)]
    L.fmx_shard_range.restype = None
    L.fmx_count_batch_multi.argtypes = [vp, i32, vp, vp, i32, vp, vp, vp]
    L.fmx_locate_batch_multi.argtypes = [vp, i32, vp, vp, i32, i32, vp, i32, vp, vp, vp]
    L.fmx_extract_batch_multi.argtypes = [vp, i32, vp, vp, i32, vp, i32, i32, vp, vp, vp]
    L.fmx_extract_boundary_batch_multi.argtypes = [vp, i32, vp, i32, u16, C.c_int, vp, i32, i32, vp, vp, vp, vp]
    L.fmx_count_locate_segments_multi.argtypes = [vp, i32, i32, vp, vp, vp, i32, i32, vp, vp, vp, vp, vp]
    L.fmx_count_batch_multi_dev.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp, vp]
    L.fmx_count_locate_segments_multi_dev.argtypes = [vp, i32, i32, vp, vp, vp, vp, i32, vp, vp, vp, vp, vp, vp, vp]
    L.fmx_multi_synchronize.argtypes = [vp, i32, vp]
    L.fmx_wavelet_build.argtypes = [vp, C.c_int64, i32, P(vp)]
    L.fmx_wavelet_rank_batch.argtypes = [vp, vp, vp, i32, vp, vp]
    L.fmx_wavelet_inverse_select_batch.argtypes = [vp, vp, i32, vp, vp]
    L.fmx_rrr_build.argtypes = [vp, C.c_int64, i32, P(vp)]
    L.fmx_rrr_rank_ones_batch.argtypes = [vp, vp, i32, vp]
    L.fmx_rrr_access_batch.argtypes = [vp, vp, i32, vp, vp]
    L.fmx_count_batch_is_planned.argtypes = [vp, i32]
    L.fmx_rrr_rank_ones_batch_dev.argtypes = [vp, vp, i32, vp, vp]
    L.fmx_rrr_access_batch_dev.argtypes = [vp, vp, i32, vp, vp, vp]
    L.fmx_convert_byte_pattern.argtypes = [vp, i32, i32, vp, P(i32)]
    L.fmx_status_message.argtypes = [C.c_int]
    L.fmx_status_message.restype = C.c_char_p
    L.fmx_status_kind.argtypes = [C.c_int]
    L.fmx_last_error.restype = C.c_char_p
    L.fmx_release_scratch.restype = None
    L.fmx_set_option.argtypes = [C.c_char_p, C.c_int]
    L.fmx_synth_log.argtypes = [u64, i32, vp]
    L.fmx_synth_log_multichar.argtypes = [u64, i32, i32, vp]
    L.fmx_synth_patterns.argtypes = [u64, vp, i32, i32, i32, vp, vp, vp]
    return L


lib = _load()

# experiments: FMX_OPTIONS="coarse_bits=10,groups_per_cu=8" applies fmx_set_option at import (tools/, profiling runs)
ENV_OPTIONS = {}  # what FMX_OPTIONS set (a caller that changes one of these for a while puts THIS value back, not the library's default)
for _kv in filter(None, os.environ.get("FMX_OPTIONS", "").split(",")):
    _k, _, _v = _kv.partition("=")
    if lib.fmx_set_option(_k.strip().encode(), int(_v)) != 0:
        raise ValueError("FMX_OPTIONS: bad option %r" % _kv)
    ENV_OPTIONS[_k.strip()] = int(_v)


class FmxError(RuntimeError):
    def __init__(self, code, where):
        self.code = code
        msg = lib.fmx_last_error()
        super().__init__("%s failed (%d): %s" % (where, code, msg.decode() if msg else ""))


def check(rc, where):
    if rc != OK:
        if rc == E_ALPHABET:
            raise ValueError("Input has more than 32767 different symbols")  # FM:423-426
        if rc == E_VERSION:
            raise IOError(lib.fmx_last_error().decode())  # SER:46-56
        raise FmxError(rc, where)
