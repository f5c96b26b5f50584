"""index4j_amd — MI355X-native engine for the backward-search path of dynatrace-oss/index4j.

The package is a thin host mirror (``FmIndex`` / ``FmIndexBuilder``) over ``libfmx.so``, the C-ABI
library holding the host-side index builder / serializer and the HIP kernels (csrc/).  It has no
CPU query path; the HIP library must be built (``__graft_entry__.build()``)."""
from ._lib import FmxError, LIB_PATH, SYMBOLS, lib  # noqa: F401
from .fmindex import (  # noqa: F401
    FmIndex,
    FmIndexBuilder,
    as_chars,
    chars_to_str,
    pack_patterns,
    raise_for_status,
    synth_log,
    synth_log_multichar,
    synth_patterns,
)
from .replicas import ReplicaSet, SegmentReplicaSet, shard_range  # noqa: F401
from .rrr import RrrVector  # noqa: F401
from .segments import SegmentedFmIndex, cut_points  # noqa: F401
from .wavelet import WaveletFixedBlockBoosting  # noqa: F401

__all__ = ["ReplicaSet", "SegmentReplicaSet", "shard_range", "WaveletFixedBlockBoosting", "RrrVector", "SegmentedFmIndex", "cut_points", "FmIndex", "FmIndexBuilder", "FmxError", "as_chars", "chars_to_str", "pack_patterns",
           "raise_for_status", "synth_log", "synth_log_multichar", "synth_patterns", "lib", "LIB_PATH", "SYMBOLS"]
