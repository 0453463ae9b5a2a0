"""suffix_amd -- MI355X (gfx950) engine for the suffix-array hot path of
BurntSushi/suffix: SuffixTable::new (SA construction), lcp_lens (LCP) and
batched positions()/contains(), behind the C ABI in include/suffix_hip.h.

Only what the path needs lives here: csrc/ (HIP kernels + C ABI), the ctypes
binding, the `SuffixTable` mirror of the reference API, device-resident entry
points for torch tensors, and the range-partitioned multi-GPU build.
"""
from ._lib import Engine, SuffixHipError, default_engine  # noqa: F401
from .table import SuffixTable  # noqa: F401

__all__ = ["SuffixTable", "Engine", "SuffixHipError", "default_engine"]
