"""ctypes binding of libsuffix_hip.so (C ABI: include/suffix_hip.h).

The product loads exactly one native library: the in-tree `libsuffix_hip.so`
built by hipcc for gfx950 (`python -c "import __graft_entry__ as g; g.build()"`).
There is NO CPU fallback: if the library is missing, or no HIP device is
visible, every compute entry point raises.

`Engine(lib_path=...)` exists so the test-suite can bind the same ABI exported
by the kernel-logic emulator (tests/emu/libsuffix_emu.so) and development
scripts the hook-enabled build (scripts/_devlib.py); product code never passes
it, and the package reads no environment variable.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(_HERE, "libsuffix_hip.so")

SFX_OK = 0
SFX_ERR_TOO_LARGE = 2
SFX_ERR_NO_DEVICE = 3

_vp, _u64, _u32, _int = ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_int


class KernelStat(ctypes.Structure):
    _fields_ = [("name", ctypes.c_char * 48), ("launches", _u64),
                ("total_ms", ctypes.c_double), ("algo_bytes", ctypes.c_double)]


class BuildStats(ctypes.Structure):
    _fields_ = [("n", _u64), ("sigma", _u32), ("bits_per_symbol", _u32), ("key_bits", _u32),
                ("symbols_per_key", _u32), ("rounds", _u32), ("reserved", _u32),
                ("active_after_initial", _u64), ("radix_passes", _u64),
                ("elements_sorted", _u64), ("small_bucket_resolved", _u64),
                ("tile_sorted", _u64), ("large_sorted", _u64), ("text_rounds", _u32), ("rank_rounds", _u32),
                ("deep_gathers", _u64)]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_ if k != "reserved"}


# every symbol include/suffix_hip.h declares: (name, restype, argtypes)
ABI = [
    ("sfx_strerror", ctypes.c_char_p, [_int]),
    ("sfx_device_count", _int, []),
    ("sfx_last_hip_error", ctypes.c_char_p, []),
    ("sfx_build_sa_u32", _int, [_vp, _u64, _vp]),
    ("sfx_release_cached_buffers", None, []),
    ("sfx_build_sa_u64", _int, [_vp, _u64, _vp]),
    ("sfx_widen_u32_to_u64_dev", _int, [_vp, _u64, _vp, _vp]),
    ("sfx_sa_workspace_bytes", _u64, [_u64]),
    ("sfx_build_sa_u32_dev", _int, [_vp, _u64, _vp, _vp, _u64, _vp]),
    ("sfx_build_lcp_u32", _int, [_vp, _u64, _vp, _vp]),
    ("sfx_lcp_workspace_bytes", _u64, [_u64]),
    ("sfx_build_lcp_u32_dev", _int, [_vp, _u64, _vp, _vp, _vp, _u64, _vp]),
    ("sfx_build_sa_lcp_u32", _int, [_vp, _u64, _vp, _vp]),
    ("sfx_sa_lcp_workspace_bytes", _u64, [_u64]),
    ("sfx_build_sa_lcp_u32_dev", _int, [_vp, _u64, _vp, _vp, _vp, _u64, _vp]),
    ("sfx_index_create", _int, [_vp, _u64, _vp, ctypes.POINTER(_vp)]),
    ("sfx_index_create_dev", _int, [_vp, _u64, _vp, _vp, ctypes.POINTER(_vp)]),
    ("sfx_index_query_dev", _int, [_vp, _vp, _vp, _u64, _vp, _vp, _vp, _vp, _vp]),
    ("sfx_index_destroy", None, [_vp]),
    ("sfx_index_len", _u64, [_vp]),
    ("sfx_index_table", _int, [_vp, _vp]),
    ("sfx_positions_batch", _int, [_vp, _vp, _vp, _u64, _vp, _vp]),
    ("sfx_contains_batch", _int, [_vp, _vp, _vp, _u64, _vp, _vp]),
    ("sfx_query_batch_dev", _int, [_vp, _u64, _vp, _vp, _vp, _u64, _vp, _vp, _vp, _vp, _vp]),
    ("sfx_lcp_intervals_workspace_bytes", _u64, [_u64]),
    ("sfx_lcp_intervals_dev", _int, [_vp, _u64, _vp, _vp, _vp, _vp, _vp, _vp, _u64, _vp]),
    ("sfx_doc_lookup_dev", _int, [_vp, _u64, _vp, _u64, _vp, _vp, _vp]),
    ("sfx_byte_histogram_dev", _int, [_vp, _u64, _u64, _vp, _vp]),
    ("sfx_key_histogram_dev", _int, [_vp, _u64, _u64, _u64, _vp, _int, _vp, _vp]),
    ("sfx_sa_range_workspace_bytes", _u64, [_u64, _u64]),
    ("sfx_build_sa_range_u32_dev", _int, [_vp, _u64, _vp, _int, _u32, _u32, _u64, _vp,
                                          ctypes.POINTER(_u64), _vp, _u64, _vp]),
    ("sfx_pack_text_dev", _int, [_vp, _u64, _vp, _vp, _vp, _u64, _vp]),
    ("sfx_build_sa_range_packed_u32_dev", _int, [_vp, _u64, _vp, _int, _u32, _u32, _u64, _vp,
                                                 ctypes.POINTER(_u64), _vp, _u64, _vp]),
    ("sfx_build_lcp_range_u32_dev", _int, [_vp, _u64, _vp, _u64, _u32, _vp, _vp]),
    ("sfx_query_batch_range_dev", _int, [_vp, _u64, _vp, _u64, _vp, _vp, _u64, _vp, _vp, _vp, _vp, _vp]),
    ("sfx_microbench", _int, [_int, _u64, _int, _int, _int, ctypes.POINTER(ctypes.c_double)]),
    ("sfx_profile_enable", None, [_int]),
    ("sfx_profile_reset", None, []),
    ("sfx_profile_report", _int, [ctypes.POINTER(KernelStat), _int]),
    ("sfx_last_build_stats", None, [ctypes.POINTER(BuildStats)]),
    ("sfx_build_stats_read", _u64, [_vp, _u64]),
    ("sfx_set_option", _int, [_int, _u64]),
    ("sfx_get_option", _u64, [_int]),
]
SFX_OPT_TINY_MAX = 1


class SuffixHipError(RuntimeError):
    pass


class Engine:
    """One loaded copy of the C ABI."""

    def __init__(self, lib_path=None):
        path = lib_path or DEFAULT_LIB
        if not os.path.exists(path):
            raise SuffixHipError(
                f"{path} not found: the HIP extension is not built (run "
                f"`python -c 'import __graft_entry__ as g; g.build()'`). "
                f"suffix_amd has no CPU fallback.")
        self.path = path
        # PyTorch wheels bundle their own libamdhip64.so (same SONAME as /opt/rocm's).
        # Two HIP runtimes in one process do not share the GPU ("No HIP GPUs are
        # available" from whichever initialises second), so let torch's copy load
        # first: the dynamic loader then resolves our NEEDED libamdhip64.so.7 to it.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        self.lib = ctypes.CDLL(path)
        for name, restype, argtypes in ABI:
            fn = getattr(self.lib, name)          # AttributeError = ABI symbol missing
            fn.restype = restype
            fn.argtypes = argtypes

    # -- helpers -----------------------------------------------------------------
    def check(self, status, what):
        if status != SFX_OK:
            msg = self.lib.sfx_strerror(status).decode()
            detail = self.lib.sfx_last_hip_error().decode()
            if status == SFX_ERR_TOO_LARGE:
                # the reference panics here (src/table.rs:380)
                raise OverflowError(f"{what}: {msg}")
            raise SuffixHipError(f"{what}: {msg}" + (f" [{detail}]" if detail else ""))

    def device_count(self):
        return int(self.lib.sfx_device_count())

    def require_device(self):
        if self.device_count() <= 0:
            raise SuffixHipError("no HIP device visible; suffix_amd has no CPU fallback")

    def build_stats(self):
        # the size-aware form: a library whose struct is longer than this binding's copies only what fits
        s = BuildStats()
        self.lib.sfx_build_stats_read(ctypes.byref(s), ctypes.sizeof(s))
        return s.as_dict()

    MB_COPY, MB_SCATTER4, MB_GATHER1, MB_GATHER4, MB_RUNSCATTER = range(5)

    def microbench(self, kind, nbytes, param=0, param2=0, reps=5):
        """GB/s (algorithmic bytes) of one memory-system micro-benchmark (sfx_microbench)."""
        g = ctypes.c_double(0.0)
        self.check(self.lib.sfx_microbench(kind, nbytes, param, param2, reps, ctypes.byref(g)), "sfx_microbench")
        return float(g.value)

    def release_cached_buffers(self):
        """Return the pooled device buffers of the host-pointer entry points to the driver."""
        self.lib.sfx_release_cached_buffers()

    def profile(self, on):
        self.lib.sfx_profile_enable(1 if on else 0)

    def profile_reset(self):
        self.lib.sfx_profile_reset()

    def profile_report(self):
        arr = (KernelStat * 64)()
        n = self.lib.sfx_profile_report(arr, 64)
        return [{"name": arr[i].name.decode(), "launches": int(arr[i].launches),
                 "total_ms": float(arr[i].total_ms), "algo_bytes": float(arr[i].algo_bytes)}
                for i in range(min(n, 64))]


_default = None


def default_engine():
    global _default
    if _default is None:
        _default = Engine()
    return _default


def set_default_engine(engine):
    """Development and test harnesses only (scripts/_devlib.py): every later `default_engine()` returns `engine`."""
    global _default
    _default = engine
