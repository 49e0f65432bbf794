"""ctypes binding of libpsa_amd.so (the C ABI declared in include/psa.h).

This is plumbing only: every search runs in the hand-written HIP kernels
behind the C ABI.  There is no Python or CPU implementation of the search in
this package; if the shared library is missing, importing fails, and if no
HIP device is usable every compute call raises PsaError(PSA_ERR_DEVICE).

Reference interface mirrored here (GuyKabiri/Parallel-Sequence-Alignment):
  gpu_run_program(ProgramData*, Mutant*, first, last)   cuda_funcs.h:33
  ProgramData / Mutant                                  program_data.h:6-11, mutant.h:6-10
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass
from typing import Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PSA_AMD_LIB") or os.path.join(_HERE, "libpsa_amd.so")   # override: kernel experiments

PSA_OK, PSA_ERR_ARG, PSA_ERR_ALPHABET, PSA_ERR_DEVICE, PSA_ERR_STATE = range(5)
EXCHANGE_RCCL, EXCHANGE_HOST = 0, 1
# the reference's own (C++-mangled) name of gpu_run_program (cuda_funcs.h:33), exported next to the C name
CXX_GPU_RUN_PROGRAM = "_Z15gpu_run_programP5_dataP7_mutantii"
ENGINE_AUTO, ENGINE_SEQUENTIAL, ENGINE_FILTER = 0, 1, 2
MAX_SEQ1 = 1 << 30          # PSA_MAX_SEQ1 (include/psa_types.h): longer Seq1 -> PSA_ERR_ARG

# every symbol include/psa.h declares (tests check the library exports all of them)
EXPORTED_SYMBOLS = (
    "gpu_run_program", "gpu_run_program_layout",
    "psa_create", "psa_destroy", "psa_set_weights",
    "psa_set_seq1", "psa_set_seq2", "psa_set_seq1_device", "psa_set_seq2_device",
    "psa_set_sequences", "psa_set_sequences_device",
    "psa_set_engine", "psa_set_batch_streams", "psa_search", "psa_search_async", "psa_search_wait", "psa_search_device",
    "psa_raw_retry_count", "psa_scores", "psa_streamed_count", "psa_stream_retry_count",
    "psa_search_batch", "psa_search_batch_device",
    "psa_search_batch_minmax", "psa_search_batch_minmax_device",
    "psa_fallback_count", "psa_reselect_count", "psa_unfused_count", "psa_duo_retry_count",
    "psa_scratch_fallback_count", "psa_scratch_shrink_count", "psa_share_sweep_count", "psa_share_pause_count",
    "psa_group_create", "psa_group_create_ex", "psa_group_destroy", "psa_group_size", "psa_group_set_weights",
    "psa_group_set_seq1", "psa_group_set_seq2", "psa_search_sharded", "psa_search_batch_sharded", "psa_group_last_timing",
    "psa_group_set_timeout",
    "psa_fold", "psa_shard_range", "psa_query_block", "psa_query_block_max", "psa_batch_unpack", "psa_build_tables",
    "psa_set_profiling", "psa_last_timing", "psa_profile_totals", "psa_last_kernel_queries",
    "psa_generate_sequence", "psa_last_error", "psa_version",
)
# test hooks of libpsa_debug.so only (include/psa_debug.h; the product library does not export them)
DEBUG_SYMBOLS = ("psa_debug_counts", "psa_debug_rescore")


class PsaError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"psa status {status}: {message}")
        self.status = status


class Result(C.Structure):
    """psa_result (include/psa.h) — 32 bytes."""
    _fields_ = [("score", C.c_double), ("offset", C.c_int32), ("char_offset", C.c_int32),
                ("ch", C.c_int32), ("shard", C.c_int32), ("pairs", C.c_int64)]

    def as_tuple(self):
        return (self.score, self.offset, self.char_offset, chr(self.ch) if self.ch else "")


class Mutant(C.Structure):
    """Mutant (mutant.h:6-10)."""
    _fields_ = [("offset", C.c_int), ("char_offset", C.c_int), ("ch", C.c_char)]


def _load() -> C.CDLL:
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no fallback implementation.")
    lib = C.CDLL(LIB_PATH)
    vp, i, d = C.c_void_p, C.c_int, C.c_double
    dp = C.POINTER(C.c_double)
    lib.psa_last_error.restype = C.c_char_p
    lib.psa_version.restype = C.c_char_p
    lib.psa_create.argtypes = [i, vp, C.POINTER(vp)]
    lib.psa_destroy.argtypes = [vp]
    lib.psa_destroy.restype = None
    lib.psa_set_weights.argtypes = [vp, dp, i]
    lib.psa_set_seq1.argtypes = [vp, C.c_char_p, i]
    lib.psa_set_seq2.argtypes = [vp, C.c_char_p, i]
    lib.psa_set_seq1_device.argtypes = [vp, vp, i]
    lib.psa_set_seq2_device.argtypes = [vp, vp, i]
    lib.psa_set_sequences.argtypes = [vp, C.c_char_p, i, C.c_char_p, i]
    lib.psa_set_sequences_device.argtypes = [vp, vp, i, vp, i]
    lib.psa_set_engine.argtypes = [vp, i]
    lib.psa_set_batch_streams.argtypes = [vp, i]
    lib.psa_search.argtypes = [vp, i, i, C.POINTER(Result)]
    lib.psa_search_async.argtypes = [vp, i, i]
    lib.psa_search_device.argtypes = [vp, vp, i, vp, i, i, i, C.POINTER(Result)]
    lib.psa_raw_retry_count.argtypes = [vp]
    lib.psa_streamed_count.argtypes = [vp]
    lib.psa_stream_retry_count.argtypes = [vp]
    lib.psa_search_wait.argtypes = [vp, C.POINTER(Result)]
    lib.psa_scores.argtypes = [vp, i, i, vp, vp, vp]
    lib.psa_search_batch.argtypes = [vp, C.c_char_p, i, i, C.c_size_t, i, i, C.POINTER(Result)]
    lib.psa_search_batch_device.argtypes = [vp, vp, i, i, C.c_size_t, i, i, C.POINTER(Result)]
    lib.psa_search_batch_minmax.argtypes = [vp, C.c_char_p, i, i, C.c_size_t, i, i, C.POINTER(Result), C.POINTER(Result)]
    lib.psa_search_batch_minmax_device.argtypes = [vp, vp, i, i, C.c_size_t, i, i, C.POINTER(Result), C.POINTER(Result)]
    if hasattr(lib, "psa_debug_counts"):                   # libpsa_debug.so
        lib.psa_debug_counts.argtypes = [vp, i, i, vp, vp, vp]
        lib.psa_debug_rescore.argtypes = [vp, vp, i, vp]
    lib.psa_fallback_count.argtypes = [vp]
    # (symbols newer than round 4 are bound when present: tools/shard_ab.sh loads older builds of the library for A/B runs;
    # __graft_entry__.build() and the tests check EXPORTED_SYMBOLS against the in-tree build)
    if hasattr(lib, "psa_scratch_fallback_count"):
        lib.psa_scratch_fallback_count.argtypes = [vp]
        lib.psa_scratch_shrink_count.argtypes = [vp]
    if hasattr(lib, "psa_share_sweep_count"):
        lib.psa_share_sweep_count.argtypes = [vp]
        lib.psa_share_sweep_count.restype = C.c_longlong
    if hasattr(lib, "psa_share_pause_count"):
        lib.psa_share_pause_count.argtypes = [vp]
    lib.psa_reselect_count.argtypes = [vp]
    lib.psa_duo_retry_count.argtypes = [vp]
    lib.psa_unfused_count.argtypes = [vp]
    lib.psa_group_create.argtypes = [C.POINTER(i), i, C.POINTER(vp)]
    lib.psa_group_create_ex.argtypes = [C.POINTER(i), i, i, C.POINTER(vp)]
    lib.psa_group_destroy.argtypes = [vp]
    lib.psa_group_destroy.restype = None
    lib.psa_group_size.argtypes = [vp]
    lib.psa_group_set_weights.argtypes = [vp, dp, i]
    lib.psa_group_set_seq1.argtypes = [vp, C.c_char_p, i]
    lib.psa_group_set_seq2.argtypes = [vp, C.c_char_p, i]
    lib.psa_search_sharded.argtypes = [vp, i, i, C.POINTER(Result), C.POINTER(Result)]
    lib.psa_search_batch_sharded.argtypes = [vp, C.c_char_p, i, i, C.c_size_t, i, i, C.POINTER(Result), C.POINTER(Result)]
    lib.psa_group_last_timing.argtypes = [vp, dp, dp]
    if hasattr(lib, "psa_group_set_timeout"):
        lib.psa_group_set_timeout.argtypes = [vp, C.c_double]
    lib.psa_fold.argtypes = [C.POINTER(Result), i, i, C.POINTER(Result)]
    lib.psa_shard_range.argtypes = [i, i, i, C.POINTER(i), C.POINTER(i)]
    lib.psa_shard_range.restype = None
    lib.psa_query_block.argtypes = [i, i, i, C.POINTER(i), C.POINTER(i)]
    lib.psa_query_block.restype = None
    lib.psa_query_block_max.argtypes = [i, i]
    lib.psa_batch_unpack.argtypes = [C.POINTER(Result), i, i, C.POINTER(Result), C.POINTER(Result)]
    lib.psa_build_tables.argtypes = [dp, i, vp, vp, vp, vp]
    lib.psa_set_profiling.argtypes = [vp, i]
    lib.psa_last_timing.argtypes = [vp, i, C.POINTER(C.c_float), C.POINTER(C.c_char_p)]
    lib.psa_profile_totals.argtypes = [vp, dp, dp, C.POINTER(i)]
    lib.psa_last_kernel_queries.argtypes = [vp]
    lib.psa_generate_sequence.argtypes = [C.c_uint64, i, C.c_char_p]
    lib.psa_generate_sequence.restype = None
    lib.gpu_run_program.argtypes = [vp, C.POINTER(Mutant), i, i]
    lib.gpu_run_program.restype = d
    lib.gpu_run_program_layout.argtypes = [vp, C.c_size_t, C.c_size_t, C.POINTER(Mutant), i, i]
    lib.gpu_run_program_layout.restype = d
    cxx = getattr(lib, CXX_GPU_RUN_PROGRAM)
    cxx.argtypes = [vp, C.POINTER(Mutant), i, i]
    cxx.restype = d
    return lib


lib = _load()


def _check(status: int) -> None:
    if status != PSA_OK:
        raise PsaError(status, lib.psa_last_error().decode())


def _weights(w: Sequence[float]):
    if len(w) != 4:
        raise ValueError("weights must have 4 entries (w1..w4, def.h:12-15)")
    return (C.c_double * 4)(*[float(x) for x in w])


def version() -> str:
    return lib.psa_version().decode()


def generate_sequence(seed: int, n: int) -> bytes:
    """Uniform A..Z letters from the library's counter-based SplitMix64 stream."""
    buf = C.create_string_buffer(n + 1)
    lib.psa_generate_sequence(C.c_uint64(seed & (2**64 - 1)), n, buf)
    return buf.raw[:n]


def generate_sequence_np(seed: int, n: int) -> np.ndarray:
    """Same stream as generate_sequence, vectorised (uint8 array of ASCII codes)."""
    with np.errstate(over="ignore"):
        z = np.uint64(seed & (2**64 - 1)) + (np.arange(n, dtype=np.uint64) + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z ^= z >> np.uint64(31)
        return (np.uint64(65) + (((z >> np.uint64(32)) * np.uint64(26)) >> np.uint64(32))).astype(np.uint8)


@dataclass
class Tables:
    P: np.ndarray      # [32,32] float64, index [c1][c2]
    SUB: np.ndarray    # [32,32] uint8 (ASCII, 0 = none)
    DELTA: np.ndarray  # [32,32] float64
    SIGN: np.ndarray   # [32,32] uint8 (ASCII sign, 0 = padding)


def build_tables(weights: Sequence[float], is_max: bool) -> Tables:
    P = np.zeros((32, 32), np.float64)
    SUB = np.zeros((32, 32), np.uint8)
    DELTA = np.zeros((32, 32), np.float64)
    SIGN = np.zeros((32, 32), np.uint8)
    _check(lib.psa_build_tables(_weights(weights), int(bool(is_max)), P.ctypes.data, SUB.ctypes.data,
                                DELTA.ctypes.data, SIGN.ctypes.data))
    return Tables(P, SUB, DELTA, SIGN)


def shard_range(total_offsets: int, nshards: int, rank: int):
    lo, hi = C.c_int(), C.c_int()
    lib.psa_shard_range(total_offsets, nshards, rank, C.byref(lo), C.byref(hi))
    return lo.value, hi.value


def query_block(nq: int, nshards: int, rank: int):
    """Queries [lo, hi) of shard `rank` when nq queries are dealt to nshards shards (psa_query_block)."""
    lo, hi = C.c_int(), C.c_int()
    lib.psa_query_block(nq, nshards, rank, C.byref(lo), C.byref(hi))
    return lo.value, hi.value


def query_block_max(nq: int, nshards: int) -> int:
    return lib.psa_query_block_max(nq, nshards)


def batch_unpack(gathered: Sequence[Result], nq: int, nshards: int, minmax: bool = False):
    """The batch exchange's gathered blocks -> results in query order (psa_batch_unpack)."""
    arr = (Result * len(gathered))(*gathered)
    hi, lo = (Result * nq)(), (Result * nq)()
    _check(lib.psa_batch_unpack(arr, nq, nshards, hi, lo if minmax else None))
    return (list(hi), list(lo)) if minmax else list(hi)


def fold(records: Sequence[Result], is_max: bool) -> Result:
    arr = (Result * len(records))(*records)
    out = Result()
    _check(lib.psa_fold(arr, len(records), int(bool(is_max)), C.byref(out)))
    return out


class Context:
    """Persistent device context (psa_create .. psa_destroy)."""

    def __init__(self, device: int = 0, stream: Optional[int] = None):
        self._h = C.c_void_p()
        _check(lib.psa_create(device, C.c_void_p(stream or 0), C.byref(self._h)))
        self.n = self.m = None

    def close(self):
        if getattr(self, "_h", None) and self._h.value and lib is not None:      # (lib is None at interpreter exit)
            lib.psa_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def set_weights(self, weights: Sequence[float], is_max: bool):
        _check(lib.psa_set_weights(self._h, _weights(weights), int(bool(is_max))))

    def set_seq1(self, seq1: bytes):
        _check(lib.psa_set_seq1(self._h, seq1, len(seq1)))
        self.n = len(seq1)

    def set_seq2(self, seq2: bytes):
        _check(lib.psa_set_seq2(self._h, seq2, len(seq2)))
        self.m = len(seq2)

    def set_sequences(self, seq1: bytes, seq2: bytes):
        """Both sequences at once: one encode launch (psa_set_sequences)."""
        _check(lib.psa_set_sequences(self._h, seq1, len(seq1), seq2, len(seq2)))
        self.n, self.m = len(seq1), len(seq2)

    def set_sequences_device(self, ptr1: int, n: int, ptr2: int, m: int):
        _check(lib.psa_set_sequences_device(self._h, C.c_void_p(ptr1), n, C.c_void_p(ptr2), m))
        self.n, self.m = n, m

    def set_seq1_device(self, ptr: int, n: int):
        _check(lib.psa_set_seq1_device(self._h, C.c_void_p(ptr), n))
        self.n = n

    def set_seq2_device(self, ptr: int, m: int):
        _check(lib.psa_set_seq2_device(self._h, C.c_void_p(ptr), m))
        self.m = m

    def set_engine(self, engine: int):
        _check(lib.psa_set_engine(self._h, engine))

    def set_batch_streams(self, streams: int):
        _check(lib.psa_set_batch_streams(self._h, streams))

    def total_offsets(self) -> int:
        return self.n - self.m + 1

    def search(self, first: Optional[int] = None, last: Optional[int] = None) -> Result:
        first = 0 if first is None else first
        last = self.total_offsets() if last is None else last
        out = Result()
        _check(lib.psa_search(self._h, first, last, C.byref(out)))
        return out

    def search_device(self, ptr1: int, n: int, ptr2: int, m: int, first: Optional[int] = None, last: Optional[int] = None) -> Result:
        """One-shot search from raw ASCII in device memory: one kernel launch (psa_search_device)."""
        first = 0 if first is None else first
        last = n - m + 1 if last is None else last
        out = Result()
        _check(lib.psa_search_device(self._h, C.c_void_p(ptr1), n, C.c_void_p(ptr2), m, first, last, C.byref(out)))
        self.n = self.m = None
        return out

    def raw_retry_count(self) -> int:
        """One-shot searches the one launch did not settle (redone through set_sequences_device + search)."""
        return lib.psa_raw_retry_count(self._h)

    def streamed_count(self) -> int:
        """Host-to-device transfers of this context that were streamed (the encode launch follows the host's copy)."""
        return lib.psa_streamed_count(self._h)

    def stream_retry_count(self) -> int:
        """... and how many of those were redone the plain way (the kernel gave up waiting for the host)."""
        return lib.psa_stream_retry_count(self._h)

    def search_async(self, first: int, last: int):
        _check(lib.psa_search_async(self._h, first, last))

    def search_wait(self) -> Result:
        out = Result()
        _check(lib.psa_search_wait(self._h, C.byref(out)))
        return out

    def search_batch(self, queries: Sequence[bytes], first: Optional[int] = None, last: Optional[int] = None):
        """Equal-length host queries -> list of Result (psa_search_batch)."""
        m = len(queries[0])
        if any(len(q) != m for q in queries):
            raise ValueError("all queries of a batch must have the same length")
        first = 0 if first is None else first
        last = self.n - m + 1 if last is None else last
        out = (Result * len(queries))()
        _check(lib.psa_search_batch(self._h, b"".join(queries), len(queries), m, m, first, last, out))
        self.m = None
        return list(out)

    def search_batch_device(self, ptr: int, nq: int, m: int, stride: int,
                            first: Optional[int] = None, last: Optional[int] = None):
        """Queries already in HBM, `stride` bytes apart (psa_search_batch_device)."""
        first = 0 if first is None else first
        last = self.n - m + 1 if last is None else last
        out = (Result * nq)()
        _check(lib.psa_search_batch_device(self._h, C.c_void_p(ptr), nq, m, stride, first, last, out))
        self.m = None
        return list(out)

    def search_batch_minmax(self, queries: Sequence[bytes], first: Optional[int] = None, last: Optional[int] = None):
        """Equal-length host queries -> (MAX results, MIN results) from one shared count pass."""
        m = len(queries[0])
        if any(len(q) != m for q in queries):
            raise ValueError("all queries of a batch must have the same length")
        first = 0 if first is None else first
        last = self.n - m + 1 if last is None else last
        hi = (Result * len(queries))()
        lo = (Result * len(queries))()
        _check(lib.psa_search_batch_minmax(self._h, b"".join(queries), len(queries), m, m, first, last, hi, lo))
        self.m = None
        return list(hi), list(lo)

    def search_batch_minmax_device(self, ptr: int, nq: int, m: int, stride: int,
                                   first: Optional[int] = None, last: Optional[int] = None):
        """Queries already in HBM (psa_search_batch_minmax_device)."""
        first = 0 if first is None else first
        last = self.n - m + 1 if last is None else last
        hi = (Result * nq)()
        lo = (Result * nq)()
        _check(lib.psa_search_batch_minmax_device(self._h, C.c_void_p(ptr), nq, m, stride, first, last, hi, lo))
        self.m = None
        return list(hi), list(lo)

    def scores(self, first: Optional[int] = None, last: Optional[int] = None):
        first = 0 if first is None else first
        last = self.total_offsets() if last is None else last
        span = last - first
        s = np.empty(span, np.float64)
        ci = np.empty(span, np.int32)
        ch = np.empty(span, np.int32)
        _check(lib.psa_scores(self._h, first, last, s.ctypes.data, ci.ctypes.data, ch.ctypes.data))
        return s, ci, ch

    def debug_counts(self, first: Optional[int] = None, last: Optional[int] = None):
        """Filter engine's exact (n_star, n_colon, n_dot) per offset (libpsa_debug.so only)."""
        if not hasattr(lib, "psa_debug_counts"):
            raise PsaError(PSA_ERR_STATE, "psa_debug_counts is a test hook of libpsa_debug.so (PSA_AMD_LIB), not of libpsa_amd.so")
        first = 0 if first is None else first
        last = self.total_offsets() if last is None else last
        span = last - first
        a, b, c = (np.empty(span, np.uint32) for _ in range(3))
        _check(lib.psa_debug_counts(self._h, first, last, a.ctypes.data, b.ctypes.data, c.ctypes.data))
        return a, b, c

    def debug_rescore(self, offsets) -> np.ndarray:
        """Exact re-score of the given offsets (filter engine's rescore kernel; libpsa_debug.so only)."""
        if not hasattr(lib, "psa_debug_rescore"):
            raise PsaError(PSA_ERR_STATE, "psa_debug_rescore is a test hook of libpsa_debug.so (PSA_AMD_LIB), not of libpsa_amd.so")
        offs = np.ascontiguousarray(offsets, np.int32)
        out = np.empty(len(offs), np.float64)
        _check(lib.psa_debug_rescore(self._h, offs.ctypes.data, len(offs), out.ctypes.data))
        return out

    def fallback_count(self) -> int:
        return lib.psa_fallback_count(self._h)

    def scratch_fallback_count(self) -> int:
        """Searches / batches answered by the sequential engine because their per-chunk counts exceed the context's cap."""
        return lib.psa_scratch_fallback_count(self._h)

    def share_sweep_count(self) -> int:
        """Slices of shared combines (offset shards) a tile's last workgroup had to take over (0 on an unshared GPU)."""
        return lib.psa_share_sweep_count(self._h)

    def share_pause_count(self) -> int:
        """Times the context paused the sharing of combines (64 launches) because its GPU turned out to be shared."""
        return lib.psa_share_pause_count(self._h)

    def scratch_shrink_count(self) -> int:
        """Times a failed allocation of per-chunk counts lowered that cap."""
        return lib.psa_scratch_shrink_count(self._h)

    def duo_retry_count(self) -> int:
        """Batches redone without the batch count kernel after one of its group barriers timed out."""
        return lib.psa_duo_retry_count(self._h)

    def unfused_count(self) -> int:
        """Single filter searches finished by a second launch (psa_rescore_kernel) instead of inside the count kernel."""
        return lib.psa_unfused_count(self._h)

    def reselect_count(self) -> int:
        """Searches whose candidate selection ran twice (overflow against the running bound)."""
        return lib.psa_reselect_count(self._h)

    def set_profiling(self, on: bool):
        _check(lib.psa_set_profiling(self._h, int(on)))

    def last_timing(self, which: int):
        ms = C.c_float()
        name = C.c_char_p()
        _check(lib.psa_last_timing(self._h, which, C.byref(ms), C.byref(name)))
        return ms.value, (name.value or b"").decode()

    def last_kernel_queries(self) -> int:
        """Queries covered by the kernel launch last_timing(1) timed (1 for a single search)."""
        return lib.psa_last_kernel_queries(self._h)

    def profile_totals(self):
        """(sum of search ms, sum of dominant-kernel ms, searches) since set_profiling(True)."""
        a, b, k = C.c_double(), C.c_double(), C.c_int()
        _check(lib.psa_profile_totals(self._h, C.byref(a), C.byref(b), C.byref(k)))
        return a.value, b.value, k.value


class Group:
    """Device group (psa_group_create .. psa_group_destroy): one search sharded over several GPUs of the node,
    per-device records exchanged by one RCCL all-gather (the reference's MPI layer, cpu_funcs.c:49-94)."""

    def __init__(self, devices: Sequence[int], exchange: int = EXCHANGE_RCCL):
        """exchange = EXCHANGE_HOST: the records meet in host memory (no RCCL; a device may be listed several times)."""
        self._h = C.c_void_p()
        arr = (C.c_int * len(devices))(*devices)
        _check(lib.psa_group_create_ex(arr, len(devices), exchange, C.byref(self._h)))
        self.n = self.m = None

    def close(self):
        if getattr(self, "_h", None) and self._h.value and lib is not None:
            lib.psa_group_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def size(self) -> int:
        return lib.psa_group_size(self._h)

    def set_weights(self, weights: Sequence[float], is_max: bool):
        _check(lib.psa_group_set_weights(self._h, _weights(weights), int(bool(is_max))))

    def set_seq1(self, seq1: bytes):
        _check(lib.psa_group_set_seq1(self._h, seq1, len(seq1)))
        self.n = len(seq1)

    def set_seq2(self, seq2: bytes):
        _check(lib.psa_group_set_seq2(self._h, seq2, len(seq2)))
        self.m = len(seq2)

    def search_batch(self, queries: Sequence[bytes], minmax: bool = False,
                     first: Optional[int] = None, last: Optional[int] = None):
        """Equal-length host queries dealt to the group's devices -> list of Result (or (MAX, MIN) lists)."""
        m = len(queries[0])
        if any(len(q) != m for q in queries):
            raise ValueError("all queries of a batch must have the same length")
        first = 0 if first is None else first
        last = self.n - m + 1 if last is None else last
        hi = (Result * len(queries))()
        lo = (Result * len(queries))()
        _check(lib.psa_search_batch_sharded(self._h, b"".join(queries), len(queries), m, m, first, last, hi,
                                            lo if minmax else None))
        return (list(hi), list(lo)) if minmax else list(hi)

    def set_timeout(self, seconds: float):
        """How long a group call waits for the devices before it gives up with PSA_ERR_DEVICE (default 60 s)."""
        _check(lib.psa_group_set_timeout(self._h, float(seconds)))

    def last_timing(self):
        """(issue_us, total_us) of the last group call: host time until every device's work was enqueued / until it returned."""
        a, b = C.c_double(), C.c_double()
        _check(lib.psa_group_last_timing(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def search(self, first: Optional[int] = None, last: Optional[int] = None):
        """-> (folded Result, [per-device Result])"""
        first = 0 if first is None else first
        last = self.n - self.m + 1 if last is None else last
        out = Result()
        shards = (Result * self.size())()
        _check(lib.psa_search_sharded(self._h, first, last, C.byref(out), shards))
        return out, list(shards)


def search(seq1: bytes, seq2: bytes, weights: Sequence[float], is_max: bool,
           first: Optional[int] = None, last: Optional[int] = None, device: int = 0) -> Result:
    """One-shot search on `device` (the body of gpu_run_program)."""
    with Context(device) as ctx:
        ctx.set_weights(weights, is_max)
        ctx.set_seq1(seq1)
        ctx.set_seq2(seq2)
        return ctx.search(first, last)
