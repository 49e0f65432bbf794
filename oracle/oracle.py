"""ctypes binding of the CPU oracle (oracle/psa_oracle.c).

TEST INFRASTRUCTURE, NOT PRODUCT CODE: only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "build", "libpsa_oracle.so")
CLI_PATH = os.path.join(_HERE, "build", "psa_oracle_cli")


def build(force: bool = False) -> None:
    """gcc build of the oracle (a few seconds)."""
    if force or not (os.path.exists(LIB_PATH) and os.path.exists(CLI_PATH)):
        subprocess.run(["make", "-C", _HERE, "all"], check=True, stdout=subprocess.DEVNULL)


class Mutant(C.Structure):
    _fields_ = [("offset", C.c_int), ("char_offset", C.c_int), ("ch", C.c_char)]


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB_PATH)
        dp = C.POINTER(C.c_double)
        L.orc_pair_sign.argtypes = [C.c_char, C.c_char]
        L.orc_pair_sign.restype = C.c_char
        L.orc_table_sign.argtypes = [C.c_char, C.c_char]
        L.orc_table_sign.restype = C.c_char
        L.orc_weight.argtypes = [C.c_char, dp]
        L.orc_weight.restype = C.c_double
        L.orc_substitute.argtypes = [C.c_char, C.c_char, dp, C.c_int]
        L.orc_substitute.restype = C.c_char
        L.orc_is_swapable.argtypes = [C.POINTER(Mutant), C.POINTER(Mutant), C.c_double, C.c_double, C.c_int]
        L.orc_score_offset.argtypes = [C.c_char_p, C.c_char_p, dp, C.c_int, C.c_int, C.POINTER(Mutant)]
        L.orc_score_offset.restype = C.c_double
        L.orc_score_and_signs.argtypes = [C.c_char_p, C.c_char_p, dp, C.c_int, C.c_char_p]
        L.orc_score_and_signs.restype = C.c_double
        L.orc_search_range.argtypes = [C.c_char_p, C.c_char_p, dp, C.c_int, C.c_int, C.c_int, C.POINTER(Mutant)]
        L.orc_search_range.restype = C.c_double
        L.orc_search_omp.argtypes = [C.c_char_p, C.c_char_p, dp, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(Mutant)]
        L.orc_search_omp.restype = C.c_double
        L.orc_search_ranks.argtypes = [C.c_char_p, C.c_char_p, dp, C.c_int, C.c_int, C.c_int, C.POINTER(Mutant)]
        L.orc_search_ranks.restype = C.c_double
        L.orc_scores.argtypes = [C.c_char_p, C.c_char_p, dp, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_scores.restype = None
        _lib = L
    return _lib


def _w(weights):
    return (C.c_double * 4)(*[float(x) for x in weights])


def _z(seq) -> bytes:
    return bytes(seq) + b"\0"


def pair_sign(a: str, b: str) -> str:
    return lib().orc_pair_sign(a.encode(), b.encode()).decode()


def table_sign(a: str, b: str) -> str:
    r = lib().orc_table_sign(a.encode(), b.encode())
    return r.decode() if r != b"\0" else ""


def weight(sign: str, weights) -> float:
    return lib().orc_weight(sign.encode(), _w(weights))


def substitute(c1: str, c2: str, weights, is_max: bool) -> str:
    r = lib().orc_substitute(c1.encode(), c2.encode(), _w(weights), int(is_max))
    return r.decode() if r != b"\0" else ""


def score_offset(seq1: bytes, seq2: bytes, weights, is_max: bool, offset: int):
    m = Mutant()
    s = lib().orc_score_offset(_z(seq1), _z(seq2), _w(weights), int(is_max), offset, C.byref(m))
    return s, m.char_offset, (m.ch.decode() if m.ch != b"\0" else "")


def score_and_signs(seq1: bytes, seq2: bytes, weights, offset: int):
    buf = C.create_string_buffer(len(seq2) + 1)
    s = lib().orc_score_and_signs(_z(seq1), _z(seq2), _w(weights), offset, buf)
    return s, buf.value.decode()


def search(seq1: bytes, seq2: bytes, weights, is_max: bool, first=None, last=None, threads: int = 1):
    """(score, offset, char_offset, ch) of the best mutant over [first,last)."""
    total = len(seq1) - len(seq2) + 1
    first = 0 if first is None else first
    last = total if last is None else last
    m = Mutant()
    s = lib().orc_search_omp(_z(seq1), _z(seq2), _w(weights), int(is_max), first, last, threads, C.byref(m))
    return s, m.offset, m.char_offset, (m.ch.decode() if m.ch != b"\0" else "")


def search_ranks(seq1: bytes, seq2: bytes, weights, is_max: bool, nproc: int, threads: int = 1):
    m = Mutant()
    s = lib().orc_search_ranks(_z(seq1), _z(seq2), _w(weights), int(is_max), nproc, threads, C.byref(m))
    return s, m.offset, m.char_offset, (m.ch.decode() if m.ch != b"\0" else "")


def scores(seq1: bytes, seq2: bytes, weights, is_max: bool, first=None, last=None):
    """Per-offset (scores, char_offsets, chs) arrays (OpenMP over offsets)."""
    total = len(seq1) - len(seq2) + 1
    first = 0 if first is None else first
    last = total if last is None else last
    span = last - first
    s = np.empty(span, np.float64)
    ci = np.empty(span, np.int32)
    ch = np.empty(span, np.uint8)
    lib().orc_scores(_z(seq1), _z(seq2), _w(weights), int(is_max), first, last,
                     s.ctypes.data, ci.ctypes.data, ch.ctypes.data)
    return s, ci, ch


def read_records(path: str):
    """All stacked 7-token records of an input.txt-format file
    (weights, seq1, seq2, is_max); same tokenisation as fscanf %lf/%s."""
    toks = open(path).read().split()
    out = []
    for k in range(0, len(toks) - 6, 7):
        w = [float(t) for t in toks[k:k + 4]]
        out.append((w, toks[k + 4].encode(), toks[k + 5].encode(), toks[k + 6] == "maximum"))
    return out
