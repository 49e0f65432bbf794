#!/usr/bin/env python3
"""Randomised soak of the HIP library AGAINST THE CPU ORACLE (test infrastructure: it lives under tests/ because only
tests may use oracle/; tools/soak.py compares the two HIP engines with each other on larger shapes).

    python tests/soak_oracle.py [seconds] [seed]

Every case: random lengths (up to 1M x 640 / 120k x 4000: what the oracle's OpenMP loop answers in about a second),
alphabet, weights, mode and sub-range; the oracle's result (cpu_funcs.c:222-300 restated) must equal, bit for bit,
  * the filter engine's blocking search, the AUTO engine's, the sequential engine's,
  * the one-shot search from raw ASCII in device memory (one launch),
  * the batch path (MAX and MIN from one pass) for the query and its reverse,
  * a group of three contexts over the offset shard (host exchange, launcher threads) and its query shard,
  * gpu_run_program on a ProgramData holding the pair (when it fits the stock 10001 / 5001 layout)."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "parallel-sequence-alignment_amd"))
import torch  # noqa: E402,F401  (its HIP runtime first, see tests/conftest.py)
import psa_amd as psa  # noqa: E402
from oracle import oracle  # noqa: E402

oracle.build()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
threads = min(16, os.cpu_count() or 1)
alphabets = [b"ABCDEFGHIJKLMNOPQRSTUVWXYZ", b"ACDEFGHIKLMNPQRSTVWY", b"ACGT", b"AL-", b"NDEQKSTA-MILV", b"AB", b"A"]
shapes = [(3000, 1), (3000, 7), (9711, 2131), (40_000, 33), (40_000, 1025), (120_000, 4000), (250_000, 640),
          (1_000_000, 640), (60_000, 2500), (10_000, 4999), (5000, 4999), (700, 700)]


def same(res, want, what, case):
    got = (res.score, res.offset, res.char_offset, chr(res.ch) if res.ch else "")
    assert got == tuple(want), (what, got, tuple(want), case)


t_end = time.time() + budget
t_note = time.time() + 60.0
cases = 0
with psa.Context(0) as cf, psa.Context(0) as cs, psa.Context(0) as ca, psa.Group([0, 0, 0], psa.EXCHANGE_HOST) as grp:
    cf.set_engine(psa.ENGINE_FILTER)
    cs.set_engine(psa.ENGINE_SEQUENTIAL)
    while time.time() < t_end:
        n, m = shapes[int(rng.integers(len(shapes)))]
        if rng.random() < 0.5:                       # not only the listed lengths
            m = int(rng.integers(1, m + 1))
            n = int(rng.integers(m, n + 1))
        ab = np.frombuffer(alphabets[int(rng.integers(len(alphabets)))], np.uint8)
        s1 = rng.choice(ab, n).astype(np.uint8).tobytes()
        s2 = rng.choice(ab, m).astype(np.uint8).tobytes()
        if rng.random() < 0.3 and n > m:             # Seq2 cut out of Seq1: an exact match somewhere
            o = int(rng.integers(0, n - m))
            s2 = s1[o:o + m]
        w = [float(x) for x in rng.choice([0.0, 0.1, 0.25, 1.0, 1.5, 2.6, 0.3, 0.2, 7.0, 13.7, 1e-3, 1e3], 4)]
        is_max = bool(rng.integers(2))
        total = n - m + 1
        first = int(rng.integers(0, total))
        last = int(rng.integers(first + 1, total + 1))
        case = (n, m, bytes(ab), w, is_max, first, last, seed, cases)
        want_full = oracle.search(s1, s2, w, is_max, threads=threads)
        want_sub = oracle.search(s1, s2, w, is_max, first, last, threads=threads)
        want_rev = oracle.search(s1, s2[::-1], w, is_max, first, last, threads=threads)
        want_other = oracle.search(s1, s2, w, not is_max, first, last, threads=threads)
        for c, name in ((cf, "filter"), (cs, "sequential"), (ca, "auto")):
            c.set_weights(w, is_max)
            c.set_sequences(s1, s2)
            same(c.search(), want_full, name + " full", case)
            same(c.search(first, last), want_sub, name + " sub-range", case)
        d1 = torch.frombuffer(bytearray(s1), dtype=torch.uint8).cuda()
        d2 = torch.frombuffer(bytearray(s2), dtype=torch.uint8).cuda()
        torch.cuda.synchronize()
        same(cf.search_device(d1.data_ptr(), n, d2.data_ptr(), m, first, last), want_sub, "one-shot raw", case)
        cf.set_seq1(s1)
        got = cf.search_batch([s2, s2[::-1]], first, last)
        same(got[0], want_sub, "batch[0]", case)
        same(got[1], want_rev, "batch[1]", case)
        hi, lo = cf.search_batch_minmax([s2, s2[::-1], s2], first, last)
        same((hi if is_max else lo)[2], want_sub, "minmax, own mode", case)
        same((lo if is_max else hi)[0], want_other, "minmax, other mode", case)
        grp.set_weights(w, is_max)
        grp.set_seq1(s1)
        grp.set_seq2(s2)
        folded, _ = grp.search(first, last)
        same(folded, want_sub, "group, offset shard", case)
        gq = grp.search_batch([s2, s2[::-1], s2, s2[::-1]], False, first, last)
        same(gq[2], want_sub, "group, query shard [2]", case)
        same(gq[3], want_rev, "group, query shard [3]", case)
        if n <= 10000 and m <= 5000:
            # the stock ProgramData layout (def.h:35-36: is_max@0, weights@8, seq1@40, seq2@10041, 15048 bytes)
            buf = bytearray(15048)
            buf[0:4] = int(is_max).to_bytes(4, "little")
            buf[8:40] = np.array(w, np.float64).tobytes()
            buf[40:40 + n] = s1
            buf[10041:10041 + m] = s2
            pd = (C.c_char * len(buf)).from_buffer(buf)
            mu = psa.Mutant()
            score = psa.lib.gpu_run_program(C.addressof(pd), C.byref(mu), first, last)
            assert (score, mu.offset, mu.char_offset, mu.ch.decode().strip("\0")) == tuple(want_sub), ("gpu_run_program", case)
        cases += 1
        if time.time() > t_note:
            print(f"{cases} cases so far, all equal to the oracle", flush=True)
            t_note = time.time() + 60.0
    print(f"soak vs oracle: {cases} cases in {budget:.0f} s, seed {seed}: every engine and entry point equals the oracle; "
          f"filter context: fallbacks {cf.fallback_count()} reselects {cf.reselect_count()} unfused {cf.unfused_count()}")
