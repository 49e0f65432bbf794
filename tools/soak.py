#!/usr/bin/env python3
"""Randomised soak: filter engine (single, batch, MIN+MAX batch) against the sequential engine on random shapes,
weights, alphabets and ranges for a given number of seconds.   python tools/soak.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "parallel-sequence-alignment_amd"))
import torch  # noqa: F401,E402  (its HIP runtime first, see tests/conftest.py)
import psa_amd as psa  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
key = lambda r: (r.score, r.offset, r.char_offset, r.ch)
alphabets = [b"ABCDEFGHIJKLMNOPQRSTUVWXYZ", b"ACDEFGHIKLMNPQRSTVWY", b"ACGT", b"AL-", b"NDEQKSTA-MILV", b"AB"]
t_end = time.time() + budget
t_note = time.time() + 60.0
cases = 0
with psa.Context(0) as cf, psa.Context(0) as cs, psa.Context(0) as ca:      # ca: PSA_ENGINE_AUTO
    cf.set_engine(psa.ENGINE_FILTER)
    cs.set_engine(psa.ENGINE_SEQUENTIAL)
    while time.time() < t_end:
        big = os.environ.get("SOAK_BIG") == "1"            # only shapes that keep the GPU busy
        n = int(rng.choice([250_000, 1_000_000, 2_500_000] if big else [3000, 40_000, 250_000, 1_000_000, 2_500_000]))
        m = int(min(n - 1, rng.choice([640, 1025, 2500, 9999, 11_264, 11_265, 13_000, 20_000] if big else
                                      [1, 7, 33, 100, 640, 1025, 2500, 9999, 11_264, 11_265, 13_000, 20_000])))
        nq = int(rng.integers(2, 40)) if big else int(rng.integers(1, 6))
        ab = np.frombuffer(alphabets[int(rng.integers(len(alphabets)))], np.uint8)
        s1 = rng.choice(ab, n).astype(np.uint8).tobytes()
        qs = [rng.choice(ab, m).astype(np.uint8).tobytes() for _ in range(nq)]
        if rng.random() < 0.5 and n - m > 1:
            o = int(rng.integers(0, n - m))
            qs[0] = s1[o:o + m]
        w = [float(x) for x in rng.choice([0.0, 0.1, 0.25, 1.0, 1.5, 2.6, 0.3, 0.2, 7.0, 1e-3, 1e3], 4)]
        total = n - m + 1
        first = int(rng.integers(0, max(1, total // 2)))
        last = int(rng.integers(first + 1, total + 1))
        if os.environ.get("SOAK_SHARD") == "1":            # offset shards: few tiles, several char chunks (the shared combine)
            last = min(total, first + int(rng.choice([1, 37, 4064, 4065, 30_000, 123_750, 247_500, 400_000])))
        for c in (cf, cs, ca):
            c.set_weights(w, True)
            c.set_seq1(s1)
        hi, lo = cf.search_batch_minmax(qs, first, last)
        one = cf.search_batch(qs, first, last)
        cf.set_seq2(qs[-1])
        single = cf.search(first, last)
        assert key(single) == key(one[-1]), ("single vs batch", n, m, nq, w, first, last)
        ca.set_sequences(s1, qs[-1])                        # both sequences by one encode launch
        assert key(ca.search(first, last)) == key(single), ("auto engine", n, m, nq, w, first, last)
        d1 = torch.frombuffer(bytearray(s1), dtype=torch.uint8).cuda()
        d2 = torch.frombuffer(bytearray(qs[-1]), dtype=torch.uint8).cuda()
        torch.cuda.synchronize()
        for c in (cf, ca):                                  # one-shot search from raw ASCII: one launch
            assert key(c.search_device(d1.data_ptr(), n, d2.data_ptr(), m, first, last)) == key(single), ("one-shot", n, m, w, first, last)
            c.set_seq1(s1)
        assert [key(r) for r in one] == [key(r) for r in hi], ("batch vs minmax", n, m, nq, w, first, last)
        for mode, got in ((True, hi), (False, lo)):
            cs.set_weights(w, mode)
            for q, r in zip(qs, got):
                cs.set_seq2(q)
                assert key(r) == key(cs.search(first, last)), ("vs sequential", n, m, nq, w, first, last, mode)
        cases += 1
        if time.time() > t_note:                      # a sign of life every minute (long silent runs get killed)
            print(f"soak: {cases} cases so far", flush=True)
            t_note = time.time() + 60.0
    print(f"soak: {cases} random cases ok; filter context: {cf.fallback_count()} sequential-engine fallbacks (ties), "
          f"{cf.reselect_count()} reselects, {cf.unfused_count()} second launches (> 16 survivors or a long Seq2), "
          f"{cf.duo_retry_count()} retried batches (a group barrier gave up: expected 0), "
          f"{cf.raw_retry_count()} one-shot searches redone through the two-launch path")
    assert cf.duo_retry_count() == 0 and ca.duo_retry_count() == 0
