#!/usr/bin/env python3
"""Batches of very many queries: the filter engine's batch path (pools of per-query scalars, lists, records) against the
sequential engine, MIN+MAX from one pass included.   python tools/big_nq.py ["n,m,nq" ...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "parallel-sequence-alignment_amd"))
import torch  # noqa: F401,E402
import psa_amd as psa  # noqa: E402

W = [1.5, 2.6, 0.3, 0.2]
cases = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or [
    (100_000, 1000, 5000), (1_000_000, 10_000, 2048), (20_000, 700, 20_000), (9711, 2131, 4097)]
key = lambda r: (r.score, r.offset, r.char_offset, r.ch)
for n, m, nq in cases:
    rng = np.random.default_rng(nq)
    s1 = rng.integers(65, 91, size=n, dtype=np.uint8).tobytes()
    qs = [rng.integers(65, 91, size=m, dtype=np.uint8).tobytes() for _ in range(nq)]
    try:
        with psa.Context(0) as cf, psa.Context(0) as cs:
            cf.set_engine(psa.ENGINE_FILTER)
            cs.set_engine(psa.ENGINE_SEQUENTIAL)
            for c in (cf, cs):
                c.set_weights(W, True)
                c.set_seq1(s1)
            cf.search_batch(qs[:8])
            t0 = time.time()
            hi, lo = cf.search_batch_minmax(qs)
            dt = time.time() - t0
            one = cf.search_batch(qs)
            ref_hi = cs.search_batch(qs)
            cs.set_weights(W, False)
            ref_lo = cs.search_batch(qs)
            assert [key(r) for r in hi] == [key(r) for r in ref_hi] == [key(r) for r in one]
            assert [key(r) for r in lo] == [key(r) for r in ref_lo]
            print(f"n={n} m={m} nq={nq}: ok; MIN+MAX batch {1e3 * dt:.1f} ms ({nq * (n - m + 1) * m / dt:.3g} pairs/s per mode); "
                  f"fallbacks {cf.fallback_count()} duo retries {cf.duo_retry_count()}", flush=True)
    except Exception as e:            # noqa: BLE001
        print(f"n={n} m={m} nq={nq}: FAILED: {type(e).__name__}: {e}", flush=True)
