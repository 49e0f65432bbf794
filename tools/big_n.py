#!/usr/bin/env python3
"""Very long Seq1 (towards the int limit of the C ABI; 288 GB of HBM hold it easily): the filter engine on the whole
range against the sequential engine on sub-ranges, a planted copy of Seq2 found, both modes.
    python tools/big_n.py [n ...]      (default: 100M, 600M, 2^31 - 64)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "parallel-sequence-alignment_amd"))
import torch  # noqa: F401,E402
import psa_amd as psa  # noqa: E402

W = [1.5, 2.6, 0.3, 0.2]
sizes = [int(x) for x in sys.argv[1:]] or [100_000_000, 600_000_000, 2**31 - 64]
key = lambda r: (r.score, r.offset, r.char_offset, r.ch)
m = 10_000
for n in sizes:
    t0 = time.time()
    rng = np.random.default_rng(n % 1000)
    s1 = rng.integers(65, 91, size=n, dtype=np.uint8)
    s2 = rng.integers(65, 91, size=m, dtype=np.uint8)
    total = n - m + 1
    plant = total - 1_234_567
    s1[plant:plant + m] = s2
    s1b, s2b = s1.tobytes(), s2.tobytes()
    del s1
    print(f"n = {n}: inputs generated in {time.time() - t0:.1f} s", flush=True)
    try:
        with psa.Context(0) as cf, psa.Context(0) as cs:
            cf.set_engine(psa.ENGINE_FILTER)
            cs.set_engine(psa.ENGINE_SEQUENTIAL)
            for c in (cf, cs):
                c.set_seq1(s1b)
                c.set_seq2(s2b)
            for is_max in (True, False):
                for c in (cf, cs):
                    c.set_weights(W, is_max)
                t1 = time.time()
                whole = cf.search()
                dt = time.time() - t1
                print(f"  {'MAX' if is_max else 'MIN'} whole range: offset {whole.offset} score {whole.score!r} in {1e3 * dt:.1f} ms "
                      f"({total * m / dt:.3g} pairs/s); fallbacks {cf.fallback_count()}", flush=True)
                if is_max:
                    assert whole.offset == plant, (whole.offset, plant)
                lo, hi = total - 3_000_000, total
                a, b = cf.search(lo, hi), cs.search(lo, hi)
                assert key(a) == key(b), ("tail sub-range", key(a), key(b))
                lo, hi = max(0, whole.offset - 1_000_000), min(total, whole.offset + 1_000_000)
                b = cs.search(lo, hi)
                assert key(b) == key(whole), ("around the winner", key(whole), key(b))
                a, b = cf.search(0, 2_000_000), cs.search(0, 2_000_000)
                assert key(a) == key(b), ("head sub-range", key(a), key(b))
        print(f"n = {n}: ok ({time.time() - t0:.1f} s)", flush=True)
    except Exception as e:            # noqa: BLE001 — report and go on to the next size
        print(f"n = {n}: FAILED: {type(e).__name__}: {e}", flush=True)
