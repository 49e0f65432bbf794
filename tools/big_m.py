#!/usr/bin/env python3
"""Very long Seq2 (beyond what the filter engine's chunking plans: 24 chunks x 15,872 chars): which engine answers, how
long it takes, and that the AUTO engine, the filter engine (when it can plan) and the sequential engine agree.
    python tools/big_m.py ["n,m" ...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "parallel-sequence-alignment_amd"))
import torch  # noqa: F401,E402
import psa_amd as psa  # noqa: E402

W = [1.5, 2.6, 0.3, 0.2]
cases = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or [
    (1_000_000, 100_000), (1_000_000, 380_000), (1_000_000, 400_000), (6_000_000, 5_000_000), (3_000_000, 3_000_000)]
key = lambda r: (r.score, r.offset, r.char_offset, r.ch)
for n, m in cases:
    rng = np.random.default_rng(n % 1000 + m % 1000)
    s1 = rng.integers(65, 91, size=n, dtype=np.uint8)
    s2 = rng.integers(65, 91, size=m, dtype=np.uint8)
    total = n - m + 1
    if total > 10:                      # a near-copy somewhere: the winner is not an accident of noise
        at = total // 3
        s2[:] = s1[at:at + m]
        s2[::7] = rng.integers(65, 91, size=len(s2[::7]), dtype=np.uint8)
    s1b, s2b = s1.tobytes(), s2.tobytes()
    try:
        with psa.Context(0) as ca, psa.Context(0) as cs, psa.Context(0) as cf:
            cs.set_engine(psa.ENGINE_SEQUENTIAL)
            cf.set_engine(psa.ENGINE_FILTER)
            for is_max in (True, False):
                out = []
                for c, name in ((ca, "auto"), (cs, "sequential"), (cf, "filter")):
                    c.set_weights(W, is_max)
                    c.set_sequences(s1b, s2b)
                    c.search()
                    t0 = time.time()
                    r = c.search()
                    dt = time.time() - t0
                    out.append(key(r))
                    print(f"  n={n} m={m} {'MAX' if is_max else 'MIN'} {name:10s}: offset {r.offset} score {r.score!r} "
                          f"{1e3 * dt:.2f} ms ({total * m / dt:.3g} pairs/s) fallbacks {c.fallback_count()}", flush=True)
                assert out[0] == out[1] == out[2], out
        print(f"n={n} m={m}: ok", flush=True)
    except Exception as e:            # noqa: BLE001
        print(f"n={n} m={m}: FAILED: {type(e).__name__}: {e}", flush=True)
