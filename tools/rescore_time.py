#!/usr/bin/env python3
"""Time of one candidate's exact re-score (psa_rescore_kernel) at 1M x 10k: psa_debug_rescore with 2048 and
64 candidates on the 64-workgroup grid; the difference is 31 candidates per workgroup back to back."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "parallel-sequence-alignment_amd"))
import psa_amd  # noqa: E402

n, m = 1_000_000, 10_000
ctx = psa_amd.Context()
ctx.set_weights([1.5, 2.6, 0.3, 0.2], True)
ctx.set_seq1(psa_amd.generate_sequence(1002, n))
ctx.set_seq2(psa_amd.generate_sequence(2002, m))
rng = np.random.default_rng(5)
offs = rng.integers(0, n - m + 1, 2048).astype(np.int32)


def timed(k, reps=20):
    ctx.debug_rescore(offs[:k])
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.debug_rescore(offs[:k])
    return (time.perf_counter() - t0) / reps


a, b = timed(64), timed(2048)
print(f"64 candidates {a*1e6:.1f} us/call, 2048 candidates {b*1e6:.1f} us/call -> {(b-a)/31*1e6:.2f} us per candidate")
