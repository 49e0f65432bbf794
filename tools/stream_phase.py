#!/usr/bin/env python3
"""GPU-side timeline of the streamed encode launch (libpsa_phase.so: -DPSA_PHASE_TIMING): per workgroup, when it
started, when the host's publication let it go, when its symbols were encoded (100 MHz clock, us from the first start)."""
import ctypes as C
import os
import sys

import numpy as np
import torch  # noqa: F401

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "parallel-sequence-alignment_amd"))
import psa_amd as psa  # noqa: E402

n, m = 1_000_000, 10_000
h1, h2 = psa.generate_sequence(1002, n), psa.generate_sequence(2002, m)
ctx = psa.Context(0)
ctx.set_weights([1.5, 2.6, 0.3, 0.2], True)
for _ in range(4):
    ctx.set_sequences(h1, h2)
buf = (C.c_ulonglong * (1024 * 8))()
psa.lib.psa_debug_phase_ts.restype = C.c_int
assert psa.lib.psa_debug_phase_ts(buf) == 0
ts = np.frombuffer(buf, dtype=np.uint64).reshape(1024, 8).astype(np.int64)
nb = (n + 49152 + 63) // 64 * 64 // 32 // 256 + 2        # workgroup 0 is the poller
ts = ts[:nb]
t0 = ts[:, 0].min()
for b in list(range(0, nb, 8)) + [nb - 1]:
    print(f"workgroup {b:4d} (bytes from {max(b - 1, 0) * 8192:8d}): start {(ts[b,0]-t0)/100:6.2f}  released {(ts[b,1]-t0)/100:6.2f}  encoded {(ts[b,2]-t0)/100:6.2f} us")
ctx.close()
