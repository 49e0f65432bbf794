#!/usr/bin/env python3
"""Timeline of the two wave groups of one workgroup of psa_count_duo_kernel (debug build -DPSA_PHASE_TIMING):
  PSA_AMD_LIB=.../libpsa_phase.so python tools/duo_timing.py"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "parallel-sequence-alignment_amd"))
import psa_amd as psa  # noqa: E402

n, m, nq = 1_000_000, 10_000, 16
s1 = torch.from_numpy(psa.generate_sequence_np(1002, n)).cuda()
qs = torch.from_numpy(np.stack([psa.generate_sequence_np(2002 + q, m) for q in range(nq)])).cuda()
ctx = psa.Context(0)
ctx.set_weights([1.5, 2.6, 0.3, 0.2], True)
ctx.set_seq1_device(s1.data_ptr(), n)
for _ in range(2):
    ctx.search_batch_device(qs.data_ptr(), nq, m, m)
buf = (C.c_ulonglong * (1024 * 8))()
psa.lib.psa_debug_phase_ts.restype = C.c_int
assert psa.lib.psa_debug_phase_ts(buf) == 0
ts = np.frombuffer(buf, dtype=np.uint64).reshape(1024, 8).astype(np.int64)
t0 = ts[600][0]
print("times in us from group 0's first main-loop start; per query: main start, main end (wave 0), merged, expanded, done")
for g in range(2):
    for qi in range(nq // 2):
        row = ts[600 + g * 16 + qi]
        print(f"group {g} query {2*qi+g:2d}: " + "  ".join(f"{(row[k]-t0)/100:8.2f}" for k in range(5)))
