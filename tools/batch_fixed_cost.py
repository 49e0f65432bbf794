#!/usr/bin/env python3
"""Fixed cost of a batch call: time of psa_search_batch_device (+ the Seq1 encode, as in bench.py's step)
for nq = 1..64 queries of 1M x 10k, and the linear fit  t = a + b * nq."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "parallel-sequence-alignment_amd"))
import psa_amd as psa  # noqa: E402

n, m = 1_000_000, 10_000
s1 = torch.from_numpy(psa.generate_sequence_np(1002, n)).cuda()
qs = torch.from_numpy(np.stack([psa.generate_sequence_np(2002 + q, m) for q in range(64)])).cuda()
ctx = psa.Context(0)
ctx.set_weights([1.5, 2.6, 0.3, 0.2], True)
xs, ys = [], []
for nq in (1, 2, 4, 8, 16, 32, 64):
    def step():
        ctx.set_seq1_device(s1.data_ptr(), n)
        return ctx.search_batch_device(qs.data_ptr(), nq, m, m)
    for _ in range(3):
        step()
    ts = []
    for _ in range(15):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        step()
        ts.append(time.perf_counter() - t0)
    t = sorted(ts)[len(ts) // 2]
    xs.append(nq)
    ys.append(t * 1e3)
    print(f"nq {nq:3d}: {t*1e3:.4f} ms")
b, a = np.polyfit(xs[2:], ys[2:], 1)
print(f"fit over nq >= 4: fixed {a:.4f} ms + {b:.4f} ms per query")
