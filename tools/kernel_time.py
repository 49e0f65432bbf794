#!/usr/bin/env python3
"""Duration of ONE launch of the batch count kernel (32 queries, 1M x 10k) and of the single-search count
kernel, from the library's HIP events.  For A/B runs of library variants on one GPU:
  PSA_AMD_LIB=.../libpsa_<variant>.so python tools/kernel_time.py [reps]
Results are not checked (experiment builds may compute garbage)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "parallel-sequence-alignment_amd"))
import psa_amd as psa  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
n, m, nq = 1_000_000, 10_000, 32
s1 = torch.from_numpy(psa.generate_sequence_np(1002, n)).cuda()
qs = torch.from_numpy(np.stack([psa.generate_sequence_np(2002 + q, m) for q in range(nq)])).cuda()
ctx = psa.Context(0)
ctx.set_engine(psa.ENGINE_FILTER)
ctx.set_weights([1.5, 2.6, 0.3, 0.2], True)
ctx.set_seq1_device(s1.data_ptr(), n)
ctx.set_profiling(True)
duo, single = [], []
for k in range(reps + 1):
    try:
        ctx.search_batch_device(qs.data_ptr(), nq, m, m)
        t, name = ctx.last_timing(1)
        if k:
            duo.append(t)
    except psa.PsaError as e:
        print("batch:", e)
for k in range(reps + 1):
    ctx.set_seq2_device(qs[0].data_ptr(), m)
    try:
        ctx.search()
        t, name1 = ctx.last_timing(1)
        if k:
            single.append(t)
    except psa.PsaError as e:
        print("single:", e)
tag = os.path.basename(os.environ.get("PSA_AMD_LIB", "libpsa_amd.so"))
print(f"{tag:28s} duo(32q) {np.median(duo):.4f} ms = {1e3*np.median(duo)/32:.2f} us/query   single {1e3*np.median(single):.2f} us")
ctx.close()
