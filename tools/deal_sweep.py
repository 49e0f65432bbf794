#!/usr/bin/env python3
"""Shares of a SIMD's four waves in the single-search count kernel (debug library knob PSA_DEAL=a,b,c,d permille by age,
oldest first): blocking search and kernel time of the 1M x 10k search for every deal given on the command line.
  PSA_AMD_LIB=.../libpsa_debug.so python tools/deal_sweep.py 340,270,250,140 300,270,250,180 ..."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "parallel-sequence-alignment_amd"))
import psa_amd as psa  # noqa: E402

n, m = 1_000_000, 10_000
t1 = torch.from_numpy(psa.generate_sequence_np(1002, n)).cuda()
t2 = torch.from_numpy(psa.generate_sequence_np(2002, m)).cuda()
for rep in range(3):                      # (the first pass warms the GPU up: read the later ones)
    for deal in sys.argv[1:]:
        os.environ["PSA_DEAL"] = deal
        ctx = psa.Context(0, torch.cuda.current_stream().cuda_stream)
        ctx.set_weights([1.5, 2.6, 0.3, 0.2], True)
        ctx.set_sequences_device(t1.data_ptr(), n, t2.data_ptr(), m)
        lat = []
        for k in range(40):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            r = ctx.search()
            lat.append(time.perf_counter() - t0)
        ctx.set_profiling(True)
        for k in range(10):
            ctx.search()
        tot, kern, cnt = ctx.profile_totals()
        print(f"deal {deal:20s} blocking {1e6 * sorted(lat[5:])[17]:6.1f} us  kernel {1e3 * kern / cnt:6.1f} us  ({r.offset})", flush=True)
        ctx.close()
