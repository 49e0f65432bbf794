#!/usr/bin/env python3
"""One rank's share of an N-way offset shard of the 1M x 10k search on this GPU (N = 1, 2, 4, 8): blocking latency and
the launch's duration from the library's HIP events.  PSA_FILTER_CHUNKS=k (debug library) forces the chunk count."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "parallel-sequence-alignment_amd"))
import psa_amd as psa
n, m = 1_000_000, 10_000
t1 = torch.from_numpy(psa.generate_sequence_np(1002, n)).cuda(); t2 = torch.from_numpy(psa.generate_sequence_np(2002, m)).cuda()
ctx = psa.Context(0, torch.cuda.current_stream().cuda_stream); ctx.set_weights([1.5,2.6,0.3,0.2], True)
ctx.set_sequences_device(t1.data_ptr(), n, t2.data_ptr(), m)
total = n - m + 1
for w in (1, 2, 4, 8):
    lo, hi = psa.shard_range(total, w, 0)
    ctx.set_profiling(False)
    lat = []
    for k in range(30):
        torch.cuda.synchronize(); t0 = time.perf_counter(); ctx.search(lo, hi); lat.append(time.perf_counter() - t0)
    ctx.set_profiling(True)
    for k in range(10): ctx.search(lo, hi)
    tot, kern, cnt = ctx.profile_totals()
    print(f"1/{w} shard: blocking {1e6*sorted(lat[5:])[12]:.1f} us, kernel (HIP events) {1e3*kern/cnt:.1f} us")
