#!/usr/bin/env python3
"""Latency of a ONE-SHOT search (psa_search_device: raw ASCII in HBM, one launch) over a SMALL offset range under a LONG
Seq1: the launch's converter workgroups validate the Seq1 bytes outside the tiles' windows, and the tiles wait for them —
with a fixed 16 converters that wait grew with |Seq1|; now one converter per 128 KB of those bytes, as far as
the spare CUs go.   python tools/subrange_oneshot.py [log2_n=27] [m=1000] [range=100000]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "parallel-sequence-alignment_amd"))
import psa_amd as psa
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 27
m = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
span = int(sys.argv[3]) if len(sys.argv) > 3 else 100_000
n = 1 << logn
t1 = torch.randint(65, 91, (n,), dtype=torch.uint8, device="cuda", generator=torch.Generator(device="cuda").manual_seed(5))
t2 = torch.randint(65, 91, (m,), dtype=torch.uint8, device="cuda", generator=torch.Generator(device="cuda").manual_seed(6))
torch.cuda.synchronize()
with psa.Context(0) as c:
    c.set_weights([1.5, 2.6, 0.3, 0.2], True)
    for lo in (0, n // 2, n - m + 1 - span):
        ts = []
        for k in range(24):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            r = c.search_device(t1.data_ptr(), n, t2.data_ptr(), m, lo, lo + span)
            ts.append(time.perf_counter() - t0)
        print(f"n = 2^{logn}, m = {m}, offsets [{lo}, {lo + span}): one-shot {1e6 * sorted(ts[4:])[10]:8.1f} us  (offset {r.offset}; two-launch retries {c.raw_retry_count()})")
    c.set_sequences_device(t1.data_ptr(), n, t2.data_ptr(), m)
    ts = []
    for k in range(24):
        torch.cuda.synchronize(); t0 = time.perf_counter(); c.search(n // 2, n // 2 + span); ts.append(time.perf_counter() - t0)
    print(f"  the same range with the sequences set beforehand (encode launch not timed): {1e6 * sorted(ts[4:])[10]:8.1f} us")
