#!/usr/bin/env python3
"""Blocking-search latency of both engines over a grid of shapes (fits the AUTO cost model, psa_device.hip use_filter)."""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "parallel-sequence-alignment_amd"))
import psa_amd as psa
shapes = [(1000, 300), (3000, 1000), (5000, 2000), (9711, 2131), (20000, 500), (20000, 1000), (50000, 200), (50000, 1000),
          (100000, 100), (100000, 300), (100000, 1000), (300000, 100), (300000, 300), (1000000, 30), (1000000, 100), (1000000, 1000),
          (1000000, 3000), (20000, 10000)]
with psa.Context(0) as c:
    c.set_weights([1.5, 2.6, 0.3, 0.2], True)
    for n, m in shapes:
        s1, s2 = psa.generate_sequence(5, n), psa.generate_sequence(6, m)
        c.set_seq1(s1); c.set_seq2(s2)
        out = []
        for eng in (psa.ENGINE_SEQUENTIAL, psa.ENGINE_FILTER):
            c.set_engine(eng)
            ts = []
            for k in range(25):
                t0 = time.perf_counter(); r = c.search(); ts.append(time.perf_counter() - t0)
            out.append(sorted(ts[5:])[10] * 1e6)
        print(f"{n:8d} x {m:5d}  pairs {(n-m+1)*m:.2e}  seq {out[0]:7.1f} us  filter {out[1]:7.1f} us")
