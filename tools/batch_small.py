#!/usr/bin/env python3
"""Batch throughput for reference-sized inputs (few offset tiles): python tools/batch_small.py [n m nq]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "parallel-sequence-alignment_amd"))
import torch  # noqa
import psa_amd as psa
shapes = [(9711, 2131, 256), (10000, 5000, 256), (50000, 1000, 256), (200000, 2000, 256), (1000000, 10000, 256)]
if len(sys.argv) > 3: shapes = [tuple(map(int, sys.argv[1:4]))]
with psa.Context(0) as c:
    c.set_weights([1.5, 2.6, 0.3, 0.2], True)
    for n, m, nq in shapes:
        s1 = psa.generate_sequence(5, n); qs = [psa.generate_sequence(100 + q, m) for q in range(nq)]
        c.set_seq1(s1)
        c.search_batch(qs)
        ts = []
        for k in range(5):
            t0 = time.perf_counter(); r = c.search_batch(qs); ts.append(time.perf_counter() - t0)
        t = sorted(ts)[2]
        c.set_seq2(qs[0]); t0 = time.perf_counter(); c.search(); t1 = time.perf_counter() - t0
        c.search(); t0 = time.perf_counter(); one = c.search(); t1 = time.perf_counter() - t0
        assert (one.offset, one.score) == (r[0].offset, r[0].score)
        print(f"{n:8d} x {m:5d} x {nq} queries: batch {t*1e3:8.3f} ms = {t/nq*1e6:7.1f} us/query ({(n-m+1)*m*nq/t:.3g} pairs/s); one blocking search {t1*1e6:6.1f} us")
