#!/usr/bin/env python3
"""Latency of the smaller BASELINE.json configs (host buffers -> result on host, context reused):
configs[0] bundled input.txt record 0 (9711 x 2131, MIN) and configs[1] synthetic 100k x 1k MAX."""
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "parallel-sequence-alignment_amd"))
import psa_amd as psa  # noqa: E402

lines = open(os.path.join(ROOT, "tests", "golden", "reference_input.txt")).read().split()
w0 = [float(x) for x in lines[:4]]
cases = [("configs[0] input.txt record 0", w0, lines[4].encode(), lines[5].encode(), lines[6] == "maximum"),
         ("configs[1] 100k x 1k MAX", [1.5, 2.6, 0.3, 0.2], psa.generate_sequence(1001, 100_000), psa.generate_sequence(2001, 1_000), True)]
with psa.Context(0) as c:
    for name, w, s1, s2, is_max in cases:
        c.set_weights(w, is_max)
        ts = []
        for k in range(30):
            t0 = time.perf_counter()
            c.set_seq1(s1)
            c.set_seq2(s2)
            r = c.search()
            ts.append(time.perf_counter() - t0)
        ts = sorted(ts[5:])
        pairs = (len(s1) - len(s2) + 1) * len(s2)
        print(f"{name}: {ts[len(ts)//2]*1e6:.1f} us per search from host buffers ({pairs/ts[len(ts)//2]:.3g} pairs/s), "
              f"offset {r.offset} score {r.score!r}")
