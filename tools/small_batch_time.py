#!/usr/bin/env python3
"""Where a big batch of small searches spends its time: wall time of the C call (queries already joined) for several
batch sizes, one mode and MIN+MAX, and the per-query cost.   python tools/small_batch_time.py [n m]"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "parallel-sequence-alignment_amd"))
import torch  # noqa: F401,E402
import psa_amd as psa  # noqa: E402

n, m = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (100_000, 1000)
rng = np.random.default_rng(5)
s1 = rng.integers(65, 91, size=n, dtype=np.uint8).tobytes()
with psa.Context(0) as c:
    c.set_weights([1.5, 2.6, 0.3, 0.2], True)
    c.set_seq1(s1)
    for nq in (128, 512, 2048, 8192):
        buf = rng.integers(65, 91, size=nq * m, dtype=np.uint8).tobytes()
        hi = (psa.Result * nq)()
        lo = (psa.Result * nq)()
        total = n - m + 1
        d = torch.frombuffer(bytearray(buf), dtype=torch.uint8).cuda()
        torch.cuda.synchronize()
        for name, call in (("host, one mode", lambda: psa.lib.psa_search_batch(c._h, buf, nq, m, m, 0, total, hi)),
                           ("host, MIN+MAX", lambda: psa.lib.psa_search_batch_minmax(c._h, buf, nq, m, m, 0, total, hi, lo)),
                           ("device, one mode", lambda: psa.lib.psa_search_batch_device(c._h, d.data_ptr(), nq, m, m, 0, total, hi))):
            assert call() == 0
            t = []
            for _ in range(3):
                t0 = time.perf_counter()
                assert call() == 0
                t.append(time.perf_counter() - t0)
            dt = min(t)
            print(f"n={n} m={m} nq={nq:5d} {name:17s}: {1e3 * dt:8.2f} ms = {1e6 * dt / nq:6.2f} us per query "
                  f"({nq * total * m / dt:.3g} pairs/s)", flush=True)
