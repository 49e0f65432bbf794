import os, sys, time
ROOT = "/root/repo"
sys.path.insert(0, os.path.join(ROOT, "parallel-sequence-alignment_amd"))
import psa_amd as psa
shapes = [(1000, 300), (3000, 1000), (9711, 2131), (20000, 500), (50000, 200), (100000, 1000), (300000, 100), (1000000, 30)]
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
