#!/usr/bin/env python3
"""Blocking-search latency of the filter engine for forced workgroup shapes / chunk counts (debug library:
PSA_FILTER_CHUNKS), to fit plan_filter's cost model.
  PSA_AMD_LIB=.../libpsa_debug.so python tools/plan_sweep.py n m first last"""
import os, sys, time, subprocess
if len(sys.argv) > 5:            # child: one forced plan
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "parallel-sequence-alignment_amd"))
    import psa_amd as psa
    n, m, first, last = map(int, sys.argv[1:5])
    with psa.Context(0) as c:
        c.set_engine(psa.ENGINE_FILTER); c.set_weights([1.5, 2.6, 0.3, 0.2], True)
        c.set_seq1(psa.generate_sequence(1002, n)); c.set_seq2(psa.generate_sequence(2002, m))
        try:
            ts = []
            for k in range(40):
                t0 = time.perf_counter(); r = c.search(first, last); ts.append(time.perf_counter() - t0)
            c.set_profiling(True); c.search(first, last); km, name = c.last_timing(1)
            print(f"{sorted(ts[8:])[16]*1e6:8.1f} us/search  kernel {km*1e3:7.1f} us {name}  offset {r.offset}")
        except psa.PsaError as e:
            print("n/a", str(e)[:60])
    sys.exit(0)
n, m, first, last = map(int, sys.argv[1:5])
for chunks in (0, 1, 2, 3, 4, 5, 6, 8, 10, 16):
    env = dict(os.environ)
    if chunks: env["PSA_FILTER_CHUNKS"] = str(chunks)
    out = subprocess.run([sys.executable, __file__, str(n), str(m), str(first), str(last), "x"], env=env, capture_output=True, text=True).stdout.strip()
    print(f"chunks {chunks if chunks else 'auto':>4}: {out}", flush=True)
