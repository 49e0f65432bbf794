#!/usr/bin/env python3
"""Where the FIRST gpu_run_program of a fresh process spends its time (the drop-in program's timed region,
cpu_funcs.c:57-62, is that call): stage stamps of the debug library (PSA_TRACE_INIT=1, libpsa_debug.so), then the same
call twice more (warm).  No torch in this process: the HIP runtime is initialised by the library's first call, as in
psa_main.

  PSA_TRACE_INIT=1 PSA_AMD_LIB=.../libpsa_debug.so python tools/cold_start.py [n] [m]
"""
import ctypes as C
import os
import sys
import time

t_start = time.perf_counter()
import numpy as np  # noqa: E402

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "parallel-sequence-alignment_amd"))
t0 = time.perf_counter()
import psa_amd as psa  # noqa: E402
t_loaded = time.perf_counter()

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000
W = [1.5, 2.6, 0.3, 0.2]
h1, h2 = psa.generate_sequence(1002, n), psa.generate_sequence(2002, m)
cap1, cap2 = n + 1, m + 1
blob = bytearray(40 + cap1 + cap2)
blob[0:4] = (1).to_bytes(4, "little")
blob[8:40] = np.asarray(W, np.float64).tobytes()
blob[40:40 + n] = h1
blob[40 + cap1:40 + cap1 + m] = h2
data = (C.c_char * len(blob)).from_buffer(blob)
mutant = psa.Mutant()
print(f"dlopen of the library (+ its HIP runtime dependency): {1e3 * (t_loaded - t0):.1f} ms", file=sys.stderr)
for k in range(3):
    t0 = time.perf_counter()
    score = psa.lib.gpu_run_program_layout(C.addressof(data), cap1, cap2, C.byref(mutant), 0, n - m + 1)
    dt = time.perf_counter() - t0
    print(f"gpu_run_program call {k}: {1e3 * dt:9.3f} ms   -> offset {mutant.offset} score {score!r}", file=sys.stderr)
