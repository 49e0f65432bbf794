#!/usr/bin/env python3
"""Phase timestamps of a CHUNKED single search (a 1/8 offset shard of configs[2], PSA_FILTER_CHUNKS to force the chunk
count): debug build with -DPSA_PHASE_TIMING -DPSA_EXPERIMENT (make variant V=phasedbg EXTRA=\"...\"),
  PSA_AMD_LIB=.../libpsa_phasedbg.so PSA_FILTER_CHUNKS=5 python tools/phase_chunks.py"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "parallel-sequence-alignment_amd"))
import psa_amd
n, m, first, last = 1_000_000, 10_000, 0, 123750
ctx = psa_amd.Context(); ctx.set_engine(2); ctx.set_weights([1.5,2.6,0.3,0.2], True)
ctx.set_seq1(psa_amd.generate_sequence(1002, n)); ctx.set_seq2(psa_amd.generate_sequence(2002, m))
for _ in range(3): r = ctx.search(first, last)
buf = (C.c_ulonglong * (1024 * 8))()
psa_amd.lib.psa_debug_phase_ts.restype = C.c_int
assert psa_amd.lib.psa_debug_phase_ts(buf) == 0
ts = np.frombuffer(buf, dtype=np.uint64).reshape(1024, 8).astype(np.int64)[:155]
t0 = ts[:, 0].min()
names = ["build", "main(w0)", "barrier", "adds", "cells->counts+publish/combine", "bounds+select"]
last_wg = ts[:, 6] > ts[:, 5]   # those that ran phase 6 in this launch (the last arrivers) — others keep stale values
print("start spread", (ts[:,0].max()-t0)/100, "end", (ts[:,5].max()-t0)/100)
for k, nm in enumerate(names[:5]):
    d = (ts[:, k+1]-ts[:, k])/100.0
    print(f"{nm:32s} mean {d.mean():7.2f} min {d.min():7.2f} max {d.max():7.2f}")
d = (ts[:,6]-ts[:,5])/100.0
print("bounds (valid for last arrivers)", np.sort(d)[-35:].round(1))
print("phase5 end times (us from start) per WG:", np.sort((ts[:,5]-t0)/100.0).round(1)[::8])
