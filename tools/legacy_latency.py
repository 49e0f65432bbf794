import os, sys, time, ctypes as C
import numpy as np, torch
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "parallel-sequence-alignment_amd"))
import psa_amd as psa
n, m = 1_000_000, 10_000
h1 = psa.generate_sequence_np(1002, n).tobytes(); h2 = psa.generate_sequence_np(2002, m).tobytes()
cap1, cap2 = n + 1, m + 1
blob = bytearray(40 + cap1 + cap2)
blob[0:4] = (1).to_bytes(4, "little")
blob[8:40] = np.asarray([1.5, 2.6, 0.3, 0.2], np.float64).tobytes()
blob[40:40 + n] = h1
blob[40 + cap1:40 + cap1 + m] = h2
data = (C.c_char * len(blob)).from_buffer(blob)
mu = psa.Mutant()
lat = []
for k in range(120):
    t0 = time.perf_counter()
    s = psa.lib.gpu_run_program_layout(C.addressof(data), cap1, cap2, C.byref(mu), 0, n - m + 1)
    lat.append(time.perf_counter() - t0)
lat = sorted(lat[20:])
print(f"gpu_run_program median {1e6*lat[len(lat)//2]:.1f} p10 {1e6*lat[len(lat)//10]:.1f} p90 {1e6*lat[9*len(lat)//10]:.1f} us; offset {mu.offset}; streamed {psa.lib.psa_streamed_count(None)} redone {psa.lib.psa_stream_retry_count(None)}")
with psa.Context(0) as c:
    c.set_weights([1.5, 2.6, 0.3, 0.2], True)
    lat = []
    for k in range(120):
        t0 = time.perf_counter()
        c.set_sequences(h1, h2); r = c.search()
        lat.append(time.perf_counter() - t0)
    lat = sorted(lat[20:])
    print(f"set_sequences + search median {1e6*lat[len(lat)//2]:.1f} p10 {1e6*lat[len(lat)//10]:.1f} us")
