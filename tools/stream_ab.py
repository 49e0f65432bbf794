#!/usr/bin/env python3
"""Host sequences -> device, streamed against plain (debug library knobs PSA_STREAM_MIN / PSA_STREAM_CHUNK):
median wall time of psa_set_sequences + psa_search and of gpu_run_program, 1M x 10k, one GPU."""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "parallel-sequence-alignment_amd"))
import psa_amd as psa  # noqa: E402

n, m = 1_000_000, 10_000
W = [1.5, 2.6, 0.3, 0.2]
s1, s2 = psa.generate_sequence_np(1002, n), psa.generate_sequence_np(2002, m)
h1, h2 = s1.tobytes(), s2.tobytes()
ctx = psa.Context(0, torch.cuda.current_stream().cuda_stream)
ctx.set_weights(W, True)


def med(fn, reps=40, warm=5):
    t = []
    for k in range(reps + warm):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        if k >= warm:
            t.append(time.perf_counter() - t0)
    return 1e6 * sorted(t)[len(t) // 2]


cap1, cap2 = n + 1, m + 1
blob = bytearray(40 + cap1 + cap2)
blob[0:4] = (1).to_bytes(4, "little")
blob[8:40] = np.asarray(W, np.float64).tobytes()
blob[40:40 + n] = h1
blob[40 + cap1:40 + cap1 + m] = h2
data = (C.c_char * len(blob)).from_buffer(blob)
mutant = psa.Mutant()


def host_pair():
    ctx.set_sequences(h1, h2)
    return ctx.search()


def only_set():
    ctx.set_sequences(h1, h2)


def legacy():
    return psa.lib.gpu_run_program_layout(C.addressof(data), cap1, cap2, C.byref(mutant), 0, n - m + 1)


knobs = {k: os.environ.get(k) for k in ("PSA_STREAM_MIN", "PSA_STREAM_CHUNK")}
print(f"{knobs}: set_sequences {med(only_set):6.1f} us, + search {med(host_pair):6.1f} us, gpu_run_program {med(legacy):6.1f} us "
      f"(streamed {ctx.streamed_count()} / {psa.lib.psa_streamed_count(None)})", flush=True)
ctx.close()
