#!/usr/bin/env python3
"""Where the end-to-end latency of ONE blocking 1M x 10k search goes (one GPU, median of 30):
inputs in HBM (two encode launches / one / none) and inputs in host memory (psa_set_seq1 + psa_set_seq2,
psa_set_sequences, gpu_run_program).  PSA_AMD_LIB selects a library variant."""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "parallel-sequence-alignment_amd"))
import psa_amd as psa  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000
W = [1.5, 2.6, 0.3, 0.2]
s1, s2 = psa.generate_sequence_np(1002, n), psa.generate_sequence_np(2002, m)
t1, t2 = torch.from_numpy(s1).cuda(), torch.from_numpy(s2).cuda()
h1, h2 = s1.tobytes(), s2.tobytes()
ctx = psa.Context(0, torch.cuda.current_stream().cuda_stream)
ctx.set_weights(W, True)


def med(fn, reps=30, warm=5):
    t = []
    for k in range(reps + warm):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        if k >= warm:
            t.append(time.perf_counter() - t0)
    return 1e6 * sorted(t)[len(t) // 2]


def two_encodes():
    ctx.set_seq1_device(t1.data_ptr(), n)
    ctx.set_seq2_device(t2.data_ptr(), m)
    return ctx.search()


def one_encode():
    ctx.set_sequences_device(t1.data_ptr(), n, t2.data_ptr(), m)
    return ctx.search()


def host_two():
    ctx.set_seq1(h1)
    ctx.set_seq2(h2)
    return ctx.search()


def host_pair():
    ctx.set_sequences(h1, h2)
    return ctx.search()


cap1, cap2 = n + 1, m + 1
blob = bytearray(40 + cap1 + cap2)
blob[0:4] = (1).to_bytes(4, "little")
blob[8:40] = np.asarray(W, np.float64).tobytes()
blob[40:40 + n] = h1
blob[40 + cap1:40 + cap1 + m] = h2
data = (C.c_char * len(blob)).from_buffer(blob)
mutant = psa.Mutant()


def legacy():
    return psa.lib.gpu_run_program_layout(C.addressof(data), cap1, cap2, C.byref(mutant), 0, n - m + 1)


def one_shot():
    return ctx.search_device(t1.data_ptr(), n, t2.data_ptr(), m)


r = two_encodes()
print(f"result {r.offset} {r.score!r}  unfused {ctx.unfused_count()}")
print(f"HBM inputs, set_seq1_device + set_seq2_device + search : {med(two_encodes):7.1f} us")
print(f"HBM inputs, set_sequences_device + search              : {med(one_encode):7.1f} us")
print(f"HBM inputs, psa_search_device (ONE launch)             : {med(one_shot):7.1f} us   (retries {ctx.raw_retry_count()})")
ctx.set_sequences_device(t1.data_ptr(), n, t2.data_ptr(), m)
print(f"sequences already set, search only                     : {med(ctx.search):7.1f} us")
print(f"host inputs, set_seq1 + set_seq2 + search              : {med(host_two):7.1f} us")
print(f"host inputs, set_sequences + search                    : {med(host_pair):7.1f} us")
print(f"gpu_run_program (ProgramData in host memory)           : {med(legacy):7.1f} us")
print(f"host transfers streamed (encode launch follows the copy): context {ctx.streamed_count()} (redone {ctx.stream_retry_count()}), "
      f"gpu_run_program {psa.lib.psa_streamed_count(None)} (redone {psa.lib.psa_stream_retry_count(None)})")
ctx.set_profiling(True)
for _ in range(10):
    ctx.search()
tot, kern, k = ctx.profile_totals()
print(f"HIP events: whole search {1e3*tot/k:.1f} us, kernel {1e3*kern/k:.1f} us ({ctx.last_timing(1)[1]})")
ctx.close()
