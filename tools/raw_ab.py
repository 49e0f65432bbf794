#!/usr/bin/env python3
"""One-shot search from raw ASCII (psa_search_device): blocking latency for the debug library's PSA_RAW_CONVERTERS
settings given on the command line (0 = every workgroup writes Seq2's class order itself; N = the first N do)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "parallel-sequence-alignment_amd"))
import psa_amd as psa  # noqa: E402

n, m = 1_000_000, 10_000
t1 = torch.from_numpy(psa.generate_sequence_np(1002, n)).cuda()
t2 = torch.from_numpy(psa.generate_sequence_np(2002, m)).cuda()
for rep in range(3):
    for conv in sys.argv[1:]:
        os.environ["PSA_RAW_CONVERTERS"] = conv
        ctx = psa.Context(0, torch.cuda.current_stream().cuda_stream)
        ctx.set_weights([1.5, 2.6, 0.3, 0.2], True)
        lat = []
        for k in range(45):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            r = ctx.search_device(t1.data_ptr(), n, t2.data_ptr(), m)
            lat.append(time.perf_counter() - t0)
        print(f"converters {conv:>4s}: blocking {1e6 * sorted(lat[5:])[20]:6.1f} us  ({r.offset}, retries {ctx.raw_retry_count()})", flush=True)
        ctx.close()
