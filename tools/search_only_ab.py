#!/usr/bin/env python3
"""Same-GPU A/B of library variants on the blocking 1M x 10k search (sequences already set): alternating rounds,
median of 200 searches each.   python tools/search_only_ab.py amd nosplit amd:HIP_FORCE_DEV_KERNARG=1,X=Y ...
(psa_amd/libpsa_<name>.so, optionally with environment settings for that child).
Each variant runs in a child process (one library per process)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CHILD = r'''
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.environ["PSA_ROOT"], "parallel-sequence-alignment_amd"))
import psa_amd as psa
n, m = 1_000_000, 10_000
alphabet = os.environ.get("AB_ALPHABET", "")
if alphabet:
    rng = np.random.default_rng(4); ab = np.frombuffer(alphabet.encode(), np.uint8)
    s1, s2 = rng.choice(ab, n), rng.choice(ab, m)
else:
    s1, s2 = psa.generate_sequence_np(1002, n), psa.generate_sequence_np(2002, m)
t1, t2 = torch.from_numpy(s1).cuda(), torch.from_numpy(s2).cuda()
ctx = psa.Context(0, torch.cuda.current_stream().cuda_stream)
ctx.set_weights([1.5, 2.6, 0.3, 0.2], True)
ctx.set_sequences_device(t1.data_ptr(), n, t2.data_ptr(), m)
lat = []
for k in range(220):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = ctx.search(); lat.append(time.perf_counter() - t0)
lat = sorted(lat[20:])
print(f"{1e6 * lat[len(lat) // 2]:.2f} {1e6 * lat[len(lat) // 10]:.2f} {r.offset}")
'''
root = os.path.dirname(HERE)
for rep in range(3):
    for spec in sys.argv[1:]:
        name, _, settings = spec.partition(":")
        env = dict(os.environ, PSA_ROOT=root, PSA_AMD_LIB=os.path.join(root, "parallel-sequence-alignment_amd", "psa_amd", f"libpsa_{name}.so"))
        env.update(kv.split("=", 1) for kv in settings.split(",") if kv)
        out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
        print(f"{spec:36s} search-only median / p10 us, offset: {out.stdout.strip() or out.stderr[-300:]}", flush=True)
