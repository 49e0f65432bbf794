import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "parallel-sequence-alignment_amd"))
import psa_amd as psa
n, m = 1_000_000, 10_000
t1 = torch.from_numpy(psa.generate_sequence_np(1002, n)).cuda(); tq = torch.from_numpy(psa.generate_sequence_np(2002, m)).cuda()
ctx = psa.Context(0, torch.cuda.current_stream().cuda_stream); ctx.set_weights([1.5,2.6,0.3,0.2], True)
lat=[]
for k in range(40):
    torch.cuda.synchronize(); t0=time.perf_counter()
    ctx.set_seq1_device(t1.data_ptr(), n); ctx.set_seq2_device(tq.data_ptr(), m); r=ctx.search()
    lat.append(time.perf_counter()-t0)
print(os.path.basename(os.environ.get("PSA_AMD_LIB","amd")), "single search latency us", 1e6*sorted(lat[5:])[17], r.offset, r.score)
ctx.close()
