import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "parallel-sequence-alignment_amd"))
import psa_amd as psa
n, m = 1_000_000, 10_000
t1 = torch.from_numpy(psa.generate_sequence_np(1002, n)).cuda(); t2 = torch.from_numpy(psa.generate_sequence_np(2002, m)).cuda()
ctx = psa.Context(0); ctx.set_weights([1.5,2.6,0.3,0.2], True)
ctx.set_sequences_device(t1.data_ptr(), n, t2.data_ptr(), m)
for k in range(12):
    torch.cuda.synchronize(); r = ctx.search()
