#!/usr/bin/env python3
"""Phase timestamps of a CHUNKED single search (a 1/N offset shard of configs[2]; PSA_FILTER_CHUNKS forces the chunk
count): debug build with -DPSA_EXPERIMENT -DPSA_PHASE_TIMING (make variant V=phasedbg EXTRA="..."),
  PSA_AMD_LIB=.../libpsa_phasedbg.so [PSA_FILTER_CHUNKS=5] python tools/phase_chunks.py [shards=8]
Prints the workgroups' phases (us) and, for the tiles' LAST ARRIVERS (the workgroups that add up a tile's chunk counts and
select), the sub-phases of the hand-off: what the planner's chunk term (psa_device.hip, plan_filter) is fitted on."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "parallel-sequence-alignment_amd"))
import psa_amd
shards = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n, m = 1_000_000, 10_000
first, last = psa_amd.shard_range(n - m + 1, shards, 0)
ctx = psa_amd.Context(); ctx.set_engine(2); ctx.set_weights([1.5, 2.6, 0.3, 0.2], True)
ctx.set_seq1(psa_amd.generate_sequence(1002, n)); ctx.set_seq2(psa_amd.generate_sequence(2002, m))
for _ in range(4): r = ctx.search(first, last)
buf = (C.c_ulonglong * (1024 * 8))()
cbuf = (C.c_ulonglong * (1024 * 16))()
psa_amd.lib.psa_debug_phase_ts.restype = C.c_int
psa_amd.lib.psa_debug_chunk_ts.restype = C.c_int
assert psa_amd.lib.psa_debug_phase_ts(buf) == 0 and psa_amd.lib.psa_debug_chunk_ts(cbuf) == 0
ts = np.frombuffer(buf, dtype=np.uint64).reshape(1024, 8).astype(np.int64)
cs = np.frombuffer(cbuf, dtype=np.uint64).reshape(1024, 16).astype(np.int64)
tail = ts[1002]                       # finish_search's stamps (the search's last workgroup)
used = ts[:1000, 0] > 0
ts, cs = ts[:1000][used], cs[:1000][used]
fresh = ts[:, 0] >= ts[:, 0].max() - 10_000          # this launch's workgroups (stale rows of larger earlier grids are older)
ts, cs = ts[fresh], cs[fresh]
t0 = ts[:, 0].min()
print(f"1/{shards} shard, offsets [{first},{last}): {len(ts)} workgroups; start spread {(ts[:,0].max()-t0)/100:.2f} us")
for k, nm in enumerate(["build planes", "main loop (wave 0)", "barrier + dump", "take + dump"]):
    d = (ts[:, k + 1] - ts[:, k]) / 100.0
    print(f"  {nm:24s} mean {d.mean():6.2f}  min {d.min():6.2f}  max {d.max():6.2f}")
la = ts[:, 6] > ts[:, 4]              # the last arrivers ran the selection in this launch
print(f"  sum_columns + read_cells  mean {((cs[:,0]-ts[:,4])/100.0).mean():6.2f}")
names = ["publish stores drained (vmcnt 0)", "barrier", "arrival atomic (round trip)", "barrier"]
for k, nm in enumerate(names):
    d = (cs[:, k + 1] - cs[:, k]) / 100.0
    print(f"  {nm:34s} mean {d.mean():6.2f}  min {d.min():6.2f}  max {d.max():6.2f}")
print(f"last arrivers: {la.sum()} (one per tile)")
lcs, lts = cs[la], ts[la]
names = ["combine: others' counts loaded + added", "bounds + wave max", "barrier", "atomicMax on the bound (round trip)", "barrier"]
for k, nm in enumerate(names):
    d = (lcs[:, k + 5] - lcs[:, k + 4]) / 100.0
    print(f"  {nm:38s} mean {d.mean():6.2f}  min {d.min():6.2f}  max {d.max():6.2f}")
d = (lts[:, 6] - lcs[:, 9]) / 100.0
print(f"  {'candidate selection':38s} mean {d.mean():6.2f}  min {d.min():6.2f}  max {d.max():6.2f}")
print(f"timeline (us from the first workgroup's start): counts final {((lts[:,5]-t0)/100.0).min():.1f}..{((lts[:,5]-t0)/100.0).max():.1f}, "
      f"selection done {((lts[:,6]-t0)/100.0).min():.1f}..{((lts[:,6]-t0)/100.0).max():.1f}")
if tail[0] > 0:
    print("finish_search stamps (us from the first workgroup's start):", [round((int(x) - t0) / 100.0, 1) for x in tail[:7] if x > 0])
