#!/usr/bin/env python3
"""Phase timestamps of psa_count_kernel (debug build with -DPSA_PHASE_TIMING, libpsa_phase.so).

  PSA_AMD_LIB=.../libpsa_phase.so python tools/phase_timing.py [n] [m]
prints, over all workgroups, the mean/min/max duration of each phase in microseconds."""
import ctypes as C
import os
import sys

import numpy as np
import torch  # noqa: F401  (its HIP runtime first, see tests/conftest.py)

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "parallel-sequence-alignment_amd"))
import psa_amd  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000
first = int(sys.argv[3]) if len(sys.argv) > 3 else 0
last = int(sys.argv[4]) if len(sys.argv) > 4 else n - m + 1
raw = os.environ.get("PHASE_RAW") == "1"          # the one-shot search from raw ASCII (psa_count_kernel<16, raw>)
# PHASE_ALPHABET=ACGT (or any letters; a trailing :w1,w2,.. gives relative frequencies): a low-entropy input instead of
# uniform A..Z;  PHASE_MIN=1: a minimum search
alphabet = os.environ.get("PHASE_ALPHABET", "")
is_max = os.environ.get("PHASE_MIN") != "1"


def make(seed, length):
    if not alphabet:
        return psa_amd.generate_sequence_np(seed, length)
    letters, _, freq = alphabet.partition(":")
    p = None
    if freq:
        p = np.array([float(x) for x in freq.split(",")])
        p /= p.sum()
    return np.random.default_rng(seed).choice(np.frombuffer(letters.encode(), np.uint8), length, p=p)


seq1_np, seq2_np = make(1002, n), make(2002, m)
ctx = psa_amd.Context()
ctx.set_weights([1.5, 2.6, 0.3, 0.2], is_max)
ctx.set_seq1(seq1_np.tobytes())
ctx.set_seq2(seq2_np.tobytes())
if raw:
    d1 = torch.from_numpy(seq1_np).cuda()
    d2 = torch.from_numpy(seq2_np).cuda()
    torch.cuda.synchronize()
for _ in range(3):
    r = ctx.search_device(d1.data_ptr(), n, d2.data_ptr(), m, first, last) if raw else ctx.search(first, last)
if raw:
    ctx.set_seq1(seq1_np.tobytes())
    ctx.set_seq2(seq2_np.tobytes())
lib = psa_amd._lib if hasattr(psa_amd, "_lib") else psa_amd.lib
buf = (C.c_ulonglong * (1024 * 8))()
lib.psa_debug_phase_ts.restype = C.c_int
assert lib.psa_debug_phase_ts(buf) == 0
ts = np.frombuffer(buf, dtype=np.uint64).reshape(1024, 8).astype(np.int64)
nt = 300           # rows 300.. hold other stamps (raw converters, batch kernel, atomics at 700.., tail and re-score at 1000..)
ts = ts[:nt]
ts = ts[ts[:, 0] > 0]
t0 = ts[:, 0].min()
names = ["build planes", "main loop (to the LAST wave's end)", "barrier, waves 8-15 store planes, barrier", "waves 0-7 add + store, barrier",
         "column sums, conversion, counts read", "bounds+select"]
print(f"{len(ts)} workgroups; start spread {(ts[:,0].max()-t0)/100:.2f} us; last end {(ts[:,6].max()-t0)/100:.2f} us")
for k, name in enumerate(names):
    d = (ts[:, k + 1] - ts[:, k]) / 100.0
    print(f"{name:44s} mean {d.mean():7.2f}  min {d.min():7.2f}  max {d.max():7.2f} us")
print("result", r)
if raw:
    full0 = np.frombuffer(buf, dtype=np.uint64).reshape(1024, 8).astype(np.int64)
    ex = full0[300:300 + len(ts)]
    for k, name in enumerate(["converter wave: symbol stores issued", "last wave: plane work done, stores drained (arrival next)",
                              "plane build's barrier passed (wait for the arrivals next)"]):
        d = (ex[:, k] - ts[:, 0]) / 100.0
        d = d[ex[:, k] > 0]
        if len(d):
            print(f"raw  {name:62s} mean {d.mean():6.2f}  min {d.min():6.2f}  max {d.max():6.2f} us after the workgroup's start")
tail = ts[ts[:, 7] > 0]
for row in tail:     # the workgroup that finished the search (finish_search)
    print(f"fused tail (last tile workgroup): select end -> tail written {(row[7] - row[6]) / 100.0:.2f} us; "
          f"kernel start -> end {(row[7] - t0) / 100.0:.2f} us")

full = np.frombuffer(buf, dtype=np.uint64).reshape(1024, 8).astype(np.int64)
ft, fc = full[1002], full[1003]
names = ["arrival -> lists filtered, table", "pass 1: classify + counts (all waves)", "pass 2: block prefix + level 1",
         "pass 3: level 2 | head chain", "pass 4: sum (one wave per survivor)", "fold + tail"]
print(f"fused tail of the last search: {fc[0]} candidates listed, {fc[1]} survive the final bound, {fc[2]} segments of "
      f"survivor 0 went to level 2")
for k, name in enumerate(names):
    print(f"tail {name:36s} {(ft[k + 1] - ft[k]) / 100.0:7.2f} us")

at = full[700:700 + min(len(ts), 244)]
if at[:, 0].max() > 0:
    d_max = (at[:, 1] - at[:, 0]) / 100.0
    d_arr = (at[:, 3] - at[:, 2]) / 100.0
    order = np.argsort(at[:, 4])
    print(f"atomicMax on the running bound, per workgroup: mean {d_max.mean():.2f} min {d_max.min():.2f} max {d_max.max():.2f} us; "
          f"arrival atomic: mean {d_arr.mean():.2f} min {d_arr.min():.2f} max {d_arr.max():.2f} us; the last 8 arrivers' "
          f"atomicMax {np.round(d_max[order[-8:]], 2).tolist()} arrival {np.round(d_arr[order[-8:]], 2).tolist()}; "
          f"arrival times of the last 8 relative to the last: {np.round((at[order[-8:], 2] - at[order[-1], 2]) / 100.0, 2).tolist()}")
p1 = full[1005]
if p1[0] > 0:
    print(f"pass 1 (wave 0): tail phase-1 stamp -> loads issued {(p1[0] - ft[1]) / 100.0:.2f} us; loads returned +{(p1[1] - p1[0]) / 100.0:.2f}; "
          f"lookups + codes stored +{(p1[2] - p1[1]) / 100.0:.2f}; segment done, padding +{(p1[3] - p1[2]) / 100.0:.2f}; wave max + scan "
          f"+{(p1[4] - p1[3]) / 100.0:.2f}; last wave there {(p1[5] - ft[1]) / 100.0:.2f} us after the phase-1 stamp")
st = full[1004]
for k, name in enumerate(["true additions compacted from the bitmap", "counts at the head", "events prepared (first 64)", "events applied (to the end)"]):
    print(f"sum  {name:36s} {(st[k + 1] - st[k]) / 100.0:7.2f} us")

# re-score kernel (workgroup 0, its first candidate)
ctx.debug_rescore(np.array([r.offset, 5, 77777], dtype=np.int32))
assert lib.psa_debug_phase_ts(buf) == 0
ts = np.frombuffer(buf, dtype=np.uint64).reshape(1024, 8).astype(np.int64)[1000]
names = ["start -> tables + scalars", "classify + counts (all waves)", "certificate + sum (wave 0)", "key reduce + record", "arrive + fold + tail"]
for k, name in enumerate(names):
    print(f"rescore {name:28s} {(ts[k + 1] - ts[k]) / 100.0:7.2f} us")
walk = np.frombuffer(buf, dtype=np.uint64).reshape(1024, 8)[1001]
print(f"rescore walk of the first candidate: {walk[0]} chain steps (<= 64 chars each), {walk[1]} scan steps (<= 512 chars), "
      f"{walk[2]} binade changes")
