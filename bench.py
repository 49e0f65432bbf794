#!/usr/bin/env python3
"""bench.py — headline benchmark of the offset x mutation search on MI355X.

Metric (BASELINE.json): char-pair scores/sec, whole job, on synthetic
|Seq1|=1M, |Seq2|=10k MAX searches (configs[2] sizes; uniform A..Z from the
library's seeded generator, weights 1.5 2.6 0.3 0.2).

A step is one pass of the hot path over one batch: Q independent Seq2 queries
(default 256 as in BASELINE.json configs[4], seeds 2002+q) against the same Seq1
(seed 1002) — every offset of every query scored, every best mutant known on the
host.  `value` starts from the raw ASCII sequences resident in HBM (symbol
encoding is inside the timed region), as the bench contract asks.  The SAME step
handed over as pageable HOST buffers — the reference's own timed region,
cpu_funcs.c:57-62: inputs in host memory -> best mutants on the host, PCIe
included — is reported next to it as `timed_region_8d`.
Query 0 is BASELINE.json configs[2]; the run checks its MAX and MIN answers
against the committed oracle goldens and exits non-zero on any mismatch, on a
retried batch or on a sequential-engine fallback (`verified`).

  python bench.py [--gpus N] [--steps K] [--warmup W]

`--gpus N` with N > 1 launches N ranks itself (one per GPU, torch.distributed.run
on 127.0.0.1) BEFORE anything touches the GPU, unless it already runs under a
launcher (WORLD_SIZE set): `python -m torch.distributed.run --nproc-per-node N
... bench.py --gpus N` works the same.  It exits non-zero if fewer than N devices
are visible or the world size differs from N.

N > 1, default `--scaling weak`: every rank searches its own Q queries (seeds
continue per rank; per-GPU work fixed), no data-path collective; ONE RCCL
all-reduce(sum) per step of a one-hot [N*Q,4] int64 buffer leaves all 32-byte
records on every rank (the argmax exchange of north_star).  `--scaling strong`
deals the Q queries of configs[4] to the ranks instead.  `--mode shard` runs
configs[3]: the offsets split like the reference's MPI ranks (cpu_funcs.c:128-133),
every rank searches its shard — one query, or `--queries Q` queries in flight per
rank (throughput mode) — one all-reduce of the per-rank records, fold with the
is_swapable order (cpu_funcs.c:71-94 replaced by one collective).
"""
import argparse
import glob
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "parallel-sequence-alignment_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

WEIGHTS = [1.5, 2.6, 0.3, 0.2]          # README.md:57
HBM_PEAK_GBS = 8000.0                   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
VALU_PEAK = 1024 * 2.4e9 / 2            # wave64 instructions/s: 1024 SIMDs, one full-rate VALU op per 2 cycles at 2.4 GHz
BYTES_PER_PAIR = 2                      # 1 B of Seq1 + 1 B of Seq2 per char pair
SEED_SEQ1, SEED_QUERY0 = 1002, 2002
PROFILE_LAUNCH_QUERIES = 128            # queries per launch of the batch count kernel (the library's default), and what the
                                        # committed PMC passes ran with (tools/profile_round.sh: --queries 128)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--n", type=int, default=1_000_000)
    ap.add_argument("--m", type=int, default=10_000)
    ap.add_argument("--queries", type=int, default=None,
                    help="Seq2 queries per step: 256 in batch mode; in shard mode queries in flight per rank (default 1)")
    ap.add_argument("--mode", choices=("batch", "shard"), default="batch")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="N > 1, batch mode: weak = Q queries per rank, strong = Q queries dealt to the ranks")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="only the timed region, the kernel probe and the self-check (profiling passes)")
    ap.add_argument("--engine", type=int, default=0)
    ap.add_argument("--batch-streams", type=int, default=1, choices=(1, 2),
                    help="2: a batch's count launches alternate between two streams (psa_set_batch_streams)")
    ap.add_argument("--emulate-world", type=int, default=0,
                    help="analysis only: time rank 0's share of a W-rank strong-scaling run on one GPU (no collective)")
    ap.add_argument("--backend", choices=("nccl", "gloo"), default="nccl",
                    help="collective backend of the record exchange (gloo: rehearsal, records travel through host memory)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="rehearsal on a one-GPU box: every rank uses device 0 (needs --backend gloo; not a scaling measurement)")
    ap.add_argument("--stub-search", action="store_true",
                    help="launcher / exchange rehearsal without a GPU: gloo, records made up from the query index")
    ap.add_argument("--group-leg", action="store_true",
                    help="internal: ONE process drives --gpus N devices through the library's own multi-GPU entry points "
                         "(psa_search_sharded / psa_search_batch_sharded: launcher threads, RCCL all-gather) and prints the "
                         "`c_abi_group` object; bench.py runs this as a child process and merges it into its line")
    ap.add_argument("--no-group-leg", action="store_true", help="skip the c_abi_group child process")
    args = ap.parse_args(argv)
    if args.queries is None:
        args.queries = 256 if args.mode == "batch" else 1
    return args


# --------------------------------------------------------------------------------------
# launcher: --gpus N without a launcher around us
# --------------------------------------------------------------------------------------
def self_launch(args):
    """Start N rank processes (torch.distributed.run, one node, 127.0.0.1) and return their exit code.
    Nothing in this process has touched the GPU: torch.cuda.device_count() only counts."""
    if not args.stub_search:
        import torch
        have = torch.cuda.device_count()
        if have < (1 if args.share_gpu else args.gpus):
            print(f"bench.py: --gpus {args.gpus} but only {have} HIP device(s) visible", file=sys.stderr)
            return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC only on this pool (RCCL)
    return subprocess.run(cmd, env=env).returncode


# --------------------------------------------------------------------------------------
# record exchange (the only collective): one all-reduce(sum) of a one-hot buffer
# --------------------------------------------------------------------------------------
def gather_records(psa, dist, torch, mine, first_slot, total_slots, device):
    """Every rank writes its 32-byte records into its own rows of a zeroed [total_slots,4] int64 buffer; the
    all-reduce(sum) is the gather of all records."""
    buf = torch.zeros((total_slots, 4), dtype=torch.int64, device=device)
    if mine:
        rows = torch.frombuffer(bytearray(b"".join(bytes(r) for r in mine)), dtype=torch.int64).view(len(mine), 4)
        buf[first_slot:first_slot + len(mine)].copy_(rows, non_blocking=True)
    dist.all_reduce(buf, op=dist.ReduceOp.SUM)
    host = buf.cpu().numpy().tobytes()
    return [psa.Result.from_buffer_copy(host[32 * r: 32 * (r + 1)]) for r in range(total_slots)]


def exchange_and_fold(psa, dist, torch, rec, rank, world, is_max, device):
    """configs[3]: per-rank shard records -> one all-reduce -> is_swapable fold on every rank."""
    recs = gather_records(psa, dist, torch, [rec], rank, world, device)
    for r, x in enumerate(recs):
        x.shard = r
    return psa.fold(recs, is_max)


def exchange_and_fold_many(psa, dist, torch, recs, rank, world, is_max, device):
    """Throughput-mode shard: every rank holds one record per query; one all-reduce of [world * Q] records, then the
    is_swapable fold per query."""
    nq = len(recs)
    allr = gather_records(psa, dist, torch, recs, rank * nq, world * nq, device)
    out = []
    for q in range(nq):
        per_rank = [allr[r * nq + q] for r in range(world)]
        for r, x in enumerate(per_rank):
            x.shard = r
        out.append(psa.fold(per_rank, is_max))
    return out


def query_block(nq, world, rank):
    """Contiguous block of queries of `rank`: as even as possible (psa_query_block, the library's own deal)."""
    import psa_amd as psa
    return psa.query_block(nq, world, rank)


def host_cores():
    """CPUs this process may really use: the affinity mask, capped by the cgroup CPU quota (the GPU boxes expose
    all 256 hardware threads in the mask but grant a share of them)."""
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            cores = min(cores, max(1, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        try:
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0:
                cores = min(cores, max(1, -(-quota // period)))
        except (OSError, ValueError):
            pass
    return cores


def cpu_baseline(s1, s2, total, m, budget_s=9.0, threads=None):
    """The oracle (reference-faithful per-pair work, OpenMP static split like
    cpu_funcs.c:174-199) timed on this host's cores over a contiguous slice of
    query 0's offsets sized for ~budget_s of CPU time."""
    from oracle import oracle
    cores = host_cores()
    if threads:
        cores = min(cores, threads)
    probe = min(total, 64 * cores)
    t0 = time.perf_counter()
    oracle.search(s1, s2, WEIGHTS, True, 0, probe, threads=cores)
    dt = max(time.perf_counter() - t0, 1e-6)
    sample = int(min(total, max(probe, probe * budget_s / dt)))
    t0 = time.perf_counter()
    oracle.search(s1, s2, WEIGHTS, True, 0, sample, threads=cores)
    dt = time.perf_counter() - t0
    return {"value": sample * m / dt, "unit": "char-pair scores/s", "cores": cores, "kind": "port",
            "sample": f"offsets [0,{sample}) of query 0 (1M x 10k MAX), {cores} OpenMP threads, {dt:.2f} s"}


def csrc_sha256():
    """Hash of the kernel sources the library is built from: the committed PMC counters are only evidence for the
    binary that was profiled (profiles/rNN_meta.json records this hash at profiling time)."""
    h = hashlib.sha256()
    for path in sorted(glob.glob(os.path.join(PKG, "csrc", "*"))):
        if path.endswith((".hip", ".hpp", ".cpp")):
            h.update(os.path.basename(path).encode())
            h.update(open(path, "rb").read())
    return h.hexdigest()


def latest_profile(pattern):
    """The newest committed rocprofv3 summary of that kind (profiles/rNN_*)."""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))
    return files[-1] if files else None


def golden_case(name):
    path = os.path.join(ROOT, "tests", "golden", "synthetic_golden.json")
    for c in json.load(open(path))["cases"]:
        if c["name"] == name:
            return c
    return None


def matches_golden(res, g):
    return (res.offset, res.char_offset, chr(res.ch) if res.ch else "", "%.17g" % res.score) == \
           (g["offset"], g["char_offset"], g["ch"], g["score"])


def stub_main(args, world, rank):
    """Launcher / exchange rehearsal on CPU (tests): gloo, no library compute — every rank makes up the records of
    its queries from the query index, the real exchange code gathers them, rank 0 prints the one JSON line."""
    import torch
    import torch.distributed as dist
    import psa_amd as psa
    if world > 1:
        dist.init_process_group("gloo")
    nq = args.queries
    per_rank = nq if args.scaling == "weak" else None
    total_q = nq * world if args.scaling == "weak" else nq
    lo, hi = (rank * nq, (rank + 1) * nq) if per_rank else query_block(nq, world, rank)
    mine = []
    for q in range(lo, hi):
        r = psa.Result()
        r.score, r.offset, r.char_offset, r.ch, r.shard, r.pairs = -1.5 * q, 7 * q + 1, q % 11, 65 + q % 26, q, 1000 + q
        mine.append(r)
    t0 = time.perf_counter()
    recs = gather_records(psa, dist, torch, mine, lo, total_q, torch.device("cpu")) if world > 1 else mine
    elapsed = time.perf_counter() - t0
    ok = all((r.offset, r.shard, r.pairs) == (7 * q + 1, q, 1000 + q) and r.score == -1.5 * q for q, r in enumerate(recs))
    if rank == 0:
        print(json.dumps({"metric": "stub (launcher rehearsal, no search)", "value": 0.0, "unit": "char-pair scores/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed,
                          "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f64",
                          "data": "stub", "config": {"workload": "stub", "records": len(recs)},
                          "verified": {"exchange": bool(ok)}}), flush=True)
    if world > 1:
        dist.destroy_process_group()
    return 0 if ok else 1


# --------------------------------------------------------------------------------------
# c_abi_group: the library's own multi-GPU product path (one process, psa_group_*)
# --------------------------------------------------------------------------------------
def group_leg_main(args):
    """BASELINE.json configs[3] and [4] through the C ABI's device group, in ONE process (a single process
    drives the node's devices): psa_search_sharded — the offsets split by the reference's rank rule
    (cpu_funcs.c:128-133), every device served by its own persistent launcher thread, one exchange of the 32-byte
    records (RCCL ncclAllGather; cpu_funcs.c:71-94 replaced), psa_fold — and psa_search_batch_sharded (64 and 256 queries
    dealt to the devices).  Answers are checked against a single context on device 0.
    With N >= 2 visible devices the group is devices 0..N-1 with the RCCL exchange (and once more with the host exchange
    for comparison).  On a ONE-GPU box the N-way path cannot run RCCL (one rank per GPU), so it is rehearsed with the host
    exchange over N = 2, 4, 8 contexts on device 0 — that measures the host-side cost per call (issue time of the
    launcher threads), not scaling — next to an RCCL group of the one device.  No torch in this process."""
    import ctypes as C
    import numpy as np
    import psa_amd as psa

    n, m = args.n, args.m
    total = n - m + 1
    ndev_visible = C.c_int(0)
    try:
        hip = C.CDLL("libamdhip64.so")
        hip.hipGetDeviceCount(C.byref(ndev_visible))
    except OSError:
        pass
    have = ndev_visible.value
    if have < 1:
        print(json.dumps({"error": "no HIP device visible"}))
        return 1
    want = args.gpus
    h1 = psa.generate_sequence_np(SEED_SEQ1, n).tobytes()
    nq_big = 256 if (n, m) == (1_000_000, 10_000) else min(256, args.queries or 256)
    hq = [psa.generate_sequence_np(SEED_QUERY0 + q, m).tobytes() for q in range(nq_big)]

    def wall_us(fn, reps, warm=3, idle=False):
        """Median wall time of fn(); idle: every device drained before each call (as the parent's latency probes do with
        torch.cuda.synchronize: a blocking search returns when its result tail lands, a few us before its kernel has
        retired, so calls back to back would each pay the previous kernel's end)."""
        t = []
        for k in range(reps + warm):
            if idle:
                for d in range(have):
                    hip.hipSetDevice(d)
                    hip.hipDeviceSynchronize()
                hip.hipSetDevice(0)
            t0 = time.perf_counter()
            fn()
            if k >= warm:
                t.append(time.perf_counter() - t0)
        return 1e6 * sorted(t)[len(t) // 2]

    # the single-context answers; and what ONE blocking search costs a single context in this process, called the same way
    # the groups are timed below (the bare C entry point, result buffers allocated beforehand): the exchange's cost is the
    # difference to a group's `c_call_us`
    with psa.Context(0) as ctx:
        ctx.set_weights(WEIGHTS, True)
        ctx.set_seq1(h1)
        ctx.set_seq2(hq[0])
        ref0 = ctx.search(0, total)
        one_result = psa.Result()
        single_c_call_us = wall_us(lambda: psa.lib.psa_search(ctx._h, 0, total, C.byref(one_result)), 40, warm=5, idle=True)
        ref_batch = ctx.search_batch(hq)
        ref_hi, ref_lo = ctx.search_batch_minmax(hq[:64])
    key = lambda r: (r.score, r.offset, r.char_offset, r.ch)

    def run_group(devices, exchange, label):
        out = {"devices": list(devices), "exchange": label}
        with psa.Group(devices, exchange) as g:
            g.set_weights(WEIGHTS, True)
            g.set_seq1(h1)
            g.set_seq2(hq[0])
            res, shards = g.search(0, total)
            ok = key(res) == key(ref0) and res.pairs == total * m
            issue = []

            def one():
                g.search(0, total)
                issue.append(g.last_timing()[0])
            blocking = wall_us(one, 24, idle=True)
            issue = sorted(issue[3:])
            best, records = psa.Result(), (psa.Result * len(devices))()
            c_call = wall_us(lambda: psa.lib.psa_search_sharded(g._h, 0, total, C.byref(best), records), 40, warm=5, idle=True)
            out["sharded_search"] = {
                "blocking_us": blocking, "c_call_us": c_call, "single_context_c_call_us": single_c_call_us,
                "group_cost_us": c_call - single_c_call_us,
                "host_issue_us": issue[len(issue) // 2], "host_issue_us_max": issue[-1],
                "pairs_per_s": total * m / (blocking * 1e-6), "equals_single_context": bool(ok),
                "shard_offsets": [psa.shard_range(total, len(devices), r) for r in (0, len(devices) - 1)]}
            batches = {}
            for nq in (64, nq_big):
                got = g.search_batch(hq[:nq])
                same = [key(r) for r in got] == [key(r) for r in ref_batch[:nq]]
                ms = wall_us(lambda: g.search_batch(hq[:nq]), 3, warm=1) / 1e3
                batches[str(nq)] = {"ms": ms, "pairs_per_s": nq * total * m / (ms * 1e-3), "equals_single_context": bool(same)}
            hi, lo = g.search_batch(hq[:64], minmax=True)
            batches["64_minmax_equals_single_context"] = bool([key(r) for r in hi] == [key(r) for r in ref_hi] and
                                                              [key(r) for r in lo] == [key(r) for r in ref_lo])
            out["batch_sharded"] = batches
        out["ok"] = bool(out["sharded_search"]["equals_single_context"] and batches["64_minmax_equals_single_context"] and
                         all(v["equals_single_context"] for k, v in batches.items() if isinstance(v, dict)))
        return out

    legs = []
    if have >= 2 and want >= 2:
        devs = list(range(min(want, have)))
        legs.append(run_group(devs, psa.EXCHANGE_RCCL, "RCCL ncclAllGather over xGMI, one rank per GPU"))
        legs.append(run_group(devs, psa.EXCHANGE_HOST, "host memory (comparison: same devices, no RCCL)"))
        mode = f"{len(devs)} GPUs, one process"
    else:
        legs.append(run_group([0], psa.EXCHANGE_RCCL, "RCCL ncclAllGather, a communicator of ONE device"))
        for k in (2, 4, 8):
            legs.append(run_group([0] * k, psa.EXCHANGE_HOST,
                                  f"host memory: {k} contexts on ONE GPU (rehearsal of the {k}-way path: host-side cost per call, "
                                  f"not scaling — the {k} shard kernels share one GPU)"))
        mode = "one GPU: N-way paths rehearsed with the host exchange"
    print(json.dumps({"mode": mode, "visible_devices": have, "groups": legs, "ok": all(l["ok"] for l in legs),
                      "note": "psa_search_sharded / psa_search_batch_sharded through psa.Group (ctypes): configs[3] = one blocking "
                              "1M x 10k MAX search, offsets sharded; configs[4] = 64 / 256 queries dealt to the devices.  "
                              "blocking_us = through the Python binding, c_call_us / single_context_c_call_us = the bare C entry points "
                              "(psa_search_sharded / psa_search of one context), same process, buffers allocated beforehand, the "
                              "devices idle before every call: c_call_us - single_context_c_call_us is what the group and its "
                              "exchange add; "
                              "host_issue_us = from the call's entry until the LAST device's launch had been enqueued (persistent "
                              "launcher thread per device; one thread issuing 8 launches in turn took ~40 us).  N > 1 with RCCL has "
                              "only ever run where this object says exchange = RCCL with more than one device"}), flush=True)
    return 0


def run_group_leg(args, gpus):
    """Child process (fresh HIP runtime, no torch): a hang or crash of the multi-GPU leg cannot take the headline line down."""
    cmd = [sys.executable, os.path.abspath(__file__), "--group-leg", "--gpus", str(gpus), "--n", str(args.n), "--m", str(args.m)]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK", "ROLE_RANK", "LOCAL_WORLD_SIZE"):
        env.pop(k, None)
    try:
        p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
    except subprocess.TimeoutExpired:
        return {"error": "the c_abi_group child process did not finish within 240 s", "ok": False}
    for ln in reversed(p.stdout.splitlines()):
        if ln.startswith("{"):
            try:
                return json.loads(ln)
            except ValueError:
                break
    return {"error": f"c_abi_group child exited with {p.returncode}: {p.stderr[-600:]}", "ok": False}



def median_us(fn, sync, reps=24, warm=4):
    """Median wall time of fn() in microseconds, device idle before every call."""
    t = []
    for k in range(reps + warm):
        sync()
        t0 = time.perf_counter()
        fn()
        if k >= warm:
            t.append(time.perf_counter() - t0)
    return 1e6 * sorted(t)[len(t) // 2]


def main():
    args = parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.group_leg:
        raise SystemExit(group_leg_main(args))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: the line would misreport n_gpus")
    if args.stub_search:
        raise SystemExit(stub_main(args, world, rank))

    import numpy as np
    import torch
    import psa_amd as psa

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    if args.share_gpu:
        if args.backend != "gloo":
            raise SystemExit("bench.py: --share-gpu is a rehearsal and needs --backend gloo (RCCL wants one GPU per rank)")
        local_rank = 0
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit(f"bench.py: rank {rank} has no device {local_rank} ({torch.cuda.device_count()} visible)")
    dist = None
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)
    xdev = device if args.backend == "nccl" else torch.device("cpu")       # where the exchanged records live
    if world > 1:
        import torch.distributed as dist
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group("gloo")

    n, m = args.n, args.m
    total = n - m + 1
    emulated = args.emulate_world > 1 and world == 1
    weak = args.mode == "batch" and args.scaling == "weak" and not emulated
    eff_world = args.emulate_world if emulated else world
    eff_rank = 0 if emulated else rank
    extras = rank == 0 and world == 1 and not emulated and not args.no_extras
    # queries this rank holds: weak = its own Q (seeds continue across ranks), strong / shard = all Q
    nq = args.queries
    seed0 = SEED_QUERY0 + (rank * nq if weak else 0)
    s1 = psa.generate_sequence_np(SEED_SEQ1, n)
    queries = np.stack([psa.generate_sequence_np(seed0 + q, m) for q in range(nq)])
    t1 = torch.from_numpy(s1).to(device)          # inputs resident in HBM before the timed region
    tq = torch.from_numpy(queries).to(device)
    h1 = s1.tobytes()
    hq_all = [queries[q].tobytes() for q in range(nq)]

    stream = torch.cuda.current_stream(device)
    ctx = psa.Context(local_rank, stream.cuda_stream)
    ctx.set_engine(args.engine)
    ctx.set_batch_streams(args.batch_streams)
    ctx.set_weights(WEIGHTS, True)

    def sync():
        torch.cuda.synchronize(device)

    if args.mode == "batch":
        qlo, qhi = (0, nq) if weak else query_block(nq, eff_world, eff_rank)
        total_q = nq * world if weak else nq                      # records exchanged per step
        slot0 = rank * nq if weak else qlo
        first, last = 0, total
        pairs_step = (qhi - qlo if emulated else total_q) * total * m

        def step():
            ctx.set_seq1_device(t1.data_ptr(), n)          # from raw ASCII: encode, once per batch
            mine = ctx.search_batch_device(tq[qlo].data_ptr(), qhi - qlo, m, m) if qhi > qlo else []
            if world > 1:
                return gather_records(psa, dist, torch, mine, slot0, total_q, xdev)
            return mine
    else:
        qlo, qhi, slot0, total_q = 0, nq, 0, nq
        first, last = psa.shard_range(total, eff_world, eff_rank)
        pairs_step = nq * (last - first if emulated else total) * m

        def step():
            if nq == 1:
                recs = [ctx.search_device(t1.data_ptr(), n, tq[0].data_ptr(), m, first, last)]
            else:                                                 # throughput mode: nq queries in flight on this shard
                ctx.set_seq1_device(t1.data_ptr(), n)
                recs = ctx.search_batch_device(tq.data_ptr(), nq, m, m, first, last)
            if world > 1:
                recs = exchange_and_fold_many(psa, dist, torch, recs, rank, world, True, xdev)
            return recs

    def timed(fn, steps):
        if dist:
            dist.barrier()
        sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            out = fn()
        sync()
        if dist:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        if dist:
            t = torch.tensor([elapsed], dtype=torch.float64, device=xdev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        return elapsed, out

    for _ in range(args.warmup):
        results = step()
    elapsed, results = timed(step, args.steps)
    value = pairs_step * args.steps / elapsed

    # configs[4] asks for MIN+MAX: both searches of every query from one shared class-count pass
    # (psa_search_batch_minmax).  Reported beside the headline, which stays the MAX-only batch.
    minmax = None
    mm = None
    region_8d = None
    from_host_equal = True
    if args.mode == "batch":
        def step_minmax():
            ctx.set_seq1_device(t1.data_ptr(), n)
            hi, lo = ctx.search_batch_minmax_device(tq[qlo].data_ptr(), qhi - qlo, m, m) if qhi > qlo else ([], [])
            if world > 1:
                return gather_records(psa, dist, torch, hi + lo, 2 * slot0, 2 * total_q, xdev)
            return hi + lo
        step_minmax()
        mm_elapsed, mm = timed(step_minmax, args.steps)
        minmax = {"ms_per_step": 1e3 * mm_elapsed / args.steps,
                  "searches_per_s": 2 * (pairs_step // (total * m)) * args.steps / mm_elapsed,
                  "equivalent_pair_scores_per_s": 2 * pairs_step * args.steps / mm_elapsed,
                  "note": "MIN and MAX result of every query per step; the class counts are shared by the two modes, "
                          "bounds / candidates / exact re-score are per mode; 'equivalent' counts the pairs two "
                          "separate searches would score"}

        # SURVEY.md 8(d) / cpu_funcs.c:57-62: the same step with the sequences handed over as pageable HOST buffers
        # (psa_set_seq1 + psa_search_batch: H2D over PCIe inside the timed region).
        hq = hq_all[qlo:qhi]

        def step_from_host():
            ctx.set_seq1(h1)
            return ctx.search_batch(hq) if hq else []
        step_from_host()
        fh_steps = max(2, args.steps // 2)
        fh_elapsed, fh = timed(step_from_host, fh_steps)
        region_8d = {"ms_per_step": 1e3 * fh_elapsed / fh_steps, "value": pairs_step * fh_steps / fh_elapsed,
                     "unit": "char-pair scores/s",
                     "note": "the reference's timed region (cpu_funcs.c:57-62; SURVEY.md 8(d)): inputs in pageable host memory -> "
                             "every best mutant on the host: psa_set_seq1 + psa_search_batch, 1 MB + Q x 10 kB over PCIe per "
                             "step.  THIS `value` is the from-host-buffers (PCIe-inclusive) throughput; the line's top-level "
                             "`value` is the same step from HBM-resident inputs, which the bench contract asks for as headline"}
        from_host_equal = [r.as_tuple() for r in fh] == [r.as_tuple() for r in (results[slot0:slot0 + qhi - qlo] if world > 1 else results)]

    # ---- N > 1, batch mode: configs[3] as well — the OFFSETS of query 0 (and of 64 queries in flight) sharded over the
    # ranks with the reference's rule (cpu_funcs.c:128-133), one exchange of the per-rank records, is_swapable fold
    offset_shard = None
    if world > 1 and args.mode == "batch" and not emulated:
        s_lo, s_hi = psa.shard_range(total, world, rank)
        nb = min(64, nq)
        # every rank searches its shard of the SAME queries (seeds 2002..; the weak-scaling batch gave each rank its own)
        sq = tq if not weak else torch.from_numpy(
            np.stack([psa.generate_sequence_np(SEED_QUERY0 + q, m) for q in range(nb)])).to(device)

        def shard_one():
            rec = ctx.search_device(t1.data_ptr(), n, sq[0].data_ptr(), m, s_lo, s_hi)
            return exchange_and_fold_many(psa, dist, torch, [rec], rank, world, True, xdev)

        def shard_many():
            ctx.set_seq1_device(t1.data_ptr(), n)
            recs = ctx.search_batch_device(sq.data_ptr(), nb, m, m, s_lo, s_hi)
            return exchange_and_fold_many(psa, dist, torch, recs, rank, world, True, xdev)
        shard_one()
        one_steps = 20
        one_elapsed, one_out = timed(shard_one, one_steps)
        shard_many()
        many_steps = max(2, args.steps)
        many_elapsed, many_out = timed(shard_many, many_steps)
        offset_shard = {
            "blocking_search_us": 1e6 * one_elapsed / one_steps,
            "blocking_search_pairs_per_s": total * m * one_steps / one_elapsed,
            "in_flight_queries": nb, "in_flight_ms_per_step": 1e3 * many_elapsed / many_steps,
            "in_flight_pairs_per_s": nb * total * m * many_steps / many_elapsed,
            "result_query0": {"offset": one_out[0].offset, "score": one_out[0].score},
            "in_flight_query0_equals_blocking": (one_out[0].offset, one_out[0].score) == (many_out[0].offset, many_out[0].score),
            "note": f"BASELINE.json configs[3]: offsets of one 1M x 10k MAX search split over {world} ranks (psa_shard_range), every "
                    "rank searches its shard from raw ASCII in HBM, ONE all-reduce of the per-rank 32-byte records, is_swapable "
                    "fold; blocking = one search at a time (latency-bound: ~45 us of a search do not shard), in flight = "
                    f"{nb} queries per step on every rank's shard (throughput).  max over ranks, barrier on both sides"}

    # ---- end-to-end latency of ONE blocking search of query 0 ------------------------------------------------
    # (library profiling off: its event records would sit between the kernels)
    ctx.set_profiling(False)
    lo1, hi1 = (first, last) if args.mode == "shard" else (0, total)
    h2 = hq_all[0]
    p1, p2 = t1.data_ptr(), tq[0].data_ptr()

    def one_search():                       # raw ASCII in HBM -> best mutant on the host: ONE launch
        return ctx.search_device(p1, n, p2, m, lo1, hi1)

    def one_search_encode_first():          # the same as an encode launch + a search launch
        ctx.set_sequences_device(p1, n, p2, m)
        return ctx.search(lo1, hi1)

    def one_search_two_setters():           # the same through the two separate setters (two encode launches)
        ctx.set_seq1_device(p1, n)
        ctx.set_seq2_device(p2, m)
        return ctx.search(lo1, hi1)

    def one_search_from_host():             # host buffers: H2D over PCIe included
        ctx.set_sequences(h1, h2)
        return ctx.search(lo1, hi1)

    single = {"latency_us": median_us(one_search, sync),
              "encode_launch_first_us": median_us(one_search_encode_first, sync, reps=12),
              "two_setters_us": median_us(one_search_two_setters, sync, reps=12)}
    ctx.set_sequences_device(p1, n, p2, m)
    search_only_us = median_us(lambda: ctx.search(lo1, hi1), sync)
    # The same probe right behind a few milliseconds of GPU work: the probes above follow seconds of GPU idleness (the CPU
    # baseline), and isolated 0.08 ms kernels with idle gaps between them do not pull the clocks up — the count loop is
    # VALU-issue-bound, so the difference is the clock (the shard_emulation leg, which runs behind other GPU work,
    # measures the same search a few microseconds faster for the same reason)
    search_only_warm_us = None
    if qhi - qlo >= 2 and not args.no_extras:        # (profiling passes: no launches of another size in the counters' means)
        ctx.set_seq1_device(p1, n)
        for _ in range(3):
            ctx.search_batch_device(tq[qlo].data_ptr(), min(qhi - qlo, 64), m, m, lo1, hi1)
        ctx.set_sequences_device(p1, n, p2, m)
        search_only_warm_us = median_us(lambda: ctx.search(lo1, hi1), sync, reps=16, warm=2)
    single.update({
              "search_only_us": search_only_us, "search_only_behind_gpu_work_us": search_only_warm_us,
              "from_host_buffers_us": median_us(one_search_from_host, sync, reps=12),
              "one_shot_retries": ctx.raw_retry_count(),
              "host_transfers_streamed": ctx.streamed_count(), "host_transfers_redone": ctx.stream_retry_count(),
              "note": "median wall time of blocking calls, device idle before each: latency = psa_search_device (raw ASCII in "
                      "HBM -> result, ONE launch: the count kernel reads the ASCII itself); encode_launch_first = "
                      "psa_set_sequences_device + psa_search (two launches), two_setters = three; search_only = the sequences "
                      "already set (another offset range, another mode); search_only_behind_gpu_work = the same right behind ~10 ms of batch "
                      "work (GPU clocks up); from_host_buffers = psa_set_sequences + psa_search"})
    one = one_search()
    # ... and through the reference's own entry point gpu_run_program (a ProgramData struct in host memory)
    legacy_us = None
    legacy_equal = True
    if rank == 0:
        import ctypes as C
        cap1, cap2 = n + 1, m + 1
        blob = bytearray(40 + cap1 + cap2)
        blob[0:4] = (1).to_bytes(4, "little")
        blob[8:40] = np.asarray(WEIGHTS, np.float64).tobytes()
        blob[40:40 + n] = h1
        blob[40 + cap1:40 + cap1 + m] = h2
        data = (C.c_char * len(blob)).from_buffer(blob)
        mutant = psa.Mutant()
        os.environ.setdefault("PSA_DEVICE", str(local_rank))
        score = [0.0]

        def legacy():
            score[0] = psa.lib.gpu_run_program_layout(C.addressof(data), cap1, cap2, C.byref(mutant), lo1, hi1)
        legacy_us = median_us(legacy, sync, reps=40)          # (12 gave 147 us once on a box whose other probes were normal)
        legacy_equal = (mutant.offset, score[0]) == (one.offset, one.score)

    # ---- the CPU baseline: behind the latency probes (13 s of all-core OpenMP work in front of them cost every probe
    # 5-7 us: 83.7 instead of 78.0 us per blocking search in the same process, gpurun_out/bench_r5d.json / _r5e.json — a
    # host that has just run hot polls and launches more slowly, and the GPU's clocks have gone down), between GPU legs
    # (a utilisation sampler that looks every few seconds sees GPU work before and after)
    cpu = cpu4 = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(h1, hq_all[0], total, m)
        cpu4 = cpu_baseline(h1, hq_all[0], total, m, budget_s=4.0, threads=4)     # the reference's THREADS_COUNT, def.h:18
        # the GPU has idled for ~13 s: the first kernels behind that run at idle clocks (0.119 instead of 0.079 ms per
        # single-search kernel by HIP events, gpurun_out/bench_r5f.json) — a few batches bring the clocks back before
        # the kernel-time probes the roofline is computed from
        if qhi - qlo >= 2:
            ctx.set_seq1_device(p1, n)
            for _ in range(4):
                ctx.search_batch_device(tq[qlo].data_ptr(), qhi - qlo, m, m, lo1, hi1)
            sync()

    # ---- kernels: HIP events recorded by the library on the stream it launches the kernel on ------------------
    # (1) the one kernel of a single search; (2) the dominant kernel of the timed region: in batch mode one launch
    # of the batch count kernel serves several queries — timed alone (a batch of exactly one launch).
    ctx.set_sequences_device(p1, n, p2, m)
    ctx.set_profiling(True)
    for k in range(10):
        one = ctx.search(lo1, hi1)
    _, kernel_ms, profiled = ctx.profile_totals()
    _, single_kernel_name = ctx.last_timing(1)
    single_kernel_s = kernel_ms / max(profiled, 1) / 1e3
    kernel_name, kernel_s, kernel_queries = single_kernel_name, single_kernel_s, 1
    if qhi - qlo >= 2:
        probe = min(qhi - qlo, PROFILE_LAUNCH_QUERIES)          # one launch of the batch kernel, as in the timed region
        times = []
        try:
            for k in range(6):
                ctx.search_batch_device(tq[qlo].data_ptr(), probe, m, m, lo1, hi1)
                t_ms, name = ctx.last_timing(1)
                if k >= 1:
                    times.append(t_ms)
            kernel_name, kernel_queries = name, ctx.last_kernel_queries()
            kernel_s = sorted(times)[len(times) // 2] / 1e3
        except psa.PsaError:
            pass            # the sequential engine's batches are not timed per launch: keep the single-search kernel
    ctx.set_profiling(False)
    pairs_launch = (hi1 - lo1) * m * kernel_queries
    algorithmic_gbs = BYTES_PER_PAIR * pairs_launch / kernel_s / 1e9

    # ---- self-check (rank 0's query 0 is BASELINE.json configs[2]): committed oracle goldens, batch == single search,
    # and the library's own health counters: a retried batch or a fallback inside the timed legs fails the run
    verified = None
    if rank == 0 and args.mode == "batch" and (n, m) == (1_000_000, 10_000) and qlo == 0:
        g_max, g_min = golden_case("c3_max"), golden_case("c3_min")
        mine_max = results[0]
        nmine = qhi - qlo
        mine_min = mm[(2 * slot0 if world > 1 else 0) + nmine] if mm else None
        singles = 0
        for q in range(0, nmine, max(1, nmine // 8)):                       # a few queries against single searches
            ctx.set_seq2_device(tq[q].data_ptr(), m)
            r = ctx.search(0, total)
            b = results[(slot0 if world > 1 else 0) + q]
            if (r.score, r.offset, r.char_offset, r.ch) != (b.score, b.offset, b.char_offset, b.ch):
                singles = -10**6
            singles += 1
        verified = {"query0_max_equals_golden": bool(g_max and matches_golden(mine_max, g_max)),
                    "query0_min_equals_golden": bool(g_min and mine_min is not None and matches_golden(mine_min, g_min)),
                    "batch_equals_single_search_queries": singles,
                    "batch_from_host_equals_batch": bool(from_host_equal),
                    "gpu_run_program_equals_psa_search": bool(legacy_equal),
                    "duo_retries": ctx.duo_retry_count(),
                    "reselects": ctx.reselect_count(),
                    "sequential_engine_fallbacks": ctx.fallback_count(),
                    "second_launch_single_searches": ctx.unfused_count(),
                    "golden": "tests/golden/synthetic_golden.json c3_max / c3_min (CPU oracle, 48 s each)"}

    # ---- extras (one GPU, rank 0): the other engine, the offset shard emulated, a low-entropy input -----------
    sequential = shard_emulation = low_entropy = low_protein = None
    if extras and (n, m) == (1_000_000, 10_000):
        # the literal north_star kernel: lane per offset, fp64 sum in the reference's order for EVERY offset
        ctx.set_engine(psa.ENGINE_SEQUENTIAL)
        ctx.set_sequences_device(p1, n, p2, m)
        seq_res = ctx.search(0, total)
        seq_us = median_us(lambda: ctx.search(0, total), sync, reps=8, warm=2)
        ctx.set_engine(args.engine)
        sequential = {"ms_per_search": seq_us / 1e3, "value": total * m / (seq_us * 1e-6), "unit": "char-pair scores/s",
                      "equals_filter_engine": (seq_res.score, seq_res.offset, seq_res.char_offset, seq_res.ch) ==
                                              (one.score, one.offset, one.char_offset, one.ch),
                      "note": "PSA_ENGINE_SEQUENTIAL (psa_seq_kernel): every offset's fp64 sum in the reference's order, "
                              "LDS pair-score table, wave-shuffle arg-best — the kernel north_star describes; the default "
                              "filter engine returns the same bits"}
        # One rank's share of an N-way offset shard (cpu_funcs.c:128-133) on this one GPU, no exchange: a blocking
        # single search (sequences already on the device) and 64 searches in flight (psa_search_batch on the shard)
        ctx.set_sequences_device(p1, n, p2, m)
        nb = min(64, nq)
        per_search, per_batch = {}, {}
        for w in (1, 2, 4, 8):
            lo_w, hi_w = psa.shard_range(total, w, 0)
            per_search[str(w)] = median_us(lambda: ctx.search(lo_w, hi_w), sync, reps=16)
            if nb >= 2:
                ctx.set_seq1_device(p1, n)
                per_batch[str(w)] = median_us(lambda: ctx.search_batch_device(tq.data_ptr(), nb, m, m, lo_w, hi_w), sync,
                                              reps=5, warm=1) / 1e3
                ctx.set_sequences_device(p1, n, p2, m)
        # ... and all the step's queries in flight on the shard (configs[4]'s per-GPU batch under an offset split): the fixed
        # ~70 us of a batch (launch, planes, the groups' skew, the last epilogue and re-scores) weigh less
        per_big = {}
        if nq >= 128:
            ctx.set_seq1_device(p1, n)
            for w in (1, 8):
                lo_w, hi_w = psa.shard_range(total, w, 0)
                per_big[str(w)] = median_us(lambda: ctx.search_batch_device(tq.data_ptr(), nq, m, m, lo_w, hi_w), sync,
                                            reps=3, warm=1) / 1e3
            ctx.set_sequences_device(p1, n, p2, m)
        shard_emulation = {
            "blocking_search_us_per_rank": per_search,
            # slices of shared combines that a tile's last workgroup had to take over (their own workgroups waited in vain):
            # 0 unless something else was running on this GPU during the probe — the 1/2, 1/4, 1/8 figures are then too high
            "slices_taken_over": ctx.share_sweep_count() if hasattr(ctx, "share_sweep_count") else None,
            "blocking_search_speedup_at_8": per_search["1"] / per_search["8"],
            "batch_of_%d_ms_per_rank" % nb: per_batch,
            "batch_speedup_at_8": (per_batch["1"] / per_batch["8"]) if per_batch else None,
            **({"batch_of_%d_ms_per_rank" % nq: per_big, "batch_of_%d_speedup_at_8" % nq: per_big["1"] / per_big["8"]}
               if per_big else {}),
            "note": "rank 0's shard of an N-way split timed on ONE GPU (N = 1, 2, 4, 8), exchange not included: an "
                    "emulation, not a scaling measurement.  A single 0.1 ms search is dominated by fixed costs (launch, plane "
                    "build, epilogue, exact re-score) and cannot scale near-linearly; searches in flight do"}
        # Plausible low-entropy workloads: a four-letter alphabet (far more near-ties than uniform A..Z, and a running sum
        # that hovers around zero) and the 20 amino-acid letters with their natural frequencies (the draws of
        # tests/golden/make_golden.py --acgt / --protein, whose oracle answers the GPU tests compare with)
        # (all inputs first, then the probes back to back, each behind 20 unmeasured searches of its own: the uniform
        # input is measured in the same state as the others — GPU clocks sag within milliseconds of idleness, and the
        # generation of a 1M-letter draw on the host is 20 ms of it)
        low_entropy, low_protein = {}, {}
        prot = np.asarray([7.8, 5.1, 4.5, 5.4, 1.9, 4.3, 6.3, 7.4, 2.2, 5.1, 9.1, 5.7, 2.2, 3.9, 5.2, 7.1, 5.8, 1.3, 3.2, 6.4])
        inputs = [(None, t1, tq[0])]
        for dest, seed, letters, freq in ((low_entropy, 4, b"ACGT", None), (low_protein, 5, b"ARNDCQEGHILKMFPSTWYV", prot / prot.sum())):
            rng = np.random.default_rng(seed)
            ab = np.frombuffer(letters, np.uint8)
            inputs.append((dest, torch.from_numpy(rng.choice(ab, n, p=freq)).to(device), torch.from_numpy(rng.choice(ab, m, p=freq)).to(device)))
        sync()
        for is_max in (True, False):
            ctx.set_weights(WEIGHTS, is_max)
            samples = [[] for _ in inputs]
            found, counters = [None] * len(inputs), [[0, 0, 0] for _ in inputs]
            for rnd in range(4):                     # the inputs in turn, several rounds: clock drift hits all alike
                for k, (dest, a1, a2) in enumerate(inputs):
                    before = (ctx.fallback_count(), ctx.reselect_count(), ctx.unfused_count())
                    ctx.set_sequences_device(a1.data_ptr(), n, a2.data_ptr(), m)
                    found[k] = ctx.search(0, total)
                    for rep in range(12):
                        sync()
                        t0 = time.perf_counter()
                        ctx.search(0, total)
                        if rep >= 4:
                            samples[k].append(time.perf_counter() - t0)
                    after = (ctx.fallback_count(), ctx.reselect_count(), ctx.unfused_count())
                    for j in range(3):
                        counters[k][j] += after[j] - before[j]
            med = [1e6 * sorted(x)[len(x) // 2] for x in samples]
            for k, (dest, a1, a2) in enumerate(inputs):
                if dest is None:
                    continue
                dest["max" if is_max else "min"] = {
                    "search_only_us": med[k], "uniform_a_z_us": med[0], "ratio_to_uniform_a_z": med[k] / med[0],
                    "offset": found[k].offset, "score": found[k].score,
                    "fallbacks": counters[k][0], "reselects": counters[k][1], "second_launches": counters[k][2]}
        low_entropy["note"] = ("|Seq1|=1M, |Seq2|=10k uniform over ACGT, README weights: same engine, counters = what the searches "
                               "of the probe needed; ratio = to the same probe on the uniform A..Z headline input, taken back to back "
                               "(tests/test_gpu_parity.py checks the answers against committed oracle goldens)")
        low_protein["note"] = "the same with the 20 amino-acid letters at their natural frequencies (Robinson & Robinson)"
        ctx.set_weights(WEIGHTS, True)

    # ---- SURVEY.md 8(d)'s other timed region: process start -> output.txt closed, the drop-in program on configs[2]
    drop_in = None
    exe = os.path.join(PKG, "host", "psa_main")
    if extras and (n, m) == (1_000_000, 10_000) and os.path.exists(exe):
        import tempfile
        with tempfile.TemporaryDirectory() as tmp:
            with open(os.path.join(tmp, "input.txt"), "wb") as f:           # cpu_funcs.c:358-365: weights, seq1, seq2, mode
                f.write(b"1.5 2.6 0.3 0.2\n" + h1 + b"\n" + h2 + b"\nmaximum\n")
            walls, inner, ok = [], [], True
            env = dict(os.environ, PSA_DEVICE=str(local_rank))
            for _ in range(3):
                t0 = time.perf_counter()
                p = subprocess.run([exe], cwd=tmp, env=env, capture_output=True, text=True)
                walls.append(time.perf_counter() - t0)
                try:
                    lines = open(os.path.join(tmp, "output.txt")).read().split("\n")
                    ok = ok and p.returncode == 0 and lines[1] == "%d %g" % (one.offset, one.score) and \
                        lines[0] == h2[:one.char_offset].decode() + chr(one.ch) + h2[one.char_offset + 1:].decode()
                    inner.append(float(p.stdout.split("total time:")[1].split()[0]))
                except (OSError, IndexError, ValueError):
                    ok = False
            drop_in = {"wall_ms": 1e3 * sorted(walls)[1], "program_total_time_ms": 1e3 * sorted(inner)[1] if len(inner) == 3 else None,
                       "output_equals_search": bool(ok),
                       "note": "psa_main on a 1M x 10k input.txt, all offsets on the GPU: process start -> output.txt closed "
                               "(median of 3 runs; HIP runtime + context creation dominate); program_total_time = the program's "
                               "own 'total time' line = the reference's timed region cpu_funcs.c:57-62 in a fresh process"}

    traffic = None
    tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if os.path.exists(tpath):
        traffic = json.load(open(tpath)).get(kernel_name)

    # The bound of the count kernels is VALU issue (bit-sliced counting out of LDS; HBM sees a few MB per launch):
    # wave-instructions per launch from the committed PMC profile (SQ_INSTS_VALU) over the live kernel time, against
    # 1024 SIMDs x 1 wave64 instruction per 2 cycles at 2.4 GHz.  Only if the profile is of THIS source.
    valu = None
    valu_stale = None
    cpath = latest_profile("r*_counters.csv")
    if cpath:
        import csv
        meta_path = cpath.replace("_counters.csv", "_meta.json")
        meta = json.load(open(meta_path)) if os.path.exists(meta_path) else {}
        for row in csv.DictReader(open(cpath)):
            if row["counter"] == "SQ_INSTS_VALU" and kernel_name.split("<")[0] in row["kernel"] and \
                    (("<" not in kernel_name) or kernel_name in row["kernel"]) and "seq" not in os.path.basename(cpath):
                insts = float(row["mean_per_dispatch"])
                if "duo" in kernel_name:
                    insts *= kernel_queries / float(PROFILE_LAUNCH_QUERIES)   # the profile's launches served that many queries each
                valu = {"wave_instructions_per_launch": insts, "achieved_per_s": insts / kernel_s,
                        "source": os.path.relpath(cpath, ROOT)}
                break
        if valu and meta.get("csrc_sha256") != csrc_sha256():
            valu_stale = (f"{os.path.relpath(cpath, ROOT)} was collected from other kernel sources (its "
                          f"{os.path.basename(meta_path)} records csrc_sha256 {str(meta.get('csrc_sha256'))[:12]}, this tree is "
                          f"{csrc_sha256()[:12]}): run tools/profile_round.sh")

    # Calibration of the achievable HBM bandwidth on this GPU: device-to-device copy of 1 GiB,
    # read + write counted.  Beside it, the kernel's MEASURED HBM rate = PMC traffic per launch / duration.
    copy_gbs = None
    if rank == 0:
        src = torch.empty(1 << 30, dtype=torch.uint8, device=device)
        dst = torch.empty_like(src)
        dst.copy_(src)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            dst.copy_(src)
        e1.record()
        torch.cuda.synchronize(device)
        copy_gbs = 10 * 2 * (1 << 30) / (e0.elapsed_time(e1) * 1e-3) / 1e9
        del src, dst

    q0 = results[0]
    if valu and not valu_stale:
        roofline = {"bound": "valu", "achieved": valu["achieved_per_s"] / 1e9, "peak": VALU_PEAK / 1e9,
                    "unit": "G wave64-instructions/s", "frac": valu["achieved_per_s"] / VALU_PEAK}
    elif valu:  # counters exist but belong to another build of the kernels: no fraction is better than a stale one
        roofline = {"bound": "valu", "achieved": None, "peak": VALU_PEAK / 1e9, "unit": "G wave64-instructions/s",
                    "frac": None, "frac_withheld": valu_stale}
    else:       # no committed PMC profile of this kernel: only the algorithmic-bytes model is available
        roofline = {"bound": "hbm", "achieved": algorithmic_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": algorithmic_gbs / HBM_PEAK_GBS}
    roofline.update({
        "traffic": traffic, "kernel": kernel_name, "kernel_ms": 1e3 * kernel_s, "queries_per_launch": kernel_queries,
        "valu": dict(valu, peak_per_s=VALU_PEAK, note="peak = 1024 SIMDs x 2.4 GHz / 2 cycles per full-rate wave64 "
                     "instruction; instructions per launch from the committed rocprofv3 PMC pass.  75 % of the loop is "
                     "v_bitop3_b32, whose 8-byte encoding issues at 1.19 ns per SIMD at this occupancy (tools/ubench/ops3): "
                     "the mix-specific ceiling is ~0.72 of this peak (DESIGN.md 4.4)") if valu and not valu_stale else None,
        "hbm_algorithmic": {"achieved": algorithmic_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "ratio_to_peak": algorithmic_gbs / HBM_PEAK_GBS,
                            "bytes_per_launch": BYTES_PER_PAIR * pairs_launch,
                            "note": "SURVEY.md 8(d) model, 2 B per char pair: NOT binding — tiles are reused from LDS, "
                                    "the kernel's physical HBM traffic is `traffic` bytes per launch"},
        "single_search_kernel": {"kernel": single_kernel_name, "kernel_ms": 1e3 * single_kernel_s,
                                 "note": "one launch = one search: counts, candidates, exact re-score, fold, result tail",
                                 "hbm_algorithmic_GBps": BYTES_PER_PAIR * (hi1 - lo1) * m / single_kernel_s / 1e9},
        "measured_hbm_GBps": (traffic / kernel_s / 1e9) if traffic else None,
        "hbm_copy_calibration_GBps": copy_gbs})
    if args.mode == "batch":
        workload = (f"batch of {nq} independent searches per step and GPU: Seq2 queries (|Seq2|={m}, seeds "
                    f"{SEED_QUERY0}..) vs one |Seq1|={n} (seed {SEED_SEQ1}), MAX, weights 1.5 2.6 0.3 0.2; "
                    f"query 0 = BASELINE.json configs[2]; "
                    + (f"{world} GPU(s) x {nq} queries (BASELINE.json configs[4] batch per GPU)" if weak or world == 1
                       else f"{nq} queries dealt to {world} GPU(s) (BASELINE.json configs[4])"))
    else:
        workload = (f"{nq} search(es) |Seq1|={n} |Seq2|={m} MAX (BASELINE.json configs[2]/[3]) per step, offsets sharded "
                    f"over {world} GPU(s)" + (", all in flight on every rank's shard (throughput mode)" if nq > 1 else ""))
    line = {
        "metric": "char-pair scores/sec (whole job), |Seq1|=1M |Seq2|=10k MAX searches",
        "value": value, "unit": "char-pair scores/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True,
        "scaling": "weak" if (weak or world == 1) else "strong", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {
            "workload": workload,
            "offsets_per_query": total, "queries_per_step": (total_q if args.mode == "batch" else nq),
            "pairs_per_step": pairs_step,
            "parallelism": (f"query-shard x{world}" if args.mode == "batch" else f"offset-shard x{world}"),
            "exchange": "none" if world == 1 else
                        ("1 RCCL all-reduce(sum) of one-hot int64 records per step" if args.backend == "nccl" else
                         "REHEARSAL: gloo all-reduce through host memory" + (", all ranks on one GPU" if args.share_gpu else "")),
            "timed_region": "raw ASCII in HBM -> encode -> search -> records on host (timed_region_8d: the same from host buffers)"},
        "result_query0": {"offset": q0.offset, "char_offset": q0.char_offset,
                          "ch": chr(q0.ch) if q0.ch else "", "score": q0.score},
        "verified": verified,
        "single_search_latency_ms": single["latency_us"] / 1e3,
        "single_search_latency_from_host_buffers_ms": single["from_host_buffers_us"] / 1e3,
        "gpu_run_program_latency_ms": legacy_us / 1e3 if legacy_us else None,
        "single_search": single,
        "roofline": roofline,
    }
    if region_8d:
        line["timed_region_8d"] = region_8d
        line["batch_from_host"] = region_8d                      # (the name earlier rounds used)
    if minmax:
        line["minmax_batch"] = minmax
    if sequential:
        line["sequential_engine"] = sequential
    if shard_emulation:
        line["shard_emulation"] = shard_emulation
    if low_entropy:
        line["low_entropy_acgt"] = low_entropy
        line["low_entropy_protein"] = low_protein
    if drop_in:
        line["drop_in_program"] = drop_in
    if offset_shard:
        line["offset_shard"] = offset_shard
    if emulated:
        line["config"]["emulated_world"] = args.emulate_world
    if cpu:
        line["cpu_baseline"] = cpu
        line["cpu_baseline_4_threads"] = cpu4
    failed = False
    if verified is not None:
        failed = not (verified["query0_max_equals_golden"] and verified["query0_min_equals_golden"] and
                      verified["batch_equals_single_search_queries"] > 0 and verified["batch_from_host_equals_batch"] and
                      verified["gpu_run_program_equals_psa_search"] and verified["duo_retries"] == 0 and
                      verified["sequential_engine_fallbacks"] == 0)
    if sequential and not sequential["equals_filter_engine"]:
        failed = True
    if drop_in and not drop_in["output_equals_search"]:
        failed = True
    # ---- the library's own multi-GPU path (psa_search_sharded / psa_search_batch_sharded: launcher threads + RCCL
    # all-gather inside libpsa_amd.so), driven by ONE child process over all `world` devices, after every rank has
    # finished and the torch process group is gone (rank 0 only; the other ranks are idle or have exited)
    ctx.close()
    if dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0 and not args.no_group_leg and not emulated:
        del t1, tq
        torch.cuda.empty_cache()
        group = run_group_leg(args, world)
        line["c_abi_group"] = group
        # On one GPU this leg is deterministic and tested (its failure is a bug: the run fails).  With N > 1 it is the
        # first time the library's RCCL exchange meets real hardware: a problem there is reported in the object
        # ("ok": false, "error") but does not take the N-rank headline measurement down with it.
        if not group.get("ok", False) and world == 1:
            failed = True
    if rank == 0:
        print(json.dumps(line), flush=True)
        if failed:
            print("bench.py: SELF-CHECK FAILED (see \"verified\" / \"c_abi_group\")", file=sys.stderr)
    if failed:
        raise SystemExit(1)


if __name__ == "__main__":
    main()
