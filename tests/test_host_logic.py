"""Host-side logic of libpsa_amd (no GPU): table build vs the oracle's
per-pair functions, the shard split, the fold order, the generator, the ABI."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT

WEIGHT_SETS = ([1, 3, 4, 2], [1.5, 2.6, 0.3, 0.2], [0.8, 0.54, 2.6, 13.7], [2, 1.5, 1.1, 1.3],
               [1, 1, 1, 1], [5, 4, 3, 2], [0, 0, 0, 0], [1e-3, 7.25, 7.25, 0.5])
SYMBOLS = [chr(ord("A") + k) for k in range(26)] + ["-"]


def test_library_exports_every_declared_symbol(psa):
    header = open(os.path.join(ROOT, "include", "psa.h")).read()
    declared = set(re.findall(r"^\s*(?:psa_status|double|void|int|long long|const char\*)\s+(\w+)\s*\(", header, re.M))
    assert declared == set(psa.EXPORTED_SYMBOLS)
    for name in declared:
        assert hasattr(psa.lib, name), name
    assert "gfx950" in psa.version()


@pytest.mark.parametrize("is_max", [True, False])
@pytest.mark.parametrize("w", WEIGHT_SETS)
def test_tables_match_oracle(psa, oracle, w, is_max):
    """P / SUB / DELTA / SIGN for all 27x27 pairs == the oracle's per-pair evaluation
    (get_hashtable_sign, get_weight, get_substitute; cpu_funcs.c:277-285)."""
    t = psa.build_tables(w, is_max)
    for i, a in enumerate(SYMBOLS):
        for j, b in enumerate(SYMBOLS):
            sign = oracle.table_sign(a, b)
            assert chr(t.SIGN[i, j]) == sign
            p = oracle.weight(sign, w)
            assert t.P[i, j] == p
            sub = oracle.substitute(a, b, w, is_max)
            assert chr(t.SUB[i, j]) == sub
            delta = oracle.weight(oracle.table_sign(a, sub), w) - p
            assert t.DELTA[i, j] == delta
    assert not t.SIGN[27:, :].any() and not t.P[27:, :].any() and not t.P[:, 27:].any()


def test_tables_reject_bad_weights(psa):
    with pytest.raises(psa.PsaError) as e:
        psa.build_tables([1, float("nan"), 1, 1], True)
    assert e.value.status == psa.PSA_ERR_ARG


def test_shard_range_is_reference_rule(psa):
    """cpu_funcs.c:128-133: equal slices, last shard takes the remainder."""
    for total in (1, 7, 8, 990001, 99001, 7581):
        for n in (1, 2, 3, 4, 8):
            ranges = [psa.shard_range(total, n, r) for r in range(n)]
            per = total // n
            for r, (lo, hi) in enumerate(ranges):
                assert lo == per * r
                assert hi == lo + per + (total % n if r == n - 1 else 0)
            assert ranges[0][0] == 0 and ranges[-1][1] == total


@pytest.mark.parametrize("nq,nshards", [(3, 8), (7, 4), (100, 8), (256, 8), (1, 1), (5, 5), (9, 2), (257, 3)])
def test_query_blocks_cover_the_batch_and_fit_the_exchange_block(psa, nq, nshards):
    """psa_query_block: contiguous blocks, as even as possible, every block <= psa_query_block_max (the size of a
    device's block in the batch exchange — a block that is larger would overrun it)."""
    blocks = [psa.query_block(nq, nshards, r) for r in range(nshards)]
    per = psa.query_block_max(nq, nshards)
    assert per == -(-nq // nshards)
    assert blocks[0][0] == 0 and blocks[-1][1] == nq
    for (lo, hi), (lo2, _) in zip(blocks, blocks[1:]):
        assert hi == lo2
    sizes = [hi - lo for lo, hi in blocks]
    assert all(0 <= s <= per for s in sizes) and max(sizes) - min(sizes) <= 1
    assert sizes == sorted(sizes, reverse=True)


@pytest.mark.parametrize("minmax", [False, True])
@pytest.mark.parametrize("nq,nshards", [(3, 8), (7, 4), (100, 8), (256, 8), (2, 3)])
def test_batch_exchange_layout_round_trip(psa, nq, nshards, minmax):
    """The batch exchange without devices: every shard packs its block [mode][per] as psa_search_batch_sharded does, a
    fake gather puts the blocks side by side, psa_batch_unpack returns every query's record(s) in query order."""
    per = psa.query_block_max(nq, nshards)

    def rec(q, mode):
        r = psa.Result()
        r.score, r.offset, r.char_offset, r.ch, r.shard, r.pairs = q + 0.25 * mode, 10 * q + mode, q % 7, 65 + q % 26, -7, 1000 + q
        return r

    pad = psa.Result()
    pad.offset, pad.shard = -1, -1
    gathered = []
    for shard in range(nshards):
        lo, hi = psa.query_block(nq, nshards, shard)
        block = [pad] * (2 * per)
        for q in range(lo, hi):
            block[q - lo] = rec(q, 0)
            if minmax:
                block[per + q - lo] = rec(q, 1)
        gathered += block
    out = psa.batch_unpack(gathered, nq, nshards, minmax)
    hi_, lo_ = out if minmax else (out, None)
    assert [(r.score, r.offset, r.shard, r.pairs) for r in hi_] == [(float(q), 10 * q, q, 1000 + q) for q in range(nq)]
    if minmax:
        assert [(r.score, r.offset, r.shard) for r in lo_] == [(q + 0.25, 10 * q + 1, q) for q in range(nq)]


def test_only_the_c_abi_is_exported(psa):
    """Built with -fvisibility=hidden and a version script: the dynamic symbols are the C ABI of psa.h plus the
    reference's C++-mangled name of gpu_run_program (cuda_funcs.h:33; the reference has no extern "C")."""
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", psa.LIB_PATH], check=True, capture_output=True, text=True).stdout
    names = {ln.split()[-1] for ln in out.splitlines() if ln.strip()}
    assert names == set(psa.EXPORTED_SYMBOLS) | {psa.CXX_GPU_RUN_PROGRAM}


def test_test_hooks_live_in_the_debug_library_only(psa, psa_debug):
    """psa_debug_counts / psa_debug_rescore are declared in include/psa_debug.h and exported by
    libpsa_debug.so (-DPSA_EXPERIMENT) only; the product library's dynamic symbols are the reference boundary plus the
    psa_* product calls (test_only_the_c_abi_is_exported)."""
    import subprocess
    header = open(os.path.join(ROOT, "include", "psa_debug.h")).read()
    declared = set(re.findall(r"^\s*(?:psa_status|double|void|int|long long|const char\*)\s+(\w+)\s*\(", header, re.M))
    assert declared == set(psa.DEBUG_SYMBOLS)
    out = subprocess.run(["nm", "-D", "--defined-only", psa_debug.LIB_PATH], check=True, capture_output=True, text=True).stdout
    names = {ln.split()[-1] for ln in out.splitlines() if ln.strip()}
    assert names == set(psa.EXPORTED_SYMBOLS) | set(psa.DEBUG_SYMBOLS) | {psa.CXX_GPU_RUN_PROGRAM}
    assert not any(hasattr(psa.lib, name) for name in psa.DEBUG_SYMBOLS)
    with pytest.raises(psa.PsaError):                       # the binding says where the hook lives instead of crashing
        psa.Context.debug_counts(object.__new__(psa.Context), 0, 1)


def test_plain_symbol_detects_a_caller_with_other_array_limits(psa, tmp_path):
    """gpu_run_program by name reads the stock 10001 / 5001 layout.  A caller that binds it with this repo's
    1,000,001 / 10,001 struct (no psa.h) is told so: no NUL inside seq1[10001] -> message + exit(EXIT_FAILURE), like
    every error of the reference's entry point (cuda_funcs.cu:44-48); a short Seq1 leaves an empty seq2 at the stock
    offset -> message, 'nothing found'.  Both before any device is touched."""
    import subprocess
    import sys
    from conftest import PKG
    script = tmp_path / "wrong_layout.py"
    script.write_text(
        "import ctypes as C, sys\n"
        f"sys.path.insert(0, {PKG!r})\n"
        "import psa_amd as psa\n"
        "n = int(sys.argv[1]); cap1, cap2 = 1000001, 10001\n"
        "blob = bytearray(40 + cap1 + cap2)\n"
        "blob[0:4] = (1).to_bytes(4, 'little')\n"
        "blob[40:40 + n] = b'A' * n\n"
        "blob[40 + cap1:40 + cap1 + 5] = b'ACDEF'\n"
        "data = (C.c_char * len(blob)).from_buffer(blob)\n"
        "m = psa.Mutant()\n"
        "s = psa.lib.gpu_run_program(C.addressof(data), C.byref(m), 0, 10)\n"
        "print('returned', s, m.offset)\n")
    run = subprocess.run([sys.executable, str(script), "20000"], capture_output=True, text=True, timeout=120)
    assert run.returncode == 1 and "other array limits" in run.stderr and "returned" not in run.stdout
    run = subprocess.run([sys.executable, str(script), "300"], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0 and "seq2 is empty" in run.stderr
    assert run.stdout.split()[:3] == ["returned", "-inf", "-1"]


def test_fold_order_matches_is_swapable(psa, oracle):
    """psa_fold == repeated is_swapable (cuda_funcs.cu:290-307), any arrival order."""
    rng = np.random.default_rng(7)
    for is_max in (True, False):
        for _ in range(200):
            k = int(rng.integers(1, 9))
            recs = []
            for shard in range(k):
                r = psa.Result()
                r.score = float(rng.integers(-3, 4))          # many exact ties
                r.offset = int(rng.integers(0, 6)) if rng.random() > 0.15 else -1
                r.char_offset = int(rng.integers(0, 4)) if r.offset >= 0 else -1
                r.ch = ord("A") if r.offset >= 0 else 0
                r.shard = shard
                r.pairs = 10
                recs.append(r)
            got = psa.fold(recs, is_max)
            best_s, best_m = (-np.inf if is_max else np.inf), None
            for r in recs:
                if r.offset < 0:
                    continue
                m2 = oracle.Mutant(r.offset, r.char_offset, b"A")
                if best_m is None or oracle.lib().orc_is_swapable(C.byref(best_m), C.byref(m2), best_s, r.score, int(is_max)):
                    best_s, best_m = r.score, m2
            if best_m is None:
                assert got.offset == -1 and got.ch == 0
            else:
                assert (got.score, got.offset, got.char_offset) == (best_s, best_m.offset, best_m.char_offset)
            assert got.pairs == 10 * k


def test_generator_c_equals_numpy(psa):
    for seed, n in ((0, 1), (1001, 1000), (1002, 4097), (2**63 + 5, 333)):
        a = np.frombuffer(psa.generate_sequence(seed, n), np.uint8)
        b = psa.generate_sequence_np(seed, n)
        assert np.array_equal(a, b)
        assert a.min() >= ord("A") and a.max() <= ord("Z")
    big = psa.generate_sequence_np(1002, 1_000_000)
    counts = np.bincount(big - 65, minlength=26)
    assert counts.min() > 37000 and counts.max() < 40000   # uniform over A..Z (main.c:73,81)


def test_no_device_fails_loudly(psa):
    """Without a usable HIP device there is no fallback: psa_create reports PSA_ERR_DEVICE."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(psa.PsaError) as e:
        psa.Context(0)
    assert e.value.status == psa.PSA_ERR_DEVICE


def test_group_needs_a_device(psa):
    """psa_group_create fails loudly (PSA_ERR_DEVICE) without a HIP device; bad arguments are PSA_ERR_ARG."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a device is present")
    with pytest.raises(psa.PsaError) as e:
        psa.Group([0])
    assert e.value.status == psa.PSA_ERR_DEVICE


def test_null_handles_are_argument_errors_not_crashes(psa):
    """The round-5 additions of the C ABI take a NULL context / group like the older calls do: an error code or a zero
    counter, never a crash (no device needed: every check precedes the first HIP call)."""
    lib = psa.lib
    assert lib.psa_group_set_timeout(None, 1.0) == psa.PSA_ERR_ARG
    assert b"psa_group_set_timeout" in lib.psa_last_error()
    for name in ("psa_scratch_fallback_count", "psa_scratch_shrink_count", "psa_share_pause_count", "psa_fallback_count",
                 "psa_reselect_count", "psa_unfused_count", "psa_duo_retry_count"):
        assert getattr(lib, name)(None) == 0, name
    assert lib.psa_share_sweep_count(None) == 0
    assert lib.psa_group_size(None) == 0
    lib.psa_group_destroy(None)                      # (a no-op)
    out = psa.Result()
    assert lib.psa_search_sharded(None, 0, 1, C.byref(out), None) == psa.PSA_ERR_ARG
    assert lib.psa_search_batch_sharded(None, b"A", 1, 1, 1, 0, 1, C.byref(out), None) == psa.PSA_ERR_ARG


def _write_record(path, rec):
    w, s1, s2, is_max = rec
    with open(path, "w") as f:
        f.write("%s %s %s %s\n%s\n%s\n%s\n" % (repr(w[0]), repr(w[1]), repr(w[2]), repr(w[3]), s1.decode(), s2.decode(),
                                             "maximum" if is_max else "minimum"))


def _expected_output(oracle, rec):
    w, s1, s2, is_max = rec
    score, off, ci, ch = oracle.search(s1, s2, w, is_max, threads=4)
    mut = bytearray(s2)
    mut[ci] = ord(ch)
    return mut.decode() + "\n%d %s" % (off, "%g" % score)


@pytest.mark.parametrize("arg", ["0", "-100"])
def test_drop_in_program_cpu_share_only(psa, oracle, reference_records, tmp_path, arg):
    """`psa_main 0` / `psa_main -100` (main.c:30-42): the whole search on the host threads of the program
    (divide_execute_tasks with cuda_percentage 0, cpu_funcs.c:144-199) — no GPU needed.  output.txt of the bundled
    input byte-for-byte; the other distinct records against the oracle (record 5: sequential-rounding trap)."""
    import shutil
    import subprocess
    from conftest import GOLDEN, PKG, DISTINCT_RECORDS
    exe = os.path.join(PKG, "host", "psa_main")
    shutil.copy(os.path.join(GOLDEN, "reference_input.txt"), tmp_path / "input.txt")
    run = subprocess.run([exe, arg], cwd=tmp_path, capture_output=True, text=True, timeout=120)
    assert run.returncode == 0, run.stdout + run.stderr
    assert (tmp_path / "output.txt").read_bytes() == open(os.path.join(GOLDEN, "reference_output.txt"), "rb").read()
    lines = run.stdout.splitlines()
    assert lines[0] == ("threads= 1, processes= 1" if arg == "-100" else "threads= 4, processes= 1")
    assert lines[1] == "CUDA percentage set to 0"
    for rec in DISTINCT_RECORDS[1:]:
        d = tmp_path / f"rec{rec}"
        d.mkdir()
        _write_record(d / "input.txt", reference_records[rec])
        run = subprocess.run([exe, arg], cwd=d, capture_output=True, text=True, timeout=120)
        assert run.returncode == 0, run.stdout + run.stderr
        assert (d / "output.txt").read_text() == _expected_output(oracle, reference_records[rec]), rec


def test_drop_in_program_sign_table(psa):
    """`psa_main --table`: the 27x27 sign table in print_hash's layout (cpu_funcs.c:322-349) equals the README's
    golden table (README.md:105-133, which shows '_' as blanks) — no device needed."""
    import subprocess
    from conftest import GOLDEN, PKG
    run = subprocess.run([os.path.join(PKG, "host", "psa_main"), "--table"], capture_output=True, text=True, timeout=60)
    assert run.returncode == 0, run.stderr
    got = [ln.replace("_", " ").rstrip() for ln in run.stdout.splitlines()[2:]]
    want = [ln.rstrip() for ln in open(os.path.join(GOLDEN, "readme_sign_table.txt")).read().split("\n") if ln.strip()]
    want = [ln for ln in want if "|" in ln]
    assert len(got) == 27 and got == want
    assert run.stdout.splitlines()[0].split() == list("ABCDEFGHIJKLMNOPQRSTUVWXYZ-")


def test_stock_reference_caller_links(psa, tmp_path):
    """The reference-style callers (their own structs and prototype, no header of this repo) compile and link against
    libpsa_amd.so by symbol name — the link-time half of the drop-in claim (the runs are GPU tests): as C, and as
    C++ without extern "C" the way the reference is built (Makefile:9-12), which refers to the mangled name."""
    import subprocess
    from conftest import PKG, ROOT
    lib_dir = os.path.join(PKG, "psa_amd")
    subprocess.run(["gcc", "-O2", "-std=c11", "-Wall", "-Werror", "-o", str(tmp_path / "stock_caller"),
                    os.path.join(ROOT, "tests", "c", "stock_caller.c"), "-L", lib_dir, "-lpsa_amd",
                    f"-Wl,-rpath,{lib_dir}"], check=True, capture_output=True, text=True)
    exe = tmp_path / "stock_caller_cxx"
    subprocess.run(["g++", "-O2", "-Wall", "-Werror", "-o", str(exe), os.path.join(ROOT, "tests", "c", "stock_caller.cpp"),
                    "-L", lib_dir, "-lpsa_amd", f"-Wl,-rpath,{lib_dir}"], check=True, capture_output=True, text=True)
    undefined = subprocess.run(["nm", "-u", str(exe)], check=True, capture_output=True, text=True).stdout
    assert psa.CXX_GPU_RUN_PROGRAM in undefined


def test_reference_make_targets(tmp_path, psa):
    """`make runseq RUNDIR=...` (the reference's own target name, Makefile:19-20: one host thread, no GPU) on the bundled
    input.txt writes the reference's output.txt byte for byte."""
    import shutil
    import subprocess
    from conftest import GOLDEN, PKG
    shutil.copy(os.path.join(GOLDEN, "reference_input.txt"), tmp_path / "input.txt")
    run = subprocess.run(["make", "-C", PKG, "runseq", f"RUNDIR={tmp_path}"], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, run.stderr[-2000:]
    assert "total time:" in run.stdout
    assert (tmp_path / "output.txt").read_bytes() == open(os.path.join(GOLDEN, "reference_output.txt"), "rb").read()
