"""Address / undefined-behaviour sanitizer runs of the host code (CPU builds only: GPU AddressSanitizer is not
available on this pool).  SURVEY.md §5.2: the reference has real UB on its CPU path (quirks 10, 11); the oracle and
the drop-in program's host scorer must be clean on every stacked record of the reference's input.txt."""
import os
import shutil
import subprocess

import pytest

from conftest import GOLDEN, PKG, ROOT, DISTINCT_RECORDS

SAN = ["-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-fno-sanitize-recover=undefined"]
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")


def _clean(stderr):
    return "AddressSanitizer" not in stderr and "runtime error" not in stderr


def test_oracle_under_sanitizers(tmp_path):
    """oracle/psa_oracle.c through its CLI, -fsanitize=address,undefined, every distinct record (4 threads, 2 ranks)."""
    exe = os.path.join(ROOT, "oracle", "build", "psa_oracle_cli_asan")
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "build/psa_oracle_cli_asan"], check=True,
                   stdout=subprocess.DEVNULL)
    ref = subprocess.run([os.path.join(ROOT, "oracle", "build", "psa_oracle_cli"),
                          os.path.join(GOLDEN, "reference_input.txt"), str(tmp_path / "plain.txt"), "0", "4", "2"],
                         capture_output=True, text=True, timeout=120)
    assert ref.returncode == 0
    for rec in DISTINCT_RECORDS:
        run = subprocess.run([exe, os.path.join(GOLDEN, "reference_input.txt"), str(tmp_path / f"out{rec}.txt"),
                              str(rec), "4", "2"], capture_output=True, text=True, timeout=300, env=ENV)
        assert run.returncode == 0 and _clean(run.stderr), run.stderr[-3000:]
    assert (tmp_path / "out0.txt").read_bytes() == open(os.path.join(GOLDEN, "reference_output.txt"), "rb").read()


def test_drop_in_host_scorer_under_sanitizers(psa, reference_records, oracle, tmp_path):
    """host/psa_main.c (parser, hybrid split with the CPU share only, table-driven scorer, writer) rebuilt with
    -fsanitize=address,undefined: `psa_main 0`, `-100`, `--table` on the bundled input and on the smallest records."""
    from test_host_logic import _write_record, _expected_output
    exe = tmp_path / "psa_main_san"
    lib_dir = os.path.join(PKG, "psa_amd")
    cmd = ["gcc", "-fopenmp", "-std=c11", "-D_POSIX_C_SOURCE=200809L", "-Wall", "-Wextra"] + SAN + [
        "-I", os.path.join(ROOT, "include"), "-o", str(exe), os.path.join(PKG, "host", "psa_main.c"),
        "-L", lib_dir, "-lpsa_amd", f"-Wl,-rpath,{lib_dir}", "-lm"]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    shutil.copy(os.path.join(GOLDEN, "reference_input.txt"), tmp_path / "input.txt")
    for arg in ("0", "-100"):
        run = subprocess.run([str(exe), arg], cwd=tmp_path, capture_output=True, text=True, timeout=300, env=ENV)
        assert run.returncode == 0 and _clean(run.stderr), run.stdout + run.stderr[-3000:]
        assert (tmp_path / "output.txt").read_bytes() == open(os.path.join(GOLDEN, "reference_output.txt"), "rb").read()
    run = subprocess.run([str(exe), "--table"], cwd=tmp_path, capture_output=True, text=True, timeout=60, env=ENV)
    assert run.returncode == 0 and _clean(run.stderr)
    for rec in (8, 10, 11):                    # the tiny records: n - m = 1, trailing blank in the file, 6 vs 5
        d = tmp_path / f"rec{rec}"
        d.mkdir()
        _write_record(d / "input.txt", reference_records[rec])
        run = subprocess.run([str(exe), "0"], cwd=d, capture_output=True, text=True, timeout=60, env=ENV)
        assert run.returncode == 0 and _clean(run.stderr), run.stderr[-3000:]
        assert (d / "output.txt").read_text() == _expected_output(oracle, reference_records[rec])
    # a missing and a malformed input end with exit code 2, not with a sanitizer report
    empty = tmp_path / "empty"
    empty.mkdir()
    run = subprocess.run([str(exe), "0"], cwd=empty, capture_output=True, text=True, timeout=60, env=ENV)
    assert run.returncode == 2 and _clean(run.stderr)
    (empty / "input.txt").write_text("1 2 3\nABC\n")
    run = subprocess.run([str(exe), "0"], cwd=empty, capture_output=True, text=True, timeout=60, env=ENV)
    assert run.returncode == 2 and _clean(run.stderr)


def test_group_launcher_threads_under_thread_sanitizer(tmp_path):
    """csrc/psa_group_workers.hpp — the device group's persistent launcher threads (generation doorbell, hot spin,
    condition-variable sleep, job pointer published through the doorbell) — built with g++ -fsanitize=thread and driven by
    tests/c/group_workers_tsan.cpp: > 10^4 run() rounds over 8 workers, pauses around and beyond the hot-spin time between
    them (both the spin and the sleep / wake-up paths, and the window in which a thread is on its way to sleep while the
    doorbell rings), destroys while hot, asleep and unstarted.  The pool has no HIP call of its own (the device selection is
    an injected callable), so this is the product's code, not a model of it.  VERDICT r4 weak #6, SURVEY.md §5.2."""
    exe = tmp_path / "group_workers_tsan"
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-pthread", "-Wall", "-Wextra",
           "-I", os.path.join(PKG, "csrc"), os.path.join(ROOT, "tests", "c", "group_workers_tsan.cpp"), "-o", str(exe)]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=0:second_deadlock_stack=1")
    run = subprocess.run([str(exe), "12000", "8", "200"], capture_output=True, text=True, timeout=600, env=env)
    assert run.returncode == 0, run.stdout + run.stderr[-3000:]
    assert "ThreadSanitizer" not in run.stderr and "0 failures" in run.stdout, run.stderr[-3000:]
