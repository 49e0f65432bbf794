#!/usr/bin/env python3
"""The drop-in program's own timed region (cpu_funcs.c:57-62 = its first gpu_run_program) in a fresh process, with the
debug library's stage stamps (PSA_TRACE_INIT=1): alone on the GPU, and — as bench.py's drop_in_program leg runs it — while
another process (the bench itself: torch + a library context + 10 GB of buffers) holds the same GPU.
  python tools/drop_in_trace.py"""
import os, shutil, subprocess, sys, tempfile, time
import torch   # first (its bundled HIP runtime must be the first one loaded: tests/conftest.py); no device call until below
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
PKG = os.path.join(ROOT, "parallel-sequence-alignment_amd")
sys.path.insert(0, PKG)
exe = os.path.join(PKG, "host", "psa_main")
with tempfile.TemporaryDirectory() as tmp:
    # psa_main finds libpsa_amd.so through its RUNPATH; LD_LIBRARY_PATH comes first: the debug build under that name
    shutil.copy(os.path.join(PKG, "psa_amd", "libpsa_debug.so"), os.path.join(tmp, "libpsa_amd.so"))
    import numpy as np
    import psa_amd as psa          # (only for the generator: no device call in this process yet)
    h1, h2 = psa.generate_sequence(1002, 1_000_000), psa.generate_sequence(2002, 10_000)
    with open(os.path.join(tmp, "input.txt"), "wb") as f:
        f.write(b"1.5 2.6 0.3 0.2\n" + h1 + b"\n" + h2 + b"\nmaximum\n")
    env = dict(os.environ, LD_LIBRARY_PATH=tmp + ":" + os.environ.get("LD_LIBRARY_PATH", ""), PSA_TRACE_INIT="1")

    def run(label, reps=3):
        for k in range(reps):
            t0 = time.perf_counter()
            p = subprocess.run([exe], cwd=tmp, env=env, capture_output=True, text=True)
            wall = time.perf_counter() - t0
            total = p.stdout.split("total time:")[1].split()[0] if "total time:" in p.stdout else "?"
            print(f"== {label}, run {k}: wall {1e3 * wall:.1f} ms, the program's total time {1e3 * float(total):.1f} ms")
            if k == reps - 1:
                print(p.stderr.strip())

    run("alone on the GPU")
    hold = torch.empty(10 << 30, dtype=torch.uint8, device="cuda")      # what the bench process holds meanwhile
    ctx = psa.Context(0)
    ctx.set_weights([1.5, 2.6, 0.3, 0.2], True)
    ctx.set_sequences(h1, h2)
    ctx.search()
    torch.cuda.synchronize()
    run("next to a process that holds the GPU (torch context, 10 GB, a library context)")
