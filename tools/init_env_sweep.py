#!/usr/bin/env python3
"""The drop-in program's timed region (its first gpu_run_program: HIP runtime initialisation dominates) under a few
runtime environment settings, 6 fresh processes each: is any of them worth setting in psa_main before the first HIP call?
  python tools/init_env_sweep.py"""
import os, subprocess, sys, tempfile, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
PKG = os.path.join(ROOT, "parallel-sequence-alignment_amd")
sys.path.insert(0, PKG)
import psa_amd as psa
exe = os.path.join(PKG, "host", "psa_main")
settings = [("default", {}), ("HSA_ENABLE_SDMA=0", {"HSA_ENABLE_SDMA": "0"}), ("HSA_ENABLE_INTERRUPT=0", {"HSA_ENABLE_INTERRUPT": "0"}),
            ("ROCR_VISIBLE_DEVICES=0", {"ROCR_VISIBLE_DEVICES": "0"}), ("HIP_VISIBLE_DEVICES=0", {"HIP_VISIBLE_DEVICES": "0"}),
            ("GPU_MAX_HW_QUEUES=1", {"GPU_MAX_HW_QUEUES": "1"}), ("HSA_TOOLS_LIB= (none)", {"HSA_TOOLS_LIB": ""}),
            ("AMD_LOG_LEVEL=0", {"AMD_LOG_LEVEL": "0"}), ("default again", {})]
with tempfile.TemporaryDirectory() as tmp:
    with open(os.path.join(tmp, "input.txt"), "wb") as f:
        f.write(b"1.5 2.6 0.3 0.2\n" + psa.generate_sequence(1002, 1_000_000) + b"\n" + psa.generate_sequence(2002, 10_000) + b"\nmaximum\n")
    for label, env in settings:
        inner, wall = [], []
        for k in range(6):
            t0 = time.perf_counter()
            p = subprocess.run([exe], cwd=tmp, env=dict(os.environ, **env), capture_output=True, text=True)
            wall.append(time.perf_counter() - t0)
            if "total time:" in p.stdout:
                inner.append(float(p.stdout.split("total time:")[1].split()[0]))
        inner.sort(); wall.sort()
        print(f"{label:28s} total time ms: " + " ".join(f"{1e3 * x:6.1f}" for x in inner) + f"   wall median {1e3 * wall[len(wall) // 2]:6.1f}", flush=True)
