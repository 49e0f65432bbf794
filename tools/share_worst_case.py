#!/usr/bin/env python3
"""The shared combine's worst case: every waiting chunk workgroup leaves at once (debug library, PSA_SHARE_SPINS=0), so every
tile's last arriver does its own slice and then all the others in one pass — against the normal case and against no
sharing at all (PSA_SHARE=0), 1/8 and 1/4 offset shards of 1M x 10k.
  PSA_AMD_LIB=.../libpsa_debug.so python tools/share_worst_case.py"""
import os, subprocess, sys, time
if len(sys.argv) > 1:
    import torch
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "parallel-sequence-alignment_amd"))
    import psa_amd as psa
    n, m = 1_000_000, 10_000
    with psa.Context(0) as c:
        c.set_weights([1.5, 2.6, 0.3, 0.2], True)
        c.set_sequences(psa.generate_sequence(1002, n), psa.generate_sequence(2002, m))
        for w in (4, 8):
            lo, hi = psa.shard_range(n - m + 1, w, 0)
            ts = []
            for k in range(40):
                torch.cuda.synchronize(); t0 = time.perf_counter(); c.search(lo, hi); ts.append(time.perf_counter() - t0)
            print(f"  1/{w} shard: blocking {1e6 * sorted(ts[8:])[16]:.1f} us; slices taken over so far {c.share_sweep_count()}; "
                  f"reselects {c.reselect_count()}, second launches {c.unfused_count()}, sequential fallbacks {c.fallback_count()}")
    sys.exit(0)
for label, env in (("normal", {}),
                   ("every owner leaves, every launch shares (PSA_SHARE_SPINS=0 PSA_SHARE_PAUSE=0)", {"PSA_SHARE_SPINS": "0", "PSA_SHARE_PAUSE": "0"}),
                   ("every owner leaves, the context reacts: 64 launches without sharing after such a search (PSA_SHARE_SPINS=0)", {"PSA_SHARE_SPINS": "0"}),
                   ("no sharing (PSA_SHARE=0)", {"PSA_SHARE": "0"})):
    print(label, flush=True)
    subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, **env))
