#!/bin/bash
# A/B of two builds of the library in one process group (same GPU): alternating runs
for rep in 1 2 3; do
for v in prev amd; do
  PSA_AMD_LIB=$GRAFT_REPO_ROOT/parallel-sequence-alignment_amd/psa_amd/libpsa_$v.so python bench.py --no-cpu-baseline --steps 3 --queries 32 2>/dev/null > /tmp/ab.json
  python - <<PY
import json
d=json.load(open("/tmp/ab.json"))
print("$v", "kernel_ms", round(d["roofline"]["kernel_ms"],4), "ms/query", round(d["ms_per_step"]/32,4), "lat", round(d["single_search_latency_ms"],4))
PY
done; done
