#!/bin/bash
# A/B of library builds / env settings in one gpurun call (same GPU): alternating runs.
#   tools/ab.sh "<lib-suffix>[:ENV=VAL,...]" ...     e.g. tools/ab.sh prev amd:PSA_BATCH_STREAMS=1 amd
LIBDIR=$GRAFT_REPO_ROOT/parallel-sequence-alignment_amd/psa_amd
for rep in 1 2 3; do
for spec in "$@"; do
  v=${spec%%:*}; envs=""
  if [[ "$spec" == *:* ]]; then envs=$(echo "${spec#*:}" | tr ',' ' '); fi
  env $envs PSA_AMD_LIB=$LIBDIR/libpsa_$v.so python bench.py --no-cpu-baseline --steps 3 --queries 64 2>/dev/null > /tmp/ab.json
  python - "$spec" <<'PY'
import json,sys
d=json.load(open("/tmp/ab.json"))
print(f"{sys.argv[1]:32s} kernel_ms {d['roofline']['kernel_ms']:.4f}  ms/query {d['ms_per_step']/64:.4f}  lat {d['single_search_latency_ms']:.4f}  value {d['value']:.4g}")
PY
done; done
