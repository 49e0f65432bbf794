#!/bin/bash
# A/B of library builds / env settings on the full 256-query bench: tools/ab256.sh "<lib-suffix>[:ENV=VAL,...]" ...
LIBDIR=$GRAFT_REPO_ROOT/parallel-sequence-alignment_amd/psa_amd
for rep in 1 2 3; do
for spec in "$@"; do
  v=${spec%%:*}; envs=""
  if [[ "$spec" == *:* ]]; then envs=$(echo "${spec#*:}" | tr ',' ' '); fi
  env $envs PSA_AMD_LIB=$LIBDIR/libpsa_$v.so python bench.py --no-cpu-baseline --no-extras --no-group-leg --steps 3 2>/dev/null > /tmp/ab.json
  python - "$spec" <<'PY'
import json,sys
d=json.load(open("/tmp/ab.json"))
mm=d.get("minmax_batch",{})
print(f"{sys.argv[1]:32s} ms/query {d['ms_per_step']/256:.4f}  minmax ms/query {mm.get('ms_per_step',0)/256:.4f}  value {d['value']:.4g}")
PY
done; done
