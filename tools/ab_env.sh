#!/bin/bash
# A/B of experiment-knob settings in one gpurun call: tools/ab_env.sh "A=1,B=2" "A=0" ...
# (the knobs exist only in the debug library, which this script selects)
export PSA_AMD_LIB=${PSA_AMD_LIB:-$GRAFT_REPO_ROOT/parallel-sequence-alignment_amd/psa_amd/libpsa_debug.so}
for rep in 1 2 3; do
for spec in "$@"; do
  envs=$(echo "$spec" | tr ',' ' ')
  env $envs python bench.py --no-cpu-baseline --steps 3 --queries 64 2>/dev/null > /tmp/ab.json
  python - "$spec" <<'PY'
import json,sys
d=json.load(open("/tmp/ab.json"))
mm=d.get("minmax_batch",{})
print(f"{sys.argv[1]:40s} ms/query {d['ms_per_step']/64:.4f}  minmax ms/query {mm.get('ms_per_step',0)/64:.4f}  lat {d['single_search_latency_ms']:.4f}")
PY
done; done
