#!/bin/bash
# count-kernel shapes x chunk counts: full 1M x 10k search and the 8-way offset shard
for cfg in "0 0 1" "1 0 1" "0 0 8" "1 0 8" "0 2 8" "0 4 8" "1 2 8" "1 4 8"; do
  set -- $cfg
  PSA_FILTER_SHAPE=$1 PSA_FILTER_CHUNKS=$2 python bench.py --no-cpu-baseline --steps 5 --mode shard --emulate-world $3 --engine 2 2>/dev/null > /tmp/sweep.json
  python - <<PY
import json
d=json.load(open("/tmp/sweep.json"))
print("shape $1 chunks $2 world $3:", d["roofline"]["kernel"], "kernel_ms", round(d["roofline"]["kernel_ms"],4), "search_ms", round(d["single_search_latency_ms"],4), "step_ms", round(d["ms_per_step"],4))
PY
done
