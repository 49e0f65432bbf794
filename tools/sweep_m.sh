#!/bin/bash
# count-kernel time vs |Seq2| (fixed |Seq1| = 1M): separates the per-workgroup fixed cost from the per-char cost
for m in 512 2560 5120 10240; do
  python bench.py --no-cpu-baseline --steps 2 --mode shard --m $m --engine 2 2>/dev/null > /tmp/sweep.json
  python - <<PY
import json
d=json.load(open("/tmp/sweep.json"))
print($m, d["roofline"]["kernel"], round(d["roofline"]["kernel_ms"],4), round(d["single_search_latency_ms"],4))
PY
done
