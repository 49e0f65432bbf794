#!/bin/bash
# Same-GPU A/B of library builds on the offset-shard latencies (tools/shard_time.py): alternating runs.
#   tools/shard_ab.sh r4 amd        -> psa_amd/libpsa_r4.so against psa_amd/libpsa_amd.so
LIBDIR=$GRAFT_REPO_ROOT/parallel-sequence-alignment_amd/psa_amd
for rep in 1 2 3; do
for v in "$@"; do
  echo "== $v (run $rep)"
  PSA_AMD_LIB=$LIBDIR/libpsa_$v.so python tools/shard_time.py 2>/dev/null | grep shard
done; done
