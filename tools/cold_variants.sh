#!/bin/bash
# First gpu_run_program of a fresh process: the library's stage stamps (debug library, PSA_TRACE_INIT), with the context's
# own stream (product behaviour), the null stream and hipStreamPerThread (experiments: what a created stream costs).
L=$GRAFT_REPO_ROOT/parallel-sequence-alignment_amd/psa_amd/libpsa_debug.so
for how in "" null perthread; do
  echo "=== PSA_MAIN_STREAM='$how'"
  if [ -z "$how" ]; then PSA_TRACE_INIT=1 PSA_AMD_LIB=$L python tools/cold_start.py 2>&1
  else PSA_MAIN_STREAM=$how PSA_TRACE_INIT=1 PSA_AMD_LIB=$L python tools/cold_start.py 2>&1; fi
done
