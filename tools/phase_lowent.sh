#!/bin/bash
# Phase breakdown of a single 1M x 10k search on uniform A..Z, ACGT and a protein-like 20-letter distribution (both modes).
# Needs psa_amd/libpsa_phase.so (make variant V=phase EXTRA="-DPSA_EXPERIMENT -DPSA_PHASE_TIMING").
L=$GRAFT_REPO_ROOT/parallel-sequence-alignment_amd/psa_amd/libpsa_phase.so
PROT="ARNDCQEGHILKMFPSTWYV:7.8,5.1,4.5,5.4,1.9,4.3,6.3,7.4,2.2,5.1,9.1,5.7,2.2,3.9,5.2,7.1,5.8,1.3,3.2,6.4"
for spec in "|0" "|1" "ACGT|0" "ACGT|1" "$PROT|0" "$PROT|1"; do
  a=${spec%|*}; mn=${spec##*|}
  echo "=== alphabet '${a%%:*}' min=$mn"
  PHASE_ALPHABET="$a" PHASE_MIN=$mn PSA_AMD_LIB=$L python tools/phase_timing.py 2>&1 | grep -v "^rescore\|^raw"
done
