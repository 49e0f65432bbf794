#!/bin/bash
# rocprofv3 passes behind profiles/<tag>_* (run on the GPU box through gpurun; every pass is its own run,
# --pmc never together with a trace domain):   tools/profile_round.sh r03
set -e
tag=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
B="$R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-group-leg"
rocprofv3 --kernel-trace --stats -d $O/prof_${tag}_stats --output-format csv -- python3 $B > $O/prof_${tag}_stats.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/prof_${tag}_twin --output-format csv -- python3 $B --batch-streams 2 > $O/prof_${tag}_twin.log 2>&1
P="$R/bench.py --steps 1 --warmup 1 --queries 128 --no-cpu-baseline --no-extras --no-group-leg"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -d $O/prof_${tag}_pmc1 --output-format csv -- python3 $P > $O/prof_${tag}_pmc1.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS -d $O/prof_${tag}_pmc2 --output-format csv -- python3 $P > $O/prof_${tag}_pmc2.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $O/prof_${tag}_pmc3 --output-format csv -- python3 $P > $O/prof_${tag}_pmc3.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/prof_${tag}_pmc4 --output-format csv -- python3 $P > $O/prof_${tag}_pmc4.log 2>&1
echo profile passes done
