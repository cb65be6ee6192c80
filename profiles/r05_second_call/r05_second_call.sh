#!/bin/bash
# Round 5, SECOND gpurun call (~8 GPU-minutes): where the time between and inside the fit kernels goes.
#   /usr/local/graft/bin/gpurun --timeout 700 -- 'bash tools/r05_second_call.sh'
# 1. tools/launch_floor.py: the cost of a kernel boundary on one stream (eager / hipGraph replay) under the runtime's dispatch knobs.
# 2. an SQ stall-reason PMC pass (one pass, counters only) of the headline bench for the default library and for whichever of the
#    first call's variants won (set VARIANT_ENV below), summarised per kernel by tools/pmc_sq_summary.py.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
export TMPDIR=/tmp
VARIANT_ENV=${VARIANT_ENV:-"PRX_FIT_FLAGS=65"}
timeout 200 python $R/tools/launch_floor.py > $R/gpurun_out/r05_launch_floor.log 2>&1; echo "launch_floor rc=$?"
cat $R/gpurun_out/r05_launch_floor.log
cd /tmp
COUNTERS="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES"
for tag in base variant; do
    if [ $tag = variant ]; then export $VARIANT_ENV; fi
    timeout 240 rocprofv3 --pmc $COUNTERS -d /tmp/prof_sq_$tag -- python $R/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-other-modes --profile-steps 0 --phase-steps 0 > $R/gpurun_out/r05_sq_$tag.log 2>&1; echo "pmc $tag rc=$?"
    db=$(find /tmp/prof_sq_$tag -name "*.db" | head -1)
    [ -n "$db" ] && python $R/tools/rocpd_to_csv.py counters "$db" /tmp/r05_sq_$tag.csv && python $R/tools/pmc_sq_summary.py /tmp/r05_sq_$tag.csv 15 $R/gpurun_out/r05_cfg1_sq_$tag.csv
    head -12 $R/gpurun_out/r05_cfg1_sq_$tag.csv 2>/dev/null | cut -c1-200
done
