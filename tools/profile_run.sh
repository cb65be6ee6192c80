#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 kernel statistics and, optionally, the two HBM-traffic PMC passes of one
# bench.py command; summaries land in gpurun_out/<tag>_* (copy what should be judged into profiles/).
#   tools/profile_run.sh <tag> stats|pmc|both <iterations in the trace> [bench.py args...]
# PMC passes are separate runs with --pmc only (MI355X_MICROARCH.md, rocprofv3 section; gpurun refuses --pmc combined with
# the API traces).
tag=$1; what=$2; iters=$3; shift 3
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
cd /tmp
todb() { find "$1" -name "*.db" | head -1; }
# a box that is handed out again keeps its /tmp: an output directory left by an earlier call of the same tag would be read instead of this run's
rm -rf /tmp/prof_${tag}_k /tmp/prof_${tag}_FETCH_SIZE /tmp/prof_${tag}_WRITE_SIZE /tmp/${tag}_raw_stats.csv /tmp/${tag}_FETCH_SIZE.csv /tmp/${tag}_WRITE_SIZE.csv
if [ "$what" = stats ] || [ "$what" = both ]; then
    rocprofv3 --kernel-trace --stats -d /tmp/prof_${tag}_k -- python $R/bench.py "$@" --no-cpu-baseline --no-other-modes --profile-steps 0 --phase-steps 0 > $R/gpurun_out/${tag}_rocprof_stats.log 2>&1
    db=$(todb /tmp/prof_${tag}_k)
    [ -n "$db" ] && python $R/tools/rocpd_to_csv.py stats "$db" /tmp/${tag}_raw_stats.csv && python $R/tools/kernel_stats_summary.py /tmp/${tag}_raw_stats.csv $iters $R/gpurun_out/${tag}_kernel_stats.csv
fi
if [ "$what" = pmc ] || [ "$what" = both ]; then
    for c in FETCH_SIZE WRITE_SIZE; do
        rocprofv3 --pmc $c -d /tmp/prof_${tag}_$c -- python $R/bench.py "$@" --no-cpu-baseline --no-other-modes --profile-steps 0 --phase-steps 0 > $R/gpurun_out/${tag}_rocprof_$c.log 2>&1
        db=$(todb /tmp/prof_${tag}_$c)
        [ -n "$db" ] && python $R/tools/rocpd_to_csv.py counters "$db" /tmp/${tag}_$c.csv
    done
    python $R/tools/pmc_summary.py /tmp/${tag}_FETCH_SIZE.csv /tmp/${tag}_WRITE_SIZE.csv $iters $R/gpurun_out/${tag}_pmc_hbm_traffic.csv $R/gpurun_out/${tag}_hbm_traffic.json
fi
