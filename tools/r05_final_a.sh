#!/bin/bash
# Round 5, evidence call A (~14 GPU-minutes): the full GPU suite, the headline's kernel statistics / HBM-traffic PMC passes / SQ
# counter pass / per-shape GEMM table, and the full bench line (CPU baseline, other modes, parity) with roofline.traffic read
# from the PMC summary of THIS call.    gpurun --timeout 1500 -- 'bash tools/r05_final_a.sh'
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r05_pytest_gpu_final.txt 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r05_pytest_gpu_final.txt | cut -c1-200
bash tools/profile_run.sh r05_cfg1 both 35 --steps 30 --warmup 5 > gpurun_out/r05_profile_run.log 2>&1; echo "profile_run rc=$?"
head -12 gpurun_out/r05_cfg1_kernel_stats.csv | cut -c1-150
cat gpurun_out/r05_cfg1_pmc_hbm_traffic.csv | cut -c1-200
cp gpurun_out/r05_cfg1_hbm_traffic.json profiles/r05_cfg1_hbm_traffic.json 2>/dev/null      # bench.py quotes it (roofline.traffic)
cd /tmp
COUNTERS="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES"
timeout 240 rocprofv3 --pmc $COUNTERS -d /tmp/prof_sq -- python $R/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-other-modes --profile-steps 0 --phase-steps 0 > $R/gpurun_out/r05_sq.log 2>&1; echo "pmc sq rc=$?"
db=$(find /tmp/prof_sq -name "*.db" | head -1)
[ -n "$db" ] && python $R/tools/rocpd_to_csv.py counters "$db" /tmp/r05_sq.csv && python $R/tools/pmc_sq_summary.py /tmp/r05_sq.csv 15 $R/gpurun_out/r05_cfg1_sq_counters.csv
head -14 $R/gpurun_out/r05_cfg1_sq_counters.csv | cut -c1-200
cd $R
timeout 200 python tools/gemm_shapes.py 3 cfg1 > gpurun_out/r05_cfg1_gemm_shapes.txt 2>&1; echo "gemm_shapes rc=$?"; head -4 gpurun_out/r05_cfg1_gemm_shapes.txt
timeout 400 python bench.py > gpurun_out/r05_bench_cfg1.json 2> gpurun_out/r05_bench_cfg1.err; echo "bench rc=$?"
grep '^{' gpurun_out/r05_bench_cfg1.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('cfg1', d['value'], 'it/s', d['ms_per_step'], 'ms; engine', r['gemm_ms_per_step'], 'ms', r['achieved'], 'TF frac', r['frac'], 'traffic', r['traffic'])
print('other', {k:(v['value'], v['frac_of_mfma_peak']) for k,v in (d.get('other_precisions') or {}).items()})
print('parity', d.get('parity_vs_oracle'))
print('cpu', d.get('cpu_baseline'))"
