#!/bin/bash
# Round 6, the evidence call: everything the round's numbers are quoted from, taken on ONE box from ONE tree in ONE gpurun call.
#   gpurun --timeout 3000 -- 'bash tools/r06_final.sh <tree-hash>'
# Writes gpurun_out/r06_*; every file starts with / carries the tree hash it was taken from.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
H=${1:-unknown}
mkdir -p $R/gpurun_out
export TMPDIR=/tmp
cd $R
hdr() { echo "# tree $H; $(date -u +%FT%TZ); $1"; }
# 1. the GPU suite
( hdr "python -m pytest tests -m gpu -q"; timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -12 ) > gpurun_out/r06_pytest_gpu_final.txt; tail -2 gpurun_out/r06_pytest_gpu_final.txt | cut -c1-200
# 2. smoke
( hdr "python __graft_entry__.py smoke"; timeout 600 python __graft_entry__.py smoke 2>&1 | grep smoke ) > gpurun_out/r06_smoke_final.txt; cut -c1-300 gpurun_out/r06_smoke_final.txt | tail -3
# 3. kernel statistics of the three configurations + the headline's HBM-traffic PMC passes (separate --pmc runs)
bash tools/profile_run.sh r06_cfg1 both 35 --steps 30 --warmup 5 > gpurun_out/r06_profile_cfg1.log 2>&1
cp gpurun_out/r06_cfg1_hbm_traffic.json profiles/r06_cfg1_hbm_traffic.json 2>/dev/null      # bench.py quotes it (roofline.traffic), labelled as a profile
bash tools/profile_run.sh r06_cfg1_ref stats 12 --precision ref --steps 10 --warmup 2 > gpurun_out/r06_profile_cfg1_ref.log 2>&1
bash tools/profile_run.sh r06_cfg2 stats 8 --config cfg2 --steps 6 --warmup 2 > gpurun_out/r06_profile_cfg2.log 2>&1
bash tools/profile_run.sh r06_cfg3 stats 8 --config cfg3 --steps 6 --warmup 2 --no-graph > gpurun_out/r06_profile_cfg3.log 2>&1
for f in r06_cfg1 r06_cfg1_ref r06_cfg2 r06_cfg3; do
  [ -f gpurun_out/${f}_kernel_stats.csv ] && sed -i "1i # tree $H" gpurun_out/${f}_kernel_stats.csv && head -6 gpurun_out/${f}_kernel_stats.csv | cut -c1-160
done
[ -f gpurun_out/r06_cfg1_pmc_hbm_traffic.csv ] && sed -i "1i # tree $H" gpurun_out/r06_cfg1_pmc_hbm_traffic.csv && cat gpurun_out/r06_cfg1_pmc_hbm_traffic.csv | cut -c1-200 | head -8
# 4. SQ counters of the headline (one --pmc pass)
cd /tmp
COUNTERS="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES"
rm -rf /tmp/prof_sq /tmp/r06_sq.csv
timeout 300 rocprofv3 --pmc $COUNTERS -d /tmp/prof_sq -- python $R/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-other-modes --profile-steps 0 --phase-steps 0 > $R/gpurun_out/r06_sq.log 2>&1
db=$(find /tmp/prof_sq -name "*.db" | head -1)
[ -n "$db" ] && python $R/tools/rocpd_to_csv.py counters "$db" /tmp/r06_sq.csv && python $R/tools/pmc_sq_summary.py /tmp/r06_sq.csv 15 $R/gpurun_out/r06_cfg1_sq_counters.csv && sed -i "1i # tree $H" $R/gpurun_out/r06_cfg1_sq_counters.csv
cd $R
# 5. device-side phase stamps of the fit kernels, fp16 and ref
( hdr "tools/fit_trace.py"; timeout 300 python tools/fit_trace.py --out gpurun_out/r06_fit_trace_final.csv 2>&1 ) > gpurun_out/r06_fit_trace_final.txt
( hdr "tools/fit_trace.py --precision ref"; timeout 300 python tools/fit_trace.py --precision ref --iters 2 2>&1 ) > gpurun_out/r06_fit_trace_final_ref.txt
# 6. the bench lines: headline in the driver's format (CPU baseline, other modes, parity), ref as the timed leg, configs[2], configs[3]
( timeout 600 python bench.py > gpurun_out/r06_bench_cfg1.json ) 2> gpurun_out/r06_bench_cfg1.err
( timeout 300 python bench.py --precision ref --steps 30 --warmup 5 --no-other-modes --no-cpu-baseline > gpurun_out/r06_bench_cfg1_ref.json ) 2> gpurun_out/r06_bench_cfg1_ref.err
( timeout 900 python bench.py --config cfg2 > gpurun_out/r06_bench_cfg2.json ) 2> gpurun_out/r06_bench_cfg2.err
( timeout 900 python bench.py --config cfg3 > gpurun_out/r06_bench_cfg3.json ) 2> gpurun_out/r06_bench_cfg3.err
for c in cfg1 cfg1_ref cfg2 cfg3; do
  grep '^{' gpurun_out/r06_bench_$c.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}
print('$c', d['dtype'], d['value'], 'it/s', d['ms_per_step'], 'ms; engine', r.get('gemm_ms_per_step'), 'ms', r.get('achieved'), 'TF frac', r.get('frac'), 'traffic', r.get('traffic'))
print('   other', {k:(v['value'], v['frac_of_mfma_peak']) for k,v in (d.get('other_precisions') or {}).items()})
print('   parity', d.get('parity_vs_oracle'))
print('   cpu', (d.get('cpu_baseline') or {}).get('value'), (d.get('cpu_baseline') or {}).get('cores'))"
done
# 7. what one rank of an N-GPU run does (single GPU, forced 1-rank group: the collectives are issued and timed): the modelled table
for n in 64 32 16 8; do
  ( timeout 300 env PRX_FORCE_DIST=1 MASTER_PORT=2957$((n % 10)) python bench.py --cutn $n --steps 30 --warmup 5 --profile-steps 3 --no-cpu-baseline --no-other-modes > gpurun_out/r06_modelled_cutn$n.json ) 2> gpurun_out/r06_modelled_cutn$n.err
  grep '^{' gpurun_out/r06_modelled_cutn$n.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('cutn $n', d['value'], 'it/s', d['ms_per_step'], 'ms; collectives', d.get('collectives_ms_per_step'))"
done
# 8. configs[2]: the engine's per-shape table with the row-streaming kernels (gemmrow*.hip) and on the tiled kernels they replaced, and
#    the same-box bench line of the round's starting arithmetic layout (tiled kernels, fp32 residual streams in the ModifiedResNet runner)
( hdr "tools/gemm_shapes.py 3 cfg2"; timeout 600 python tools/gemm_shapes.py 3 cfg2 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r06_cfg2_gemm_shapes_row.txt
( hdr "PRX_GEMM_ROWK=0 tools/gemm_shapes.py 3 cfg2"; PRX_GEMM_ROWK=0 timeout 600 python tools/gemm_shapes.py 3 cfg2 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r06_cfg2_gemm_shapes_tiled.txt
( PRX_GEMM_ROWK=0 PRX_RN_LEAN=0 timeout 900 python bench.py --config cfg2 --no-cpu-baseline --no-other-modes > gpurun_out/r06_bench_cfg2_tiled_fp32streams.json ) 2> gpurun_out/r06_bench_cfg2_tiled_fp32streams.err
grep '^{' gpurun_out/r06_bench_cfg2_tiled_fp32streams.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}
print('cfg2 tiled kernels + fp32 streams (same box)', d['value'], 'it/s', d['ms_per_step'], 'ms; engine', r.get('gemm_ms_per_step'))"
head -12 gpurun_out/r06_cfg2_gemm_shapes_row.txt
