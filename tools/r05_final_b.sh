#!/bin/bash
# Round 5, evidence call B (~10 GPU-minutes): configs[2] / configs[3] bench lines and kernel statistics, the modelled strong-scaling
# table (one GPU running the per-rank shard with every collective of the sharded iteration in place), the stand-alone fit-kernel
# tables and the launch floor.    gpurun --timeout 1200 -- 'bash tools/r05_final_b.sh'
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
export TMPDIR=/tmp
cd $R
for c in cfg2 cfg3; do
  timeout 400 python bench.py --config $c > gpurun_out/r05_bench_$c.json 2> gpurun_out/r05_bench_$c.err; echo "bench $c rc=$?"
  grep '^{' gpurun_out/r05_bench_$c.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}
print('$c', d['value'], 'it/s', d['ms_per_step'], 'ms; engine', r.get('gemm_ms_per_step'), 'ms', r.get('achieved'), 'TF frac', r.get('frac'), d['config'].get('launch'))"
done
bash tools/profile_run.sh r05_cfg3 stats 12 --config cfg3 --steps 8 --warmup 4 > gpurun_out/r05_profile_run_cfg3.log 2>&1; echo "profile cfg3 rc=$?"
head -16 gpurun_out/r05_cfg3_kernel_stats.csv | cut -c1-150
bash tools/profile_run.sh r05_cfg2 stats 14 --config cfg2 --steps 10 --warmup 4 > gpurun_out/r05_profile_run_cfg2.log 2>&1; echo "profile cfg2 rc=$?"
for n in 64 32 16 8; do
  PRX_FORCE_DIST=1 timeout 200 python bench.py --cutn $n --steps 40 --warmup 8 --no-cpu-baseline --no-other-modes --profile-steps 0 --phase-steps 0 > gpurun_out/r05_modelled_cutn$n.json 2>/dev/null
  grep '^{' gpurun_out/r05_modelled_cutn$n.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('modelled cutn $n', d['value'], 'it/s', d['ms_per_step'], 'ms', d.get('collectives_ms_per_step'))"
done
timeout 150 python tools/fit_bench.py > gpurun_out/r05_fit_bench.txt 2>&1; echo "fit_bench rc=$?"
timeout 150 python tools/fit_conv_bench.py > gpurun_out/r05_fit_conv_bench.txt 2>&1; echo "fit_conv_bench rc=$?"
timeout 150 python tools/lib_gemm_compare.py > gpurun_out/r05_lib_gemm_compare.txt 2>&1; echo "lib_gemm_compare rc=$?"
