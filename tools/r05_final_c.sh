#!/bin/bash
# Round 5, last GPU call: the full GPU suite and the bench lines of the FINAL tree (after the descriptor / scalar-epilogue fix).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r05_pytest_gpu_final.txt 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" gpurun_out/r05_pytest_gpu_final.txt | tail -2
grep -E "independent-oracle|image_rel_l2_max" gpurun_out/r05_pytest_gpu_final.txt | head -3
timeout 400 python bench.py > gpurun_out/r05_bench_cfg1.json 2> gpurun_out/r05_bench_cfg1.err; echo "bench rc=$?"
grep '^{' gpurun_out/r05_bench_cfg1.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('cfg1', d['value'], 'it/s', d['ms_per_step'], 'ms; engine', r['gemm_ms_per_step'], 'ms', r['achieved'], 'TF frac', r['frac'], 'traffic', r['traffic'])
print('other', {k:(v['value'], v['frac_of_mfma_peak']) for k,v in (d.get('other_precisions') or {}).items()})
print('parity', {k:v for k,v in (d.get('parity_vs_oracle') or {}).items() if k!='what'})
print('cpu', (d.get('cpu_baseline') or {}).get('value'))"
timeout 300 python bench.py --config cfg2 --no-cpu-baseline > gpurun_out/r05_bench_cfg2.json 2> gpurun_out/r05_bench_cfg2.err; echo "bench cfg2 rc=$?"
grep '^{' gpurun_out/r05_bench_cfg2.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('cfg2', d['value'], 'it/s', d['ms_per_step'], 'ms; engine', r['gemm_ms_per_step'], 'ms', r['achieved'], 'TF frac', r['frac'])"
