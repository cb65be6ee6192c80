cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r02m_gpu_suite.log 2>&1; echo "suite rc=$?" > gpurun_out/r02m_rc.txt
timeout 600 python bench.py > gpurun_out/r02m_bench_cfg1.log 2>&1
timeout 900 python bench.py --config cfg2 > gpurun_out/r02m_bench_cfg2.log 2>&1
timeout 900 python bench.py --config cfg3 > gpurun_out/r02m_bench_cfg3.log 2>&1
timeout 300 python bench.py --precision f32 --no-cpu-baseline > gpurun_out/r02m_bench_cfg1_f32.log 2>&1
timeout 600 bash tools/profile_run.sh r02m_cfg1 both 35 --steps 30 --warmup 5
timeout 400 bash tools/profile_run.sh r02m_cfg2 stats 8 --config cfg2 --steps 6 --warmup 2
timeout 400 bash tools/profile_run.sh r02m_cfg3 stats 5 --config cfg3 --steps 4 --warmup 1
timeout 200 python tools/mha_bench.py > gpurun_out/r02m_mha_bench.log 2>&1
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r02m_smoke.log 2>&1
echo done >> gpurun_out/r02m_rc.txt
