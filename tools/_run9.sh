cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_f32_mode_gpu.py -x -q -m gpu > gpurun_out/r02i_tests.log 2>&1; echo "kern rc=$?" > gpurun_out/r02i_rc.txt
timeout 900 python -m pytest tests/test_e2e_gpu.py tests/test_path_gpu.py -x -q -m gpu -k "cfg2 or config2 or config3 or ensemble or resnet or style or rn50" > gpurun_out/r02i_e2e.log 2>&1; echo "e2e rc=$?" >> gpurun_out/r02i_rc.txt
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r02i_bench_cfg1.log 2>&1
timeout 300 python bench.py --config cfg2 --no-cpu-baseline > gpurun_out/r02i_bench_cfg2.log 2>&1
timeout 400 python bench.py --config cfg3 --no-cpu-baseline > gpurun_out/r02i_bench_cfg3.log 2>&1
timeout 400 bash tools/profile_run.sh r02i_cfg2 stats 8 --config cfg2 --steps 6 --warmup 2
timeout 400 bash tools/profile_run.sh r02i_cfg3 stats 5 --config cfg3 --steps 4 --warmup 1
echo done >> gpurun_out/r02i_rc.txt
