cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python tools/cutout_bench.py > gpurun_out/r02h_cutbench.log 2>&1
timeout 900 python -m pytest tests/test_path_gpu.py tests/test_determinism_gpu.py tests/test_f32_mode_gpu.py -x -q -m gpu > gpurun_out/r02h_tests.log 2>&1; echo "path rc=$?" > gpurun_out/r02h_rc.txt
timeout 600 python -m pytest tests/test_e2e_gpu.py -x -q -m gpu -k "cfg2 or config2 or ensemble or resnet" > gpurun_out/r02h_e2e.log 2>&1; echo "e2e rc=$?" >> gpurun_out/r02h_rc.txt
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r02h_bench_cfg1.log 2>&1
timeout 300 python bench.py --config cfg2 --no-cpu-baseline > gpurun_out/r02h_bench_cfg2.log 2>&1
timeout 400 bash tools/profile_run.sh r02h_cfg2 stats 8 --config cfg2 --steps 6 --warmup 2
echo done >> gpurun_out/r02h_rc.txt
