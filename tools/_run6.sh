cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_path_gpu.py tests/test_determinism_gpu.py -x -q -m gpu > gpurun_out/r02f_tests.log 2>&1; echo "tests rc=$?" > gpurun_out/r02f_rc.txt
timeout 300 python tools/cutout_bench.py > gpurun_out/r02f_cutbench.log 2>&1
timeout 300 python -m pytest tests/test_e2e_gpu.py -x -q -m gpu -k "style or cfg3" > gpurun_out/r02f_style_tests.log 2>&1; echo "style rc=$?" >> gpurun_out/r02f_rc.txt
timeout 400 python bench.py --config cfg3 --no-cpu-baseline > gpurun_out/r02f_bench_cfg3.log 2>&1
timeout 300 python bench.py --config cfg3 --cutn 32 --no-cpu-baseline > gpurun_out/r02f_bench_cfg3_c32.log 2>&1
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r02f_bench_cfg1.log 2>&1
timeout 300 python bench.py --config cfg2 --no-cpu-baseline > gpurun_out/r02f_bench_cfg2.log 2>&1
timeout 300 python tools/lib_gemm_compare.py > gpurun_out/r02f_libgemm.log 2>&1
timeout 400 bash tools/profile_run.sh r02f_cfg1 stats 35 --steps 30 --warmup 5
timeout 400 bash tools/profile_run.sh r02f_cfg3 stats 5 --config cfg3 --steps 4 --warmup 1
echo done >> gpurun_out/r02f_rc.txt
