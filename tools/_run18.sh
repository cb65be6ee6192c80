cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 200 python tools/lib_gemm_compare.py > gpurun_out/r02r_libgemm.log 2>&1
timeout 300 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm" > gpurun_out/r02r_tests.log 2>&1; echo "tests rc=$?" > gpurun_out/r02r_rc.txt
timeout 200 python bench.py --no-cpu-baseline --phase-steps 0 > gpurun_out/r02r_cfg1.log 2>&1
timeout 200 python bench.py --no-cpu-baseline --phase-steps 0 --config cfg2 > gpurun_out/r02r_cfg2.log 2>&1
echo done >> gpurun_out/r02r_rc.txt
