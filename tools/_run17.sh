cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export PRX_BIG_TILE=2
PRX_BIG_TILE_WAVES=8 timeout 200 python tools/lib_gemm_compare.py > gpurun_out/r02q_w8_libgemm.log 2>&1
PRX_BIG_TILE_WAVES=4 timeout 200 python tools/lib_gemm_compare.py > gpurun_out/r02q_w4_libgemm.log 2>&1
PRX_BIG_TILE_WAVES=4 timeout 200 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm" > gpurun_out/r02q_w4_tests.log 2>&1; echo "w4 tests rc=$?" > gpurun_out/r02q_rc.txt
PRX_BIG_TILE_WAVES=4 timeout 400 python bench.py --config cfg3 --no-cpu-baseline --phase-steps 0 > gpurun_out/r02q_w4_cfg3.log 2>&1
echo done >> gpurun_out/r02q_rc.txt
