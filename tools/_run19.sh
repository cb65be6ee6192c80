cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "mha" > gpurun_out/r02s_mha_tests.log 2>&1; echo "mha rc=$?" > gpurun_out/r02s_rc.txt
timeout 200 python tools/mha_bench.py > gpurun_out/r02s_mha_bench.log 2>&1
timeout 600 python -m pytest tests/test_e2e_gpu.py tests/test_path_gpu.py -x -q -m gpu -k "config2 or config3 or b16 or l14 or vit or resnet" > gpurun_out/r02s_e2e.log 2>&1; echo "e2e rc=$?" >> gpurun_out/r02s_rc.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r02s_smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/r02s_rc.txt
echo done >> gpurun_out/r02s_rc.txt
