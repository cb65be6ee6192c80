cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "mha" > gpurun_out/r02g_mha_tests.log 2>&1; echo "mha rc=$?" > gpurun_out/r02g_rc.txt
timeout 200 python tools/mha_bench.py > gpurun_out/r02g_mha_bench.log 2>&1
PRX_MHA_TILES=1 timeout 200 python tools/mha_bench.py > gpurun_out/r02g_mha_bench_tiles.log 2>&1
timeout 900 python -m pytest tests/test_path_gpu.py tests/test_determinism_gpu.py -x -q -m gpu > gpurun_out/r02g_tests.log 2>&1; echo "path rc=$?" >> gpurun_out/r02g_rc.txt
timeout 300 python tools/cutout_bench.py > gpurun_out/r02g_cutbench.log 2>&1
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r02g_bench_cfg1.log 2>&1
timeout 300 python bench.py --config cfg2 --no-cpu-baseline > gpurun_out/r02g_bench_cfg2.log 2>&1
timeout 400 python bench.py --config cfg3 --no-cpu-baseline > gpurun_out/r02g_bench_cfg3.log 2>&1
timeout 400 bash tools/profile_run.sh r02g_cfg1 stats 35 --steps 30 --warmup 5
timeout 400 bash tools/profile_run.sh r02g_cfg2 stats 8 --config cfg2 --steps 6 --warmup 2
timeout 400 bash tools/profile_run.sh r02g_cfg3 stats 5 --config cfg3 --steps 4 --warmup 1
timeout 600 python -m pytest tests/test_e2e_gpu.py -x -q -m gpu -k "cfg2 or cfg3 or ensemble or resnet or b16 or l14" > gpurun_out/r02g_e2e.log 2>&1; echo "e2e rc=$?" >> gpurun_out/r02g_rc.txt
echo done >> gpurun_out/r02g_rc.txt
