cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r02o_smoke.log 2>&1; echo "smoke rc=$?" > gpurun_out/r02o_rc.txt
timeout 300 python -m pytest tests/test_path_gpu.py -q -m gpu -k "test_clip_resnet_vs_oracle" > gpurun_out/r02o_rn.log 2>&1; echo "rn rc=$?" >> gpurun_out/r02o_rc.txt
PRX_FORCE_DIST=1 timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r02o_bench_dist.log 2>&1; echo "dist rc=$?" >> gpurun_out/r02o_rc.txt
timeout 300 python -m pytest tests/test_f32_mode_gpu.py tests/test_e2e_gpu.py -x -q -m gpu -k "headline or smoke or rccl" > gpurun_out/r02o_e2e.log 2>&1; echo "e2e rc=$?" >> gpurun_out/r02o_rc.txt
echo done >> gpurun_out/r02o_rc.txt
