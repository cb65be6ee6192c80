cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 400 python bench.py --config cfg3 --no-cpu-baseline --phase-steps 0 > gpurun_out/r02j_cfg3_base.log 2>&1
PRX_BIG_TILE=2 timeout 400 python bench.py --config cfg3 --no-cpu-baseline --phase-steps 0 > gpurun_out/r02j_cfg3_big2.log 2>&1
PRX_BIG_TILE=4 timeout 400 python bench.py --config cfg3 --no-cpu-baseline --phase-steps 0 > gpurun_out/r02j_cfg3_big4.log 2>&1
timeout 300 python bench.py --config cfg2 --no-cpu-baseline --phase-steps 0 > gpurun_out/r02j_cfg2_base.log 2>&1
PRX_BIG_TILE=2 timeout 300 python bench.py --config cfg2 --no-cpu-baseline --phase-steps 0 > gpurun_out/r02j_cfg2_big2.log 2>&1
PRX_BIG_TILE=2 timeout 300 python bench.py --no-cpu-baseline --phase-steps 0 > gpurun_out/r02j_cfg1_big2.log 2>&1
timeout 600 python -m pytest tests/test_e2e_gpu.py tests/test_path_gpu.py -x -q -m gpu -k "resnet or rn50 or config2" > gpurun_out/r02j_rn.log 2>&1; echo "rn rc=$?" > gpurun_out/r02j_rc.txt
echo done >> gpurun_out/r02j_rc.txt
