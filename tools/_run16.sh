cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
PRX_CUSTOM_BACKWARD_LAST=0 timeout 400 python bench.py --config cfg3 --no-cpu-baseline --phase-steps 0 > gpurun_out/r02p_cfg3_one_pass.log 2>&1
timeout 400 python bench.py --config cfg3 --no-cpu-baseline --phase-steps 0 > gpurun_out/r02p_cfg3_split.log 2>&1
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r02p_gpu_suite.log 2>&1; echo "suite rc=$?" > gpurun_out/r02p_rc.txt
echo done >> gpurun_out/r02p_rc.txt
