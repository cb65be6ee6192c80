cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for i in 1 2 3; do timeout 300 python -m pytest tests/test_path_gpu.py -x -q -m gpu -s -k "test_clip_resnet_vs_oracle" 2>&1 | grep -E "emb rel|passed|failed" ; done > gpurun_out/r02k_rn.log 2>&1
timeout 400 python -m pytest tests/test_f32_mode_gpu.py -x -q -m gpu -s -k "resnet or rn" > gpurun_out/r02k_rn_f32.log 2>&1
echo done > gpurun_out/r02k_rc.txt
