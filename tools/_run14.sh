cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_path_gpu.py -q -m gpu -s -k "test_clip_resnet_vs_oracle" 2>&1 | grep -E "emb rel|passed|failed|^E  " > gpurun_out/r02n_rn.log 2>&1
for v in base il w83 w84; do
  case $v in base) export PRX_GEMM_INTERLEAVE=0 PRX_GEMM_W8=0;; il) export PRX_GEMM_INTERLEAVE=1 PRX_GEMM_W8=0;; w83) export PRX_GEMM_INTERLEAVE=0 PRX_GEMM_W8=3;; w84) export PRX_GEMM_INTERLEAVE=0 PRX_GEMM_W8=4;; esac
  if [ $v != base ]; then timeout 200 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm" > gpurun_out/r02n_${v}_tests.log 2>&1; echo "$v tests rc=$?" >> gpurun_out/r02n_rc.txt; fi
  timeout 200 python tools/lib_gemm_compare.py > gpurun_out/r02n_${v}_libgemm.log 2>&1
  timeout 200 python bench.py --no-cpu-baseline --phase-steps 0 > gpurun_out/r02n_${v}_cfg1.log 2>&1
done
echo done >> gpurun_out/r02n_rc.txt
