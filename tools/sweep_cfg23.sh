#!/bin/bash
# One gpurun call: in-pipeline per-shape tile sweeps of configs[2] and configs[3] (tools/gemm_rules.py), i.e. the two measurements
# DESIGN.md section 6 names as the first ones of the next round: RN50x4's HBM-bound 1x1 convolutions on the smaller tiles, and the
# 8-phase kernel against the 4-wave kernels at K = 768 / 1024.  Plus the stand-alone cost of each fused
# epilogue per kernel family (tools/gemm_epilogue_cost.py).  ~5 GPU-minutes.
#   /usr/local/graft/bin/gpurun --timeout 420 -- 'bash tools/sweep_cfg23.sh'
set -u
mkdir -p gpurun_out
timeout 140 python tools/gemm_rules.py 16 3 cfg2 > gpurun_out/rules_cfg2.log 2>&1; echo "cfg2 rc=$?"
timeout 120 python tools/gemm_rules.py 6 3 cfg3 > gpurun_out/rules_cfg3.log 2>&1; echo "cfg3 rc=$?"
timeout 90 python tools/gemm_epilogue_cost.py fp16 > gpurun_out/epilogue_cost.log 2>&1; echo "epilogue rc=$?"
grep -A40 "beats the heuristic" gpurun_out/rules_cfg2.log | head -40
grep -A20 "beats the heuristic" gpurun_out/rules_cfg3.log | head -20
cut -c1-400 gpurun_out/epilogue_cost.log
