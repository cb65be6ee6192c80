#!/bin/bash
# kernel timeline of one steady-state headline iteration (+ env knobs A/B of the launch path)
#   gpurun --timeout 600 -- 'bash tools/r05_timeline.sh'
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
export TMPDIR=/tmp
cd /tmp
B="--steps 40 --warmup 8 --no-other-modes --no-cpu-baseline --profile-steps 0 --phase-steps 0"
for tag in default; do
  timeout 200 rocprofv3 --kernel-trace -d /tmp/prof_tl_$tag -- python $R/bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-other-modes --profile-steps 0 --phase-steps 0 > $R/gpurun_out/r05_tl_$tag.log 2>&1; echo "trace $tag rc=$?"
  db=$(find /tmp/prof_tl_$tag -name "*.db" | head -1)
  [ -n "$db" ] && python $R/tools/kernel_timeline.py "$db" $R/gpurun_out/r05_cfg1_timeline_$tag.csv
done
cd $R
line() { grep '^{' $1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$2', d['value'], d['ms_per_step'])"; }
python bench.py $B > gpurun_out/r05_knob_base.json 2>/dev/null; line gpurun_out/r05_knob_base.json base
HIP_FORCE_DEV_KERNARG=0 python bench.py $B > gpurun_out/r05_knob_kernarg0.json 2>/dev/null; line gpurun_out/r05_knob_kernarg0.json kernarg0
AMD_DIRECT_DISPATCH=0 python bench.py $B > gpurun_out/r05_knob_dd0.json 2>/dev/null; line gpurun_out/r05_knob_dd0.json direct_dispatch0
python bench.py $B --graph > gpurun_out/r05_knob_graph.json 2>/dev/null; line gpurun_out/r05_knob_graph.json graph
HIP_FORCE_DEV_KERNARG=0 python bench.py $B --graph > gpurun_out/r05_knob_graph_k0.json 2>/dev/null; line gpurun_out/r05_knob_graph_k0.json graph_kernarg0
PRX_VIT_CLS_TAIL=1 python bench.py $B > gpurun_out/r05_knob_cls.json 2>/dev/null; line gpurun_out/r05_knob_cls.json cls
PRX_VIT_CLS_TAIL=1 PRX_LIB_PATH=libprx_hip_sched.so python bench.py $B > gpurun_out/r05_knob_cls_sched.json 2>/dev/null; line gpurun_out/r05_knob_cls_sched.json cls_sched
python bench.py $B > gpurun_out/r05_knob_base2.json 2>/dev/null; line gpurun_out/r05_knob_base2.json base_again
