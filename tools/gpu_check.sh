#!/bin/bash
# One gpurun call of the round's working loop: the tests next to what changed, the headline bench, and (optionally) a kernel
# timeline / kernel statistics of it.   gpurun --timeout 900 -- 'bash tools/gpu_check.sh <tag> "<pytest -k expr>" [timeline] [ref]'
# Everything lands in gpurun_out/<tag>_*; copy what should be kept into profiles/.
set -u
AB_ENV=${AB_ENV:-}
tag=$1; kexpr=${2:-}; shift 2
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
export TMPDIR=/tmp
cd $R
if [ -n "$kexpr" ]; then
  timeout 600 python -m pytest tests -m gpu -x -q -k "$kexpr" > gpurun_out/${tag}_pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/${tag}_pytest.log | cut -c1-300
fi
B="--steps 40 --warmup 8 --no-other-modes --no-cpu-baseline"
line() { grep '^{' $1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; print('$2', d['value'], 'it/s', d['ms_per_step'], 'ms; engine', r.get('gemm_ms_per_step'), 'ms frac', r.get('frac'), 'launches', r.get('launches_per_step'))"; }
timeout 200 python bench.py $B > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; line gpurun_out/${tag}_bench.json fp16
for what in "$@"; do
  case $what in
  timeline)
    cd /tmp
    timeout 200 rocprofv3 --kernel-trace -d /tmp/prof_tl_$tag -- python $R/bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-other-modes --profile-steps 0 --phase-steps 0 > $R/gpurun_out/${tag}_tl.log 2>&1
    db=$(find /tmp/prof_tl_$tag -name "*.db" | head -1)
    [ -n "$db" ] && python $R/tools/kernel_timeline.py "$db" $R/gpurun_out/${tag}_timeline.csv | tail -8
    cd $R ;;
  ref)
    timeout 200 python bench.py $B --precision ref > gpurun_out/${tag}_bench_ref.json 2> gpurun_out/${tag}_bench_ref.err; line gpurun_out/${tag}_bench_ref.json ref ;;
  reftimeline)
    cd /tmp
    timeout 200 rocprofv3 --kernel-trace -d /tmp/prof_tlr_$tag -- python $R/bench.py --steps 8 --warmup 3 --precision ref --no-cpu-baseline --no-other-modes --profile-steps 0 --phase-steps 0 > $R/gpurun_out/${tag}_tlr.log 2>&1
    db=$(find /tmp/prof_tlr_$tag -name "*.db" | head -1)
    [ -n "$db" ] && python $R/tools/kernel_timeline.py "$db" $R/gpurun_out/${tag}_timeline_ref.csv | tail -8
    cd $R ;;
  ab)        # same-box A/B against the snapshot of the last accepted tree (tools/ab_snapshot.sh): base, new, base, new
    for i in 1 2; do
      timeout 200 python .ab_base/bench.py $B --profile-steps 0 --phase-steps 0 > gpurun_out/${tag}_ab_base$i.json 2>/dev/null; line gpurun_out/${tag}_ab_base$i.json "base$i"
      timeout 200 env $AB_ENV python bench.py $B --profile-steps 0 --phase-steps 0 > gpurun_out/${tag}_ab_new$i.json 2>/dev/null; line gpurun_out/${tag}_ab_new$i.json "new$i [$AB_ENV]"
    done ;;
  smoke)
    timeout 300 python __graft_entry__.py smoke > gpurun_out/${tag}_smoke.log 2>&1; echo "smoke rc=$?"; grep "smoke" gpurun_out/${tag}_smoke.log | cut -c1-400 ;;
  esac
done
