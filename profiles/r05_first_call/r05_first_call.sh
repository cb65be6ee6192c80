#!/bin/bash
# Round 5, FIRST gpurun call (~15 GPU-minutes): everything that was prepared on the CPU emulation in round 4 (no GPU minutes
# were left) gets its first run on the device, shortest and most hang-prone first, each step under its own timeout.
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/r05_first_call.sh'
# 1. producer-wave fit kernels (gemmfit.hip NPROD = 4): the barrier protocol relies on ended waves not being waited for.
# 2. the same build's headline bench: baseline, PRX_FIT_FLAGS=65 (producer waves on every tile that has the variant),
#    PRX_VIT_CLS_TAIL=1 (the last ViT block's projection / MLP on the class-token rows only), PRX_VIT_LANES=2 / 4 (the tower as
#    concurrent chunk chains on streams), the alternative build with the fit kernels' stage schedule pinned (libprx_hip_sched.so), and combinations.
# 3. stand-alone per-shape timings of the two (tools/fit_bench.py: `producer waves` column).
# 4. the front end on the HIP parts and the full GPU suite.
set -u
mkdir -p gpurun_out
export PRX_TEST_EXPERIMENTAL=1
timeout 150 python -m pytest tests/test_kernels_gpu.py tests/test_zz_frontend_gpu.py -q -k "producer_wave or fft_drawer_hip or tower_lanes" > gpurun_out/r05_producer_tests.log 2>&1; echo "producer tests rc=$?"
tail -3 gpurun_out/r05_producer_tests.log
timeout 240 python -m pytest tests/test_kernels_gpu.py -q -k "random_shapes_on_the_device" > gpurun_out/r05_sweeps.log 2>&1; echo "random-shape sweeps on the device rc=$?"
tail -25 gpurun_out/r05_sweeps.log | cut -c1-300
timeout 200 python bench.py --steps 40 --warmup 8 --no-other-modes --no-cpu-baseline > gpurun_out/r05_bench_base.json 2> gpurun_out/r05_bench_base.err; echo "bench base rc=$?"
PRX_FIT_FLAGS=65 timeout 200 python bench.py --steps 40 --warmup 8 --no-other-modes --no-cpu-baseline > gpurun_out/r05_bench_prod.json 2> gpurun_out/r05_bench_prod.err; echo "bench producers rc=$?"
PRX_VIT_CLS_TAIL=1 timeout 200 python bench.py --steps 40 --warmup 8 --no-other-modes --no-cpu-baseline > gpurun_out/r05_bench_cls.json 2> gpurun_out/r05_bench_cls.err; echo "bench class-token tail rc=$?"
PRX_VIT_LANES=2 timeout 200 python bench.py --steps 40 --warmup 8 --no-other-modes --no-cpu-baseline > gpurun_out/r05_bench_lanes2.json 2> gpurun_out/r05_bench_lanes2.err; echo "bench 2 lanes rc=$?"
PRX_VIT_LANES=4 timeout 200 python bench.py --steps 40 --warmup 8 --no-other-modes --no-cpu-baseline > gpurun_out/r05_bench_lanes4.json 2> gpurun_out/r05_bench_lanes4.err; echo "bench 4 lanes rc=$?"
PRX_LIB_PATH=libprx_hip_sched.so timeout 200 python bench.py --steps 40 --warmup 8 --no-other-modes --no-cpu-baseline > gpurun_out/r05_bench_sched.json 2> gpurun_out/r05_bench_sched.err; echo "bench pinned schedule rc=$?"
PRX_LIB_PATH=libprx_hip_sched.so PRX_FIT_FLAGS=65 timeout 200 python bench.py --steps 40 --warmup 8 --no-other-modes --no-cpu-baseline > gpurun_out/r05_bench_schedprod.json 2> gpurun_out/r05_bench_schedprod.err; echo "bench pinned schedule + producers rc=$?"
PRX_VIT_CLS_TAIL=1 PRX_FIT_FLAGS=65 timeout 200 python bench.py --steps 40 --warmup 8 --no-other-modes --no-cpu-baseline > gpurun_out/r05_bench_both.json 2> gpurun_out/r05_bench_both.err; echo "bench both rc=$?"
python - <<'PY'
import json
for tag in ("base", "prod", "cls", "lanes2", "lanes4", "sched", "schedprod", "both"):
    try:
        line = [l for l in open(f"gpurun_out/r05_bench_{tag}.json") if l.startswith("{")][-1]
        d = json.loads(line)
        print(tag, d["value"], d["unit"], "ms/step", d["ms_per_step"], "roofline", d.get("roofline", {}).get("frac"))
    except Exception as e:
        print(tag, "no line:", e)
PY
timeout 150 python tools/fit_bench.py > gpurun_out/r05_fit_bench.log 2>&1; echo "fit_bench rc=$?"
PRX_LIB_PATH=libprx_hip_sched.so timeout 150 python tools/fit_bench.py > gpurun_out/r05_fit_bench_sched.log 2>&1; echo "fit_bench (pinned schedule) rc=$?"
cut -c1-60,150-400 gpurun_out/r05_fit_bench_sched.log | head -12
cut -c1-60,150-400 gpurun_out/r05_fit_bench.log | head -20
timeout 150 python tools/fit_conv_bench.py > gpurun_out/r05_fit_conv_bench.log 2>&1; echo "fit_conv_bench rc=$?"
cut -c1-400 gpurun_out/r05_fit_conv_bench.log | head -12
unset PRX_TEST_EXPERIMENTAL
timeout 120 python -m pytest tests/test_zz_frontend_gpu.py -x -q > gpurun_out/r05_frontend.log 2>&1; echo "frontend rc=$?"; tail -2 gpurun_out/r05_frontend.log
