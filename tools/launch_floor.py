"""Diagnostic (run on the GPU box): what one kernel boundary costs on a stream.

The headline iteration is 480 launches and its smallest kernels sit on a 4.2-5 us floor whatever they do
(profiles/r04_cfg1_kernel_stats.csv); replaying the iteration from a hipGraph did not move the step time, so the floor is inside
the device, not on the host.  This script times a chain of N dependent launches of a kernel that does (almost) nothing -- a
256-float row norm from the library's own C ABI -- back to back on torch's current stream, eagerly and replayed from a
hipGraph, and repeats the measurement in child processes under the HIP / ROCr dispatch knobs that can be set from the
environment.  us per launch = wall time of the chain / N (HIP events around the chain, device idle before it).

    python tools/launch_floor.py            # parent: runs every variant in a child process and prints a table
"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

VARIANTS = [
    ("default", {}),
    ("HIP_FORCE_DEV_KERNARG=1", {"HIP_FORCE_DEV_KERNARG": "1"}),
    ("HIP_FORCE_DEV_KERNARG=0", {"HIP_FORCE_DEV_KERNARG": "0"}),
    ("AMD_DIRECT_DISPATCH=0", {"AMD_DIRECT_DISPATCH": "0"}),
    ("HSA_ENABLE_INTERRUPT=0", {"HSA_ENABLE_INTERRUPT": "0"}),
    ("GPU_MAX_HW_QUEUES=1", {"GPU_MAX_HW_QUEUES": "1"}),
    ("HSA_OVERRIDE_CPU_AFFINITY_DEBUG=0", {"HSA_OVERRIDE_CPU_AFFINITY_DEBUG": "0"}),
    ("ROC_ACTIVE_WAIT_TIMEOUT=1000", {"ROC_ACTIVE_WAIT_TIMEOUT": "1000"}),
]


def child():
    import torch
    from pixray_amd import _lib
    from pixray_amd._lib import call
    _lib.load()
    dev = "cuda"
    x = torch.randn(4, 256, device=dev)
    out = torch.empty(4, device=dev)
    big = torch.randn(3200, 768, device=dev)
    obig = torch.empty(3200, device=dev)
    s = _lib.current_stream()

    def chain(n, t, o, rows):
        for _ in range(n):
            call("prx_k_sqnorm_rows", t, o, rows, t.shape[1], s)

    def timed(fn, reps=5):
        best = 1e9
        for _ in range(reps):
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); fn(); b.record()
            torch.cuda.synchronize()
            best = min(best, a.elapsed_time(b))
        return best

    n = 2000
    chain(50, x, out, 4)
    eager_small = timed(lambda: chain(n, x, out, 4)) * 1e3 / n
    eager_big = timed(lambda: chain(n, big, obig, 3200)) * 1e3 / n
    t0 = time.perf_counter(); chain(n, x, out, 4); host = (time.perf_counter() - t0) * 1e6 / n
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    try:
        with torch.cuda.stream(side):
            s2 = _lib.current_stream()
            for _ in range(3):
                call("prx_k_sqnorm_rows", x, out, 4, 256, s2)
            torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=side):
                s3 = _lib.current_stream()
                for _ in range(n):
                    call("prx_k_sqnorm_rows", x, out, 4, 256, s3)
        graph_small = timed(g.replay) * 1e3 / n
    except Exception as e:                      # noqa: BLE001
        graph_small = float("nan")
        print("graph capture failed:", e, file=sys.stderr)
    print(f"RESULT eager 4x256 {eager_small:6.2f} us/launch | eager 3200x768 (9.8 MB read) {eager_big:6.2f} | host enqueue {host:6.2f} | "
          f"hipGraph replay 4x256 {graph_small:6.2f}")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child()
        sys.exit(0)
    for name, env in VARIANTS:
        e = dict(os.environ)
        e.update(env)
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=e, capture_output=True, text=True, timeout=120)
            line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")]
            print(f"{name:36s} {line[0][7:] if line else 'failed: ' + r.stderr.strip().splitlines()[-1][:120] if r.stderr.strip() else 'no output'}", flush=True)
        except subprocess.TimeoutExpired:
            print(f"{name:36s} timed out", flush=True)
