"""Per-shape table of the GEMM-engine launches of one optimisation iteration (run on the GPU box).

    python tools/gemm_shapes.py [steps] [cfg1|cfg2|cfg3] [cutn]

Enables the engine's HIP-event profiling for a few iterations of the headline session, dumps one row per launch
(PRX_GEMM_PROFILE_DUMP) and aggregates by (M, N, K, A mode, tile, split-K)."""
import collections
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
config = sys.argv[2] if len(sys.argv) > 2 else "cfg1"
cutn = int(sys.argv[3]) if len(sys.argv) > 3 else None
path = os.path.join(tempfile.gettempdir(), "prx_gemm_dump.csv")
if os.path.exists(path):
    os.remove(path)
os.environ["PRX_GEMM_PROFILE_DUMP"] = path
import ctypes
import torch
from pixray_amd import _lib, api

dev = torch.device("cuda", 0)
custom, largs = (), None
if config == "cfg3":
    import numpy as np
    import bench
    np.random.seed(0)
    custom, largs = bench.cfg3_custom_losses(dev, None)
sess = api.build_workload(config, num_cuts=cutn, device=dev, custom_losses=custom, args=largs)
for i in range(3):
    sess.train(i)
torch.cuda.synchronize()
prof = api.GemmProfile(sess)
prof.enable(True)
for i in range(steps):
    sess.train(3 + i)
torch.cuda.synchronize()
prof.enable(False)
_ms, _fl, _n = prof.collect()
ms, fl, n = ctypes.c_double(_ms), ctypes.c_double(_fl), ctypes.c_longlong(_n)
agg = collections.OrderedDict()
for line in open(path):
    M, N, K, mode, bm, bn, sp, us = line.strip().split(",")
    key = (int(M), int(N), int(K), int(mode), int(bm), int(bn), int(sp))
    a = agg.setdefault(key, [0, 0.0])
    a[0] += 1; a[1] += float(us)
print(f"total {ms.value / steps:.3f} ms/iter, {fl.value / steps / 1e9:.0f} GFLOP/iter, {n.value // steps} launches/iter, "
      f"{fl.value / ms.value / 1e9:.0f} TFLOP/s")
print(f"{'M':>6} {'N':>5} {'K':>5} mode tile       sp  n/it  avg_us  ms/it   TF     (fit: gemmfit.hip, one workgroup per CU)")
for key, (cnt, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    M, N, K, mode, bm, bn, sp = key
    avg = us / cnt
    tile = f"row{bm - 2000}x{bn}" if bm >= 2000 else (f"fit{bm - 1000}x{bn}" if bm >= 1000 else f"{bm}x{bn}")
    print(f"{M:6d} {N:5d} {K:5d} {mode:4d} {tile:<10s} {sp:3d} {cnt / steps:5.1f} {avg:7.1f} {us / steps / 1e3:6.3f} {2.0 * M * N * K / avg / 1e6:5.0f}")
