"""In-pipeline tile / split-K sweep per GEMM shape (run on the GPU box).

For each of the heaviest problem shapes of the headline iteration, try every tile shape x a few split-K factors through
`prx_gemm_tile_rule`, time that shape's launches with the engine's HIP events while the rest of the iteration runs
unchanged (cold weights, real neighbours), and print what beats the heuristic by more than the noise.

    python tools/gemm_rules.py [top shapes] [steps] [cfg1|cfg2|cfg3] [cutn]

Candidates include 256x256 (the 8-phase kernel on the whole problem, no row peeling) wherever it is eligible, and the 4-wave
tiles for shapes the planner gives to the 8-phase kernel -- a rule replaces the plan for its own shape only."""
import collections
import ctypes
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
top = int(sys.argv[1]) if len(sys.argv) > 1 else 14
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
config = sys.argv[3] if len(sys.argv) > 3 else "cfg1"
cutn = int(sys.argv[4]) if len(sys.argv) > 4 else None
path = os.path.join(tempfile.gettempdir(), "prx_gemm_rules.csv")
os.environ["PRX_GEMM_PROFILE_DUMP"] = path
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
lib = _lib.load()
prof = api.GemmProfile(sess)
it = [0]


def run(n):
    for _ in range(n):
        sess.train(it[0]); it[0] += 1
    torch.cuda.synchronize()


def measure():
    """{(M,N,K,mode): (launches/iter, us per launch, (bm,bn,splits))}, total ms/iter"""
    if os.path.exists(path):
        os.remove(path)
    prof.enable(True)
    run(steps)
    prof.enable(False)
    _ms, _fl, _n = prof.collect()
    ms = ctypes.c_double(_ms)
    agg = collections.OrderedDict()
    for line in open(path):
        M, N, K, mode, bm, bn, sp, us = line.strip().split(",")
        a = agg.setdefault((int(M), int(N), int(K), int(mode)), [0, 0.0, None])
        a[0] += 1; a[1] += float(us); a[2] = (int(bm), int(bn), int(sp))
    return {k: (v[0] / steps, v[1] / v[0], v[2]) for k, v in agg.items()}, ms.value / steps


run(4)
base, base_ms = measure()
base2, base_ms2 = measure()
print(f"baseline GEMM engine {base_ms:.3f} / {base_ms2:.3f} ms per iteration (two passes = the noise)")
order = sorted(base, key=lambda k: -base[k][0] * base[k][1])[:top]
wins = []
for key in order:
    M, N, K, mode = key
    cnt, us0, cfg0 = base[key]
    us0b = base2[key][1]
    kt = (K + 63) // 64
    if cfg0[0] >= 1000:          # profile records code a fit tile (gemmfit.hip) as bm + 1000: the fit planner owns this shape -- nothing to sweep here
        print(f"{M:6d} {N:5d} {K:5d} m{mode} x{cnt:4.1f}  fit tile {cfg0[0] - 1000}x{cfg0[1]}: {us0:6.1f} / {us0b:6.1f} us | planner-owned, not swept")
        continue
    row = f"{M:6d} {N:5d} {K:5d} m{mode} x{cnt:4.1f}  heuristic {cfg0[0]}x{cfg0[1]} s{cfg0[2]}: {us0:6.1f} / {us0b:6.1f} us |"
    best = (min(us0, us0b), cfg0)
    for bm, bn in ((256, 256), (128, 128), (128, 64), (64, 64)):
        if N <= 64 and bn > 64:
            continue
        if bm == 256 and (mode != 0 or K % 128 or M < 2048 or N < 256):      # the 8-phase kernel: row-major 16-bit operands, K % 128 == 0
            continue
        tiles = -(-M // bm) * -(-N // bn)
        cands = {1}
        if tiles <= 256:
            for tgt in (192, 256, 320, 448, 640):
                s = max(1, min(round(tgt / tiles), kt // 2, 32))
                cands.add(s)
        for sp in sorted(cands):
            if (bm, bn, sp) == cfg0:
                continue
            prof.tile_rule(M, N, K, mode, bm, bn, sp)
            try:
                r, _ = measure()
            except Exception as e:                      # a shape a tile cannot take (fused-stat constraints)
                print("   skip", key, bm, bn, sp, str(e)[:80])
                prof.tile_rule(M, N, K, mode, 0, 0, 0)
                continue
            prof.tile_rule(M, N, K, mode, 0, 0, 0)
            us = r[key][1]
            row += f" {bm}x{bn}s{sp}:{us:6.1f}"
            if us < best[0]:
                best = (us, (bm, bn, sp))
    print(row)
    if best[1] != cfg0 and best[0] < 0.96 * min(us0, us0b):
        wins.append((key, cfg0, best[1], us0, best[0], cnt))
print("\nbeats the heuristic by > 4 %:")
for key, c0, c1, u0, u1, cnt in wins:
    print(f"  {key}: {c0} {u0:.1f} us -> {c1} {u1:.1f} us  (x{cnt:.0f}/iter = {(u0 - u1) * cnt / 1e3:.3f} ms)")
# confirm the winners together
for key, c0, c1, *_ in wins:
    prof.tile_rule(*key, *c1)
if wins:
    _, ms_all = measure()
    print(f"all winners applied: GEMM engine {ms_all:.3f} ms per iteration (baseline {base_ms:.3f} / {base_ms2:.3f})")
