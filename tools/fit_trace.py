"""Diagnostic (run on the GPU box): device-side phase times of every fit-kernel launch of ONE headline iteration.

    make -C pixray_amd/csrc trace && python tools/fit_trace.py [--config cfg1] [--precision fp16] [--out gpurun_out/fit_trace.csv]

Loads the diagnostic twin of the library (libprx_hip_trace.so: gemmfit.hip built with -DPRX_FIT_TRACE), in which every wave of
a fit kernel keeps s_memtime stamps of its phases in scalar registers and writes them out at its end.  Prints, per launch
shape, the phases in microseconds (s_memtime ticks, calibrated against the constant 100 MHz s_memrealtime clock over every
wave's lifetime) as median over workgroups of the per-workgroup LAST wave:
    setup   entry -> DMA coordinates ready          fill    -> first stage landed (first barrier)
    loop    -> K loop done                          ksum    -> K groups summed through LDS
    epi     -> epilogue issued (last store issued)  drain   -> stores acknowledged (s_waitcnt vmcnt(0))
and the workgroup's span (first entry of any of its waves -> its last acknowledgement; median over workgroups), i.e. the
kernel's device-side duration without the dispatch boundary and without the dispatch skew between workgroups (the s_memtime
counters of different XCDs are not synchronised)."""
import argparse
import collections
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
import numpy as np
import torch
from pixray_amd import _lib

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="cfg1")
ap.add_argument("--precision", default="fp16")
ap.add_argument("--out", default=None)
ap.add_argument("--iters", type=int, default=3)
args = ap.parse_args()

_lib.LIB_PATH = os.path.join(ROOT, "pixray_amd", "csrc", "libprx_hip_trace.so")
lib = _lib.load()
from pixray_amd import api

dev = torch.device("cuda", 0)
torch.manual_seed(1234)
sess = api.build_workload(args.config, precision=args.precision, device=dev)
it = 0
for _ in range(6):
    sess.train(it); it += 1
torch.cuda.synchronize()

buf = torch.zeros(64 << 20, dtype=torch.uint8, device=dev)
lib.prx_fit_trace_begin.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
lib.prx_fit_trace_begin.restype = None
lib.prx_fit_trace_end.argtypes = [ctypes.c_char_p, ctypes.c_size_t]
lib.prx_fit_trace_end.restype = ctypes.c_longlong

rows = collections.defaultdict(list)
ticks_per_us = []
REP = os.environ.get("PRX_FIT_TRACE_REP", "0") not in ("", "0")
if REP:
    print("PRX_FIT_TRACE_REP: the epilogue runs twice; column 'setup' = the SECOND (warm instruction cache) pass, 'epi' = the first, 'loop' = entry -> loop end")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for rep in range(args.iters):
    buf.zero_()
    torch.cuda.synchronize()
    lib.prx_fit_trace_begin(buf.data_ptr(), buf.numel())
    e0.record()
    sess.train(it); it += 1
    e1.record()
    torch.cuda.synchronize()
    text = ctypes.create_string_buffer(1 << 20)
    lib.prx_fit_trace_end(text, len(text))
    host = buf.cpu().numpy().view(np.uint64)
    step_ms = e0.elapsed_time(e1)
    for li, line in enumerate(text.value.decode().splitlines()):
        f = [int(x) for x in line.split()]
        off, grid = f[0], f[1]
        t = host[off: off + grid * 64].reshape(grid, 8, 8).astype(np.int64)      # [workgroup][wave][slot]
        st = t[:, :, :7]
        if (st[:, :, 0] == 0).any():
            continue
        real = (t[:, :, 7] >> 32).astype(np.float64)                          # wave lifetimes in 10 ns ticks
        life = (st[:, :, 6] - st[:, :, 0]).astype(np.float64)
        ticks_per_us.append(float(np.median(life[real > 50] / real[real > 50]) * 100.0) if (real > 50).any() else float("nan"))
        ph = np.diff(st, axis=2)                                              # [wg][wave][6 phases]
        last = st[:, :, 6].argmax(axis=1)                                     # the workgroup's last wave
        phl = ph[np.arange(grid), last]                                       # [wg][6]
        if REP:                                                               # slots 1, 2 hold the second epilogue pass's begin / end
            phl = phl.copy()
            phl[:, 0] = (st[:, :, 2] - st[:, :, 1])[np.arange(grid), last]    # "setup" column = the warm pass
            phl[:, 1] = 0; phl[:, 2] = st[np.arange(grid), last, 3] - st[np.arange(grid), last, 0]
        # the s_memtime counters of different XCDs are not synchronised: spans are taken per workgroup
        span = np.median(st[:, :, 6].max(axis=1) - st[:, :, 0].min(axis=1))
        entry_spread = np.median(st[:, :, 0].max(axis=1) - st[:, :, 0].min(axis=1))
        rows[tuple(f[2:])].append((li, np.median(phl, axis=0), phl.max(axis=0), span, entry_spread, step_ms))

tpu = float(np.nanmedian(ticks_per_us))
print(f"s_memtime ticks per microsecond (calibrated against s_memrealtime over every wave's lifetime): {tpu:.1f}")
TICK_US = 1.0 / tpu
hdr = "tile      M     N     K  mode act outs(f32,16) ops stats r16 | n | setup  fill   loop   ksum   epi  drain | span  entry-spread (us; median of per-WG last wave)"
print(hdr)
out_lines = ["bm,bn,M,N,K,a_mode,up,act,out_f32,out16,operands,stats,row16,flags,launches,setup,fill,loop,ksum,epi,drain,span,entry_spread"]
tot = 0.0
for key, recs in sorted(rows.items(), key=lambda kv: -sum(r[3] for r in kv[1])):
    bm, bn, M, N, K, amode, up, act, of32, o16, ops, stats, r16, flags = key
    med = np.median(np.stack([r[1] for r in recs]), axis=0) * TICK_US
    span = np.median([r[3] for r in recs]) * TICK_US
    spread = np.median([r[4] for r in recs]) * TICK_US
    n = len(recs) / args.iters
    tot += span * n
    print(f"{bm:3d}x{bm and bn:<3d} {M:5d} {N:5d} {K:5d}  {amode}{up}   {act}   {of32},{o16}        {ops}   {stats}    {r16}  |{n:5.1f}| " +
          " ".join(f"{v:6.2f}" for v in med) + f" | {span:6.2f} {spread:6.2f}")
    out_lines.append(",".join(str(v) for v in key) + f",{n:.1f}," + ",".join(f"{v:.3f}" for v in med) + f",{span:.3f},{spread:.3f}")
print(f"sum of spans per iteration: {tot:.1f} us; iteration (events, traced build): {np.median([r[5] for rs in rows.values() for r in rs]):.3f} ms")
if args.out:
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    with open(args.out, "w") as fh:
        fh.write("\n".join(out_lines) + "\n")
