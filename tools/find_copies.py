"""Which host-side operation issues the small device-to-device copies / fills of one iteration (run on the GPU box).

rocprofv3 shows ~34 `__amd_rocclr_copyBuffer` + ~9 `fillBuffer` launches per iteration; this attributes each memcpy /
memset the torch profiler sees to the innermost CPU-side range (autograd node, record_function range or aten op) whose
time span contains the runtime call that issued it."""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import ProfilerActivity, profile, record_function

from pixray_amd import api

dev = torch.device("cuda", 0)
sess = api.build_vqgan_clip_session(size=(256, 256), vqgan_model="imagenet_f16_16384", clip_model="ViT-B/32", num_cuts=64,
                                    learning_rate=0.2, iterations=10 ** 9, seed=0, device=dev)
for i in range(3):
    sess.train(i)
torch.cuda.synchronize()


def one_iteration(it):
    sess.cur_iteration = it
    for opt in sess.opts:
        opt.zero_grad(set_to_none=True)
    with record_function("PRX host_prep"):
        sess._host_prep(it)
    with record_function("PRX ascend_txt"):
        losses = sess.ascend_txt()
    with record_function("PRX sum"):
        loss = sum(losses)
    with record_function("PRX backward"):
        loss.backward()
    with record_function("PRX opt.step"):
        for opt in sess.opts:
            opt.step()
    with record_function("PRX clip_z"):
        sess._clip_z()


N = 4
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for k in range(N):
        one_iteration(3 + k)
    torch.cuda.synchronize()

evs = prof.events()
cpu = [e for e in evs if e.device_type == torch.autograd.DeviceType.CPU]
rt = [e for e in cpu if "emcpy" in e.name or "emset" in e.name]          # hipMemcpyAsync / hipMemsetAsync runtime calls
ranges = [e for e in cpu if e not in rt and (e.time_range.end - e.time_range.start) > 0]
count = collections.Counter()
for r in rt:
    t = r.time_range.start
    best = None
    for e in ranges:
        if e.time_range.start <= t <= e.time_range.end:
            if best is None or (e.time_range.end - e.time_range.start) < (best.time_range.end - best.time_range.start):
                best = e
    outer = [e.name for e in ranges if e.name.startswith("PRX") and e.time_range.start <= t <= e.time_range.end]
    count[(r.name, best.name if best else "?", outer[0] if outer else "?")] += 1
print(f"runtime memcpy/memset calls per iteration (over {N} iterations):")
for (name, inner, outer), c in sorted(count.items(), key=lambda kv: -kv[1]):
    print(f"  {c / N:6.2f}  {name:24s} in {inner[:60]:60s} [{outer}]")
gpu = collections.Counter(e.name for e in evs if e.device_type != torch.autograd.DeviceType.CPU and ("emcpy" in e.name or "emset" in e.name))
print("device-side memcpy/memset events per iteration:", {k: v / N for k, v in gpu.items()})
