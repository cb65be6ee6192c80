#!/usr/bin/env python
"""bench.py -- optimisation iterations/sec of the pixray hot path on MI355X.

One "step" = one body of the reference's train() (pixray.py:1448-1487, batches=1): VqganDrawer.synth ->
MakeCutouts (host-drawn augmentation parameters, device noise) -> CLIP ViT encode_image -> Prompt loss ->
backward to z -> (N>1: all-reduce of dL/d(image)) -> Adam -> clip_z, at BASELINE.json configs[1]:
VQGAN imagenet_f16_16384 256x256 + ViT-B/32 + 64 cutouts, seeded random weights of the real architectures.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (see the keys below).  `roofline` is measured live with HIP events around every
launch of the dominant kernel family (the bf16 MFMA GEMM / implicit-GEMM conv engine) on the launch stream;
`cpu_baseline` times the CPU oracle (a port: the reference itself is not importable offline) on the host cores.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# algorithmic work per iteration at the headline config
GFLOP_DECODER = 506.0 + 2.1      # decoder fwd+bwd + VQ distance GEMM (replicated on every rank)
GFLOP_CLIP_PER_CUT = 8.82 + 8.91  # ViT-B/32 fwd + bwd(dgrad) per cutout
PEAK_BF16_TFLOPS = 2500.0         # MI355X_MICROARCH.md: dense bf16 MFMA


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--cutn", type=int, default=64)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--graph", action="store_true",
                    help="replay the iteration from a captured hipGraph (measured neutral on MI355X: the loop is GPU-bound)")
    ap.add_argument("--cpu-iters", type=int, default=2)
    ap.add_argument("--profile-steps", type=int, default=3)
    args = ap.parse_args()

    import torch
    from pixray_amd import _lib, api

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with WORLD_SIZE={args.gpus} (got {world})")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    group = None
    force_dist = os.environ.get("PRX_FORCE_DIST") == "1"      # exercise the RCCL code path on a single GPU (tests)
    if world > 1 or force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        group = dist.group.WORLD
    torch.manual_seed(1234 + rank)          # per-rank device noise streams (host-side draws are seeded identically)

    sess = api.build_vqgan_clip_session(size=(256, 256), vqgan_model="imagenet_f16_16384", clip_model="ViT-B/32",
                                        num_cuts=args.cutn, learning_rate=0.2, iterations=10 ** 9, seed=0, device=dev,
                                        group=group, rank=rank, world_size=world)
    if force_dist and world == 1:
        sess.world_size = 1
        for p_ in sess.perceptors.values():
            p_.group = group              # min/max + renorm-gradient all-reduces over the 1-rank group
        sess._force_hook_group = group

    def barrier():
        if world > 1 or force_dist:
            import torch.distributed as dist
            dist.barrier(group=group)
        torch.cuda.synchronize(dev)

    it = 0
    graphed = False
    if args.graph and world == 1:
        graphed = sess.enable_graph(warmup=max(args.warmup - 1, 1))      # warm-up iterations run inside
        it = sess.cur_iteration
    for _ in range(0 if graphed else args.warmup):
        sess.train(it); it += 1
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        sess.train(it); it += 1
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
        elapsed = float(t.item())
    ms_per_step = 1e3 * elapsed / args.steps
    value = args.steps / elapsed
    loss = float(sum(l.detach() for l in sess.last_losses))

    # ---- roofline leg: per-launch HIP-event timing of the GEMM engine over a few extra steps ------------------
    # (eager launches: events cannot be recorded around the nodes of a replayed graph)
    roofline = None
    sess._drop_graph()
    if rank == 0:
        import ctypes
        prof = api.GemmProfile(sess)
        prof.enable(True)
        for _ in range(args.profile_steps):
            sess.train(it); it += 1
        torch.cuda.synchronize(dev)
        prof.enable(False)
        _ms, _fl, _n = prof.collect()
        ms, fl, n = ctypes.c_double(_ms), ctypes.c_double(_fl), ctypes.c_longlong(_n)
        rc = 0
        # A bracket [event, kernel, event] also times the events' own timestamp packets.  An EMPTY bracket on this stream
        # measures ~5 us; around a kernel about half of that is hidden behind the kernel's own dispatch/drain, and taking
        # half of the empty-bracket time off every launch reproduces rocprofv3's kernel durations for the same command
        # within 2 % (profiles/r01_e_3stage_c64_kernel_stats.csv: 22.4 us per launch incl. split-K reduce passes).
        pairs = []
        for _ in range(200):
            a_, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a_.record(); b_.record(); pairs.append((a_, b_))
        torch.cuda.synchronize(dev)
        ev_over_ms = 0.5 * sorted(x.elapsed_time(y) for x, y in pairs)[len(pairs) // 2]
        raw_ms = ms.value
        if rc == 0 and ms.value > 0:
            ms.value = max(ms.value - ev_over_ms * n.value, 0.5 * ms.value)
            achieved = fl.value / (ms.value * 1e-3) / 1e12
            roofline = {"bound": "mfma", "kernel": "gemm_glds_kernel<BM,BN,AMODE,STAGES,..> (bf16 MFMA GEMM / implicit 3x3 conv, all launches)",
                        "achieved": round(achieved, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                        "frac": round(achieved / PEAK_BF16_TFLOPS, 4),
                        # HBM-side bytes per GEMM launch from the PMC passes committed in profiles/r01_h_pmc_hbm_traffic.csv
                        # (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs; FETCH doubled per the gfx950
                        # correction of MI355X_MICROARCH.md); not re-measured live
                        "traffic": 42.5e6, "traffic_source": "profiles/r01_h_pmc_hbm_traffic.csv",
                        "launches_per_step": n.value // args.profile_steps,
                        "gemm_gflop_per_step": round(fl.value / args.profile_steps / 1e9, 1),
                        "gemm_ms_per_step": round(ms.value / args.profile_steps, 3),
                        "avg_launch_us": round(1e3 * ms.value / max(n.value, 1), 2),
                        "event_overhead_us_removed_per_launch": round(1e3 * ev_over_ms, 2),
                        "avg_launch_us_raw_events": round(1e3 * raw_ms / max(n.value, 1), 2)}
    elif world > 1:
        for _ in range(args.profile_steps):      # keep ranks in lock-step through the collectives
            sess.train(it); it += 1

    per_gpu_gflop = GFLOP_DECODER + GFLOP_CLIP_PER_CUT * args.cutn / world
    iter_frac = value * per_gpu_gflop * 1e9 / (PEAK_BF16_TFLOPS * 1e12)

    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import step_ref
        r = step_ref.time_oracle_iterations(n_iters=args.cpu_iters, warmup=1, cutn=args.cutn)
        cpu_baseline = {"value": round(r["iters_per_sec"], 4), "unit": "iterations/s", "cores": r["threads"],
                        "kind": "port",
                        "sample": f"{args.cpu_iters} full iterations (same config, fp32 torch CPU oracle, "
                                  f"{r['threads']} threads of {r['cores']} host cores) after 1 warm-up"}

    if rank == 0:
        out = {
            "metric": "optimisation iters/sec, VQGAN 256^2 + ViT-B/32 + 64 cutouts",
            "value": round(value, 3), "unit": "iterations/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "vqgan imagenet_f16_16384 256x256 + CLIP ViT-B/32 + %d cutouts, 1 prompt, Adam lr 0.2"
                                   % args.cutn,
                       "weights": "seeded random, real architectures", "cutouts_per_gpu": args.cutn // world,
                       "parallelism": f"cutout-sharded x{world}, all-reduce of dL/d(image)" if world > 1 else "single GPU",
                       "launch": "hipGraph replay" if graphed else "eager"},
            "final_loss": round(loss, 5),
            "per_gpu_gflop_per_step": round(per_gpu_gflop, 1),
            "iter_mfma_frac": round(iter_frac, 4),
            "roofline": roofline, "cpu_baseline": cpu_baseline,
        }
    if world > 1 or force_dist:
        import torch.distributed as dist
        dist.destroy_process_group()      # RCCL prints its version banner here: keep the JSON line last
    sys.stderr.flush()
    try:                                   # RCCL writes its banner through C stdio: drain it before the JSON line
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
