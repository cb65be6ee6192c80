"""GPU test of the C-ABI exchange step (`prx_comm` / `prx_allreduce_grad`, include/prx.h, csrc/comm.hip): the one-shot
direct-write all-reduce over IPC-mapped peer windows.

Only one GPU is reachable, and RCCL refuses two ranks on one device -- but the protocol of this collective (handle exchange,
window mapping, push / flag / wait / ordered sum, slot parity across back-to-back calls) does not care whether the peer
window sits on another GPU: two PROCESSES sharing cuda:0 exercise every line of it.  The xGMI transport and its timing are
what the driver's multi-GPU run adds (bench.py: collectives_ms_per_step)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)        # host-side bootstrap only (handle exchange)
    torch.cuda.set_device(0)
    from pixray_amd.comm import OneShotComm
    comm = OneShotComm(dist.group.WORLD, rank, world, max_bytes=4 << 20)
    res = {}
    # (a) the image-gradient payload of the headline configuration (786 KB), several back-to-back calls: exercises both slot
    # parities and the "at most one call ahead" argument; each rank contributes a different, seeded vector
    outs = []
    for it in range(6):
        g = torch.Generator().manual_seed(100 * it + rank)
        x = torch.randn(1, 3, 256, 256, generator=g).cuda()
        comm.all_reduce_sum_(x)
        outs.append(x.cpu())
    torch.cuda.synchronize()
    res["outs"] = [o.numpy() for o in outs]
    # (b) a scalar-sized, unaligned vector (the min / max renormalisation sums)
    v = torch.tensor([1.0 + rank, -2.0 * (rank + 1), 0.5], device="cuda")[1:]       # 2 floats at a 4-byte offset
    comm.all_reduce_sum_(v)
    res["small"] = v.cpu().numpy()
    # (c) the two scalar-sized collectives of the sharded iteration on the same exchange: MAX of the {-min, max} pair (slip.py:21-36),
    # SUM of the four fp64 renormalisation sums
    mm = torch.tensor([-0.25 - rank, 0.75 + 0.125 * rank], device="cuda")
    comm.all_reduce_(mm, "max")
    res["max"] = mm.cpu().numpy()
    acc = torch.tensor([1e-9 * (rank + 1), 1.0 + rank, 3.0, 2.0 ** 40 * (rank + 1)], device="cuda", dtype=torch.float64)
    comm.all_reduce_(acc, "sum")
    res["f64"] = acc.cpu().numpy()
    comm.check()
    res["status"] = comm.status()
    q.put((rank, res))
    dist.barrier()
    comm.close()
    dist.destroy_process_group()


def test_oneshot_allreduce_two_processes_one_device():
    import numpy as np
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        got = {}
        for _ in range(world):
            r, res = q.get(timeout=240)
            got[r] = res
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
    finally:
        for p in procs:
            if p.is_alive():
                p.terminate()
    assert got[0]["status"] == 0 and got[1]["status"] == 0
    for it in range(6):
        want = sum(torch.randn(1, 3, 256, 256, generator=torch.Generator().manual_seed(100 * it + r)) for r in range(world))
        a, b = got[0]["outs"][it], got[1]["outs"][it]
        assert np.array_equal(a, b), "ranks diverged"                 # summed in rank order on every rank: bit-identical
        assert np.array_equal(a, want.numpy()), it                     # 2 addends: the fp32 sum is exact to the last bit
    assert np.allclose(got[0]["small"], [-6.0, 1.0]) and np.array_equal(got[0]["small"], got[1]["small"])
    for r in range(world):
        assert np.array_equal(got[r]["max"], np.float32([-0.25, 0.875]))
        assert np.array_equal(got[r]["f64"], np.float64([1e-9 * 1 + 1e-9 * 2, 3.0, 6.0, 3 * 2.0 ** 40]))      # summed in rank order


def test_bench_gpus_2_bare_launch_two_ranks_on_one_device():
    """`python bench.py --gpus 2` with no launcher: spawns its two ranks itself, shards the cutouts, runs all three collectives of
    the iteration on the C-ABI one-shot exchange (both ranks on cuda:0, gloo for the bootstrap only) and prints ONE JSON line.
    The sharded loss equals the single-process run's to fp16 round-off of the tower batch split."""
    import json
    import subprocess
    root = os.path.dirname(HERE)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(PRX_ONE_DEVICE="1", PRX_DIST_BACKEND="gloo")
    common = ["--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-other-modes", "--profile-steps", "1", "--phase-steps", "0"]
    recs = {}
    for n in (2, 1):
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(n)] + common, capture_output=True, text=True,
                             timeout=900, env=env if n > 1 else {k: v for k, v in env.items() if not k.startswith("PRX_")})
        assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
        lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1, out.stdout[-2000:]
        recs[n] = json.loads(lines[0])
    r2, r1 = recs[2], recs[1]
    assert r2["n_gpus"] == 2 and r2["config"]["cutouts_per_gpu"] == 32 and "one-shot" in r2["config"]["exchange"]
    assert r2["value"] > 0 and np.isfinite(r2["final_loss"])
    assert abs(r2["final_loss"] - r1["final_loss"]) < 5e-2 * abs(r1["final_loss"]), (r2["final_loss"], r1["final_loss"])
    # the same two ranks with the iteration captured in a hipGraph: the exchange's sequence numbers live on the device (csrc/comm.hip),
    # so the three collectives replay like any other launch -- same losses as the eager sharded run
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--graph"] + common, capture_output=True, text=True,
                         timeout=900, env=env)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    rg = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert rg["config"]["launch"] == "hipGraph replay", rg["config"]["launch"]
    assert abs(rg["final_loss"] - r2["final_loss"]) < 2e-2 * abs(r2["final_loss"]), (rg["final_loss"], r2["final_loss"])


def test_oneshot_allreduce_world_one_is_the_identity():
    from pixray_amd.comm import OneShotComm
    comm = OneShotComm(None, 0, 1, max_bytes=1 << 20)
    x = torch.randn(1024, device="cuda")
    y = x.clone()
    comm.all_reduce_sum_(y)
    assert torch.equal(x, y) and comm.status() == 0
    comm.close()
