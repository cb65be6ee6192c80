"""Which part of a configs[3]-shaped session refuses hipGraph capture?  Builds small sessions that differ in one piece
(canvas shape, plugin stack) and prints Session.enable_graph's verdict for each (run on the GPU box)."""
import argparse
import os
import sys
import types
import warnings

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pixray_amd import api, weights
from pixray_amd import style_loss as sl
from pixray_amd.cutouts import MakeCutouts
from pixray_amd.engine import Session
from pixray_amd.fft_drawer import FftDrawer
from pixray_amd.interfaces import LossInterface
from pixray_amd.perceptor import get_clip_perceptor
from pixray_amd.prompt import Prompt

DEV = torch.device("cuda", 0)


class Saturation(LossInterface):
    supports_graph_replay = True

    def get_loss(self, cur_cutouts, out, args, globals=None, lossGlobals=None):
        cut = next(iter(cur_cutouts.values()))
        return -cut.std(dim=1).mean() * 0.2


def build(size, losses, skip="1", max_hw=None):
    st = types.SimpleNamespace(size=size, fft_use="fft", fft_decay=1.5, fft_lrate=0.3)
    dr = FftDrawer(st)
    dr.load_model(st, DEV)
    dr.init_from_tensor(None)
    perc = get_clip_perceptor("tiny-B/32", DEV, max_batch=8)
    mk = MakeCutouts(224, 8, generator=torch.Generator().manual_seed(3), aspect_width=size[0] / size[1])
    pm = Prompt(api.seeded_unit_vectors(1, 128, 9).to(DEV), 1.0, float("-inf")).to(DEV)
    args = sl.StyleLoss.add_settings(argparse.ArgumentParser()).parse_args(["--styleloss_skip", skip, "--styleloss_content_weight", "8"])
    custom = []
    if "style" in losses:
        ext = None
        if max_hw:
            ext = sl.Vgg16Extractor(params=weights.synthetic_vgg16_params(0), device=DEV, max_hw=max_hw)
        style = sl.StyleLoss(extractor=ext, vgg_params=weights.synthetic_vgg16_params(0),
                             style_image=torch.rand(1, 3, 50, 60, generator=torch.Generator().manual_seed(4)), device=DEV)
        args = style.parse_settings(args)
        custom.append({"loss": style, "weight": 1.0})
    if "sat" in losses:
        custom.append({"loss": Saturation(device=DEV), "weight": 1.0})
    return Session(dr, {"tiny-B/32": perc}, {224: mk}, {"tiny-B/32": [pm]}, args=args, seed=1, custom_losses=custom)


cases = [("96x80 no plugins", (96, 80), ()), ("128x128 no plugins", (128, 128), ()), ("96x80 saturation", (96, 80), ("sat",)),
         ("128x128 style skip0 reserved", (128, 128), ("style",), "0", (128, 128)),
         ("128x128 style skip1", (128, 128), ("style",), "1"),
         ("96x80 style skip0 reserved", (96, 80), ("style",), "0", (96, 96)),
         ("96x80 style skip1", (96, 80), ("style",), "1"),
         ("96x80 style+sat skip1", (96, 80), ("style", "sat"), "1")]
only = sys.argv[1:]
for c in cases:
    name, size, losses = c[0], c[1], c[2]
    if only and not any(o in name for o in only):
        continue
    np.random.seed(0)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        sess = build(size, losses, *(c[3:]))
        ok = sess.enable_graph(warmup=2)
        if ok:
            for _ in range(3):
                sess.train()
            torch.cuda.synchronize()
    print(f"{name:32s} -> {'captured + replayed' if ok else 'REFUSED: ' + str(sess.graph_error).splitlines()[0]}", flush=True)


def interleave(mode):
    """two sessions like tests/test_e2e_gpu.py's replay test; prints after every synchronised step (who faults, when)"""
    np.random.seed(0)
    a = build((96, 80), ("style", "sat"), "1")
    b = build((96, 80), ("style", "sat"), "1")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for it in range(3):
            a.train(it)
        torch.cuda.synchronize(); print("a eager 0..2 ok", flush=True)
        np.random.seed(0)
        assert b.enable_graph(warmup=2), b.graph_error
        b.train(2)
        torch.cuda.synchronize(); print("b first replay ok", flush=True)
        for it in range(3, 7):
            if mode == "alloc":        # no second session: churn the allocator between replays instead
                junk = [torch.randint(0, 1 << 30, (1 << 20,), device=DEV) for _ in range(64)]
                torch.cuda.synchronize(); del junk
            else:
                state = np.random.get_state()
                a.train(it)
                torch.cuda.synchronize(); print(f"a.train({it}) ok", flush=True)
                np.random.set_state(state)
            b.train(it)
            torch.cuda.synchronize(); print(f"b.train({it}) ok", [round(float(l), 5) for l in b.last_losses], flush=True)


if only and only[0].startswith("interleave"):
    interleave(only[0].split(":")[-1])


def churn(which):
    """one replayed session; between replays the allocator is churned (freed blocks get overwritten with random integers),
    so a captured kernel that reads memory nobody keeps alive faults instead of silently reading stale data"""
    cfg = {"none": ((96, 80), ()), "sat": ((96, 80), ("sat",)), "style0": ((128, 128), ("style",), "0", (128, 128)),
           "style1": ((96, 80), ("style",), "1"), "style0s": ((96, 80), ("style",), "0", (96, 96))}[which]
    np.random.seed(0)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        b = build(*cfg)
        assert b.enable_graph(warmup=2), b.graph_error
        for it in range(4):
            junk = [torch.randint(0, 1 << 30, (1 << 20,), device=DEV) for _ in range(64)]
            torch.cuda.synchronize(); del junk
            b.train()
            torch.cuda.synchronize()
    print(f"churn {which}: 4 replays ok", [round(float(l), 5) for l in b.last_losses], flush=True)


if only and only[0].startswith("churn"):
    churn(only[0].split(":")[-1])


def piece(which):
    """capture loss(x).backward() of ONE part of the StyleLoss arithmetic, then replay under allocator churn"""
    torch.manual_seed(0)
    np.random.seed(0)
    H = W = 128
    x = torch.rand(1, 3, H, W, device=DEV, requires_grad=True)
    ext = sl.Vgg16Extractor(params=weights.synthetic_vgg16_params(0), device=DEV, max_hw=(H, W))
    shapes = sl.vgg_map_shapes(H, W)
    xx, xy = sl._sample_grid(H, W)
    rows, wts = sl._draw_bilinear_tables(shapes, xx[:1024], xy[:1024])
    rows_d, wts_d = torch.from_numpy(rows).to(DEV), torch.from_numpy(wts).to(DEV)
    srows = torch.from_numpy(sl._draw_hypercolumn_rows(shapes, 1000)).to(DEV)
    cols_a = torch.rand(1, 2181, 1024, 1, device=DEV, requires_grad=True)
    cols_b = torch.rand(1, 2181, 1024, 1, device=DEV)
    sty = torch.rand(1, 2179, 5000, 1, device=DEV)

    def f():
        if which == "vgg":
            return sum(t.mean() for t in ext(x))
        if which == "hyper":
            fa, fb = ext(x), ext(x * 0.5)
            a, b = sl._bilinear_columns_dev(fa, fb, rows_d, wts_d)
            return a.mean() + b.mean()
        if which == "gather":
            with torch.no_grad():
                g = sl._gather_hypercolumns([t.detach() for t in ext(x)], srows)
            return g.mean() * x.mean()
        if which == "selfsim":
            return sl._self_similarity_loss(cols_a, cols_b)
        if which == "remd":
            return sl._remd(cols_a[:, :2179], sty)
        if which == "remd3":
            return sl._remd(cols_a[:, :3], sty[:, :3])
        if which == "moment":
            return sl._moment_loss(cols_a[:, :-2], sty)
        if which == "mm":          # the relaxed-EMD distance product on its own: [1024, 2179] x [2179, 5000]
            return torch.mm(cols_a[0, :2179, :, 0].t(), sty[0, :, :, 0]).mean()
        if which == "mm3":
            return torch.mm(cols_a[0, :3, :, 0].t(), sty[0, :3, :, 0]).mean()
        if which == "min":
            M = cols_a[0, :1024, :, 0]
            return torch.max(M.min(1)[0].mean(), M.min(0)[0].mean())
        Xc, Yc = sl._columns(cols_a[:, :2179]), sl._columns(sty)
        X3, Y3 = sl._columns(cols_a[:, :3]), sl._columns(sty[:, :3])
        if which == "cos":
            return sl._cos_dist(Xc, Yc).mean()
        if which == "cosmin1":
            return sl._cos_dist(Xc, Yc).min(1)[0].mean()
        if which == "cosmin0":
            return sl._cos_dist(Xc, Yc).min(0)[0].mean()
        if which == "minbig":
            M = cols_a[0, :1250, :, 0].reshape(250, 5120)[:, :5000]
            return torch.max(M.min(1)[0].mean(), M.min(0)[0].mean())
        if which == "yuv":
            C = sl._const("yuv", sl._YUV, X3.device)
            return torch.mm(C, X3.t()).t().mean()
        if which == "l2":
            return sl._l2_dist(X3, Y3).mean()
        if which == "sqsumY":
            return (Yc ** 2).sum(1).mean() * cols_a.mean()
        if which == "sqsumX":
            return (Xc ** 2).sum(1).mean()
        if which == "normdiv":
            xn = torch.sqrt((Xc ** 2).sum(1).view(-1, 1))
            return (torch.mm(Xc, Yc.t()) / xn).mean()
        if which == "normdivY":
            yn = torch.sqrt((Yc ** 2).sum(1).view(1, -1))
            return (torch.mm(Xc, Yc.t()) / yn).mean()
        if which == "rsub":
            return (1. - torch.mm(Xc, Yc.t())).mean()
        if which == "cos3":
            return sl._cos_dist(X3, Y3).mean()
        if which == "pyramid":
            return sl._fold_pyramid(sl._laplace_pyramid(x, 5)).square().mean()
        raise SystemExit(which)

    leaves = [x, cols_a]
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            for l in leaves:
                l.grad = None
            f().backward()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    for l in leaves:
        l.grad = None
    le = f()
    le.backward()
    print(f"piece {which}: eager loss {float(le):.6f} grad norms {[float(l.grad.norm()) if l.grad is not None else None for l in leaves]}", flush=True)
    del le
    g = torch.cuda.CUDAGraph()
    for l in leaves:
        l.grad = None
    with torch.cuda.graph(g):
        loss = f()
        loss.backward()
    g.replay()
    torch.cuda.synchronize()
    print(f"piece {which}: first replay loss {float(loss):.6f} grad norms {[float(l.grad.norm()) if l.grad is not None else None for l in leaves]}", flush=True)
    if os.environ.get("SNAP"):
        snap = torch.cuda.memory_snapshot()
        pools = {}
        for seg in snap:
            pools.setdefault(tuple(seg.get("segment_pool_id", (0, 0))), []).append((seg["address"], seg["address"] + seg["total_size"], seg.get("stream", 0)))
        for pid, segs in pools.items():
            print("pool", pid, "segments", len(segs), "bytes", sum(b - a for a, b, _ in segs), "streams", sorted({s_ for _, _, s_ in segs})[:4], flush=True)
        priv = [se for pid, segs in pools.items() if pid != (0, 0) for se in segs]
        print("loss ptr in private pool:", any(a <= loss.data_ptr() < b for a, b, _ in priv), flush=True)
        junk = [torch.randint(0, 1 << 30, (1 << 20,), device=DEV) for _ in range(64)]
        inside = sum(any(a <= j.data_ptr() < b for a, b, _ in priv) for j in junk)
        print(f"junk tensors whose memory lies inside a private-pool segment: {inside} of {len(junk)}", flush=True)
        torch.cuda.synchronize(); del junk
    mode = int(os.environ.get("CHURN", "1"))       # 0: plain replays; 1: overwrite freed memory; 2: allocate without writing
    vals = []
    for it in range(4):
        if mode == 1:
            junk = [torch.randint(0, 1 << 30, (1 << 20,), device=DEV) for _ in range(64)]
            torch.cuda.synchronize(); del junk
        elif mode == 2:
            junk = [torch.empty(1 << 20, dtype=torch.int64, device=DEV) for _ in range(64)]
            torch.cuda.synchronize(); del junk
        g.replay()
        torch.cuda.synchronize()
        vals.append(round(float(loss), 6))
    print(f"piece {which}: churn mode {mode}: 4 replays ok, losses {vals}", flush=True)


if only and only[0].startswith("piece"):
    piece(only[0].split(":")[-1])
