"""Where does a fast mode's gradient error come from?  Two sessions (exact-f32 and the fast precision) on the same reduced /
headline configuration and draws; hooks capture dL/d(embeddings), dL/d(cutouts) and dL/d(image) on both, and the decoder is
additionally fed the f32 session's image gradient so that its own backward error is seen in isolation.
    python tools/grad_stage_probe.py [reduced|wide|headline] [fp16|bf16]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import step_ref  # noqa: E402


def rel(a, b):
    a, b = a.double().flatten().cpu(), b.double().flatten().cpu()
    return float((a - b).norm() / b.norm()), float(a @ b / (a.norm() * b.norm()))


def run(case, prec):
    kw = dict(vqgan_model="tiny_f4", clip_model="tiny-B/32", size=(64, 64), cutn=8, seed=0)
    if case == "wide":
        kw.update(size=(112, 64), seed=3)
    elif case == "headline":
        kw = dict(vqgan_model="imagenet_f16_16384", clip_model="ViT-B/32", size=(256, 256), cutn=64, seed=0)
    cap = {}
    for p in ("f32", prec):
        sess = step_ref._build_hip(kw["vqgan_model"], kw["clip_model"], kw["size"], kw["cutn"], kw["seed"], "cuda:0", precision=p)
        S = next(iter(sess.cutoutsTable))
        mk = sess.cutoutsTable[S]
        mk.fixed_params = step_ref._draws(kw["cutn"], S, kw["seed"], 0, aspect=kw["size"][0] / kw["size"][1])
        c = cap[p] = {}
        real_mk = mk.forward

        def mk_fwd(x, *a, _c=c, _f=real_mk, **k):
            x.register_hook(lambda g, _c=_c: _c.__setitem__("g_img", g.detach().clone()))
            _c["img"] = x.detach().clone()
            out = _f(x, *a, **k)
            out.register_hook(lambda g, _c=_c: _c.__setitem__("g_cut", g.detach().clone()))
            _c["cut"] = out.detach().clone()
            return out
        mk.forward = mk_fwd
        losses = sess.ascend_txt()
        sess.last_embeds.register_hook(lambda g, _c=c: _c.__setitem__("g_emb", g.detach().clone())) if sess.last_embeds.requires_grad else None
        sum(losses).backward()
        c["dz"] = sess.drawer.get_z().grad.detach().clone()
        c["sess"] = sess
    a, b = cap[prec], cap["f32"]
    print(f"== {case} {prec} vs f32 (rel-L2, cosine)")
    for k in ("img", "cut", "g_cut", "g_img", "dz"):
        if k in a and k in b:
            print(f"   {k:6s}", rel(a[k], b[k]))
    # decoder backward alone: feed the f32 image gradient to the fast session's decoder
    s = a["sess"]
    z = s.drawer.get_z()
    z.grad = None
    img = s.drawer.synth(0)
    img.backward(b["g_img"])
    print("   dz from the f32 image gradient through the fast decoder backward:", rel(z.grad, b["dz"]))
    print("   |g_img| amax", float(b["g_img"].abs().max()), "median", float(b["g_img"].abs().median()), "|g_cut| amax", float(b["g_cut"].abs().max()),
          "median", float(b["g_cut"].abs().median()))


if __name__ == "__main__":
    case = sys.argv[1] if len(sys.argv) > 1 else "reduced"
    for prec in (sys.argv[2:] or ["fp16", "bf16"]):
        run(case, prec)
