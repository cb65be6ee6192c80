cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
PRX_MHA_TILES=1 timeout 300 python -m pytest tests/test_path_gpu.py -x -q -m gpu -s -k "test_clip_resnet_vs_oracle" 2>&1 | grep -E "emb rel|passed|failed" > gpurun_out/r02l_rn_tiles.log 2>&1
timeout 300 python - > gpurun_out/r02l_rn_dbg.log 2>&1 <<'PY'
import torch
from pixray_amd import ops, weights
cfg = weights.CLIP_RESNET_CONFIGS["RN50x4"]
p = weights.synthetic_clip_resnet_params(cfg, seed=3)
g = torch.Generator().manual_seed(17)
R = cfg.input_resolution
for n in (2, 4, 8):
    low = torch.rand(n, 3, R // 8, R // 8, generator=g)
    cut = (torch.nn.functional.interpolate(low, size=(R, R), mode="bilinear", align_corners=False) + 0.05 * torch.randn(n, 3, R, R, generator=g))
    ge = torch.randn(n, cfg.output_dim, generator=g)
    res = {}
    for prec in ("f32", "bf16"):
        h = ops.ClipResNetHandle(cfg, p, max_batch=8, device="cuda", precision=prec)
        cd = cut.to("cuda").requires_grad_(True)
        emb = ops.clip_encode_image(cd, h)
        (gd,) = torch.autograd.grad(emb, cd, ge.to("cuda"))
        res[prec] = (emb.detach().double().cpu(), gd.double().cpu())
    a, b = res["bf16"][1], res["f32"][1]
    cos = float((a.flatten() @ b.flatten()) / (a.norm() * b.norm()))
    print(n, "emb rel", float((res["bf16"][0] - res["f32"][0]).norm() / res["f32"][0].norm()), "grad norm ratio bf16/f32", float(a.norm() / b.norm()), "cos", cos,
          "rel", float((a - b).norm() / b.norm()))
    for i in range(n):
        print("   cutout", i, "norm ratio", float(a[i].norm() / b[i].norm()))
PY
echo done > gpurun_out/r02l_rc.txt
