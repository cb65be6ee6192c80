"""GPU probe of the three operand precisions against the CPU oracle: headline, the reduced / widescreen graphs of smoke(),
cfg2 at the per-GPU shard size, and the RN50x4 tower on its own.  Prints one JSON object per case (the numbers quoted in
DESIGN.md section 4 / BASELINE.md section 3); `PRX_GRAD_SCALE_LOG2` can be swept from the environment."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

from oracle import step_ref, workload_ref  # noqa: E402


def main():
    which = sys.argv[1:] or ["reduced", "wide", "headline", "cfg2"]
    out = {}
    t0 = time.time()
    for case in which:
        if case == "reduced":
            kw = dict(vqgan_model="tiny_f4", clip_model="tiny-B/32", size=(64, 64), cutn=8, seed=0)
        elif case == "wide":
            kw = dict(vqgan_model="tiny_f4", clip_model="tiny-B/32", size=(112, 64), cutn=8, seed=0)
        elif case == "headline":
            kw = dict()
        else:
            kw = None
        if kw is not None:
            out[case] = {}
            for prec in ("f32", "fp16", "bf16"):
                r = step_ref.compare_one_iteration(precision=prec, **kw)
                out[case][prec] = {k: r[k] for k in ("dz_rel_l2", "dz_cosine", "loss_abs_err", "embeds_rel_l2", "image_rel_l2", "indices_equal")}
        elif case == "cfg2":
            out[case] = workload_ref.compare_workload("cfg2", 16, precisions=("f32", "fp16", "bf16"))
        elif case == "cfg3":
            import bench
            out[case] = workload_ref.compare_workload(
                "cfg3", 32, precisions=("f32", "fp16", "bf16"),
                custom_factory=lambda prec: [{"loss": bench.make_saturation_loss("cuda:0"), "weight": 1.0}],
                custom_ref=[{"loss": workload_ref.SaturationLossRef(), "weight": 1.0}])
        print(f"[{time.time() - t0:6.1f}s] {case}: {json.dumps(out[case])}", flush=True)
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/fp16_probe.json", "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
