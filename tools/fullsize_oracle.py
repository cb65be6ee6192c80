"""Full-size CPU-oracle gradients for BASELINE.json configs[2] / configs[3] -> tests/golden/fullsize_<cfg>.npz.

    python tools/fullsize_oracle.py cfg2            # VQGAN 512^2 + ViT-B/16 + RN50x4, 128 cutouts each
    python tools/fullsize_oracle.py cfg3            # fft 512^2 + ViT-L/14, 256 cutouts + StyleLoss + SaturationLoss

The oracle (oracle/fullsize_ref.py: workload_ref.iteration evaluated in cutout chunks, same arithmetic) takes ~1 h / ~5 h on
the 8 cores of the build container, which is why its output is a committed fixture; the GPU test
(tests/test_fullsize_gpu.py) runs the HIP path at the same size from the same seeds and compares.  Everything the HIP side
needs is a function of (workload, seed): weights, start point and augmentation draws are regenerated there.
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

from oracle import fullsize_ref, workload_ref  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("workload", choices=["cfg1", "cfg2", "cfg3"])
    ap.add_argument("--cutn", type=int, default=0)
    ap.add_argument("--chunk", type=int, default=16)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--np-seed", dest="np_seed", type=int, default=0, help="numpy seed of the StyleLoss draws (cfg3)")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    from pixray_amd import api
    if a.threads:
        torch.set_num_threads(a.threads)
    cutn = a.cutn or api.WORKLOADS[a.workload]["num_cuts"]
    custom, largs = [], None
    if a.workload == "cfg3":
        # BASELINE.json configs[3]'s custom_loss stack: StyleLoss (the plugin's STROTSS arithmetic on the CPU VGG16 oracle, the
        # reference's own schedule of one extractor pass per style draw, Losses/StyleLoss.py:24-47,458-500) + SaturationLoss.
        # STROTSS samples from numpy's global generator: seeded here, and identically in tests/test_fullsize_gpu.py
        import bench
        custom, largs = bench.cfg3_custom_losses("cpu", None, on_cpu=True)
        np.random.seed(a.np_seed)
    t0 = time.perf_counter()

    def log(msg):
        print(f"[{time.perf_counter() - t0:8.1f}s] {msg}", flush=True)

    r = fullsize_ref.iteration_chunked(a.workload, cutn, a.seed, custom=custom, args=largs, chunk=a.chunk, log=log)
    out = a.out or os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", f"fullsize_{a.workload}.npz")
    np.savez_compressed(out, grad=r["grad"].numpy(), losses=np.asarray(r["losses"], np.float64), embeds=r["embeds"].numpy(),
                        img_mean=np.float64(r["img"].double().mean()), img_sq=np.float64((r["img"].double() ** 2).mean()),
                        start_sq=np.float64((r["start"].double() ** 2).sum()), cutn=cutn, seed=a.seed,
                        np_seed=a.np_seed, n_terms=len(r["losses"]), seconds=time.perf_counter() - t0, threads=torch.get_num_threads())
    log(f"wrote {out}: losses {r['losses']} |grad| {float(r['grad'].norm()):.6e}")


if __name__ == "__main__":
    main()
