"""Diagnostic (run on the GPU box): the measured parity figures of the two runners alone -- CLIP ViT tower (embeddings, gradient to the
cutouts) and VQGAN decoder (image, gradient to z) against the oracle in the default precision -- the numbers the per-runner gates of
tests/test_path_gpu.py are set from.    python tools/runner_gates.py"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import torch
import test_path_gpu as tp
from test_path_gpu import rel_l2, cosine
for name, n in [("tiny-B/32", 4), ("ViT-B/32", 8), ("test-B/16", 3), ("test-L/14", 2)]:
    ref, out, gref, gd = tp._clip_case(name, n, 5)
    print("clip", name, "emb rel %.2e cos %.6f | grad rel %.2e cos %.6f" % (rel_l2(out, ref), cosine(out, ref), rel_l2(gd, gref), cosine(gd, gref)))
for name, hw in [("tiny_f4", 16), ("imagenet_f16_16384", 16), ("imagenet_f16_16384", 32), ("tiny_f4", (12, 20)), ("imagenet_f16_16384", (14, 25))]:
    ref, out, gref, gd, idx_ref, idx = tp._vqgan_case(name, hw, 9)
    print("vqgan", name, hw, "img rel %.2e | grad rel %.2e cos %.6f idx_equal %s" % (rel_l2(out, ref), rel_l2(gd, gref), cosine(gd, gref), bool(torch.equal(idx, idx_ref))))
