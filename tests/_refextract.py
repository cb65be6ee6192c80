"""Run pixray's own pure-torch fragments straight from the read-only reference checkout.

`import pixray` is impossible offline (kornia, clip, taming, ... are missing), but single definitions can be
pulled out of the source by AST and exec'd.  Only used in this container (tests skip when /root/reference is
absent, e.g. on the GPU box); nothing is copied into the repository."""
import ast
import os

import torch
import torch.nn.functional as F
from torch import nn

REF = "/root/reference"


def available() -> bool:
    return os.path.isfile(os.path.join(REF, "pixray.py"))


def extract(relpath: str, names, extra_ns=None) -> dict:
    src = open(os.path.join(REF, relpath)).read()
    tree = ast.parse(src)
    ns = {"torch": torch, "F": F, "nn": nn}
    if extra_ns:
        ns.update(extra_ns)
    wanted = set(names)
    for node in tree.body:
        name = getattr(node, "name", None)
        if isinstance(node, ast.Assign) and len(node.targets) == 1 and isinstance(node.targets[0], ast.Name):
            name = node.targets[0].id
        if name in wanted:
            code = ast.get_source_segment(src, node)
            exec(compile(code, f"{relpath}:{name}", "exec"), ns)
            wanted.discard(name)
    if wanted:
        raise KeyError(f"{relpath}: not found: {sorted(wanted)}")
    return ns


def pixray_prompt_ns():
    return extract("pixray.py", ["ReplaceGrad", "replace_grad", "spherical_dist_loss", "Prompt", "is_number", "parse_prompt"])


def vqgan_ns():
    return extract("vqgan.py", ["ReplaceGrad", "replace_grad", "vector_quantize", "ClampWithGrad", "clamp_with_grad"])


STROTSS_NAMES = ["Vgg16_Extractor", "tensor_resample", "laplacian", "make_laplace_pyramid", "fold_laplace_pyramid", "sample_indices",
                 "spatial_feature_extract", "pairwise_distances_cos", "pairwise_distances_sq_l2", "distmat", "content_loss",
                 "rgb_to_yuv", "style_loss", "moment_loss", "calculate_loss", "scale_loss", "strotss_loss"]


def styleloss_ns():
    """pixray's own STROTSS functions and `Vgg16_Extractor` class (Losses/StyleLoss.py); torchvision is absent, so the
    extractor is instantiated without its __init__ (see `reference_vgg_extractor`)."""
    import math
    import numpy as np
    return extract("Losses/StyleLoss.py", STROTSS_NAMES, {"np": np, "math": math})


def reference_vgg_extractor(ns, params, space="uniform"):
    """the reference's Vgg16_Extractor around an nn.Sequential laid out like torchvision's vgg16().features (Conv2d,
    ReLU(inplace), MaxPool2d per cfg 'D'), filled from `params`; __init__ (which downloads the pretrained net) is bypassed"""
    cfg = (64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512, "M")
    layers, cin = [], 3
    for v in cfg:
        if v == "M":
            layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
        else:
            layers += [nn.Conv2d(cin, v, kernel_size=3, padding=1), nn.ReLU(inplace=True)]
            cin = v
    seq = nn.Sequential(*layers)
    seq.load_state_dict({k[len("features."):]: v for k, v in params.items()})
    ex = ns["Vgg16_Extractor"].__new__(ns["Vgg16_Extractor"])
    nn.Module.__init__(ex)
    ex.vgg_layers = seq
    for p_ in ex.parameters():
        p_.requires_grad = False
    ex.capture_layers = [1, 3, 6, 8, 11, 13, 15, 22, 29]
    ex.space = space
    return ex
