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
