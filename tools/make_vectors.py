"""Extracts the reference's default vector prompt into package data (run in the build container, where /root/reference is present):

    python tools/make_vectors.py

`--vector_prompts textoff` is pixray's DEFAULT (pixray.py:1732): every run adds, per perceptor, a Prompt on a precomputed CLIP-space
vector from `vectors/textoff.json` at weight 0.1 (pixray.py:887-915).  The table is data the reference ships, keyed by CLIP
model name; this keeps the entries of the towers this package implements, value for value (the reference reads them into a
FloatTensor), in the reference's own json layout {model name: [[...]]} so that a user's own vector files load the same way.
The GPU box has no /root/reference, hence the committed copy under pixray_amd/vectors/."""
import json
import os

SRC = "/root/reference/vectors/textoff.json"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEEP = ["RN50", "RN101", "RN50x4", "ViT-B/32", "ViT-B/16"]        # weights.CLIP_CONFIGS / CLIP_RESNET_CONFIGS names present in the table

with open(SRC) as f:
    table = json.load(f)
out = {k: table[k] for k in KEEP}
dst = os.path.join(ROOT, "pixray_amd", "vectors", "textoff.json")
with open(dst, "w") as f:
    json.dump(out, f)
print(dst, {k: (len(v), len(v[0])) for k, v in out.items()})
