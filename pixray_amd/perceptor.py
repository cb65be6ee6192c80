"""Perceptor wrapper with the reference's `CLIP_Base` surface (/root/reference/slip.py:44-74): attributes
`input_resolution`, `output_dim`; `encode_image(imgs) -> [N, D]` L2-normalised and differentiable; frozen
weights.  The arithmetic is the HIP CLIP ViT runner (pixray_amd/csrc/vit.hip)."""
import torch

from . import ops
from .weights import CLIP_CONFIGS, ClipVitConfig, synthetic_clip_vit_params


class ClipVitPerceptor:
    def __init__(self, cfg: ClipVitConfig, params, device, max_batch: int = 64, group=None):
        self.cfg = cfg
        self.device = torch.device(device)
        self.input_resolution = cfg.input_resolution
        self.output_dim = cfg.output_dim
        self.group = group          # torch.distributed group when the cutout batch is sharded
        self.handle = ops.ClipVitHandle(cfg, params, max_batch, self.device)

    def preprocess(self, imgs, input_range=None):
        raise NotImplementedError("preprocessing (slip.py:21-42,58-60) is fused into encode_image on this path")

    def encode_image(self, imgs, input_range=None, apply_preprocess=True):
        """slip.py:62-66.  `input_range` is ignored exactly as the reference ignores it (slip.py:64)."""
        if not apply_preprocess:
            raise NotImplementedError("apply_preprocess=False is not supported by the fused path")
        if imgs.shape[0] > self.handle.max_batch:
            raise ValueError(f"batch {imgs.shape[0]} exceeds the perceptor capacity {self.handle.max_batch}")
        return ops.clip_encode_image(imgs, self.handle, self.group)

    def encode_text(self, text):
        raise NotImplementedError(
            "the CLIP text tower + BPE tokenizer (slip.py:68-70) are outside the hot path (SURVEY.md §8f-1); "
            "pass precomputed text embeddings as vector prompts")


def get_clip_perceptor(clip_model_name, device, params=None, max_batch=64, seed=0, group=None):
    """slip.py:173-186 equivalent for the ViT family; `params` is an OpenAI `visual.*` state dict (random-init
    weights of the real architecture are synthesised when none is given: no checkpoints exist offline)."""
    if clip_model_name not in CLIP_CONFIGS:
        raise KeyError(f"unknown / unsupported perceptor {clip_model_name!r} (supported: {sorted(CLIP_CONFIGS)})")
    cfg = CLIP_CONFIGS[clip_model_name]
    if params is None:
        params = synthetic_clip_vit_params(cfg, seed)
    return ClipVitPerceptor(cfg, params, device, max_batch=max_batch, group=group)
