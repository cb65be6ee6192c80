"""Perceptor wrapper with the reference's `CLIP_Base` surface (/root/reference/slip.py:44-74): attributes
`input_resolution`, `output_dim`; `encode_image(imgs) -> [N, D]` L2-normalised and differentiable; frozen
weights; `encode_text(text) -> [n, D]` (raw projection, as slip.py:68-70); `encode_texts`.  The arithmetic is the HIP
CLIP ViT runner (pixray_amd/csrc/vit.hip) and the HIP text tower (pixray_amd/csrc/clip_text.hip, built on first use)."""
import torch

from . import ops
from .weights import (CLIP_CONFIGS, CLIP_RESNET_CONFIGS, CLIP_TEXT_CONFIGS, ClipResNetConfig, synthetic_clip_resnet_params, ClipTextConfig, ClipVitConfig, clip_text_param_shapes,
                      synthetic_clip_text_params, synthetic_clip_vit_params)


class ClipVitPerceptor:
    def __init__(self, cfg: ClipVitConfig, params, device, max_batch: int = 64, group=None, text_cfg: ClipTextConfig = None,
                 text_params=None, tokenizer=None, seed: int = 0, precision=None):
        self.cfg = cfg
        self.text_cfg = text_cfg
        self.text_params = text_params      # OpenAI state-dict entries of the text side (token_embedding.weight, ...)
        self.tokenizer = tokenizer          # a pixray_amd.tokenizer.BpeTokenizer; default: the process-wide one
        self._seed = seed
        self._text_handle = None
        self.device = torch.device(device)
        self.input_resolution = cfg.input_resolution
        self.output_dim = cfg.output_dim
        self.group = group          # torch.distributed group when the cutout batch is sharded
        if isinstance(cfg, ClipResNetConfig):       # ModifiedResNet family (RN50x4, ...): same protocol, different runner
            self.handle = ops.ClipResNetHandle(cfg, params, max_batch, self.device, precision=precision)
        else:
            self.handle = ops.ClipVitHandle(cfg, params, max_batch, self.device, precision=precision)

    CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)      # slip.py:55
    CLIP_STD = (0.26862954, 0.26130258, 0.27577711)

    def preprocess(self, imgs, input_range=None):
        """slip.py:58-60: `adjust_range(imgs, [0, 1], input_range)` (slip.py:21-42: the batch's own min / max when no range is
        given), torchvision `Resize(R)` (shorter side to R, bilinear, tensors are not antialiased) + `CenterCrop(R)` +
        `Normalize(mean, std)`.  Plain differentiable tensor arithmetic on the caller's device: it is not on the loop's path
        (the loop's `encode_image` fuses range + normalisation into the tower's first kernel); it exists so that a caller who
        preprocesses once and encodes with `apply_preprocess=False`, as the reference allows, finds the same two methods."""
        import torch.nn.functional as F
        imgs = imgs.float()
        lo = imgs.min() if input_range is None else torch.as_tensor(float(input_range[0]), device=imgs.device)
        x = imgs - lo
        hi = x.max() if input_range is None else torch.as_tensor(float(input_range[1]), device=imgs.device) - lo
        x = torch.where(hi != 0, x / torch.where(hi != 0, hi, torch.ones_like(hi)), x)       # slip.py:32-33: divided only when the span is not 0
        R = self.input_resolution
        h, w = x.shape[-2:]
        if min(h, w) != R:
            nh, nw = (R, max(R, int(R * w / h))) if h <= w else (max(R, int(R * h / w)), R)
            x = F.interpolate(x, size=(nh, nw), mode="bilinear", align_corners=False, antialias=False)
            h, w = nh, nw
        top, left = int(round((h - R) / 2.0)), int(round((w - R) / 2.0))
        x = x[..., top:top + R, left:left + R]
        mean = torch.tensor(self.CLIP_MEAN, device=x.device).view(1, 3, 1, 1)
        std = torch.tensor(self.CLIP_STD, device=x.device).view(1, 3, 1, 1)
        return (x - mean) / std

    def encode_image(self, imgs, input_range=None, apply_preprocess=True):
        """slip.py:62-66.  `input_range` is ignored exactly as the reference ignores it (slip.py:64).  With
        `apply_preprocess=False` the images are taken as already preprocessed ([n, 3, R, R], normalised): the tower's first kernel
        fuses `(x - lo) / (hi - lo)` and the channel normalisation, so they are handed to it un-normalised with the fixed range
        (0, 1) -- the same values enter the patch embedding, and the gradient flows back through both affine maps."""
        if imgs.shape[0] > self.handle.max_batch:
            raise ValueError(f"batch {imgs.shape[0]} exceeds the perceptor capacity {self.handle.max_batch}")
        if not apply_preprocess:
            mean = torch.tensor(self.CLIP_MEAN, device=imgs.device).view(1, 3, 1, 1)
            std = torch.tensor(self.CLIP_STD, device=imgs.device).view(1, 3, 1, 1)
            return ops.clip_encode_image(imgs.float() * std + mean, self.handle, fixed_range=(0.0, 1.0))
        return ops.clip_encode_image(imgs, self.handle, self.group, getattr(self, "comm", None))

    # -- text side (slip.py:68-74) ---------------------------------------------------------------------------------
    def _text(self):
        if self._text_handle is None:
            if self.text_cfg is None:
                raise NotImplementedError(f"no text tower configuration for perceptor {self.cfg.name!r}")
            if self.text_cfg.output_dim != self.cfg.output_dim:
                raise ValueError("text and image towers must share the embedding size")
            params = self.text_params
            if params is None:                  # no checkpoints offline: seeded random weights of the real architecture
                params = synthetic_clip_text_params(self.text_cfg, self._seed)
            missing = [k for k in clip_text_param_shapes(self.text_cfg) if k not in params]
            if missing:
                raise KeyError(f"text tower state dict is missing {missing[0]} (+{len(missing) - 1} more)")
            self._text_handle = ops.ClipTextHandle(self.text_cfg, params, max_batch=16, device=self.device)
        return self._text_handle

    def tokenize(self, text):
        from .tokenizer import default_tokenizer
        tok = self.tokenizer if self.tokenizer is not None else default_tokenizer()
        if tok.vocab_size != self.text_cfg.vocab_size:
            raise ValueError(f"tokenizer has {tok.vocab_size} ids, the text tower expects {self.text_cfg.vocab_size}")
        return tok.tokenize(text, self.text_cfg.context_length)

    def encode_text(self, text):
        """slip.py:68-70: `clip.tokenize(text)` -> `model.encode_text` -> `.float()`; `text` may also be an integer tensor
        of token ids [n, context_length] (what `clip.tokenize` returns; pixray.py:868-870 passes those)."""
        h = self._text()
        tokens = text if torch.is_tensor(text) else self.tokenize(text)
        outs = [ops.clip_encode_text_tokens(tokens[i:i + h.max_batch], h) for i in range(0, tokens.shape[0], h.max_batch)]
        return torch.cat(outs, 0).float()

    def encode_texts(self, texts):
        """slip.py:72-74"""
        e = torch.stack([self.encode_text(t).detach().clone() for t in texts])
        return e / e.norm(dim=-1, keepdim=True)


def get_clip_perceptor(clip_model_name, device, params=None, max_batch=64, seed=0, group=None, text_params=None,
                       tokenizer=None, precision=None):
    """slip.py:173-186 equivalent for the ViT family; `params` is an OpenAI `visual.*` state dict and `text_params` the
    text-side entries of the same checkpoint (random-init weights of the real architectures are synthesised when none
    are given: no checkpoints exist offline).  `precision`: "fp16" (default) | "bf16" (fast paths) | "f32" (exact-f32 MFMA parity mode)."""
    if clip_model_name in CLIP_RESNET_CONFIGS:
        cfg = CLIP_RESNET_CONFIGS[clip_model_name]
        if params is None:
            params = synthetic_clip_resnet_params(cfg, seed)
        return ClipVitPerceptor(cfg, params, device, max_batch=max_batch, group=group, text_cfg=CLIP_TEXT_CONFIGS.get(clip_model_name),
                                text_params=text_params, tokenizer=tokenizer, seed=seed, precision=precision)
    if clip_model_name not in CLIP_CONFIGS:
        raise KeyError(f"unknown / unsupported perceptor {clip_model_name!r} "
                       f"(supported: {sorted(CLIP_CONFIGS) + sorted(CLIP_RESNET_CONFIGS)})")
    cfg = CLIP_CONFIGS[clip_model_name]
    if params is None:
        params = synthetic_clip_vit_params(cfg, seed)
    return ClipVitPerceptor(cfg, params, device, max_batch=max_batch, group=group, text_cfg=CLIP_TEXT_CONFIGS.get(clip_model_name),
                            text_params=text_params, tokenizer=tokenizer, seed=seed, precision=precision)
