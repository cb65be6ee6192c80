"""`Prompt` (reference: /root/reference/pixray.py:268-280) on the fused HIP prompt-loss kernel.

Same constructor, buffers (`embed`, `weight`, `stop`) and `forward(input[cutn, D]) -> scalar`; the result
carries a `grad_fn`.  `denom` (optional) is the number of (cutout, embed) pairs the mean runs over when the
cutout batch is sharded across ranks (global cutn * n_embed); by default the local count, as in the
reference."""
import torch
from torch import nn

from . import ops


class Prompt(nn.Module):
    def __init__(self, embed, weight=1., stop=float('-inf')):
        super().__init__()
        self.register_buffer('embed', embed)
        self.register_buffer('weight', torch.as_tensor(weight))
        self.register_buffer('stop', torch.as_tensor(stop))
        self._w = float(torch.as_tensor(weight))      # host copies: no device sync per iteration
        self._s = float(torch.as_tensor(stop))
        self.denom = None

    def forward(self, input):
        return ops.prompt_loss(input, self.embed, self._w, self._s, self.denom)


def parse_prompt(prompt):
    """`text`, `text:weight` or `text:weight:stop` (reference: pixray.py:290-321)."""
    text, weight, stop = prompt, 1, float('-inf')
    nums = []
    while len(nums) < 2:
        vals = text.rsplit(':', 1)
        if len(vals) > 1 and _is_number(vals[1]):
            nums.append(float(vals[1]))
            text = vals[0]
        else:
            break
    if len(nums) == 1:
        weight = nums[0]
    elif len(nums) == 2:
        weight, stop = nums[1], nums[0]
    return text, weight, stop


def _is_number(s):
    try:
        float(s)
        return True
    except ValueError:
        return False
