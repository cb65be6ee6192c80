"""CPU fp32 restatement of the StyleLoss plugin's VGG16 extractor -- TEST INFRASTRUCTURE ONLY (never imported by
pixray_amd/).

Reference: `Vgg16_Extractor` (/root/reference/Losses/StyleLoss.py:24-47): `torchvision.models.vgg16(pretrained=True)
.features`, frozen; `forward_base` walks the Sequential and keeps the input plus the outputs of layers
[1,3,6,8,11,13,15,22,29] (ReLU outputs relu1_1 .. relu5_3; the ReLUs are in-place); `forward` first maps a [-1,1] image to
ImageNet-normalised space unless space == 'vgg'.  torchvision is not installed here and no VGG16 checkpoint exists offline,
so the layer list is restated from torchvision's published cfg "D" (conv3x3 pad 1 + ReLU, 2x2/2 max-pool after blocks of
2,2,3,3,3 convs).  Pins: (1) tests/test_oracle_pins.py::test_vgg_oracle_vs_the_reference_extractor_class_run_live runs the
reference's OWN class (AST-extracted, executed) around an nn.Sequential of that layout and requires the same ten maps;
(2) tests/golden/styleloss_golden.npz (STROTSS value + image gradient through the reference's class and functions) is
reproduced on this extractor (tests/test_style_loss.py).  What stays from knowledge: that torchvision's `vgg16().features`
IS cfg "D" with these state-dict keys (`features.{0,2,5,...}.weight/bias`; tests/test_style_loss.py::
test_torchvision_vgg16_checkpoint_adapter covers the key layout the loader expects)."""
from typing import Dict, List

import torch
import torch.nn.functional as F

CFG = (64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512)
CAPTURE = (1, 3, 6, 8, 11, 13, 15, 22, 29)
MEAN = (0.485, 0.456, 0.406)
STD = (0.229, 0.224, 0.225)


def normalise(x: torch.Tensor, space: str = "uniform") -> torch.Tensor:
    """StyleLoss.py:41-45"""
    if space != "vgg":
        x = (x + 1.0) / 2.0
        x = x - torch.tensor(MEAN, dtype=x.dtype, device=x.device).view(1, -1, 1, 1)
        x = x / torch.tensor(STD, dtype=x.dtype, device=x.device).view(1, -1, 1, 1)
    return x


def forward_base(p: Dict[str, torch.Tensor], x: torch.Tensor) -> List[torch.Tensor]:
    """StyleLoss.py:34-39: [x, relu1_1, relu1_2, relu2_1, relu2_2, relu3_1, relu3_2, relu3_3, relu4_3, relu5_3] (NCHW)"""
    feat = [x]
    idx = 0
    for v in CFG:
        if v == "M":
            x = F.max_pool2d(x, 2, 2)
            idx += 1
            continue
        x = F.relu(F.conv2d(x, p[f"features.{idx}.weight"], p[f"features.{idx}.bias"], padding=1))
        if idx + 1 in CAPTURE:
            feat.append(x)
        idx += 2
    return feat


def forward(p, x, space: str = "uniform"):
    return forward_base(p, normalise(x, space))
