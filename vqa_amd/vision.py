"""The reference's convolutional / Swin vision wrappers (model/vision_encoder.py:8-127) with their constructor
arguments, module names (so the reference's checkpoints load key for key) and the `(v_seq (B,N,out), v_global
(B,out))` contract that the fusion modules consume.

These encoders sit in front of the fusion hot path and are library code on this GPU as they are in the
reference: convolutions, batch norm and pooling run through torch (MIOpen), the Swin backbone is the
`transformers` SwinModel.  What is specific here: the backbones are built from their architecture alone, because a
pretrained checkpoint cannot be fetched by name without a network -- `weights=` takes a local state dict file
(`.safetensors`, or a `.pt`/`.pth` read with `weights_only=True`), a Swin `model_name` that is a local directory is
loaded from it, anything else starts from a random initialisation -- and the convolutions run in channels-last
memory format on the GPU, the layout MIOpen's implicit-GEMM kernels want on this hardware.
"""
from __future__ import annotations

import os

import torch
import torch.nn as nn


def _read_local_state(path: str) -> dict:
    """A state dict from a local file, read without executing anything the file holds."""
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file
        return load_file(path)
    return torch.load(path, map_location="cpu", weights_only=True)


def _channels_last(images: torch.Tensor) -> torch.Tensor:
    return images.contiguous(memory_format=torch.channels_last) if images.is_cuda else images


class CNN(nn.Module):
    """Three conv3x3-BN-ReLU-maxpool stages; the 28 x 28 map of a 224 x 224 image is the token sequence
    (784 tokens of 512 channels), its average through a Linear the global feature (model/vision_encoder.py:8-44)."""

    def __init__(self, out_features=512):
        super().__init__()

        def stage(cin, cout):
            return nn.Sequential(nn.Conv2d(cin, cout, kernel_size=3, stride=1, padding=1), nn.BatchNorm2d(cout),
                                 nn.ReLU(), nn.MaxPool2d(2))
        self.layer1 = stage(3, 64)
        self.layer2 = stage(64, 128)
        self.layer3 = stage(128, 512)
        self.pool = nn.AdaptiveAvgPool2d((1, 1))
        self.flatten = nn.Flatten()
        self.fc = nn.Linear(512, out_features)
        self.output_dim = out_features

    def forward(self, images):
        fmap = self.layer3(self.layer2(self.layer1(_channels_last(images))))  # (B, 512, H/8, W/8)
        if fmap.is_contiguous(memory_format=torch.channels_last):
            # channels-last storage IS the (B, H*W, C) token layout: contiguous tokens without a copy
            v_seq = fmap.permute(0, 2, 3, 1).reshape(fmap.shape[0], -1, fmap.shape[1])
        else:
            v_seq = fmap.flatten(2).permute(0, 2, 1)
        v_global = self.fc(self.flatten(self.pool(fmap)))
        return v_seq, v_global


class _Bottleneck(nn.Module):
    """The 1x1 - 3x3 - 1x1 residual block of ResNet-50 (He et al. 2015), stride on the 3x3 convolution, with the
    attribute names of the torchvision implementation that the reference instantiates."""
    expansion = 4

    def __init__(self, cin, width, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, width, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(width)
        self.conv2 = nn.Conv2d(width, width, 3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(width)
        self.conv3 = nn.Conv2d(width, width * self.expansion, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(width * self.expansion)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.relu(self.bn2(self.conv2(y)))
        y = self.bn3(self.conv3(y))
        return self.relu(y + (x if self.downsample is None else self.downsample(x)))


def _resnet50_features() -> nn.Sequential:
    """conv1, bn1, relu, maxpool, layer1..layer4 of ResNet-50 -- the children the reference keeps
    (`list(resnet.children())[:-2]`, model/vision_encoder.py:53-55), so the keys are `0.weight`, `1.*`, `4.0.conv1.weight` ..."""
    def layer(cin, width, blocks, stride):
        down = nn.Sequential(nn.Conv2d(cin, width * 4, 1, stride=stride, bias=False), nn.BatchNorm2d(width * 4))
        mods = [_Bottleneck(cin, width, stride, down)]
        mods += [_Bottleneck(width * 4, width) for _ in range(blocks - 1)]
        return nn.Sequential(*mods)
    net = nn.Sequential(nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False), nn.BatchNorm2d(64), nn.ReLU(inplace=True),
                        nn.MaxPool2d(3, stride=2, padding=1),
                        layer(64, 64, 3, 1), layer(256, 128, 4, 2), layer(512, 256, 6, 2), layer(1024, 512, 3, 2))
    for m in net.modules():  # the initialisation of the architecture's authors
        if isinstance(m, nn.Conv2d):
            nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
        elif isinstance(m, nn.BatchNorm2d):
            nn.init.ones_(m.weight)
            nn.init.zeros_(m.bias)
    return net


class ResNet50(nn.Module):
    """ResNet-50 up to its last 7 x 7 x 2048 map; a 1x1 convolution makes the 49 tokens, the pooled map through a
    Linear the global feature (model/vision_encoder.py:46-80).  `weights`: a local file with the backbone's state
    dict (torchvision key names, e.g. `conv1.weight`, `layer1.0.bn1.weight`); without it the backbone is randomly
    initialised -- the reference downloads ImageNet weights, which needs a network."""

    def __init__(self, out_features=512, weights: str | None = None):
        super().__init__()
        self.resnet_features = _resnet50_features()
        if weights is not None:
            self.load_backbone(weights)
        resnet_output_dim = 2048
        self.seq_projection = nn.Conv2d(resnet_output_dim, out_features, kernel_size=1)
        self.pool = nn.AdaptiveAvgPool2d((1, 1))
        self.flatten = nn.Flatten()
        self.global_projection = nn.Linear(resnet_output_dim, out_features)
        self.output_dim = out_features

    # the child names of a torchvision ResNet, in the order of the Sequential that holds them here
    _CHILDREN = ("conv1", "bn1", "relu", "maxpool", "layer1", "layer2", "layer3", "layer4")

    def load_backbone(self, path: str) -> None:
        state = _read_local_state(path)
        index = {n: str(i) for i, n in enumerate(self._CHILDREN)}
        renamed = {}
        for k, v in state.items():
            head, _, rest = k.partition(".")
            if head in index:  # (the classifier head `fc.*` of the full network is not part of the features)
                renamed[index[head] + "." + rest] = v
        self.resnet_features.load_state_dict(renamed, strict=True)

    def forward(self, images):
        fmap = self.resnet_features(_channels_last(images))  # (B, 2048, 7, 7)
        v_global = self.global_projection(self.flatten(self.pool(fmap)))
        v_seq = self.seq_projection(fmap).flatten(2).permute(0, 2, 1)  # (B, 49, out)
        return v_seq, v_global


class SwinTransformer(nn.Module):
    """Swin-T: the 49 final-stage tokens and the pooled output, each through its own Linear
    (model/vision_encoder.py:83-127).  `model_name` that is a local directory is loaded from it
    (`SwinModel.from_pretrained`); a hub name cannot be fetched offline, so the backbone is then built from
    `config` (default: the Swin-T configuration that name denotes -- patch 4, window 7, widths 96..768, depths
    2-2-6-2) with a random initialisation."""

    def __init__(self, model_name='microsoft/swin-tiny-patch4-window7-224', out_features=512, config=None):
        super().__init__()
        from transformers import SwinConfig, SwinModel
        if config is None and os.path.isdir(model_name):
            self.swin = SwinModel.from_pretrained(model_name, local_files_only=True)
        else:
            self.swin = SwinModel(config if config is not None else SwinConfig())
        swin_output_dim = self.swin.config.hidden_size
        self.seq_projection = nn.Linear(swin_output_dim, out_features)
        self.global_projection = nn.Linear(swin_output_dim, out_features)
        self.output_dim = out_features

    def forward(self, images):
        outputs = self.swin(pixel_values=images)
        return self.seq_projection(outputs.last_hidden_state), self.global_projection(outputs.pooler_output)


def get_vision_registry() -> dict:
    """The `Vision:` names of the reference's configuration files (utils/util.py:50-54)."""
    return {"CNN": CNN, "ResNet50": ResNet50, "SwinTransformer": SwinTransformer}
