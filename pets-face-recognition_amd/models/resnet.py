"""ResNet feature-extractor backbones with torchvision-compatible structure and state-dict names.

The reference builds its backbone with `torchvision.models.resnet50(pretrained=True)` and replaces `fc` by
`Linear(2048, 512)` (/root/reference/configs/dog_fe/fe_dogs_config.py:102-103).  torchvision is third-party (and not
installed); this module restates the ResNet v1.5 definition (stride on the 3x3, bias-free convs, BN eps 1e-5 /
momentum 0.1, Kaiming-normal fan-out init, zero_init_residual=False) so that `state_dict()` carries exactly the
torchvision keys (`conv1.weight`, `layer1.0.downsample.0.weight`, `bn1.num_batches_tracked`, `fc.bias`, …).

Execution:
  * CUDA (HIP) tensors → the hand-written gfx950 kernels via `models._fe_engine.FEEngine` (no torch math;
    raises if libpfr_hip.so is missing);
  * CPU tensors → the plain `torch.nn` layers this module is made of (the reference's own CPU behaviour; used by
    BASELINE config 1 "PyTorch CPU via main.py" and never selected implicitly for a CUDA input).
"""
import torch
import torch.nn as nn


def _conv(cin, cout, k, stride=1, pad=0):
    return nn.Conv2d(cin, cout, k, stride=stride, padding=pad, bias=False)


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = _conv(inplanes, planes, 3, stride, 1)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = _conv(planes, planes, 3, 1, 1)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        identity = x
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        if self.downsample is not None:
            identity = self.downsample(x)
        return self.relu(out + identity)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = _conv(inplanes, planes, 1)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = _conv(planes, planes, 3, stride, 1)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = _conv(planes, planes * 4, 1)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        identity = x
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        if self.downsample is not None:
            identity = self.downsample(x)
        return self.relu(out + identity)


class ResNet(nn.Module):
    def __init__(self, block, layers, num_classes=1000, compute_dtype=None):
        super().__init__()
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2)
        self.layer3 = self._make_layer(block, 256, layers[2], stride=2)
        self.layer4 = self._make_layer(block, 512, layers[3], stride=2)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512 * block.expansion, num_classes)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        # compute dtype of the HIP path: torch.bfloat16 (throughput) or torch.float32 (parity); None → env / bf16
        self.compute_dtype = compute_dtype
        self._engine = None

    def _make_layer(self, block, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(_conv(self.inplanes, planes * block.expansion, 1, stride),
                                       nn.BatchNorm2d(planes * block.expansion))
        layers = [block(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self.inplanes, planes))
        return nn.Sequential(*layers)

    # ---- reference-equivalent CPU path (plain torch.nn layers)
    def _forward_torch(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        x = torch.flatten(self.avgpool(x), 1)
        return self.fc(x)

    # ---- HIP path
    def hip_engine(self, device=None):
        from ._fe_engine import FEEngine

        if self._engine is None or not self._engine.matches(self):
            self._engine = FEEngine(self, device or next(self.parameters()).device, self.compute_dtype)
        return self._engine

    def forward(self, x):
        if x.is_cuda:
            from ._fe_engine import fe_forward

            return fe_forward(self, x)
        return self._forward_torch(x)

    def _apply(self, fn, *a, **kw):
        # parameters may move (model.to(device)); the engine re-adopts them lazily
        self._engine = None
        return super()._apply(fn, *a, **kw)


def resnet18(num_classes=1000, pretrained=False, **kw):
    _no_pretrained(pretrained)
    return ResNet(BasicBlock, [2, 2, 2, 2], num_classes, **kw)


def resnet34(num_classes=1000, pretrained=False, **kw):
    _no_pretrained(pretrained)
    return ResNet(BasicBlock, [3, 4, 6, 3], num_classes, **kw)


def resnet50(num_classes=1000, pretrained=False, **kw):
    _no_pretrained(pretrained)
    return ResNet(Bottleneck, [3, 4, 6, 3], num_classes, **kw)


def resnet101(num_classes=1000, pretrained=False, **kw):
    _no_pretrained(pretrained)
    return ResNet(Bottleneck, [3, 4, 23, 3], num_classes, **kw)


def _no_pretrained(pretrained):
    if pretrained:
        import warnings

        warnings.warn("pretrained=True ignored: ImageNet weights are not available offline; load a reference-format "
                      "state_dict with load_state_dict() instead (keys are torchvision-compatible).")
