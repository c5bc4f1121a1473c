"""Functional fp32 CPU restatement of torchvision's ResNet (v1.5) forward, driven by a torchvision-named state dict.

Follows the call site /root/reference/configs/dog_fe/fe_dogs_config.py:102-103 (`torchvision.models.resnet50`, `fc`
replaced by Linear(2048, 512)); input is ToTensor() output in [0,1] (fe_dogs_config.py:17-32), train-mode BatchNorm
with eps 1e-5, momentum 0.1 and unbiased running variance.  Independent of the product code: plain
torch.nn.functional ops on CPU."""
import torch
import torch.nn.functional as F

ARCH = {
    "resnet18": ("basic", [2, 2, 2, 2]),
    "resnet34": ("basic", [3, 4, 6, 3]),
    "resnet50": ("bottleneck", [3, 4, 6, 3]),
    "resnet101": ("bottleneck", [3, 4, 23, 3]),
    # test-only prefix of ResNet-50 (stem + all of layer1 + the first block of layers 2-4): the geometries the BN-input-free form of
    # conv3 + bn3 applies to, small enough for a 256-image CPU oracle step (tests/test_model_gpu.py)
    "resnet50_l1": ("bottleneck", [3, 1, 1, 1]),
    # test-only one-block-per-layer nets: shallow enough that two correct bf16 evaluations of a train step agree to ~1e-2 in the gradient
    # (a 16-block random-init net amplifies every rounding difference to ~25 %: tools/diag_bf16cond.py), so a wrong kernel shows
    "resnet14b": ("bottleneck", [1, 1, 1, 1]),
    "resnet10": ("basic", [1, 1, 1, 1]),
}


def init_state_dict(arch, emb_dim=512, seed=0):
    """Seeded torchvision-style initialisation (Kaiming-normal fan-out convs, BN γ=1 β=0, default Linear init)."""
    kind, layers = ARCH[arch]
    g = torch.Generator().manual_seed(seed)
    sd = {}

    def conv(name, cout, cin, k):
        std = (2.0 / (cout * k * k)) ** 0.5
        sd[name + ".weight"] = torch.randn(cout, cin, k, k, generator=g) * std

    def bn(name, c):
        sd[name + ".weight"] = torch.ones(c)
        sd[name + ".bias"] = torch.zeros(c)
        sd[name + ".running_mean"] = torch.zeros(c)
        sd[name + ".running_var"] = torch.ones(c)
        sd[name + ".num_batches_tracked"] = torch.zeros((), dtype=torch.long)

    conv("conv1", 64, 3, 7)
    bn("bn1", 64)
    exp = 1 if kind == "basic" else 4
    inpl = 64
    for li, (planes, nblk) in enumerate(zip([64, 128, 256, 512], layers)):
        for b in range(nblk):
            stride = 2 if (b == 0 and li > 0) else 1
            p = f"layer{li + 1}.{b}"
            if kind == "basic":
                conv(p + ".conv1", planes, inpl, 3); bn(p + ".bn1", planes)
                conv(p + ".conv2", planes, planes, 3); bn(p + ".bn2", planes)
            else:
                conv(p + ".conv1", planes, inpl, 1); bn(p + ".bn1", planes)
                conv(p + ".conv2", planes, planes, 3); bn(p + ".bn2", planes)
                conv(p + ".conv3", planes * 4, planes, 1); bn(p + ".bn3", planes * 4)
            if stride != 1 or inpl != planes * exp:
                conv(p + ".downsample.0", planes * exp, inpl, 1); bn(p + ".downsample.1", planes * exp)
            inpl = planes * exp
    bound = 1.0 / (inpl ** 0.5)
    sd["fc.weight"] = (torch.rand(emb_dim, inpl, generator=g) * 2 - 1) * bound
    sd["fc.bias"] = (torch.rand(emb_dim, generator=g) * 2 - 1) * bound
    return sd


def _bn(sd, name, x, train, new_stats):
    rm, rv = sd[name + ".running_mean"].clone(), sd[name + ".running_var"].clone()
    y = F.batch_norm(x, rm, rv, sd[name + ".weight"], sd[name + ".bias"], training=train, momentum=0.1, eps=1e-5)
    if train and new_stats is not None:
        new_stats[name + ".running_mean"] = rm
        new_stats[name + ".running_var"] = rv
    return y


def bf16_round(t):
    """straight-through bf16 rounding: models WHERE the bf16 path stores/feeds bf16 (not part of the reference)"""
    return t + (t.detach().bfloat16().to(t.dtype) - t.detach())


class _RoundBoth(torch.autograd.Function):
    """bf16 rounding of the value in forward AND of the incoming gradient in backward: a tensor the bf16 path keeps in bf16 in both
    directions (every activation it stores has a bf16 gradient tensor of the same shape)"""

    @staticmethod
    def forward(ctx, t):
        return t.bfloat16().to(t.dtype)

    @staticmethod
    def backward(ctx, g):
        return g.bfloat16().to(g.dtype)


class _RoundGrad(torch.autograd.Function):
    """identity in forward, bf16 rounding of the gradient in backward (a bf16 gradient tensor whose forward value is exact already)"""

    @staticmethod
    def forward(ctx, t):
        return t.view_as(t)

    @staticmethod
    def backward(ctx, g):
        return g.bfloat16().to(g.dtype)


def bf16_round_fb(t):
    """`quant` callable for forward(): bf16 storage points in BOTH directions — what the HIP bf16 path does: the gradients of raw conv
    outputs (BatchNorm-backward apply), of normalised activations (data gradients), of block outputs (the residual joins), of the stem's
    pooled output, of the pooled features and of the embedding are bf16 tensors; weight gradients stay fp32 (split-K fp32 slabs), so
    weights are rounded straight-through only.  Not part of the reference: test infrastructure for the bf16 gradient tolerances."""
    return _RoundBoth.apply(t)


bf16_round_fb.weights = bf16_round                      # weights: forward rounding, fp32 gradient
bf16_round_fb.grad_only = _RoundGrad.apply              # gradient-only storage points


def forward(sd, x, arch, train=True, new_stats=None, taps=None, quant=None):
    """sd: name → tensor (parameters may require grad).  Returns the embedding (N × emb_dim).
    quant=None is the reference arithmetic (fp32).  quant=bf16_round emulates the storage points of the bf16 path
    (inputs, weights, raw conv outputs, normalised activations, block outputs) for the bf16 deviation tests;
    quant=bf16_round_fb additionally rounds the GRADIENTS at the tensors the bf16 path stores in bf16 (see there)."""
    kind, layers = ARCH[arch]
    q = quant or (lambda t: t)
    qw = getattr(quant, "weights", q)                   # weights
    qg = getattr(quant, "grad_only", lambda t: t)       # gradient-only storage points

    def conv(inp, name, **kw):
        return q(F.conv2d(inp, qw(sd[name + ".weight"]), **kw))

    x = conv(qw(x), "conv1", stride=2, padding=3)
    x = q(F.relu(_bn(sd, "bn1", x, train, new_stats)))
    x = qg(F.max_pool2d(x, 3, 2, 1))
    if taps is not None:
        taps["stem"] = x
    for li, nblk in enumerate(layers):
        for b in range(nblk):
            p = f"layer{li + 1}.{b}"
            stride = 2 if (b == 0 and li > 0) else 1
            idn = x
            if kind == "basic":
                o = conv(x, p + ".conv1", stride=stride, padding=1)
                o = q(F.relu(_bn(sd, p + ".bn1", o, train, new_stats)))
                o = conv(o, p + ".conv2", padding=1)
                o = _bn(sd, p + ".bn2", o, train, new_stats)
            else:
                o = conv(x, p + ".conv1")
                o = q(F.relu(_bn(sd, p + ".bn1", o, train, new_stats)))
                o = conv(o, p + ".conv2", stride=stride, padding=1)
                o = q(F.relu(_bn(sd, p + ".bn2", o, train, new_stats)))
                o = conv(o, p + ".conv3")
                o = _bn(sd, p + ".bn3", o, train, new_stats)
            if (p + ".downsample.0.weight") in sd:
                idn = conv(x, p + ".downsample.0", stride=stride)
                idn = _bn(sd, p + ".downsample.1", idn, train, new_stats)
            x = q(F.relu(o + idn))
            if taps is not None:
                taps[p] = x
    x = q(x.mean(dim=(2, 3)))
    return qg(F.linear(x, qw(sd["fc.weight"]), sd["fc.bias"]))


def param_names(sd):
    return [k for k in sd if not (k.endswith("running_mean") or k.endswith("running_var") or k.endswith("num_batches_tracked"))]
