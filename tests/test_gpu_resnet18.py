"""resnet18.ResNet18 (the torchvision ResNet-18 layout over the general convolution kernels; the reference holds no such
model and no ternary / INQ quantiser: parity unpinned).  With INQ layers before their first stage the network is a
plain fp32 ResNet-18, so loss and every gradient can be checked against the same network written with stock torch
operators; the quantised modes are checked for what the papers prescribe (ternary codes, frozen powers of two) and for
a training step that learns.  Inputs are 64 x 64 so that the file runs in seconds; the layer shapes at 224 x 224 are
covered kernel by kernel in test_gpu_conv_general.py, and the whole model at 224 x 224 (batch 4) by the first test
below and by the half-frozen INQ step."""
import argparse

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from quantized_cnn_amd import ternary as T
    from quantized_cnn_amd.resnet18 import ResNet18
    from quantized_cnn_amd.train import CapturedStep
DEV = "cuda"


def config(mode_q):
    return argparse.Namespace(mode="train", mode_q=mode_q, prim_init="uniform", num_level_conv=32, num_level_fc=32, l2_reg=1e-4)


def stock_forward(model, x):
    """The same network through torch's own operators (batch statistics, no running-average update)."""
    def bn(m, t):
        return F.batch_norm(t, None, None, m.weight, m.bias, True, 0.0, m.eps)
    out = F.max_pool2d(F.relu(bn(model.bn0, F.conv2d(x, model.conv0.weight, stride=2, padding=3))), 3, 2, 1)
    for layer in (model.layer1, model.layer2, model.layer3, model.layer4):
        for blk in layer:
            o = F.relu(bn(blk.bn1, F.conv2d(out, blk.conv1.weight, stride=blk.conv1.stride, padding=1)))
            o = bn(blk.bn2, F.conv2d(o, blk.conv2.weight, stride=1, padding=1))
            sc = out if blk.down is None else bn(blk.bn_down, F.conv2d(out, blk.down.weight, stride=blk.down.stride))
            out = F.relu(o + sc)
    return model.fc(torch.flatten(F.adaptive_avg_pool2d(out, 1), 1))


def _float64_reference(model, x, y, weights=None):
    """Loss and gradients of the same network through torch's own operators in float64 on the host (weights: optional
    {conv module name: tensor} replacing convolution weights, e.g. an INQ layer's effective weights)."""
    import copy
    ref = copy.deepcopy(model).cpu().double()
    if weights:
        mods = dict(ref.named_modules())
        for n, w in weights.items():
            mods[n].weight.data.copy_(w.detach().cpu().double())
    ref.zero_grad(set_to_none=True)
    loss = F.cross_entropy(stock_forward(ref, x.cpu().double()), y.cpu())
    loss.backward()
    return float(loss.detach()), {n: p.grad for n, p in ref.named_parameters()}


def _near_float64(got, want, name):
    """Gradient of one tensor against the float64 evaluation at 224 x 224: relative L2 error <= 2e-2 and largest
    deviation <= 0.1 of the largest entry.  This is as tight as any fp32 evaluation of this network can be held: with
    ~2e7 activations some ReLU / max-pool inputs lie within rounding of a decision boundary, a decision taken the other
    way in a 14 x 14 or 7 x 7 layer moves a per-channel sum over 4 * 196 terms by 1e-3 .. 1e-2 of itself, and every layer
    upstream inherits it.  Measured against float64 (batch 4 and 16): this library L2 3.8e-3 median / 6.0e-3 worst,
    largest deviation 4.7e-2 .. 6.8e-2; torch's own fp32 operators on the same GPU 2.0e-3 / 3.6e-3 and 4.2e-2.  A wiring
    error (a wrong tap, stride, halo or shortcut at one of the 224-sized shapes) shows as O(1)."""
    d = got.detach().cpu().double() - want
    l2 = float(d.norm()) / max(float(want.norm()), 1e-300)
    mx = float(d.abs().max()) / max(float(want.abs().max()), 1e-300)
    assert l2 <= 2e-2 and mx <= 0.1, (name, l2, mx)


@pytest.mark.parametrize("size", [64, 224])
def test_unfrozen_inq_resnet18_equals_the_stock_network(size):
    """size 224: the input size BASELINE's INQ ResNet-18 configuration names (every layer at its stated shape, batch 4).
    Checked against the stock operators on the GPU (fp32, other summation order) and, at 224, against a float64
    evaluation on the host, which is what both fp32 results approximate."""
    torch.manual_seed(0)
    model = ResNet18(config("inq"), num_classes=10).to(DEV).train()
    x = torch.randn(4, 3, size, size, device=DEV)
    y = torch.randint(0, 10, (4,), device=DEV)
    loss = F.cross_entropy(model(x), y)
    loss.backward()
    got = {n: p.grad.clone() for n, p in model.named_parameters()}
    model.zero_grad(set_to_none=True)
    if size == 224:
        l64, g64 = _float64_reference(model, x, y)
        assert abs(float(loss.detach()) - l64) <= 1e-5 * abs(l64)
        for n, p in model.named_parameters():
            _near_float64(got[n], g64[n], n)
        return
    ref = F.cross_entropy(stock_forward(model, x), y)
    ref.backward()
    assert abs(float(loss.detach()) - float(ref.detach())) <= 1e-5 * abs(float(ref.detach()))
    for n, p in model.named_parameters():
        assert p.grad is not None and got[n] is not None, n
        err = float((got[n] - p.grad).abs().max())
        assert err <= 2e-3 * float(p.grad.abs().max()) + 1e-7, (n, err, float(p.grad.abs().max()))  # (17 layers of fp32 in another summation order)


@pytest.mark.parametrize("mode_q", ["adaptive", "twn", "ttq", "inq"])
def test_resnet18_training_step_learns_in_every_quantiser_mode(mode_q):
    torch.manual_seed(1)
    model = ResNet18(config(mode_q), num_classes=10).to(DEV).train()
    if mode_q == "inq":
        for m in model.modules():
            if isinstance(m, T.INQConv2d):
                m.quantize_portion(0.5)
    opt = torch.optim.SGD(model.parameters(), lr=1e-2, momentum=0.9)
    x = torch.randn(8, 3, 64, 64, device=DEV)
    y = torch.randint(0, 10, (8,), device=DEV)
    losses = []
    for _ in range(6):
        opt.zero_grad(set_to_none=True)
        loss = F.cross_entropy(model(x), y)
        loss.backward()
        assert all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for p in model.parameters()), mode_q
        opt.step()
        losses.append(float(loss.detach()))
    assert all(l == l for l in losses) and losses[-1] < losses[0], (mode_q, losses)
    if mode_q in ("twn", "ttq"):  # every layer computes with three values
        for m in model.modules():
            if isinstance(m, T.TernaryConv2d):
                assert m.quantize()["Wt"].unique().numel() <= 3
    if mode_q == "inq":  # frozen weights are powers of two (or zero) and receive no gradient
        for m in model.modules():
            if isinstance(m, T.INQConv2d):
                fz = m.effective_weight().detach()[m.mask == 0].abs()
                nz = fz[fz > 0]
                assert torch.equal(torch.log2(nz), torch.log2(nz).round())
                assert float(m.weight.grad[m.mask == 0].abs().max()) == 0.0


def test_half_frozen_inq_resnet18_at_224_equals_the_stock_network_on_its_effective_weights():
    """INQ at its first stage (half of every layer frozen at powers of two), 224 x 224, batch 4: loss and the gradients
    of the trainable coordinates equal those of the stock network evaluated on the effective weights; frozen
    coordinates receive exactly zero."""
    torch.manual_seed(3)
    model = ResNet18(config("inq"), num_classes=10).to(DEV).train()
    convs = [(n, m) for n, m in model.named_modules() if isinstance(m, T.INQConv2d)]
    assert len(convs) == 19
    for _, m in convs:
        m.quantize_portion(0.5)
    x = torch.randn(4, 3, 224, 224, device=DEV)
    y = torch.randint(0, 10, (4,), device=DEV)
    loss = F.cross_entropy(model(x), y)
    loss.backward()
    got = {n: p.grad.clone() for n, p in model.named_parameters()}
    l64, g64 = _float64_reference(model, x, y, {n: m.effective_weight() for n, m in convs})
    assert abs(float(loss.detach()) - l64) <= 1e-5 * abs(l64)
    masks = {n + ".weight": m.mask.bool().cpu() for n, m in convs}
    for n, p in model.named_parameters():
        want = g64[n] if n not in masks else torch.where(masks[n], g64[n], torch.zeros_like(g64[n]))
        _near_float64(got[n], want, n)
        if n in masks:
            assert float(got[n].cpu()[~masks[n]].abs().max()) == 0.0, n


def test_resnet18_step_as_one_graph_equals_eager_steps():
    def fresh():
        torch.manual_seed(2)
        model = ResNet18(config("ttq"), num_classes=10).to(DEV).train()
        return model, torch.optim.SGD(model.parameters(), lr=1e-2, momentum=0.9)
    x = torch.randn(4, 3, 64, 64, device=DEV)
    y = torch.randint(0, 10, (4,), device=DEV)
    m1, o1 = fresh()
    eager = []
    for _ in range(4):
        o1.zero_grad(set_to_none=True)
        loss = F.cross_entropy(m1(x), y)
        loss.backward()
        o1.step()
        eager.append(float(loss.detach()))
    m2, o2 = fresh()
    step = CapturedStep(m2, o2, F.cross_entropy, warmup=1)
    graph = [float(step(x, y)) for _ in range(3)]
    assert graph == eager[1:]


@pytest.mark.parametrize("shape", [(2, 3, 8, 8), (3, 5, 7, 9), (1, 2, 1, 1), (2, 4, 2, 3), (4, 64, 56, 56), (1, 1, 113, 31)])
def test_max_pool_kernels_equal_torch(shape):
    """aq_maxpool3s2_fwd / _bwd against nn.MaxPool2d(3, 2, 1): values bit-equal; gradients equal (an input pixel sums at most
    four window gradients); ties (quantised inputs) route the gradient to the first maximum in scan order as torch does."""
    import quantized_cnn_amd.ops as ops
    from quantized_cnn_amd.resnet18 import max_pool_3s2
    g = torch.Generator(device=DEV).manual_seed(sum(shape))
    for ties in (False, True):
        x = torch.randn(shape, device=DEV, generator=g)
        if ties:
            x = (x * 2).round() / 2
        pool = torch.nn.MaxPool2d(3, 2, 1)
        xr = x.clone().requires_grad_(True)
        yr = pool(xr)
        dy = torch.randn(yr.shape, device=DEV, generator=g)
        yr.backward(dy)
        xg = x.clone().requires_grad_(True)
        y = max_pool_3s2(pool, xg)
        assert torch.equal(y, yr)
        y.backward(dy)
        assert float((xg.grad - xr.grad).abs().max()) <= 1e-6 * max(float(xr.grad.abs().max()), 1e-6)
        y2, idx = ops.maxpool3s2_forward(x)
        assert int(idx.max()) <= 8 and torch.equal(y2, yr.detach())
    xn = torch.randn(1, 1, 6, 6, device=DEV)
    xn[0, 0, 2, 2] = float("nan")
    yn, yt = ops.maxpool3s2_forward(xn)[0], torch.nn.functional.max_pool2d(xn, 3, 2, 1)
    assert torch.equal(torch.isnan(yn), torch.isnan(yt)) and torch.equal(yn[~torch.isnan(yn)], yt[~torch.isnan(yt)])
    assert max_pool_3s2(torch.nn.MaxPool2d(2, 2), xn[:, :, :4, :4].nan_to_num()).shape == (1, 1, 2, 2)  # other pools: stock operator


@pytest.mark.parametrize("shape", [(4, 64, 56, 56), (3, 5, 7, 9), (2, 16, 8, 8), (8, 128, 4, 4)])
def test_bn_add_relu_equals_stock_operators(shape):
    """layers.bn_add_relu, the tail of a post-activation block as one kernel each way, against relu(batch_norm(x) + add) on
    torch's operators: output, running statistics, and the gradients of x, add, weight and bias."""
    import quantized_cnn_amd.layers as layers
    g = torch.Generator(device=DEV).manual_seed(sum(shape))
    x = torch.randn(shape, device=DEV, generator=g)
    add = torch.randn(shape, device=DEV, generator=g)
    dy = torch.randn(shape, device=DEV, generator=g)
    outs = []
    for fused in (True, False):
        bn = torch.nn.BatchNorm2d(shape[1]).to(DEV).train()
        with torch.no_grad():
            bn.weight.copy_(torch.linspace(0.5, 1.5, shape[1]))
            bn.bias.copy_(torch.linspace(-0.3, 0.3, shape[1]))
        xa, aa = x.clone().requires_grad_(True), add.clone().requires_grad_(True)
        y = layers.bn_add_relu(bn, xa, aa) if fused else F.relu(bn(xa) + aa)
        y.backward(dy)
        outs.append((y.detach(), xa.grad, aa.grad, bn.weight.grad, bn.bias.grad, bn.running_mean.clone(), bn.running_var.clone()))
    names = ("y", "dx", "dadd", "dweight", "dbias", "running_mean", "running_var")
    for n, a, b in zip(names, outs[0], outs[1]):
        assert float((a - b).abs().max()) <= 2e-5 * max(float(b.abs().max()), 1e-3), (n, float((a - b).abs().max()), float(b.abs().max()))
    assert torch.equal(outs[0][2] != 0, outs[1][2] != 0) or float(((outs[0][2] != 0) != (outs[1][2] != 0)).float().mean()) < 1e-5
