"""Ternary (TWN, TTQ) and incremental power-of-two (INQ) quantisers against their CPU restatement.

PARITY UNPINNED - no reference implementation: the reference repository cites these methods (README.md:3-29) and
implements none of them, so the checker (oracle/aq_oracle.c: aqo_tern_*, aqo_inq_*) is itself written from the
papers.  What the tests establish is that the HIP kernels compute what that restatement computes:
codes, masks and rounded weights bit for bit when both sides use the same threshold; thresholds and scales
(fp32 fixed-order sums against float64) at rel 2e-6 / 1e-5; masked gradient sums at 2e-5 of their scale.
"""
import argparse

import numpy as np
import pytest
import torch

from oracle import oracle as O

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    import quantized_cnn_amd.ops as ops
    import quantized_cnn_amd.ternary as T
DEV = "cuda"


def cu(a, dtype=torch.float32):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV, dtype)


@pytest.mark.parametrize("n", [1, 3, 5, 16, 17, 255, 2048, 2049, 36864, 1_000_003, 40_000_001])  # the last: second-level partials, streaming loads
@pytest.mark.parametrize("mode,from_max,factor", [("twn", False, 0.7), ("ttq", True, 0.05), ("ttq", False, 0.7)])
def test_ternary_forward_vs_oracle(n, mode, from_max, factor):
    rs = np.random.RandomState(n % 9973 + int(factor * 100))
    w = (rs.standard_normal(n) * 0.05).astype(np.float32)
    Wp, Wn = np.float32(0.11), np.float32(0.07)
    kw = dict(Wp=[cu([Wp])], Wn=[cu([Wn])]) if mode == "ttq" else {}
    out = ops.tern_forward([cu(w)], mode, factor, from_max, packed=True, **kw)[0]
    stats, q = out["stats"].cpu().numpy(), out["q"].cpu().numpy()
    ref = O.tern_forward(w, mode, factor, from_max, Wp, Wn)
    assert abs(stats[0] - ref["delta"]) <= 2e-6 * ref["delta"] + 1e-30
    # codes: bit-exact for the kernel's own threshold ...
    same = O.tern_codes(w, stats[0])
    codes = out["codes"].cpu().numpy()
    assert np.array_equal(codes, same["codes"])
    # ... and against the oracle's threshold wherever |w| is not within rounding of it
    far = np.abs(np.abs(w) - ref["delta"]) > 4e-6 * ref["delta"]
    assert np.array_equal(codes[far], ref["codes"][far])
    assert np.array_equal(out["packed"].cpu().numpy(), O.pack_codes(codes, 2))
    assert stats[3] == same["npos"] + same["nneg"]
    if mode == "ttq":
        assert stats[1] == Wp and stats[2] == Wn
    else:
        alpha = same["sum_abs"] / max(same["npos"] + same["nneg"], 1)
        assert abs(stats[1] - alpha) <= 1e-5 * alpha + 1e-30 and stats[1] == stats[2]
    assert q[0] == 0.0 and q[1] == stats[1] and q[2] == -stats[2]
    assert np.array_equal(out["Wt"].cpu().numpy(), q[codes])


def test_ternary_forward_many_layers_one_call_and_determinism():
    rs = np.random.RandomState(1)
    shapes = [(16, 16, 3, 3)] * 6 + [(32, 16, 3, 3)] + [(32, 32, 3, 3)] * 5 + [(64, 32, 3, 3)] + [(64, 64, 3, 3)] * 5 + [(64, 10)]
    shapes = shapes * 2  # 38 tensors: more than one call's worth of 32
    ws = [cu(rs.standard_normal(s) * (0.02 + 0.01 * i)) for i, s in enumerate(shapes)]
    a = ops.tern_forward(ws, "twn")
    b = ops.tern_forward(ws, "twn")
    for w, x, y in zip(ws, a, b):
        assert torch.equal(x["codes"], y["codes"]) and torch.equal(x["stats"], y["stats"]) and torch.equal(x["Wt"], y["Wt"])
        ref = O.tern_forward(w.cpu().numpy(), "twn")
        assert abs(float(x["stats"][0]) - ref["delta"]) <= 2e-6 * ref["delta"]
        assert x["codes"].shape == w.shape
        same = O.tern_codes(w.cpu().numpy(), x["stats"][0].item())
        assert np.array_equal(x["codes"].cpu().numpy(), same["codes"])
    # edge cases: all zeros (no code fires, alpha 0), constants
    z = ops.tern_forward([torch.zeros(100, device=DEV), torch.full((7,), 0.5, device=DEV)], "twn")
    assert int(z[0]["codes"].sum()) == 0 and float(z[0]["stats"][1]) == 0.0 and float(z[0]["stats"][3]) == 0.0
    assert int(z[1]["codes"].min()) == 1 and abs(float(z[1]["stats"][1]) - 0.5) < 1e-7  # 0.5 > 0.7 * 0.5
    with pytest.raises(ValueError):
        ops.tern_forward(ws[:1], "ttq")


@pytest.mark.parametrize("n", [1, 7, 2048, 70001, 40_000_001])
def test_ternary_backward_vs_oracle(n):
    rs = np.random.RandomState(n)
    w = (rs.standard_normal(n) * 0.05).astype(np.float32)
    g = rs.standard_normal(n).astype(np.float32)
    wd, gd = cu(w), cu(g)
    # TWN: straight through
    dW, _, _ = ops.tern_backward([wd], [gd], "twn")
    assert torch.equal(dW[0], gd)
    # TTQ
    Wp, Wn = np.float32(0.13), np.float32(0.09)
    fw = ops.tern_forward([wd], "ttq", 0.05, True, [cu([Wp])], [cu([Wn])])[0]
    delta = float(fw["stats"][0])
    dW, dp, dn = ops.tern_backward([wd], [gd], "ttq", stats=[fw["stats"]], Wp=[cu([Wp])], Wn=[cu([Wn])])
    dw_ref, dp_ref, dn_ref = O.tern_backward(w, g, "ttq", delta, Wp, Wn)
    assert np.array_equal(dW[0].cpu().numpy().view(np.uint32), dw_ref.view(np.uint32))
    scale = np.abs(g).sum()
    assert abs(float(dp[0]) - dp_ref) <= 2e-5 * scale and abs(float(dn[0]) - dn_ref) <= 2e-5 * scale
    # the same scale gradients from autograd of the forward formula
    wt_, gt_ = torch.from_numpy(w).double(), torch.from_numpy(g).double()
    p, q = torch.tensor(float(Wp), dtype=torch.double, requires_grad=True), torch.tensor(float(Wn), dtype=torch.double, requires_grad=True)
    wq = p * (wt_ > delta) - q * (wt_ < -delta)
    (wq * gt_).sum().backward()
    assert abs(dp_ref - float(p.grad)) <= 1e-9 * scale and abs(dn_ref - float(q.grad)) <= 1e-9 * scale
    # INQ mask
    mask = (rs.random_sample(n) < 0.4).astype(np.uint8)
    dW, _, _ = ops.tern_backward([wd], [gd], "inq", masks=[cu(mask, torch.uint8)])
    assert np.array_equal(dW[0].cpu().numpy(), np.where(mask != 0, g, np.float32(0.0)))


def test_inq_partition_vs_oracle_stage_by_stage():
    rs = np.random.RandomState(5)
    n = 50_001
    w = (rs.standard_normal(n) * 0.08).astype(np.float32)
    n1, n2 = O.inq_exponents(np.abs(w).max(), 5)
    assert n1 - n2 == 7
    # decision boundaries of the level rule, zero, a subnormal, values above 1.5 * 2^n1 and below 2^(n2-1)
    edge = []
    for k in range(n2 - 2, n1 + 2):
        for m in (0.5, 0.75, 1.0, 1.5):
            v = np.float32(m * 2.0 ** k)
            edge += [v, np.nextafter(v, np.float32(0)), np.nextafter(v, np.float32(4)), -v]
    edge += [0.0, -0.0, 1e-41, 3.0 * 2.0 ** n1]
    w[: len(edge)] = np.array(edge, np.float32)
    wd = cu(w)
    md = torch.ones(n, dtype=torch.uint8, device=DEV)
    cd = torch.full((n,), 7, dtype=torch.uint8, device=DEV)
    wo, mo = w.copy(), np.ones(n, np.uint8)
    for frac in (0.5, 0.75, 0.875, 1.0):
        free = np.abs(wo[mo == 1])
        k = int(round(frac * n)) - int((mo == 0).sum())
        tau = np.sort(free)[::-1][k - 1] if frac < 1.0 else 0.0
        ops.inq_partition(wd, md, tau, n1, n2, cd)
        wo, mo, co = O.inq_partition(wo, mo, tau, n1, n2)
        assert np.array_equal(wd.cpu().numpy().view(np.uint32), wo.view(np.uint32)), frac
        assert np.array_equal(md.cpu().numpy(), mo) and np.array_equal(cd.cpu().numpy(), co), frac
    assert mo.sum() == 0
    levels = np.unique(np.abs(wo))
    assert set(levels) <= {0.0} | {2.0 ** k for k in range(n2, n1 + 1)}
    assert co.max() <= 16 and len(np.unique(co)) == 17  # zero + 8 exponents x 2 signs: 5 bits
    # codes decode to the weights
    q = np.array([0.0] + [s * 2.0 ** k for k in range(n2, n1 + 1) for s in (1, -1)], np.float32)
    assert np.array_equal(q[co], wo)


@pytest.mark.parametrize("mode_q", ["twn", "ttq"])
def test_ternary_conv_module_forward_and_gradients(mode_q):
    torch.manual_seed(3)
    conv = T.TernaryConv2d(8, 12, 3, 2, 1, mode_q).to(DEV)
    x = torch.randn(4, 8, 10, 10, device=DEV, requires_grad=True)
    y = conv(x)
    gy = torch.randn_like(y)
    y.backward(gy)
    w = conv.weight.detach().cpu().numpy()
    ref = O.tern_forward(w, mode_q, 0.7, False, *((float(conv.Wp.detach()), float(conv.Wn.detach())) if mode_q == "ttq" else (None, None)))
    Wt = torch.from_numpy(ref["Wt"]).to(DEV)
    xt = x.detach().clone().requires_grad_(True)
    wt = Wt.clone().requires_grad_(True)
    yt = torch.nn.functional.conv2d(xt, wt, stride=2, padding=1)
    yt.backward(gy)
    assert (y - yt).abs().max().item() <= 2e-5 * yt.abs().max().item()
    assert (x.grad - xt.grad).abs().max().item() <= 2e-5 * xt.grad.abs().max().item()
    dWt = wt.grad.cpu().numpy()
    if mode_q == "twn":
        assert np.abs(conv.weight.grad.cpu().numpy() - dWt).max() <= 2e-5 * np.abs(dWt).max()
        assert conv.Wp is None
    else:
        dw_ref, dp_ref, dn_ref = O.tern_backward(w, dWt, "ttq", ref["delta"], ref["pos"], ref["neg"])
        assert np.abs(conv.weight.grad.cpu().numpy() - dw_ref).max() <= 2e-5 * np.abs(dw_ref).max()
        scale = np.abs(dWt).sum()
        assert abs(float(conv.Wp.grad) - dp_ref) <= 2e-5 * scale and abs(float(conv.Wn.grad) - dn_ref) <= 2e-5 * scale


def test_ternary_resnet20_trains_and_inq_stages():
    from quantized_cnn_amd.train import TrainStep
    torch.manual_seed(0)
    c = argparse.Namespace(mode_q="ttq", l2_reg=5e-4, prim_lr=1e-2, q_lr=1e-3, lr=1e-3)
    model = T.ResNet(c, 3).to(DEV).train()
    x = torch.randn(16, 3, 32, 32, device=DEV)
    y = torch.randint(0, 10, (16,), device=DEV)
    step = TrainStep(model, c, T)
    losses = [float(step(x, y)) for _ in range(6)]
    assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses
    # INQ: stages freeze the right share, frozen weights are powers of two and stop moving
    ci = argparse.Namespace(mode_q="inq", l2_reg=0.0, prim_lr=1e-2, q_lr=1e-3, lr=1e-2)
    net = T.ResNet(ci, 1).to(DEV).train()
    convs = [m for m in net.modules() if isinstance(m, T.INQConv2d)]
    st = TrainStep(net, ci, T)
    xb, yb = torch.randn(8, 3, 32, 32, device=DEV), torch.randint(0, 10, (8,), device=DEV)
    for frac in (0.5, 0.75, 1.0):
        for m in convs:
            m.quantize_portion(frac)
        frozen_before = [m.effective_weight().detach()[m.mask == 0].clone() for m in convs]
        for m in convs:
            share = 1.0 - float(m.mask.float().mean())
            assert abs(share - frac) <= 2.0 / m.weight.numel() + 1e-3, (frac, share)
        st(xb, yb)
        torch.nn.functional.cross_entropy(net(xb), yb).backward()
        for m, fb in zip(convs, frozen_before):
            assert torch.equal(m.effective_weight().detach()[m.mask == 0], fb)  # frozen levels stay whatever the optimizer does
            assert float(m.weight.grad[m.mask == 0].abs().max()) == 0.0         # and receive no gradient
            assert frac == 1.0 or float(m.weight.grad[m.mask == 1].abs().max()) > 0.0
            fz = m.effective_weight().detach()[m.mask == 0].abs()
            nz = fz[fz > 0]
            assert torch.equal(torch.log2(nz), torch.log2(nz).round())
            m.weight.grad = None
    m = convs[0]
    xin = torch.randn(2, m.weight.shape[1], 12, 12, device=DEV)
    with torch.no_grad():
        assert torch.equal(m.forward_codes(xin), m(xin)) or (m.forward_codes(xin) - m(xin)).abs().max().item() <= 1e-6


@pytest.mark.parametrize("mode_q", ["twn", "ttq"])
def test_model_wide_quantiser_node_equals_per_layer_path(mode_q, monkeypatch):
    """ternary.ResNet quantises all layers in one call and applies the quantiser's gradient rule to all of them in one
    call (one autograd node); every layer on its own (the path a single TernaryConv2d takes) must give the same
    loss and the same gradients to rounding."""
    torch.manual_seed(5)
    c = argparse.Namespace(mode_q=mode_q)
    model = T.ResNet(c, 1).to(DEV).train()
    x = torch.randn(8, 3, 32, 32, device=DEV)
    y = torch.randint(0, 10, (8,), device=DEV)

    def grads():
        model.zero_grad(set_to_none=True)
        loss = torch.nn.functional.cross_entropy(model(x), y)
        loss.backward()
        return float(loss), {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}

    for m in model.modules():  # batch-norm running statistics would differ between the two passes otherwise
        if isinstance(m, torch.nn.BatchNorm2d):
            m.momentum = 0.0
    l1, g1 = grads()
    monkeypatch.setattr(T, "quantize_all", lambda ls: None)
    l2, g2 = grads()
    # (the model-wide path convolves with the decoded weights on the fused nodes of the adaptive layers, a single layer
    # from its codes: other kernels, other summation order)
    assert abs(l1 - l2) <= 2e-6 * abs(l2)
    assert g1.keys() == g2.keys() and any(".Wp" in n for n in g1) == (mode_q == "ttq")
    for n in g1:
        assert float((g1[n] - g2[n]).abs().max()) <= 2e-4 * float(g2[n].abs().max()) + 1e-7, n


def test_model_wide_inq_node_equals_per_layer_path(monkeypatch):
    """ternary.ResNet hands every INQ layer its effective weight from one autograd node (gradients masked for all layers in
    one call); each layer computing its own must give the same loss and gradients bit for bit (same kernels, same order)."""
    torch.manual_seed(9)
    c = argparse.Namespace(mode_q="inq")
    model = T.ResNet(c, 1).to(DEV).train()
    for m in model.modules():
        if isinstance(m, T.INQConv2d):
            m.quantize_portion(0.5)
        if isinstance(m, torch.nn.BatchNorm2d):
            m.momentum = 0.0
    x = torch.randn(8, 3, 32, 32, device=DEV)
    y = torch.randint(0, 10, (8,), device=DEV)

    def grads():
        model.zero_grad(set_to_none=True)
        loss = torch.nn.functional.cross_entropy(model(x), y)
        loss.backward()
        return float(loss.detach()), {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}

    l1, g1 = grads()
    monkeypatch.setattr(T, "inq_all", lambda ls: None)
    l2, g2 = grads()
    assert l1 == l2 and g1.keys() == g2.keys()
    for n in g1:
        assert torch.equal(g1[n], g2[n]), n
    frozen = [m for m in model.modules() if isinstance(m, T.INQConv2d)][0]
    assert float(frozen.weight.grad[frozen.mask == 0].abs().max()) == 0.0


def test_captured_step_replays_the_eager_steps():
    """train.CapturedStep (the whole step as one HIP graph) against the same steps run eagerly from the same start:
    same kernels in the same order, so parameters, momentum buffers and batch-norm statistics agree bit for bit."""
    from quantized_cnn_amd.train import CapturedStep
    c = argparse.Namespace(mode_q="ttq")
    xs = [torch.randn(8, 3, 32, 32, device=DEV, generator=torch.Generator(device=DEV).manual_seed(20 + i)) for i in range(5)]
    ys = [torch.randint(0, 10, (8,), device=DEV, generator=torch.Generator(device=DEV).manual_seed(40 + i)) for i in range(5)]

    def fresh():
        torch.manual_seed(1)
        model = T.ResNet(c, 1).to(DEV).train()
        return model, torch.optim.SGD(model.parameters(), lr=1e-2, momentum=0.9)

    m1, o1 = fresh()
    eager_losses = []
    for x, y in zip(xs[:1] + xs, ys[:1] + ys):  # (the captured step's one warm-up step is a real step on the first batch)
        o1.zero_grad(set_to_none=True)
        loss = torch.nn.functional.cross_entropy(m1(x), y)
        loss.backward()
        o1.step()
        eager_losses.append(float(loss.detach()))
    m2, o2 = fresh()
    with pytest.raises(ValueError):
        CapturedStep(m2, o2, torch.nn.functional.cross_entropy, warmup=0)
    step = CapturedStep(m2, o2, torch.nn.functional.cross_entropy, warmup=1)
    graph_losses = [float(step(x, y)) for x, y in zip(xs, ys)]
    assert graph_losses == eager_losses[1:]
    for (n, a), (_, b) in zip(m1.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a, b), n
    with pytest.raises(ValueError):
        step(xs[0][:4], ys[0][:4])


def test_inq_layer_survives_a_state_dict_round_trip():
    """mask, codes and exponents are buffers: a layer rebuilt from a state_dict computes with the same frozen levels,
    keeps them out of the gradient, and continues with the next stage on the same exponents."""
    torch.manual_seed(4)
    a = T.INQConv2d(16, 32, 3, 1, 1).to(DEV)
    a.quantize_portion(0.5)
    b = T.INQConv2d(16, 32, 3, 1, 1).to(DEV)
    b.load_state_dict(a.state_dict())
    assert torch.equal(a.effective_weight().detach(), b.effective_weight().detach())
    assert torch.equal(a.exponents, b.exponents)
    with torch.no_grad():
        b.weight.add_(0.123)  # whatever an optimizer does to the frozen coordinates must not show
    eff = b.effective_weight().detach()
    assert torch.equal(eff[b.mask == 0], a.effective_weight().detach()[a.mask == 0])
    x = torch.randn(2, 16, 8, 8, device=DEV)
    b(x).sum().backward()
    assert float(b.weight.grad[b.mask == 0].abs().max()) == 0.0
    b.quantize_portion(0.75)
    assert abs(1.0 - float(b.mask.float().mean()) - 0.75) <= 1e-3 and torch.equal(a.exponents, b.exponents)


def test_captured_step_follows_inq_stages():
    """A step captured as one HIP graph BEFORE an INQ stage must train on the new partition after
    quantize_portion(): the boolean masks and decoded frozen values the graph reads are persistent tensors that
    the stage rewrites in place.  The same schedule run eagerly from the same start gives the same parameters
    bit for bit (same kernels, same order)."""
    from quantized_cnn_amd.train import CapturedStep
    c = argparse.Namespace(mode_q="inq")
    g = torch.Generator(device=DEV).manual_seed(7)
    xs = [torch.randn(8, 3, 32, 32, device=DEV, generator=g) for _ in range(6)]
    ys = [torch.randint(0, 10, (8,), device=DEV, generator=g) for _ in range(6)]

    def fresh():
        torch.manual_seed(2)
        model = T.ResNet(c, 1).to(DEV).train()
        return model, torch.optim.SGD(model.parameters(), lr=1e-2, momentum=0.9), [m for m in model.modules() if isinstance(m, T.INQConv2d)]

    def eager_step(model, opt, x, y):
        opt.zero_grad(set_to_none=True)
        loss = torch.nn.functional.cross_entropy(model(x), y)
        loss.backward()
        opt.step()
        return float(loss.detach())

    schedule = [(None, 0), (None, 1), (0.5, 2), (None, 3), (0.75, 4), (1.0, 5)]  # (stage before the step, batch)
    m1, o1, c1 = fresh()
    eager = [eager_step(m1, o1, xs[0], ys[0])]  # the captured step's warm-up step
    for frac, i in schedule:
        if frac is not None:
            for m in c1:
                m.quantize_portion(frac)
        eager.append(eager_step(m1, o1, xs[i], ys[i]))
    m2, o2, c2 = fresh()
    step = CapturedStep(m2, o2, torch.nn.functional.cross_entropy, warmup=1)
    graph = []
    for frac, i in schedule:
        if frac is not None:
            for m in c2:
                m.quantize_portion(frac)
        graph.append(float(step(xs[i], ys[i])))
    assert graph == eager[1:]
    for (n, a), (_, b) in zip(m1.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a, b), n
    for m in c2:  # every weight frozen at a power of two (or zero), whatever momentum did to the parameter
        w = m.effective_weight().detach().abs()
        nz = w[w > 0]
        assert int(m.mask.sum()) == 0 and torch.equal(torch.log2(nz), torch.log2(nz).round())
    # a state_dict load into the captured model is seen by the next replay as well
    sd = {k: v.clone() for k, v in m1.state_dict().items()}
    l_e = eager_step(m1, o1, xs[0], ys[0])
    m2.load_state_dict(sd)
    assert float(step(xs[0], ys[0])) == l_e


@pytest.mark.parametrize("mode_q", ["ttq", "inq"])
def test_checkpoint_round_trip_of_models_without_adaptive_layers(mode_q, tmp_path):
    """Ternary / INQ models have no p_c / q_level, so train.make_optimizers holds one optimizer ("rest"); the
    checkpoint keeps the reference's eight keys (adp_qtz_fan.py:331-340) with empty entries for the absent groups and
    restores the one optimizer from the entry of its role."""
    from quantized_cnn_amd import checkpoint as ck
    from quantized_cnn_amd.train import TrainStep
    c = argparse.Namespace(mode_q=mode_q, l2_reg=5e-4, lr=1e-3)
    torch.manual_seed(3)
    x = torch.randn(8, 3, 32, 32, device=DEV)
    y = torch.randint(0, 10, (8,), device=DEV)
    model = T.ResNet(c, 1).to(DEV).train()
    if mode_q == "inq":
        for m in model.modules():
            if isinstance(m, T.INQConv2d):
                m.quantize_portion(0.5)
    step = TrainStep(model, c, T)
    assert step.opts.roles == ("rest",) and step.opts.by_role("prim") is None
    for _ in range(2):
        step(x, y)
    path = str(tmp_path / "ck.pth")
    ck.save_checkpoint(path, model, step.opts, 1, 3, 17, 0.25)
    res = torch.load(path, map_location="cpu", weights_only=True)
    assert set(res) == set(ck.CHECKPOINT_KEYS) and res["optimizer_prim"]["param_groups"] == [] and res["optimizer"]["param_groups"]
    after = float(step(x, y))
    model2 = T.ResNet(c, 1).to(DEV).train()
    step2 = TrainStep(model2, c, T)
    assert ck.load_checkpoint(path, model2, step2.opts) == (1, 3, 17, 0.25)
    assert float(step2(x, y)) == after
    for (n, a), (_, b) in zip(model.state_dict().items(), model2.state_dict().items()):
        assert torch.equal(a, b), n
    with pytest.raises(ValueError):
        ck.save_checkpoint(path, model, [step.opts[0]], 1, 0, 0, 0.0)  # a bare list that is not the reference's three
