"""HIP kernels, called through the C ABI, against the CPU oracle and the reference fixtures.

Bars: with AQ_FLAG_EXACT (and always when codes are emitted) W, codes and dp_c are BIT-EXACT against
the oracle (same IEEE-754 operation sequence).  The default fast arithmetic (exp on the transcendental
unit) is held to |W - W_oracle| <= 2e-6*max|q| + 2e-6*|W| and dp_c to 3e-5 of the row scale.
Sums across rows (sum W^2, dq) and the convolutions are fp32 reductions in a different (fixed) order
than the oracle's double accumulation: rel 2e-5 of the result's scale.
"""
import numpy as np
import pytest
import torch

from oracle import oracle as O

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    import quantized_cnn_amd.ops as ops
DEV = "cuda"


def cu(a, dtype=torch.float32):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV, dtype)


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def assert_bits_equal(got, want, what):
    g, w = bits(got), bits(want)
    nan = np.isnan(got) & np.isnan(want)
    bad = (g != w) & ~nan
    assert not bad.any(), "%s: %d of %d values differ, first at %s: %r vs %r" % (
        what, bad.sum(), bad.size, np.argwhere(bad)[0], got[tuple(np.argwhere(bad)[0])], want[tuple(np.argwhere(bad)[0])])


def hint_levels(L, s=0.3):
    return O.levels_hint(s, L) if L % 2 == 0 else O.levels_fix(s, L + 1)


def rand_levels(rs, L):
    return (rs.uniform(-1, 1, L) * 0.25).astype(np.float32)


@pytest.mark.parametrize("L", [1, 3, 4, 8, 31, 32, 33, 48, 64, 65, 100, 128])
@pytest.mark.parametrize("rows", [1, 7, 64, 65, 255, 1000, 4099])
def test_synth_forward_bit_exact(L, rows):
    rs = np.random.RandomState(L * 10007 + rows)
    p = rs.standard_normal((rows, L)).astype(np.float32)
    q = rand_levels(rs, L)
    beta = float(rs.choice([1, 2, 3, 4, 0.5, -1.5]))
    ref = O.synth_fwd(p, q, beta)
    out = ops.synth_forward([cu(p)], [cu(q)], beta, soft=True, codes=True, sumsq=True)
    assert_bits_equal(out["W"][0].cpu().numpy(), ref["W"], "W")
    assert np.array_equal(out["idx"][0].cpu().numpy().astype(np.int32), ref["idx"])
    assert_bits_equal(out["W_hard"][0].cpu().numpy(), ref["W_hard"], "W_hard")
    ss = float(out["sumsq"].cpu()[0])
    assert abs(ss - ref["sumsq"]) <= 2e-5 * ref["sumsq"] + 1e-30
    # codes alone (what a hard-mode forward pass asks for): the arg-max-of-p kernel, same codes bit for bit
    out4 = ops.synth_forward([cu(p)], [cu(q)], beta, soft=False, codes=True)
    assert np.array_equal(out4["idx"][0].cpu().numpy().astype(np.int32), ref["idx"])
    assert_bits_equal(out4["W_hard"][0].cpu().numpy(), ref["W_hard"], "W_hard(codes only)")
    # soft-only launch with the exact flag gives the same bits
    out2 = ops.synth_forward([cu(p)], [cu(q)], beta, soft=True, codes=False, sumsq=False, exact=True)
    assert_bits_equal(out2["W"][0].cpu().numpy(), ref["W"], "W(soft only)")
    # default fast arithmetic (the training path): same W within fp32 tolerance
    out3 = ops.synth_forward([cu(p)], [cu(q)], beta, soft=True, sumsq=True, exact=False)
    Wf = out3["W"][0].cpu().numpy()
    assert np.all(np.abs(Wf - ref["W"]) <= 2e-6 * np.abs(q).max() + 2e-6 * np.abs(ref["W"]))
    assert abs(float(out3["sumsq"].cpu()[0]) - ref["sumsq"]) <= 2e-5 * ref["sumsq"] + 1e-30


def test_synth_forward_reference_fixtures(golden):
    """Same inputs as the reference run: codes equal to the reference's wherever the reference's own
    arg-max was not decided by fp32 rounding; W within fp32 tolerance of the reference."""
    g = golden("synth")
    for c in range(int(g["ncase"])):
        k = "c%02d_" % c
        beta = float(g[k + "meta"][2])
        p, q = g[k + "p_c"], g[k + "q"]
        out = ops.synth_forward([cu(p)], [cu(q)], beta, soft=True, codes=True)
        W = out["W"][0].cpu().numpy()
        idx = out["idx"][0].cpu().numpy()
        ref = O.synth_fwd(p, q, beta)
        assert_bits_equal(W, ref["W"], "W case %d" % c)
        assert np.array_equal(idx.astype(np.int32), ref["idx"])
        flag = g[k + "rounding_tie"]
        assert np.all((idx == g[k + "idx"]) | flag)
        only = ops.synth_forward([cu(p)], [cu(q)], beta, soft=False, codes=True)
        assert np.array_equal(only["idx"][0].cpu().numpy(), idx), c  # special rows included: ties, zeros, huge, tiny
        ok = ~np.isnan(g[k + "W"])
        assert np.all(np.abs(W[ok] - g[k + "W"][ok]) <= 1e-6 * np.abs(q).max() + 2e-6 * np.abs(g[k + "W"][ok]))
        assert np.array_equal(np.isnan(W), np.isnan(g[k + "W"]))


@pytest.mark.parametrize("L", [4, 31, 32, 33, 64, 100])
@pytest.mark.parametrize("rows", [1, 63, 64, 300, 4099])
def test_synth_backward_bit_exact_dp(L, rows):
    rs = np.random.RandomState(L * 7919 + rows)
    p = rs.random_sample((rows, L)).astype(np.float32)
    q = rand_levels(rs, L)
    dW = rs.standard_normal(rows).astype(np.float32)
    beta, wd = 2.0, 1e-3
    dp_ref, dq_ref = O.synth_bwd(p, q, beta, dW, wd)
    dp, dq = ops.synth_backward([cu(p)], [cu(q)], beta, [cu(dW)], wd, exact=True)
    assert_bits_equal(dp[0].cpu().numpy(), dp_ref, "dp_c")
    got = dq[0].cpu().numpy().astype(np.float64)
    assert np.abs(got - dq_ref).max() <= 2e-5 * np.abs(dq_ref).max() + 1e-12
    # dW = None (regulariser only) and frozen levels (no dq)
    dp_ref, _ = O.synth_bwd(p, q, beta, None, wd)
    dp, dq = ops.synth_backward([cu(p)], [cu(q)], beta, [None], wd, need_dq=[False], exact=True)
    assert dq[0] is None
    assert_bits_equal(dp[0].cpu().numpy(), dp_ref, "dp_c(reg only)")
    # default fast arithmetic
    dp_ref, dq_ref = O.synth_bwd(p, q, beta, dW, wd)
    dp, dq = ops.synth_backward([cu(p)], [cu(q)], beta, [cu(dW)], wd, exact=False)
    got = dp[0].cpu().numpy()
    tol = 3e-5 * np.abs(dp_ref).max(axis=1, keepdims=True) + 1e-6 * np.abs(dp_ref).max()
    assert np.all(np.abs(got - dp_ref) <= tol)
    assert np.abs(dq[0].cpu().numpy() - dq_ref).max() <= 2e-5 * np.abs(dq_ref).max() + 1e-12


@pytest.mark.parametrize("L", [31, 32, 64, 100])
@pytest.mark.parametrize("beta", [1.0, 4.0, 64.0, 0.03, -2.0, 0.0])
def test_codes_only_kernel_on_near_ties(L, beta):
    """Rows built to sit on the decision boundary of the codes-only kernel: the two (or three) largest
    coefficients 0 to 40 ulps apart, in either index order, scattered over the lanes; some rows with an exact
    tie, a NaN, an infinity or all zeros.  Codes must be those of the exact sequence (= the oracle's)."""
    rs = np.random.RandomState(int(L * 31 + (beta + 70) * 1000))
    rows = 6000
    p = rs.random_sample((rows, L)).astype(np.float32) * 0.9
    for r in range(rows):
        i, j, k = rs.choice(L, 3, replace=False)
        top = np.float32(0.95 + 0.04 * rs.random_sample())
        p[r, i] = top
        p[r, j] = (top.view(np.uint32) - np.uint32(rs.randint(0, 41))).view(np.float32)
        if r % 3 == 0:
            p[r, k] = (top.view(np.uint32) - np.uint32(rs.randint(0, 9))).view(np.float32)
        if r % 7 == 0:
            p[r] = -p[r]  # all-negative rows: the maximum is the element closest to zero
    p[11] = 0.0
    p[12, 3] = np.nan
    p[13, 5] = np.inf
    p[14] = 1e-25
    p[15] = 3e19
    q = rand_levels(rs, L)
    ref = O.synth_fwd(p, q, beta)
    out = ops.synth_forward([cu(p)], [cu(q)], beta, soft=False, codes=True)
    got = out["idx"][0].cpu().numpy().astype(np.int32)
    assert np.array_equal(got, ref["idx"]), np.argwhere(got != ref["idx"])[:5]
    assert_bits_equal(out["W_hard"][0].cpu().numpy(), ref["W_hard"], "W_hard")


def test_synth_multi_segment_resnet20_shapes():
    """All 19 quantized layers of ResNet-20 (fix: L=31, others L=32, mixed) in one call."""
    rs = np.random.RandomState(5)
    shapes = [(16, 16)] * 6 + [(32, 16)] + [(32, 32)] * 5 + [(64, 32)] + [(64, 64)] * 5
    ps, qs, Ls = [], [], []
    for i, (co, ci) in enumerate(shapes):
        L = 31 if i % 3 == 0 else 32
        ps.append(rs.random_sample((co, ci, 3, 3, L)).astype(np.float32))
        qs.append(hint_levels(L, 0.1 + 0.05 * i))
        Ls.append(L)
    ps.append(rs.random_sample((64, 10, 32)).astype(np.float32))
    qs.append(hint_levels(32, 0.9))
    beta = 3.0
    out = ops.synth_forward([cu(p) for p in ps], [cu(q) for q in qs], beta, soft=True, codes=True, sumsq=True)
    dWs = [rs.standard_normal(p.shape[:-1]).astype(np.float32) for p in ps]
    need = [i % 2 == 0 for i in range(len(ps))]
    dps, dqs = ops.synth_backward([cu(p) for p in ps], [cu(q) for q in qs], beta, [cu(d) for d in dWs], 1e-3, need_dq=need, exact=True)
    ss = out["sumsq"].cpu().numpy()
    for i, (p, q) in enumerate(zip(ps, qs)):
        ref = O.synth_fwd(p, q, beta)
        assert out["W"][i].shape == p.shape[:-1]
        assert_bits_equal(out["W"][i].cpu().numpy(), ref["W"], "W[%d]" % i)
        assert np.array_equal(out["idx"][i].cpu().numpy().astype(np.int32), ref["idx"])
        assert abs(ss[i] - ref["sumsq"]) <= 2e-5 * ref["sumsq"]
        dp_ref, dq_ref = O.synth_bwd(p, q, beta, dWs[i], 1e-3)
        assert_bits_equal(dps[i].cpu().numpy(), dp_ref, "dp[%d]" % i)
        if need[i]:
            assert np.abs(dqs[i].cpu().numpy() - dq_ref).max() <= 2e-5 * np.abs(dq_ref).max()
        else:
            assert dqs[i] is None


def test_synth_is_run_to_run_deterministic_and_handles_empty():
    rs = np.random.RandomState(3)
    p, q = cu(rs.random_sample((70001, 32)).astype(np.float32)), cu(hint_levels(32))
    dW = cu(rs.standard_normal(70001).astype(np.float32))
    a = ops.synth_forward([p], [q], 4.0, sumsq=True)
    b = ops.synth_forward([p], [q], 4.0, sumsq=True)
    assert torch.equal(a["sumsq"], b["sumsq"]) and torch.equal(a["W"][0], b["W"][0])
    d1 = ops.synth_backward([p], [q], 4.0, [dW], 0.0)
    d2 = ops.synth_backward([p], [q], 4.0, [dW], 0.0)
    assert torch.equal(d1[1][0], d2[1][0]) and torch.equal(d1[0][0], d2[0][0])
    e = ops.synth_forward([p[:0]], [q], 1.0, sumsq=True)
    assert e["W"][0].numel() == 0 and float(e["sumsq"][0]) == 0.0
    with pytest.raises(RuntimeError, match="beta"):
        ops.synth_forward([p], [q], 1e3)
    with pytest.raises(ValueError):
        ops.synth_forward([p], [q[:5]], 1.0)
    with pytest.raises(ValueError):
        ops.synth_forward([p.double()], [q], 1.0)


CONV_SHAPES = [  # cin, cout, stride, H, W, N
    (16, 16, 1, 32, 32, 2), (16, 32, 2, 32, 32, 2), (32, 32, 1, 16, 16, 3), (32, 64, 2, 16, 16, 2), (64, 64, 1, 8, 8, 5),
    (3, 7, 1, 5, 5, 2), (5, 3, 2, 6, 6, 2), (16, 16, 1, 9, 9, 1), (16, 32, 2, 9, 9, 1), (64, 64, 1, 4, 4, 1),
    (17, 33, 1, 13, 7, 2), (33, 17, 2, 11, 20, 3), (1, 1, 1, 1, 1, 1), (4, 4, 2, 1, 1, 1), (8, 80, 1, 40, 36, 1),
    # stride-2 dgrad by parity classes (Cout a multiple of 32, even H and W) with ragged class tiles, and its
    # neighbours that fall back to the zero-dilated image (odd size / other channel counts)
    (8, 32, 2, 24, 24, 3), (5, 64, 2, 20, 12, 2), (16, 32, 2, 2, 2, 1), (8, 32, 2, 23, 24, 2), (8, 48, 2, 24, 24, 2), (12, 96, 2, 6, 36, 1),
]


@pytest.mark.parametrize("shape", CONV_SHAPES)
def test_conv_forward_dgrad_wgrad_vs_oracle(shape):
    cin, cout, stride, H, Wd, N = shape
    rs = np.random.RandomState(sum(shape) * 31 + H)
    x = rs.standard_normal((N, cin, H, Wd)).astype(np.float32)
    W = (rs.standard_normal((cout, cin, 3, 3)) * 0.2).astype(np.float32)
    y_ref = O.conv_fwd(x, W, stride, 1)
    y = ops.conv_forward(cu(x), cu(W), stride).cpu().numpy()
    assert y.shape == y_ref.shape
    assert np.abs(y - y_ref).max() <= 2e-5 * np.abs(y_ref).max() + 1e-30
    dy = rs.standard_normal(y_ref.shape).astype(np.float32)
    dx_ref = O.conv_dgrad(dy, W, x.shape, stride, 1)
    dx = ops.conv_dgrad(cu(dy), cu(W), x.shape, stride).cpu().numpy()
    assert np.abs(dx - dx_ref).max() <= 2e-5 * np.abs(dx_ref).max() + 1e-30
    dW_ref = O.conv_wgrad(x, dy, 3, stride, 1)
    dW = ops.conv_wgrad(cu(x), cu(dy), stride).cpu().numpy()
    assert np.abs(dW - dW_ref).max() <= 2e-5 * np.abs(dW_ref).max() + 1e-30


def test_conv_asymmetric_identity_probe():
    """One-hot weights and ramp inputs: catches transposed fragments / swapped taps exactly."""
    cin, cout, H = 16, 16, 8
    x = np.arange(2 * cin * H * H, dtype=np.float32).reshape(2, cin, H, H) / 7.0
    for tap in range(9):
        W = np.zeros((cout, cin, 3, 3), np.float32)
        for co in range(cout):
            W[co, (co * 5 + 3) % cin, tap // 3, tap % 3] = 1.0
        y = ops.conv_forward(cu(x), cu(W), 1).cpu().numpy()
        assert np.array_equal(y, O.conv_fwd(x, W, 1, 1)), tap


def test_conv_reference_fixtures(golden):
    g = golden("conv")
    for c in range(int(g["ncase"])):
        k = "c%02d_" % c
        cin, cout, stride, H, N = [int(v) for v in g[k + "shape"]]
        x, W, dy = cu(g[k + "x"]), cu(g[k + "W"]), cu(g[k + "dy"])
        y = ops.conv_forward(x, W, stride).cpu().numpy()
        assert np.abs(y - g[k + "y"]).max() <= 2e-5 * np.abs(g[k + "y"]).max()
        dx = ops.conv_dgrad(dy, W, x.shape, stride).cpu().numpy()
        assert np.abs(dx - g[k + "dx"]).max() <= 2e-5 * np.abs(g[k + "dx"]).max()
        dW = ops.conv_wgrad(x, dy, stride).cpu().numpy()
        assert np.abs(dW - g[k + "dW"]).max() <= 2e-5 * np.abs(g[k + "dW"]).max()


def test_conv_codes_equals_conv_of_decoded_weights():
    rs = np.random.RandomState(8)
    for (cin, cout, stride, H) in [(16, 32, 2, 16), (32, 32, 1, 8), (5, 3, 1, 6)]:
        L = 31
        q = O.levels_fix(0.4, 32)
        idx = rs.randint(0, L, (cout, cin, 3, 3)).astype(np.uint8)
        x = rs.standard_normal((2, cin, H, H)).astype(np.float32)
        y = ops.conv_forward_codes(cu(x), cu(idx, torch.uint8), cu(q), stride)
        y2 = ops.conv_forward(cu(x), cu(q[idx]), stride)
        assert torch.equal(y, y2)


def test_conv_full_batch_linearity_and_determinism():
    """BASELINE-size layer (batch 128): conv(a*x1 + x2) == a*conv(x1) + conv(x2) within fp32, wgrad reproducible."""
    torch.manual_seed(0)
    for (cin, cout, stride, H) in [(16, 16, 1, 32), (32, 64, 2, 16), (64, 64, 1, 8)]:
        x1 = torch.randn(128, cin, H, H, device=DEV)
        x2 = torch.randn(128, cin, H, H, device=DEV)
        W = torch.randn(cout, cin, 3, 3, device=DEV) * 0.1
        y = ops.conv_forward(2.0 * x1 + x2, W, stride)
        y12 = 2.0 * ops.conv_forward(x1, W, stride) + ops.conv_forward(x2, W, stride)
        assert (y - y12).abs().max() <= 2e-5 * y.abs().max()
        dy = torch.randn_like(y)
        a, b = ops.conv_wgrad(x1, dy, stride), ops.conv_wgrad(x1, dy, stride)
        assert torch.equal(a, b)
        # <dy, conv(x)> == <dgrad(dy), x> == <wgrad(x, dy), W>   (adjoint identities)
        lhs = (dy.double() * ops.conv_forward(x1, W, stride).double()).sum()
        mid = (ops.conv_dgrad(dy, W, x1.shape, stride).double() * x1.double()).sum()
        rhs = (a.double() * W.double()).sum()
        assert abs(lhs - mid) <= 1e-5 * abs(lhs) + 1e-3 and abs(lhs - rhs) <= 1e-5 * abs(lhs) + 1e-3


def test_linear_vs_oracle_and_fixture(golden):
    g = golden("linear")
    for c in range(int(g["ncase"])):
        k = "c%02d_" % c
        x, W, b, gy = g[k + "x"], g[k + "W"], g[k + "bias"], g[k + "g"]
        y = ops.linear_forward(cu(x), cu(W), cu(b)).cpu().numpy()
        assert np.abs(y - g[k + "y"]).max() <= 1e-5 * np.abs(g[k + "y"]).max()
        dx, dW, db = ops.linear_backward(cu(x), cu(W), cu(gy))
        assert np.abs(dx.cpu().numpy() - g[k + "dx"]).max() <= 1e-5 * np.abs(g[k + "dx"]).max()
        assert np.abs(dW.cpu().numpy() - g[k + "dW"]).max() <= 1e-5 * np.abs(g[k + "dW"]).max()
        assert np.abs(db.cpu().numpy() - g[k + "dbias"]).max() <= 1e-5 * np.abs(g[k + "dbias"]).max()


# ------------------------------------------------------------------------- batch norm (+ ReLU)
# fp32 reductions in a fixed tree order against the oracle's double accumulation: 1e-5 of the tensor's
# scale for y / dx, rel 2e-5 for per-channel sums; elements whose pre-activation is within 1e-6 of zero
# may fall on either side of the ReLU and are excluded from the backward comparison.

BN_SHAPES = [(128, 16, 32, 32), (128, 32, 16, 16), (128, 64, 8, 8), (8, 16, 32, 32), (3, 5, 7, 3), (2, 1, 1, 1), (5, 7, 1, 1),
             (130, 3, 5, 5), (1, 4, 50, 50), (2, 2, 64, 64)]


@pytest.mark.parametrize("relu", [True, False])
@pytest.mark.parametrize("shape", BN_SHAPES)
def test_bn_relu_forward_backward_vs_oracle(shape, relu):
    rs = np.random.RandomState(sum(shape) + int(relu))
    N, Cn = shape[0], shape[1]
    x = (rs.standard_normal(shape) * rs.uniform(0.5, 3.0, (1, Cn, 1, 1)) + rs.uniform(-2, 2, (1, Cn, 1, 1))).astype(np.float32)
    gamma = (1.0 + 0.3 * rs.standard_normal(Cn)).astype(np.float32)
    beta = (0.3 * rs.standard_normal(Cn)).astype(np.float32)
    dy = rs.standard_normal(shape).astype(np.float32)
    res = rs.standard_normal(shape).astype(np.float32)
    rm0, rv0 = rs.standard_normal(Cn).astype(np.float32), rs.uniform(0.5, 2, Cn).astype(np.float32)
    rm, rv, nbt = cu(rm0), cu(rv0), torch.tensor(41, device=DEV, dtype=torch.int64)
    xd, gd, bd = cu(x), cu(gamma), cu(beta)
    y, mean, rstd = ops.bn_relu_forward(xd, gd, bd, rm, rv, nbt, 1e-5, 0.1, relu=relu)
    yo, mo, ro, vo = O.bn_relu_fwd(x, gamma, beta, 1e-5, relu)
    assert np.abs(y.cpu().numpy() - yo).max() <= 1e-5 * max(1.0, np.abs(yo).max())
    assert np.allclose(mean.cpu().numpy(), mo, rtol=1e-5, atol=1e-6)
    assert np.allclose(rstd.cpu().numpy(), ro, rtol=2e-5)
    assert np.allclose(rm.cpu().numpy(), 0.9 * rm0 + 0.1 * mo, rtol=1e-5, atol=1e-6)
    assert np.allclose(rv.cpu().numpy(), 0.9 * rv0 + 0.1 * vo, rtol=2e-5)
    assert int(nbt) == 42
    # evaluation mode: running statistics, nothing updated
    ye = ops.bn_relu_forward(xd, gd, bd, rm, rv, None, 1e-5, 0.0, relu=relu, use_running_stats=True)[0]
    t = (x - rm.cpu().numpy().reshape(1, -1, 1, 1)) / np.sqrt(rv.cpu().numpy().reshape(1, -1, 1, 1) + 1e-5) * gamma.reshape(1, -1, 1, 1) + beta.reshape(1, -1, 1, 1)
    t = np.maximum(t, 0) if relu else t
    assert np.abs(ye.cpu().numpy() - t).max() <= 1e-5 * max(1.0, np.abs(t).max())
    assert int(nbt) == 42
    # backward, with and without the shortcut gradient
    for r in (None, res):
        dx, dg, db = ops.bn_relu_backward(cu(dy), xd, gd, bd, mean, rstd, res=cu(r) if r is not None else None, relu=relu)
        dxo, dgo, dbo = O.bn_relu_bwd(dy, x, gamma, beta, r, 1e-5, relu)
        pre = (x - mo.reshape(1, -1, 1, 1)) * ro.reshape(1, -1, 1, 1) * gamma.reshape(1, -1, 1, 1) + beta.reshape(1, -1, 1, 1)
        near = (np.abs(pre) < 1e-6) & relu
        if near.any():
            continue  # a borderline ReLU decision changes the channel sums: skip this (rare) draw
        scale = max(1.0, np.abs(dxo).max())
        assert np.abs(dx.cpu().numpy() - dxo).max() <= 2e-5 * scale
        n_per = x.size // Cn
        assert np.abs(dg.cpu().numpy() - dgo).max() <= 2e-5 * max(1.0, np.abs(dgo).max(), np.sqrt(n_per))
        assert np.abs(db.cpu().numpy() - dbo).max() <= 2e-5 * max(1.0, np.abs(dbo).max(), np.sqrt(n_per))


def test_bn_large_mean_does_not_cancel():
    """mean >> std: the two-pass slice statistics keep the variance accurate."""
    rs = np.random.RandomState(5)
    x = (1000.0 + 0.01 * rs.standard_normal((64, 8, 16, 16))).astype(np.float32)
    g, b = np.ones(8, np.float32), np.zeros(8, np.float32)
    y, mean, rstd = ops.bn_relu_forward(cu(x), cu(g), cu(b), None, None, None, 1e-5, 0.1, relu=False)
    yo, mo, ro, _ = O.bn_relu_fwd(x, g, b, 1e-5, False)
    assert np.allclose(rstd.cpu().numpy(), ro, rtol=1e-3)
    assert np.abs(y.cpu().numpy() - yo).max() <= 2e-2  # inputs carry 6e-5 absolute rounding, amplified by rstd ~ 100


def test_bn_matches_torch_batch_norm_and_is_deterministic():
    torch.manual_seed(3)
    x = torch.randn(128, 32, 16, 16, device=DEV) * 2 + 0.5
    bn = torch.nn.BatchNorm2d(32).to(DEV)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.normal_(0, 0.2)
    rm, rv, nbt = bn.running_mean.clone(), bn.running_var.clone(), bn.num_batches_tracked.clone()
    y1, m1, r1 = ops.bn_relu_forward(x, bn.weight.detach(), bn.bias.detach(), rm, rv, nbt, bn.eps, bn.momentum)
    y2, m2, r2 = ops.bn_relu_forward(x, bn.weight.detach(), bn.bias.detach(), rm.clone(), rv.clone(), nbt.clone(), bn.eps, bn.momentum)
    assert torch.equal(y1, y2) and torch.equal(m1, m2) and torch.equal(r1, r2)
    yt = torch.relu(bn(x))
    assert (y1 - yt).abs().max().item() <= 1e-5 * yt.abs().max().item()
    assert torch.allclose(rm, bn.running_mean, rtol=1e-5, atol=1e-6) and torch.allclose(rv, bn.running_var, rtol=1e-5)
    assert int(nbt) == int(bn.num_batches_tracked) == 1


def test_conv_forward_with_shortcut_add():
    rs = np.random.RandomState(8)
    for (N, cin, cout, s, H) in [(4, 16, 16, 1, 32), (3, 16, 32, 2, 16), (2, 5, 7, 1, 9)]:
        x, W = cu(rs.standard_normal((N, cin, H, H))), cu(rs.standard_normal((cout, cin, 3, 3)) * 0.1)
        y0 = ops.conv_forward(x, W, s)
        res = cu(rs.standard_normal(tuple(y0.shape)))
        y1 = ops.conv_forward(x, W, s, res=res)
        assert torch.equal(y1, y0 + res)  # one fp32 addition after the same accumulation


def test_bn_argument_errors():
    x = torch.zeros(1, 3, 1, 1, device=DEV)
    g = torch.ones(3, device=DEV)
    with pytest.raises(RuntimeError, match="more than one value"):
        ops.bn_relu_forward(x, g, g, None, None, None)
    with pytest.raises(RuntimeError, match="HIP device only"):
        ops.bn_relu_forward(torch.zeros(2, 3, 2, 2), g, g, None, None, None)


@pytest.mark.parametrize("shape", [(128, 16, 32, 32), (4, 32, 16, 16), (2, 3, 7, 5), (1, 1, 2, 2)])
def test_shortcut_pool_pad_forward_backward(shape):
    """AvgPool2d(2) + channel zero padding: exact up to the order of the four additions (1 ulp)."""
    rs = np.random.RandomState(sum(shape))
    N, Cn, H, Wd = shape
    front, back = Cn // 2, Cn - Cn // 2
    x = rs.standard_normal(shape).astype(np.float32)
    Ho, Wo = H // 2, Wd // 2
    pooled = x[:, :, :2 * Ho, :2 * Wo].reshape(N, Cn, Ho, 2, Wo, 2).astype(np.float64).mean(axis=(3, 5))
    want = np.zeros((N, Cn + front + back, Ho, Wo), np.float64)
    want[:, front:front + Cn] = pooled
    y = ops.shortcut_forward(cu(x), front, back)
    assert y.shape == want.shape
    assert np.abs(y.cpu().numpy() - want).max() <= 3e-7 * max(1.0, np.abs(want).max())
    dy = rs.standard_normal(want.shape).astype(np.float32)
    dx = ops.shortcut_backward(cu(dy), shape, front, back).cpu().numpy()
    wdx = np.zeros(shape, np.float32)
    wdx[:, :, :2 * Ho, :2 * Wo] = np.repeat(np.repeat(dy[:, front:front + Cn], 2, axis=2), 2, axis=3) * 0.25
    assert np.array_equal(dx, wdx)


def test_wgrad_partial_plus_batched_reduce_equals_wgrad():
    rs = np.random.RandomState(12)
    jobs, want, outs = [], [], []
    for (N, cin, cout, s, H) in [(16, 16, 16, 1, 32), (16, 16, 32, 2, 32), (16, 32, 32, 1, 16), (8, 64, 64, 1, 8), (2, 3, 5, 1, 7)] * 8:
        x, dy_shape = cu(rs.standard_normal((N, cin, H, H))), ops.conv_out_shape((N, cin, H, H), cout, s)
        dy = cu(rs.standard_normal(dy_shape))
        want.append(ops.conv_wgrad(x, dy, s))
        ws, nslab = ops.conv_wgrad_partial(x, dy, s)
        out = torch.empty_like(want[-1])
        jobs.append((ws, out, nslab))
        outs.append(out)
    assert len(jobs) == 40  # more than one launch of 32 jobs
    ops.conv_wgrad_reduce(jobs)
    for a, b in zip(outs, want):
        assert torch.equal(a, b)


# ------------------------------------------------------------------ Adam fused into the synthesis gradient
# gradient bit-exact as in synth_backward (exact mode); the Adam arithmetic runs in fp32 with fused
# multiply-adds against the oracle's double evaluation: |dp| <= 2e-6*lr-scale, moments rel 1e-6.

@pytest.mark.parametrize("L", [4, 31, 32, 64, 100])
def test_synth_backward_adam_vs_oracle(L):
    rs = np.random.RandomState(900 + L)
    rows = 777
    p0 = rs.standard_normal((rows, L)).astype(np.float32)
    q = rand_levels(rs, L)
    lr, b1, b2, eps = 1e-2, 0.9, 0.999, 1e-8
    p, m, v = cu(p0), torch.zeros(rows, L, device=DEV), torch.zeros(rows, L, device=DEV)
    step = torch.zeros((), device=DEV)
    po, mo, vo = p0.copy(), np.zeros_like(p0), np.zeros_like(p0)
    for t in range(1, 4):
        dW = rs.standard_normal(rows).astype(np.float32)
        dp_ref, dq_ref = ops.synth_backward([p.clone()], [cu(q)], 2.0, [cu(dW)], 1e-3, exact=True)
        g_dev = torch.empty_like(p)
        dq = ops.synth_backward_adam([p], [cu(q)], 2.0, [cu(dW)], [m], [v], [step], lr, (b1, b2), eps, wd=1e-3, exact=True, out_dp=[g_dev])
        assert float(step) == t
        assert torch.equal(g_dev, dp_ref[0])          # same gradient, bit for bit
        assert torch.equal(dq[0], dq_ref[0])
        g_or, _ = O.synth_bwd(po, q, 2.0, dW, 1e-3)   # oracle gradient at the oracle's own parameters
        O.adam_step(po, g_or, mo, vo, t, lr, b1, b2, eps)
        assert np.abs(p.cpu().numpy() - po).max() <= 2e-4 * lr * t + 1e-6, t   # noise-level gradients may flip the +-lr step
        frac_off = (np.abs(p.cpu().numpy() - po) > 1e-6).mean()
        assert frac_off <= 0.01, (t, frac_off)
        assert np.abs(m.cpu().numpy() - mo).max() <= 1e-5 * max(np.abs(mo).max(), 1e-6)
        assert np.abs(v.cpu().numpy() - vo).max() <= 1e-5 * max(np.abs(vo).max(), 1e-12)


def test_synth_backward_adam_equals_torch_fused_adam_on_resnet20_shapes():
    """All 19 layers in one call against synth_backward followed by torch.optim.Adam(fused=True)."""
    torch.manual_seed(4)
    shapes = [(16, 16, 3, 3)] * 6 + [(32, 16, 3, 3)] + [(32, 32, 3, 3)] * 5 + [(64, 32, 3, 3)] + [(64, 64, 3, 3)] * 5 + [(64, 10)]
    L = 32
    ps = [torch.rand(*s, L, device=DEV) for s in shapes]
    qs = [torch.randn(L, device=DEV) * 0.2 for _ in shapes]
    ref_p = [torch.nn.Parameter(p.clone()) for p in ps]
    opt = torch.optim.Adam(ref_p, lr=1e-2, fused=True)
    m = [torch.zeros_like(p) for p in ps]
    v = [torch.zeros_like(p) for p in ps]
    steps = [torch.zeros((), device=DEV) for _ in ps]
    for t in range(3):
        dW = [torch.randn(s, device=DEV) for s in shapes]
        dp, _ = ops.synth_backward([p.detach() for p in ref_p], qs, 1.0, dW, 1e-3)
        for p, g in zip(ref_p, dp):
            p.grad = g
        opt.step()
        ops.synth_backward_adam(ps, qs, 1.0, dW, m, v, steps, 1e-2, wd=1e-3)
        for a, b in zip(ps, ref_p):
            d = (a - b.detach()).abs()
            assert d.max().item() <= 2.1e-2 * (t + 1)
            assert (d > 2e-6).float().mean().item() <= 0.01
    st = opt.state[ref_p[0]]
    assert torch.allclose(m[0], st["exp_avg"], rtol=1e-4, atol=1e-9) and torch.allclose(v[0], st["exp_avg_sq"], rtol=1e-4, atol=1e-12)
    assert float(steps[0]) == float(st["step"]) == 3


def test_synth_backward_adam_argument_errors():
    p, q = torch.rand(8, 32, device=DEV), torch.rand(32, device=DEV)
    with pytest.raises(ValueError, match="exp_avg"):
        ops.synth_backward_adam([p], [q], 1.0, [None], [torch.zeros(8, 31, device=DEV)], [torch.zeros_like(p)], [torch.zeros((), device=DEV)], 1e-2)
    with pytest.raises(RuntimeError, match="out of range"):
        ops.synth_backward_adam([p], [q], 1.0, [None], [torch.zeros_like(p)], [torch.zeros_like(p)], [torch.zeros((), device=DEV)], 1e-2, betas=(1.5, 0.9))


@pytest.mark.parametrize("shape", [(128, 64, 8, 8), (5, 3, 4, 4), (2, 7, 1, 1), (3, 2, 13, 11)])
def test_bn_relu_with_fused_mean_pool(shape):
    """relu(bn(x)) followed by the mean over the plane, fused: equals the unfused kernels to 1e-6 of
    scale (other summation order), backward included."""
    rs = np.random.RandomState(sum(shape))
    N, Cn = shape[0], shape[1]
    x = cu(rs.standard_normal(shape) * 1.7 + 0.4)
    g, b = cu(1.0 + 0.2 * rs.standard_normal(Cn)), cu(0.3 * rs.standard_normal(Cn))
    rm, rv = torch.zeros(Cn, device=DEV), torch.ones(Cn, device=DEV)
    rm2, rv2 = rm.clone(), rv.clone()
    y, mean, rstd = ops.bn_relu_forward(x, g, b, rm, rv, None)
    pooled, mean2, rstd2 = ops.bn_relu_forward(x, g, b, rm2, rv2, None, pool=True)
    assert pooled.shape == (N, Cn)
    assert torch.equal(mean, mean2) and torch.equal(rstd, rstd2) and torch.equal(rm, rm2) and torch.equal(rv, rv2)
    want = y.double().mean(dim=(2, 3))
    assert (pooled.double() - want).abs().max().item() <= 1e-6 * max(1.0, want.abs().max().item())
    dpool = cu(rs.standard_normal((N, Cn)))
    HW = shape[2] * shape[3]
    dy = (dpool / HW).view(N, Cn, 1, 1).expand(shape).contiguous()
    dx1, dg1, db1 = ops.bn_relu_backward(dy, x, g, b, mean, rstd)
    dx2, dg2, db2 = ops.bn_relu_backward(dpool, x, g, b, mean, rstd, pool=True)
    assert (dx1 - dx2).abs().max().item() <= 1e-6 * max(1.0, dx1.abs().max().item())
    assert torch.allclose(dg1, dg2, rtol=1e-5, atol=1e-6) and torch.allclose(db1, db2, rtol=1e-5, atol=1e-6)
    # running-statistics mode
    pe = ops.bn_relu_forward(x, g, b, rm, rv, None, use_running_stats=True, pool=True)[0]
    ye = ops.bn_relu_forward(x, g, b, rm, rv, None, use_running_stats=True)[0]
    assert (pe.double() - ye.double().mean(dim=(2, 3))).abs().max().item() <= 1e-6 * max(1.0, ye.abs().max().item())


# ------------------------------------------- convolution that normalises its input while staging it
# Same operation sequence as the separate kernels (statistics merge, fma + max, MFMA accumulation order),
# so results are bit-identical to bn_relu_forward followed by conv_forward / conv_wgrad.

@pytest.mark.parametrize("shape", [(16, 16, 1, 32, 128), (16, 32, 2, 32, 128), (32, 32, 1, 16, 128), (32, 64, 2, 16, 128), (64, 64, 1, 8, 128),
                                   (16, 16, 1, 32, 3), (64, 64, 1, 8, 2), (32, 32, 1, 16, 5)])
def test_bnconv_equals_separate_kernels(shape):
    cin, cout, s, H, N = shape
    rs = np.random.RandomState(sum(shape))
    x = cu(rs.standard_normal((N, cin, H, H)) * 1.3 + 0.2)
    W = cu(rs.standard_normal((cout, cin, 3, 3)) * 0.1)
    g, b = cu(1.0 + 0.2 * rs.standard_normal(cin)), cu(0.2 * rs.standard_normal(cin))
    assert ops.bnconv_supported(x.shape, cout, s)
    rm1, rv1, n1 = torch.zeros(cin, device=DEV), torch.ones(cin, device=DEV), torch.zeros((), dtype=torch.int64, device=DEV)
    rm2, rv2, n2 = rm1.clone(), rv1.clone(), n1.clone()
    z, mean, rstd = ops.bn_relu_forward(x, g, b, rm1, rv1, n1)
    y_ref = ops.conv_forward(z, W, s)
    res = cu(rs.standard_normal(tuple(y_ref.shape)))
    y, mean2, rstd2 = ops.bnconv_forward(x, g, b, rm2, rv2, n2, 1e-5, 0.1, W, s)
    assert torch.equal(mean, mean2) and torch.equal(rstd, rstd2) and torch.equal(rm1, rm2) and torch.equal(rv1, rv2) and int(n2) == 1
    assert torch.equal(y, y_ref)
    y2, _, _ = ops.bnconv_forward(x, g, b, None, None, None, 1e-5, 0.1, W, s, res=res)
    assert torch.equal(y2, y_ref + res)
    dy = cu(rs.standard_normal(tuple(y_ref.shape)))
    ws, nslab = ops.bnconv_wgrad_partial(x, g, b, mean, rstd, dy, s)
    dW = torch.empty_like(W)
    ops.conv_wgrad_reduce([(ws, dW, nslab)])
    want = ops.conv_wgrad(z, dy, s)  # (shapes with 16-channel multiples run on another kernel there: same sums, other order)
    assert (dW - want).abs().max().item() <= 2e-6 * want.abs().max().item()


def test_bnconv_unsupported_shapes_are_reported():
    assert not ops.bnconv_supported((2, 8, 40, 36), 80, 1)      # tiles narrower than the output
    x = torch.zeros(2, 8, 40, 36, device=DEV)
    with pytest.raises(RuntimeError, match="aq_bnconv_supported"):
        ops.bnconv_forward(x, torch.ones(8, device=DEV), torch.zeros(8, device=DEV), None, None, None, 1e-5, 0.1,
                           torch.zeros(80, 8, 3, 3, device=DEV), 1)


# --------------------------------------- convolution that leaves its output's channel statistics behind
# The epilogue sums per workgroup what the statistics pass sums per slice: other grouping, so the merged
# mean / variance agree to rounding (a few 1e-7 relative), not to the bit.

@pytest.mark.parametrize("shape", [(3, 16, 1, 32, 128), (16, 16, 1, 32, 128), (16, 32, 2, 32, 128), (32, 32, 1, 16, 128), (32, 64, 2, 16, 128),
                                   (64, 64, 1, 8, 128), (16, 16, 1, 32, 3), (64, 64, 1, 8, 1), (32, 32, 1, 16, 5), (8, 20, 1, 16, 7)])
@pytest.mark.parametrize("with_res", [False, True])
def test_conv_epilogue_statistics_feed_batch_norm(shape, with_res):
    cin, cout, s, H, N = shape
    rs = np.random.RandomState(sum(shape) + with_res)
    x = cu(rs.standard_normal((N, cin, H, H)) * 1.3 + 0.2)
    W = cu(rs.standard_normal((cout, cin, 3, 3)) * 0.1)
    y_ref = ops.conv_forward(x, W, s)
    res = cu(rs.standard_normal(tuple(y_ref.shape)) + 0.5) if with_res else None
    if with_res:
        y_ref = ops.conv_forward(x, W, s, res)
    y, st = ops.conv_forward(x, W, s, res, stats=True)
    assert st is not None, "ResNet-20 shapes have the statistics epilogue"
    assert torch.equal(y, y_ref)
    g, b = cu(1.0 + 0.2 * rs.standard_normal(cout)), cu(0.2 * rs.standard_normal(cout))
    for pool in (False, True):
        rm1, rv1, n1 = torch.zeros(cout, device=DEV), torch.ones(cout, device=DEV), torch.zeros((), dtype=torch.int64, device=DEV)
        rm2, rv2, n2 = rm1.clone(), rv1.clone(), n1.clone()
        z1, mean1, rstd1 = ops.bn_relu_forward(y, g, b, rm1, rv1, n1, pool=pool)
        z2, mean2, rstd2 = ops.bn_relu_forward(y, g, b, rm2, rv2, n2, pool=pool, conv_stats=st)
        # against float64 statistics of y
        yd = y.double()
        m64, v64 = yd.mean(dim=(0, 2, 3)), yd.var(dim=(0, 2, 3), unbiased=False)
        scale = max(1.0, float(m64.abs().max()))
        assert (mean2.double() - m64).abs().max().item() <= 2e-6 * scale
        assert ((rstd2.double() - (v64 + 1e-5).rsqrt()) / (v64 + 1e-5).rsqrt()).abs().max().item() <= 5e-6
        assert torch.allclose(mean1, mean2, rtol=0, atol=2e-6 * scale) and torch.allclose(rstd1, rstd2, rtol=1e-5, atol=0)
        assert torch.allclose(rm1, rm2, rtol=0, atol=1e-6 * scale) and torch.allclose(rv1, rv2, rtol=1e-5, atol=1e-7) and int(n2) == 1
        assert (z1 - z2).abs().max().item() <= 2e-5 * max(1.0, float(z1.abs().max()))
    # determinism
    y3, st3 = ops.conv_forward(x, W, s, res, stats=True)
    assert torch.equal(st3[0], st[0])


def test_conv_epilogue_statistics_unsupported_shapes_and_errors():
    x = torch.randn(2, 8, 40, 36, device=DEV)
    W = torch.randn(12, 8, 3, 3, device=DEV)
    y, st = ops.conv_forward(x, W, 1, stats=True)  # tiles narrower than the output: no epilogue, plain convolution
    assert st is None and torch.equal(y, ops.conv_forward(x, W, 1))
    lib = ops._lib.lib()
    assert lib.aq_conv_stats_bytes(2, 8, 40, 36, 12, 3, 1, 1) == 0
    assert lib.aq_conv_stats_bytes(128, 16, 32, 32, 16, 3, 1, 1) > 0
    # a descriptor that does not describe the tensor is refused
    x2 = torch.randn(4, 16, 32, 32, device=DEV)
    y2, st2 = ops.conv_forward(x2, torch.randn(16, 16, 3, 3, device=DEV), 1, stats=True)
    g, b = torch.ones(16, device=DEV), torch.zeros(16, device=DEV)
    with pytest.raises(RuntimeError, match="do not describe"):
        ops.bn_relu_forward(y2[:2].contiguous(), g, b, conv_stats=st2)


@pytest.mark.parametrize("shape", [(16, 16, 1, 32, 128), (16, 32, 2, 32, 128), (32, 32, 1, 16, 128), (32, 64, 2, 16, 128), (64, 64, 1, 8, 128),
                                   (16, 16, 1, 32, 3), (64, 64, 1, 8, 1), (20, 8, 1, 16, 7)])
@pytest.mark.parametrize("with_res", [False, True])
def test_dgrad_epilogue_sums_feed_batch_norm_backward(shape, with_res):
    """aq_conv_dgrad_bn + aq_bn_relu_bwd(dgrad_sums) against aq_conv_dgrad + aq_bn_relu_bwd: dz bit-equal, the
    channel sums grouped differently (per workgroup instead of per slice), so dgamma / dbeta / dx to rounding."""
    cin, cout, s, H, N = shape
    rs = np.random.RandomState(sum(shape) + 7 * with_res)
    xb = cu(rs.standard_normal((N, cin, H, H)) * 1.3 + 0.2)
    W = cu(rs.standard_normal((cout, cin, 3, 3)) * 0.1)
    g, b = cu(1.0 + 0.2 * rs.standard_normal(cin)), cu(0.2 * rs.standard_normal(cin))
    z, mean, rstd = ops.bn_relu_forward(xb, g, b)
    y = ops.conv_forward(z, W, s)
    dy = cu(rs.standard_normal(tuple(y.shape)))
    res = cu(rs.standard_normal(tuple(xb.shape))) if with_res else None
    dz_ref = ops.conv_dgrad(dy, W, xb.shape, s)
    dx_ref, dg_ref, db_ref = ops.bn_relu_backward(dz_ref, xb, g, b, mean, rstd, res=res)
    dz, sums = ops.conv_dgrad(dy, W, xb.shape, s, bn=(xb, g, b, mean, rstd))
    assert sums is not None and torch.equal(dz, dz_ref)
    dx, dg, db = ops.bn_relu_backward(dz, xb, g, b, mean, rstd, res=res, dgrad_sums=sums)
    # float64 reference of the two sums
    live = (z > 0).double()
    gq = dz_ref.double() * live
    xh = (xb.double() - mean.double().view(1, -1, 1, 1)) * rstd.double().view(1, -1, 1, 1)
    s1, s2 = gq.sum(dim=(0, 2, 3)), (gq * xh).sum(dim=(0, 2, 3))
    mag1 = gq.abs().sum(dim=(0, 2, 3)).clamp_min(1e-30)
    mag2 = (gq * xh).abs().sum(dim=(0, 2, 3)).clamp_min(1e-30)
    assert ((db.double() - s1).abs() / mag1).max().item() <= 2e-6 and ((dg.double() - s2).abs() / mag2).max().item() <= 2e-6
    assert ((db_ref.double() - s1).abs() / mag1).max().item() <= 2e-6
    assert (dx - dx_ref).abs().max().item() <= 1e-5 * max(1.0, float(dx_ref.abs().max()))
    dz2, sums2 = ops.conv_dgrad(dy, W, xb.shape, s, bn=(xb, g, b, mean, rstd))
    assert torch.equal(sums2[0], sums[0])


def test_dgrad_epilogue_unsupported_shape_falls_back():
    xb = torch.randn(2, 8, 40, 36, device=DEV)
    W = torch.randn(12, 8, 3, 3, device=DEV)
    g, b = torch.ones(8, device=DEV), torch.zeros(8, device=DEV)
    z, mean, rstd = ops.bn_relu_forward(xb, g, b)
    dy = torch.randn(2, 12, 40, 36, device=DEV)
    dz, sums = ops.conv_dgrad(dy, W, xb.shape, 1, bn=(xb, g, b, mean, rstd))
    assert sums is None and torch.equal(dz, ops.conv_dgrad(dy, W, xb.shape, 1))
    assert ops._lib.lib().aq_conv_dgrad_bn_bytes(2, 8, 40, 36, 12, 3, 1, 1) == 0


@pytest.mark.parametrize("shape", [(16, 16, 1, 32, 128), (16, 32, 2, 32, 128), (32, 32, 1, 16, 128), (32, 64, 2, 16, 128), (64, 64, 1, 8, 128),
                                   (16, 16, 1, 32, 3), (64, 64, 1, 8, 1), (32, 32, 1, 16, 5), (20, 8, 1, 16, 7), (8, 12, 1, 40, 2)])
@pytest.mark.parametrize("with_bn", [False, True])
def test_conv_backward_one_call_equals_separate_calls(shape, with_bn):
    """aq_conv_bwd (input gradient and weight-gradient slabs side by side in one launch where a kernel pair
    exists) gives the bits of aq_conv_dgrad[_bn] and the sums of aq_conv_wgrad_partial."""
    cin, cout, s, H, N = shape
    rs = np.random.RandomState(sum(shape) + 3 * with_bn)
    Wd = 36 if H == 40 else H
    xb = cu(rs.standard_normal((N, cin, H, Wd)) * 1.3 + 0.2)
    W = cu(rs.standard_normal((cout, cin, 3, 3)) * 0.1)
    g, b = cu(1.0 + 0.2 * rs.standard_normal(cin)), cu(0.2 * rs.standard_normal(cin))
    z, mean, rstd = ops.bn_relu_forward(xb, g, b)
    y = ops.conv_forward(z, W, s)
    dy = cu(rs.standard_normal(tuple(y.shape)))
    bn = (xb, g, b, mean, rstd) if with_bn else None
    dW_ref = ops.conv_wgrad(z, dy, s)
    if with_bn:
        dz_ref, sums_ref = ops.conv_dgrad(dy, W, z.shape, s, bn=bn)
    else:
        dz_ref, sums_ref = ops.conv_dgrad(dy, W, z.shape, s), None
    for _ in range(2):
        dz, sums, ws, nslab = ops.conv_backward(dy, W, z, s, bn=bn)
        dW = torch.empty_like(W)
        ops.conv_wgrad_reduce([(ws, dW, nslab)])
        assert torch.equal(dz, dz_ref)
        # (the separate weight gradient of shapes with 16-channel multiples runs on another kernel: same sums, other order)
        assert (dW - dW_ref).abs().max().item() <= 2e-6 * dW_ref.abs().max().item()
        assert (sums is None) == (sums_ref is None)
        if sums is not None:  # the sums may be split over another number of workgroups: compare what they merge to
            got = ops.bn_relu_backward(dz, xb, g, b, mean, rstd, dgrad_sums=sums)
            want = ops.bn_relu_backward(dz_ref, xb, g, b, mean, rstd, dgrad_sums=sums_ref)
            for u, v in zip(got, want):
                assert (u - v).abs().max().item() <= 1e-5 * max(1.0, float(v.abs().max()))


# ------------------------------------------------------------------------------------ loss tail

@pytest.mark.parametrize("shape", [(128, 10), (1, 10), (300, 7), (5, 1), (64, 100)])
def test_ce_loss_and_sumsq_vs_oracle_and_torch(shape):
    """fp32 exp/log against the float64 oracle: 2e-6 relative on the loss, 5e-7 absolute on the gradient
    (entries are at most 1/N in magnitude)."""
    rs = np.random.RandomState(sum(shape))
    N, Cn = shape
    x = (rs.standard_normal(shape) * 3).astype(np.float32)
    y = rs.randint(0, Cn, N)
    a = [cu(rs.standard_normal(n)) for n in (1, 5, 432, 16, 36864, 64)]
    ssa = ops.sumsq(a)
    for t, s in zip(a, ssa.tolist()):
        assert abs(s - float((t.double() ** 2).sum())) <= 2e-6 * float((t.double() ** 2).sum())
    ssb = cu(rs.uniform(0, 2, 19))
    vals, d = ops.ce_loss(cu(x), torch.from_numpy(y).cuda(), 5e-4, ssa, ssb)
    loss, dref = O.ce_loss(x, y)
    ce, reg, tot = vals.tolist()
    assert abs(ce - loss) <= 2e-6 * max(1.0, abs(loss))
    want_reg = 5e-4 * (float(ssa.double().sum()) + float(ssb.double().sum()))
    assert abs(reg - want_reg) <= 2e-6 * want_reg and abs(tot - (ce + reg)) <= 1e-6 * abs(tot)
    assert np.abs(d.cpu().numpy() - dref).max() <= 5e-7
    # the stock operator agrees
    xt = cu(x).requires_grad_(True)
    lt = torch.nn.functional.cross_entropy(xt, torch.from_numpy(y).cuda())
    lt.backward()
    assert abs(float(lt) - ce) <= 2e-6 * max(1.0, abs(ce)) and (xt.grad - d).abs().max().item() <= 5e-7
    # value only; a label out of range poisons the loss
    _, dh = ops.ce_loss(cu(x), torch.from_numpy(y).cuda(), grad_scale=0.5)  # a power of two: exactly half
    assert torch.equal(dh, d * 0.5)
    v2, d2 = ops.ce_loss(cu(x), torch.from_numpy(y).cuda(), need_grad=False)
    assert d2 is None and v2[0].item() == ce and v2[1].item() == 0.0
    bad = torch.from_numpy(y).cuda().clone()
    bad[0] = Cn
    assert np.isnan(ops.ce_loss(cu(x), bad)[0][0].item())


# ------------------------------------------------------------------------------ small-tensor Adam

def test_adam_step_vs_oracle_and_torch():
    """aq_adam_step on a mixed bag of small tensors (more than one launch's worth): against the C oracle's
    Adam and against torch.optim.Adam(fused=True) fed the same gradients; reg adds reg*p to the gradient."""
    rs = np.random.RandomState(5)
    sizes = [32] * 20 + [432, 10, 1, 7] + [16, 16, 32, 32, 64, 64] * 6 + [36864]
    ps = [cu(rs.standard_normal(n)) for n in sizes]
    regs = [2e-4 if i % 3 == 0 else 0.0 for i in range(len(sizes))]
    ref = [torch.nn.Parameter(p.clone()) for p in ps]
    opt = torch.optim.Adam(ref, lr=1e-3, fused=True)
    m = [torch.zeros_like(p) for p in ps]
    v = [torch.zeros_like(p) for p in ps]
    steps = [torch.zeros((), device=DEV) for _ in ps]
    op, om, ov = [p.cpu().numpy().copy() for p in ps], [np.zeros(n, np.float32) for n in sizes], [np.zeros(n, np.float32) for n in sizes]
    for t in range(1, 4):
        gs = [cu(rs.standard_normal(n) * 0.1) for n in sizes]
        for r, g, reg in zip(ref, gs, regs):
            r.grad = g + reg * r.detach()
        opt.step()
        for i, g in enumerate(gs):
            gi = (g.cpu().numpy() + np.float32(regs[i]) * op[i]).astype(np.float32)
            O.adam_step(op[i], gi, om[i], ov[i], t, 1e-3)
        g_in = [g.clone() for g in gs]
        ops.adam_step(ps, g_in, m, v, steps, 1e-3, regs=regs)
        for i in range(len(sizes)):
            if regs[i]:
                assert (g_in[i] - ref[i].grad).abs().max().item() <= 1e-7 * max(1.0, float(ref[i].grad.abs().max()))
            else:
                assert torch.equal(g_in[i], gs[i])
            assert np.abs(ps[i].cpu().numpy() - op[i]).max() <= 2e-6, (t, i)
            assert (ps[i] - ref[i].detach()).abs().max().item() <= 2e-6, (t, i)
            assert float(steps[i]) == t
    st = opt.state[ref[-1]]
    assert torch.allclose(m[-1], st["exp_avg"], rtol=1e-4, atol=1e-8) and torch.allclose(v[-1], st["exp_avg_sq"], rtol=1e-4, atol=1e-10)


def test_adam_step_argument_errors():
    p = torch.zeros(4, device=DEV)
    st = torch.zeros((), device=DEV)
    with pytest.raises(RuntimeError, match="share a step counter"):
        ops.adam_step([p, p.clone()], [p.clone(), p.clone()], [p.clone(), p.clone()], [p.clone(), p.clone()], [st, st], 1e-3)
    with pytest.raises(RuntimeError, match="out of range"):
        ops.adam_step([p], [p.clone()], [p.clone()], [p.clone()], [st], 1e-3, betas=(1.0, 0.9))
    with pytest.raises(ValueError, match="elements"):
        ops.adam_step([p], [torch.zeros(3, device=DEV)], [p.clone()], [p.clone()], [st], 1e-3)


@pytest.mark.parametrize("shape", [(128, 16, 32, 32), (4, 32, 16, 16), (2, 3, 6, 10), (3, 5, 7, 9)])
def test_bn_backward_takes_the_pooled_shortcut_gradient(shape):
    """aq_bn_relu_bwd_shortcut == aq_shortcut_bwd followed by aq_bn_relu_bwd(res) (bit for bit when both walk the
    tensor the same way, i.e. for widths that are multiples of 4)."""
    N, Cn, H, Wd = shape
    rs = np.random.RandomState(sum(shape))
    x = cu(rs.standard_normal(shape) * 1.2 + 0.1)
    g, b = cu(1.0 + 0.2 * rs.standard_normal(Cn)), cu(0.2 * rs.standard_normal(Cn))
    _, mean, rstd = ops.bn_relu_forward(x, g, b)
    dy = cu(rs.standard_normal(shape))
    front, back = Cn // 2, Cn - Cn // 2
    ds = cu(rs.standard_normal((N, Cn + front + back, H // 2, Wd // 2)))
    res = ops.shortcut_backward(ds, shape, front, back)
    want = ops.bn_relu_backward(dy, x, g, b, mean, rstd, res=res)
    got = ops.bn_relu_backward(dy, x, g, b, mean, rstd, shortcut=(ds, front, back))
    for u, v in zip(got, want):
        if Wd % 4 == 0:
            assert torch.equal(u, v)
        else:  # rows of another width are walked element by element, the reference call four at a time: other partial sums
            assert (u - v).abs().max().item() <= 2e-6 * max(1.0, float(v.abs().max()))
    with pytest.raises(ValueError, match="does not belong"):
        ops.bn_relu_backward(dy, x, g, b, mean, rstd, shortcut=(ds, front + 1, back))


@pytest.mark.parametrize("shape", [(128, 16, 32, 32), (4, 32, 16, 16), (2, 3, 6, 10), (3, 5, 7, 9)])
def test_bn_forward_also_writes_the_shortcut(shape):
    N, Cn, H, Wd = shape
    rs = np.random.RandomState(sum(shape) + 1)
    x = cu(rs.standard_normal(shape) * 1.2 + 0.1)
    g, b = cu(1.0 + 0.2 * rs.standard_normal(Cn)), cu(0.2 * rs.standard_normal(Cn))
    front, back = Cn // 2, Cn - Cn // 2
    rm1, rv1 = torch.zeros(Cn, device=DEV), torch.ones(Cn, device=DEV)
    rm2, rv2 = rm1.clone(), rv1.clone()
    y1, m1, r1 = ops.bn_relu_forward(x, g, b, rm1, rv1, None)
    y2, m2, r2, s = ops.bn_relu_forward(x, g, b, rm2, rv2, None, shortcut=(front, back))
    assert torch.equal(y1, y2) and torch.equal(m1, m2) and torch.equal(r1, r2) and torch.equal(rm1, rm2) and torch.equal(rv1, rv2)
    assert torch.equal(s, ops.shortcut_forward(x, front, back))
    want = torch.nn.functional.pad(torch.nn.functional.avg_pool2d(x, 2), (0, 0, 0, 0, front, back))
    assert torch.equal(s, want)


@pytest.mark.parametrize("shape", [(16, 16, 1, 32, 128), (16, 32, 2, 32, 4), (64, 64, 1, 8, 3), (8, 80, 1, 40, 1), (5, 3, 2, 6, 2)])
def test_conv_dgrad_adds_the_shortcut_gradient(shape):
    """aq_conv_dgrad_add == aq_conv_dgrad + res with the addition as the last operation (bit-equal to doing it
    afterwards); res may alias the output."""
    cin, cout, s, H, N = shape
    rs = np.random.RandomState(sum(shape))
    Wd = 36 if H == 40 else H
    W = cu(rs.standard_normal((cout, cin, 3, 3)) * 0.1)
    xs = (N, cin, H, Wd)
    dy = cu(rs.standard_normal(ops.conv_out_shape(xs, cout, s)))
    res = cu(rs.standard_normal(xs))
    want = ops.conv_dgrad(dy, W, xs, s) + res
    assert torch.equal(ops.conv_dgrad(dy, W, xs, s, res=res), want)
    with pytest.raises(ValueError, match="excludes bn"):
        ops.conv_dgrad(dy, W, xs, s, res=res, bn=(res, None, None, None, None))


@pytest.mark.parametrize("shape", [(16, 16, 1, 32, 5), (32, 64, 2, 16, 3), (64, 64, 1, 8, 9), (128, 64, 1, 14, 2), (3, 16, 1, 32, 2), (20, 24, 1, 9, 3)],
                         ids=lambda s: "x".join(map(str, s)))
def test_grouped_weight_gradient_equals_single_launches(shape):
    """aq_conv_wgrad_partial_group (layers of one shape in one launch; what the fused training step runs at the end of
    the backward pass) against one aq_conv_wgrad per layer (autograd of adp_qtz_fan.py:574-601): the same sums, cut
    into fewer slabs when several layers share the launch, so equal within fp32 summation order (2e-6 of the largest
    element; bit-equal on the kernels whose slab count does not depend on the group), and reproducible bit for bit."""
    cin, cout, stride, H, nl = shape
    g = torch.Generator(device=DEV).manual_seed(11)
    xs = [torch.randn(6, cin, H, H, device=DEV, generator=g) for _ in range(nl)]
    Ho = (H + 2 - 3) // stride + 1
    dys = [torch.randn(6, cout, Ho, Ho, device=DEV, generator=g) for _ in range(nl)]

    def grouped():
        res = ops.conv_wgrad_partial_group(xs, dys, stride)
        outs = [torch.empty(cout, cin, 3, 3, device=DEV) for _ in range(nl)]
        ops.conv_wgrad_reduce([(ws, o, ns) for (ws, ns), o in zip(res, outs)])
        return outs
    outs, again = grouped(), grouped()
    for x, dy, o, o2 in zip(xs, dys, outs, again):
        want = ops.conv_wgrad(x, dy, stride)
        assert (o - want).abs().max().item() <= 2e-6 * want.abs().max().item()
        assert torch.equal(o, o2)
        if cin % 16 or cout % 16:
            assert torch.equal(o, want)
