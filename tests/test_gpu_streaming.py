"""The large-tensor instantiations of the synthesis kernels against the CPU oracle.

Calls that move more than 128 MiB switch aq_synth_fwd / aq_synth_bwd / aq_synth_bwd_adam to their streaming
instantiations (nontemporal 16-byte accesses; for row lengths that are no multiple of 4 the LDS-staged
nontemporal stores).  These are the kernels behind the standalone 25 M .. 1 B-coefficient sweep and the
roofline figure, so they get the same bars as the small cases: W / codes / dp_c bit-exact in exact
arithmetic, 2e-6 (W) and 3e-5 of the row scale (dp_c) in fast arithmetic, cross-row sums at rel 2e-5.
The 1 B-coefficient case (4 GB of p_c: row offsets past 2^31 bytes, every level of the partial-sum tree)
is checked by properties: spot windows against the oracle, run-to-run equality, finiteness.
"""
import numpy as np
import pytest
import torch

from oracle import oracle as O

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    import quantized_cnn_amd.ops as ops
DEV = "cuda"
STREAM_BYTES = 128 << 20  # the library's switch: bytes of p_c per call (x6 with the fused Adam update)


def cu(a, dtype=torch.float32):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV, dtype)


def bits_equal(got, want):
    g = np.ascontiguousarray(got, dtype=np.float32).view(np.uint32)
    w = np.ascontiguousarray(want, dtype=np.float32).view(np.uint32)
    return bool(np.all((g == w) | (np.isnan(got) & np.isnan(want))))


def levels(L, s=0.3):
    return O.levels_hint(s, L) if L % 2 == 0 else O.levels_fix(s, L + 1)


def make_case(L, rows, seed):
    rs = np.random.RandomState(seed)
    p = rs.random_sample((rows, L)).astype(np.float32)
    dW = rs.standard_normal(rows).astype(np.float32)
    return p, levels(L), dW


@pytest.mark.parametrize("L", [32, 31])
def test_streaming_forward_vs_oracle(L):
    rows = 1_200_000 if L == 32 else 1_240_000
    p, q, _ = make_case(L, rows, 11 + L)
    assert p.nbytes > STREAM_BYTES
    beta = 4.0
    ref = O.synth_fwd(p, q, beta)
    pd, qd = cu(p), cu(q)
    hard = ops.synth_forward([pd], [qd], beta, soft=True, codes=True, sumsq=True)
    assert bits_equal(hard["W"][0].cpu().numpy(), ref["W"])
    assert np.array_equal(hard["idx"][0].cpu().numpy().astype(np.int32), ref["idx"])
    assert bits_equal(hard["W_hard"][0].cpu().numpy(), ref["W_hard"])
    assert abs(float(hard["sumsq"][0]) - ref["sumsq"]) <= 2e-5 * ref["sumsq"]
    only = ops.synth_forward([pd], [qd], beta, soft=False, codes=True)  # codes alone: the arg-max-of-p instantiation
    assert torch.equal(only["idx"][0], hard["idx"][0]) and torch.equal(only["W_hard"][0], hard["W_hard"][0])
    soft = ops.synth_forward([pd], [qd], beta, soft=True, sumsq=True, exact=True)
    assert bits_equal(soft["W"][0].cpu().numpy(), ref["W"])
    assert torch.equal(soft["sumsq"], hard["sumsq"])
    fast = ops.synth_forward([pd], [qd], beta, soft=True, sumsq=True, exact=False)
    Wf = fast["W"][0].cpu().numpy()
    assert np.all(np.abs(Wf - ref["W"]) <= 2e-6 * np.abs(q).max() + 2e-6 * np.abs(ref["W"]))
    assert abs(float(fast["sumsq"][0]) - ref["sumsq"]) <= 2e-5 * ref["sumsq"]


@pytest.mark.parametrize("L", [32, 31])
def test_streaming_backward_vs_oracle(L):
    rows = 1_200_000 if L == 32 else 1_240_000
    p, q, dW = make_case(L, rows, 23 + L)
    beta, wd = 4.0, 1e-3
    dp_ref, dq_ref = O.synth_bwd(p, q, beta, dW, wd)
    pd, qd, gd = cu(p), cu(q), cu(dW)
    dp, dq = ops.synth_backward([pd], [qd], beta, [gd], wd, exact=True)
    assert bits_equal(dp[0].cpu().numpy(), dp_ref)
    assert np.abs(dq[0].cpu().numpy() - dq_ref).max() <= 2e-5 * np.abs(dq_ref).max()
    dp, dq = ops.synth_backward([pd], [qd], beta, [gd], wd, exact=False)
    got = dp[0].cpu().numpy()
    tol = 3e-5 * np.abs(dp_ref).max(axis=1, keepdims=True) + 1e-6 * np.abs(dp_ref).max()
    assert np.all(np.abs(got - dp_ref) <= tol)
    assert np.abs(dq[0].cpu().numpy() - dq_ref).max() <= 2e-5 * np.abs(dq_ref).max()


def test_streaming_mixed_layers_in_one_call():
    """Layers of 32, 31 and 64 levels in one streaming call: the 32-level layers run the 32-rows-per-wavefront forms
    of the fast-arithmetic kernels, the others 64 rows, so their partial sums are laid out per layer; row counts
    that are no multiple of either (tails), fast and exact arithmetic."""
    shapes = [(32, 700_037), (31, 300_011), (64, 100_003), (32, 300_001)]
    beta, wd = 2.0, 1e-3
    cases = [make_case(L, rows, 71 + i) for i, (L, rows) in enumerate(shapes)]
    assert sum(c[0].nbytes for c in cases) > STREAM_BYTES
    pd, qd, gd = [cu(c[0]) for c in cases], [cu(c[1]) for c in cases], [cu(c[2]) for c in cases]
    refs = [O.synth_fwd(c[0], c[1], beta) for c in cases]
    brefs = [O.synth_bwd(c[0], c[1], beta, c[2], wd) for c in cases]
    for exact in (False, True):
        out = ops.synth_forward(pd, qd, beta, soft=True, sumsq=True, exact=exact)
        dp, dq = ops.synth_backward(pd, qd, beta, gd, wd, exact=exact)
        for i, (c, ref, (dp_ref, dq_ref)) in enumerate(zip(cases, refs, brefs)):
            W = out["W"][i].cpu().numpy()
            got = dp[i].cpu().numpy()
            if exact:
                assert bits_equal(W, ref["W"]), i
                assert bits_equal(got, dp_ref), i
            else:
                assert np.all(np.abs(W - ref["W"]) <= 2e-6 * np.abs(c[1]).max() + 2e-6 * np.abs(ref["W"])), i
                tol = 3e-5 * np.abs(dp_ref).max(axis=1, keepdims=True) + 1e-6 * np.abs(dp_ref).max()
                assert np.all(np.abs(got - dp_ref) <= tol), i
            assert abs(float(out["sumsq"][i]) - ref["sumsq"]) <= 2e-5 * ref["sumsq"], i
            assert np.abs(dq[i].cpu().numpy() - dq_ref).max() <= 2e-5 * np.abs(dq_ref).max(), i


@pytest.mark.parametrize("L", [32, 31])
@pytest.mark.parametrize("exact", [True, False])
def test_streaming_backward_adam_vs_oracle(L, exact):
    """p, m, v read and written: the switch counts 6 passes, so 540 000 rows (69 MB of p) are past it, and past the
    row count from which 32-level layers run 32 rows per wavefront; p alone stays below the switch, so the plain
    gradient call on the same rows is the cached 64-row instantiation."""
    rows = 540_000 if L == 32 else 540_004
    p0, q, _ = make_case(L, rows, 37 + L)
    assert p0.nbytes * 6 > STREAM_BYTES
    rs = np.random.RandomState(5)
    lr, b1, b2, eps, beta, wd = 1e-2, 0.9, 0.999, 1e-8, 2.0, 1e-3
    p, m, v = cu(p0), torch.zeros(rows, L, device=DEV), torch.zeros(rows, L, device=DEV)
    step = torch.zeros((), device=DEV)
    po, mo, vo = p0.copy(), np.zeros_like(p0), np.zeros_like(p0)
    qd = cu(q)
    for t in (1, 2):
        dW = rs.standard_normal(rows).astype(np.float32)
        gd = cu(dW)
        g_small = ops.synth_backward([p.clone()], [qd], beta, [gd], wd, exact=exact)  # 69 MB: the cached instantiation
        g_dev = torch.empty_like(p)
        dq = ops.synth_backward_adam([p], [qd], beta, [gd], [m], [v], [step], lr, (b1, b2), eps, wd=wd, exact=exact, out_dp=[g_dev])
        assert float(step) == t
        assert torch.equal(g_dev, g_small[0][0])  # streaming and cached instantiations: same bits
        # (dq is a sum over rows: the two instantiations may group their per-wavefront partials differently, 64 against
        # 32 rows per wavefront, so the sums agree to rounding, not bit for bit)
        assert float((dq[0] - g_small[1][0]).abs().max()) <= 2e-6 * float(g_small[1][0].abs().max())
        g_or, dq_or = O.synth_bwd(po, q, beta, dW, wd)
        if exact and t == 1:
            assert bits_equal(g_dev.cpu().numpy(), g_or)
        assert np.abs(dq[0].cpu().numpy() - dq_or).max() <= 2e-5 * np.abs(dq_or).max()
        O.adam_step(po, g_or, mo, vo, t, lr, b1, b2, eps)
        d = np.abs(p.cpu().numpy() - po)
        assert d.max() <= 2.0 * lr * t * 1.01          # no coordinate further than Adam can move it
        assert (d > 1e-6).mean() <= 0.01, (t, (d > 1e-6).mean())
        assert np.abs(m.cpu().numpy() - mo).max() <= (1e-5 if exact else 1e-4) * max(np.abs(mo).max(), 1e-6)
        assert np.abs(v.cpu().numpy() - vo).max() <= (1e-5 if exact else 1e-4) * max(np.abs(vo).max(), 1e-12)


def _windows(rows, width=4096):
    """Row windows of a large tensor: both ends, the rows around byte offsets 2^31 and 2^32 of a 128-byte-row
    tensor, and a misaligned stretch in the middle."""
    starts = [0, (1 << 24) - width // 2, (1 << 25) - width // 2, rows // 2 + 37, rows - width]
    return [(s, min(s + width, rows)) for s in starts if 0 <= s < rows]


@pytest.mark.parametrize("L", [32, 31])
def test_one_billion_coefficients_properties(L):
    """The largest size of the standalone sweep: 31 250 000 rows x 32 (4 GB of p_c; L = 31: 3.9 GB)."""
    rows = 31_250_000
    g = torch.Generator(device=DEV).manual_seed(1234 + L)
    p = torch.rand(rows, L, device=DEV, generator=g)
    dW = torch.randn(rows, device=DEV, generator=g) + 0.5  # non-zero mean: the sums over rows are large against their rounding error
    q = levels(L)
    qd = cu(q)
    beta, wd = 4.0, 1e-3
    hard = ops.synth_forward([p], [qd], beta, soft=True, codes=True, sumsq=True)
    again = ops.synth_forward([p], [qd], beta, soft=True, codes=True, sumsq=True)
    assert torch.equal(hard["W"][0], again["W"][0]) and torch.equal(hard["idx"][0], again["idx"][0])
    assert torch.equal(hard["sumsq"], again["sumsq"])
    del again
    only = ops.synth_forward([p], [qd], beta, soft=False, codes=True)
    assert torch.equal(only["idx"][0], hard["idx"][0]) and torch.equal(only["W_hard"][0], hard["W_hard"][0])
    del only
    fast = ops.synth_forward([p], [qd], beta, soft=True, sumsq=True, exact=False)
    assert bool(torch.isfinite(fast["W"][0]).all())
    assert float((fast["W"][0] - hard["W"][0]).abs().max()) <= 4e-6 * float(np.abs(q).max())
    # sum W^2 against a float64 sum of the kernel's own W: pins every level of the partial-sum tree
    ss64 = float((hard["W"][0].double() ** 2).sum())
    assert abs(float(hard["sumsq"][0]) - ss64) <= 2e-5 * ss64
    assert abs(float(fast["sumsq"][0]) - ss64) <= 2e-5 * ss64
    # the arg-max of s_c is the arg-max of p_c (normalisation and softmax are monotone for beta > 0);
    # rows whose maximum leads by more than rounding can blur (two p a few ulps apart may give the same exp)
    # must agree everywhere
    top2 = p.topk(2, dim=1)
    unique = top2.values[:, 0] - top2.values[:, 1] > 1e-5
    assert bool((hard["idx"][0].long()[unique] == top2.indices[:, 0][unique]).all())
    del top2, unique
    dp = torch.empty_like(p)
    _, dq = ops.synth_backward([p], [qd], beta, [dW], wd, exact=True, out_dp=[dp])
    dp2 = torch.empty_like(p)
    _, dq2 = ops.synth_backward([p], [qd], beta, [dW], wd, exact=True, out_dp=[dp2])
    assert torch.equal(dq[0], dq2[0])
    same = True
    for a in range(0, rows, 1 << 22):  # chunked: torch.equal on 4 GB allocates nothing, but keep peaks low
        same = same and torch.equal(dp[a:a + (1 << 22)], dp2[a:a + (1 << 22)])
    assert same
    del dp2
    assert bool(torch.isfinite(dq[0]).all())
    for a, b in _windows(rows):
        pw = p[a:b].cpu().numpy()
        ref = O.synth_fwd(pw, q, beta)
        assert bits_equal(hard["W"][0][a:b].cpu().numpy(), ref["W"]), (a, b)
        assert np.array_equal(hard["idx"][0][a:b].cpu().numpy().astype(np.int32), ref["idx"]), (a, b)
        dp_ref, _ = O.synth_bwd(pw, q, beta, dW[a:b].cpu().numpy(), wd)
        assert bits_equal(dp[a:b].cpu().numpy(), dp_ref), (a, b)
    # dq = sum_rows s_c * g: the gradient identity sum_l dq_l * q_l = sum_rows g * W pins the whole tree
    gW = hard["W"][0].double()
    gtot = float(((dW.double() + wd * gW) * gW).sum())
    lhs = float((dq[0].double() * qd.double()).sum())
    assert abs(lhs - gtot) <= 2e-5 * max(abs(gtot), float((dq[0].double() * qd.double()).abs().sum()))


def test_three_level_partial_sum_tree():
    """70 M rows of 4 levels: 1.09 M per-wavefront partials of dq, more than one block of the second level can
    hold, so the fixed-order sum runs over three levels.  dq and sum W^2 against the oracle (float64)."""
    rows, L = 70_000_000, 4
    g = torch.Generator(device=DEV).manual_seed(77)
    p = torch.rand(rows, L, device=DEV, generator=g) + 0.1
    dW = torch.randn(rows, device=DEV, generator=g) + 0.5  # non-zero mean: the sums over rows are large against their rounding error
    q = np.array([-0.25, -0.0625, 0.125, 0.5], np.float32)
    qd = cu(q)
    beta, wd = 2.0, 0.0
    out = ops.synth_forward([p], [qd], beta, soft=True, sumsq=True, exact=True)
    dp, dq = ops.synth_backward([p], [qd], beta, [dW], wd, exact=True)
    pw, gw = p.cpu().numpy(), dW.cpu().numpy()
    ref = O.synth_fwd(pw, q, beta, want_idx=False)
    assert abs(float(out["sumsq"][0]) - ref["sumsq"]) <= 2e-5 * ref["sumsq"]
    _, dq_ref = O.synth_bwd(pw, q, beta, gw, wd, want_dp=False)
    assert np.abs(dq[0].cpu().numpy() - dq_ref).max() <= 2e-5 * np.abs(dq_ref).max() + 1e-3
    a, b = rows - 5000, rows
    dp_ref, _ = O.synth_bwd(pw[a:b], q, beta, gw[a:b], wd)
    assert bits_equal(dp[0][a:b].cpu().numpy(), dp_ref)
    assert bits_equal(out["W"][0][a:b].cpu().numpy(), ref["W"][a:b])
