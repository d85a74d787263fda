"""Host-side logic of the drop-in modules on the CPU (no kernels run): level initialisers of the product's own
MyConv2d / MyLinear against the values the reference's classes produced (tests/golden/levels.npz), and the
per-epoch beta / learning-rate schedule against the reference's loop head (adp_qtz_fan.py:268-278)."""
import argparse

import numpy as np
import torch


def _cfg(L, liar=True):
    return argparse.Namespace(mode="train", prim_init="uniform", num_level_conv=L, num_level_fc=L, l2_reg=5e-4, liar=liar)


def test_product_level_initialisers_match_reference(golden):
    import quantized_cnn_amd.adp_qtz_fix as fix
    import quantized_cnn_amd.adp_qtz_hint as hint
    g = golden("levels")
    v = torch.zeros(10)
    n = int(g["ncase"])
    seen = set()
    for i in range(n):
        s, L, liar = g["arg_%02d" % i]
        s, L, liar = float(s), int(L), bool(liar)
        seen.add((L, liar))
        c = _cfg(L, liar)
        conv, fc = fix.MyConv2d(c, s, 2, 2, 3, 1, 1), fix.MyLinear(c, s, v, 4, 10)
        assert conv.q_level.shape == (L - 1,) and not conv.q_level.requires_grad and conv.p_c.shape == (2, 2, 3, 3, L - 1)
        assert fc.p_c.shape == (4, 10, L - 1) and not fc.q_level.requires_grad
        assert np.array_equal(conv.q_level.detach().numpy(), g["fix_%02d" % i]), (s, L, liar)
        assert np.array_equal(fc.q_level.detach().numpy(), g["fixfc_%02d" % i]), (s, L, liar)
        hconv, hfc = hint.MyConv2d(c, s, 2, 2, 3, 1, 1), hint.MyLinear(c, s, v, 4, 10)
        assert hconv.q_level.requires_grad and hconv.p_c.shape == (2, 2, 3, 3, L)
        assert np.array_equal(hconv.q_level.detach().numpy(), g["hint_%02d" % i]), (s, L, liar)
        assert np.array_equal(hfc.q_level.detach().numpy(), g["hintfc_%02d" % i]), (s, L, liar)
    assert seen == {(32, True), (32, False), (8, True), (8, False)}


def test_fan_level_bound_and_bias_init(golden):
    import quantized_cnn_amd.adp_qtz_fan as fan
    g = golden("levels")
    torch.manual_seed(0)
    for fan_in, obs in g["fan_bound_obs"]:
        cin = int(fan_in) // 9 if int(fan_in) != 64 else None
        m = fan.MyConv2d(_cfg(4096), cin, 1, 3, 1, 1) if cin else fan.MyLinear(_cfg(4096), 64, 1)
        b = np.sqrt(3.0) * np.sqrt(2.0 / 6.0) / np.sqrt(fan_in)  # adp_qtz_fan.py:551-560
        mx = float(m.q_level.detach().abs().max())
        assert mx <= b * (1 + 1e-6) and mx >= 0.99 * b and abs(obs - b) <= 0.01 * b
    lin = fan.MyLinear(_cfg(32), 64, 10)
    assert float(lin.bias.abs().max()) <= 1 / 8.0  # U(-1/sqrt(64), 1/sqrt(64)), adp_qtz_fan.py:629-631


def _reference_loop_head(epoch, beta, lrs, beta_list, epoch_list):
    """adp_qtz_fan.py:269-278 restated: one if / elif chain."""
    if epoch == epoch_list[1]:
        beta = beta_list[1]
    elif epoch == epoch_list[2]:
        beta = beta_list[2]
    elif epoch == epoch_list[3]:
        beta = beta_list[3]
    elif epoch == epoch_list[4]:
        lrs = [lr / 5 for lr in lrs]
    return beta, lrs


def test_beta_schedule_follows_the_reference_loop():
    from quantized_cnn_amd.train import beta_schedule
    for beta_list, epoch_list, epochs in (
            ("[1, 2, 3, 4]", "[0, 70, 140, 210, 300]", 350),     # the reference's defaults (adp_qtz_fan.py:123-129)
            ("[1, 2, 3, 4]", "[0, 2, 2, 5, 5]", 8),              # epochs listed twice: first match only
            ("[2,5,7,9]", "[0,1,2,3,4]", 6)):
        c = argparse.Namespace(beta_list=beta_list, epoch_list=epoch_list)
        bl = [int(v) for v in beta_list.strip("[]").split(",")]
        el = [int(v) for v in epoch_list.strip("[]").split(",")]
        params = [torch.nn.Parameter(torch.zeros(1)) for _ in range(3)]
        opts = [torch.optim.Adam([p], lr=lr) for p, lr in zip(params, (1e-2, 1e-3, 1e-2))]
        beta, ref_beta, ref_lrs = bl[0], bl[0], [1e-2, 1e-3, 1e-2]
        for epoch in range(epochs):
            beta = beta_schedule(c, epoch, beta, opts)
            ref_beta, ref_lrs = _reference_loop_head(epoch, ref_beta, ref_lrs, bl, el)
            assert beta == ref_beta, (epoch, beta, ref_beta)
            assert [o.param_groups[0]["lr"] for o in opts] == ref_lrs, epoch


def test_ternary_and_inq_oracle_known_answers():
    """The CPU restatement of TWN / TTQ / INQ (parity unpinned: written from the cited papers, no reference code)
    on hand-computed cases."""
    from oracle import oracle as O
    w = np.array([0.5, -0.1, 0.05, -0.6, 0.0, 0.3], np.float32)
    r = O.tern_forward(w, "twn")                     # mean|w| = 1.55 / 6, delta = 0.7 * that = 0.18083
    assert abs(r["delta"] - 0.7 * 1.55 / 6) < 1e-7
    assert list(r["codes"]) == [1, 0, 0, 2, 0, 1]
    assert abs(r["pos"] - (0.5 + 0.6 + 0.3) / 3) < 1e-7 and r["pos"] == r["neg"]
    assert np.allclose(r["Wt"], np.array([1, 0, 0, -1, 0, 1]) * r["pos"])
    t = O.tern_forward(w, "ttq", 0.05, True, 0.7, 0.2)   # delta = 0.05 * max|w| = 0.03
    assert list(t["codes"]) == [1, 2, 1, 2, 0, 1] and np.allclose(t["Wt"], [0.7, -0.2, 0.7, -0.2, 0, 0.7])
    g = np.array([1.0, 2.0, 3.0, 4.0, 5.0, 6.0], np.float32)
    dw, dp, dn = O.tern_backward(w, g, "ttq", t["delta"], 0.7, 0.2)
    assert dp == 1 + 3 + 6 and dn == -(2 + 4)
    assert np.allclose(dw, [0.7, 0.4, 2.1, 0.8, 5.0, 4.2])
    # INQ, 5 bits, max|w| = 0.6: n1 = floor(log2(0.8)) = -1, n2 = n1 - 7 = -8
    n1, n2 = O.inq_exponents(0.6, 5)
    assert (n1, n2) == (-1, -8)
    v = np.array([0.6, 0.374, 0.375, -0.2, 0.0029, 0.0019, 0.0, -0.74, 0.75], np.float32)
    q, m, c = O.inq_partition(v, np.ones(9, np.uint8), 0.0, n1, n2)
    #  0.6 -> 0.5 (in [0.375, 0.75)); 0.374 -> 0.25; 0.375 -> 0.5; -0.2 -> -0.25 (in [0.1875, 0.375));
    #  0.0029 -> 2^-8 (in [2^-9, 1.5 * 2^-8)); 0.0019 -> 0 (< 2^-9 = 0.00195); 0.75 -> clamps to 2^n1 = 0.5
    assert list(q) == [0.5, 0.25, 0.5, -0.25, 2.0 ** -8, 0.0, 0.0, -0.5, 0.5]
    assert m.sum() == 0 and list(c) == [15, 13, 15, 14, 1, 0, 0, 16, 15]
    q2, m2, c2 = O.inq_partition(v, np.ones(9, np.uint8), 0.37, n1, n2)   # only magnitudes >= 0.37 join
    assert list(m2) == [0, 0, 0, 1, 1, 1, 1, 0, 0] and q2[3] == np.float32(-0.2) and c2[3] == 255


def test_pkt_oracle_properties():
    """oracle.pkt_loss (parity unpinned, restated from Passalis & Tefas 2018): zero for features with the same cosine
    geometry, positive otherwise, invariant to a rescaling of rows."""
    from oracle import oracle as O
    rs = np.random.RandomState(0)
    t = rs.standard_normal((40, 8))
    assert abs(O.pkt_loss(t * 3.0, t)) < 1e-12
    s = rs.standard_normal((40, 5))
    a = O.pkt_loss(s, t)
    assert a > 0 and abs(O.pkt_loss(s * rs.uniform(0.5, 2.0, (40, 1)), t) - a) < 1e-9
    # two points: K = [[1, k], [k, 1]] with k = (cos + 1) / 2; orthogonal student, identical teacher rows
    s2, t2 = np.array([[1.0, 0.0], [0.0, 1.0]]), np.array([[1.0, 0.0], [2.0, 0.0]])
    ps = np.array([[1, .5], [.5, 1]]) / 1.5
    pt = np.full((2, 2), 0.5)
    assert abs(O.pkt_loss(s2, t2) - np.mean(pt * np.log((pt + 1e-7) / (ps + 1e-7)))) < 1e-7  # (eps also sits in the normalisation)
