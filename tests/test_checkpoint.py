"""Checkpoint interchange with the reference (tests/golden/ckpt_fan.pt was written by the reference's
model and optimizers with the dictionary layout of adp_qtz_fan.py:331-340) and the packed export."""
import argparse
import os

import numpy as np
import pytest
import torch

from conftest import ROOT
from oracle import oracle as O

CKPT = os.path.join(ROOT, "tests", "golden", "ckpt_fan.pt")


def small_cfg(mode="train"):
    return argparse.Namespace(mode=mode, prim_init="uniform", num_level_conv=4, num_level_fc=4, l2_reg=5e-4, liar=True,
                              prim_lr=1e-2, q_lr=1e-3, lr=1e-2)


def test_reference_checkpoint_loads_into_model_and_optimizers(tmp_path):
    import quantized_cnn_amd.adp_qtz_fan as fan
    from quantized_cnn_amd import checkpoint as ck
    from quantized_cnn_amd.train import make_optimizers
    model = fan.ResNet(small_cfg(), 1)
    opts = make_optimizers(model, small_cfg())
    beta, epoch, it, best = ck.load_checkpoint(CKPT, model, opts)
    assert (beta, epoch, it) == (2, 3, 1) and abs(best - 18.75) < 1e-6
    raw = ck._safe_load(CKPT)
    assert list(raw["model"].keys()) == list(model.state_dict().keys())
    for k, v in raw["model"].items():
        assert torch.equal(v, model.state_dict()[k]), k
    # Adam state: two steps taken, moments present for every p_c
    st = opts[0].state_dict()["state"]
    assert len(st) == 7 and all(float(s["step"]) == 2 for s in st.values())
    # round trip through this library's writer gives the same file content
    out = str(tmp_path / "again.pt")
    ck.save_checkpoint(out, model, opts, beta, epoch, it, best)
    again = ck._safe_load(out)
    assert set(again.keys()) == set(ck.CHECKPOINT_KEYS) == set(raw.keys())
    for k in ("optimizer_prim", "optimizer_q", "optimizer"):
        assert again[k]["param_groups"] == raw[k]["param_groups"]
        for i, s in raw[k]["state"].items():
            for name, t in s.items():
                assert torch.equal(torch.as_tensor(t), torch.as_tensor(again[k]["state"][i][name])), (k, i, name)
    ck.save_bestmodel(str(tmp_path / "best.pt"), model, beta)
    m2 = fan.ResNet(small_cfg(), 1)
    assert ck.load_bestmodel(str(tmp_path / "best.pt"), m2) == 2
    assert all(torch.equal(a, b) for a, b in zip(m2.state_dict().values(), model.state_dict().values()))
    with pytest.raises(KeyError, match="not a training checkpoint"):
        ck.load_checkpoint(str(tmp_path / "best.pt"), model)


@pytest.mark.parametrize("bits", [1, 2, 3, 4, 5, 6, 7, 8])
def test_oracle_pack_unpack_round_trip(bits):
    rs = np.random.RandomState(bits)
    for n in (0, 1, 7, 8, 9, 1000, 4099):
        idx = rs.randint(0, 1 << bits, n).astype(np.uint8)
        packed = O.pack_codes(idx, bits)
        assert packed.size == (n * bits + 7) // 8
        assert np.array_equal(O.unpack_codes(packed, n, bits), idx)
    # known answer, 5 bits: codes 1, 2, 3 -> bits 10000 01000 11000 (LSB first) -> bytes 0x41, 0x0c
    assert list(O.pack_codes(np.array([1, 2, 3], np.uint8), 5)) == [0x41, 0x0C]


@pytest.mark.gpu
@pytest.mark.parametrize("bits", [1, 3, 5, 7, 8])
def test_pack_kernels_bit_exact_vs_oracle(bits):
    import quantized_cnn_amd.ops as ops
    rs = np.random.RandomState(100 + bits)
    for n in (1, 7, 8, 9, 2047, 2048, 36864, 100003):
        idx = rs.randint(0, 1 << bits, n).astype(np.uint8)
        packed = ops.pack_codes(torch.from_numpy(idx).cuda(), bits)
        assert np.array_equal(packed.cpu().numpy(), O.pack_codes(idx, bits)), n
        assert np.array_equal(ops.unpack_codes(packed, n, bits).cpu().numpy(), idx), n
    assert ops.code_bits(32) == 5 and ops.code_bits(31) == 5 and ops.code_bits(33) == 6 and ops.code_bits(2) == 1 and ops.code_bits(1) == 1
    with pytest.raises(ValueError):
        ops.unpack_codes(torch.zeros(3, dtype=torch.uint8, device="cuda"), 100, 5)


@pytest.mark.gpu
@pytest.mark.parametrize("fused", [False, True])
def test_resume_from_reference_checkpoint_continues_its_curve(golden, fused):
    """Load the reference's checkpoint, take the next optimisation step on the GPU: loss and logits equal
    the reference's own next step (tolerance 2e-5 relative: fp32, other summation order)."""
    import quantized_cnn_amd.adp_qtz_fan as fan
    from quantized_cnn_amd import checkpoint as ck
    from quantized_cnn_amd.train import TrainStep, FusedTrainStep
    g = golden("ckpt")
    c = small_cfg()
    model = fan.ResNet(c, 1).cuda().train()
    step = (FusedTrainStep if fused else TrainStep)(model, c, fan, use_graph=False)
    if fused:
        step._ensure_adam_state()
    beta, _, _, _ = ck.load_checkpoint(CKPT, model, step.opts)
    fan.beta = beta
    try:
        x, y = torch.from_numpy(g["x"]).cuda(), torch.from_numpy(g["y"]).cuda()
        loss = float(step(x, y))
    finally:
        fan.beta = 1
    assert abs(loss - float(g["loss_next"])) <= 2e-5 * float(g["loss_next"])
    ref = g["logits_next"]
    assert np.abs(step.logits.cpu().numpy() - ref).max() <= 2e-5 * np.abs(ref).max()
    sums = np.array([float(p.detach().double().sum()) for _, p in model.named_parameters()])
    assert np.abs(sums - g["p_sum_after_next"]).max() <= 1e-3 * np.abs(g["p_sum_after_next"]).max()
    assert float(step.opts[0].state_dict()["state"][0]["step"]) == 3


@pytest.mark.gpu
def test_checkpoint_loaded_after_graph_capture_reaches_the_replayed_step(golden):
    """A step object that has already captured its HIP graphs holds the addresses of the optimizer state; loading a
    checkpoint afterwards must land in those very tensors.  One captured step on other data first, then the
    reference's checkpoint, then one replay: loss, logits and step counter are the reference's next step."""
    import quantized_cnn_amd.adp_qtz_fan as fan
    from quantized_cnn_amd import checkpoint as ck
    from quantized_cnn_amd.train import FusedTrainStep
    g = golden("ckpt")
    c = small_cfg()
    model = fan.ResNet(c, 1).cuda().train()
    x, y = torch.from_numpy(g["x"]).cuda(), torch.from_numpy(g["y"]).cuda()
    fan.beta = int(g["beta"])
    try:
        step = FusedTrainStep(model, c, fan, use_graph=True)
        step(torch.randn_like(x), y)  # captures and replays once: state tensors now belong to the graph
        st0 = step.opts[0].state[step.ps[0]]
        ptrs = (st0["exp_avg"].data_ptr(), st0["exp_avg_sq"].data_ptr(), st0["step"].data_ptr())
        beta, _, _, _ = ck.load_checkpoint(CKPT, model, step.opts)
        assert beta == int(g["beta"])
        st1 = step.opts[0].state[step.ps[0]]
        assert (st1["exp_avg"].data_ptr(), st1["exp_avg_sq"].data_ptr(), st1["step"].data_ptr()) == ptrs
        assert float(st1["step"]) == 2
        loss = float(step(x, y))
    finally:
        fan.beta = 1
    assert abs(loss - float(g["loss_next"])) <= 2e-5 * float(g["loss_next"])
    ref = g["logits_next"]
    assert np.abs(step.logits.cpu().numpy() - ref).max() <= 2e-5 * np.abs(ref).max()
    assert float(step.opts[0].state_dict()["state"][0]["step"]) == 3
    sums = np.array([float(p.detach().double().sum()) for _, p in model.named_parameters()])
    assert np.abs(sums - g["p_sum_after_next"]).max() <= 1e-3 * np.abs(g["p_sum_after_next"]).max()


@pytest.mark.gpu
def test_hard_mode_and_packed_export_match_reference_test_script(golden, tmp_path):
    """The reference's test-time model (adp_qtz_fan_test.py, hard assignment, eval mode) on the same
    checkpoint; then the packed export reproduces those logits without p_c."""
    import quantized_cnn_amd.adp_qtz_fan as fan
    from quantized_cnn_amd import checkpoint as ck
    g = golden("ckpt")
    ct = small_cfg("test")
    m = fan.ResNet(ct, 1).cuda()
    fan.beta = ck.load_checkpoint(CKPT, m)[0]
    try:
        m.eval()
        x = torch.from_numpy(g["x"]).cuda()
        with torch.no_grad():
            logits = m(x)
        ref = g["hard_logits_eval"]
        assert np.abs(logits.cpu().numpy() - ref).max() <= 2e-5 * np.abs(ref).max()
        path = str(tmp_path / "packed.pt")
        ck.save_packed(path, m)
        packed = torch.load(path, weights_only=True)
        nb_codes, nb_fp32 = ck.packed_nbytes(packed)
        assert nb_codes < nb_fp32 / 10        # 2-bit codes + 4-entry codebooks against fp32 weights
        m2 = fan.ResNet(ct, 1).cuda().eval()
        with torch.no_grad():
            for l in m2.quant_layers():
                l.p_c.fill_(float("nan"))     # the packed model must not read p_c
        assert ck.load_packed(path, m2) == 2.0
        with torch.no_grad():
            logits2 = m2(x)
        assert torch.equal(logits2, logits)
    finally:
        fan.beta = 1
    # the learned fan-in levels are no powers of two; the frozen pow2 levels of the fix variant are
    import quantized_cnn_amd.adp_qtz_fix as fix
    c32 = argparse.Namespace(mode="test", prim_init="uniform", num_level_conv=32, num_level_fc=32, l2_reg=5e-4, liar=True)
    mf = fix.ResNet(c32, [0.3] * 20, torch.zeros(10)).cuda()
    pk = ck.export_packed(mf, 1.0)
    e = pk["layers"]["layer1.0.conv1"]
    assert "sign" not in packed["layers"]["layer1.0.conv1"] and e["bits"] == 5
    q = e["sign"].float() * torch.pow(2.0, e["exponent"].float())
    assert torch.equal(q, e["q_level"])
    assert e["codes"].numel() == (16 * 16 * 9 * 5 + 7) // 8
