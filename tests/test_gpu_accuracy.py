"""How close the kernels are to exact arithmetic, measured with the reference as the yardstick.

The whole-model tests hold some tensors to tolerances above 1e-5 (BatchNorm gradients 1e-4, batch-128 loss curves
2e-3 at steps 4-5).  This file measures what those tolerances rest on.  For the first training step of each variant on
the fixtures of tests/golden/train.npz (batch 8: logits, loss, regulariser, the first-step gradient of every small tensor
and of four p_c heads; batch 128: logits, loss, regulariser) the same step is evaluated in float64 (oracle/torch_port.py,
pinned to the reference, run in double precision on the host) and, per tensor,

    err_ours = rms(kernel result - float64)      err_ref = rms(reference fixture - float64)

are compared.  What holds (MI355X, fast and exact arithmetic alike), and is asserted:

  * logits, loss, regulariser, BatchNorm weight gradients, the linear bias gradient:
        err_ours <= 1.5 * err_ref + 2e-7 * rms(value)     (two units of fp32 rounding of the tensor's own scale: a tensor
                                                            the reference happens to hit exactly would admit nothing)
    measured: both sides 3e-8 .. 1e-7 of the tensor's scale.
  * gradients of p_c and of the stem weight:  err_ours <= 3 * err_ref + 2e-7 * rms(value)
    measured: ours 2.2e-6 .. 7.3e-6 of scale, the reference 2.1e-6 .. 4.4e-6 (the weight gradient sums 8192 products per
    entry on the fp32 matrix cores in one chain per tile; the reference's `mm` blocks the same sum).
  * gradients of the BatchNorm biases and of the level vectors q_level: sums over N*H*W (or over all weights of a layer)
    of terms that cancel to 1e-3 of their size, so they inherit the error of every upstream value amplified by that
    factor.  The reference computes exactly these sums with double accumulators (torch's CPU batch-norm backward) or a
    blocked `mm`; the kernels sum in fp32 (second stages in double).  Measured: ours 4.4e-6 median / 1.1e-4 worst of the
    tensor's scale, the reference 3e-6 median / 1.8e-5 worst.  Asserted: err_ours <= 2e-4 * rms(value), and the median
    over these tensors of err_ours / err_ref <= 3.
    This is the measured ground of the 1e-4 tolerance on BatchNorm gradients in test_gpu_model.py: the reference is up to
    20x closer to float64 than the kernels on these tensors, and 1e-4 of the tensor's scale is what the kernels achieve.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from quantized_cnn_amd.train import FusedTrainStep

from oracle import torch_port as TP
from test_gpu_model import build


def _rms(a):
    a = np.asarray(a, np.float64)
    return float(np.sqrt(np.mean(a * a))) if a.size else 0.0


def _float64_step(variant, model, x, y, g):
    """First training step in double precision on the host: (logits, loss, reg, {name: gradient})."""
    if variant == "fan":
        port = TP.PortModel("fan", 32)
    else:
        port = TP.PortModel(variant, 32, s=[float(v) for v in g[variant + "_s"]], v=torch.from_numpy(g[variant + "_v"]))
    port.load_state_dict({k: v.detach().cpu() for k, v in model.state_dict().items()})
    port = port.double().train()
    port.beta = 1
    logits = port(x.double())
    reg = TP.model_loss(5e-4, 1, port)
    loss = torch.nn.functional.cross_entropy(logits, y) + reg
    loss.backward()
    grads = {n: (None if p.grad is None else p.grad.numpy().copy()) for n, p in port.named_parameters()}
    return logits.detach().numpy(), float(loss), float(reg), grads


@pytest.mark.parametrize("variant", ["fan", "fix", "hint"])
@pytest.mark.parametrize("batch", [8, 128])
@pytest.mark.parametrize("exact", [False, True])
def test_first_step_is_as_close_to_float64_as_the_reference(golden, variant, batch, exact):
    import quantized_cnn_amd.ops as ops
    g = golden("train")
    k = variant + ("_" if batch == 8 else "_b128_")
    mod, c, model = build(variant, g)
    model.train()
    if batch == 8:
        x, y = torch.from_numpy(g[k + "x"]), torch.from_numpy(g[k + "y"])
    else:
        rx = np.random.RandomState(int(g[k + "xseed"]))
        x = torch.from_numpy(rx.standard_normal((128, 3, 32, 32)).astype(np.float32))
        y = torch.from_numpy(rx.randint(0, 10, 128).astype(np.int64))
    logits64, loss64, reg64, grads64 = _float64_step(variant, model, x, y, g)
    ops.EXACT_ARITHMETIC = exact
    try:
        step = FusedTrainStep(model, c, mod, use_graph=False, fuse_adam=False)
        loss = float(step(x.cuda(), y.cuda()))
    finally:
        ops.EXACT_ARITHMETIC = False
    ours = {"logits": step.logits.cpu().numpy(), "loss": np.array([loss]), "reg": np.array([float(step.reg)])}
    ref = {"logits": g[k + "logits0"], "loss": g[k + "loss"][:1], "reg": g[k + "reg"][:1]}
    f64 = {"logits": logits64, "loss": np.array([loss64]), "reg": np.array([reg64])}
    if batch == 8:
        named = dict(model.named_parameters())
        for n, p in named.items():
            key = k + "g0_all_" + n
            if key in g.files and g[key].size:
                ours["d " + n], ref["d " + n], f64["d " + n] = p.grad.cpu().numpy(), g[key], grads64[n]
        for n in ("layer1.0.conv1", "layer2.0.conv1", "layer3.2.conv2", "linear"):
            ours["d %s.p_c[:4096]" % n] = named[n + ".p_c"].grad.reshape(-1)[:4096].cpu().numpy()
            ref["d %s.p_c[:4096]" % n] = g[k + "dp0_head_" + n]
            f64["d %s.p_c[:4096]" % n] = grads64[n + ".p_c"].reshape(-1)[:4096]
        assert len(ours) >= 40  # (adp_qtz_fix has no level gradients)
    rows = []
    for name in ours:
        scale = _rms(f64[name])
        rows.append((name, _rms(ours[name] - f64[name]), _rms(ref[name] - f64[name]), scale))
    import os
    if os.environ.get("AQ_ACCURACY_TABLE"):
        for name, e_ours, e_ref, scale in rows:
            print("    %-40s ours %.3g ref %.3g scale %.3g  ours/ref %.2f  rel %.2g" % (name, e_ours, e_ref, scale, e_ours / max(e_ref, 1e-300), e_ours / max(scale, 1e-300)))
    bad, ratios = [], []
    for name, e_ours, e_ref, scale in rows:
        if name.endswith("bn1.bias") or name.endswith("bn2.bias") or name.endswith("bn0.bias") or name.endswith("bnlast.bias") or name.endswith("q_level"):
            ratios.append(e_ours / max(e_ref, 1e-300))
            ok = e_ours <= 2e-4 * scale
        elif "p_c" in name or name == "d conv0.weight":
            ok = e_ours <= 3.0 * e_ref + 2e-7 * scale
        else:
            ok = e_ours <= 1.5 * e_ref + 2e-7 * scale
        if not ok:
            bad.append((name, e_ours, e_ref, scale))
    assert not bad, bad
    if ratios:
        print("%s batch %d %s: cancelling sums (BatchNorm bias / q_level gradients): err_ours / err_ref median %.2f, worst %.1f"
              % (variant, batch, "exact" if exact else "fast", float(np.median(ratios)), max(ratios)))
        assert float(np.median(ratios)) <= 3.0, float(np.median(ratios))
