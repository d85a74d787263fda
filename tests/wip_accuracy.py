"""How close the kernels are to exact arithmetic, measured with the reference as the yardstick.

Every tolerance above 1e-5 in the whole-model tests rests on one claim: the reference's own fp32 results are that far
from an exact evaluation, so another correct fp32 implementation cannot be held tighter against them.  This file tests
the claim instead of stating it.  For the first training step of each variant on the fixtures of tests/golden/train.npz
(batch 8: logits, loss, regulariser, the first-step gradient of every small tensor and of four p_c heads; batch 128:
logits, loss, regulariser) it evaluates the same step in float64 (oracle/torch_port.py, pinned to the reference, run in
double precision on the host) and compares, per tensor,

    err_ours = rms(kernel result - float64)      err_ref = rms(reference fixture - float64)

    bound:  err_ours <= 1.5 * err_ref + 2e-7 * rms(float64 value)

(the second term is two units of fp32 rounding of the tensor's own scale: a tensor the reference happens to hit exactly
would otherwise admit nothing), and over all compared tensors of a case together the error must not exceed the
reference's at all (ratio of the summed squared relative errors <= 1).
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
        assert len(ours) >= 50
    worst, num, den, rows = (0.0, None), 0.0, 0.0, []
    for name in ours:
        scale = _rms(f64[name])
        e_ours, e_ref = _rms(ours[name] - f64[name]), _rms(ref[name] - f64[name])
        bound = 1.5 * e_ref + 2e-7 * scale
        ratio = e_ours / max(bound, 1e-300)
        worst = max(worst, (ratio, name, e_ours, e_ref, scale))
        if scale > 0:
            num += (e_ours / scale) ** 2
            den += (e_ref / scale) ** 2
        rows.append((name, e_ours, e_ref, scale, bound))
    print("%s batch %d %s: worst tensor %s err %.3g (reference %.3g, scale %.3g); all tensors: ours/reference %.3f"
          % (variant, batch, "exact" if exact else "fast", worst[1], worst[2], worst[3], worst[4], np.sqrt(num / max(den, 1e-300))))
    import os
    if os.environ.get("AQ_ACCURACY_TABLE"):
        for name, e_ours, e_ref, scale, bound in rows:
            print("    %-40s ours %.3g ref %.3g scale %.3g  ours/ref %.2f  rel %.2g" % (name, e_ours, e_ref, scale, e_ours / max(e_ref, 1e-300), e_ours / max(scale, 1e-300)))
    bad = [r for r in rows if r[1] > r[4]]
    assert not bad, bad
    assert num <= den * 1.0 + 1e-14, (num, den)
