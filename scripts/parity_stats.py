"""Prints how far the GPU steps land from the reference run in tests/golden/train.npz (development aid for
choosing the tolerances of tests/test_gpu_model.py)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_model as T  # noqa: E402
from quantized_cnn_amd.train import FusedTrainStep, TrainStep  # noqa: E402

g = np.load(os.path.join(ROOT, "tests", "golden", "train.npz"), allow_pickle=False)


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


for variant in ("fan", "fix", "hint"):
    k = variant + "_"
    for kind in ("plain", "fused", "fused_graph"):
        mod, c, model = T.build(variant, g)
        model.train()
        x, y = torch.from_numpy(g[k + "x"]).cuda(), torch.from_numpy(g[k + "y"]).cuda()
        if kind == "plain":
            step = TrainStep(model, c, mod)
            logits = model.forward(x)
            loss = step.criterion(logits, y) + mod.model_loss(c, mod.beta, model)
            loss.backward()
            named = dict(model.named_parameters())
            worst = 0.0
            for n, p in named.items():
                key = k + "g0_all_" + n
                if key in g.files and g[key].size:
                    worst = max(worst, rel(p.grad.cpu().numpy(), g[key]))
            heads = max(rel(named[n + ".p_c"].grad.reshape(-1)[:4096].cpu().numpy(), g[k + "dp0_head_" + n])
                        for n in ("layer1.0.conv1", "layer2.0.conv1", "layer3.2.conv2", "linear"))
            print(variant, kind, "logits0 %.2e  small grads worst %.2e  dp heads worst %.2e" % (rel(logits.detach().cpu().numpy(), g[k + "logits0"]), worst, heads))
            for o in step.opts:
                o.step()
                o.zero_grad()
            losses = [loss.item()]
            first = 1
        else:
            step = FusedTrainStep(model, c, mod, use_graph=(kind == "fused_graph"))
            losses, first = [], 0
        for it in range(first, int(g["steps"])):
            if it == 3:
                mod.beta = 2
            losses.append(float(step(x, y)))
        mod.beta = 1
        named = dict(model.named_parameters())
        print(variant, kind, "loss curve rel %.2e" % (np.abs(np.array(losses) - g[k + "loss"]).max() / g[k + "loss"].max()))
        for n in ("layer1.0.conv1", "layer3.2.conv2", "linear"):
            d = np.abs(named[n + ".p_c"].detach().reshape(-1)[:4096].cpu().numpy().astype(np.float64) - g[k + "p_final_head_" + n])
            print("   %s p_c: max %.2e  frac>1e-4: %.4f  frac>1e-3: %.4f frac>1e-2: %.4f" % (n, d.max(), (d > 1e-4).mean(), (d > 1e-3).mean(), (d > 1e-2).mean()))
        worst_small, worst_name = 0.0, ""
        for n, p in named.items():
            key = k + "final_all_" + n
            if key in g.files:
                lr = 1e-3 if "q_level" in n else 1e-2
                d = np.abs(p.detach().cpu().numpy().astype(np.float64) - g[key]).max() / lr
                if d > worst_small:
                    worst_small, worst_name = d, n
        print("   small tensors: worst |d|/lr %.3e at %s" % (worst_small, worst_name))
        sums = np.array([float(p.detach().double().sum()) for _, p in model.named_parameters()])
        numel = np.array([p.numel() for _, p in model.named_parameters()])
        lrs = np.array([1e-3 if "q_level" in n else 1e-2 for n, _ in model.named_parameters()])
        ds = np.abs(sums - g[k + "p_sum"]) / (numel * lrs * 5)
        print("   p_sum: worst |dsum|/(numel*lr*steps) %.3e" % ds.max())
    # batch 128
    mod, c, model = T.build(variant, g)
    model.train()
    rx = np.random.RandomState(int(g[k + "b128_xseed"]))
    x = torch.from_numpy(rx.standard_normal((128, 3, 32, 32)).astype(np.float32)).cuda()
    y = torch.from_numpy(rx.randint(0, 10, 128).astype(np.int64)).cuda()
    step = FusedTrainStep(model, c, mod, use_graph=True)
    losses = []
    for it in range(5):
        if it == 3:
            mod.beta = 2
        losses.append(float(step(x, y)))
        if it == 0:
            l0 = rel(step.logits.cpu().numpy(), g[k + "b128_logits0"])
    mod.beta = 1
    print(variant, "b128 logits0 %.2e loss curve rel %.2e" % (l0, np.abs(np.array(losses) - g[k + "b128_loss"]).max() / g[k + "b128_loss"].max()))
