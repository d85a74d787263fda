"""Development aid: how far do the fix variant's parameters drift from the reference after five steps?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import test_gpu_model as T
from quantized_cnn_amd.train import TrainStep
g = np.load(os.path.join(ROOT, "tests", "golden", "train.npz"))
for variant in ("fan", "fix", "hint"):
    k = variant + "_"
    mod, c, model = T.build(variant, g)
    model.train()
    x, y = torch.from_numpy(g[k + "x"]).cuda(), torch.from_numpy(g[k + "y"]).cuda()
    step = TrainStep(model, c, mod)
    losses = []
    for it in range(5):
        if it == 3: mod.beta = 2
        losses.append(float(step(x, y)))
    mod.beta = 1
    named = dict(model.named_parameters())
    print(variant, "loss err", np.abs(np.array(losses) - g[k + "loss"]).max())
    for n in ("layer1.0.conv1", "layer2.0.conv1", "layer3.2.conv2", "linear"):
        d = np.abs(named[n + ".p_c"].detach().reshape(-1)[:4096].cpu().numpy() - g[k + "p_final_head_" + n])
        print("   %-16s max %.4f  frac>1e-2 %.4f  frac>2e-3 %.4f" % (n, d.max(), (d > 1e-2).mean(), (d > 2e-3).mean()))
