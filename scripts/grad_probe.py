"""Development aid: step-0 gradient of conv0 against the reference fixture with pieces swapped for stock operators."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import test_gpu_model as T
import quantized_cnn_amd.layers as layers
g = np.load(os.path.join(ROOT, "tests", "golden", "train.npz"))

def run(variant, tag):
    k = variant + "_"
    mod, c, model = T.build(variant, g)
    model.train()
    x, y = torch.from_numpy(g[k + "x"]).cuda(), torch.from_numpy(g[k + "y"]).cuda()
    logits = model.forward(x)
    loss = torch.nn.functional.cross_entropy(logits, y) + mod.model_loss(c, mod.beta, model)
    loss.backward()
    named = dict(model.named_parameters())
    def rel(a, b):
        a, b = a.detach().cpu().numpy().astype(np.float64), np.asarray(b, np.float64)
        return np.abs(a - b).max() / np.abs(b).max()
    print("%-5s %-22s logits %.2e conv0 %.2e bias %.2e bn0w %.2e dp(l1) %.2e" % (
        variant, tag, rel(logits, g[k + "logits0"]), rel(named["conv0.weight"].grad, g[k + "g0_conv0"]),
        rel(named["linear.bias"].grad, g[k + "g0_bias"]), rel(named["bn0.weight"].grad, g[k + "g0_bn0_w"]),
        rel(named["layer1.0.conv1.p_c"].grad.reshape(-1)[:4096], g[k + "dp0_head_layer1.0.conv1"])))

own_bn, own_sc = layers.bn_relu, layers._Block._apply_shortcut
for variant in ("fan", "fix", "hint"):
    run(variant, "all own kernels")
    layers._Block._apply_shortcut = lambda self, x: self.shortcut(x)
    run(variant, "stock shortcut")
    layers._Block._apply_shortcut = own_sc
    layers.bn_relu = lambda bn, t, relu=True, split=False: ((torch.relu(bn(t)), t) if split else torch.relu(bn(t)))
    run(variant, "stock bn+relu")
    layers.bn_relu = own_bn
