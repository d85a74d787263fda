"""Development aid: loss curve of the bench workload under variations of the fused step."""
import os, sys, argparse
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import quantized_cnn_amd.adp_qtz_fan as fan
import quantized_cnn_amd.layers as layers
from quantized_cnn_amd.train import FusedTrainStep, TrainStep

def run(tag, graph=True, pool=True, defer=True, fuse_adam=True, plain=False):
    c = argparse.Namespace(mode="train", prim_init="uniform", num_level_conv=32, num_level_fc=32, l2_reg=5e-4, liar=True, prim_lr=1e-2, q_lr=1e-3, lr=1e-2)
    fan.beta = 1
    torch.manual_seed(0)
    model = fan.ResNet(c, 3).cuda().train()
    g = torch.Generator(device="cuda").manual_seed(1000)
    x = torch.randn(128, 3, 32, 32, device="cuda", generator=g)
    y = torch.randint(0, 10, (128,), device="cuda", generator=g)
    own_pool, own_defer = layers.bn_relu_pool, layers.defer_wgrad_reductions
    if not pool:
        layers.bn_relu_pool = lambda bn, p, t: p(layers.bn_relu(bn, t))
    if not defer:
        layers.defer_wgrad_reductions = lambda jobs: None
    try:
        step = TrainStep(model, c, fan) if plain else FusedTrainStep(model, c, fan, use_graph=graph, fuse_adam=fuse_adam)
        out = []
        for i in range(110):
            l = step(x, y)
            if i % 10 == 9:
                out.append(round(float(l), 4))
        print("%-28s" % tag, out, flush=True)
    finally:
        layers.bn_relu_pool, layers.defer_wgrad_reductions = own_pool, own_defer

run("plain TrainStep", plain=True)
run("fused graph (bench)")
run("fused eager", graph=False)
run("fused eager no pool", graph=False, pool=False)
run("fused eager no defer", graph=False, defer=False)
run("fused eager torch adam", graph=False, fuse_adam=False)
