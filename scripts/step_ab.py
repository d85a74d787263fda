"""Development aid: ms per training step (graph replay) with layer-level switches flipped.
usage: step_ab.py NAME=0|1 ...   (attributes of quantized_cnn_amd.layers)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import quantized_cnn_amd.layers as layers
import quantized_cnn_amd.adp_qtz_fan as fan
from quantized_cnn_amd.train import FusedTrainStep
import bench

def run(tag):
    torch.manual_seed(0)
    conf = bench.cfg()
    fan.beta = 1
    model = fan.ResNet(conf, 3).cuda().train()
    step = FusedTrainStep(model, conf, fan, use_graph=True)
    x = torch.randn(128, 3, 32, 32, device="cuda"); y = torch.randint(0, 10, (128,), device="cuda")
    for _ in range(10): step(x, y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(5):
        e0.record()
        for _ in range(100): step(x, y)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 100)
    print("%-40s %.4f ms/step  %.0f img/s" % (tag, best, 128 / best * 1e3), flush=True)

run("defaults")
for arg in sys.argv[1:]:
    k, v = arg.split("=")
    old = getattr(layers, k)
    setattr(layers, k, bool(int(v)))
    run(arg)
    setattr(layers, k, old)
