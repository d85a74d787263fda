"""Training-step throughput of the three quantizer variants (batch 128, synthetic CIFAR-10 shapes, graph replay,
batch in the step's static buffers): adp_qtz_fan (32 learned levels), adp_qtz_fix (31 frozen power-of-two
levels), adp_qtz_hint (32 learned levels initialised to powers of two)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import quantized_cnn_amd.adp_qtz_fan as fan
import quantized_cnn_amd.adp_qtz_fix as fix
import quantized_cnn_amd.adp_qtz_hint as hint
from quantized_cnn_amd.train import FusedTrainStep

x = torch.randn(128, 3, 32, 32, device="cuda"); y = torch.randint(0, 10, (128,), device="cuda")
for name, mod in (("fan", fan), ("fix", fix), ("hint", hint)):
    conf = bench.cfg()
    conf.liar = True
    mod.beta = 1
    torch.manual_seed(0)
    if name == "fan":
        model = mod.ResNet(conf, 3)
    else:
        model = mod.ResNet(conf, [0.1 * (i + 1) for i in range(20)], torch.zeros(10))
    model = model.cuda().train()
    step = FusedTrainStep(model, conf, mod, use_graph=True)
    step(x, y)
    bx, by = step.static_inputs()
    bx.copy_(x); by.copy_(y)
    for _ in range(10): step(bx, by)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(5):
        e0.record()
        for _ in range(100): step(bx, by)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 100)
    print("adp_qtz_%-5s %.4f ms/step  %.0f img/s" % (name, best, 128 / best * 1e3), flush=True)
