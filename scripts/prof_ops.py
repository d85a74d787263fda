"""Development aid: which aten ops of one eager fused step launch the remaining stock kernels (with python stacks)."""
import os, sys, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import quantized_cnn_amd.adp_qtz_fan as fan
from quantized_cnn_amd.train import FusedTrainStep
from torch.profiler import profile, ProfilerActivity
c = argparse.Namespace(mode="train", prim_init="uniform", num_level_conv=32, num_level_fc=32, l2_reg=5e-4, liar=True, prim_lr=1e-2, q_lr=1e-3, lr=1e-2)
torch.manual_seed(0)
model = fan.ResNet(c, 3).cuda().train()
x = torch.randn(128, 3, 32, 32, device="cuda"); y = torch.randint(0, 10, (128,), device="cuda")
step = FusedTrainStep(model, c, fan, use_graph=False)
for _ in range(3): step(x, y)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(x, y)
    torch.cuda.synchronize()
rows = [(e.key, e.count, e.device_time_total) for e in prof.key_averages() if e.device_time_total > 0 and not e.key.startswith(("aq::", "void aq::"))]
for k, n, t in sorted(rows, key=lambda r: -r[2])[:40]:
    print("%-110s %3d %8.1f us" % (k[:110], n, t))
