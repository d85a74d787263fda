"""Development aid: torch.profiler breakdown of one eager training step."""
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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(3): step(x, y)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=45, max_name_column_width=60))
