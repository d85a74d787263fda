"""Per-step losses of the batch-128 fixture cases on the GPU (plain and fused steps) beside the reference's."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_model as T  # noqa: E402
from quantized_cnn_amd.train import FusedTrainStep, TrainStep  # noqa: E402
import quantized_cnn_amd.ops as ops  # noqa: E402

g = np.load(os.path.join(ROOT, "tests", "golden", "train.npz"), allow_pickle=False)
np.set_printoptions(precision=8)
for variant in ("fan", "hint"):
    k = variant + "_"
    for kind in ("plain", "fused", "fused_exact"):
        mod, c, model = T.build(variant, g)
        model.train()
        rx = np.random.RandomState(int(g[k + "b128_xseed"]))
        x = torch.from_numpy(rx.standard_normal((128, 3, 32, 32)).astype(np.float32)).cuda()
        y = torch.from_numpy(rx.randint(0, 10, 128).astype(np.int64)).cuda()
        ops.EXACT_ARITHMETIC = kind == "fused_exact"
        step = TrainStep(model, c, mod) if kind == "plain" else FusedTrainStep(model, c, mod, use_graph=False)
        losses = []
        for it in range(5):
            if it == 3:
                mod.beta = 2
            losses.append(float(step(x, y)))
        mod.beta = 1
        ops.EXACT_ARITHMETIC = False
        print(variant, kind, np.array(losses), "diff", np.array(losses) - g[k + "b128_loss"])
