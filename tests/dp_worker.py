"""Worker of the data-parallel tests: one rank of a world_size-N job (gloo rendezvous on 127.0.0.1).

mode "cpu": exercises DataParallel's exchange logic on CPU tensors (no kernels involved).
mode "gpu": runs FusedTrainStep with the gradient exchange on cuda:0 (all ranks share the one GPU
            of the test box; the backend is gloo, the production backend is "nccl" = RCCL).
Writes its results to <out>.rank<r>.pt
"""
import argparse
import os
import sys

import torch
import torch.distributed as dist
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class Toy(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv0 = nn.Conv2d(1, 2, 3, bias=False)
        self.bn0 = nn.BatchNorm2d(2)
        self.p_c = nn.Parameter(torch.rand(4, 8))
        self.q_level = nn.Parameter(torch.rand(8))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", required=True)
    ap.add_argument("--out", required=True)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--graph", type=int, default=0)
    a = ap.parse_args()
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from quantized_cnn_amd.train import DataParallel, FusedTrainStep
    from quantized_cnn_amd import layers
    res = {}
    if a.mode == "cpu":
        torch.manual_seed(100 + rank)  # different initial parameters per rank: broadcast must fix that
        m = Toy()
        dp = DataParallel(m)
        res["params_after_broadcast"] = [p.detach().clone() for p in m.parameters()]
        assert layers._dW_reduce_hook is not None
        flat = torch.arange(10, dtype=torch.float32) * (rank + 1)
        layers._dW_reduce_hook(flat)
        res["flat"] = flat
        for i, p in enumerate(dp.dense):
            p.grad = torch.full_like(p, float(rank + 1) * (i + 1))
        dp.reduce_dense()
        res["dense"] = [p.grad.clone() for p in dp.dense]
        res["dense_names"] = [n for n, p in m.named_parameters() if "p_c" not in n and "q_level" not in n]
        dp.close()
        assert layers._dW_reduce_hook is None
    else:
        import quantized_cnn_amd.adp_qtz_hint as hint
        c = argparse.Namespace(mode="train", prim_init="uniform", num_level_conv=32, num_level_fc=32, l2_reg=5e-4, liar=True,
                               prim_lr=1e-2, q_lr=1e-3, lr=1e-2)
        hint.beta = 2
        torch.manual_seed(7 + rank)  # rank-dependent init on purpose
        model = hint.ResNet(c, [0.1 * (i + 1) for i in range(20)], torch.zeros(10)).cuda().train()
        dp = DataParallel(model)
        g = torch.Generator().manual_seed(5)
        X = torch.randn(world * 4, 3, 32, 32, generator=g)
        Y = torch.randint(0, 10, (world * 4,), generator=g)
        x, y = X[rank * 4:(rank + 1) * 4].cuda(), Y[rank * 4:(rank + 1) * 4].cuda()
        step = FusedTrainStep(model, c, hint, dp=dp, use_graph=bool(a.graph))
        losses = [float(step(x, y)) for _ in range(a.steps)]
        res["losses"] = losses
        res["state"] = {k: v.detach().cpu() for k, v in model.state_dict().items() if "running" not in k and "num_batches" not in k}
    torch.save(res, "%s.rank%d.pt" % (a.out, rank))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
