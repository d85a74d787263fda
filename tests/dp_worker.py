"""Worker of the data-parallel tests: one rank of a world_size-N job (gloo rendezvous on 127.0.0.1).

mode "cpu": exercises DataParallel's exchange logic on CPU tensors (no kernels involved).
mode "gpu": runs FusedTrainStep with the gradient exchange on cuda:0 (all ranks share the one GPU
            of the test box; the backend is gloo, the production backend is "nccl" = RCCL).
mode "gpu_eval" / "gpu_syncbn": one step with separate (unfused) gradients kept, BatchNorm on running statistics
            resp. as SyncBatchNorm: in both the two half batches must reproduce a full-batch single-process step.
mode "nccl1": a single rank on the RCCL backend; the two-graphs-around-an-all-reduce form of the step (forced at
            world size 1) against the one-graph form, bit for bit.
mode "tern_eval": one FusedTernaryStep (--net resnet20 | resnet18 at 64 x 64, --mode_q ttq | twn | inq) with BatchNorm on its
            running statistics, the exchange in several buckets; "tern_nccl1": the same step on RCCL with one rank, forced
            exchange against the one-graph form.
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
    ap.add_argument("--net", default="resnet20")
    ap.add_argument("--mode_q", default="ttq")
    a = ap.parse_args()
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    if a.mode in ("nccl1", "tern_nccl1"):
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    else:
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
    elif a.mode == "nccl1":
        import quantized_cnn_amd.adp_qtz_fan as fan
        c = argparse.Namespace(mode="train", prim_init="uniform", num_level_conv=32, num_level_fc=32, l2_reg=5e-4, liar=True,
                               prim_lr=1e-2, q_lr=1e-3, lr=1e-2)
        fan.beta = 1
        g = torch.Generator().manual_seed(5)
        x, y = torch.randn(16, 3, 32, 32, generator=g).cuda(), torch.randint(0, 10, (16,), generator=g).cuda()
        runs = []
        for force in (False, True):
            torch.manual_seed(3)
            model = fan.ResNet(c, 3).cuda().train()
            dp = DataParallel(model)
            step = FusedTrainStep(model, c, fan, dp=dp, use_graph=bool(a.graph), force_exchange=force)
            assert step._exchange == force
            losses = [float(step(x, y)) for _ in range(a.steps)]
            if a.graph:
                assert (step._graphs[1] is not None) == force  # two graphs around the all-reduce / one graph
            runs.append((losses, {k: v.detach().clone() for k, v in model.state_dict().items()}))
            dp.close()
        res["losses"], res["losses_forced"] = runs[0][0], runs[1][0]
        res["equal"] = all(torch.equal(runs[0][1][k], runs[1][1][k]) for k in runs[0][1])
        res["backend"] = dist.get_backend()
    elif a.mode in ("tern_eval", "tern_nccl1"):
        import quantized_cnn_amd.ternary as T
        from quantized_cnn_amd.resnet18 import ResNet18
        from quantized_cnn_amd.train import FusedTernaryStep
        c = argparse.Namespace(mode="train", mode_q=a.mode_q, l2_reg=5e-4, lr=1e-3, prim_init="uniform", num_level_conv=32, num_level_fc=32)

        def build(seed):
            torch.manual_seed(seed)
            model = (T.ResNet(c, 1) if a.net == "resnet20" else ResNet18(c, num_classes=10)).cuda().train()
            if a.mode_q == "inq":
                for m in model.modules():
                    if isinstance(m, T.INQConv2d):
                        m.quantize_portion(0.5)
            return model
        size = 32 if a.net == "resnet20" else 64
        g = torch.Generator().manual_seed(5)
        X = torch.randn(8, 3, size, size, generator=g)
        Y = torch.randint(0, 10, (8,), generator=g)
        if a.mode == "tern_eval":
            model = build(7 + rank)  # rank-dependent init on purpose: the broadcast must fix it
            for m in model.modules():
                if isinstance(m, nn.BatchNorm2d):
                    m.eval()
            dp = DataParallel(model)
            per = 8 // world
            x, y = X[rank * per:(rank + 1) * per].cuda(), Y[rank * per:(rank + 1) * per].cuda()
            step = FusedTernaryStep(model, c, dp=dp, use_graph=bool(a.graph), keep_grads=True, bucket_bytes=1 << 16)
            res["loss"] = float(step(x, y))
            res["reg"] = float(step.reg)
            res["nbuckets"] = -(-step.g_flat.numel() * 4 // (1 << 16))
            res["grads"] = {n: p.grad.detach().cpu().clone() for n, p in model.named_parameters() if p.grad is not None}
            res["state"] = {k: v.detach().cpu() for k, v in model.state_dict().items()}
        else:
            runs = []
            for force in (False, True):
                model = build(3)
                dp = DataParallel(model)
                step = FusedTernaryStep(model, c, dp=dp, use_graph=bool(a.graph), force_exchange=force, bucket_bytes=1 << 16)
                assert step._exchange == force
                losses = [float(step(X.cuda(), Y.cuda())) for _ in range(a.steps)]
                if a.graph:
                    assert (step._graphs[1] is not None) == force
                runs.append((losses, {k: v.detach().clone() for k, v in model.state_dict().items()}))
                dp.close()
            res["losses"], res["losses_forced"] = runs[0][0], runs[1][0]
            res["equal"] = all(torch.equal(runs[0][1][k], runs[1][1][k]) for k in runs[0][1])
            res["backend"] = dist.get_backend()
    elif a.mode in ("gpu_eval", "gpu_syncbn"):
        import quantized_cnn_amd.adp_qtz_hint as hint
        c = argparse.Namespace(mode="train", prim_init="uniform", num_level_conv=32, num_level_fc=32, l2_reg=5e-4, liar=True,
                               prim_lr=1e-2, q_lr=1e-3, lr=1e-2)
        hint.beta = 2
        torch.manual_seed(7)
        model = hint.ResNet(c, [0.1 * (i + 1) for i in range(20)], torch.zeros(10)).cuda().train()
        if a.mode == "gpu_eval":
            for m in model.modules():
                if isinstance(m, nn.BatchNorm2d):
                    m.eval()
        dp = DataParallel(model, sync_bn=(a.mode == "gpu_syncbn"))
        print("rank %d: model broadcast done" % rank, flush=True)
        if a.mode == "gpu_syncbn":
            assert any(isinstance(m, nn.SyncBatchNorm) for m in model.modules())
            try:
                FusedTrainStep(model, c, hint, dp=dp, use_graph=True)(torch.zeros(2, 3, 32, 32).cuda(), torch.zeros(2, dtype=torch.long).cuda())
                res["graph_refused"] = False
            except RuntimeError as e:
                res["graph_refused"] = "SyncBatchNorm" in str(e)
        g = torch.Generator().manual_seed(5)
        X = torch.randn(world * 4, 3, 32, 32, generator=g)
        Y = torch.randint(0, 10, (world * 4,), generator=g)
        x, y = X[rank * 4:(rank + 1) * 4].cuda(), Y[rank * 4:(rank + 1) * 4].cuda()
        print("rank %d: graph refusal checked" % rank, flush=True)
        step = FusedTrainStep(model, c, hint, dp=dp, use_graph=False, fuse_adam=False)
        step._phase_a(x, y)
        torch.cuda.synchronize()
        print("rank %d: phase A done" % rank, flush=True)
        step._comm()
        torch.cuda.synchronize()
        print("rank %d: exchange done" % rank, flush=True)
        step._phase_b()
        res["loss"] = float(step.loss)
        res["reg"] = float(step.reg)
        res["grads"] = {n: p.grad.detach().cpu().clone() for n, p in model.named_parameters() if p.grad is not None}
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
