#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING the reference.

Runs only in the build container (needs /root/reference); the GPU box and the
test-suite never execute this file, they read the .npz files it wrote.

The reference scripts are loaded unmodified with importlib under a module name
other than "__main__" (so their CLI blocks do not run).  Their top-level imports
of torchvision / tensorboardX / h5py, which are not installed here and are not
used by the classes on the quantized hot path, are satisfied by empty stand-in
modules.  Internal tensors (the synthesised weight W and its gradient) are
captured by temporarily wrapping torch.matmul / torch.index_select, i.e. by
instrumenting torch, not the reference.

Every fixture stores its INPUTS as well as the reference OUTPUTS; nothing is
re-derived from torch RNG seeds at test time.  The whole-model fixture derives
its (large) initial parameters from numpy's legacy RandomState, whose stream is
frozen by numpy's compatibility policy, and stores seeds + expected outputs.

usage: python tests/golden/make_golden.py
"""
import argparse
import importlib.util
import os
import sys
import types

import numpy as np

REF = "/root/reference/adp_qtz"
OUT = os.path.dirname(os.path.abspath(__file__))


def _stub_modules():
    import matplotlib
    matplotlib.use("Agg")
    for name in ("torchvision", "torchvision.transforms", "tensorboardX", "h5py"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    sys.modules["tensorboardX"].SummaryWriter = type("SummaryWriter", (), {})


def load_ref(script):
    _stub_modules()
    argv, sys.argv = sys.argv, [script]
    try:
        spec = importlib.util.spec_from_file_location("ref_" + script[:-3], os.path.join(REF, script))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        sys.argv = argv
    return mod


def cfg(mode="train", prim_init="uniform", L=32, Lfc=32, l2_reg=5e-4, liar=True):
    return argparse.Namespace(mode=mode, prim_init=prim_init, num_level_conv=L, num_level_fc=Lfc,
                              l2_reg=l2_reg, liar=liar)


class Spy:
    """Wraps torch.matmul / torch.index_select while a reference forward runs and keeps
    the (s_c @ q_level) or q_level[idx] result so that W and dW can be read back."""

    def __init__(self, torch):
        self.torch = torch
        self.w = []
        self.idx = []

    def __enter__(self):
        t = self.torch
        self._mm, self._is, self._max = t.matmul, t.index_select, t.max

        def matmul(a, b, *k, **kw):
            out = self._mm(a, b, *k, **kw)
            if b.dim() == 1 and a.dim() >= 3:  # (.., L) @ (L,) : weight synthesis
                if out.requires_grad:
                    out.retain_grad()
                self.w.append(out)
            return out

        def index_select(src, dim, index, *k, **kw):
            out = self._is(src, dim, index, *k, **kw)
            self.w.append(out)
            self.idx.append(index)
            return out

        t.matmul, t.index_select = matmul, index_select
        return self

    def __exit__(self, *a):
        self.torch.matmul, self.torch.index_select = self._mm, self._is


def f32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32))


def special_rows(L, rs):
    """Rows that exercise ties / degenerate inputs.  Returns (rows[S,L], names)."""
    rows, names = [], []
    r = rs.random_sample(L).astype(np.float32)
    m = r.max() + np.float32(0.25)
    t = r.copy(); t[3] = m; t[L - 2] = m
    rows.append(t); names.append("exact_tie_of_max_first_index_wins")
    rows.append(np.full(L, 0.37, np.float32)); names.append("all_equal")
    rows.append(np.zeros(L, np.float32)); names.append("all_zero_nan")
    t = rs.standard_normal(L).astype(np.float32) * np.float32(1e-20)
    rows.append(t); names.append("tiny_magnitude")
    t = rs.standard_normal(L).astype(np.float32) * np.float32(1e15)
    rows.append(t); names.append("huge_magnitude")
    t = -rs.random_sample(L).astype(np.float32) - np.float32(0.1)
    rows.append(t); names.append("all_negative")
    t = np.zeros(L, np.float32); t[L - 1] = 1.0
    rows.append(t); names.append("one_hot_last")
    return np.stack(rows), names


def gen_synth(torch):
    """(p_c, q, beta) -> W_soft, idx, W_hard, sum W^2 ; dW -> dp_c, dq  through MyConv2d of each variant."""
    out = {}
    rs = np.random.RandomState(1234)
    mods = {"fan": load_ref("adp_qtz_fan.py"), "fix": load_ref("adp_qtz_fix.py"), "hint": load_ref("adp_qtz_hint.py")}
    tests = {"fan": load_ref("adp_qtz_fan_test.py"), "fix": load_ref("adp_qtz_fix_test.py"),
             "hint": load_ref("adp_qtz_hint_test.py")}
    cout, cin = 8, 5
    case = 0
    for variant in ("fan", "fix", "hint"):
        for init in ("uniform", "normal"):
            for beta in (1, 2, 3, 4):
                if variant != "fan" and init == "normal" and beta in (2, 3):
                    continue
                c = cfg(prim_init=init)
                mod, tmod = mods[variant], tests[variant]
                for m in (mod, tmod):
                    m.beta = beta
                    m.config = c
                s = float(np.exp(rs.uniform(np.log(0.05), np.log(3.0))))
                if variant == "fan":
                    conv = mod.MyConv2d(c, cin, cout, 3, 1, 1)
                else:
                    conv = mod.MyConv2d(c, s, cin, cout, 3, 1, 1)
                L = conv.p_c.shape[-1]
                R = cout * cin * 9
                p = (rs.random_sample((R, L)) if init == "uniform" else rs.standard_normal((R, L))).astype(np.float32)
                sp, names = special_rows(L, rs)
                p[: sp.shape[0]] = sp
                q = conv.q_level.detach().numpy().copy()
                if variant == "fan":  # fan levels are random in the reference: draw them here and store them
                    q = (rs.uniform(-1, 1, L) * 0.2357).astype(np.float32)
                with torch.no_grad():
                    conv.p_c.copy_(torch.from_numpy(p.reshape(cout, cin, 3, 3, L)))
                    conv.q_level.copy_(torch.from_numpy(q))
                x = torch.from_numpy(rs.standard_normal((2, cin, 6, 6)).astype(np.float32))
                g = torch.from_numpy(rs.standard_normal((2, cout, 6, 6)).astype(np.float32))
                # soft forward + backward through the trainer's class
                with Spy(torch) as spy:
                    y = conv.forward(x)
                    (y * g).sum().backward()
                w = spy.w[0]
                dW = w.grad.detach().numpy().reshape(R)
                dp = conv.p_c.grad.detach().numpy().reshape(R, L)
                dq = None if conv.q_level.grad is None else conv.q_level.grad.detach().numpy()
                # regulariser through the reference's model_loss (adds l2_reg * sum W^2)
                holder = torch.nn.Module()
                holder.conv = conv
                conv.p_c.grad = None
                if conv.q_level.grad is not None:
                    conv.q_level.grad = None
                reg = mod.model_loss(c, beta, holder)
                reg.backward()
                dp_reg = conv.p_c.grad.detach().numpy().reshape(R, L)
                dq_reg = None if conv.q_level.grad is None else conv.q_level.grad.detach().numpy()
                # hard forward through the *_test.py class (same parameters)
                ct = cfg(mode="test", prim_init=init)
                if variant == "fan":
                    hconv = tmod.MyConv2d(ct, cin, cout, 3, 1, 1)
                else:
                    hconv = tmod.MyConv2d(ct, s, cin, cout, 3, 1, 1)
                with torch.no_grad():
                    hconv.p_c.copy_(conv.p_c)
                    hconv.q_level.copy_(conv.q_level)
                    with Spy(torch) as hspy:
                        yh = hconv.forward(x)
                # the reference's own s_c, to flag rows whose arg-max is decided by fp32 rounding
                with torch.no_grad():
                    pc = conv.p_c.reshape(R, L)
                    pn = pc / torch.sqrt(torch.sum(pc ** 2, dim=-1, keepdim=True))
                    e = torch.exp(beta * pn)
                    sc = e / torch.sum(e, dim=-1, keepdim=True)
                    top2 = torch.topk(sc, 2, dim=-1).values
                    ptop2 = torch.topk(pc, 2, dim=-1).values
                    rounding_tie = ((top2[:, 0] == top2[:, 1]) & (ptop2[:, 0] != ptop2[:, 1])).numpy()
                k = "c%02d_" % case
                out[k + "meta"] = np.array([variant, init, str(beta), str(L), str(cout), str(cin), repr(s)])
                out[k + "special"] = np.array(names)
                out[k + "p_c"] = p
                out[k + "q"] = f32(q)
                out[k + "W"] = f32(w.detach().numpy().reshape(R))
                out[k + "idx"] = hspy.idx[0].numpy().astype(np.int64).reshape(R)
                out[k + "W_hard"] = f32(hspy.w[0].numpy().reshape(R))
                out[k + "rounding_tie"] = rounding_tie
                out[k + "x"] = f32(x.numpy()); out[k + "g"] = f32(g.numpy())
                out[k + "y"] = f32(y.detach().numpy()); out[k + "y_hard"] = f32(yh.numpy())
                out[k + "dW"] = f32(dW); out[k + "dp_c"] = f32(dp)
                out[k + "dq"] = np.zeros(0, np.float32) if dq is None else f32(dq)
                out[k + "reg"] = np.float32(reg.item())
                out[k + "dp_c_reg"] = f32(dp_reg)
                out[k + "dq_reg"] = np.zeros(0, np.float32) if dq_reg is None else f32(dq_reg)
                case += 1
    out["ncase"] = np.int64(case)
    np.savez_compressed(os.path.join(OUT, "synth.npz"), **out)
    print("synth.npz: %d cases" % case)


def gen_conv(torch):
    """x, W, dy -> y, dx, dW for the five distinct (Cin, Cout, stride) layer shapes of ResNet-20
    (reduced batch / spatial size; one case at full 32x32), via the reference's 9-tap MyConv2d."""
    mod = load_ref("adp_qtz_fan.py")
    mod.beta = 2
    rs = np.random.RandomState(4321)
    out = {}
    shapes = [  # cin, cout, stride, H, N
        (16, 16, 1, 32, 2), (16, 32, 2, 32, 2), (32, 32, 1, 16, 2), (32, 64, 2, 16, 2), (64, 64, 1, 8, 3),
        (16, 16, 1, 9, 1), (16, 32, 2, 9, 1), (3, 7, 1, 5, 2), (5, 3, 2, 6, 2), (64, 64, 1, 4, 1),
    ]
    L = 4
    for i, (cin, cout, stride, H, N) in enumerate(shapes):
        c = cfg(L=L)
        conv = mod.MyConv2d(c, cin, cout, 3, stride, 1)
        with torch.no_grad():
            conv.p_c.copy_(torch.from_numpy(rs.random_sample(tuple(conv.p_c.shape)).astype(np.float32)))
            conv.q_level.copy_(torch.from_numpy((rs.standard_normal(L) * 0.2).astype(np.float32)))
        x = torch.from_numpy(rs.standard_normal((N, cin, H, H)).astype(np.float32)).requires_grad_(True)
        with Spy(torch) as spy:
            y = conv.forward(x)
            g = torch.from_numpy(rs.standard_normal(tuple(y.shape)).astype(np.float32))
            (y * g).sum().backward()
        k = "c%02d_" % i
        out[k + "shape"] = np.array([cin, cout, stride, H, N], np.int64)
        out[k + "x"] = f32(x.detach().numpy()); out[k + "W"] = f32(spy.w[0].detach().numpy())
        out[k + "y"] = f32(y.detach().numpy()); out[k + "dy"] = f32(g.numpy())
        out[k + "dx"] = f32(x.grad.numpy()); out[k + "dW"] = f32(spy.w[0].grad.numpy())
    out["ncase"] = np.int64(len(shapes))
    np.savez_compressed(os.path.join(OUT, "conv.npz"), **out)
    print("conv.npz: %d cases" % len(shapes))


def gen_linear(torch):
    out = {}
    rs = np.random.RandomState(99)
    for i, variant in enumerate(("fan", "fix", "hint")):
        mod = load_ref("adp_qtz_%s.py" % variant)
        mod.beta = 3
        c = cfg()
        mod.config = c
        v = torch.from_numpy(rs.standard_normal(10).astype(np.float32))
        lin = mod.MyLinear(c, 64, 10) if variant == "fan" else mod.MyLinear(c, 0.8, v, 64, 10)
        L = lin.p_c.shape[-1]
        with torch.no_grad():
            lin.p_c.copy_(torch.from_numpy(rs.random_sample((64, 10, L)).astype(np.float32)))
            if variant == "fan":
                lin.q_level.copy_(torch.from_numpy((rs.uniform(-1, 1, L) * 0.3).astype(np.float32)))
        x = torch.from_numpy(rs.standard_normal((7, 64)).astype(np.float32)).requires_grad_(True)
        g = torch.from_numpy(rs.standard_normal((7, 10)).astype(np.float32))
        with Spy(torch) as spy:
            y = lin.forward(x)
            (y * g).sum().backward()
        k = "c%02d_" % i
        out[k + "variant"] = np.array(variant)
        out[k + "p_c"] = f32(lin.p_c.detach().numpy()); out[k + "q"] = f32(lin.q_level.detach().numpy())
        out[k + "bias"] = f32(lin.bias.detach().numpy())
        out[k + "x"] = f32(x.detach().numpy()); out[k + "g"] = f32(g.numpy()); out[k + "y"] = f32(y.detach().numpy())
        out[k + "W"] = f32(spy.w[0].detach().numpy()); out[k + "dW"] = f32(spy.w[0].grad.numpy())
        out[k + "dx"] = f32(x.grad.numpy()); out[k + "dp_c"] = f32(lin.p_c.grad.numpy())
        out[k + "dq"] = np.zeros(0, np.float32) if lin.q_level.grad is None else f32(lin.q_level.grad.numpy())
        out[k + "dbias"] = f32(lin.bias.grad.numpy())
    out["ncase"] = np.int64(3)
    np.savez_compressed(os.path.join(OUT, "linear.npz"), **out)
    print("linear.npz: 3 cases")


def gen_levels(torch):
    """q_level known answers of fix / hint for several scales; fan's bound."""
    out = {}
    fix, hint, fan = load_ref("adp_qtz_fix.py"), load_ref("adp_qtz_hint.py"), load_ref("adp_qtz_fan.py")
    i = 0
    v = torch.zeros(10)
    for s in (0.3, 1.0, 2.7, 0.0421, 0.75, 0.70710678):
        for L in (32, 8):
            for liar in (True, False):
                c = cfg(L=L, Lfc=L, liar=liar)
                fix.config = c
                out["fix_%02d" % i] = f32(fix.MyConv2d(c, s, 2, 2, 3, 1, 1).q_level.detach().numpy())
                out["fixfc_%02d" % i] = f32(fix.MyLinear(c, s, v, 4, 10).q_level.detach().numpy())
                out["hint_%02d" % i] = f32(hint.MyConv2d(c, s, 2, 2, 3, 1, 1).q_level.detach().numpy())
                out["hintfc_%02d" % i] = f32(hint.MyLinear(c, s, v, 4, 10).q_level.detach().numpy())
                out["arg_%02d" % i] = np.array([s, L, float(liar)])
                i += 1
    out["ncase"] = np.int64(i)
    # fan: levels are U(-b, b); record b as realised by the reference for a few fan-ins
    torch.manual_seed(0)
    fb = []
    for cin in (16, 32, 64):
        m = fan.MyConv2d(cfg(L=4096), cin, 1, 3, 1, 1)
        fb.append([cin * 9, float(m.q_level.abs().max())])
    m = fan.MyLinear(cfg(Lfc=4096), 64, 1)
    fb.append([64, float(m.q_level.abs().max())])
    out["fan_bound_obs"] = np.array(fb)
    np.savez_compressed(os.path.join(OUT, "levels.npz"), **out)
    print("levels.npz: %d cases" % i)


def fill_model_from_numpy(torch, model, seed):
    """Deterministic, torch-RNG-independent initial state: every parameter, in
    named_parameters() order, from one numpy RandomState stream.  p_c ~ U[0,1); q_level kept as the
    variant initialised it, except fan (random in the reference) ~ U(-0.2,0.2); everything else
    small normal; BN weight 1 + small normal."""
    rs = np.random.RandomState(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            n = p.numel()
            if name.endswith("p_c"):
                a = rs.random_sample(n)
            elif name.endswith("q_level"):
                a = rs.uniform(-0.2, 0.2, n)
                if getattr(model, "_keep_q", False):
                    continue
            elif "bn" in name and name.endswith("weight"):
                a = 1.0 + 0.1 * rs.standard_normal(n)
            else:
                a = 0.1 * rs.standard_normal(n)
            p.copy_(torch.from_numpy(a.astype(np.float32)).reshape(p.shape))


def relu_margin(torch, model, x):
    """Smallest |pre-activation| of one training-mode forward pass: every BatchNorm output of these models feeds a
    ReLU (adp_qtz_fan.py:674-675, 690, 726, 730).  The pass runs without grad on a copy of the buffers' state."""
    vals = []
    hooks = [m.register_forward_hook(lambda mod, inp, out: vals.append(float(out.detach().abs().min())))
             for m in model.modules() if isinstance(m, torch.nn.BatchNorm2d)]
    saved = [b.clone() for b in model.buffers()]
    with torch.no_grad():
        model.forward(x)
    for b, v in zip(model.buffers(), saved):
        b.copy_(v)
    for h in hooks:
        h.remove()
    return min(vals)


def gen_train(torch):
    """Five optimisation steps (fwd + CE + model_loss + bwd + 3x Adam) of each variant's full
    ResNet-20 on a fixed synthetic batch; stores losses, logits and parameter checksums.

    Batch 8: the batch is drawn (seed after seed) until no ReLU input of the first forward pass lies within
    RELU_MARGIN of zero, so that another correct fp32 evaluation (other summation order, ~1e-6 of the activation
    scale) takes the same side of every ReLU and first-step gradients can be compared tightly; the margin found
    is stored.  Batch 128 (the benchmark's batch): loss curve, first logits and regulariser only; its inputs are
    re-derived at test time from the stored numpy RandomState seed (legacy stream, frozen by numpy's policy)."""
    out = {}
    steps, B = 5, 8
    RELU_MARGIN = 3e-6
    for variant in ("fan", "fix", "hint"):
        mod = load_ref("adp_qtz_%s.py" % variant)
        c = cfg()
        mod.config = c
        rs = np.random.RandomState(7)
        s = [float(v) for v in np.exp(rs.uniform(np.log(0.1), np.log(2.0), 20))]
        v = torch.from_numpy((0.1 * rs.standard_normal(10)).astype(np.float32))
        for tag, batch in (("", B), ("b128_", 128)):
            mod.beta = 1
            torch.manual_seed(0)
            model = mod.ResNet(c, 3) if variant == "fan" else mod.ResNet(c, s, v)
            model._keep_q = variant != "fan"
            fill_model_from_numpy(torch, model, 11)
            model.train()
            if batch == B:
                best = (-1.0, None)
                for xseed in range(1000, 4000):
                    rx = np.random.RandomState(xseed)
                    x = torch.from_numpy(rx.standard_normal((batch, 3, 32, 32)).astype(np.float32))
                    y = torch.from_numpy(rx.randint(0, 10, batch).astype(np.int64))
                    m = relu_margin(torch, model, x)
                    if m > best[0]:
                        best = (m, xseed)
                    if m >= RELU_MARGIN:
                        break
                margin, xseed = best
                rx = np.random.RandomState(xseed)
                x = torch.from_numpy(rx.standard_normal((batch, 3, 32, 32)).astype(np.float32))
                y = torch.from_numpy(rx.randint(0, 10, batch).astype(np.int64))
                print(variant, "batch 8: x seed", xseed, "relu margin %.3g" % margin)
            else:
                xseed = 500
                rx = np.random.RandomState(xseed)
                x = torch.from_numpy(rx.standard_normal((batch, 3, 32, 32)).astype(np.float32))
                y = torch.from_numpy(rx.randint(0, 10, batch).astype(np.int64))
            named = list(model.named_parameters())
            o1 = torch.optim.Adam([p for n, p in named if "p_c" in n], lr=1e-2)
            o2 = torch.optim.Adam([p for n, p in named if "q_level" in n], lr=1e-3)
            o3 = torch.optim.Adam([p for n, p in named if "p_c" not in n and "q_level" not in n], lr=1e-2)
            crit = torch.nn.CrossEntropyLoss()
            losses, regs, logit0 = [], [], None
            for it in range(steps):
                if it == 3:
                    mod.beta = 2  # exercise the sharpening-factor change
                logits = model.forward(x)
                reg = mod.model_loss(c, mod.beta, model)
                loss = crit(logits, y) + reg
                loss.backward()
                if it == 0:
                    logit0 = logits.detach().numpy().copy()
                    g0 = {n: (None if p.grad is None else p.grad.detach().numpy().copy()) for n, p in named}
                o1.step(); o2.step(); o3.step()
                o1.zero_grad(); o2.zero_grad(); o3.zero_grad()
                losses.append(loss.item()); regs.append(reg.item())
            k = variant + "_" + tag
            out[k + "loss"] = np.array(losses); out[k + "reg"] = np.array(regs); out[k + "logits0"] = f32(logit0)
            out[k + "xseed"] = np.int64(xseed)
            out[k + "p_sum"] = np.array([float(p.detach().double().sum()) for n, p in named])
            print(variant, tag, "losses", losses)
            if batch != B:
                continue
            out[k + "relu_margin"] = np.float64(margin)
            out[k + "s"] = np.array(s); out[k + "v"] = f32(v.numpy())
            out[k + "x"] = f32(x.numpy()); out[k + "y"] = y.numpy()
            for n in ("layer1.0.conv1", "layer2.0.conv1", "layer3.2.conv2", "linear"):
                out[k + "q_final_" + n] = f32(dict(named)[n + ".q_level"].detach().numpy())
                gq = g0[n + ".q_level"]
                out[k + "dq0_" + n] = np.zeros(0, np.float32) if gq is None else f32(gq)
                out[k + "dp0_head_" + n] = f32(g0[n + ".p_c"].reshape(-1)[:4096])
                out[k + "p_final_head_" + n] = f32(dict(named)[n + ".p_c"].detach().numpy().reshape(-1)[:4096])
            out[k + "g0_conv0"] = f32(g0["conv0.weight"]); out[k + "g0_bias"] = f32(g0["linear.bias"])
            out[k + "g0_bn0_w"] = f32(g0["bn0.weight"])
            # every small tensor's first-step gradient and final value (BatchNorm, bias, stem, levels): tight per-tensor checks
            for n, p in named:
                if p.numel() <= 4096:
                    out[k + "g0_all_" + n] = np.zeros(0, np.float32) if g0[n] is None else f32(g0[n])
                    out[k + "final_all_" + n] = f32(p.detach().numpy())
            out[k + "keys"] = np.array(list(model.state_dict().keys()))
            out[k + "param_names"] = np.array([n for n, _ in named])
    out["steps"] = np.int64(steps); out["seed_model"] = np.int64(11)
    np.savez_compressed(os.path.join(OUT, "train.npz"), **out)


def gen_block(torch):
    """Whole residual blocks of the reference (BatchNorm in training mode + ReLU + quantized convs +
    shortcut): parameters, input, upstream gradient -> output, input gradient, parameter gradients,
    running statistics after the forward pass."""
    mod = load_ref("adp_qtz_fan.py")
    mod.beta = 2
    rs = np.random.RandomState(2024)
    out = {}
    cases = [("residual", 16, False, 8, 4), ("residual", 16, True, 8, 3), ("preresidual", 16, False, 6, 2),
             ("residual", 32, False, 5, 2)]
    L = 4
    for i, (kind, cin, inc, H, N) in enumerate(cases):
        c = cfg(L=L)
        blk = mod.Residual(c, cin, inc) if kind == "residual" else mod.PreResidual(c, cin)
        blk.train()
        k = "b%02d_" % i
        with torch.no_grad():
            for name, p in blk.named_parameters():
                n = p.numel()
                if name.endswith("p_c"):
                    a = rs.random_sample(n)
                elif name.endswith("q_level"):
                    a = rs.uniform(-0.3, 0.3, n)
                elif name.endswith("weight"):
                    a = 1.0 + 0.2 * rs.standard_normal(n)
                else:
                    a = 0.2 * rs.standard_normal(n)
                p.copy_(torch.from_numpy(a.astype(np.float32)).reshape(p.shape))
                out[k + "param_" + name] = f32(p.detach().numpy())
        x = torch.from_numpy((rs.standard_normal((N, cin, H, H)) * 1.5 + 0.3).astype(np.float32)).requires_grad_(True)
        y = blk.forward(x * 1.0)  # (the block adds the shortcut in place: keep the leaf out of it)
        g = torch.from_numpy(rs.standard_normal(tuple(y.shape)).astype(np.float32))
        (y * g).sum().backward()
        out[k + "kind"] = np.array(kind); out[k + "shape"] = np.array([cin, int(inc), H, N], np.int64)
        out[k + "x"] = f32(x.detach().numpy()); out[k + "g"] = f32(g.numpy())
        out[k + "y"] = f32(y.detach().numpy()); out[k + "dx"] = f32(x.grad.numpy())
        for name, p in blk.named_parameters():
            out[k + "grad_" + name] = f32(p.grad.numpy())
        for name, b in blk.named_buffers():
            out[k + "buf_" + name] = np.asarray(b.detach().numpy())
    out["ncase"] = np.int64(len(cases))
    np.savez_compressed(os.path.join(OUT, "block.npz"), **out)
    print("block.npz: %d cases" % len(cases))


def gen_ckpt(torch):
    """A training checkpoint written the way the reference's train() writes it (adp_qtz_fan.py:331-340)
    after two optimisation steps of a small ResNet (n=1 block per stage, 4 levels), plus what the
    reference computes from that state: loss / logits of the next training step, and the hard-mode
    logits of adp_qtz_fan_test.py's model loaded from the same file (eval mode)."""
    mod = load_ref("adp_qtz_fan.py")
    modt = load_ref("adp_qtz_fan_test.py")
    c = cfg(L=4, Lfc=4)
    mod.beta = 2
    rs = np.random.RandomState(31)
    model = mod.ResNet(c, 1)
    fill_model_from_numpy(torch, model, 5)
    model.train()
    B = 6
    x = torch.from_numpy(rs.standard_normal((B, 3, 32, 32)).astype(np.float32))
    y = torch.from_numpy(rs.randint(0, 10, B).astype(np.int64))
    named = list(model.named_parameters())
    o1 = torch.optim.Adam([p for n, p in named if "p_c" in n], lr=1e-2)
    o2 = torch.optim.Adam([p for n, p in named if "q_level" in n], lr=1e-3)
    o3 = torch.optim.Adam([p for n, p in named if "p_c" not in n and "q_level" not in n], lr=1e-2)
    crit = torch.nn.CrossEntropyLoss()

    def step():
        logits = model.forward(x)
        loss = crit(logits, y) + mod.model_loss(c, mod.beta, model)
        loss.backward()
        o1.step(); o2.step(); o3.step()
        o1.zero_grad(); o2.zero_grad(); o3.zero_grad()
        return loss.item(), logits.detach().numpy().copy()

    losses = [step()[0], step()[0]]
    path = os.path.join(OUT, "ckpt_fan.pt")
    torch.save({
        "beta": mod.beta, "epoch": 3, "iter_idx": 1, "best_val_acc": np.mean([np.float32(12.5), np.float32(25.0)]),
        "model": model.state_dict(), "optimizer_prim": o1.state_dict(), "optimizer_q": o2.state_dict(),
        "optimizer": o3.state_dict(),
    }, path)
    # hard-mode logits of the test script's model with the same state (eval: running statistics)
    modt.beta = 2
    mt = modt.ResNet(c, 1)
    mt.load_state_dict(model.state_dict())
    mt.eval()
    with torch.no_grad():
        hard = mt.forward(x).numpy().copy()
    l3, logits3 = step()
    out = {"x": f32(x.numpy()), "y": y.numpy(), "losses_before": np.array(losses), "loss_next": np.float64(l3),
           "logits_next": f32(logits3), "hard_logits_eval": f32(hard), "beta": np.int64(2),
           "p_sum_after_next": np.array([float(p.detach().double().sum()) for n, p in named])}
    np.savez_compressed(os.path.join(OUT, "ckpt.npz"), **out)
    print("ckpt_fan.pt: %d bytes; losses %s -> %.6f" % (os.path.getsize(path), losses, l3))


def gen_baseline(torch):
    """The full-precision ResNet of adp_qtz_baseline.py (n=1 and n=3): logits, CE + model_loss, gradients
    after one backward pass, one SGD(momentum 0.9) step, and the scales / bias the fix script derives
    from such a model (adp_qtz_fix.py:208-215)."""
    mod = load_ref("adp_qtz_baseline.py")
    out = {}
    for n in (1, 3):
        rs = np.random.RandomState(70 + n)
        model = mod.ResNet(n)
        fill_model_from_numpy(torch, model, 21 + n)
        model.train()
        B = 4
        x = torch.from_numpy(rs.standard_normal((B, 3, 32, 32)).astype(np.float32))
        y = torch.from_numpy(rs.randint(0, 10, B).astype(np.int64))
        c = argparse.Namespace(l2_reg=1e-4)
        opt = torch.optim.SGD(model.parameters(), lr=0.1, momentum=0.9)
        losses = []
        k = "n%d_" % n
        for it in range(3):
            logits = model.forward(x)
            loss = torch.nn.functional.cross_entropy(logits, y) + mod.model_loss(c, model)
            loss.backward()
            if it == 0:
                named = dict(model.named_parameters())
                out[k + "logits0"] = f32(logits.detach().numpy())
                for name in ("conv0.weight", "layer2.0.conv1.weight", "layer3.0.conv2.weight", "linear.weight", "bn0.weight"):
                    out[k + "g0_" + name] = f32(named[name].grad.numpy())
            opt.step(); opt.zero_grad()
            losses.append(loss.item())
        sc, v = [], None
        for name, param in model.named_parameters():
            if ('conv' in name and 'weight' in name) or ('linear' in name and 'weight' in name):
                sc.append(torch.max(torch.abs(param)).item())
            elif 'linear' in name and 'bias' in name:
                v = param
        out[k + "x"] = f32(x.numpy()); out[k + "y"] = y.numpy(); out[k + "loss"] = np.array(losses)
        out[k + "s"] = np.array(sc, np.float64); out[k + "v"] = f32(v.detach().numpy())
        out[k + "keys"] = np.array(list(model.state_dict().keys()))
        out[k + "p_sum"] = np.array([float(p.detach().double().sum()) for _, p in model.named_parameters()])
        print("baseline n=%d losses %s, %d scales" % (n, losses, len(sc)))
    np.savez_compressed(os.path.join(OUT, "baseline.npz"), **out)


def gen_data(torch):
    """The reference's reader (adp_qtz_fan.py:452-503, `load_data`; the same text in fix / hint / baseline) run on a
    synthetic directory in the CIFAR-10 python format: six pickled batch files with a uint8 (n, 3072) b"data" array and
    a b"labels" list.  Batch sizes differ per file and data_batch_5 holds an odd number of images, so the split
    `int(len / 2)` is exercised.  Stored: the raw batch contents (the test rebuilds the directory from them) and the
    arrays the reference returns for "train" / "valid" / "test", plus the (image, label) pairs its CIFAR10Dataset
    hands out without a transform (adp_qtz_fan.py:506-530)."""
    import pickle
    import tempfile
    out = {}
    rs = np.random.RandomState(77)
    sizes = {"data_batch_1": 6, "data_batch_2": 4, "data_batch_3": 5, "data_batch_4": 3, "data_batch_5": 7, "test_batch": 5}
    with tempfile.TemporaryDirectory() as d:
        for name, n in sizes.items():
            data = rs.randint(0, 256, (n, 3072)).astype(np.uint8)
            labels = [int(v) for v in rs.randint(0, 10, n)]
            with open(os.path.join(d, name), "wb") as f:
                pickle.dump({b"batch_label": name.encode(), b"labels": labels, b"data": data,
                             b"filenames": [b"img_%d.png" % i for i in range(n)]}, f)
            out["raw_" + name + "_data"] = data
            out["raw_" + name + "_labels"] = np.array(labels, np.int64)
        for variant in ("fan", "fix", "hint", "baseline"):
            mod = load_ref("adp_qtz_%s.py" % variant)
            for split in ("train", "valid", "test"):
                data, label = mod.load_data(d, split)
                if variant == "fan":
                    out[split + "_data"] = np.ascontiguousarray(data)
                    out[split + "_label"] = np.asarray(label)
                    out[split + "_dtype"] = np.array([str(data.dtype), str(np.asarray(label).dtype)])
                else:  # the four scripts carry the same reader: one fixture serves all
                    assert np.array_equal(out[split + "_data"], data) and np.array_equal(out[split + "_label"], label), variant
            try:
                mod.load_data(d, "nope")
                raise AssertionError("reference accepted an unknown split")
            except ValueError as e:
                out["bad_split_message"] = np.array(str(e))
        fan = load_ref("adp_qtz_fan.py")
        ds = fan.CIFAR10Dataset(d, "valid", None)
        out["dataset_len"] = np.int64(len(ds))
        out["dataset_items"] = np.stack([np.asarray(ds[i][0]) for i in range(len(ds))])
        out["dataset_labels"] = np.array([int(ds[i][1]) for i in range(len(ds))], np.int64)
    out["names"] = np.array(list(sizes))
    np.savez_compressed(os.path.join(OUT, "data.npz"), **out)
    print("data.npz: train %s valid %s test %s" % (out["train_data"].shape, out["valid_data"].shape, out["test_data"].shape))


def main():
    import torch
    torch.set_num_threads(8)
    which = sys.argv[1:] or ["synth", "conv", "linear", "levels", "train", "block", "ckpt", "baseline", "data"]
    for w in which:
        globals()["gen_" + w](torch)


if __name__ == "__main__":
    main()
