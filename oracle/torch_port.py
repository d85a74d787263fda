"""CPU restatement (plain PyTorch, fp32) of the reference's quantized ResNet-20 training step.

TEST INFRASTRUCTURE ONLY (checker for tests/ and the cpu_baseline leg of bench.py); the product
package never imports this file.  Parity status: PINNED against tests/golden/train.npz (five
optimisation steps of the reference's own classes) by tests/test_oracle_golden.py.

Restates: weight synthesis adp_qtz_fan.py:563-568, conv :574-601 (either as the nine shifted
matmuls of the reference or as the equivalent F.conv2d), linear :633-648, blocks :651-734,
model_loss :737-754, optimizer split :208-213, level initialisers fix:661-665 / hint:657-662.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


def synth(p_c, q_level, beta):
    pn = p_c / torch.sqrt(torch.sum(p_c ** 2, dim=-1, keepdim=True))
    e = torch.exp(beta * pn)
    return torch.matmul(e / torch.sum(e, dim=-1, keepdim=True), q_level)


def conv_taps(x, W, stride):
    """Nine shifted matmuls accumulated tap by tap (width tap outer, height tap inner)."""
    x = F.pad(x, (1, 1, 1, 1))
    N, C, Hp, Wp = x.shape
    h, w = Hp - 2, Wp - 2
    out = None
    for i in range(3):
        for j in range(3):
            cur = x[:, :, j:j + h:stride, i:i + w:stride]
            hn, wn = cur.shape[2], cur.shape[3]
            o = torch.matmul(W[:, :, j, i], cur.reshape(N, C, hn * wn)).reshape(N, W.shape[0], hn, wn)
            out = o if out is None else out + o
    return out


class PortModel(nn.Module):
    """variant in {"fan","fix","hint"}; conv_impl in {"conv2d","taps"}; state_dict keys as the reference."""

    def __init__(self, variant="fan", L=32, s=None, v=None, conv_impl="conv2d", liar=True):
        super().__init__()
        self.variant, self.conv_impl, self.beta = variant, conv_impl, 1
        Lq = L - 1 if variant == "fix" else L
        self.conv0 = nn.Conv2d(3, 16, 3, 1, 1, bias=False)
        self.bn0 = nn.BatchNorm2d(16)
        self.specs = []  # (prefix, cin, cout, stride)
        si = iter(s[1:19]) if s is not None else None

        def qconv(prefix, cin, cout, stride):
            p = nn.Parameter(torch.rand(cout, cin, 3, 3, Lq))
            q = nn.Parameter(torch.empty(Lq), requires_grad=variant != "fix")
            self._init_q(q, L, cin * 9, next(si) if si is not None else None, liar)
            self.register_parameter_path(prefix + ".p_c", p)
            self.register_parameter_path(prefix + ".q_level", q)
            self.specs.append((prefix, cin, cout, stride))

        self._mods = {}
        plan = [("layer1", 16, 16, False), ("layer2", 16, 32, True), ("layer3", 32, 64, True)]
        for name, cin, cout, down in plan:
            setattr(self, name, nn.ModuleList([nn.Module() for _ in range(3)]))
            for b in range(3):
                blk = getattr(self, name)[b]
                bin_ = cin if b == 0 else cout
                first_pre = (name == "layer1" and b == 0)
                if not first_pre:
                    blk.bn1 = nn.BatchNorm2d(bin_)
                blk.conv1 = nn.Module()
                blk.bn2 = nn.BatchNorm2d(cout)
                blk.conv2 = nn.Module()
                qconv("%s.%d.conv1" % (name, b), bin_, cout, 2 if (down and b == 0) else 1)
                qconv("%s.%d.conv2" % (name, b), cout, cout, 1)
        self.bnlast = nn.BatchNorm2d(64)
        self.linear = nn.Module()
        self.linear.p_c = nn.Parameter(torch.rand(64, 10, Lq))
        self.linear.q_level = nn.Parameter(torch.empty(Lq), requires_grad=variant != "fix")
        self._init_q(self.linear.q_level, L, 64, s[19] if s is not None else None, liar)
        self.linear.bias = nn.Parameter(torch.zeros(10) if v is None else v.detach().clone().float())

    def register_parameter_path(self, path, p):
        mod = self
        parts = path.split(".")
        for name in parts[:-1]:
            mod = mod[int(name)] if name.isdigit() else getattr(mod, name)
        mod.register_parameter(parts[-1], p)

    def _init_q(self, q, L, fan_in, s, liar):
        with torch.no_grad():
            if self.variant == "fan":
                b = math.sqrt(3.0) * math.sqrt(2.0 / 6.0) / math.sqrt(fan_in)
                q.uniform_(-b, b)
                return
            t2 = math.floor(math.log(s, 2)) if (liar or self.variant == "hint") else round(math.log(s, 2))
            t3 = torch.arange(t2 - L / 2 + (2 if self.variant == "fix" else 1), t2 + 1)
            pos = 2 ** t3
            parts = [(-pos).sort()[0], torch.zeros(1), pos] if self.variant == "fix" else [(-pos).sort()[0], pos]
            q.copy_(torch.cat(parts))

    def _get(self, path):
        mod = self
        for name in path.split("."):
            mod = mod[int(name)] if name.isdigit() else getattr(mod, name)
        return mod

    def _conv(self, prefix, x, stride):
        m = self._get(prefix)
        W = synth(m.p_c, m.q_level, self.beta)
        return F.conv2d(x, W, stride=stride, padding=1) if self.conv_impl == "conv2d" else conv_taps(x, W, stride)

    def forward(self, x):
        out = F.relu(self.bn0(self.conv0(x)))
        for name, cin, cout, down in (("layer1", 16, 16, False), ("layer2", 16, 32, True), ("layer3", 32, 64, True)):
            for b in range(3):
                blk = getattr(self, name)[b]
                pre = "%s.%d" % (name, b)
                first = down and b == 0
                if name == "layer1" and b == 0:
                    o = self._conv(pre + ".conv1", out, 1)
                    sc = out
                else:
                    o = self._conv(pre + ".conv1", F.relu(blk.bn1(out)), 2 if first else 1)
                    sc = out
                    if first:
                        sc = F.pad(F.avg_pool2d(out, 2), (0, 0, 0, 0, cin // 2, cin // 2))
                o = self._conv(pre + ".conv2", F.relu(blk.bn2(o)), 1)
                out = o + sc
        out = F.avg_pool2d(F.relu(self.bnlast(out)), 8).view(out.size(0), -1)
        return torch.matmul(out, synth(self.linear.p_c, self.linear.q_level, self.beta)) + self.linear.bias


def model_loss(l2_reg, beta, model):
    loss, p_c = 0, None
    for name, param in model.named_parameters():
        if "weight" in name:
            loss = loss + torch.sum(param ** 2)
        if "p_c" in name:
            p_c = param
        if "q_level" in name:
            loss = loss + torch.sum(synth(p_c, param, beta) ** 2)
    return l2_reg * loss


def optimizers(model, prim_lr=1e-2, q_lr=1e-3, lr=1e-2):
    named = list(model.named_parameters())
    return (torch.optim.Adam([p for n, p in named if "p_c" in n], lr=prim_lr),
            torch.optim.Adam([p for n, p in named if "q_level" in n], lr=q_lr),
            torch.optim.Adam([p for n, p in named if "p_c" not in n and "q_level" not in n], lr=lr))


def train_step(model, opts, x, y, l2_reg=5e-4):
    logits = model(x)
    reg = model_loss(l2_reg, model.beta, model)
    loss = F.cross_entropy(logits, y) + reg
    loss.backward()
    for o in opts:
        o.step()
    for o in opts:
        o.zero_grad()
    return loss.item(), reg.item(), logits.detach()
