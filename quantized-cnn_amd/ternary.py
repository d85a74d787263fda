"""Ternary (TWN, TTQ) and incremental power-of-two (INQ) quantized layers behind the module API of the
adaptive-quantisation variants: the methods the reference's README cites (README.md:3-29) next to its own soft
level assignment.  The reference's scripts implement none of them, so nothing here can be compared with
reference code: the arithmetic follows the papers (Li, Zhang, Liu 2016; Zhu, Han, Mao, Dally 2017; Zhou et al.
2017) and is checked against a CPU restatement written from the same papers - parity unpinned.

    mode_q = "twn" : w~ = alpha * sign(w) above delta = 0.7 * mean|w|, alpha = mean of |w| above delta; straight-through
    mode_q = "ttq" : w~ = +Wp / -Wn above / below +-delta with learned Wp, Wn; scaled straight-through gradients
    mode_q = "inq" : full-precision weights whose largest-magnitude share is frozen at powers of two, stage by stage

A layer on its own, and every layer outside a training step, convolves from the 1-byte codes and a three-entry codebook,
decoded while the weight tile is staged into LDS (the scale rides in the codebook); the backward pass uses the decoded
weights.  In a training step of a model that quantises all its layers together (quantize_all) the decoded weights are
an autograd tensor and the 3x3 layers run on the fused convolution nodes of the adaptive layers.
"""
import math

import torch
import torch.nn as nn

from . import layers, ops
from .layers import model_loss  # noqa: F401  (l2 on every "weight" parameter, adp_qtz_fan.py:737-743)

beta = 1  # unused by these layers; present so that train.TrainStep can treat this module like a variant module


def _conv_quantized(x, q, layer):
    """Forward convolution on a quantised weight: from the 1-byte codes + codebook where a kernel decodes them while it
    stages the weight tile into LDS (the ResNet-20 shapes, small channel counts, 5x5 / 7x7), from the decoded weights
    on the many-channel 3x3 / 1x1 kernels, whose weight fragments go from memory straight to registers."""
    cout, cin = q["codes"].shape[0], q["codes"].shape[1]
    many = cin % 16 == 0 and ((layer.ksize == 3 and cout >= 48) or (layer.ksize == 1 and cout % 16 == 0 and layer.padding == 0))
    if many:
        return ops.conv_forward(x, q["Wt"], layer.stride, pad=layer.padding)
    return ops.conv_forward_codes(x, q["codes"], q["q"], layer.stride, layer.padding)


class _TernConv(torch.autograd.Function):
    """y = conv(x, w~): forward from codes + codebook, backward through the decoded w~ and the quantiser's
    straight-through rule.  q: pre-computed dict of ops.tern_forward for this weight (or None)."""

    @staticmethod
    def forward(ctx, x, w, Wp, Wn, layer, q):
        x = x.contiguous()
        if q is None:
            q = layer.quantize()
        ctx.layer, ctx.q = layer, q
        ctx.save_for_backward(x, w, Wp, Wn)
        return _conv_quantized(x, q, layer)

    @staticmethod
    def backward(ctx, dy):
        x, w, Wp, Wn = ctx.saved_tensors
        L, q = ctx.layer, ctx.q
        dy = dy.contiguous()
        dx = ops.conv_dgrad(dy, q["Wt"], x.shape, L.stride, pad=L.padding) if ctx.needs_input_grad[0] else None
        dWt = ops.conv_wgrad(x, dy, L.stride, ks=L.ksize, pad=L.padding)
        if L.mode_q == "ttq":
            dW, dp, dn = ops.tern_backward([w], [dWt], "ttq", stats=[q["stats"]], Wp=[Wp], Wn=[Wn])
            return dx, dW[0], dp[0].reshape(Wp.shape), dn[0].reshape(Wn.shape), None, None
        dW, _, _ = ops.tern_backward([w], [dWt], "twn")
        return dx, dW[0], None, None, None, None


class _TernLinear(torch.autograd.Function):
    """y = x @ w~ + bias with w (in, out) like the reference's MyLinear (adp_qtz_fan.py:646-647)."""

    @staticmethod
    def forward(ctx, x, w, bias, Wp, Wn, layer, q):
        x = x.contiguous()
        if q is None:
            q = layer.quantize()
        ctx.layer, ctx.q = layer, q
        ctx.save_for_backward(x, w, Wp, Wn)
        return ops.linear_forward(x, q["Wt"], bias)

    @staticmethod
    def backward(ctx, dy):
        x, w, Wp, Wn = ctx.saved_tensors
        L, q = ctx.layer, ctx.q
        dx, dWt, db = ops.linear_backward(x, q["Wt"], dy.contiguous(), need_dx=ctx.needs_input_grad[0])
        if L.mode_q == "ttq":
            dW, dp, dn = ops.tern_backward([w], [dWt], "ttq", stats=[q["stats"]], Wp=[Wp], Wn=[Wn])
            return dx, dW[0], db, dp[0].reshape(Wp.shape), dn[0].reshape(Wn.shape), None, None
        dW, _, _ = ops.tern_backward([w], [dWt], "twn")
        return dx, dW[0], db, None, None, None, None


class _QuantizeAll(torch.autograd.Function):
    """The quantised weights of ALL ternary layers of a model as one autograd node: forward = one ops.tern_forward
    call (five launches for every layer together), backward = one ops.tern_backward call once the weight gradients
    of every layer have arrived (a launch and its second stage instead of one pair per layer).
    apply(mode_q, factor, threshold_from_max, n, sink, w_0 .. w_{n-1} [, Wp_0 .. Wp_{n-1}, Wn_0 .. Wn_{n-1}])
    -> (Wt_0, .., Wt_{n-1}); sink (a list) receives the per-layer dicts of ops.tern_forward."""

    @staticmethod
    def forward(ctx, mode_q, factor, threshold_from_max, n, sink, *ts):
        ws = [t.detach() for t in ts[:n]]
        ctx.mode_q, ctx.n = mode_q, n
        if mode_q == "ttq":
            Wp, Wn = [t.detach() for t in ts[n:2 * n]], [t.detach() for t in ts[2 * n:3 * n]]
            qs = ops.tern_forward(ws, "ttq", factor, threshold_from_max, Wp, Wn)
        else:
            qs = ops.tern_forward(ws, "twn", factor, threshold_from_max)
        sink.extend(qs)
        ctx.stats = [q["stats"] for q in qs]
        ctx.save_for_backward(*ts)
        return tuple(q["Wt"] for q in qs)

    @staticmethod
    def backward(ctx, *dWt):
        ts, n = ctx.saved_tensors, ctx.n
        ws = list(ts[:n])
        g = [d.contiguous() for d in dWt]
        if ctx.mode_q == "ttq":
            Wp, Wn = list(ts[n:2 * n]), list(ts[2 * n:3 * n])
            dW, dp, dn = ops.tern_backward(ws, g, "ttq", stats=ctx.stats, Wp=Wp, Wn=Wn)
            return (None,) * 5 + tuple(dW) + tuple(d.reshape(p.shape) for d, p in zip(dp, Wp)) + tuple(d.reshape(p.shape) for d, p in zip(dn, Wn))
        dW, _, _ = ops.tern_backward(ws, g, "twn")
        return (None,) * 5 + tuple(dW)


class _ConvFromQuantized(torch.autograd.Function):
    """y = conv(x, w~) for a w~ that _QuantizeAll produced: forward from its codes + codebook, backward through the
    decoded w~; the weight gradient leaves as the gradient of w~ (the quantiser's rule is applied by _QuantizeAll)."""

    @staticmethod
    def forward(ctx, x, Wt, layer, q):
        x = x.contiguous()
        ctx.layer = layer
        ctx.save_for_backward(x, Wt)
        return _conv_quantized(x, q, layer)

    @staticmethod
    def backward(ctx, dy):
        x, Wt = ctx.saved_tensors
        L = ctx.layer
        dy = dy.contiguous()
        dx = ops.conv_dgrad(dy, Wt, x.shape, L.stride, pad=L.padding) if ctx.needs_input_grad[0] else None
        return dx, ops.conv_wgrad(x, dy, L.stride, ks=L.ksize, pad=L.padding), None, None


class _LinearFromQuantized(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, Wt, bias):
        x = x.contiguous()
        ctx.save_for_backward(x, Wt)
        return ops.linear_forward(x, Wt, bias)

    @staticmethod
    def backward(ctx, dy):
        x, Wt = ctx.saved_tensors
        dx, dWt, db = ops.linear_backward(x, Wt, dy.contiguous(), need_dx=ctx.needs_input_grad[0])
        return dx, dWt, db


class _TernBase(nn.Module):
    def _setup_quantizer(self, mode_q, factor, threshold_from_max):
        if mode_q not in ("twn", "ttq"):
            raise ValueError("mode_q must be 'twn' or 'ttq' (got %r)" % (mode_q,))
        self.mode_q, self.factor, self.threshold_from_max = mode_q, float(factor), bool(threshold_from_max)
        self._shared = None
        with torch.no_grad():
            a = self.weight.abs()
            delta = self.factor * (a.max() if self.threshold_from_max else a.mean())
            big = a[a > delta]
            init = float(big.mean()) if big.numel() else float(a.mean())
        if mode_q == "ttq":  # learned scales start at TWN's alpha (Zhu et al. 2017 initialise from the pretrained weights)
            self.Wp = nn.Parameter(torch.full((1,), init))
            self.Wn = nn.Parameter(torch.full((1,), init))
        else:
            self.register_parameter("Wp", None)
            self.register_parameter("Wn", None)

    def quantize(self, packed=False):
        """-> dict of ops.tern_forward for this layer's weight (codes, Wt, stats, q[, packed])."""
        if self._shared is not None and not packed:
            q, self._shared = self._shared, None
            return q[0] if isinstance(q, tuple) else q
        w = self.weight.detach()
        if self.mode_q == "ttq":
            return ops.tern_forward([w], "ttq", self.factor, self.threshold_from_max, [self.Wp.detach()], [self.Wn.detach()], packed=packed)[0]
        return ops.tern_forward([w], "twn", self.factor, self.threshold_from_max, packed=packed)[0]


class TernaryConv2d(_TernBase):
    def __init__(self, in_channel, out_channel, ksize=3, stride=1, padding=1, mode_q="twn", factor=0.7, threshold_from_max=False):
        super().__init__()
        self.ksize, self.stride, self.padding = ksize, stride, padding
        self.weight = nn.Parameter(torch.empty(out_channel, in_channel, ksize, ksize))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        self._setup_quantizer(mode_q, factor, threshold_from_max)

    def forward(self, x, res=None, bn=None, split=False):
        """Same call convention as layers.QConv2d.forward (res added to the output, bn applied to the input, split
        returns the raw input or its pooled shortcut next to the output)."""
        if isinstance(self._shared, tuple) and torch.is_grad_enabled():
            # Training step of a model that quantised all its layers together (quantize_all, or a fused step that also
            # hands down the slot of a flat gradient buffer): the decoded weights are an autograd tensor, so the layer runs
            # on the same fused nodes as the adaptive layers (layers.QConv2d.forward): shortcut added in the convolution's
            # epilogue, batch statistics out of the producing convolution, batch-norm backward coupled to the
            # input-gradient kernel, weight gradient deferred into the slot.
            shared, self._shared = self._shared, None
            Wt, slot = shared[1], (shared[2] if len(shared) > 2 else None)
            usual = self.ksize == 3 and self.padding == 1
            if usual and bn is not None:
                return layers.bn_relu_conv(bn, x, Wt, self.stride, slot, res, split, stats=self.training)
            if usual or slot is not None:
                xs = x
                if bn is not None:
                    x, xs = layers.bn_relu(bn, x, split=True) if split else (layers.bn_relu(bn, x), x)
                    y = layers.conv3x3(x, Wt, self.stride, slot, res, stats=self.training and usual, pad=self.padding)
                    if isinstance(split, tuple):
                        return y, layers._Shortcut.apply(xs, *split)
                    return (y, xs) if split else y
                if isinstance(split, tuple):
                    return layers.conv3x3(x, Wt, self.stride, slot, res, stats=self.training and usual, pad=self.padding), layers._Shortcut.apply(x, *split)
                return layers.conv3x3(x, Wt, self.stride, slot, res, stats=self.training and usual, split=bool(split), pad=self.padding)
            self._shared = shared  # other kernel sizes without a slot: the generic path below
        xs = x
        if bn is not None:
            if split:
                z, xs = layers.bn_relu(bn, x, split=True)
            else:
                z = layers.bn_relu(bn, x)
        else:
            z = x
        if isinstance(self._shared, tuple) and torch.is_grad_enabled():  # quantised with the model's other layers (quantize_all)
            shared, self._shared = self._shared, None
            y = _ConvFromQuantized.apply(z, shared[1], self, shared[0])
        else:
            y = _TernConv.apply(z, self.weight, self.Wp, self.Wn, self, None)
        if res is not None:
            y = y + res
        if isinstance(split, tuple):
            return y, layers._Shortcut.apply(xs, *split)
        return (y, xs) if split else y


class TernaryLinear(_TernBase):
    def __init__(self, in_feature, out_feature, mode_q="twn", factor=0.7, threshold_from_max=False):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(in_feature, out_feature))
        bound = 1 / math.sqrt(in_feature)
        nn.init.uniform_(self.weight, -bound * math.sqrt(3.0), bound * math.sqrt(3.0))
        self.bias = nn.Parameter(torch.empty(out_feature).uniform_(-bound, bound))
        self._setup_quantizer(mode_q, factor, threshold_from_max)

    def forward(self, x):
        if isinstance(self._shared, tuple) and torch.is_grad_enabled():
            shared, self._shared = self._shared, None
            if len(shared) > 2:  # a fused step: the weight gradient lands in its slot of the flat buffer
                return layers._LinearIO.apply(x, shared[1], self.bias, shared[2])
            return _LinearFromQuantized.apply(x, shared[1], self.bias)
        return _TernLinear.apply(x, self.weight, self.bias, self.Wp, self.Wn, self, None)


class _InqMask(torch.autograd.Function):
    """Effective weight of an INQ layer: the parameter where it is still trainable, the frozen level elsewhere; the
    gradient of frozen weights is zeroed (Zhou et al. 2017, eq. 8).  Taking the frozen values from the codes rather
    than from the parameter keeps them fixed under optimizers that move a coordinate without a gradient (momentum,
    Adam's running averages)."""

    @staticmethod
    def forward(ctx, w, mask, maskb, frozen):
        ctx.save_for_backward(w, mask)
        return torch.where(maskb, w, frozen)

    @staticmethod
    def backward(ctx, g):
        w, mask = ctx.saved_tensors
        dW, _, _ = ops.tern_backward([w.detach()], [g.contiguous()], "inq", masks=[mask])
        return dW[0], None, None, None


class _InqAll(torch.autograd.Function):
    """The effective weights of ALL INQ layers of a model as one autograd node, so that the masking of the gradients
    (ops.tern_backward in its INQ mode) is one call for every layer instead of one per layer.
    apply(layers, w_0, .., w_{n-1}) -> (W_0, .., W_{n-1})."""

    @staticmethod
    def forward(ctx, ls, *ws):
        ctx.masks = [l.mask for l in ls]
        ctx.save_for_backward(*ws)
        outs = []
        for l, w in zip(ls, ws):
            maskb, frozen = l._frozen_values()
            outs.append(torch.where(maskb, w, frozen))
        return tuple(outs)

    @staticmethod
    def backward(ctx, *g):
        ws = [w.detach() for w in ctx.saved_tensors]
        dW, _, _ = ops.tern_backward(ws, [d.contiguous() for d in g], "inq", masks=ctx.masks)
        return (None,) + tuple(dW)


def inq_all(ls):
    """Effective weights of all INQ layers of a model through one autograd node (training steps only); each layer picks
    its weight up in its next forward."""
    if not ls or not torch.is_grad_enabled() or not all(l.weight.requires_grad for l in ls):
        return
    for l, w in zip(ls, _InqAll.apply(ls, *[l.weight for l in ls])):
        l._shared = w


class INQConv2d(nn.Module):
    """Convolution whose weights are frozen at powers of two stage by stage: quantize_portion(0.5), (0.75), (0.875),
    (1.0) between retraining phases.  `mask` (1 = still trainable) and `codes` (level codes of the frozen weights, 5 bits
    for bits=5) are buffers, so they travel with the state_dict."""

    def __init__(self, in_channel, out_channel, ksize=3, stride=1, padding=1, bits=5):
        super().__init__()
        self.ksize, self.stride, self.padding, self.bits = ksize, stride, padding, bits
        self.weight = nn.Parameter(torch.empty(out_channel, in_channel, ksize, ksize))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        self.register_buffer("mask", torch.ones(self.weight.shape, dtype=torch.uint8))
        self.register_buffer("codes", torch.full(self.weight.shape, 255, dtype=torch.uint8))
        self.register_buffer("exponents", torch.zeros(2, dtype=torch.int32))  # (n1, n2), fixed at the first stage
        self._have_exp = False
        self._shared = None  # effective weight handed down by the owning model for the current step (inq_all)
        self._slot = None
        self._cache = None

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        """mask / codes / exponents travel with the state_dict; whether a first stage has fixed the exponents is derived
        from them, so that a reloaded layer computes with its frozen levels and continues its stages."""
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)
        self.buffers_changed()

    def buffers_changed(self):
        """To be called after mask / codes / exponents were written behind torch's version counters (through .data, by a
        collective, by a kernel): re-derives whether a stage has run and refreshes the decoded frozen values in place."""
        self._have_exp = bool((self.mask == 0).any()) or bool(self.exponents.any())
        self._refresh_frozen()

    def quantize_portion(self, fraction):
        """Freeze the largest-magnitude weights until `fraction` of all weights are frozen."""
        w = self.weight.data
        n = w.numel()
        if not self._have_exp:
            s = float(ops.absmax([w.contiguous()])[0]) if w.is_cuda else float(w.abs().max())
            n1 = int(math.floor(math.log2(4.0 * s / 3.0)))
            self.exponents.copy_(torch.tensor([n1, n1 + 1 - (2 ** (self.bits - 1)) // 2], dtype=torch.int32))
            self._have_exp = True
        frozen = n - int(self.mask.sum())
        if frozen:
            w.copy_(self.effective_weight().detach())  # drop whatever drift the optimizer left under the mask
        k = int(round(float(fraction) * n)) - frozen
        if k <= 0:
            return 0
        free = w.abs()[self.mask.bool()]
        tau = float(torch.topk(free, min(k, free.numel()), largest=True, sorted=True).values[-1])
        n1, n2 = (int(v) for v in self.exponents.tolist())
        ops.inq_partition(w, self.mask, tau, n1, n2, self.codes)
        self._refresh_frozen()  # (the kernel wrote mask and codes behind torch's version counters)
        return k

    def _refresh_frozen(self):
        """Bring the boolean mask and the decoded frozen values up to date with the buffers.  The two tensors are
        allocated once per device and from then on written IN PLACE: a training step captured as a HIP graph
        (train.CapturedStep, TrainStep(use_graph=True)) holds their addresses, so a new INQ stage or a state_dict load
        must change their contents, not replace them - replaced tensors would leave the replayed graph reading freed
        memory and the previous partition."""
        dev = self.weight.device
        q = self.codebook()
        maskb = self.mask.bool()
        frozen = q[self.codes.clamp(max=q.numel() - 1).long()]
        c = self._cache
        if c is None or c[1].device != dev or c[1].shape != maskb.shape:
            self._cache = c = [None, maskb, frozen]
        else:
            c[1].copy_(maskb)
            c[2].copy_(frozen)
        c[0] = (self.codes._version, self.mask._version, self.exponents._version, dev)

    def _frozen_values(self):
        """-> (boolean mask of the trainable weights, fp32 values of the frozen ones decoded from the codes).  Before the
        first stage the mask is all ones (the values are never selected), so the effective weight has the same form
        - where(mask, w, frozen) - in every stage, and a step captured before a stage stays valid after it.  Persistent
        tensors (_refresh_frozen), refreshed after quantize_portion, state_dict loads, device moves and whenever the
        buffers were written through torch."""
        key = (self.codes._version, self.mask._version, self.exponents._version, self.weight.device)
        if self._cache is None or self._cache[0] != key:
            self._refresh_frozen()
        return self._cache[1], self._cache[2]

    def effective_weight(self):
        """The weights the layer computes with: trainable entries from the parameter, frozen ones from their codes."""
        if self._shared is not None and torch.is_grad_enabled():  # computed with the model's other layers (inq_all)
            w, self._shared = self._shared, None
            if isinstance(w, tuple):  # a fused step: (effective weight, its slot in the flat gradient buffer)
                self._slot = w[1]
                return w[0]
            return w
        maskb, frozen = self._frozen_values()
        return _InqMask.apply(self.weight, self.mask, maskb, frozen)

    def codebook(self):
        """Level table of `codes`: index 0 = 0, 1 + 2 * (k - n2) + s = (-1)^s * 2^k."""
        n1, n2 = (int(v) for v in self.exponents.tolist())
        q = [0.0]
        for k in range(n2, n1 + 1):
            q += [2.0 ** k, -(2.0 ** k)]
        return torch.tensor(q, dtype=torch.float32, device=self.weight.device)

    def forward(self, x, res=None, bn=None, split=False):
        self._slot = None
        w = self.effective_weight()
        slot, self._slot = self._slot, None
        usual = self.ksize == 3 and self.padding == 1
        if bn is not None and usual and torch.is_grad_enabled():
            # the fused nodes of the adaptive layers (layers.QConv2d.forward)
            return layers.bn_relu_conv(bn, x, w, self.stride, slot, res, split, stats=self.training)
        xs = x
        if bn is not None:
            if split:
                z, xs = layers.bn_relu(bn, x, split=True)
            else:
                z = layers.bn_relu(bn, x)
        else:
            z = x
        if bn is None and split is True and torch.is_grad_enabled():
            return layers.conv3x3(z, w, self.stride, slot, res, stats=self.training and usual, split=True, pad=self.padding)
        y = layers.conv3x3(z, w, self.stride, slot, res, stats=self.training and usual and torch.is_grad_enabled(), pad=self.padding)
        if isinstance(split, tuple):
            return y, layers._Shortcut.apply(xs, *split)
        return (y, xs) if split else y

    def forward_codes(self, x):
        """Inference from the codes alone once every weight is frozen (decoded inside the convolution kernel)."""
        if int(self.mask.sum()) != 0:
            raise RuntimeError("forward_codes: %d weights are still trainable" % int(self.mask.sum()))
        return ops.conv_forward_codes(x.contiguous(), self.codes, self.codebook(), self.stride, self.padding)


def quantize_all(ls):
    """One quantiser call for all ternary layers of a model (five launches) instead of one per layer; with gradients on,
    one autograd node per group of equal settings, whose backward applies the quantiser's gradient rule to all layers
    in one call as well.  Each layer picks its result up in its next forward."""
    if not ls:
        return
    for mode in ("twn", "ttq"):
        grp = [l for l in ls if l.mode_q == mode]
        for key in sorted({(l.factor, l.threshold_from_max) for l in grp}):
            g = [l for l in grp if (l.factor, l.threshold_from_max) == key]
            if torch.is_grad_enabled() and all(l.weight.requires_grad for l in g):
                sink = []
                ts = [l.weight for l in g] + ([l.Wp for l in g] + [l.Wn for l in g] if mode == "ttq" else [])
                Wts = _QuantizeAll.apply(mode, key[0], key[1], len(g), sink, *ts)
                for l, q, Wt in zip(g, sink, Wts):
                    l._shared = (q, Wt)
                continue
            ws = [l.weight.detach() for l in g]
            if mode == "ttq":
                qs = ops.tern_forward(ws, "ttq", key[0], key[1], [l.Wp.detach() for l in g], [l.Wn.detach() for l in g])
            else:
                qs = ops.tern_forward(ws, "twn", key[0], key[1])
            for l, q in zip(g, qs):
                l._shared = q


def make_conv(config, in_channel, out_channel, ksize=3, stride=1, padding=1):
    """Convolution of the flavour config.mode_q names ("twn" | "ttq" | "inq").
    Threshold: config.tern_factor / config.tern_threshold_from_max; the default for BOTH ternary modes is TWN's rule
    delta = 0.7 * mean|w| (what BASELINE's north star names).  The TTQ paper itself uses delta = t * max|w| with t = 0.05:
    set tern_threshold_from_max=True, tern_factor=0.05 for the published TTQ threshold."""
    mode_q = getattr(config, "mode_q", "twn")
    if mode_q == "inq":
        return INQConv2d(in_channel, out_channel, ksize, stride, padding, bits=getattr(config, "inq_bits", 5))
    return TernaryConv2d(in_channel, out_channel, ksize, stride, padding, mode_q, getattr(config, "tern_factor", 0.7),
                         getattr(config, "tern_threshold_from_max", False))


class Residual(layers._Block):
    def __init__(self, config, in_channel, increase_dim):
        super().__init__()
        out_channel, stride = (in_channel * 2, 2) if increase_dim else (in_channel, 1)
        self.bn1 = nn.BatchNorm2d(in_channel)
        self.conv1 = make_conv(config, in_channel, out_channel, 3, stride, 1)
        self.bn2 = nn.BatchNorm2d(out_channel)
        self.conv2 = make_conv(config, out_channel, out_channel, 3, 1, 1)
        self.shortcut = self._shortcut(in_channel, increase_dim)

    def forward(self, x):
        return self._residual_forward(x)


class PreResidual(layers._Block):
    def __init__(self, config, in_channel, increase_dim=False):
        super().__init__()
        self.conv1 = make_conv(config, in_channel, in_channel, 3, 1, 1)
        self.bn2 = nn.BatchNorm2d(in_channel)
        self.conv2 = make_conv(config, in_channel, in_channel, 3, 1, 1)

    def forward(self, x):
        return self._preresidual_forward(x)


class ResNet(nn.Module):
    """The reference's ResNet-20 graph (adp_qtz_fan.py:696-734: full-precision conv0, 3 stages of n blocks, pre-activation,
    pooled zero-padded shortcuts) with ternary / INQ convolutions in place of the adaptive ones."""

    def __init__(self, config, n=3):
        super().__init__()
        self.conv0 = nn.Conv2d(3, 16, 3, 1, 1, bias=False)
        self.bn0 = nn.BatchNorm2d(16)

        def stage(in_plane, first):
            blocks = [PreResidual(config, in_plane)] if first else [Residual(config, in_plane, True)]
            width = in_plane if first else in_plane * 2
            blocks += [Residual(config, width, False) for _ in range(n - 1)]
            return nn.Sequential(*blocks)

        self.layer1, self.layer2, self.layer3 = stage(16, True), stage(16, False), stage(32, False)
        self.bnlast = nn.BatchNorm2d(64)
        self.ap = nn.AvgPool2d(8)
        mode_q = getattr(config, "mode_q", "twn")
        self.linear = TernaryLinear(64, 10, mode_q if mode_q in ("twn", "ttq") else "twn") if mode_q != "inq" else nn.Linear(64, 10)

    def quant_layers(self):
        return [m for m in self.modules() if isinstance(m, _TernBase)]

    def _quantize_all(self):
        quantize_all(self.quant_layers())

    def inq_layers(self):
        return [m for m in self.modules() if isinstance(m, INQConv2d)]

    def forward(self, x):
        if not x.is_cuda:
            raise RuntimeError("the quantized ResNet runs on the HIP device only (input is on %s)" % x.device)
        self._quantize_all()
        inq_all(self.inq_layers())
        return self.forward_body(x)

    def forward_body(self, x):
        """The graph with the quantised / effective weights of all layers already handed down (forward, or a fused step)."""
        c0 = self.conv0
        # inside a step that defers the weight-gradient reductions the stem joins them (layers.QResNet.stem)
        slot = getattr(self, "_stem_slot", None) if layers._wgrad_deferred is not None else None
        out = layers.conv3x3(x, c0.weight, 1, slot, stats=self.training)
        out = layers.bn_relu(self.bn0, out)
        out = self.layer3(self.layer2(self.layer1(out)))
        out = layers.bn_relu_pool(self.bnlast, self.ap, out)
        return self.linear(out.view(out.size(0), -1))
