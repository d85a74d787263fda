"""Tensor-level wrappers over the HIP kernels (no autograd here; see `layers` for the Functions).

Every function launches on torch's current HIP stream, allocates its outputs and workspace through
torch's caching allocator (so calls are capturable in a HIP graph), and never synchronises.
"""
import ctypes as C

import torch

from . import _lib

# Default arithmetic of the synthesis kernels: False = exp on the transcendental unit (fastest);
# True = bit-reproducible operation sequence (equal to the CPU restatement bit for bit).
EXACT_ARITHMETIC = False


def _flags(exact):
    return _lib.FLAG_EXACT if (EXACT_ARITHMETIC if exact is None else exact) else 0


def _rows_L(p):
    L = p.shape[-1]
    return p.numel() // max(L, 1), L


def _seg_array(p_list, q_list):
    n = len(p_list)
    if n == 0 or n != len(q_list):
        raise ValueError("need one q_level per p_c (got %d / %d)" % (n, len(q_list)))
    segs = (_lib.AqSeg * n)()
    for i, (p, q) in enumerate(zip(p_list, q_list)):
        rows, L = _rows_L(p)
        if q.numel() != L:
            raise ValueError("segment %d: p_c has %d levels but q_level has %d" % (i, L, q.numel()))
        segs[i].p_c = _lib.ptr(p, name="p_c[%d]" % i)
        segs[i].q = _lib.ptr(q, name="q_level[%d]" % i)
        segs[i].rows, segs[i].L = rows, L
    return segs


def _workspace(segs, n, backward, device):
    nbytes = _lib.lib().aq_synth_workspace_bytes(segs, n, int(backward))
    ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=device)
    return ws, nbytes


def synth_forward(p_list, q_list, beta, soft=True, codes=False, sumsq=False, out_W=None, exact=None):
    """Weight synthesis of several layers in one launch.

    p_list[i]: (..., L_i) fp32, q_list[i]: (L_i,).  Returns a dict with (as requested)
      "W"      list of soft weights, shape p.shape[:-1]          (adp_qtz_fan.py:563-568)
      "idx"    list of uint8 codes, "W_hard" list of q[idx]      (adp_qtz_fan_test.py:304-305)
      "sumsq"  fp32 tensor [n]: sum of W^2 per layer             (adp_qtz_fan.py:752)
    out_W: optional list of preallocated fp32 tensors to receive W.
    exact: True = bit-reproducible arithmetic, False = fast, None = module default EXACT_ARITHMETIC
    (launches that emit codes are always exact).
    """
    n = len(p_list)
    segs = _seg_array(p_list, q_list)
    dev = p_list[0].device
    out = {}
    if soft:
        out["W"] = out_W if out_W is not None else [torch.empty(p.shape[:-1], dtype=torch.float32, device=dev) for p in p_list]
        for i, w in enumerate(out["W"]):
            if w.numel() != segs[i].rows:
                raise ValueError("out_W[%d] has %d elements, expected %d" % (i, w.numel(), segs[i].rows))
            segs[i].W = _lib.ptr(w, name="W[%d]" % i)
    if codes:
        out["idx"] = [torch.empty(p.shape[:-1], dtype=torch.uint8, device=dev) for p in p_list]
        out["W_hard"] = [torch.empty(p.shape[:-1], dtype=torch.float32, device=dev) for p in p_list]
        for i in range(n):
            segs[i].idx = _lib.ptr(out["idx"][i], torch.uint8, "idx")
            segs[i].W_hard = _lib.ptr(out["W_hard"][i], name="W_hard")
    if sumsq:
        out["sumsq"] = torch.empty(n, dtype=torch.float32, device=dev)
        base = out["sumsq"].data_ptr()
        for i in range(n):
            segs[i].sumsq = base + 4 * i
    ws, nbytes = _workspace(segs, n, False, dev)
    _lib.check(_lib.lib().aq_synth_fwd(segs, n, float(beta), _flags(exact), ws.data_ptr(), nbytes, _lib.stream()))
    return out


def synth_backward(p_list, q_list, beta, dW_list, wd=0.0, need_dq=None, out_dp=None, dsumsq=None, exact=None, out_dq=None):
    """Gradient of the synthesis for several layers in one launch.

    dW_list[i]: upstream d loss/d W (shape p.shape[:-1]) or None; wd = 2*l2_reg*(d loss/d reg).
    need_dq[i] False skips dq (frozen levels).  dsumsq: optional fp32 device tensor [n] with
    d loss/d sumsq per layer (adds 2*dsumsq[i] to wd on the device, no host read-back).
    Returns (dp_list, dq_list) with None where skipped.
    """
    n = len(p_list)
    segs = _seg_array(p_list, q_list)
    dev = p_list[0].device
    need_dq = [True] * n if need_dq is None else list(need_dq)
    dp = out_dp if out_dp is not None else [torch.empty_like(p) for p in p_list]
    if out_dq is not None:
        dq = [o if need else None for o, need in zip(out_dq, need_dq)]
    else:
        dq = [torch.empty_like(q) if need else None for q, need in zip(q_list, need_dq)]
    for i in range(n):
        if dW_list[i] is not None:
            if dW_list[i].numel() != segs[i].rows:
                raise ValueError("dW[%d] has %d elements, expected %d" % (i, dW_list[i].numel(), segs[i].rows))
            segs[i].dW = _lib.ptr(dW_list[i], name="dW[%d]" % i)
        segs[i].dp_c = _lib.ptr(dp[i], name="dp_c[%d]" % i)
        if dsumsq is not None:
            segs[i].dsumsq = _lib.ptr(dsumsq, name="dsumsq") + 4 * i
        if dq[i] is not None:
            segs[i].dq = _lib.ptr(dq[i], name="dq[%d]" % i)
    ws, nbytes = _workspace(segs, n, True, dev)
    _lib.check(_lib.lib().aq_synth_bwd(segs, n, float(beta), float(wd), _flags(exact), ws.data_ptr(), nbytes, _lib.stream()))
    return dp, dq


def synth_backward_adam(p_list, q_list, beta, dW_list, exp_avg, exp_avg_sq, steps, lr, betas=(0.9, 0.999), eps=1e-8, wd=0.0,
                        need_dq=None, dsumsq=None, exact=None, out_dq=None, out_dp=None):
    """synth_backward with torch.optim.Adam's update of p_c fused in (adp_qtz_fan.py:208-209, 291):
    p_list[i], exp_avg[i], exp_avg_sq[i] are updated in place and steps[i] (one-element fp32 device
    tensors, the optimizer's state["step"]) are advanced by one.  Returns dq_list; the gradient dp_c is
    only stored when out_dp is given."""
    n = len(p_list)
    if n > 24:
        raise ValueError("synth_backward_adam: at most 24 layers per call (got %d)" % n)
    segs = _seg_array(p_list, q_list)
    dev = p_list[0].device
    need_dq = [True] * n if need_dq is None else list(need_dq)
    if out_dq is not None:
        dq = [o if need else None for o, need in zip(out_dq, need_dq)]
    else:
        dq = [torch.empty_like(q) if need else None for q, need in zip(q_list, need_dq)]
    for i in range(n):
        if dW_list[i] is not None:
            if dW_list[i].numel() != segs[i].rows:
                raise ValueError("dW[%d] has %d elements, expected %d" % (i, dW_list[i].numel(), segs[i].rows))
            segs[i].dW = _lib.ptr(dW_list[i], name="dW[%d]" % i)
        for t, what in ((exp_avg[i], "exp_avg"), (exp_avg_sq[i], "exp_avg_sq")):
            if t.shape != p_list[i].shape:
                raise ValueError("%s[%d] has shape %s, expected %s" % (what, i, tuple(t.shape), tuple(p_list[i].shape)))
        segs[i].adam_m = _lib.ptr(exp_avg[i], name="exp_avg[%d]" % i)
        segs[i].adam_v = _lib.ptr(exp_avg_sq[i], name="exp_avg_sq[%d]" % i)
        if steps[i].numel() != 1:
            raise ValueError("steps[%d] must hold one element" % i)
        segs[i].adam_step = _lib.ptr(steps[i], name="step[%d]" % i)
        if out_dp is not None:
            segs[i].dp_c = _lib.ptr(out_dp[i], name="dp_c[%d]" % i)
        if dsumsq is not None:
            segs[i].dsumsq = _lib.ptr(dsumsq, name="dsumsq") + 4 * i
        if dq[i] is not None:
            segs[i].dq = _lib.ptr(dq[i], name="dq[%d]" % i)
    nbytes = _lib.lib().aq_synth_workspace_bytes(segs, n, 2)
    ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=dev)
    _lib.check(_lib.lib().aq_synth_bwd_adam(segs, n, float(beta), float(wd), float(lr), float(betas[0]), float(betas[1]), float(eps),
                                            _flags(exact), ws.data_ptr(), nbytes, _lib.stream()))
    return dq


# ------------------------------------------------------------------------------------ convolution

def _conv_dims(x_shape, Cout, stride, ks=3, pad=1):
    N, Cin, H, Wd = x_shape
    return [int(N), int(Cin), int(H), int(Wd), int(Cout), int(ks), int(stride), int(pad)]


def conv_out_shape(x_shape, Cout, stride, ks=3, pad=1):
    N, _, H, Wd = x_shape
    return (N, Cout, (H + 2 * pad - ks) // stride + 1, (Wd + 2 * pad - ks) // stride + 1)


def conv_forward(x, W, stride=1, res=None, stats=False, pad=1):
    """y = conv2d(x, W, stride, padding=pad) [+ res], W (Cout, Cin, k, k), k in 1, 3, 5, 7  (adp_qtz_fan.py:574-601;
    res: the shortcut added after conv2, :676-677).  3x3 / padding 1, the reference's layers, has the tuned kernels.
    stats=True returns (y, st): st = (buffer, descriptor) holds per-workgroup channel statistics of y written
    by the convolution's epilogue, which bn_relu_forward(y, ..., conv_stats=st) takes in place of its own
    statistics pass; st is None for shapes without that epilogue."""
    if W.dim() != 4 or W.shape[2] != W.shape[3] or x.dim() != 4 or W.shape[1] != x.shape[1]:
        raise ValueError("conv_forward: x %s / W %s: need NCHW input and (Cout, Cin, k, k) weights" % (tuple(x.shape), tuple(W.shape)))
    ks = int(W.shape[2])
    y = torch.empty(conv_out_shape(x.shape, W.shape[0], stride, ks, pad), dtype=torch.float32, device=x.device)
    dims = _conv_dims(x.shape, W.shape[0], stride, ks, pad)
    if res is not None and tuple(res.shape) != tuple(y.shape):
        raise ValueError("conv_forward: res %s does not match the output %s" % (tuple(res.shape), tuple(y.shape)))
    if stats:
        nbytes = _lib.lib().aq_conv_stats_bytes(*dims) if x.shape[0] > 0 else 0
        if nbytes > 0:
            buf = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
            desc = _lib.AqConvStats()
            _lib.check(_lib.lib().aq_conv_fwd_stats(_lib.ptr(x, name="x"), _lib.ptr(W, name="W"), _lib.ptr(res, name="res"), _lib.ptr(y), *dims,
                                                    buf.data_ptr(), nbytes, C.byref(desc), _lib.stream()))
            return y, (buf, desc)
    if res is None:
        _lib.check(_lib.lib().aq_conv_fwd(_lib.ptr(x, name="x"), _lib.ptr(W, name="W"), _lib.ptr(y), *dims, _lib.stream()))
    else:
        if tuple(res.shape) != tuple(y.shape):
            raise ValueError("conv_forward: res %s does not match the output %s" % (tuple(res.shape), tuple(y.shape)))
        _lib.check(_lib.lib().aq_conv_fwd_add(_lib.ptr(x, name="x"), _lib.ptr(W, name="W"), _lib.ptr(res, name="res"), _lib.ptr(y), *dims, _lib.stream()))
    return (y, None) if stats else y


def conv_forward_codes(x, idx, q, stride=1, pad=1):
    """Hard-mode conv: weights given as uint8 codes (Cout, Cin, k, k) + levels q (L,)."""
    if idx.dim() != 4 or idx.shape[1] != x.shape[1] or idx.shape[2] != idx.shape[3]:
        raise ValueError("conv_forward_codes: idx %s does not match x %s" % (tuple(idx.shape), tuple(x.shape)))
    ks = int(idx.shape[2])
    y = torch.empty(conv_out_shape(x.shape, idx.shape[0], stride, ks, pad), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().aq_conv_fwd_codes(_lib.ptr(x, name="x"), _lib.ptr(idx, torch.uint8, "idx"), _lib.ptr(q, name="q"), int(q.numel()),
                                            _lib.ptr(y), *_conv_dims(x.shape, idx.shape[0], stride, ks, pad), _lib.stream()))
    return y


def conv_dgrad(dy, W, x_shape, stride=1, bn=None, res=None, pad=1):
    """dx = d loss / d x of conv_forward.  bn = (xb, gamma, beta, save_mean, save_rstd) says that x was
    relu(batch_norm(xb)): the result is then (dx, sums), where sums = (buffer, descriptor) holds the
    per-workgroup reductions that bn_relu_backward(dx, xb, ..., dgrad_sums=sums) takes in place of its own
    reduction pass (None for shapes without that epilogue).  res: tensor of x's shape added to dx in the
    kernel's epilogue (the gradient arriving over an identity shortcut); not together with bn."""
    dx = torch.empty(tuple(x_shape), dtype=torch.float32, device=dy.device)
    ks = int(W.shape[2])
    if tuple(dy.shape) != conv_out_shape(x_shape, W.shape[0], stride, ks, pad):
        raise ValueError("conv_dgrad: dy %s does not match x %s" % (tuple(dy.shape), tuple(x_shape)))
    if res is not None:
        if bn is not None or tuple(res.shape) != tuple(x_shape):
            raise ValueError("conv_dgrad: res %s must have the shape of x %s and excludes bn" % (tuple(res.shape), tuple(x_shape)))
        _lib.check(_lib.lib().aq_conv_dgrad_add(_lib.ptr(dy, name="dy"), _lib.ptr(W, name="W"), _lib.ptr(res, name="res"), _lib.ptr(dx),
                                                *_conv_dims(x_shape, W.shape[0], stride, ks, pad), _lib.stream()))
        return dx
    if bn is not None:
        xb, gamma, beta, mean, rstd = bn
        if tuple(xb.shape) != tuple(x_shape) or not xb.is_contiguous():
            raise ValueError("conv_dgrad: bn input %s must be contiguous with the shape of x %s" % (tuple(xb.shape), tuple(x_shape)))
        dims = _conv_dims(x_shape, W.shape[0], stride, ks, pad)
        nbytes = _lib.lib().aq_conv_dgrad_bn_bytes(*dims) if x_shape[0] > 0 else 0
        if nbytes > 0:
            buf = torch.empty(nbytes, dtype=torch.uint8, device=dy.device)
            desc = _lib.AqConvStats()
            _lib.check(_lib.lib().aq_conv_dgrad_bn(
                _lib.ptr(dy, name="dy"), _lib.ptr(W, name="W"), _lib.ptr(dx), *dims, _lib.ptr(xb, name="xb"), _lib.ptr(gamma, name="weight"),
                _lib.ptr(beta, name="bias"), _lib.ptr(mean, name="save_mean"), _lib.ptr(rstd, name="save_rstd"), buf.data_ptr(), nbytes,
                C.byref(desc), _lib.stream()))
            return dx, (buf, desc)
        _lib.check(_lib.lib().aq_conv_dgrad(_lib.ptr(dy, name="dy"), _lib.ptr(W, name="W"), _lib.ptr(dx), *dims, _lib.stream()))
        return dx, None
    _lib.check(_lib.lib().aq_conv_dgrad(_lib.ptr(dy, name="dy"), _lib.ptr(W, name="W"), _lib.ptr(dx), *_conv_dims(x_shape, W.shape[0], stride, ks, pad),
                                        _lib.stream()))
    return dx


def conv_backward(dy, W, x, stride=1, bn=None, pad=1):
    """Both gradients of y = conv_forward(x, W, stride) from one call (one launch where the library has a
    kernel pair for the shape): -> (dx, sums, workspace, nslab).  dx / sums as conv_dgrad(dy, W, x.shape,
    stride, bn) (sums None without bn or without a reduction epilogue for the shape); workspace / nslab as
    conv_wgrad_partial(x, dy, stride), to be finished by conv_wgrad_reduce."""
    Cout, ks = W.shape[0], int(W.shape[2])
    if tuple(dy.shape) != conv_out_shape(x.shape, Cout, stride, ks, pad) or W.shape[1] != x.shape[1]:
        raise ValueError("conv_backward: dy %s / W %s do not match x %s" % (tuple(dy.shape), tuple(W.shape), tuple(x.shape)))
    dims = _conv_dims(x.shape, Cout, stride, ks, pad)
    dx = torch.empty_like(x)
    nbytes = _lib.lib().aq_conv_wgrad_workspace_bytes(*dims)
    ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=x.device)
    nslab = C.c_int(0)
    desc = _lib.AqConvStats()
    buf, sbytes, extra = None, 0, [None] * 5
    if bn is not None:
        xb, gamma, beta, mean, rstd = bn
        if tuple(xb.shape) != tuple(x.shape) or not xb.is_contiguous():
            raise ValueError("conv_backward: bn input %s must be contiguous with the shape of x %s" % (tuple(xb.shape), tuple(x.shape)))
        sbytes = _lib.lib().aq_conv_bwd_sums_bytes(*dims)
        buf = torch.empty(max(sbytes, 16), dtype=torch.uint8, device=x.device)
        extra = [_lib.ptr(xb, name="xb"), _lib.ptr(gamma, name="weight"), _lib.ptr(beta, name="bias"), _lib.ptr(mean, name="save_mean"),
                 _lib.ptr(rstd, name="save_rstd")]
    _lib.check(_lib.lib().aq_conv_bwd(
        _lib.ptr(dy, name="dy"), _lib.ptr(W, name="W"), _lib.ptr(x, name="x"), _lib.ptr(dx), *dims, *extra,
        buf.data_ptr() if buf is not None else None, sbytes, C.byref(desc), ws.data_ptr(), nbytes, C.byref(nslab), _lib.stream()))
    sums = (buf, desc) if (bn is not None and desc.P > 0) else None
    return dx, sums, ws, nslab.value


def conv_wgrad(x, dy, stride=1, out=None, ks=3, pad=1):
    Cout = dy.shape[1]
    if tuple(dy.shape) != conv_out_shape(x.shape, Cout, stride, ks, pad):
        raise ValueError("conv_wgrad: dy %s does not match x %s" % (tuple(dy.shape), tuple(x.shape)))
    dW = out if out is not None else torch.empty((Cout, x.shape[1], ks, ks), dtype=torch.float32, device=x.device)
    if dW.numel() != Cout * x.shape[1] * ks * ks:
        raise ValueError("conv_wgrad: out has %d elements, expected %d" % (dW.numel(), Cout * x.shape[1] * ks * ks))
    dims = _conv_dims(x.shape, Cout, stride, ks, pad)
    nbytes = _lib.lib().aq_conv_wgrad_workspace_bytes(*dims)
    ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=x.device)
    _lib.check(_lib.lib().aq_conv_wgrad(_lib.ptr(x, name="x"), _lib.ptr(dy, name="dy"), _lib.ptr(dW, name="dW"), *dims, ws.data_ptr(), nbytes, _lib.stream()))
    return dW


def conv_wgrad_partial(x, dy, stride=1, ks=3, pad=1):
    """First stage of conv_wgrad only: -> (workspace tensor holding the slabs, number of slabs).  The
    caller reduces them later with conv_wgrad_reduce (several layers in one launch)."""
    Cout = dy.shape[1]
    if tuple(dy.shape) != conv_out_shape(x.shape, Cout, stride, ks, pad):
        raise ValueError("conv_wgrad_partial: dy %s does not match x %s" % (tuple(dy.shape), tuple(x.shape)))
    dims = _conv_dims(x.shape, Cout, stride, ks, pad)
    nbytes = _lib.lib().aq_conv_wgrad_workspace_bytes(*dims)
    ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=x.device)
    nslab = C.c_int(0)
    _lib.check(_lib.lib().aq_conv_wgrad_partial(_lib.ptr(x, name="x"), _lib.ptr(dy, name="dy"), *dims, ws.data_ptr(), nbytes, C.byref(nslab),
                                                _lib.stream()))
    return ws, nslab.value


def conv_wgrad_partial_group(xs, dys, stride=1, ks=3, pad=1):
    """conv_wgrad_partial for several layers of ONE shape in one launch per 8 layers -> list of (workspace, nslab)."""
    if not xs:
        return []
    x0, Cout = xs[0], dys[0].shape[1]
    for x, dy in zip(xs, dys):
        if tuple(x.shape) != tuple(x0.shape) or tuple(dy.shape) != conv_out_shape(x0.shape, Cout, stride, ks, pad):
            raise ValueError("conv_wgrad_partial_group: layers of different shapes (%s / %s against %s)" % (tuple(x.shape), tuple(dy.shape), tuple(x0.shape)))
    dims = _conv_dims(x0.shape, Cout, stride, ks, pad)
    nbytes = _lib.lib().aq_conv_wgrad_workspace_bytes(*dims)
    n = len(xs)
    wss = [torch.empty(max(nbytes, 16), dtype=torch.uint8, device=x0.device) for _ in range(n)]
    ax = (C.c_void_p * n)(*[_lib.ptr(x, name="x") for x in xs])
    ady = (C.c_void_p * n)(*[_lib.ptr(dy, name="dy") for dy in dys])
    aws = (C.c_void_p * n)(*[w.data_ptr() for w in wss])
    nslab = C.c_int(0)
    _lib.check(_lib.lib().aq_conv_wgrad_partial_group(ax, ady, aws, n, *dims, nbytes, C.byref(nslab), _lib.stream()))
    return [(w, nslab.value) for w in wss]


def conv_wgrad_reduce(jobs):
    """jobs: list of (workspace, dW tensor, nslab) from conv_wgrad_partial; one launch per 32 jobs."""
    if not jobs:
        return
    arr = (_lib.AqWgradJob * len(jobs))()
    for j, (ws, dW, nslab) in enumerate(job[:3] for job in jobs):
        arr[j].part, arr[j].dW = ws.data_ptr(), _lib.ptr(dW, name="dW")
        arr[j].n, arr[j].nslab = dW.numel(), nslab
    _lib.check(_lib.lib().aq_conv_wgrad_reduce(arr, len(jobs), _lib.stream()))


# ----------------------------------------------------------------------------------------- linear

def linear_forward(x, W, bias):
    """y = x @ W + bias with W (in, out)  (adp_qtz_fan.py:646-647)."""
    N, In = x.shape
    if W.shape[0] != In:
        raise ValueError("linear_forward: x %s / W %s" % (tuple(x.shape), tuple(W.shape)))
    Out = W.shape[1]
    y = torch.empty((N, Out), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().aq_linear_fwd(_lib.ptr(x, name="x"), _lib.ptr(W, name="W"), _lib.ptr(bias, name="bias"), _lib.ptr(y), N, In, Out, _lib.stream()))
    return y


def linear_backward(x, W, dy, need_dx=True, out_dW=None):
    N, In = x.shape
    Out = W.shape[1]
    dx = torch.empty_like(x) if need_dx else None
    if out_dW is not None and (tuple(out_dW.shape) != tuple(W.shape) or not out_dW.is_contiguous()):
        raise ValueError("linear_backward: out_dW must be contiguous with the shape of W %s" % (tuple(W.shape),))
    dW = out_dW if out_dW is not None else torch.empty_like(W)
    db = torch.empty(Out, dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().aq_linear_bwd(_lib.ptr(x, name="x"), _lib.ptr(W, name="W"), _lib.ptr(dy, name="dy"), _lib.ptr(dx), _lib.ptr(dW), _lib.ptr(db),
                                        N, In, Out, _lib.stream()))
    return dx, dW, db


# ------------------------------------------------------------------------- batch norm (+ ReLU)

def _bn_dims(x):
    if x.dim() < 2:
        raise ValueError("batch norm input needs at least 2 dimensions, got %s" % (tuple(x.shape),))
    N, Cn = int(x.shape[0]), int(x.shape[1])
    return N, Cn, int(x.numel() // max(N * Cn, 1))


def bn_relu_forward(x, gamma, beta, running_mean=None, running_var=None, num_batches_tracked=None, eps=1e-5, momentum=0.1,
                    relu=True, use_running_stats=False, pool=False, conv_stats=None, shortcut=None):
    """y = [relu](batch_norm(x)) over (N, H, W) of an NCHW tensor (adp_qtz_fan.py:674-675, 690, 726, 730).
    Batch-statistics mode returns (y, save_mean, save_rstd) and updates the running statistics in
    place like nn.BatchNorm2d in training mode; use_running_stats=True normalises with them instead.
    pool=True returns the mean of y over each (image, channel) plane, shape (N, C), instead of y (the
    network's tail, adp_qtz_fan.py:730-731).
    conv_stats: the second result of conv_forward(..., stats=True) for this very x (batch-statistics mode).
    shortcut=(pad_front, pad_back): the same launch also produces shortcut_forward(x, pad_front, pad_back); the
    result is then (y, save_mean, save_rstd, s)."""
    N, Cn, HW = _bn_dims(x)
    y = torch.empty((N, Cn), dtype=torch.float32, device=x.device) if pool else torch.empty_like(x)
    flags = (_lib.BN_RELU if relu else 0) | (_lib.BN_EVAL if use_running_stats else 0)
    mean = rstd = ws = None
    nbytes = 0
    if not use_running_stats:
        mean = torch.empty(Cn, dtype=torch.float32, device=x.device)
        rstd = torch.empty(Cn, dtype=torch.float32, device=x.device)
        if conv_stats is not None:
            ws, desc = conv_stats
            nbytes = ws.numel()
        else:
            nbytes = _lib.lib().aq_bn_workspace_bytes(N, Cn, HW)
            ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=x.device)
    desc = C.byref(conv_stats[1]) if (conv_stats is not None and not use_running_stats) else None
    if shortcut is not None:
        front, back = shortcut
        if pool or x.dim() != 4:
            raise ValueError("bn_relu_forward: the shortcut goes with the plain form on an NCHW tensor")
        s = torch.empty((N, Cn + front + back, x.shape[2] // 2, x.shape[3] // 2), dtype=torch.float32, device=x.device)
        _lib.check(_lib.lib().aq_bn_relu_fwd_shortcut(
            _lib.ptr(x, name="x"), _lib.ptr(gamma, name="weight"), _lib.ptr(beta, name="bias"), _lib.ptr(running_mean, name="running_mean"),
            _lib.ptr(running_var, name="running_var"), _lib.ptr(num_batches_tracked, torch.int64, "num_batches_tracked"), N, Cn,
            int(x.shape[2]), int(x.shape[3]), float(eps), float(momentum), flags, _lib.ptr(y), _lib.ptr(mean), _lib.ptr(rstd),
            ws.data_ptr() if ws is not None else None, nbytes, desc, int(front), int(back), _lib.ptr(s), _lib.stream()))
        return y, mean, rstd, s
    fn = _lib.lib().aq_bn_relu_pool_fwd if pool else _lib.lib().aq_bn_relu_fwd
    _lib.check(fn(
        _lib.ptr(x, name="x"), _lib.ptr(gamma, name="weight"), _lib.ptr(beta, name="bias"), _lib.ptr(running_mean, name="running_mean"),
        _lib.ptr(running_var, name="running_var"), _lib.ptr(num_batches_tracked, torch.int64, "num_batches_tracked"), N, Cn, HW,
        float(eps), float(momentum), flags, _lib.ptr(y), _lib.ptr(mean), _lib.ptr(rstd), ws.data_ptr() if ws is not None else None, nbytes,
        desc, _lib.stream()))
    return y, mean, rstd


def bn_relu_backward(dy, x, gamma, beta, save_mean, save_rstd, res=None, relu=True, need_affine_grads=True, pool=False, dgrad_sums=None,
                     shortcut=None):
    """-> (dx [+ res], dgamma, dbeta) of bn_relu_forward in batch-statistics mode.  pool=True: dy is the
    gradient of the pooled output, shape (N, C)."""
    N, Cn, HW = _bn_dims(x)
    if pool:
        if tuple(dy.shape) != (N, Cn) or res is not None:
            raise ValueError("bn_relu_backward(pool): dy must be (N, C) = (%d, %d) and res None" % (N, Cn))
    elif dy.shape != x.shape or (res is not None and res.shape != x.shape):
        raise ValueError("bn_relu_backward: dy / res must have the shape of x %s" % (tuple(x.shape),))
    if shortcut is not None:
        # shortcut = (dshort, pad_front, pad_back): gradient of ZeroPad(AvgPool2d(2)(x)) instead of a res tensor
        dshort, front, back = shortcut
        if pool or res is not None or x.dim() != 4 or tuple(dshort.shape) != (N, Cn + front + back, x.shape[2] // 2, x.shape[3] // 2):
            raise ValueError("bn_relu_backward: shortcut gradient %s does not belong to x %s with pads %d, %d" % (
                tuple(dshort.shape), tuple(x.shape), front, back))
    dx = torch.empty_like(x)
    dg = torch.empty(Cn, dtype=torch.float32, device=x.device) if need_affine_grads else None
    db = torch.empty(Cn, dtype=torch.float32, device=x.device) if need_affine_grads else None
    if dgrad_sums is not None:
        if pool or not relu:
            raise ValueError("bn_relu_backward: dgrad_sums go with the plain ReLU form only")
        ws, nbytes, desc = dgrad_sums[0], dgrad_sums[0].numel(), C.byref(dgrad_sums[1])
    else:
        nbytes, desc = _lib.lib().aq_bn_workspace_bytes(N, Cn, HW), None
        ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=x.device)
    if shortcut is not None:
        _lib.check(_lib.lib().aq_bn_relu_bwd_shortcut(
            _lib.ptr(dy, name="dy"), _lib.ptr(x, name="x"), _lib.ptr(gamma, name="weight"), _lib.ptr(beta, name="bias"),
            _lib.ptr(save_mean, name="save_mean"), _lib.ptr(save_rstd, name="save_rstd"), _lib.ptr(dshort, name="dshort"), N, Cn,
            int(x.shape[2]), int(x.shape[3]), int(front), int(back), _lib.BN_RELU if relu else 0, _lib.ptr(dx), _lib.ptr(dg), _lib.ptr(db),
            ws.data_ptr(), nbytes, desc, _lib.stream()))
        return dx, dg, db
    if pool:
        _lib.check(_lib.lib().aq_bn_relu_pool_bwd(
            _lib.ptr(dy, name="dy"), _lib.ptr(x, name="x"), _lib.ptr(gamma, name="weight"), _lib.ptr(beta, name="bias"),
            _lib.ptr(save_mean, name="save_mean"), _lib.ptr(save_rstd, name="save_rstd"), N, Cn, HW,
            _lib.BN_RELU if relu else 0, _lib.ptr(dx), _lib.ptr(dg), _lib.ptr(db), ws.data_ptr(), nbytes, _lib.stream()))
        return dx, dg, db
    _lib.check(_lib.lib().aq_bn_relu_bwd(
        _lib.ptr(dy, name="dy"), _lib.ptr(x, name="x"), _lib.ptr(gamma, name="weight"), _lib.ptr(beta, name="bias"),
        _lib.ptr(save_mean, name="save_mean"), _lib.ptr(save_rstd, name="save_rstd"), _lib.ptr(res, name="res"), N, Cn, HW,
        _lib.BN_RELU if relu else 0, _lib.ptr(dx), _lib.ptr(dg), _lib.ptr(db), ws.data_ptr(), nbytes, desc, _lib.stream()))
    return dx, dg, db


def bn_add_relu_forward(x, add, gamma, beta, running_mean, running_var, num_batches_tracked, eps=1e-5, momentum=0.1, conv_stats=None):
    """y = relu(batch_norm(x) + add) with batch statistics (the tail of a post-activation residual block) in one launch
    (plus the statistics pass unless conv_stats carries them) -> (y, save_mean, save_rstd); running statistics updated."""
    N, Cn, HW = _bn_dims(x)
    if tuple(add.shape) != tuple(x.shape):
        raise ValueError("bn_add_relu_forward: add %s must have the shape of x %s" % (tuple(add.shape), tuple(x.shape)))
    y = torch.empty_like(x)
    mean = torch.empty(Cn, dtype=torch.float32, device=x.device)
    rstd = torch.empty(Cn, dtype=torch.float32, device=x.device)
    if conv_stats is not None:
        ws, desc = conv_stats[0], C.byref(conv_stats[1])
        nbytes = ws.numel()
    else:
        nbytes, desc = _lib.lib().aq_bn_workspace_bytes(N, Cn, HW), None
        ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=x.device)
    _lib.check(_lib.lib().aq_bn_add_relu_fwd(
        _lib.ptr(x, name="x"), _lib.ptr(gamma, name="weight"), _lib.ptr(beta, name="bias"), _lib.ptr(running_mean, name="running_mean"),
        _lib.ptr(running_var, name="running_var"), _lib.ptr(num_batches_tracked, torch.int64, "num_batches_tracked"), N, Cn, HW,
        float(eps), float(momentum), _lib.ptr(add, name="add"), _lib.ptr(y), _lib.ptr(mean), _lib.ptr(rstd), ws.data_ptr(), nbytes, desc,
        _lib.stream()))
    return y, mean, rstd


def bn_add_relu_backward(dy, x, y, gamma, beta, save_mean, save_rstd):
    """-> (dx, dadd, dgamma, dbeta) of bn_add_relu_forward; y is the forward's output (its sign is the ReLU's mask)."""
    N, Cn, HW = _bn_dims(x)
    if dy.shape != x.shape or y.shape != x.shape:
        raise ValueError("bn_add_relu_backward: dy / y must have the shape of x %s" % (tuple(x.shape),))
    dx, dadd = torch.empty_like(x), torch.empty_like(x)
    dg = torch.empty(Cn, dtype=torch.float32, device=x.device)
    db = torch.empty(Cn, dtype=torch.float32, device=x.device)
    nbytes = _lib.lib().aq_bn_workspace_bytes(N, Cn, HW)
    ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=x.device)
    _lib.check(_lib.lib().aq_bn_add_relu_bwd(
        _lib.ptr(dy, name="dy"), _lib.ptr(x, name="x"), _lib.ptr(y, name="y"), _lib.ptr(gamma, name="weight"), _lib.ptr(beta, name="bias"),
        _lib.ptr(save_mean, name="save_mean"), _lib.ptr(save_rstd, name="save_rstd"), N, Cn, HW, _lib.ptr(dx), _lib.ptr(dadd), _lib.ptr(dg),
        _lib.ptr(db), ws.data_ptr(), nbytes, _lib.stream()))
    return dx, dadd, dg, db


# -------------------------------------------------------------------------------------- max pool

def maxpool3s2_forward(x):
    """nn.MaxPool2d(3, 2, 1) -> (y, idx): idx uint8 = position ky * 3 + kx of each window's first maximum."""
    N, Cn, H, Wd = x.shape
    OH, OW = (H - 1) // 2 + 1, (Wd - 1) // 2 + 1
    y = torch.empty((N, Cn, OH, OW), dtype=torch.float32, device=x.device)
    idx = torch.empty((N, Cn, OH, OW), dtype=torch.uint8, device=x.device)
    _lib.check(_lib.lib().aq_maxpool3s2_fwd(_lib.ptr(x, name="x"), _lib.ptr(y), _lib.ptr(idx, torch.uint8, "idx"), int(N), int(Cn), int(H), int(Wd),
                                            _lib.stream()))
    return y, idx


def maxpool3s2_backward(dy, idx, x_shape):
    N, Cn, H, Wd = x_shape
    if tuple(dy.shape) != (N, Cn, (H - 1) // 2 + 1, (Wd - 1) // 2 + 1) or tuple(idx.shape) != tuple(dy.shape):
        raise ValueError("maxpool3s2_backward: dy %s / idx %s do not match x %s" % (tuple(dy.shape), tuple(idx.shape), tuple(x_shape)))
    dx = torch.empty(tuple(x_shape), dtype=torch.float32, device=dy.device)
    _lib.check(_lib.lib().aq_maxpool3s2_bwd(_lib.ptr(dy, name="dy"), _lib.ptr(idx, torch.uint8, "idx"), _lib.ptr(dx), int(N), int(Cn), int(H), int(Wd),
                                            _lib.stream()))
    return dx


# -------------------------------------------------------------------------------------- shortcut

def shortcut_forward(x, pad_front, pad_back):
    """AvgPool2d(2) + zero channels in front / behind (adp_qtz_fan.py:669-671)."""
    N, Cn, H, Wd = x.shape
    y = torch.empty((N, Cn + pad_front + pad_back, H // 2, Wd // 2), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().aq_shortcut_fwd(_lib.ptr(x, name="x"), _lib.ptr(y), int(N), int(Cn), int(H), int(Wd), int(pad_front), int(pad_back),
                                          _lib.stream()))
    return y


def shortcut_backward(dy, x_shape, pad_front, pad_back):
    N, Cn, H, Wd = x_shape
    if tuple(dy.shape) != (N, Cn + pad_front + pad_back, H // 2, Wd // 2):
        raise ValueError("shortcut_backward: dy %s does not match x %s" % (tuple(dy.shape), tuple(x_shape)))
    dx = torch.empty(tuple(x_shape), dtype=torch.float32, device=dy.device)
    _lib.check(_lib.lib().aq_shortcut_bwd(_lib.ptr(dy, name="dy"), _lib.ptr(dx), int(N), int(Cn), int(H), int(Wd), int(pad_front), int(pad_back),
                                          _lib.stream()))
    return dx


# ---------------------------------------------------------------------------------- packed codes

def code_bits(L):
    """Bits per code for L levels: ceil(log2 L), at least 1."""
    return max(1, (int(L) - 1).bit_length())


def pack_codes(idx, bits):
    """uint8 codes (any shape) -> uint8 bit stream of ceil(n*bits/8) bytes (little-endian, code i in
    bits [i*bits, (i+1)*bits))."""
    n = idx.numel()
    out = torch.empty(_lib.lib().aq_packed_bytes(n, int(bits)), dtype=torch.uint8, device=idx.device)
    _lib.check(_lib.lib().aq_pack_codes(_lib.ptr(idx, torch.uint8, "idx"), n, int(bits), _lib.ptr(out, torch.uint8), _lib.stream()))
    return out


def unpack_codes(packed, n, bits):
    if packed.numel() != _lib.lib().aq_packed_bytes(int(n), int(bits)):
        raise ValueError("unpack_codes: %d bytes given, %d codes of %d bits need %d" % (
            packed.numel(), n, bits, _lib.lib().aq_packed_bytes(int(n), int(bits))))
    idx = torch.empty(int(n), dtype=torch.uint8, device=packed.device)
    _lib.check(_lib.lib().aq_unpack_codes(_lib.ptr(packed, torch.uint8, "packed"), int(n), int(bits), _lib.ptr(idx, torch.uint8), _lib.stream()))
    return idx


# --------------------------------------------------------------------- input pipeline, scales

def augment_batch(data, index=None, off_y=None, off_x=None, flip=None, pad=4, mean=(0.4915, 0.4821, 0.4462),
                  std=(0.2472, 0.2437, 0.2617), batch=None):
    """Normalised NCHW fp32 batch from the raw uint8 image array (N, H, W, C): gather `index`, zero-pad by
    `pad`, crop at (off_y, off_x) in 0..2*pad, mirror where `flip`, then ToTensor + Normalize
    (adp_qtz_fan.py:164-174).  index / off_y / off_x: int32 device tensors [B]; flip: uint8 [B]; any of
    them None = identity.  Defaults are the reference's constants."""
    if data.dim() != 4 or data.dtype != torch.uint8:
        raise ValueError("augment_batch: data must be uint8 (N, H, W, C)")
    N, H, W, Cn = data.shape
    B = int(batch if batch is not None else (index.numel() if index is not None else N))
    for t, name in ((index, "index"), (off_y, "off_y"), (off_x, "off_x"), (flip, "flip")):
        if t is not None and t.numel() != B:
            raise ValueError("augment_batch: %s has %d entries for a batch of %d" % (name, t.numel(), B))
    out = torch.empty((B, Cn, H, W), dtype=torch.float32, device=data.device)
    m = (C.c_float * Cn)(*[float(v) for v in mean][:Cn])
    sd = (C.c_float * Cn)(*[float(v) for v in std][:Cn])
    _lib.check(_lib.lib().aq_augment_batch(
        _lib.ptr(data, torch.uint8, "data"), int(N), _lib.ptr(index, torch.int32, "index"), _lib.ptr(off_y, torch.int32, "off_y"),
        _lib.ptr(off_x, torch.int32, "off_x"), _lib.ptr(flip, torch.uint8, "flip"), B, int(H), int(W), int(Cn), int(pad), m, sd,
        _lib.ptr(out), _lib.stream()))
    return out


def absmax(tensors):
    """max |v| of every tensor in the list, one launch per 32 tensors -> fp32 device tensor [n]."""
    n = len(tensors)
    out = torch.empty(n, dtype=torch.float32, device=tensors[0].device)
    ptrs = (C.c_void_p * n)(*[_lib.ptr(t, name="tensor[%d]" % i) for i, t in enumerate(tensors)])
    sizes = (C.c_longlong * n)(*[t.numel() for t in tensors])
    _lib.check(_lib.lib().aq_absmax(ptrs, sizes, n, _lib.ptr(out), _lib.stream()))
    return out


# ------------------------------------------------- convolution that normalises its input while staging it

_bnconv_ok = {}


def bnconv_supported(x_shape, Cout, stride):
    """True when conv(relu(batch_norm(x))) can run without materialising the activated tensor."""
    key = (tuple(int(v) for v in x_shape), int(Cout), int(stride))
    if key not in _bnconv_ok:
        _bnconv_ok[key] = bool(_lib.lib().aq_bnconv_supported(*_conv_dims(x_shape, Cout, stride)))
    return _bnconv_ok[key]


def bnconv_forward(x, gamma, beta, running_mean, running_var, num_batches_tracked, eps, momentum, W, stride=1, res=None):
    """y = conv3x3(relu(batch_norm(x)), W) [+ res] with batch statistics (adp_qtz_fan.py:674-677, 690-692);
    returns (y, save_mean, save_rstd) and updates the running statistics like nn.BatchNorm2d."""
    N, Cin, H, Wd = x.shape
    y = torch.empty(conv_out_shape(x.shape, W.shape[0], stride), dtype=torch.float32, device=x.device)
    if res is not None and tuple(res.shape) != tuple(y.shape):
        raise ValueError("bnconv_forward: res %s does not match the output %s" % (tuple(res.shape), tuple(y.shape)))
    mean = torch.empty(Cin, dtype=torch.float32, device=x.device)
    rstd = torch.empty(Cin, dtype=torch.float32, device=x.device)
    nbytes = _lib.lib().aq_bn_workspace_bytes(int(N), int(Cin), int(H * Wd))
    ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=x.device)
    _lib.check(_lib.lib().aq_bnconv_fwd(
        _lib.ptr(x, name="x"), _lib.ptr(gamma, name="weight"), _lib.ptr(beta, name="bias"), _lib.ptr(running_mean, name="running_mean"),
        _lib.ptr(running_var, name="running_var"), _lib.ptr(num_batches_tracked, torch.int64, "num_batches_tracked"), float(eps),
        float(momentum), _lib.ptr(W, name="W"), _lib.ptr(res, name="res"), _lib.ptr(y), _lib.ptr(mean), _lib.ptr(rstd),
        *_conv_dims(x.shape, W.shape[0], stride), ws.data_ptr(), nbytes, _lib.stream()))
    return y, mean, rstd


def bnconv_wgrad_partial(x, gamma, beta, save_mean, save_rstd, dy, stride=1):
    """conv_wgrad_partial with relu(batch_norm(x)) recomputed from the saved statistics as the operand."""
    Cout = dy.shape[1]
    if tuple(dy.shape) != conv_out_shape(x.shape, Cout, stride):
        raise ValueError("bnconv_wgrad_partial: dy %s does not match x %s" % (tuple(dy.shape), tuple(x.shape)))
    dims = _conv_dims(x.shape, Cout, stride)
    nbytes = _lib.lib().aq_conv_wgrad_workspace_bytes(*dims)
    ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=x.device)
    nslab = C.c_int(0)
    _lib.check(_lib.lib().aq_bnconv_wgrad_partial(
        _lib.ptr(x, name="x"), _lib.ptr(gamma, name="weight"), _lib.ptr(beta, name="bias"), _lib.ptr(save_mean, name="save_mean"),
        _lib.ptr(save_rstd, name="save_rstd"), _lib.ptr(dy, name="dy"), *dims, ws.data_ptr(), nbytes, C.byref(nslab), _lib.stream()))
    return ws, nslab.value


# ------------------------------------------------------------------------------------- loss tail

def sumsq(tensors):
    """Sum of squares of every tensor in the list (one launch per 32 tensors) -> fp32 device tensor [n]:
    the `torch.sum(param**2)` terms of model_loss (adp_qtz_fan.py:741-743)."""
    n = len(tensors)
    out = torch.empty(n, dtype=torch.float32, device=tensors[0].device)
    ptrs = (C.c_void_p * n)(*[_lib.ptr(t, name="tensor[%d]" % i) for i, t in enumerate(tensors)])
    sizes = (C.c_longlong * n)(*[t.numel() for t in tensors])
    nbytes = _lib.lib().aq_sumsq_workspace_bytes(sizes, n)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=tensors[0].device) if nbytes else None
    _lib.check(_lib.lib().aq_sumsq(ptrs, sizes, n, _lib.ptr(out), ws.data_ptr() if ws is not None else None, nbytes, _lib.stream()))
    return out


def ce_loss(logits, labels, l2_reg=0.0, sumsq_a=None, sumsq_b=None, need_grad=True, grad_scale=1.0):
    """CrossEntropyLoss (mean) + regulariser value in one launch (adp_qtz_fan.py:288-289):
    -> (vals, dlogits) with vals = [ce, l2_reg*(sum sumsq_a + sum sumsq_b), their sum] on the device and
    dlogits = grad_scale * d ce / d logits (None unless need_grad)."""
    if logits.dim() != 2 or labels.shape != logits.shape[:1]:
        raise ValueError("ce_loss: logits %s / labels %s" % (tuple(logits.shape), tuple(labels.shape)))
    N, Cn = logits.shape
    vals = torch.empty(3, dtype=torch.float32, device=logits.device)
    d = torch.empty_like(logits) if need_grad else None
    na = 0 if sumsq_a is None else sumsq_a.numel()
    nb = 0 if sumsq_b is None else sumsq_b.numel()
    nbytes = _lib.lib().aq_ce_loss_workspace_bytes(int(N), int(Cn))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=logits.device) if nbytes else None
    _lib.check(_lib.lib().aq_ce_loss(_lib.ptr(logits, name="logits"), _lib.ptr(labels, torch.int64, "labels"), int(N), int(Cn),
                                     _lib.ptr(sumsq_a, name="sumsq_a"), na, _lib.ptr(sumsq_b, name="sumsq_b"), nb, float(l2_reg),
                                     float(grad_scale), _lib.ptr(d), _lib.ptr(vals), ws.data_ptr() if ws is not None else None, nbytes, _lib.stream()))
    return vals, d


# ------------------------------------------------------------------------------------------ Adam

def adam_step(params, grads, exp_avgs, exp_avg_sqs, steps, lr, betas=(0.9, 0.999), eps=1e-8, regs=None):
    """torch.optim.Adam's update (no weight decay / amsgrad) of a list of tensors in one launch per 48 tensors
    (adp_qtz_fan.py:296-304: optimizer_q.step(), optimizer.step()).  steps: one-element fp32 tensors, one per
    parameter, advanced by 1.  regs[i] != 0: grads[i] += regs[i] * params[i] first (in place)."""
    n = len(params)
    if not (len(grads) == len(exp_avgs) == len(exp_avg_sqs) == len(steps) == n) or (regs is not None and len(regs) != n):
        raise ValueError("adam_step: lists of different lengths")
    if n == 0:
        return
    arr = (_lib.AqAdamItem * n)()
    for i in range(n):
        p = params[i]
        for t, name in ((grads[i], "grad"), (exp_avgs[i], "exp_avg"), (exp_avg_sqs[i], "exp_avg_sq")):
            if t.numel() != p.numel():
                raise ValueError("adam_step: %s of tensor %d has %d elements, the parameter %d" % (name, i, t.numel(), p.numel()))
        arr[i].p, arr[i].g = _lib.ptr(p, name="param"), _lib.ptr(grads[i], name="grad")
        arr[i].m, arr[i].v = _lib.ptr(exp_avgs[i], name="exp_avg"), _lib.ptr(exp_avg_sqs[i], name="exp_avg_sq")
        arr[i].step = _lib.ptr(steps[i], name="step")
        arr[i].n = p.numel()
        arr[i].reg = float(regs[i]) if regs is not None else 0.0
    _lib.check(_lib.lib().aq_adam_step(arr, n, float(lr), float(betas[0]), float(betas[1]), float(eps), _lib.stream()))


# ------------------------------------------------------ ternary / power-of-two quantisers (TWN, TTQ, INQ)

_TERN_MODES = {"twn": _lib.TERN_TWN, "ttq": _lib.TERN_TTQ, "inq": _lib.TERN_INQ}


def tern_forward(ws, mode="twn", factor=0.7, threshold_from_max=False, Wp=None, Wn=None, codes=True, packed=False, decoded=True, out_Wt=None):
    """Ternary quantisation of a list of weight tensors in one call (five small launches for all of them):
    delta = factor * mean|w| (threshold_from_max: factor * max|w|); code 1 where w > delta, 2 where w < -delta, else 0.
    mode "twn": both scales = mean of |w| above delta; "ttq": scales are the learned Wp[i], Wn[i] (one-element tensors).
    Returns one dict per tensor: "codes" uint8 (shape of w), "packed" uint8 bytes (2 bits per code), "Wt" = q[codes],
    "stats" fp32 [4] = (delta, positive scale, negative scale, number of non-zero codes), "q" fp32 [3] = (0, +pos, -neg).
    out_Wt: optional list of preallocated fp32 tensors that receive the decoded weights (e.g. views of one flat buffer)."""
    n = len(ws)
    if mode not in ("twn", "ttq"):
        raise ValueError("tern_forward: mode %r (twn or ttq)" % (mode,))
    if mode == "ttq" and (Wp is None or Wn is None or len(Wp) != n or len(Wn) != n):
        raise ValueError("tern_forward: mode ttq needs one Wp and one Wn per tensor")
    if mode == "twn" and decoded and not codes:
        raise ValueError("tern_forward: TWN decodes Wt from the codes; ask for both")
    out = []
    for base in range(0, n, 32):
        chunk = ws[base:base + 32]
        segs = (_lib.AqTernSeg * len(chunk))()
        dev = chunk[0].device
        aux = torch.empty((len(chunk), 8), dtype=torch.float32, device=dev)  # [stats(4) | q(3) | pad]
        for i, w in enumerate(chunk):
            d = {"stats": aux[i, :4], "q": aux[i, 4:7]}
            segs[i].W, segs[i].n = _lib.ptr(w, name="w[%d]" % (base + i)), w.numel()
            segs[i].stats, segs[i].q = aux.data_ptr() + 32 * i, aux.data_ptr() + 32 * i + 16
            if codes:
                d["codes"] = torch.empty(w.shape, dtype=torch.uint8, device=dev)
                segs[i].codes = d["codes"].data_ptr()
            if packed:
                d["packed"] = torch.empty((w.numel() + 3) // 4, dtype=torch.uint8, device=dev)
                segs[i].packed = d["packed"].data_ptr()
            if decoded:
                if out_Wt is not None:
                    if out_Wt[base + i].numel() != w.numel():
                        raise ValueError("tern_forward: out_Wt[%d] has %d elements, the weight %d" % (base + i, out_Wt[base + i].numel(), w.numel()))
                    d["Wt"] = out_Wt[base + i]
                else:
                    d["Wt"] = torch.empty(w.shape, dtype=torch.float32, device=dev)
                segs[i].Wt = _lib.ptr(d["Wt"], name="Wt")
            if mode == "ttq":
                segs[i].Wp = _lib.ptr(Wp[base + i], name="Wp")
                segs[i].Wn = _lib.ptr(Wn[base + i], name="Wn")
            out.append(d)
        nbytes = _lib.lib().aq_tern_workspace_bytes(segs, len(chunk))
        wsb = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=dev)
        _lib.check(_lib.lib().aq_tern_fwd(segs, len(chunk), _TERN_MODES[mode], float(factor), int(bool(threshold_from_max)), wsb.data_ptr(),
                                          nbytes, _lib.stream()))
    return out


def tern_backward(ws, dWts, mode, stats=None, Wp=None, Wn=None, masks=None):
    """Gradients of the latent weights from d loss / d (quantized weights): "twn" straight through; "ttq" scaled by
    region (stats[i] as tern_forward left it gives delta) plus (dWp[i], dWn[i]); "inq" masked by masks[i] (uint8,
    1 = trainable).  -> (dW list, dWp list | None, dWn list | None)."""
    n = len(ws)
    dWs, dps, dns = [], ([] if mode == "ttq" else None), ([] if mode == "ttq" else None)
    for base in range(0, n, 32):
        chunk = ws[base:base + 32]
        segs = (_lib.AqTernSeg * len(chunk))()
        dev = chunk[0].device
        sg = torch.empty((len(chunk), 2), dtype=torch.float32, device=dev) if mode == "ttq" else None
        for i, w in enumerate(chunk):
            g = dWts[base + i]
            if g.numel() != w.numel():
                raise ValueError("tern_backward: gradient %d has %d elements, the weight %d" % (base + i, g.numel(), w.numel()))
            dw = torch.empty_like(w)
            dWs.append(dw)
            segs[i].W, segs[i].n = _lib.ptr(w, name="w"), w.numel()
            segs[i].dWt, segs[i].dW = _lib.ptr(g, name="dWt"), _lib.ptr(dw)
            if mode == "ttq":
                segs[i].stats = _lib.ptr(stats[base + i], name="stats")
                segs[i].Wp, segs[i].Wn = _lib.ptr(Wp[base + i], name="Wp"), _lib.ptr(Wn[base + i], name="Wn")
                segs[i].dWp, segs[i].dWn = sg.data_ptr() + 8 * i, sg.data_ptr() + 8 * i + 4
                dps.append(sg[i, 0])
                dns.append(sg[i, 1])
            if mode == "inq":
                segs[i].mask = _lib.ptr(masks[base + i], torch.uint8, "mask")
        nbytes = _lib.lib().aq_tern_workspace_bytes(segs, len(chunk))
        wsb = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=dev)
        _lib.check(_lib.lib().aq_tern_bwd(segs, len(chunk), _TERN_MODES[mode], wsb.data_ptr(), nbytes, _lib.stream()))
    return dWs, dps, dns


def tern_backward_adam(ws, dWts, mode, exp_avg, exp_avg_sq, steps, lr, betas=(0.9, 0.999), eps=1e-8, reg=0.0, stats=None, Wp=None, Wn=None,
                       scale_state=None, masks=None, out_dW=None, out_dscale=None):
    """tern_backward with torch.optim.Adam's update of the latent weights (and of the TTQ scales) applied in the same
    pass: ws[i], exp_avg[i], exp_avg_sq[i] are updated in place, steps[i] (one-element fp32 tensors, the optimizer's
    state["step"]) advanced by one; reg * w is added to the gradient first (the regulariser's 2 * l2_reg * w).
    mode "ttq": Wp[i], Wn[i] are updated too, from scale_state[i] = (Wp exp_avg, Wp exp_avg_sq, Wp step, Wn exp_avg,
    Wn exp_avg_sq, Wn step).  out_dW: optional tensors that receive the latent gradients (they are not stored otherwise);
    out_dscale[i] = (dWp, dWn) one-element tensors likewise for the TTQ scale gradients."""
    n = len(ws)
    for base in range(0, n, 32):
        chunk = ws[base:base + 32]
        segs = (_lib.AqTernSeg * len(chunk))()
        dev = chunk[0].device
        for i, w in enumerate(chunk):
            j = base + i
            for t, what in ((dWts[j], "gradient"), (exp_avg[j], "exp_avg"), (exp_avg_sq[j], "exp_avg_sq")):
                if t.numel() != w.numel():
                    raise ValueError("tern_backward_adam: %s %d has %d elements, the weight %d" % (what, j, t.numel(), w.numel()))
            segs[i].W, segs[i].n = _lib.ptr(w, name="w"), w.numel()
            segs[i].dWt = _lib.ptr(dWts[j], name="dWt")
            segs[i].adam_m, segs[i].adam_v = _lib.ptr(exp_avg[j], name="exp_avg"), _lib.ptr(exp_avg_sq[j], name="exp_avg_sq")
            segs[i].adam_step = _lib.ptr(steps[j], name="step")
            if out_dW is not None:
                segs[i].dW = _lib.ptr(out_dW[j], name="dW")
            if mode == "ttq":
                segs[i].stats = _lib.ptr(stats[j], name="stats")
                segs[i].Wp, segs[i].Wn = _lib.ptr(Wp[j], name="Wp"), _lib.ptr(Wn[j], name="Wn")
                st = scale_state[j]
                segs[i].Wp_m, segs[i].Wp_v, segs[i].Wp_step = (_lib.ptr(t, name="Wp state") for t in st[:3])
                segs[i].Wn_m, segs[i].Wn_v, segs[i].Wn_step = (_lib.ptr(t, name="Wn state") for t in st[3:])
                if out_dscale is not None:
                    segs[i].dWp, segs[i].dWn = _lib.ptr(out_dscale[j][0], name="dWp"), _lib.ptr(out_dscale[j][1], name="dWn")
            if mode == "inq":
                segs[i].mask = _lib.ptr(masks[j], torch.uint8, "mask")
        nbytes = _lib.lib().aq_tern_workspace_bytes(segs, len(chunk))
        wsb = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=dev)
        _lib.check(_lib.lib().aq_tern_bwd_adam(segs, len(chunk), _TERN_MODES[mode], float(lr), float(betas[0]), float(betas[1]), float(eps), float(reg),
                                               wsb.data_ptr(), nbytes, _lib.stream()))


def inq_effective(ws, masks, frozen, out=None, sumsq=False):
    """Effective weights of INQ layers in one launch: out[i] = where(masks[i] != 0, ws[i], frozen[i]).
    sumsq=True: -> (out, ss) with ss[i] = sum(ws[i] ** 2) from the same pass (one float per layer)."""
    n = len(ws)
    outs = out if out is not None else [torch.empty_like(w) for w in ws]
    ss = torch.empty(n, dtype=torch.float32, device=ws[0].device) if sumsq and n else None
    for base in range(0, n, 32):
        chunk = ws[base:base + 32]
        segs = (_lib.AqTernSeg * len(chunk))()
        for i, w in enumerate(chunk):
            j = base + i
            if masks[j].numel() != w.numel() or frozen[j].numel() != w.numel() or outs[j].numel() != w.numel():
                raise ValueError("inq_effective: layer %d: mask / frozen / out do not match the weight's %d elements" % (j, w.numel()))
            segs[i].W, segs[i].n = _lib.ptr(w, name="w"), w.numel()
            segs[i].mask = _lib.ptr(masks[j], torch.uint8, "mask")
            segs[i].frozen = _lib.ptr(frozen[j], name="frozen")
            segs[i].Wt = _lib.ptr(outs[j], name="out")
        if ss is None:
            _lib.check(_lib.lib().aq_inq_effective(segs, len(chunk), _lib.stream()))
        else:
            nbytes = _lib.lib().aq_tern_workspace_bytes(segs, len(chunk))
            wsb = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=ws[0].device)
            _lib.check(_lib.lib().aq_inq_effective_sumsq(segs, len(chunk), ss[base:].data_ptr(), wsb.data_ptr(), nbytes, _lib.stream()))
    return (outs, ss) if sumsq else outs


def inq_partition(w, mask, tau, n1, n2, codes=None):
    """One INQ partition step in place on w (fp32) and mask (uint8): trainable weights with |w| >= tau are rounded
    to {0, +-2^n2 .. +-2^n1} and frozen; codes (uint8, optional) receives the level codes (255 = still trainable)."""
    if mask.shape != w.shape or (codes is not None and codes.shape != w.shape):
        raise ValueError("inq_partition: mask / codes must have the shape of w")
    _lib.check(_lib.lib().aq_inq_partition(_lib.ptr(w, name="w"), _lib.ptr(mask, torch.uint8, "mask"), _lib.ptr(codes, torch.uint8, "codes"),
                                           w.numel(), float(tau), int(n1), int(n2), _lib.stream()))


# ------------------------------------------------------------- probabilistic knowledge transfer (PKT) loss

def pkt_loss(student, teacher, eps=1e-7, need_grad=True, grad_scale=None):
    """Cosine-kernel PKT divergence between the batch's student and teacher features (N, Ds) / (N, Dt):
    -> (loss fp32 [1], d loss / d student or None).  grad_scale: optional one-element device tensor the gradient
    is multiplied by."""
    if student.dim() != 2 or teacher.dim() != 2 or student.shape[0] != teacher.shape[0]:
        raise ValueError("pkt_loss: student %s / teacher %s must be (N, D) matrices of one batch" % (tuple(student.shape), tuple(teacher.shape)))
    N, Ds, Dt = int(student.shape[0]), int(student.shape[1]), int(teacher.shape[1])
    loss = torch.empty(1, dtype=torch.float32, device=student.device)
    d = torch.empty_like(student) if need_grad else None
    nbytes = _lib.lib().aq_pkt_workspace_bytes(N, Ds, Dt)
    ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=student.device)
    _lib.check(_lib.lib().aq_pkt_loss(_lib.ptr(student, name="student"), _lib.ptr(teacher, name="teacher"), N, Ds, Dt, float(eps), _lib.ptr(loss),
                                      _lib.ptr(grad_scale, name="grad_scale"), _lib.ptr(d), ws.data_ptr(), nbytes, _lib.stream()))
    return loss, d


def pkt_kernel_matrix(feat, eps=1e-7):
    """(cos(x_i, x_j) + 1) / 2 for all pairs of rows of feat (N, D) -> (N, N), on the fp32 matrix cores."""
    N, D = int(feat.shape[0]), int(feat.shape[1])
    K = torch.empty((N, N), dtype=torch.float32, device=feat.device)
    nbytes = ((N * D * 4 + 255) // 256 + (N * 4 + 255) // 256) * 256
    ws = torch.empty(nbytes, dtype=torch.uint8, device=feat.device)
    _lib.check(_lib.lib().aq_pkt_kernel_matrix(_lib.ptr(feat, name="feat"), N, D, float(eps), _lib.ptr(K), ws.data_ptr(), nbytes, _lib.stream()))
    return K
