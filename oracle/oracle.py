"""ctypes front end of the CPU oracle (libaq_oracle.so, built from aq_oracle.c).

TEST INFRASTRUCTURE ONLY.  Importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg; never from the product package.  All functions take
and return numpy arrays (float32 unless stated).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "libaq_oracle.so")
    src = os.path.join(_HERE, "aq_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libaq_oracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libaq_oracle.so")
        if not os.path.exists(so):
            build()
        L = C.CDLL(so)
        fp, ip, dp = C.POINTER(C.c_float), C.POINTER(C.c_int32), C.POINTER(C.c_double)
        L.aqo_expf.argtypes, L.aqo_expf.restype = [C.c_float], C.c_float
        L.aqo_synth_fwd.argtypes = [fp, fp, C.c_int64, C.c_int, C.c_float, fp, ip, fp, dp]
        L.aqo_synth_bwd.argtypes = [fp, fp, C.c_int64, C.c_int, C.c_float, fp, C.c_float, fp, dp]
        conv = [fp, fp] + [C.c_int] * 8 + [fp]
        L.aqo_conv_fwd.argtypes = conv
        L.aqo_conv_dgrad.argtypes = conv
        L.aqo_conv_wgrad.argtypes = conv
        L.aqo_linear_fwd.argtypes = [fp, fp, fp, C.c_int, C.c_int, C.c_int, fp]
        L.aqo_linear_bwd.argtypes = [fp, fp, fp, C.c_int, C.c_int, C.c_int, fp, fp, fp]
        L.aqo_bn_relu_fwd.argtypes = [fp, fp, fp, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, fp, dp]
        L.aqo_bn_relu_bwd.argtypes = [fp, fp, fp, fp, fp, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, fp, fp, fp]
        L.aqo_adam_step.argtypes = [fp, fp, fp, fp, C.c_int64, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double]
        L.aqo_adam_step.restype = None
        u8p = C.POINTER(C.c_uint8)
        L.aqo_tern_stats.argtypes = [fp, C.c_int64, dp, dp]
        L.aqo_tern_codes.argtypes = [fp, C.c_int64, C.c_float, u8p, dp, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        L.aqo_tern_bwd.argtypes = [fp, fp, C.c_int64, C.c_int, C.c_float, C.c_float, C.c_float, u8p, fp, dp, dp]
        L.aqo_inq_partition.argtypes = [fp, u8p, u8p, C.c_int64, C.c_float, C.c_int, C.c_int]
        for f in ("aqo_tern_stats", "aqo_tern_codes", "aqo_tern_bwd", "aqo_inq_partition"):
            getattr(L, f).restype = None
        L.aqo_num_threads.restype = C.c_int
        L.aqo_set_num_threads.argtypes = [C.c_int]
        for f in ("aqo_synth_fwd", "aqo_synth_bwd", "aqo_conv_fwd", "aqo_conv_dgrad", "aqo_conv_wgrad",
                  "aqo_linear_fwd", "aqo_linear_bwd", "aqo_set_num_threads", "aqo_bn_relu_fwd", "aqo_bn_relu_bwd"):
            getattr(L, f).restype = None
        _LIB = L
    return _LIB


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(C.POINTER(C.c_float))


def num_threads():
    return lib().aqo_num_threads()


def set_num_threads(n):
    lib().aqo_set_num_threads(int(n))


def expf(x):
    x = np.asarray(x, dtype=np.float32)
    out = np.empty_like(x)
    L = lib()
    flat_in, flat_out = x.reshape(-1), out.reshape(-1)
    for i in range(flat_in.size):
        flat_out[i] = L.aqo_expf(float(flat_in[i]))
    return out


def synth_fwd(p_c, q, beta, want_idx=True):
    """p_c (..., L), q (L,) -> dict(W (...), idx int32 (...), W_hard (...), sumsq float64)."""
    p, pp = _f(p_c)
    qa, qp = _f(q)
    L = p.shape[-1]
    rows = p.size // L
    W = np.empty(p.shape[:-1], np.float32)
    Wh = np.empty(p.shape[:-1], np.float32)
    idx = np.empty(p.shape[:-1], np.int32)
    ss = C.c_double(0.0)
    lib().aqo_synth_fwd(pp, qp, rows, L, float(beta), W.ctypes.data_as(C.POINTER(C.c_float)),
                        idx.ctypes.data_as(C.POINTER(C.c_int32)) if want_idx else None,
                        Wh.ctypes.data_as(C.POINTER(C.c_float)) if want_idx else None, C.byref(ss))
    return {"W": W, "idx": idx, "W_hard": Wh, "sumsq": ss.value}


def synth_bwd(p_c, q, beta, dW, wd=0.0, want_dp=True):
    """-> dp_c (same shape as p_c), dq float64 (L,).  g = dW + wd*W per weight."""
    p, pp = _f(p_c)
    qa, qp = _f(q)
    L = p.shape[-1]
    rows = p.size // L
    if dW is None:
        g, gp = None, None
    else:
        g, gp = _f(dW)
        assert g.size == rows
    dp = np.empty_like(p) if want_dp else None
    dq = np.zeros(L, np.float64)
    lib().aqo_synth_bwd(pp, qp, rows, L, float(beta), gp, float(wd),
                        dp.ctypes.data_as(C.POINTER(C.c_float)) if want_dp else None,
                        dq.ctypes.data_as(C.POINTER(C.c_double)))
    return dp, dq


def _out_hw(H, ks, stride, pad):
    return (H + 2 * pad - ks) // stride + 1


def conv_fwd(x, W, stride=1, pad=1):
    xa, xp = _f(x)
    wa, wp = _f(W)
    N, Cin, H, Wd = xa.shape
    Cout, _, ks, _ = wa.shape
    y = np.empty((N, Cout, _out_hw(H, ks, stride, pad), _out_hw(Wd, ks, stride, pad)), np.float32)
    lib().aqo_conv_fwd(xp, wp, N, Cin, H, Wd, Cout, ks, stride, pad, y.ctypes.data_as(C.POINTER(C.c_float)))
    return y


def conv_dgrad(dy, W, x_shape, stride=1, pad=1):
    ga, gp = _f(dy)
    wa, wp = _f(W)
    N, Cin, H, Wd = x_shape
    Cout, _, ks, _ = wa.shape
    dx = np.empty(x_shape, np.float32)
    lib().aqo_conv_dgrad(gp, wp, N, Cin, H, Wd, Cout, ks, stride, pad, dx.ctypes.data_as(C.POINTER(C.c_float)))
    return dx


def conv_wgrad(x, dy, ks=3, stride=1, pad=1):
    xa, xp = _f(x)
    ga, gp = _f(dy)
    N, Cin, H, Wd = xa.shape
    Cout = ga.shape[1]
    dW = np.empty((Cout, Cin, ks, ks), np.float32)
    lib().aqo_conv_wgrad(xp, gp, N, Cin, H, Wd, Cout, ks, stride, pad, dW.ctypes.data_as(C.POINTER(C.c_float)))
    return dW


def linear_fwd(x, W, b):
    xa, xp = _f(x)
    wa, wp = _f(W)
    ba, bp = _f(b)
    N, In = xa.shape
    Out = wa.shape[1]
    y = np.empty((N, Out), np.float32)
    lib().aqo_linear_fwd(xp, wp, bp, N, In, Out, y.ctypes.data_as(C.POINTER(C.c_float)))
    return y


def linear_bwd(x, W, dy):
    xa, xp = _f(x)
    wa, wp = _f(W)
    ga, gp = _f(dy)
    N, In = xa.shape
    Out = wa.shape[1]
    dx, dW, db = np.empty((N, In), np.float32), np.empty((In, Out), np.float32), np.empty(Out, np.float32)
    fp = C.POINTER(C.c_float)
    lib().aqo_linear_bwd(xp, wp, gp, N, In, Out, dx.ctypes.data_as(fp), dW.ctypes.data_as(fp), db.ctypes.data_as(fp))
    return dx, dW, db


# ---- level initialisers (host logic of the reference, restated) -------------------------------

def bn_relu_fwd(x, gamma, beta, eps=1e-5, relu=True):
    """-> y, mean (C,), rstd (C,), unbiased variance (C,) (float64 statistics)."""
    xa, xp = _f(x)
    ga, gp = _f(gamma)
    ba, bp = _f(beta)
    N, Cn = xa.shape[0], xa.shape[1]
    HW = xa.size // (N * Cn)
    y = np.empty_like(xa)
    st = np.zeros((Cn, 3), np.float64)
    lib().aqo_bn_relu_fwd(xp, gp, bp, N, Cn, HW, float(eps), int(relu), y.ctypes.data_as(C.POINTER(C.c_float)),
                          st.ctypes.data_as(C.POINTER(C.c_double)))
    return y, st[:, 0].copy(), st[:, 1].copy(), st[:, 2].copy()


def bn_relu_bwd(dy, x, gamma, beta, res=None, eps=1e-5, relu=True):
    """-> dx (+res), dgamma, dbeta"""
    da, dyp = _f(dy)
    xa, xp = _f(x)
    ga, gp = _f(gamma)
    ba, bp = _f(beta)
    ra, rp = _f(res) if res is not None else (None, None)
    N, Cn = xa.shape[0], xa.shape[1]
    HW = xa.size // (N * Cn)
    fp = C.POINTER(C.c_float)
    dx, dg, db = np.empty_like(xa), np.empty(Cn, np.float32), np.empty(Cn, np.float32)
    lib().aqo_bn_relu_bwd(dyp, xp, gp, bp, rp, N, Cn, HW, float(eps), int(relu), dx.ctypes.data_as(fp), dg.ctypes.data_as(fp),
                          db.ctypes.data_as(fp))
    return dx, dg, db


def adam_step(p, g, m, v, t, lr, b1=0.9, b2=0.999, eps=1e-8):
    """In-place torch.optim.Adam update of fp32 arrays p, m, v for step number t (1-based)."""
    for a in (p, m, v):
        assert a.dtype == np.float32 and a.flags.c_contiguous
    ga, gp = _f(g)
    fp = C.POINTER(C.c_float)
    lib().aqo_adam_step(p.ctypes.data_as(fp), gp, m.ctypes.data_as(fp), v.ctypes.data_as(fp), p.size, int(t), lr, b1, b2, eps)


def pack_codes(idx, bits):
    """Little-endian bit stream of `bits`-bit codes: code i occupies stream bits [i*bits, (i+1)*bits)."""
    idx = np.ascontiguousarray(idx, dtype=np.uint8).reshape(-1)
    b = np.unpackbits(idx[:, None], axis=1, bitorder="little")[:, :bits]
    return np.packbits(b.reshape(-1), bitorder="little")


def unpack_codes(packed, n, bits):
    b = np.unpackbits(np.ascontiguousarray(packed, dtype=np.uint8), bitorder="little")[:n * bits].reshape(n, bits)
    full = np.zeros((n, 8), np.uint8)
    full[:, :bits] = b
    return np.packbits(full, axis=1, bitorder="little").reshape(-1)


def augment_batch(data, index, off_y, off_x, flip, pad=4, mean=(0.4915, 0.4821, 0.4462), std=(0.2472, 0.2437, 0.2617)):
    """torchvision's RandomCrop(H, padding=pad) + RandomHorizontalFlip + ToTensor + Normalize for given
    random draws (adp_qtz_fan.py:164-169), image by image, in fp32: pad with zeros, crop, mirror,
    u8/255, (v - mean)/std.  data (N, H, W, C) uint8 -> (B, C, H, W) float32."""
    data = np.asarray(data, np.uint8)
    N, H, W, Cn = data.shape
    B = len(index)
    out = np.empty((B, Cn, H, W), np.float32)
    mean = np.asarray(mean, np.float32)[:Cn]
    std = np.asarray(std, np.float32)[:Cn]
    for b in range(B):
        img = np.zeros((H + 2 * pad, W + 2 * pad, Cn), np.uint8)
        img[pad:pad + H, pad:pad + W] = data[index[b]]
        crop = img[off_y[b]:off_y[b] + H, off_x[b]:off_x[b] + W]
        if flip[b]:
            crop = crop[:, ::-1]
        t = crop.astype(np.float32) / np.float32(255.0)
        t = (t - mean) / std
        out[b] = t.transpose(2, 0, 1)
    return out


def ce_loss(logits, labels):
    """nn.CrossEntropyLoss() (mean reduction, adp_qtz_fan.py:202, 288) and d loss / d logits, in float64."""
    x = np.asarray(logits, np.float64)
    y = np.asarray(labels, np.int64)
    m = x.max(axis=1, keepdims=True)
    lse = m + np.log(np.exp(x - m).sum(axis=1, keepdims=True))
    n = x.shape[0]
    loss = float((lse[:, 0] - x[np.arange(n), y]).mean())
    d = np.exp(x - lse)
    d[np.arange(n), y] -= 1.0
    return loss, d / n


def levels_fix(s, num_level, liar=True):
    """adp_qtz_fix.py:661-665: [-2^t .. descending magnitude], 0, [2^t ..]; num_level-1 entries."""
    import math
    t2 = math.floor(math.log(s, 2)) if liar else round(math.log(s, 2))
    t3 = np.arange(t2 - num_level / 2 + 2, t2 + 1)
    pos = (2.0 ** t3).astype(np.float32)
    return np.concatenate([np.sort(-pos), np.zeros(1, np.float32), pos]).astype(np.float32)


def levels_hint(s, num_level):
    """adp_qtz_hint.py:657-662: num_level entries, no zero."""
    import math
    t2 = math.floor(math.log(s, 2))
    t3 = np.arange(t2 - num_level / 2 + 1, t2 + 1)
    pos = (2.0 ** t3).astype(np.float32)
    return np.concatenate([np.sort(-pos), pos]).astype(np.float32)


def fan_bound(fan_in):
    """adp_qtz_fan.py:551-560: q_level ~ U(-b, b)."""
    import math
    gain = math.sqrt(2.0 / (1 + 5.0))
    return math.sqrt(3.0) * gain / math.sqrt(fan_in)


# ---- ternary / power-of-two quantisers: PARITY UNPINNED (no reference implementation; restated from the papers
# cited at /root/reference/README.md:3-29, see aq_oracle.c) ---------------------------------------------------------

def tern_stats(w):
    """-> (mean |w|, max |w|) in float64."""
    a, ap = _f(w)
    mean, mx = C.c_double(0.0), C.c_double(0.0)
    lib().aqo_tern_stats(ap, a.size, C.byref(mean), C.byref(mx))
    return mean.value, mx.value


def tern_codes(w, delta):
    """Codes for a GIVEN threshold (float32) -> dict(codes uint8 (shape of w), sum_abs float64, npos, nneg)."""
    a, ap = _f(w)
    codes = np.empty(a.shape, np.uint8)
    s, p, q = C.c_double(0.0), C.c_int64(0), C.c_int64(0)
    lib().aqo_tern_codes(ap, a.size, float(np.float32(delta)), codes.ctypes.data_as(C.POINTER(C.c_uint8)), C.byref(s), C.byref(p), C.byref(q))
    return {"codes": codes, "sum_abs": s.value, "npos": p.value, "nneg": q.value}


def tern_forward(w, mode="twn", factor=0.7, threshold_from_max=False, Wp=None, Wn=None):
    """TWN / TTQ forward in float64 statistics: -> dict(delta, pos, neg, codes, Wt float32)."""
    mean, mx = tern_stats(w)
    delta = np.float32(factor * (mx if threshold_from_max else mean))
    r = tern_codes(w, delta)
    if mode == "ttq":
        pos, neg = np.float32(Wp), np.float32(Wn)
    else:
        cnt = r["npos"] + r["nneg"]
        pos = neg = np.float32(r["sum_abs"] / cnt) if cnt else np.float32(0.0)
    q = np.array([0.0, pos, -neg], np.float32)
    return {"delta": delta, "pos": pos, "neg": neg, "codes": r["codes"], "Wt": q[r["codes"]], "count": r["npos"] + r["nneg"]}


def tern_backward(w, dwt, mode, delta=0.0, pos=1.0, neg=1.0, mask=None):
    """-> (dw float32, dWp, dWn float64).  mode: "twn" | "ttq" | "inq"."""
    a, ap = _f(w)
    g, gp = _f(dwt)
    dw = np.empty_like(a)
    m = np.ascontiguousarray(mask, dtype=np.uint8) if mask is not None else None
    dwp, dwn = C.c_double(0.0), C.c_double(0.0)
    lib().aqo_tern_bwd(ap, gp, a.size, {"twn": 0, "ttq": 1, "inq": 2}[mode], float(np.float32(delta)), float(np.float32(pos)),
                       float(np.float32(neg)), m.ctypes.data_as(C.POINTER(C.c_uint8)) if m is not None else None,
                       dw.ctypes.data_as(C.POINTER(C.c_float)), C.byref(dwp), C.byref(dwn))
    return dw, dwp.value, dwn.value


def inq_exponents(max_abs, bits=5):
    """n1 = floor(log2(4 s / 3)), n2 = n1 + 1 - 2^(bits-1) / 2  (Zhou et al. 2017, eq. 3)."""
    import math
    n1 = int(math.floor(math.log2(4.0 * float(max_abs) / 3.0)))
    return n1, n1 + 1 - (2 ** (bits - 1)) // 2


def inq_partition(w, mask, tau, n1, n2):
    """One partition step on copies: -> (w, mask, codes)."""
    a = np.array(w, dtype=np.float32, copy=True, order="C")
    m = np.array(mask, dtype=np.uint8, copy=True, order="C")
    codes = np.empty(a.shape, np.uint8)
    u8 = C.POINTER(C.c_uint8)
    lib().aqo_inq_partition(a.ctypes.data_as(C.POINTER(C.c_float)), m.ctypes.data_as(u8), codes.ctypes.data_as(u8), a.size,
                            float(np.float32(tau)), int(n1), int(n2))
    return a, m, codes


def pkt_loss(student, teacher, eps=1e-7):
    """PKT divergence of Passalis & Tefas (2018) with the cosine kernel, float64: PARITY UNPINNED (no reference
    implementation; restated from the published method): rows normalised by (|x| + eps), K = (xh xh^T + 1) / 2,
    P = K / row sums, loss = mean(Pt * log((Pt + eps) / (Ps + eps)))."""
    def cond(x):
        x = np.asarray(x, np.float64)
        xh = x / (np.sqrt((x ** 2).sum(axis=1, keepdims=True)) + eps)
        xh[np.isnan(xh)] = 0.0
        K = (xh @ xh.T + 1.0) / 2.0
        return K / K.sum(axis=1, keepdims=True)
    Ps, Pt = cond(student), cond(teacher)
    return float(np.mean(Pt * np.log((Pt + eps) / (Ps + eps))))
