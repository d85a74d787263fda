"""Checkpoints in the reference's on-disk layout, and packed export of hard-mode weights.

Training checkpoints and best-model files are the dictionaries the reference's `train()` writes
with torch.save (adp_qtz_fan.py:331-340 and :376-379) and reads back at :249-263 / :421-425 /
adp_qtz_fan_test.py:147-160:

    checkpoint  {"beta", "epoch", "iter_idx", "best_val_acc", "model", "optimizer_prim",
                 "optimizer_q", "optimizer"}
    best model  {"beta", "model"}

`model` is the module's state_dict (same keys and shapes as the reference's classes), the three
optimizer entries are torch.optim.Adam state_dicts, so files move between the reference scripts and
this library in both directions.  Files are read with torch.load(weights_only=True).

Packed export: in hard ("test") mode a quantized layer's weights are q_level[idx] with idx = argmax
s_c (adp_qtz_fan_test.py:304-305), so a trained layer is fully described by its codes (5 bits each
for the 31/32-level layers) and its codebook.  `export_packed` turns a model into that form (codes
bit-packed on the GPU); for the power-of-two variants (adp_qtz_fix / adp_qtz_hint) the codebook is also
given as sign and exponent, which is what shift-add inference needs.  The reference has no packed
file format; this one is specific to this library.
"""
import math

import torch

from . import layers, ops

CHECKPOINT_KEYS = ("beta", "epoch", "iter_idx", "best_val_acc", "model", "optimizer_prim", "optimizer_q", "optimizer")
PACKED_FORMAT = "aq-packed-1"


# ------------------------------------------------------------------------------ training checkpoints

def _plain(v):
    """python scalar for numpy / tensor scalars (the reference stores e.g. a numpy float for best_val_acc)"""
    if isinstance(v, torch.Tensor):
        return v.item()
    return v.item() if hasattr(v, "item") else v


def _safe_load(path):
    """torch.load that executes nothing from the file: tensors, containers and python scalars, plus numpy
    scalars (the reference stores best_val_acc as the numpy float that np.mean returns)."""
    import numpy as np
    allow = []
    try:
        scalar = np._core.multiarray.scalar
    except AttributeError:  # numpy 1.x
        scalar = np.core.multiarray.scalar
    allow += [(scalar, "numpy.core.multiarray.scalar"), (scalar, "numpy._core.multiarray.scalar"), np.dtype]
    allow += [type(np.dtype(t)) for t in (np.float32, np.float64, np.int64, np.int32)]
    with torch.serialization.safe_globals(allow):
        return torch.load(path, map_location="cpu", weights_only=True)


def _load_optimizer(opt, state):
    """optimizer.load_state_dict also takes the saved param_groups' implementation switches (a file
    written by the reference's plain optim.Adam says fused=None, capturable=False) and leaves the step
    counters on the CPU accordingly.  The switches describe how this process runs the update, not the
    training state: keep this optimizer's own and put the counters where they then belong."""
    keep = [{k: g.get(k) for k in ("fused", "capturable", "foreach")} for g in opt.param_groups]
    # State tensors that already exist keep their storage: a step object that has captured its HIP graphs holds
    # their addresses (Adam kernels, the fused synthesis-gradient update), so the loaded values are copied INTO
    # them; load_state_dict alone would leave the graphs updating orphaned tensors.
    old = {p: dict(st) for p, st in opt.state.items()}
    opt.load_state_dict(state)
    for g, flags in zip(opt.param_groups, keep):
        g.update(flags)
        for p in g["params"]:
            st = opt.state.get(p)
            if not st:
                continue
            if (flags.get("fused") or flags.get("capturable")) and "step" in st:
                st["step"] = torch.as_tensor(st["step"], dtype=torch.float32).to(p.device)
            for k, prev in old.get(p, {}).items():
                new = st.get(k)
                if isinstance(prev, torch.Tensor) and isinstance(new, torch.Tensor) and prev.shape == new.shape and prev.dtype == new.dtype \
                        and prev.device == new.device and prev.data_ptr() != new.data_ptr():
                    prev.copy_(new)
                    st[k] = prev


_ROLE_KEYS = (("prim", "optimizer_prim"), ("q", "optimizer_q"), ("rest", "optimizer"))


def _by_role(opts):
    """{role: optimizer} for train.OptimizerSet (models without p_c / q_level hold fewer than three optimizers), or
    for a plain sequence of exactly three in the reference's order (optimizer_prim, optimizer_q, optimizer)."""
    roles = getattr(opts, "roles", None)
    if roles is None:
        if len(opts) != 3:
            raise ValueError("checkpoint: %d optimizers without roles; pass train.make_optimizers' result or the reference's "
                             "three (optimizer_prim, optimizer_q, optimizer)" % len(opts))
        roles = ("prim", "q", "rest")
    return dict(zip(roles, opts))


def save_checkpoint(path, model, opts, beta, epoch, iter_idx, best_val_acc):
    """adp_qtz_fan.py:331-340.  opts = (optimizer_prim, optimizer_q, optimizer) or a train.OptimizerSet; a group the
    model does not have (ternary / INQ models: no p_c, no q_level) is written as an empty Adam state_dict, so the file
    keeps the reference's eight keys."""
    by = _by_role(opts)
    empty = {"state": {}, "param_groups": []}
    res = {
        "beta": _plain(beta),
        "epoch": int(epoch),
        "iter_idx": int(iter_idx),
        "best_val_acc": float(_plain(best_val_acc)),
        "model": model.state_dict(),
    }
    for role, key in _ROLE_KEYS:
        res[key] = by[role].state_dict() if role in by else dict(empty)
    torch.save(res, path)


def load_checkpoint(path, model, opts=None):
    """adp_qtz_fan.py:249-263: restores the model and (when given) the optimizers, each from the entry of its role;
    returns (beta, start_epoch, iter_idx, best_val_acc).  The caller assigns beta to the variant module's global
    (`adp_qtz_fan.beta = beta`)."""
    res = _safe_load(path)
    missing = [k for k in CHECKPOINT_KEYS if k not in res]
    if missing:
        raise KeyError("%s is not a training checkpoint: missing %s" % (path, missing))
    model.load_state_dict(res["model"])
    if opts is not None:
        by = _by_role(opts)
        for role, key in _ROLE_KEYS:
            if role in by:
                if not res[key].get("param_groups"):
                    raise KeyError("%s holds no state for the %r optimizer of this model" % (path, key))
                _load_optimizer(by[role], res[key])
            elif res[key].get("param_groups"):
                raise KeyError("%s holds %r state, but the model has no such parameter group" % (path, key))
    return _plain(res["beta"]), int(res["epoch"]), int(res["iter_idx"]), float(_plain(res["best_val_acc"]))


def save_bestmodel(path, model, beta):
    """adp_qtz_fan.py:376-379"""
    torch.save({"beta": _plain(beta), "model": model.state_dict()}, path)


def load_bestmodel(path, model):
    """adp_qtz_fan.py:421-425, adp_qtz_fan_test.py:149-160 (both file kinds carry "beta" and "model");
    returns beta."""
    res = _safe_load(path)
    model.load_state_dict(res["model"])
    return _plain(res["beta"])


# ------------------------------------------------------------------------------------ packed export

def _pow2_codebook(q):
    """sign (int8: -1, 0, +1) and exponent (int8) with q == sign * 2**exponent, or None when a level is
    not an exact power of two (learned levels of adp_qtz_fan, or adp_qtz_hint after training)."""
    m, e = torch.frexp(q.detach().float().cpu())  # q = m * 2**e, |m| in [0.5, 1)
    ok = (q == 0).cpu() | (m.abs() == 0.5)
    if not bool(ok.all()):
        return None
    sign = torch.sign(q.detach().cpu()).to(torch.int8)
    expo = torch.where(sign == 0, torch.zeros_like(e), e - 1).to(torch.int8)
    return sign, expo


def export_packed(model, beta=None):
    """-> dict of plain tensors / scalars (loadable with weights_only=True):
         "format", "beta", "layers": {name: {"codes" uint8 bit stream, "bits", "shape", "q_level",
         ["sign", "exponent"]}}, "dense": state_dict entries that are not p_c / q_level
    Codes come from one hard-mode synthesis launch over all layers and are packed on the device."""
    named = [(n, m) for n, m in model.named_modules() if isinstance(m, layers._QuantBase)]
    if not named:
        raise ValueError("export_packed: the model has no quantized layers")
    b = float(named[0][1].current_beta() if beta is None else beta)
    ps = [m.p_c.detach() for _, m in named]
    qs = [m.q_level.detach() for _, m in named]
    out = ops.synth_forward(ps, qs, b, soft=False, codes=True)
    packed = {}
    for (name, m), idx in zip(named, out["idx"]):
        L = m.q_level.numel()
        bits = ops.code_bits(L)
        entry = {"codes": ops.pack_codes(idx.reshape(-1), bits).cpu(), "bits": bits, "shape": list(idx.shape),
                 "q_level": m.q_level.detach().cpu().clone()}
        p2 = _pow2_codebook(m.q_level)
        if p2 is not None:
            entry["sign"], entry["exponent"] = p2
        packed[name] = entry
    dense = {k: v.detach().cpu().clone() for k, v in model.state_dict().items() if not (k.endswith("p_c") or k.endswith("q_level"))}
    return {"format": PACKED_FORMAT, "beta": b, "layers": packed, "dense": dense}


def save_packed(path, model, beta=None):
    torch.save(export_packed(model, beta), path)


def packed_nbytes(packed):
    """Bytes of the quantized layers in packed form (codes + codebooks), and of their fp32 weights."""
    codes = sum(e["codes"].numel() + 4 * e["q_level"].numel() for e in packed["layers"].values())
    dense = sum(4 * math.prod(e["shape"]) for e in packed["layers"].values())
    return codes, dense


def load_packed(path_or_dict, model):
    """Puts packed weights into a model built in hard mode (config.mode == "test"): every quantized
    layer gets its codes and codebook pinned (`layer.set_codes`), so forward passes decode q_level[idx]
    inside the convolution kernel and never touch p_c; dense tensors (conv0, BatchNorm, linear.bias)
    are loaded as usual.  Returns beta."""
    res = path_or_dict if isinstance(path_or_dict, dict) else torch.load(path_or_dict, map_location="cpu", weights_only=True)
    if res.get("format") != PACKED_FORMAT:
        raise ValueError("not a packed export (format %r)" % (res.get("format"),))
    mods = {n: m for n, m in model.named_modules() if isinstance(m, layers._QuantBase)}
    if set(mods) != set(res["layers"]):
        raise KeyError("packed layers %s do not match the model's %s" % (sorted(res["layers"]), sorted(mods)))
    dev = next(model.parameters()).device
    for name, e in res["layers"].items():
        m = mods[name]
        shape = tuple(e["shape"])
        if shape != tuple(m.p_c.shape[:-1]) or e["q_level"].numel() != m.q_level.numel():
            raise ValueError("%s: packed shape %s / %d levels, model has %s / %d" % (
                name, shape, e["q_level"].numel(), tuple(m.p_c.shape[:-1]), m.q_level.numel()))
        idx = ops.unpack_codes(e["codes"].to(dev), math.prod(shape), int(e["bits"])).view(shape)
        m.set_codes(idx, e["q_level"].to(dev))
    missing, unexpected = model.load_state_dict(res["dense"], strict=False)
    bad = [k for k in missing if not (k.endswith("p_c") or k.endswith("q_level"))]
    if bad or unexpected:
        raise KeyError("dense tensors do not match: missing %s unexpected %s" % (bad, list(unexpected)))
    return res["beta"]
