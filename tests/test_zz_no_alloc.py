"""aq_synth_fwd_static (the entry point that allocates nothing: its partial sums live in a device array inside
the library) against aq_synth_fwd (caller-provided workspace), through the C ABI: every output bit for bit,
on two calls in a row, the second with a smaller input, so that stale partial sums of the first call would
show.  Collected last on purpose (a failure here must not hide the other tests under `pytest -x`)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from quantized_cnn_amd import _lib
DEV = "cuda"


def _segs(ps, qs, outs):
    n = len(ps)
    segs = (_lib.AqSeg * n)()
    for i, (p, q) in enumerate(zip(ps, qs)):
        segs[i].p_c, segs[i].q = _lib.ptr(p), _lib.ptr(q)
        segs[i].rows, segs[i].L = p.numel() // p.shape[-1], p.shape[-1]
        segs[i].W = _lib.ptr(outs["W"][i])
        segs[i].idx = _lib.ptr(outs["idx"][i], torch.uint8)
        segs[i].W_hard = _lib.ptr(outs["Wh"][i])
        segs[i].sumsq = outs["ss"].data_ptr() + 4 * i
    return segs


def _outputs(ps):
    return {"W": [torch.full(p.shape[:-1], float("nan"), device=DEV) for p in ps],
            "idx": [torch.full(p.shape[:-1], 255, dtype=torch.uint8, device=DEV) for p in ps],
            "Wh": [torch.full(p.shape[:-1], float("nan"), device=DEV) for p in ps],
            "ss": torch.full((len(ps),), float("nan"), device=DEV)}


def _run_both(ps, qs, beta, flags):
    L = _lib.lib()
    n = len(ps)
    a, b = _outputs(ps), _outputs(ps)
    sa, sb = _segs(ps, qs, a), _segs(ps, qs, b)
    nbytes = L.aq_synth_workspace_bytes(sa, n, 0)
    assert 0 < nbytes <= _lib.STATIC_SCRATCH_BYTES
    ws = torch.empty(nbytes, dtype=torch.uint8, device=DEV)
    _lib.check(L.aq_synth_fwd(sa, n, beta, flags, ws.data_ptr(), nbytes, _lib.stream()))
    torch.cuda.synchronize()
    before = torch.cuda.memory_stats()["allocation.all.allocated"]
    _lib.check(L.aq_synth_fwd_static(sb, n, beta, flags, _lib.stream()))
    assert torch.cuda.memory_stats()["allocation.all.allocated"] == before  # nothing through torch's allocator either
    torch.cuda.synchronize()
    for i in range(n):
        assert torch.equal(a["W"][i].view(torch.int32), b["W"][i].view(torch.int32)), i
        assert torch.equal(a["idx"][i], b["idx"][i]), i
        assert torch.equal(a["Wh"][i].view(torch.int32), b["Wh"][i].view(torch.int32)), i
    assert torch.equal(a["ss"].view(torch.int32), b["ss"].view(torch.int32))
    assert bool(torch.isfinite(b["ss"]).all())


def test_static_workspace_entry_point_equals_the_allocating_one():
    g = torch.Generator(device=DEV).manual_seed(9)
    shapes = [(64, 64, 3, 3, 32)] * 3 + [(32, 16, 3, 3, 31), (64, 10, 32), (300_001, 32)]
    ps = [torch.rand(*s, device=DEV, generator=g) for s in shapes]
    qs = [torch.randn(s[-1], device=DEV, generator=g) * 0.2 for s in shapes]
    _run_both(ps, qs, 3.0, _lib.FLAG_EXACT)
    # second call, smaller input (fewer layers, fewer rows), fast arithmetic: partial sums of the first call
    # are still in the workspace and must not leak into these results
    ps2 = [ps[0][:7].contiguous(), ps[5][:1000].contiguous()]
    _run_both(ps2, [qs[0], qs[5]], 1.0, 0)


def test_static_workspace_entry_point_rejects_oversized_requests():
    L = _lib.lib()
    rows = 40_000_000  # 625 000 partial sums + the second-level area: more than the 2 MiB workspace
    segs = (_lib.AqSeg * 1)()
    p = torch.empty(16, device=DEV)  # never touched: the size check comes first
    segs[0].p_c, segs[0].q, segs[0].rows, segs[0].L = p.data_ptr(), p.data_ptr(), rows, 4
    segs[0].sumsq = p.data_ptr()
    assert L.aq_synth_workspace_bytes(segs, 1, 0) > _lib.STATIC_SCRATCH_BYTES
    rc = L.aq_synth_fwd_static(segs, 1, 1.0, 0, _lib.stream())
    assert rc < 0 and b"static workspace" in L.aq_last_error()
