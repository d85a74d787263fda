"""HBM throughput of the ternary quantiser and its gradient kernel on 25 M .. 1 B weights (development aid)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import quantized_cnn_amd.ops as ops  # noqa: E402


def ev(fn, it=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e-3 / it


SIZES = [int(v) for v in sys.argv[1:]] or [25_000_000, 100_000_000, 250_000_000, 1_000_000_000]
for n in SIZES:
    w = torch.randn(n, device="cuda") * 0.05
    g = torch.randn(n, device="cuda")
    Wp, Wn = [torch.full((1,), 0.1, device="cuda")], [torch.full((1,), 0.08, device="cuda")]
    t_codes = ev(lambda: ops.tern_forward([w], "twn", decoded=False))                     # 2 reads of w + 1-byte codes
    t_full = ev(lambda: ops.tern_forward([w], "ttq", Wp=Wp, Wn=Wn, packed=True))          # + packed + decoded weights
    st = ops.tern_forward([w], "ttq", Wp=Wp, Wn=Wn)[0]["stats"]
    t_bwd = ev(lambda: ops.tern_backward([w], [g], "ttq", stats=[st], Wp=Wp, Wn=Wn))      # w, dwt read, dw written
    print("n=%11d  codes %.3f ms %.0f GB/s | codes+packed+Wt %.3f ms %.0f GB/s | ttq bwd %.3f ms %.0f GB/s" % (
        n, t_codes * 1e3, n * 9 / t_codes / 1e9, t_full * 1e3, n * 13.25 / t_full / 1e9, t_bwd * 1e3, n * 12 / t_bwd / 1e9), flush=True)
    del w, g
    torch.cuda.empty_cache()
