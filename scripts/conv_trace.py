"""Development aid (experiment build with -DAQ_EXPERIMENT, AQ_HIP_LIB pointing at it): per-phase
timestamps of the conv kernels for three workgroups (first / middle / last of the grid)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import quantized_cnn_amd.ops as ops, quantized_cnn_amd._lib as L
lib = L.lib()
buf = (C.c_longlong * 96)()
def trace(tag, fn, stamps):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    lib.aq_debug_read_trace(buf)  # (clears nothing; stamps are overwritten by the next launch)
    fn(); torch.cuda.synchronize()
    assert lib.aq_debug_read_trace(buf) == 0
    t = list(buf)
    base = min(t[(w * 2 + th) * 16] for w in range(3) for th in range(2) if t[(w * 2 + th) * 16] > 0)
    print(tag)
    for w, wn in enumerate(("first", "middle", "last")):
        for th in range(2):
            row = t[(w * 2 + th) * 16:(w * 2 + th) * 16 + 16]
            if row[0] == 0: continue
            print("   wg %-6s thread %3d: " % (wn, th * 256) + "  ".join("%s=%.2f" % (nm, (row[i] - base) / 100.0) for i, nm in stamps if row[i] >= base))
N = 128
FW = [(0, "start"), (1, "fetch0 issued"), (2, "panel stored"), (3, "tile0 in LDS"), (4, "mfma done"), (5, "stored"), (6, "end")]
WG = [(0, "start"), (1, "prologue")] + [(2 + i, "ph%d" % i) for i in range(10)] + [(12, "loop end"), (13, "end")]
for (cin, cout, s, H) in [(16, 16, 1, 32), (32, 32, 1, 16), (64, 64, 1, 8)]:
    x = torch.randn(N, cin, H, H, device="cuda"); W = torch.randn(cout, cin, 3, 3, device="cuda") * 0.1
    y = ops.conv_forward(x, W, s); dy = torch.randn_like(y)
    trace("fwd %d->%d %dx%d (us since first stamp)" % (cin, cout, H, H), lambda: ops.conv_forward(x, W, s), FW)
    g_, b_ = torch.ones(cin, device="cuda"), torch.zeros(cin, device="cuda")
    trace("bn+fwd", lambda: ops.bnconv_forward(x, g_, b_, None, None, None, 1e-5, 0.1, W, s), FW + [(7, "bn merged")])
    trace("dgrad", lambda: ops.conv_dgrad(dy, W, x.shape, s), FW)
    trace("wgrad", lambda: ops.conv_wgrad(x, dy, s), WG)
