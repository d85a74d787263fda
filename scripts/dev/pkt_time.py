"""Development aid: time of ops.pkt_kernel_matrix (normalise + product) at batch 1024, eager calls and as a replayed HIP
graph of ten calls (the eager call is host-bound: two ctypes launches and two allocations)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import quantized_cnn_amd.ops as ops
import bench
for D in (64, 128, 512, 100):
    f = torch.randn(1024, D, device="cuda")
    t = bench.event_time(lambda: ops.pkt_kernel_matrix(f), 20, 5)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        ops.pkt_kernel_matrix(f)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(10):
                k = ops.pkt_kernel_matrix(f)
    torch.cuda.current_stream().wait_stream(s)
    tg = bench.event_time(g.replay, 10, 3) / 10
    print("D=%d  eager %.2f us %.1f TF | in a graph %.2f us %.1f TF" % (D, t * 1e6, 2.0 * 1024 * 1024 * D / t / 1e12, tg * 1e6, 2.0 * 1024 * 1024 * D / tg / 1e12))
