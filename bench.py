#!/usr/bin/env python3
"""Benchmark of the quantized-training hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...; without a
     launcher bench.py starts the N ranks itself, and refuses when fewer than N devices are visible)

One "step" = one optimisation step of the quantized ResNet-20 (adp_qtz_fan flavour, 32 learned
levels) on a synthetic CIFAR-10 shaped batch of 128 images per GPU: one synthesis launch for the 19
quantized layers, forward (MFMA convolutions, fused BatchNorm+ReLU kernels), CrossEntropy + model_loss,
backward (conv dgrad/wgrad, BatchNorm backward, one synthesis-gradient launch with the Adam update of
p_c fused in), Adam for the two small parameter groups.  Inputs are resident in HBM before the timed region.  Prints
one JSON line (rank 0).  Extra objects on that line:
  roofline      the quantize kernel (weight-synthesis forward) on the standalone sweep tensor
                (250 M primary coefficients, L=32): algorithmic bytes (132 B per weight row) / average
                launch duration measured here with HIP events on the launch stream; peak = 8 TB/s
  roofline_more the same for the synthesis gradient kernel (260 B/row) and, MFMA-bound, for the conv
                forward / dgrad / wgrad kernels of the five ResNet-20 layer shapes (fp32 matrix peak)
  cpu_baseline  the CPU port of the same training step (oracle/torch_port.py) timed on the host cores
                for a bounded sample (N=1 only)
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E peak (spec)
FP32_MATRIX_PEAK_TF = 157.3  # v_mfma_f32_* dense peak (spec)
BATCH = 128


def cfg():
    return argparse.Namespace(mode="train", prim_init="uniform", num_level_conv=32, num_level_fc=32, l2_reg=5e-4,
                              liar=True, prim_lr=1e-2, q_lr=1e-3, lr=1e-2)


def event_time(fn, iters, warm=3):
    """Average duration of fn() in seconds, HIP events on the stream the kernels are launched on."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e-3 / iters


def kernel_rooflines(quick):
    import quantized_cnn_amd.ops as ops
    from quantized_cnn_amd._lib import last_kernel  # the instantiation the most recent call launched (aq_last_kernel)
    dev = "cuda"
    L = 32
    rows = (250_000_000 if not quick else 25_000_000) // L
    g = torch.Generator(device=dev).manual_seed(0)
    p = torch.rand(rows, L, device=dev, generator=g)
    q = torch.tensor([-(2.0 ** -k) for k in range(2, 18)][::1] + [2.0 ** -k for k in range(17, 1, -1)], device=dev)
    g.manual_seed(1)
    dW = torch.randn(rows, device=dev, generator=g)
    W = [torch.empty(rows, device=dev)]
    dp = [torch.empty_like(p)]
    t1 = event_time(lambda: ops.synth_forward([p], [q], 4.0, soft=True, out_W=W), 20, warm=10)
    k1 = last_kernel()
    t2 = event_time(lambda: ops.synth_backward([p], [q], 4.0, [dW], 1e-3, out_dp=dp), 20, warm=10)
    k2 = last_kernel()
    # HBM bytes per launch from the PMC counters (collected separately with rocprofv3 --pmc on the same
    # 250 M-element tensor, see profiles/r03_roofline_traffic.json); only valid for the full-size tensor
    traffic = traffic2 = traffic3 = traffic_codes = traffic_tern = None
    tj = {}
    tf = os.path.join(ROOT, "profiles", "r03_roofline_traffic.json")
    if os.path.exists(tf) and not quick:
        try:
            tj = json.load(open(tf))
            if tj.get("rows_per_launch") == rows:
                traffic, traffic2 = tj.get("synth_fwd_bytes_per_launch"), tj.get("synth_bwd_bytes_per_launch")
                traffic3 = tj.get("synth_bwd_adam_bytes_per_launch")
                traffic_codes, traffic_tern = tj.get("synth_fwd_codes_only_bytes_per_launch"), tj.get("tern_quantise_bytes_per_call")
        except Exception:
            traffic = traffic2 = traffic3 = traffic_codes = traffic_tern = None
            tj = {}
    a1 = rows * (4 * L + 4) / t1 / 1e9
    # traffic: not measured in this run; the committed PMC result for the same launch (FETCH_SIZE x 2 + WRITE_SIZE)
    src = "profiles/r03_roofline_traffic.json (rocprofv3 --pmc passes of this launch, not collected in this run)" if traffic is not None else None
    roof = {"bound": "hbm", "kernel": k1 + " (soft weights, fast arithmetic)", "achieved": round(a1, 1), "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": round(a1 / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": src,
            "bytes_per_row": 4 * L + 4, "rows_per_launch": rows, "avg_launch_ms": round(t1 * 1e3, 4)}
    a2 = rows * (8 * L + 4) / t2 / 1e9
    more = [{"bound": "hbm", "kernel": k2 + " (gradient of the synthesis, dp_c and dq)", "achieved": round(a2, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
             "frac": round(a2 / HBM_PEAK_GBS, 4), "traffic": traffic2, "bytes_per_row": 8 * L + 4, "rows_per_launch": rows,
             "avg_launch_ms": round(t2 * 1e3, 4)}]
    # the code-emitting forward (hard mode, what an evaluation pass runs: 1-byte code and q[code] per row)
    th = event_time(lambda: ops.synth_forward([p], [q], 4.0, soft=False, codes=True), 20, warm=10)
    ah = rows * (4 * L + 5) / th / 1e9
    more.append({"bound": "hbm", "kernel": last_kernel() + " (codes + q[code]; arg-max of p, exact sequence on near ties)",
                 "achieved": round(ah, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ah / HBM_PEAK_GBS, 4), "traffic": traffic_codes,
                 "bytes_per_row": 4 * L + 5, "rows_per_launch": rows, "avg_launch_ms": round(th * 1e3, 4)})
    # soft W, codes and q[code] together in exact arithmetic (bit-reproducible against the CPU restatement)
    th = event_time(lambda: ops.synth_forward([p], [q], 4.0, soft=True, codes=True), 20, warm=10)
    ah = rows * (4 * L + 9) / th / 1e9
    more.append({"bound": "hbm", "kernel": last_kernel() + " (W + codes + q[code], exact arithmetic)", "achieved": round(ah, 1),
                 "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ah / HBM_PEAK_GBS, 4), "traffic": tj.get("synth_fwd_exact_W_codes_bytes_per_launch"), "bytes_per_row": 4 * L + 9, "rows_per_launch": rows,
                 "avg_launch_ms": round(th * 1e3, 4)})
    # 31 levels (adp_qtz_fix): 124-byte rows, no 16-byte alignment
    L31 = 31
    p31 = p.view(-1)[:rows * L31].view(rows, L31)
    q31 = torch.cat([q[:15], torch.zeros(1, device=dev), q[17:]]).contiguous()
    dp31 = [dp[0].view(-1)[:rows * L31].view(rows, L31)]
    # (twenty warm-up launches: this instruction-heavier kernel runs 10-25 % slower right after a pause)
    t31f = event_time(lambda: ops.synth_forward([p31], [q31], 4.0, soft=True, out_W=W), 20, warm=20)
    n31f = last_kernel()
    # adp_qtz_fix freezes its levels (adp_qtz_fix.py:649-652): its gradient pass asks for no dq; the learned-level form beside it
    t31b = event_time(lambda: ops.synth_backward([p31], [q31], 4.0, [dW], 1e-3, need_dq=[False], out_dp=dp31), 20, warm=20)
    n31b = last_kernel()
    t31q = event_time(lambda: ops.synth_backward([p31], [q31], 4.0, [dW], 1e-3, out_dp=dp31), 20, warm=20)
    n31q = last_kernel()
    for nm, tt, bpr, key in ((n31f + " (L = 31)", t31f, 4 * L31 + 4, "synth_fwd_L31"),
                             (n31b + " (L = 31, frozen levels as in adp_qtz_fix: no dq)", t31b, 8 * L31 + 4, "synth_bwd_L31_no_dq"),
                             (n31q + " (L = 31, with dq)", t31q, 8 * L31 + 4, "synth_bwd_L31_dq")):
        aa = rows * bpr / tt / 1e9
        more.append({"bound": "hbm", "kernel": nm, "achieved": round(aa, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(aa / HBM_PEAK_GBS, 4),
                     "traffic": tj.get(key + "_bytes_per_launch"), "bytes_per_row": bpr, "rows_per_launch": rows, "avg_launch_ms": round(tt * 1e3, 4)})
    del p31, dp31
    # the same kernel with the Adam update of p_c fused in: p, m, v read and written, 24L+4 bytes per row
    del dp
    m1, m2, stp = torch.zeros_like(p), torch.zeros_like(p), torch.zeros((), device=dev)
    t3 = event_time(lambda: ops.synth_backward_adam([p], [q], 4.0, [dW], [m1], [m2], [stp], 1e-2, wd=1e-3), 10)
    a3 = rows * (24 * L + 4) / t3 / 1e9
    more.append({"bound": "hbm", "kernel": last_kernel() + " (Adam update of p_c fused)", "achieved": round(a3, 1),
                 "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(a3 / HBM_PEAK_GBS, 4), "traffic": traffic3, "bytes_per_row": 24 * L + 4,
                 "rows_per_launch": rows, "avg_launch_ms": round(t3 * 1e3, 4)})
    del p, dW, W, m1, m2
    torch.cuda.empty_cache()
    for (cin, cout, s, H) in [(16, 16, 1, 32), (16, 32, 2, 32), (32, 32, 1, 16), (32, 64, 2, 16), (64, 64, 1, 8)]:
        x = torch.randn(BATCH, cin, H, H, device=dev)
        Wc = torch.randn(cout, cin, 3, 3, device=dev) * 0.1
        y = ops.conv_forward(x, Wc, s)
        dy = torch.randn_like(y)
        fl = 2.0 * BATCH * cout * cin * 9 * y.shape[2] * y.shape[3]
        # launch-bound kernels: time a graph of 20 back-to-back launches so host launch latency is excluded
        for name, fn in (("conv_fwd", lambda: ops.conv_forward(x, Wc, s)), ("conv_dgrad", lambda: ops.conv_dgrad(dy, Wc, x.shape, s)),
                         ("conv_wgrad", lambda: ops.conv_wgrad(x, dy, s))):
            st = torch.cuda.Stream()
            st.wait_stream(torch.cuda.current_stream())
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.stream(st):
                fn()
                torch.cuda.synchronize()
                with torch.cuda.graph(gr, stream=st):
                    for _ in range(20):
                        fn()
            torch.cuda.current_stream().wait_stream(st)
            t = event_time(gr.replay, 5, 2) / 20
            more.append({"bound": "mfma", "kernel": "%s %d->%d s%d %dx%d N=%d" % (name, cin, cout, s, H, H, BATCH),
                         "achieved": round(fl / t / 1e12, 2), "peak": FP32_MATRIX_PEAK_TF, "unit": "TFLOP/s",
                         "frac": round(fl / t / 1e12 / FP32_MATRIX_PEAK_TF, 4), "avg_launch_us": round(t * 1e6, 2)})
        # the weight gradients as the training step runs them: the layers of one shape together at the end of the backward
        # pass (conv_wgrad_partial_group + the batched slab reduction); per-layer average
        nl = {(16, 16): 6, (32, 32): 5, (64, 64): 5}.get((cin, cout), 1)
        if nl > 1:
            xs = [torch.randn_like(x) for _ in range(nl)]
            dys = [torch.randn_like(dy) for _ in range(nl)]
            outs = [torch.empty(cout, cin, 3, 3, device=dev) for _ in range(nl)]

            def grouped():
                res = ops.conv_wgrad_partial_group(xs, dys, s)
                ops.conv_wgrad_reduce([(ws, o, ns) for (ws, ns), o in zip(res, outs)])
            st = torch.cuda.Stream()
            st.wait_stream(torch.cuda.current_stream())
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.stream(st):
                grouped()
                torch.cuda.synchronize()
                with torch.cuda.graph(gr, stream=st):
                    for _ in range(10):
                        grouped()
            torch.cuda.current_stream().wait_stream(st)
            t = event_time(gr.replay, 5, 2) / (10 * nl)
            more.append({"bound": "mfma", "kernel": "conv_wgrad %d->%d s%d %dx%d N=%d, %d layers per launch as in the training step (per layer)" % (
                cin, cout, s, H, H, BATCH, nl), "achieved": round(fl / t / 1e12, 2), "peak": FP32_MATRIX_PEAK_TF, "unit": "TFLOP/s",
                "frac": round(fl / t / 1e12 / FP32_MATRIX_PEAK_TF, 4), "avg_launch_us": round(t * 1e6, 2)})
            del xs, dys, outs
    # ---- kernels of the quantiser modes and shapes that have no reference code (parity unpinned) ----
    torch.cuda.empty_cache()
    n = 250_000_000 if not quick else 25_000_000
    w = torch.randn(n, device=dev) * 0.05
    gw = torch.randn(n, device=dev)
    tq = event_time(lambda: ops.tern_forward([w], "twn", decoded=False), 10)
    more.append({"bound": "hbm", "kernel": "ternary quantiser, TWN: aq::tern_absstat_kernel + %s + second stages (threshold 0.7*mean|w|, 1-byte codes)" % last_kernel(),
                 "achieved": round(n * 9 / tq / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(n * 9 / tq / 1e9 / HBM_PEAK_GBS, 4),
                 "traffic": traffic_tern if not quick else None, "bytes_per_weight": 9, "weights_per_call": n, "avg_call_ms": round(tq * 1e3, 4), "parity": "unpinned (no reference implementation)"})
    Wp, Wn = [torch.full((1,), 0.1, device=dev)], [torch.full((1,), 0.08, device=dev)]
    st = ops.tern_forward([w], "ttq", Wp=Wp, Wn=Wn)[0]["stats"]
    tb = event_time(lambda: ops.tern_backward([w], [gw], "ttq", stats=[st], Wp=Wp, Wn=Wn), 10)
    more.append({"bound": "hbm", "kernel": last_kernel() + " (TTQ straight-through + scale gradients)", "achieved": round(n * 12 / tb / 1e9, 1),
                 "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(n * 12 / tb / 1e9 / HBM_PEAK_GBS, 4), "bytes_per_weight": 12, "weights_per_call": n,
                 "avg_call_ms": round(tb * 1e3, 4), "parity": "unpinned (no reference implementation)"})
    del w, gw
    torch.cuda.empty_cache()
    B18 = 32
    for (cin, cout, ks, s, pd, H) in [(3, 64, 7, 2, 3, 224), (64, 64, 3, 1, 1, 56), (128, 128, 3, 1, 1, 28), (256, 256, 3, 1, 1, 14), (512, 512, 3, 1, 1, 7),
                                      (128, 256, 3, 2, 1, 28), (64, 128, 1, 2, 0, 56)]:
        x = torch.randn(B18, cin, H, H, device=dev)
        Wc = torch.randn(cout, cin, ks, ks, device=dev) * 0.05
        y = ops.conv_forward(x, Wc, s, pad=pd)
        dy = torch.randn_like(y)
        fl = 2.0 * B18 * cout * cin * ks * ks * y.shape[2] * y.shape[3]
        for name, fn in (("conv_fwd", lambda: ops.conv_forward(x, Wc, s, pad=pd)), ("conv_dgrad", lambda: ops.conv_dgrad(dy, Wc, x.shape, s, pad=pd)),
                         ("conv_wgrad", lambda: ops.conv_wgrad(x, dy, s, ks=ks, pad=pd))):
            if name == "conv_dgrad" and cin == 3:
                continue  # the stem's input needs no gradient
            t = event_time(fn, 5, 2)
            more.append({"bound": "mfma", "kernel": "%s %d->%d k%d s%d %dx%d N=%d (ResNet-18 shape: stem, F(2x2,3x3), many-channel and 1x1 kernels of aq_conv_*.hip; no reference model)" % (
                name, cin, cout, ks, s, H, H, B18), "achieved": round(fl / t / 1e12, 2), "peak": FP32_MATRIX_PEAK_TF, "unit": "TFLOP/s",
                "frac": round(fl / t / 1e12 / FP32_MATRIX_PEAK_TF, 4), "avg_launch_us": round(t * 1e6, 2)})
    def graph_time(fn, reps=20):
        """Average duration of fn() inside a replayed HIP graph of `reps` calls (host launch cost excluded)."""
        st = torch.cuda.Stream()
        st.wait_stream(torch.cuda.current_stream())
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.stream(st):
            fn()
            torch.cuda.synchronize()
            with torch.cuda.graph(gr, stream=st):
                for _ in range(reps):
                    fn()
        torch.cuda.current_stream().wait_stream(st)
        return event_time(gr.replay, 5, 2) / reps

    for D in (64, 512):
        f = torch.randn(1024, D, device=dev)
        tch = torch.randn(1024, 512, device=dev)
        t = graph_time(lambda: ops.pkt_kernel_matrix(f))
        fl = 2.0 * 1024 * 1024 * D
        more.append({"bound": "mfma", "kernel": "PKT cosine-kernel matrix, batch 1024 x %d features (aq::pkt_gemm_nt_kernel<true>: row normalisation inside the product, "
                                                "upper triangle computed and mirrored)" % D,
                     "achieved": round(fl / t / 1e12, 2), "peak": FP32_MATRIX_PEAK_TF, "unit": "TFLOP/s", "frac": round(fl / t / 1e12 / FP32_MATRIX_PEAK_TF, 4),
                     "avg_call_us": round(t * 1e6, 2), "parity": "unpinned (no reference implementation)"})
        # the distillation loss itself (student 1024 x D against a 1024 x 512 teacher): forward alone, and forward + gradient
        # (two kernel matrices, row probabilities, divergence; the Jacobian matrix and one more product of 2 * N^2 * D flops)
        tl = graph_time(lambda: ops.pkt_loss(f, tch, need_grad=False))
        tg = graph_time(lambda: ops.pkt_loss(f, tch, need_grad=True))
        flf = 2.0 * 1024 * 1024 * (D + 512)
        flg = flf + 2.0 * 1024 * 1024 * D
        for nm, tt, ff in (("PKT loss forward", tl, flf), ("PKT loss forward + d loss / d student", tg, flg)):
            more.append({"bound": "mfma", "kernel": "%s, batch 1024, student %d / teacher 512 features (aq_pkt_loss: %d launches)" % (nm, D, 3 if tt is tl else 7),
                         "achieved": round(ff / tt / 1e12, 2), "peak": FP32_MATRIX_PEAK_TF, "unit": "TFLOP/s", "frac": round(ff / tt / 1e12 / FP32_MATRIX_PEAK_TF, 4),
                         "avg_call_us": round(tt * 1e6, 2), "parity": "unpinned (no reference implementation)"})
    return roof, more


def build_workload(name, rank, dp_factory, args):
    """-> dict(step, x, y, batch, workload, parity, model).  name:
         adaptive      the reference's model: ResNet-20 with adp_qtz_fan layers (32 learned levels, soft assignment), batch 128
         ttq_resnet20  BASELINE configs[1] read literally: the same graph with TTQ convolutions (learned Wp / Wn), batch 128
         twn_resnet20  the same with TWN convolutions
         inq_resnet18  BASELINE configs[2] read literally: ResNet-18 at 224 x 224, INQ 5-bit powers of two, half frozen, batch 32
    The last three have no reference implementation (parity unpinned)."""
    from quantized_cnn_amd.train import FusedTernaryStep, FusedTrainStep
    c = cfg()
    torch.manual_seed(0)
    g = torch.Generator(device="cuda").manual_seed(1000 + rank)
    if name == "adaptive":
        import quantized_cnn_amd.adp_qtz_fan as fan
        fan.beta = 1
        model = fan.ResNet(c, 3).cuda().train()
        dp = dp_factory(model)
        step = FusedTrainStep(model, c, fan, dp=dp, use_graph=not args.no_graph, force_exchange=args.force_exchange and dp is not None,
                              overlap_wgrad=args.overlap_wgrad)
        batch, size, classes = BATCH, 32, 10
        desc = ("quantized ResNet-20 as the reference implements it (adp_qtz_fan: 19 layers x 32 learned levels, soft assignment) on synthetic "
                "CIFAR-10 32x32, full step fwd+CE+model_loss+bwd+3xAdam (train.FusedTrainStep)")
        parity = "pinned to the reference (tests/golden)"
    elif name in ("ttq_resnet20", "twn_resnet20"):
        import quantized_cnn_amd.ternary as ternary
        c.mode_q = name[:3]
        c.lr = 1e-3  # (at the reference's 1e-2 the learned TTQ scales make the loss spike every few dozen steps, in every arrangement of the step)
        model = ternary.ResNet(c, 3).cuda().train()
        dp = dp_factory(model)
        step = FusedTernaryStep(model, c, dp=dp, use_graph=not args.no_graph, force_exchange=args.force_exchange and dp is not None)
        batch, size, classes = BATCH, 32, 10
        desc = ("BASELINE configs[1] read literally: ResNet-20 with %s convolutions (quantized_cnn_amd.ternary.ResNet%s; threshold 0.7 * mean|w|) on "
                "synthetic CIFAR-10 32x32, full step fwd+CE+l2+bwd+Adam (train.FusedTernaryStep)"
                % (name[:3].upper(), ", learned Wp / Wn" if name.startswith("ttq") else ""))
        parity = "unpinned (no reference implementation)"
    elif name == "inq_resnet18":
        from quantized_cnn_amd.resnet18 import ResNet18
        c.mode_q = "inq"
        c.lr = 1e-3
        model = ResNet18(c).cuda().train()
        for m in model.modules():
            if hasattr(m, "quantize_portion"):
                m.quantize_portion(0.5)
        dp = dp_factory(model)
        step = FusedTernaryStep(model, c, dp=dp, use_graph=not args.no_graph, force_exchange=args.force_exchange and dp is not None)
        batch, size, classes = 32, 224, 1000
        desc = ("BASELINE configs[2] read literally: ResNet-18 at 224x224 with INQ convolutions (5-bit powers of two, 50 % frozen) on synthetic "
                "ImageNet-shaped data, full step fwd+CE+l2+bwd+Adam (train.FusedTernaryStep), gradient exchange in 8 MiB buckets")
        parity = "unpinned (no reference model or INQ implementation)"
    else:
        raise SystemExit("bench.py: unknown workload %r" % name)
    x = torch.randn(batch, 3, size, size, device="cuda", generator=g)
    y = torch.randint(0, classes, (batch,), device="cuda", generator=g)
    return dict(step=step, x=x, y=y, batch=batch, workload=desc, parity=parity, model=model, dp=dp)


def time_workload(w, steps, warmup, dist=None):
    """W untimed steps, then exactly K timed steps between barriers + device synchronisation; -> (seconds, last loss)."""
    step, x, y = w["step"], w["x"], w["y"]
    step(x, y)
    bufs = step.static_inputs()
    if bufs is not None:  # the batch lives in the buffers the captured step reads (as a device-side loader leaves it)
        bufs[0].copy_(x)
        bufs[1].copy_(y)
        x, y = bufs
    for _ in range(warmup):
        step(x, y)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step(x, y)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    return time.perf_counter() - t0, float(step.loss)


def side_workload(name, steps, warmup, args):
    """One of the other workloads, timed on this GPU beside the headline (no exchange)."""
    w = build_workload(name, 0, lambda m: None, argparse.Namespace(no_graph=args.no_graph, force_exchange=False, overlap_wgrad=False))
    dt, loss = time_workload(w, steps, warmup)
    dt /= steps
    out = {"value": round(w["batch"] / dt, 1), "unit": "img/s", "ms_per_step": round(dt * 1e3, 4), "final_loss": round(loss, 5),
           "per_gpu_batch": w["batch"], "workload": w["workload"], "parity": w["parity"]}
    if name == "inq_resnet18":
        out["conv_tflops"] = round(3 * conv_flops_per_image(w["model"], 224) * w["batch"] / dt / 1e12, 1)
    del w
    torch.cuda.empty_cache()
    return out


def conv_flops_per_image(model, size):
    """Forward multiply-add flops (2 per MAC) of the model's convolutions for one size x size image, from the modules'
    own shapes (a forward hook on every convolution-like module records its output plane)."""
    total = [0]
    hooks = []

    def hook(m, inp, out):
        o = out[0] if isinstance(out, tuple) else out
        w = m.weight
        total[0] += 2 * w.shape[0] * w.shape[1] * w.shape[2] * w.shape[3] * o.shape[2] * o.shape[3]
    for m in model.modules():
        w = getattr(m, "weight", None)
        if w is not None and w.dim() == 4 and not isinstance(m, torch.nn.Conv2d):
            hooks.append(m.register_forward_hook(hook))
    c0 = model.conv0
    with torch.no_grad():
        was = model.training
        model.eval()
        model(torch.zeros(1, 3, size, size, device="cuda"))
        model.train(was)
    for h in hooks:
        h.remove()
    ho = (size + 2 * c0.padding[0] - c0.kernel_size[0]) // c0.stride[0] + 1
    return total[0] + 2 * c0.weight.numel() * ho * ho


def cpu_baseline(quick):
    """CPU port of the same step (oracle/torch_port.py, pinned to the reference by tests) on the host."""
    from oracle import torch_port as TP
    nthreads = min(16, os.cpu_count() or 1)
    torch.set_num_threads(nthreads)
    torch.manual_seed(0)
    out = {}
    for impl, warm, steps in (("conv2d", 5 if not quick else 1, 20 if not quick else 2), ("taps", 1, 2 if not quick else 1)):
        model = TP.PortModel("fan", 32, conv_impl=impl)
        model.train()
        opts = TP.optimizers(model)
        x, y = torch.randn(BATCH, 3, 32, 32), torch.randint(0, 10, (BATCH,))
        for _ in range(warm):
            TP.train_step(model, opts, x, y)
        t0 = time.perf_counter()
        for _ in range(steps):
            TP.train_step(model, opts, x, y)
        out[impl] = BATCH * steps / (time.perf_counter() - t0)
    res = {"value": round(out["conv2d"], 1), "unit": "img/s", "cores": nthreads, "kind": "port",
           "sample": "20 steps of the same workload (batch 128) after 5 warm-up steps, torch CPU fp32, F.conv2d formulation "
                     "(the reference's nine-tap gather formulation: 2 steps after 1)",
           "value_nine_tap_formulation": round(out["taps"], 1)}
    # the quantize kernels on the CPU (C restatement, OpenMP over rows), same algorithmic byte counts as the GPU figures
    import numpy as np
    from oracle import oracle as O
    O.set_num_threads(nthreads)
    # sizes of the stand-alone sweep (25 M ... 250 M primary coefficients; the GPU rows above use 250 M); L = 31 at the smallest
    sizes = (25_000_000, 100_000_000, 250_000_000) if not quick else (2_000_000,)
    sweep = []
    rng = np.random.default_rng(0)
    for elems in sizes:
        for L in ((32, 31) if elems == sizes[0] else (32,)):
            rows = elems // L
            p = rng.random((rows, L), dtype=np.float32)
            q = (O.levels_hint(0.3, 32) if L == 32 else O.levels_fix(0.3, 32)).astype(np.float32)
            dW = rng.standard_normal(rows, dtype=np.float32)
            O.synth_fwd(p[:1000], q, 4.0)
            t0 = time.perf_counter(); O.synth_fwd(p, q, 4.0, want_idx=False); tf = time.perf_counter() - t0
            t0 = time.perf_counter(); O.synth_bwd(p, q, 4.0, dW, 1e-3); tb = time.perf_counter() - t0
            sweep.append({"L": L, "elements": elems, "rows": rows, "synth_fwd_GBps": round(rows * (4 * L + 4) / tf / 1e9, 2),
                          "synth_bwd_GBps": round(rows * (8 * L + 4) / tb / 1e9, 2)})
            del p, dW
    res["quantize_kernels_cpu"] = {"cores": O.num_threads(), "sweep": sweep}
    return res


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start N ranks (one per GPU) through torch.distributed.run
    and hand on their exit code.  The parent never initialises the GPU (device_count() alone does not)."""
    import socket
    import subprocess
    n = torch.cuda.device_count()
    if n < args.gpus:
        raise SystemExit("bench.py: --gpus %d asked for but only %d HIP device(s) visible; not running a smaller job under that name"
                         % (args.gpus, n))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-graph", action="store_true", help="run the step eagerly instead of replaying HIP graphs")
    ap.add_argument("--quick", action="store_true", help="smaller roofline tensor and CPU sample (development)")
    ap.add_argument("--skip-extras", action="store_true", help="skip roofline and cpu_baseline measurements")
    ap.add_argument("--overlap-wgrad", action="store_true", help="weight-gradient groups on a second stream beside the rest of the backward pass")
    ap.add_argument("--workload", default="adaptive", choices=("adaptive", "ttq_resnet20", "twn_resnet20", "inq_resnet18"),
                    help="what the timed loop (value, ms_per_step, --gpus N) runs; the others are reported beside it on one GPU")
    ap.add_argument("--force-exchange", action="store_true",
                    help="run the multi-GPU form of the step (two HIP graphs around an RCCL all-reduce) even on one GPU")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return self_launch(args)  # before anything touches the GPU in this process
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit("bench.py: --gpus %d disagrees with WORLD_SIZE=%d of the launcher" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the hot path has no CPU fallback")
    torch.cuda.set_device(local)
    dist = None
    if world > 1 or args.force_exchange:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))

    from quantized_cnn_amd.train import DataParallel

    w = build_workload(args.workload, rank, (lambda m: DataParallel(m)) if dist is not None else (lambda m: None), args)
    dp = w["dp"]
    batch = w["batch"]
    elapsed, loss = time_workload(w, args.steps, args.warmup, dist)
    if dist is not None:
        t = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        out = {
            "metric": "train img/s (quantized ResNet-20, synthetic CIFAR-10) + quantize-kernel HBM GB/s vs peak",
            "value": round(world * batch * args.steps / elapsed, 1), "unit": "img/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": w["workload"], "workload_name": args.workload, "parity": w["parity"],
                       "global_batch": world * batch, "per_gpu_batch": batch, "parallelism": "dp%d" % world,
                       "hip_graphs": not args.no_graph,
                       "exchange": ("none" if dp is None else "rccl all-reduce of the flat gradient buffer" +
                                    (" in %d MiB buckets" % (w["step"].bucket_bytes >> 20) if hasattr(w["step"], "bucket_bytes") else ""))},
            "final_loss": round(loss, 5),
        }
        if not args.skip_extras:
            roof, more = kernel_rooflines(args.quick)
            out["roofline"] = roof
            out["roofline_more"] = more
            del w
            torch.cuda.empty_cache()
            # the other workloads on this GPU: second top-level values, each with its own workload text
            for name, k, wu in (("adaptive", 100, 20), ("ttq_resnet20", 100, 20), ("twn_resnet20", 100, 20), ("inq_resnet18", 20, 5)):
                if name != args.workload:
                    out[name] = side_workload(name, 20 if args.quick and k > 20 else (5 if args.quick else k), 2 if args.quick else wu, args)
            if world == 1:
                out["cpu_baseline"] = cpu_baseline(args.quick)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
