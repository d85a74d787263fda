"""Training step of the quantized ResNet: the loop body of the reference's `train()`
(adp_qtz_fan.py:280-297) as a reusable object, with optional HIP-graph capture of the whole step
and optional data parallelism (one process per GPU, RCCL through torch.distributed).

    step = TrainStep(model, config, module)        # module: adp_qtz_fan / _fix / _hint (owns `beta`)
    loss = step(x, y)                              # fwd + CE + model_loss + bwd + 3x Adam + zero_grad

Data parallelism (pure DP, parameters replicated, batch sharded):
  * d loss/d W of all quantized layers lives in one flat buffer (1.07 MB for ResNet-20) that is
    all-reduced ONCE inside the backward pass, before the synthesis-gradient kernel turns it into
    dp_c (34 MB) and dq on every rank.  dp_c/dq are linear in dW with a Jacobian that depends only on
    replicated parameters, so this equals all-reducing dp_c at 1/32 of the bytes.
  * gradients of the remaining (dense) parameters - conv0, BatchNorm, linear.bias - are
    all-reduced as a second small flat buffer after backward.
  * the regulariser's gradient is added locally on every rank (it does not depend on the batch).
  * BatchNorm statistics are per replica unless `sync_bn=True` converts the model to SyncBatchNorm.
"""
import torch
import torch.nn as nn

from . import layers


def split_parameters(model):
    """The reference's three optimizer groups, selected by parameter name (adp_qtz_fan.py:208-213)."""
    named = list(model.named_parameters())
    prim = [p for n, p in named if "p_c" in n]
    qlev = [p for n, p in named if "q_level" in n]
    rest = [p for n, p in named if "p_c" not in n and "q_level" not in n]
    return prim, qlev, rest


class OptimizerSet(tuple):
    """The optimizers of a model in the reference's step order (adp_qtz_fan.py:291-293), holding only the groups
    that have parameters, each with its role: `roles[i]` in ("prim", "q", "rest") names what `self[i]` updates, and
    `by_role("q")` returns that optimizer or None.  Code that needs the reference's three positions (checkpoints:
    optimizer_prim / optimizer_q / optimizer) goes by role, not by index."""

    ROLES = ("prim", "q", "rest")

    def __new__(cls, pairs):
        self = super().__new__(cls, [o for _, o in pairs])
        self.roles = tuple(r for r, _ in pairs)
        return self

    def by_role(self, role):
        return self[self.roles.index(role)] if role in self.roles else None


def make_optimizers(model, config, capturable=False, fused=False):
    """fused=True selects torch's single-kernel multi-tensor Adam (same update rule, one pass over
    the 34 MB of p_c instead of about ten)."""
    prim, qlev, rest = split_parameters(model)
    kw = dict(capturable=True) if capturable else {}
    if fused:
        kw["fused"] = True
    # q_level of the `fix` variant has requires_grad=False: Adam skips parameters without grad
    groups = (("prim", prim, getattr(config, "prim_lr", 1e-2)), ("q", qlev, getattr(config, "q_lr", 1e-3)),
              ("rest", rest, getattr(config, "lr", 1e-2)))
    # (models without adaptive layers, e.g. the ternary ones, have no p_c / q_level: only the groups that exist)
    return OptimizerSet([(role, torch.optim.Adam(g, lr=lr, **kw)) for role, g, lr in groups if g])


class DataParallel:
    """Gradient exchange for pure data parallelism over torch.distributed (backend "nccl" = RCCL on
    ROCm, "gloo" on CPU for tests).  Installs the flat-dW all-reduce hook of `layers`."""

    def __init__(self, model, process_group=None, sync_bn=False):
        import torch.distributed as dist
        self.dist, self.group = dist, process_group
        self.world = dist.get_world_size(process_group)
        self.model = model
        # gloo (tests, CPU rehearsals) moves device tensors through the host; RCCL works on them in place
        self.host_staged = dist.get_backend(process_group) == "gloo"
        if sync_bn and self.world > 1:
            nn.SyncBatchNorm.convert_sync_batchnorm(model, process_group)
        _, _, rest = split_parameters(model)
        self.dense = [p for p in rest if p.requires_grad]
        layers.set_dW_reduce_hook(self.reduce_flat if self.world > 1 else None)
        # identical initial parameters on every rank
        if self.world > 1:
            for t in list(model.parameters()) + list(model.buffers()):
                if self.host_staged and t.is_cuda:
                    h = t.data.cpu()
                    dist.broadcast(h, 0, group=process_group)
                    t.data.copy_(h)
                else:
                    dist.broadcast(t.data, 0, group=process_group)
            for m in model.modules():  # state derived from buffers (INQ layers: decoded frozen levels) follows the broadcast
                if hasattr(m, "buffers_changed"):
                    m.buffers_changed()

    def reduce_flat(self, flat, average=True):
        """In-place mean (average=False: sum, for gradients already scaled by 1/world) over ranks of one flat
        gradient buffer: a single all-reduce."""
        if self.host_staged:
            h = flat.cpu() if flat.is_cuda else flat
            self.dist.all_reduce(h, group=self.group)
            if h is not flat:
                flat.copy_(h)
        else:
            self.dist.all_reduce(flat, group=self.group)
        if average:
            flat.div_(self.world)

    def reduce_buckets(self, buckets):
        """In-place SUM over ranks of several pieces of a gradient buffer (views of one flat tensor): every piece an
        asynchronous all-reduce, issued back to back in the given order and awaited together, so that the
        collectives pipeline on the links instead of running one after the other behind host round trips."""
        if self.host_staged:
            hs = [b.cpu() if b.is_cuda else b for b in buckets]
            works = [self.dist.all_reduce(h, group=self.group, async_op=True) for h in hs]
            for w in works:
                w.wait()
            for b, h in zip(buckets, hs):
                if h is not b:
                    b.copy_(h)
            return
        works = [self.dist.all_reduce(b, group=self.group, async_op=True) for b in buckets]
        for w in works:
            w.wait()  # (makes the current stream wait for the collective; the host does not block)

    def reduce_dense(self):
        if self.world == 1:
            return
        grads = [p.grad for p in self.dense if p.grad is not None]
        if not grads:
            return
        flat = torch.cat([g.reshape(-1) for g in grads])
        self.reduce_flat(flat)
        off = 0
        for g in grads:
            n = g.numel()
            g.copy_(flat[off:off + n].view_as(g))
            off += n

    def close(self):
        layers.set_dW_reduce_hook(None)


class TrainStep:
    """One optimisation step with the reference's order of operations: forward, CrossEntropy +
    model_loss, backward, optimizer_prim / optimizer_q / optimizer .step(), three zero_grad()."""

    def __init__(self, model, config, module, dp=None, use_graph=False):
        self.model, self.config, self.module, self.dp = model, config, module, dp
        self.criterion = nn.CrossEntropyLoss()
        self.use_graph = use_graph
        self.opts = make_optimizers(model, config, capturable=use_graph)
        self._graph = None
        self._key = None

    # -- eager ---------------------------------------------------------------------------------
    def _body(self, x, y):
        logits = self.model.forward(x)
        loss = self.criterion(logits, y) + self.module.model_loss(self.config, self.module.beta, self.model)
        loss.backward()
        if self.dp is not None:
            self.dp.reduce_dense()
        for o in self.opts:
            o.step()
        return loss, logits

    def __call__(self, x, y):
        if not self.use_graph:
            loss, logits = self._body(x, y)
            for o in self.opts:
                o.zero_grad()
            self.logits = logits.detach()
            return loss.detach()
        return self._graph_step(x, y)

    # -- whole step captured in one HIP graph ----------------------------------------------------
    def _graph_step(self, x, y):
        key = (tuple(x.shape), float(self.module.beta), tuple(g["lr"] for o in self.opts for g in o.param_groups))
        if self._graph is None or key != self._key:
            self._capture(x, y)
            self._key = key
        self._sx.copy_(x)
        self._sy.copy_(y)
        self._graph.replay()
        self.logits = self._slogits
        return self._sloss

    def _capture(self, x, y):
        """Parameters and optimizer state are updated by the warm-up iterations, exactly as eager
        steps would; afterwards every replay is one more step."""
        _refuse_sync_bn_under_capture(self.model, self.dp)
        if self.dp is not None and self.dp.world > 1:
            raise RuntimeError("graph capture of the data-parallel step is not supported; use use_graph=False")
        self._sx, self._sy = x.clone(), y.clone()
        self._graph = None
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for o in self.opts:
                o.zero_grad(set_to_none=True)
            g = torch.cuda.CUDAGraph()
            # capture without a warm-up step so that step counts stay those of the reference loop;
            # Adam state is created lazily on the first step, so initialise it first (no update)
            for o in self.opts:
                for group in o.param_groups:
                    for p in group["params"]:
                        if p.requires_grad and len(o.state[p]) == 0:
                            o.state[p]["step"] = torch.zeros((), dtype=torch.float32, device=p.device)
                            o.state[p]["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                            o.state[p]["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=s):
                loss, logits = self._body(self._sx, self._sy)
                self._sloss, self._slogits = loss.detach(), logits.detach()
                for o in self.opts:
                    o.zero_grad(set_to_none=True)
        torch.cuda.current_stream().wait_stream(s)
        self._graph = g


class CapturedStep:
    """A training step of any model on torch autograd with torch optimizers - forward, loss_fn(model(x), y), backward,
    optimizer.step() - captured once and replayed as ONE HIP graph (the library's kernels are launched on torch's current
    stream, so they are captured like stock operators).  For the models outside the reference's three-optimizer loop
    (ternary.ResNet, resnet18.ResNet18, single layers); the reference's own step order is TrainStep / FusedTrainStep.
    `warmup` (>= 1) eager steps on the first batch run first (they are real steps: optimizer state is created there), then
    the capture; every call
    copies (x, y) into the graph's input buffers and replays.  Shapes must stay fixed; SyncBatchNorm is refused."""

    def __init__(self, model, optimizers, loss_fn, warmup=2):
        if warmup < 1:  # optimizers create their state (momentum, Adam's averages) in their first step: a capture of THAT
            # step would replay the state's creation, not its update
            raise ValueError("CapturedStep: at least one eager warm-up step is needed before the capture")
        self.model, self.loss_fn, self.warmup = model, loss_fn, warmup
        self.opts = list(optimizers) if isinstance(optimizers, (list, tuple)) else [optimizers]
        self._graph = None

    def _body(self):
        out = self.model(self._sx)
        loss = self.loss_fn(out, self._sy)
        loss.backward()
        for o in self.opts:
            o.step()
        self.loss, self.output = loss.detach(), out.detach()

    def _capture(self, x, y):
        if any(isinstance(m, nn.SyncBatchNorm) for m in self.model.modules()):
            raise RuntimeError("CapturedStep: the model holds SyncBatchNorm modules; their collectives cannot be captured in a HIP graph")
        self._sx, self._sy = x.clone(), y.clone()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(self.warmup):
                for o in self.opts:
                    o.zero_grad(set_to_none=True)
                self._body()
        torch.cuda.current_stream().wait_stream(s)
        for o in self.opts:
            o.zero_grad(set_to_none=True)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self._body()
        self._graph = g

    def __call__(self, x, y):
        if self._graph is None:
            self._capture(x, y)  # (the capture itself does not execute: the replay below is this call's step)
        elif tuple(x.shape) != tuple(self._sx.shape) or tuple(y.shape) != tuple(self._sy.shape):
            raise ValueError("CapturedStep: batch shape changed (%s -> %s); build a new step" % (tuple(self._sx.shape), tuple(x.shape)))
        self._sx.copy_(x)
        self._sy.copy_(y)
        self._graph.replay()
        return self.loss


def _refuse_sync_bn_under_capture(model, dp):
    """SyncBatchNorm runs torch.distributed collectives inside its forward and backward; captured into a HIP
    graph they would be replayed outside the process group's ordering.  Such a model runs eagerly."""
    if dp is not None and dp.world > 1 and any(isinstance(m, nn.SyncBatchNorm) for m in model.modules()):
        raise RuntimeError("the model holds SyncBatchNorm modules (DataParallel(sync_bn=True)): their collectives cannot be "
                           "captured in a HIP graph; build the step with use_graph=False")


def beta_schedule(config, epoch, beta, opts):
    """Per-epoch schedule of the reference (adp_qtz_fan.py:269-278), to be called at the top of every epoch: beta
    steps to beta_list[k] at epoch_list[k] (k = 1, 2, 3), learning rates of the three optimizers are divided by 5
    at epoch_list[4].  One chain of alternatives, as in the reference: an epoch listed twice takes the first
    match only.  Returns the new beta."""
    beta_list = list(map(int, config.beta_list.strip("[]").split(",")))
    epoch_list = list(map(int, config.epoch_list.strip("[]").split(",")))
    if epoch == epoch_list[1]:
        beta = beta_list[1]
    elif epoch == epoch_list[2]:
        beta = beta_list[2]
    elif epoch == epoch_list[3]:
        beta = beta_list[3]
    elif epoch == epoch_list[4]:
        for o in opts:
            o.param_groups[0]["lr"] /= 5
    return beta


class FusedTrainStep:
    """Same arithmetic as TrainStep, arranged for HIP graphs and data parallelism:

        graph A : ONE synthesis launch (W of all layers + sum W^2); forward; CrossEntropy; backward
                  with the synthesised W as leaves -> d loss/d W of every quantized layer lands in one
                  flat buffer; for the exchange the dense gradients are gathered behind them
        comm    : one all-reduce of that buffer (only when world_size > 1; never captured)
        graph B : ONE synthesis-gradient launch (dW + 2*l2_reg*W -> dp_c, dq) with Adam for p_c inside;
                  Adam of the two small groups, one launch each, with the regulariser gradient of the
                  dense "weight" tensors folded in

    The graphs are captured once per (input shape, beta, learning rates) and replayed (a single graph
    holding both phases when there is no exchange); with use_graph=False the same two phases run eagerly.
    """

    def __init__(self, model, config, module, dp=None, use_graph=True, fuse_adam=True, overlap_wgrad=False, force_exchange=False):
        """force_exchange: keep the exchange (the flat-buffer all-reduce between the two graphs) even when the
        process group has a single rank, where it changes no value: the path a multi-GPU job takes, runnable
        on one GPU.
        fuse_adam: apply optimizer_prim's Adam update of p_c inside the synthesis-gradient kernel (the
        optimizer object keeps its state tensors and state_dict format; its step() is not called).
        overlap_wgrad: run the weight-gradient launches (one per group of layers of a shape, started when the backward
        pass leaves their residual stage) on a second HIP stream beside the rest of the backward pass (a parallel
        branch of the captured graph).  Off by default: measured slower on MI355X / ROCm 7.2 than the single-stream
        chain that runs all groups after the pass (108.2k against 110.2k img/s in one session; with one
        cross-stream edge per layer, before the layers were grouped, 77.0k against 82.2k): the input-gradient
        kernels beside them already occupy every CU, so the groups take their time from the chain."""
        self.overlap_wgrad = bool(overlap_wgrad)
        self._side = torch.cuda.Stream() if self.overlap_wgrad else None
        self.model, self.config, self.module, self.dp = model, config, module, dp
        if force_exchange and dp is None:
            raise ValueError("force_exchange needs a DataParallel object (dp=...) to run the exchange with")
        self.force_exchange = bool(force_exchange)
        self.use_graph = use_graph
        self.criterion = nn.CrossEntropyLoss()
        self.opts = make_optimizers(model, config, capturable=use_graph, fused=True)
        self.qlayers = model.quant_layers()
        dev = next(model.parameters()).device
        named = list(model.named_parameters())
        self.dense = [p for n, p in named if "p_c" not in n and "q_level" not in n and p.requires_grad]
        self.dense_reg = [p for n, p in named if "weight" in n and "p_c" not in n and "q_level" not in n]
        self.ps = [l.p_c for l in self.qlayers]
        self.qs = [l.q_level for l in self.qlayers]
        self.sizes = [p.numel() // p.shape[-1] for p in self.ps]
        nW = sum(self.sizes)
        nD = sum(p.numel() for p in self.dense)
        self.W_flat = torch.empty(nW, dtype=torch.float32, device=dev)
        self.g_flat = torch.zeros(nW + nD, dtype=torch.float32, device=dev)  # [dW of all layers | dense grads]
        self.Ws, self.slots, off = [], [], 0
        for p, n in zip(self.ps, self.sizes):
            self.Ws.append(self.W_flat[off:off + n].view(p.shape[:-1]))
            self.slots.append(self.g_flat[off:off + n].view(p.shape[:-1]))
            off += n
        self.nW = nW
        g0 = self.opts[0].param_groups[0]
        self.fuse_adam = bool(fuse_adam) and len(self.ps) <= 24 and not (g0["amsgrad"] or g0["maximize"] or g0["weight_decay"])
        # the two small groups (q_level vectors; conv0, BatchNorm, linear bias): one launch each instead of the
        # stock optimizer's multi-tensor kernels, with the regulariser's gradient folded in
        self.small_adam = self.fuse_adam and all(not (g["amsgrad"] or g["maximize"] or g["weight_decay"])
                                                 for o in self.opts[1:] for g in o.param_groups)
        self.dp_out = None if self.fuse_adam else [torch.zeros_like(p) for p in self.ps]
        self.dq_out = [torch.zeros_like(q) if q.requires_grad else None for q in self.qs]
        self.dss = torch.full((len(self.ps),), float(config.l2_reg), dtype=torch.float32, device=dev)
        # the stem convolution's weight gradient joins the batched slab reduction through a buffer of its own
        c0 = getattr(model, "conv0", None)
        self.stem_slot = self.stem_w = None
        if isinstance(c0, nn.Conv2d) and c0.kernel_size == (3, 3) and c0.padding == (1, 1) and c0.stride in ((1, 1), (2, 2)) and c0.bias is None \
                and c0.weight.requires_grad:
            self.stem_w = c0.weight
            self.stem_slot = torch.zeros_like(c0.weight)
        self._graphs = None
        self._key = None
        self.loss = self.reg = self.logits = None

    @property
    def _exchange(self):
        return self.dp is not None and (self.dp.world > 1 or self.force_exchange)

    # ---- the two phases ---------------------------------------------------------------------
    def _phase_a(self, x, y):
        from . import ops
        beta = float(self.module.beta)
        for p in self.dense:
            p.grad = None  # autograd then assigns instead of launching one accumulate kernel per tensor
        # the stem's weight gradient lands in THIS step's buffer (several step objects may share one model)
        self.model._stem_slot = self.stem_slot
        out = ops.synth_forward([p.detach() for p in self.ps], [q.detach() for q in self.qs], beta, soft=True, sumsq=True,
                                out_W=self.Ws)
        leaves = []
        for l, W, slot in zip(self.qlayers, self.Ws, self.slots):
            w = W.detach().requires_grad_(True)
            leaves.append(w)
            l._shared = (w, slot)
        self.model._synth_cache = None
        # Weight-gradient reductions are deferred from the forward pass on (the stem looks at the switch when
        # it runs), to one launch after the backward pass.
        jobs = []
        prev = layers.defer_wgrad_reductions(jobs, self._side if self.overlap_wgrad else None)
        try:
            logits = self.model.forward_body(x)
            # loss tail in two launches: sums of squares of the dense "weight" tensors, then cross entropy, its
            # gradient and the loss value (the regulariser's value enters here, its gradients in phase B:
            # 2*l2_reg*w is added to the batch gradient after the exchange, being the same on every rank)
            dense_ss = ops.sumsq([p.detach() for p in self.dense_reg]) if self.dense_reg else None
            # data parallel: the loss gradient leaves scaled by 1/world, so that the exchange is a plain sum
            vals, dlogits = ops.ce_loss(logits.detach().contiguous(), y, float(self.config.l2_reg), out["sumsq"], dense_ss,
                                        grad_scale=1.0 / self.dp.world if self._exchange else 1.0)
            logits.backward(dlogits)
        finally:
            layers.defer_wgrad_reductions(*prev)
        layers.run_deferred_wgrads(jobs)  # first-stage kernels that waited for the end of the pass, one launch per layer shape
        if self.overlap_wgrad:
            torch.cuda.current_stream().wait_stream(self._side)  # join: all weight-gradient slabs written
        ops.conv_wgrad_reduce(jobs)  # the slab sums of all quantized convolutions in one launch
        del jobs
        if self.stem_slot is not None:
            self.stem_w.grad = self.stem_slot
        for w in leaves:
            w.grad = None
        if self._exchange:
            # dense gradients behind dW in the flat exchange buffer (one kernel)
            torch.cat([p.grad.reshape(-1) for p in self.dense], out=self.g_flat[self.nW:])
        self.reg, self.loss = vals[1], vals[2]
        self.logits = logits.detach()

    def _phase_b(self):
        from . import ops
        need_dq = [q.requires_grad for q in self.qs]
        if self._exchange:
            # dense gradients come back from the exchange buffer (averaged over ranks)
            off = self.nW
            views = []
            for p in self.dense:
                views.append(self.g_flat[off:off + p.numel()].view_as(p))
                off += p.numel()
            if self.small_adam:
                for p, v in zip(self.dense, views):  # the update reads (and completes) the gradients where they are
                    p.grad = v
            else:
                torch._foreach_copy_([p.grad for p in self.dense], views)
        if self.dense_reg and not self.small_adam:
            torch._foreach_add_([p.grad for p in self.dense_reg], [p.detach() for p in self.dense_reg], alpha=2.0 * float(self.config.l2_reg))
        ps, qs = [p.detach() for p in self.ps], [q.detach() for q in self.qs]
        if self.fuse_adam:
            self._ensure_adam_state()
            grp = self.opts[0].param_groups[0]
            st = [self.opts[0].state[p] for p in self.ps]
            ops.synth_backward_adam(ps, qs, float(self.module.beta), self.slots, [t["exp_avg"] for t in st], [t["exp_avg_sq"] for t in st],
                                    [t["step"] for t in st], lr=grp["lr"], betas=grp["betas"], eps=grp["eps"], need_dq=need_dq,
                                    out_dq=self.dq_out, dsumsq=self.dss)
        else:
            ops.synth_backward(ps, qs, float(self.module.beta), self.slots, 0.0, need_dq=need_dq, out_dp=self.dp_out, out_dq=self.dq_out,
                               dsumsq=self.dss)
            for p, g in zip(self.ps, self.dp_out):
                p.grad = g
        for q, g in zip(self.qs, self.dq_out):
            if g is not None:
                q.grad = g
        if self.small_adam:
            reg_ids, coef = {id(p) for p in self.dense_reg}, 2.0 * float(self.config.l2_reg)
            for o in self.opts[1:]:
                for grp in o.param_groups:
                    prm = [p for p in grp["params"] if p.grad is not None]
                    st = [o.state[p] for p in prm]
                    ops.adam_step([p.detach() for p in prm], [p.grad for p in prm], [t["exp_avg"] for t in st], [t["exp_avg_sq"] for t in st],
                                  [t["step"] for t in st], lr=grp["lr"], betas=grp["betas"], eps=grp["eps"],
                                  regs=[coef if id(p) in reg_ids else 0.0 for p in prm])
            return
        for o in (self.opts[1:] if self.fuse_adam else self.opts):
            o.step()

    def _comm(self):
        if self._exchange:
            self.dp.reduce_flat(self.g_flat, average=False)

    # ---- driver ------------------------------------------------------------------------------
    def __call__(self, x, y):
        if not self.use_graph:
            self._phase_a(x, y)
            self._comm()
            self._phase_b()
            return self.loss
        key = (tuple(x.shape), float(self.module.beta), tuple(g["lr"] for o in self.opts for g in o.param_groups), self._exchange)
        if self._graphs is None or key != self._key:
            self._capture(x, y)
            self._key = key
        if x is not self._sx:
            self._sx.copy_(x)
        if y is not self._sy:
            self._sy.copy_(y)
        self._graphs[0].replay()
        if self._graphs[1] is not None:
            self._comm()
            self._graphs[1].replay()
        return self.loss

    def invalidate(self):
        """Drop the captured graphs; the next call captures again.  Needed after anything that replaces tensors
        the graphs hold addresses of (parameters or optimizer state re-created, not copied into)."""
        self._graphs = None
        self._key = None

    def static_inputs(self):
        """(x, y) buffers the captured graphs read, or None before the first graph step.  A producer that
        writes a batch straight into them (and passes these very tensors to the step) saves the two copies
        per step."""
        return (self._sx, self._sy) if (self.use_graph and self._graphs is not None) else None

    def _ensure_adam_state(self):
        """Adam creates its state lazily in step(); create it up front (same tensors, same state_dict
        layout) for the graph capture and for the fused p_c update."""
        for o in self.opts:
            for group in o.param_groups:
                for p in group["params"]:
                    if p.requires_grad and len(o.state[p]) == 0:
                        o.state[p]["step"] = torch.zeros((), dtype=torch.float32, device=p.device)
                        o.state[p]["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                        o.state[p]["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)

    def _capture(self, x, y):
        _refuse_sync_bn_under_capture(self.model, self.dp)
        self._sx, self._sy = x.clone(), y.clone()
        self._ensure_adam_state()
        for q, g in zip(self.qs, self.dq_out):
            if g is not None:
                q.grad = g
        if self.dp_out is not None:
            for p, g in zip(self.ps, self.dp_out):
                p.grad = g
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        torch.cuda.synchronize()
        ga, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with torch.cuda.stream(s):
            # one eager pass of phase A so that libraries used by the stock ops (BatchNorm, pooling)
            # finish their lazy initialisation outside the capture; it changes no parameter, and the
            # BatchNorm running statistics it touches are put back
            bufs = [b for b in self.model.buffers()]
            saved = [b.clone() for b in bufs]
            self._phase_a(self._sx, self._sy)
            for b, v in zip(bufs, saved):
                b.copy_(v)
            torch.cuda.synchronize()
            if self._exchange:
                with torch.cuda.graph(ga, stream=s):
                    self._phase_a(self._sx, self._sy)
                with torch.cuda.graph(gb, stream=s, pool=ga.pool()):
                    self._phase_b()
            else:  # nothing happens between the phases: one graph, one replay per step
                gb = None
                with torch.cuda.graph(ga, stream=s):
                    self._phase_a(self._sx, self._sy)
                    self._phase_b()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        self._graphs = (ga, gb)
        # capture does not execute: parameters are untouched until the first replay


class FusedTernaryStep:
    """FusedTrainStep's arrangement for the models whose layers are ternary (TWN / TTQ) or incremental power-of-two
    (INQ) quantised - ternary.ResNet, resnet18.ResNet18 in those modes (the quantisers the reference's README cites;
    no reference implementation, parity unpinned):

        graph A : ONE quantiser call for every layer (aq_tern_fwd / aq_inq_effective) -> quantised weights in one flat
                  buffer; forward; CrossEntropy + l2 on the "weight" tensors (adp_qtz_fan.py:737-743); backward with the
                  quantised weights as leaves -> d loss / d w~ of every layer in one flat buffer, the weight gradients
                  of layers of one shape as one launch each; the dense gradients gathered behind them for the exchange
        comm    : all-reduce of that buffer in buckets (only when world_size > 1; never captured), see `bucket_bytes`
        graph B : the quantiser's gradient rule (straight-through scaling / masks, d loss / d Wp, d loss / d Wn as masked
                  sums) with Adam of the latent weights and of the scales in the same pass (aq_tern_bwd_adam); Adam of
                  the remaining parameters in one call (aq_adam_step), regulariser gradient folded in

    Data parallel: d loss / d Wp, d loss / d Wn and the latent gradients are linear in d loss / d w~ with coefficients
    that depend only on replicated parameters, so reducing d loss / d w~ and applying the rule on every rank equals
    reducing the rule's outputs; INQ masks the REDUCED gradient on every rank (mask and frozen levels are replicated
    state), so replicas stay bit-identical: every rank applies the same arithmetic to the same reduced buffer.

    bucket_bytes: the exchange buffer [d w~ of all layers | dense gradients] is all-reduced in consecutive pieces of
    about this many bytes, issued back to back as asynchronous collectives and awaited together.  Default 8 MiB: over
    xGMI a ring all-reduce is bound per link (~153 GB/s), pieces of 4-16 MiB reach the link rate while several in flight
    keep all seven links busy; ResNet-20 (1.1 MB) is a single piece, ResNet-18 (45 MB) six."""

    def __init__(self, model, config, module=None, dp=None, use_graph=True, force_exchange=False, bucket_bytes=8 << 20, keep_grads=False):
        """keep_grads: also store what autograd would have left in .grad of the latent weights and of the TTQ scales (the
        fused update does not need them in memory; tests and gradient statistics do)."""
        from . import ternary
        self.model, self.config, self.dp = model, config, dp
        self.module = module if module is not None else ternary
        if force_exchange and dp is None:
            raise ValueError("force_exchange needs a DataParallel object (dp=...) to run the exchange with")
        self.force_exchange, self.use_graph, self.bucket_bytes = bool(force_exchange), bool(use_graph), int(bucket_bytes)
        if not hasattr(model, "forward_body"):
            raise TypeError("FusedTernaryStep: the model needs forward_body(x) (ternary.ResNet, resnet18.ResNet18)")
        self.tern = [m for m in model.modules() if isinstance(m, ternary._TernBase)]
        self.inq = [m for m in model.modules() if isinstance(m, ternary.INQConv2d)]
        if bool(self.tern) == bool(self.inq):
            raise ValueError("FusedTernaryStep: the model must hold ternary layers or INQ layers (found %d / %d)" % (len(self.tern), len(self.inq)))
        if any(isinstance(m, layers._QuantBase) for m in model.modules()):
            raise ValueError("FusedTernaryStep: adaptive (p_c / q_level) layers belong to FusedTrainStep")
        self.qlayers = self.tern or self.inq
        if self.tern:
            keys = {(l.mode_q, l.factor, l.threshold_from_max) for l in self.tern}
            if len(keys) != 1:
                raise ValueError("FusedTernaryStep: all ternary layers must share mode, factor and threshold rule (found %s)" % sorted(keys))
            self.mode, self.factor, self.from_max = next(iter(keys))
        else:
            self.mode = "inq"
        dev = next(model.parameters()).device
        self.ws = [l.weight for l in self.qlayers]
        self.scales = [(l.Wp, l.Wn) for l in self.qlayers] if self.mode == "ttq" else []
        own = {id(p) for p in self.ws} | {id(p) for pair in self.scales for p in pair}
        named = list(model.named_parameters())
        self.dense = [p for n, p in named if id(p) not in own and p.requires_grad]
        self.reg_all = [p for n, p in named if "weight" in n]                       # the regulariser's tensors (value)
        self.dense_reg = {id(p) for n, p in named if "weight" in n and id(p) not in own}
        # INQ: the pass that builds the effective weights also sums the squares of the latent tensors it reads
        wid = {id(w) for w in self.ws}
        self.reg_q = self.mode == "inq" and all(id(w) in {id(p) for p in self.reg_all} for w in self.ws)
        self.reg_rest = [p for p in self.reg_all if id(p) not in wid]
        self.opts = make_optimizers(model, config, capturable=use_graph, fused=True)
        if self.opts.roles != ("rest",):
            raise ValueError("FusedTernaryStep: expected one optimizer over all parameters, found roles %s" % (self.opts.roles,))
        g0 = self.opts[0].param_groups[0]
        if g0["amsgrad"] or g0["maximize"] or g0["weight_decay"]:
            raise ValueError("FusedTernaryStep: plain Adam only (no amsgrad / maximize / weight_decay)")
        nW = sum(w.numel() for w in self.ws)
        nD = sum(p.numel() for p in self.dense)
        self.nW = nW
        self.W_flat = torch.empty(nW, dtype=torch.float32, device=dev)
        self.g_flat = torch.zeros(nW + nD, dtype=torch.float32, device=dev)  # [d loss / d w~ of all layers | dense grads]
        self.Wts, self.slots, off = [], [], 0
        for w in self.ws:
            n = w.numel()
            self.Wts.append(self.W_flat[off:off + n].view(w.shape))
            self.slots.append(self.g_flat[off:off + n].view(w.shape))
            off += n
        c0 = getattr(model, "conv0", None)
        self.stem_slot = self.stem_w = None
        if isinstance(c0, nn.Conv2d) and c0.bias is None and c0.weight.requires_grad and c0.kernel_size == (3, 3) and c0.padding == (1, 1) \
                and c0.stride in ((1, 1), (2, 2)):
            self.stem_w, self.stem_slot = c0.weight, torch.zeros_like(c0.weight)
        self.keep_grads = bool(keep_grads)
        self.dW_out = [torch.zeros_like(w) for w in self.ws] if keep_grads else None
        self.dscale_out = [(torch.zeros_like(p), torch.zeros_like(n)) for p, n in self.scales] if keep_grads else None
        self._graphs = self._key = None
        self.loss = self.reg = self.logits = None

    @property
    def _exchange(self):
        return self.dp is not None and (self.dp.world > 1 or self.force_exchange)

    def _ensure_adam_state(self):
        o = self.opts[0]
        for group in o.param_groups:
            for p in group["params"]:
                if p.requires_grad and len(o.state[p]) == 0:
                    o.state[p]["step"] = torch.zeros((), dtype=torch.float32, device=p.device)
                    o.state[p]["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    o.state[p]["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)

    # ---- the two phases ---------------------------------------------------------------------
    def _phase_a(self, x, y):
        from . import ops
        for p in self.dense:
            p.grad = None
        self.model._stem_slot = self.stem_slot
        ws = [w.detach() for w in self.ws]
        if self.mode == "inq":
            fz = [l._frozen_values() for l in self.qlayers]  # persistent tensors, rewritten in place by every INQ stage
            # (the same pass leaves sum w^2 of every latent tensor for the regulariser: aq_sumsq then reads the other tensors only)
            _, ss_q = ops.inq_effective(ws, [l.mask for l in self.qlayers], [f[1] for f in fz], out=self.Wts, sumsq=bool(self.reg_q))
            for l, W, slot in zip(self.qlayers, self.Wts, self.slots):
                l._shared = (W.detach().requires_grad_(True), slot)
        else:
            kw = dict(Wp=[p.detach() for p, _ in self.scales], Wn=[n.detach() for _, n in self.scales]) if self.mode == "ttq" else {}
            self.q = ops.tern_forward(ws, self.mode, self.factor, self.from_max, out_Wt=self.Wts, **kw)
            for l, q, W, slot in zip(self.qlayers, self.q, self.Wts, self.slots):
                l._shared = (q, W.detach().requires_grad_(True), slot)
        jobs = []
        prev = layers.defer_wgrad_reductions(jobs, None)
        try:
            logits = self.model.forward_body(x)
            if self.mode == "inq" and self.reg_q:
                ss = ops.sumsq([p.detach() for p in self.reg_rest]) if self.reg_rest else None
            else:
                ss, ss_q = (ops.sumsq([p.detach() for p in self.reg_all]) if self.reg_all else None), None
            vals, dlogits = ops.ce_loss(logits.detach().contiguous(), y, float(getattr(self.config, "l2_reg", 0.0)), ss_q, ss,
                                        grad_scale=1.0 / self.dp.world if self._exchange else 1.0)
            logits.backward(dlogits)
        finally:
            layers.defer_wgrad_reductions(*prev)
        layers.run_deferred_wgrads(jobs)
        ops.conv_wgrad_reduce(jobs)
        del jobs
        if self.stem_slot is not None:
            self.stem_w.grad = self.stem_slot
        for l in self.qlayers:
            l._shared = None
        if self._exchange:
            torch.cat([p.grad.reshape(-1) for p in self.dense], out=self.g_flat[self.nW:])
        self.reg, self.loss = vals[1], vals[2]
        self.logits = logits.detach()

    def _phase_b(self):
        from . import ops
        o = self.opts[0]
        grp = o.param_groups[0]
        coef = 2.0 * float(getattr(self.config, "l2_reg", 0.0))
        if self._exchange:
            off = self.nW
            for p in self.dense:  # the update reads (and completes) the reduced gradients where they lie
                p.grad = self.g_flat[off:off + p.numel()].view_as(p)
                off += p.numel()
        st = [o.state[w] for w in self.ws]
        kw = {}
        if self.mode == "ttq":
            sst = [(o.state[p], o.state[n]) for p, n in self.scales]
            kw = dict(stats=[q["stats"] for q in self.q], Wp=[p.detach() for p, _ in self.scales], Wn=[n.detach() for _, n in self.scales],
                      scale_state=[(a["exp_avg"], a["exp_avg_sq"], a["step"], b["exp_avg"], b["exp_avg_sq"], b["step"]) for a, b in sst])
        elif self.mode == "inq":
            kw = dict(masks=[l.mask for l in self.qlayers])
        ops.tern_backward_adam([w.detach() for w in self.ws], self.slots, self.mode, [t["exp_avg"] for t in st], [t["exp_avg_sq"] for t in st],
                               [t["step"] for t in st], lr=grp["lr"], betas=grp["betas"], eps=grp["eps"], reg=coef, out_dW=self.dW_out,
                               out_dscale=self.dscale_out if self.mode == "ttq" else None, **kw)
        if self.keep_grads:
            for w, g in zip(self.ws, self.dW_out):
                w.grad = g
            for (p, n), (gp, gn) in zip(self.scales, self.dscale_out or []):
                p.grad, n.grad = gp, gn
        prm = [p for p in self.dense if p.grad is not None]
        dst = [o.state[p] for p in prm]
        ops.adam_step([p.detach() for p in prm], [p.grad for p in prm], [t["exp_avg"] for t in dst], [t["exp_avg_sq"] for t in dst],
                      [t["step"] for t in dst], lr=grp["lr"], betas=grp["betas"], eps=grp["eps"],
                      regs=[coef if id(p) in self.dense_reg else 0.0 for p in prm])

    def _comm(self):
        """The exchange: the flat buffer in consecutive buckets, asynchronous all-reduces issued back to back."""
        if not self._exchange:
            return
        n = self.g_flat.numel()
        per = max(1, self.bucket_bytes // 4)
        nb = max(1, (n + per - 1) // per)
        per = (n + nb - 1) // nb
        self.dp.reduce_buckets([self.g_flat[i:min(i + per, n)] for i in range(0, n, per)])

    # ---- driver ------------------------------------------------------------------------------
    def __call__(self, x, y):
        if not self.use_graph:
            self._ensure_adam_state()
            self._phase_a(x, y)
            self._comm()
            self._phase_b()
            return self.loss
        key = (tuple(x.shape), tuple(g["lr"] for g in self.opts[0].param_groups), self._exchange)
        if self._graphs is None or key != self._key:
            self._capture(x, y)
            self._key = key
        if x is not self._sx:
            self._sx.copy_(x)
        if y is not self._sy:
            self._sy.copy_(y)
        self._graphs[0].replay()
        if self._graphs[1] is not None:
            self._comm()
            self._graphs[1].replay()
        return self.loss

    def invalidate(self):
        self._graphs = self._key = None

    def static_inputs(self):
        return (self._sx, self._sy) if (self.use_graph and self._graphs is not None) else None

    def _capture(self, x, y):
        _refuse_sync_bn_under_capture(self.model, self.dp)
        self._sx, self._sy = x.clone(), y.clone()
        self._ensure_adam_state()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        torch.cuda.synchronize()
        ga, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with torch.cuda.stream(s):
            # (only the BatchNorm running statistics: the pass writes nothing else, and INQ layers key their decoded
            # frozen levels on the version counters of their own buffers)
            bufs = [b for m in self.model.modules() if isinstance(m, nn.modules.batchnorm._BatchNorm) for b in m.buffers(recurse=False)]
            saved = [b.clone() for b in bufs]
            self._phase_a(self._sx, self._sy)  # lazy initialisation of whatever stock operators the model still uses
            for b, v in zip(bufs, saved):
                b.copy_(v)
            torch.cuda.synchronize()
            if self._exchange:
                with torch.cuda.graph(ga, stream=s):
                    self._phase_a(self._sx, self._sy)
                with torch.cuda.graph(gb, stream=s, pool=ga.pool()):
                    self._phase_b()
            else:
                gb = None
                with torch.cuda.graph(ga, stream=s):
                    self._phase_a(self._sx, self._sy)
                    self._phase_b()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        self._graphs = (ga, gb)
