"""train() step of the reference (src/train_cnn_lstm.py:131-150).

Two optimiser paths behind the same train(batch, model, criterion, optimizer) call:
  * any torch.optim optimiser (the reference passes torch.optim.Adam, train_cnn_lstm.py:363): gradients are clamped to
    [-5, 5] on the device (vocr_clamp, NaN preserved like torch's clamp_) and optimizer.step() runs as given;
  * FlatClampAdam (the fast path, `make_optimizer(model)`): all parameters live in ONE flat fp32 buffer (params / grads /
    Adam moments), so zero_grad is one memset, the data-parallel exchange is one RCCL all-reduce over xGMI on the flat
    gradient, and clamp(+-5) + Adam is one HBM-streaming kernel (vocr_clamp_adam) instead of ~150 small launches.

Data parallelism: one process per GPU, full replica, per-rank batch; gradients are SUMMED across ranks
before the clamp (the reference's loss is a batch SUM, so this equals one big batch of B*N)."""
import os

import torch
import torch.distributed as dist

from . import ops
from ._lib import prof_range


def _dp_active(group=None):
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or
                                                              os.environ.get("VOCR_FORCE_DIST") == "1")


class FlatClampAdam(object):
    """torch.optim.Adam(lr, betas, eps, weight_decay) semantics + the reference's elementwise grad clamp, on a
    flat parameter buffer.  API subset of torch.optim.Optimizer used by the reference loop: zero_grad(), step(),
    param_groups[0]['lr'], state_dict()/load_state_dict()."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, clamp=5.0, named_split=None):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("no parameters")
        dev = self.params[0].device      # buffers may be built on the host (tests of the DP exchange); step() needs the GPU
        n = sum(p.numel() for p in self.params)
        self.flat_p = torch.empty(n, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=dev)
        self._offsets = {}
        off = 0
        with torch.no_grad():
            for p in self.params:
                k = p.numel()
                self.flat_p[off:off + k].copy_(p.data.reshape(-1))
                p.data = self.flat_p[off:off + k].view_as(p)          # parameter storage now aliases the flat buffer
                p.grad = self.flat_g[off:off + k].view_as(p)          # autograd accumulates in place into the flat grads
                self._offsets[id(p)] = off
                # every parameter now owns a persistent gradient buffer that zero_grad() clears once per step: the backward
                # kernels may write its gradient straight there (ops._sinks).  The mark is per PARAMETER, so other models
                # and optimisers in the process are unaffected.
                if dev.type == "cuda":
                    p._vocr_sink_owner = self
                off += k
        self.param_groups = [dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, clamp=clamp)]
        self.step_count = 0
        self._written = set()            # parameters whose gradient a backward kernel stored directly since zero_grad()
        # Two all-reduce buckets: [0, split) = parameters whose gradients appear last (rapid_ds + cnn), [split, n) =
        # bridge + LSTM + prob (88 % of the bytes at H=512), final as soon as the backward reaches the CNN, so their
        # exchange over xGMI is launched there and hides under the CNN backward (make_optimizer registers the hook).
        self._split = int(named_split) if named_split is not None else 0
        self._tail_work = None
        self._comm_stream = None
        self._group = None               # process group of BOTH buckets (None = the default group); see set_group()
        self._time_comm = False          # bench.py: HIP events around both buckets of the exchange
        self._comm_events = []           # (bucket name, start event, end event)

    # ---- direct-gradient bookkeeping (ops._sinks)
    def owns_grads(self, params):
        base = self.flat_g.data_ptr()
        for p in params:
            off = self._offsets.get(id(p))
            if off is None or p.grad.data_ptr() != base + off * 4:
                return False
        return True

    def note_direct_write(self, params):
        for p in params:
            if id(p) in self._written:
                raise RuntimeError("vistaocr_amd.FlatClampAdam: second backward() before zero_grad(): the backward kernels "
                                   "store weight gradients directly into the flat buffer (they do not accumulate). Call "
                                   "optimizer.zero_grad() between backward passes, or use a torch.optim optimiser for "
                                   "gradient accumulation.")
            self._written.add(id(p))

    def set_group(self, group):
        """Data-parallel process group for the gradient exchange (both buckets); None = the default group."""
        self._finish_tail()
        self._group = group

    def time_comm(self, on):
        """Record HIP events around the two all-reduce buckets of the following steps (bench.py's allreduce_ms_per_step)."""
        self._time_comm = bool(on) and self.flat_g.is_cuda

    def comm_times_ms(self):
        """{"sequence_bucket": ms, "cnn_bucket": ms, "steps": n}: mean duration of each bucket's all-reduce over the steps recorded
        since the last call, from 'inputs ready' to 'reduced' on the stream that carries it (None if nothing was recorded)."""
        if not self._comm_events:
            return None
        torch.cuda.synchronize(self.flat_g.device)
        acc, cnt = {}, {}
        for name, e0, e1 in self._comm_events:
            acc[name] = acc.get(name, 0.0) + e0.elapsed_time(e1)
            cnt[name] = cnt.get(name, 0) + 1
        self._comm_events = []
        out = {k: round(acc[k] / cnt[k], 4) for k in acc}
        out["steps"] = max(cnt.values())
        out["bytes"] = {"sequence_bucket": 4 * (self.flat_g.numel() - self._split), "cnn_bucket": 4 * self._split}
        return out

    def _finish_tail(self):
        """Wait for a sequence-side all-reduce that a backward pass started and nobody has collected yet."""
        if self._tail_work is not None:
            self._tail_work.wait()
            self._tail_work = None

    def zero_grad(self, set_to_none=False):
        self._finish_tail()              # never clear a buffer a collective is still reducing
        self.flat_g.zero_()
        self._written.clear()
        off = 0
        for p in self.params:                                         # re-attach if someone set .grad = None
            k = p.numel()
            if p.grad is None or p.grad.data_ptr() != self.flat_g.data_ptr() + off * 4:
                p.grad = self.flat_g[off:off + k].view_as(p)
            off += k

    _dp_active = staticmethod(_dp_active)

    def _start_tail_allreduce(self):
        """Backward hook: bridge/LSTM/prob gradients are final -> start their all-reduce asynchronously."""
        if self._split > 0 and self._tail_work is None and _dp_active(self._group):
            if self.flat_g.is_cuda:
                # The exchange must see the weight-gradient kernels still queued on the side stream (LSTM layer 0's), but the
                # MAIN stream must not wait for them: it goes straight on into the CNN backward, which those kernels overlap.
                # So the collective is enqueued from a stream of its own that waits for both.
                dev = self.flat_g.device
                if self._comm_stream is None:
                    self._comm_stream = torch.cuda.Stream(device=dev)
                self._comm_stream.wait_stream(torch.cuda.current_stream(dev))
                self._comm_stream.wait_stream(ops.side_stream(dev))
                with torch.cuda.stream(self._comm_stream):
                    if self._time_comm:
                        e0 = torch.cuda.Event(enable_timing=True)
                        e0.record()
                    self._tail_work = dist.all_reduce(self.flat_g[self._split:], op=dist.ReduceOp.SUM, group=self._group, async_op=True)
                    if self._time_comm:
                        self._tail_work.wait()          # orders the (dedicated) comm stream behind the collective, not the host
                        e1 = torch.cuda.Event(enable_timing=True)
                        e1.record()
                        self._comm_events.append(("sequence_bucket", e0, e1))
            else:
                self._tail_work = dist.all_reduce(self.flat_g[self._split:], op=dist.ReduceOp.SUM, group=self._group, async_op=True)

    def all_reduce_grads(self, group=None):
        """Sum gradients over data-parallel ranks (RCCL over xGMI when the backend is 'nccl').  `group`: a process group
        other than the one set with set_group() is only accepted while no bucket is in flight on the latter."""
        if self.flat_g.is_cuda:
            ops.join_side_stream(self.flat_g.device)
        if group is None:
            group = self._group
        elif group is not self._group:
            if self._tail_work is not None:
                raise RuntimeError("vistaocr_amd.FlatClampAdam: the sequence-side bucket is already being reduced on the group "
                                   "given to set_group(); pass the same group here (or call set_group(group) before backward)")
        if not _dp_active(group):
            self._finish_tail()
            return
        e0 = None
        if self._time_comm:
            e0 = torch.cuda.Event(enable_timing=True)
            e0.record()
        if self._tail_work is not None:                 # tail bucket already in flight since mid-backward
            dist.all_reduce(self.flat_g[:self._split], op=dist.ReduceOp.SUM, group=group)
            name = "cnn_bucket"
        else:
            dist.all_reduce(self.flat_g, op=dist.ReduceOp.SUM, group=group)
            name = "whole_gradient"
        if e0 is not None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            self._comm_events.append((name, e0, e1))
        self._finish_tail()

    def step(self, grad_scale=1.0):
        g = self.param_groups[0]
        if self._tail_work is not None:
            # a backward pass started the sequence-side bucket's exchange but all_reduce_grads() never ran: the CNN bucket is
            # unreduced and the collective may still be writing flat_g - refuse instead of racing it
            self._finish_tail()
            raise RuntimeError("vistaocr_amd.FlatClampAdam.step(): a data-parallel process group is active and backward() has "
                               "started the gradient exchange; call optimizer.all_reduce_grads() between backward() and step()")
        ops.join_side_stream(self.flat_g.device)
        self.step_count += 1
        ops.clamp_adam(self.flat_p, self.flat_g, self.exp_avg, self.exp_avg_sq, g["lr"], g["betas"][0], g["betas"][1],
                       g["eps"], g["weight_decay"], g["clamp"], grad_scale, self.step_count)

    def check_health(self):
        """Synchronising check of the device health words (LSTM hand-off timeout, NaN gradient)."""
        ops.check_health_sync(self.flat_g.device)

    def state_dict(self):
        """In torch.optim.Adam's own format ({'state': {i: {'step', 'exp_avg', 'exp_avg_sq'}}, 'param_groups': [...]}, the
        parameters numbered in model.parameters() order), so the 'optimizer' entry of a snapshot (src/train_cnn_lstm.py:429)
        loads into a torch.optim.Adam over the same model and vice versa.  The moments are views of the flat buffers."""
        g = self.param_groups[0]
        state, off = {}, 0
        for i, p in enumerate(self.params):
            k = p.numel()
            state[i] = {"step": torch.tensor(float(self.step_count)), "exp_avg": self.exp_avg[off:off + k].view_as(p),
                        "exp_avg_sq": self.exp_avg_sq[off:off + k].view_as(p)}
            off += k
        group = {"lr": g["lr"], "betas": tuple(g["betas"]), "eps": g["eps"], "weight_decay": g["weight_decay"], "amsgrad": False,
                 "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
                 "decoupled_weight_decay": False, "clamp": g["clamp"], "params": list(range(len(self.params)))}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        if "state" not in sd:                                         # round-1 snapshots: flat moments
            self.step_count = int(sd["step"])
            self.exp_avg.copy_(sd["exp_avg"])
            self.exp_avg_sq.copy_(sd["exp_avg_sq"])
            self.param_groups[0].update({k: v for k, v in sd["param_groups"][0].items()})
            return
        if sd["state"] and len(sd["state"]) != len(self.params):
            raise ValueError("optimizer state has %d parameters, the model %d" % (len(sd["state"]), len(self.params)))
        off = 0
        with torch.no_grad():
            for i, p in enumerate(self.params):
                k = p.numel()
                st = sd["state"].get(i)
                if st is not None:
                    self.exp_avg[off:off + k].copy_(st["exp_avg"].reshape(-1))
                    self.exp_avg_sq[off:off + k].copy_(st["exp_avg_sq"].reshape(-1))
                    self.step_count = int(float(st["step"]))
                off += k
        g = sd["param_groups"][0]
        for key in ("lr", "betas", "eps", "weight_decay", "clamp"):
            if key in g:
                self.param_groups[0][key] = g[key]


def make_optimizer(model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, clamp=5.0):
    """FlatClampAdam over model.parameters() with the bucket boundary placed after the CNN parameters; the model's
    "sequence gradients are final" backward milestone starts the big bucket's all-reduce."""
    n_cnn = sum(p.numel() for p in list(model.rapid_ds.parameters()) + list(model.cnn.parameters()) if p.requires_grad)
    opt = FlatClampAdam(model.parameters(), lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, clamp=clamp, named_split=n_cnn)
    hooks = getattr(model, "_vocr_hooks", None)
    if hooks is not None and n_cnn > 0:
        hooks["sequence_grads_ready"] = [opt._start_tail_allreduce]
    return opt


def seed_rank(model, rank, base=0):
    """Per-rank randomness of a data-parallel replica: call AFTER the identical initialisation.  Re-seeds torch's
    CPU and device generators (FractionalMaxPool2d samples are torch.rand draws per forward, src/models/cnnlstm.py:127,130) with
    base + 1000*rank and offsets the model's counter-based dropout stream (nn.LSTM's inter-layer dropout,
    src/train_cnn_lstm.py:331), so ranks draw different samples and masks while their weights stay identical."""
    torch.manual_seed(int(base) + 1000 * int(rank))
    if torch.cuda.is_available():
        torch.cuda.manual_seed(int(base) + 1000 * int(rank))
    if hasattr(model, "dropout_seed"):
        model.dropout_seed = 0x5EED + 1000003 * int(rank)
        model._dropout_calls = 0


GRAD_CLAMP = 5.0          # src/train_cnn_lstm.py:143-145


def _generic_update(model, optimizer):
    """The reference's own tail of train() for an optimiser that is not FlatClampAdam: (all-reduce,) clamp every
    gradient to [-5, 5] on the device, optimizer.step()."""
    grads = [p.grad for p in model.parameters() if p.grad is not None]
    if grads and grads[0].is_cuda:
        ops.join_side_stream(grads[0].device)
    if _dp_active():
        for g in grads:
            dist.all_reduce(g, op=dist.ReduceOp.SUM)
    for g in grads:
        if not g.is_contiguous():
            g.data = g.data.contiguous()
        ops.clamp_(g.data, GRAD_CLAMP)
    optimizer.step()


def _step(batch, model, criterion, optimizer, want_float):
    input_tensor, target, input_widths, target_widths, metadata = batch
    input_tensor = input_tensor.cuda(non_blocking=True)
    optimizer.zero_grad()
    with prof_range("train.forward"):
        model_output, model_output_actual_lengths = model(input_tensor, input_widths)
    with prof_range("train.ctc"):
        loss = criterion(model_output, target, model_output_actual_lengths, target_widths)
    # The float the reference returns is known as soon as the forward pass is done: its copy to pinned host memory is
    # queued HERE, ahead of the backward kernels, and only that copy is waited for at the end.  The host then gets the
    # value while the device is still busy with this step's backward + Adam and starts queueing the next step at once
    # (with loss.item() after optimizer.step() the device ran dry at the start of every step: 23.6 vs 22.2 ms/step).
    # The device health words ride along: they cover the previous step's backward/optimiser and this step's forward.
    ev = None
    if want_float and loss.is_cuda:
        host, host_health = _pinned(loss.device)
        host.copy_(loss.detach().reshape(-1)[:1], non_blocking=True)
        host_health.copy_(ops.health(loss.device), non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
    with prof_range("train.backward"):
        loss.backward()
    if hasattr(optimizer, "all_reduce_grads"):
        with prof_range("train.exchange"):
            optimizer.all_reduce_grads()
        with prof_range("train.clamp_adam"):
            optimizer.step()
    else:
        with prof_range("train.exchange_clamp_step"):
            _generic_update(model, optimizer)
    if not want_float:
        return loss.detach()
    if ev is None:
        return loss.data[0].item()
    ev.synchronize()
    ops.check_health(host_health.tolist(), loss.device)
    return float(host[0])


def train(batch, model, criterion, optimizer):
    """One optimisation step — src/train_cnn_lstm.py:131-150; returns the batch-summed loss as a Python float.
    `optimizer` is a FlatClampAdam (clamp fused into its step) or any torch.optim optimiser (device clamp, then step())."""
    return _step(batch, model, criterion, optimizer, True)


_PINNED = {}


def _pinned(device):
    key = str(device)
    if key not in _PINNED:
        _PINNED[key] = (torch.empty(1, dtype=torch.float32).pin_memory(), torch.zeros(2, dtype=torch.int32).pin_memory())
    return _PINNED[key]


def train_async(batch, model, criterion, optimizer):
    """Same step without the device->host sync of the returned float (returns the loss tensor; no health check —
    call optimizer.check_health() or train() now and then)."""
    return _step(batch, model, criterion, optimizer, False)
