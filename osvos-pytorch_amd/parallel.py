"""Data-parallel gradient exchange: one process per GPU, torch.distributed ('nccl' = RCCL over
xGMI on ROCm; 'gloo' on CPU for the tests).  The reference has no distributed code at all
(single device, gradient *accumulation* only: train_online.py:140-149, train_parent.py:163-172);
this is the one exchange step the data-parallel version of those loops needs.

Each rank runs its own micro-batches and accumulates locally; once per optimizer step the flat
gradient buffer (15.27 M fp32 = 61 MB, frozen deconv weights carry no gradient) is summed across
ranks with ONE all-reduce -- a single large collective, sized for the per-link-bound xGMI ring
rather than many small buckets -- and every rank applies the identical SGD update.  From the second optimizer step on (gradients living in
the flat arena, written in place by the backward kernels) the collective is issued as SEVEN chunks in the order the backward completes
them (head, side_prep, stages.4 ... stages.0), each on a communication stream that waits for that group's gradient-ready event
(osvos_net_arm_grad_events): the deep layers' 50 MB travel over xGMI while conv3_x .. conv1_x are still being differentiated.

With the reference's per-frame class-balance weights (osvos_layers.py:30-32) W ranks x (nAveGrad/W)
micro-batches reproduce the single-process gradient exactly (up to summation order) whenever W
divides nAveGrad: use average=False and keep `loss /= nAveGrad`.  average=True divides the sum by
the world size (weak scaling: every rank keeps nAveGrad local micro-batches)."""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def _grad_group(name):
    """Completion group of a parameter inside osvos_net_backward (include/osvos_hip.h, osvos_net_arm_grad_events):
    0 = score_dsn + fuse, 1 = side_prep, 2..6 = stages.4 .. stages.0; modules that are not an OSVOS tree: one group."""
    if name.startswith(("score_dsn.", "fuse.")):
        return 0
    if name.startswith("side_prep."):
        return 1
    if name.startswith("stages."):
        return 2 + (4 - int(name.split(".")[1]))
    return 0


def comm_port():
    """TCP port of the id store of the C-ABI communicator: OSVOS_COMM_PORT, else MASTER_PORT + 1"""
    p = os.environ.get("OSVOS_COMM_PORT")
    if p:
        port = int(p)
        if not (1024 <= port <= 65535):
            raise ValueError("OSVOS_COMM_PORT = %s: need a port in 1024..65535" % p)
        return port
    return int(os.environ.get("MASTER_PORT", "29500")) + 1


def _store_timeout():
    import datetime
    return datetime.timedelta(seconds=float(os.environ.get("OSVOS_COMM_TIMEOUT", "120")))


class AbiCommunicator:
    """RCCL communicator held through the C ABI (``osvos_comm_*``, csrc/comm.cpp) instead of torch.distributed: the gradient sum is
    then a plain ``ncclAllReduce`` enqueued by the library -- and, chunked behind the backward's gradient-ready events, costs the host
    seven stream-wait + enqueue pairs instead of seven torch.distributed work objects.  The 128-byte RCCL id travels from rank 0 to
    the others through a ``torch.distributed.TCPStore`` (no process group needed) at MASTER_ADDR : OSVOS_COMM_PORT -- an explicit port of
    the job's own; without the variable MASTER_PORT + 1 is used, and a port that is already taken (a second job on the node) is reported as
    such instead of hanging the rendezvous."""

    def __init__(self, rank, world, device):
        import ctypes as C
        from ._lib import check, lib
        self._C, self._check, self._lib = C, check, lib()
        self.rank, self.world, self.device = rank, world, torch.device(device)
        ident = (C.c_char * 128)()
        store = None
        if world > 1:      # the rendezvous first: a taken port is reported before RCCL or the device are touched
            port = comm_port()
            try:
                store = dist.TCPStore(os.environ.get("MASTER_ADDR", "127.0.0.1"), port, world, rank == 0, timeout=_store_timeout())
            except Exception as e:      # (rank 0: EADDRINUSE; others: nobody listening within the timeout)
                raise RuntimeError("osvos communicator: cannot %s the id store at %s:%d (%s).  Another job on this node probably uses the port: "
                                   "give every job its own OSVOS_COMM_PORT (default MASTER_PORT + 1)."
                                   % ("open" if rank == 0 else "reach", os.environ.get("MASTER_ADDR", "127.0.0.1"), port, e)) from e
        if rank == 0:
            check(self._lib.osvos_comm_unique_id(ident), "comm_unique_id")
        if store is not None:
            if rank == 0:
                store.set("osvos_comm_id", bytes(ident.raw))
            else:
                raw = store.get("osvos_comm_id")
                C.memmove(ident, raw, 128)
        self.handle = C.c_void_p()
        with torch.cuda.device(self.device):
            check(self._lib.osvos_comm_init(C.byref(self.handle), rank, world, ident), "comm_init")

    def all_reduce(self, flat):
        """in-place sum over the ranks of a contiguous float32 (gradients) or float64 (class counts, loss statistics: exact) CUDA tensor"""
        C = self._C
        if flat.dtype not in (torch.float32, torch.float64) or not flat.is_contiguous():
            raise RuntimeError("AbiCommunicator.all_reduce: contiguous float32 / float64 tensors only (got %s)" % flat.dtype)
        fn = self._lib.osvos_comm_allreduce_f32 if flat.dtype == torch.float32 else self._lib.osvos_comm_allreduce_f64
        self._check(fn(self.handle, C.c_void_p(flat.data_ptr()), flat.numel(), C.c_void_p(torch.cuda.current_stream(flat.device).cuda_stream)),
                    "comm_allreduce")

    def all_reduce_chunks(self, flat, slices, events, comm_stream):
        C = self._C
        n = len(slices)
        first = (C.c_size_t * n)(*[a for _, a, _ in slices])
        count = (C.c_size_t * n)(*[b - a for _, a, b in slices])
        evs = (C.c_void_p * n)(*[C.c_void_p(events[g].cuda_event) for g, _, _ in slices])
        self._check(self._lib.osvos_comm_allreduce_chunks_f32(self.handle, C.c_void_p(flat.data_ptr()), first, count, evs, n,
                                                              C.c_void_p(comm_stream.cuda_stream)), "comm_allreduce_chunks")

    def close(self):
        if self.handle:
            self._check(self._lib.osvos_comm_destroy(self.handle), "comm_destroy")
            self.handle = self._C.c_void_p()


class GradientAllReducer:
    """Gradients live in ONE flat buffer: after the first backward every ``p.grad`` is re-pointed to a view into it
    (``attach``), the network's backward accumulates into those views in place, the collective runs on the flat buffer
    itself and ``zero_grads`` is one memset -- no flatten / unflatten copies (2 x 61 MB per optimizer step before) and no
    per-tensor allocations.  If somebody replaces a ``.grad`` (``optimizer.zero_grad()`` with set_to_none) the next call
    copies it back in and re-attaches."""

    def __init__(self, module, average=False, process_group=None, always=False, overlap=True, comm=None):
        named = [(n, p) for n, p in module.named_parameters() if p.requires_grad]
        # arena order = the order in which a backward COMPLETES the gradients (head, side_prep, stages.4 ... stages.0): every group is
        # one contiguous slice, so a group can be reduced as soon as its ready-event fires while shallower layers still compute
        named.sort(key=lambda np_: _grad_group(np_[0]))            # stable: state_dict order inside a group
        self.params = [p for _, p in named]
        self._group_of = {id(p): _grad_group(n) for n, p in named}
        self.module = module
        self.average = average
        self.group = process_group
        self.always = always          # run the collective even in a 1-rank group (exercises RCCL on one GPU)
        self.comm = comm              # AbiCommunicator: RCCL through the C ABI instead of torch.distributed (OSVOS_DP_BACKEND=abi in the scripts)
        # Overlap is OPT-IN (OSVOS_DP_OVERLAP=1).  Measured on one MI355X with a one-rank RCCL group (profiles/r02_dp_selfcheck.txt,
        # bench.py --force-dist): the blocking single collective costs nothing (211.6 vs 212.5 frames/s), while seven chunked
        # collectives behind events cost the HOST ~10 ms per optimizer step through torch.distributed's work bookkeeping (146
        # frames/s; the GPU timeline itself is unchanged) -- more than a 61 MB all-reduce over xGMI can take.  The mechanism and
        # its bit-identity check stay (tools/dp_selfcheck.py runs both paths); it is switched on by those who measure a gain.
        self.overlap = overlap and hasattr(module, "_runtime") and os.environ.get("OSVOS_DP_OVERLAP", "0") == "1"
        self._flat = None
        self._views = {}              # id(param) -> view into _flat
        self._slices = []             # [(group, start, stop)] of the flat buffer, in completion order
        self._events = None
        self._comm_stream = None
        self._armed = False
        self.overlapped_steps = 0

    def attach(self):
        """Point the .grad of every parameter that has one at its slice of the flat buffer.  Returns the flat buffer
        (None when no parameter has a gradient yet)."""
        have = [p for p in self.params if p.grad is not None]
        if not have:
            return None
        total = sum(p.grad.numel() for p in have)
        layout_ok = (self._flat is not None and self._flat.numel() == total and self._flat.device == have[0].grad.device
                     and len(self._views) == len(have) and all(id(p) in self._views for p in have))
        if not layout_ok:
            self._flat = torch.empty(total, device=have[0].grad.device, dtype=have[0].grad.dtype)
            self._views, off = {}, 0
            self._slices = []
            for p in have:
                n = p.grad.numel()
                self._views[id(p)] = self._flat[off:off + n].view_as(p.grad)
                g = self._group_of[id(p)]
                if self._slices and self._slices[-1][0] == g:
                    self._slices[-1] = (g, self._slices[-1][1], off + n)
                else:
                    self._slices.append((g, off, off + n))
                off += n
        stale = [p for p in have if p.grad.data_ptr() != self._views[id(p)].data_ptr()]
        if stale:
            torch._foreach_copy_([self._views[id(p)] for p in stale], [p.grad for p in stale])
            for p in stale:
                p.grad = self._views[id(p)]
        return self._flat

    def zero_grads(self):
        """Replacement for ``optimizer.zero_grad()``: keeps the views alive, one memset.  Falls back to setting the
        gradients to None before the first ``attach``."""
        if self._flat is None:
            for p in self.params:
                p.grad = None
            return
        self._flat.zero_()
        for p in self.params:
            v = self._views.get(id(p))
            if v is not None:
                p.grad = v

    def _world(self):
        return self.comm.world if self.comm is not None else dist.get_world_size(self.group)

    def _active(self):
        if self.comm is not None:
            return self.comm.world > 1 or self.always
        return dist.is_initialized() and (dist.get_world_size(self.group) > 1 or self.always)

    def arm(self):
        """Call right BEFORE the backward of the LAST micro-batch of an optimizer step.  If the gradients already live in the flat
        arena (every .grad is a view of it: the in-place accumulating backward then writes FINAL values straight into it), the next
        osvos_net_backward records one event per completion group and ``all_reduce`` reduces group by group on a communication
        stream, each chunk as soon as its event fires -- overlapped with the backward of the shallower layers.  Returns False (and
        ``all_reduce`` falls back to one blocking collective after the backward) before the arena is attached, on CPU, or when
        the module is not an OSVOS tree."""
        self._armed = False
        if not (self.overlap and self._active() and self._flat is not None and self._flat.is_cuda):
            return False
        rt = self.module._runtime
        if not getattr(rt, "inplace_accumulate", False):
            return False
        if any(p.grad is None or p.grad.data_ptr() != self._views[id(p)].data_ptr() for p in self.params if id(p) in self._views):
            return False
        dev = self._flat.device
        if self._events is None:
            self._events = [torch.cuda.Event() for _ in range(7)]
            # the forward's side-branch stream is idle during a backward: issue the collectives from it rather than from yet another
            # stream (ROCm maps streams onto GPU_MAX_HW_QUEUES hardware queues; one stream too many and two of the backward's alias)
            self._comm_stream = rt._stream_for(dev, "side")
            for e in self._events:
                e.record(torch.cuda.current_stream(dev))       # torch creates the hipEvent_t lazily, at the first record
        rt.grad_events = [e.cuda_event for e in self._events]
        self._armed = True
        return True

    def all_reduce(self):
        """Sum (or average) the .grad of every parameter that has one across all ranks."""
        armed, self._armed = self._armed, False
        if armed:      # only if the backward really recorded the events (it does so only when it accumulated in place into the arena)
            rt = self.module._runtime
            armed, rt.grad_events_recorded = bool(rt.grad_events_recorded), False
            rt.grad_events = None
        flat = self.attach()
        if flat is None:
            return
        if not self._active():
            return
        if self.comm is not None:
            if armed:
                main = torch.cuda.current_stream(flat.device)
                self.comm.all_reduce_chunks(flat, self._slices, self._events, self._comm_stream)
                main.wait_stream(self._comm_stream)
                self.overlapped_steps += 1
            else:
                self.comm.all_reduce(flat)
        elif armed:
            main = torch.cuda.current_stream(flat.device)
            with torch.cuda.stream(self._comm_stream):
                for g, a, b in self._slices:                     # completion order: head, side_prep, stages.4 ... stages.0
                    self._comm_stream.wait_event(self._events[g])
                    # synchronous-style call: the collective is ordered after this stream's work and this stream waits for it;
                    # the host does not block
                    dist.all_reduce(flat[a:b], op=dist.ReduceOp.SUM, group=self.group)
            main.wait_stream(self._comm_stream)                  # the optimizer's stream waits for the collectives, not the host
            self.overlapped_steps += 1
        else:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
        if self.average:
            flat.div_(self._world())

    def broadcast_parameters(self, src=0):
        """Make every rank start from rank `src`'s weights: ONE collective over a flat copy of all parameters (52 tensors, 61 MB) instead of
        one per tensor.  torch.distributed: a broadcast; the C-ABI communicator has one collective, so the broadcast is a sum to which every
        other rank contributes zeros."""
        ps = [p for p in self.module.parameters()]
        if not ps or not self._active_or_world():
            return
        with torch.no_grad():
            flat = torch.cat([p.data.reshape(-1) for p in ps])
            if self.comm is not None:
                if self.comm.rank != src:
                    flat.zero_()
                self.comm.all_reduce(flat)
            else:
                dist.broadcast(flat, src=src, group=self.group)
            off = 0
            for p in ps:
                n = p.numel()
                p.data.copy_(flat[off:off + n].view_as(p.data))
                off += n

    def _active_or_world(self):
        if self.comm is not None:
            return self.comm.world > 1
        return dist.is_initialized() and dist.get_world_size(self.group) > 1


def global_class_counts(label, group=None, comm=None):
    """The two numbers a batch that is SHARDED over ranks must exchange for the class-balanced loss to equal the single-process loss of the
    whole batch (reference layers/osvos_layers.py:28-34 counts positives and negatives over the whole input tensor and :46 divides
    by the batch size): returns (n_pos, n_total, n_images) of the GLOBAL batch as float64 tensors on the label's device -- one all-reduce of
    three doubles.  Without a process group (or with one rank) these are the local counts.

    Use: ``loss = cbce_with_counts(output, label, n_pos, n_total, n_images)`` on every rank, gradients summed over ranks -> exactly the
    single-process batch loss.  The reference scripts train with batch 1 per micro-batch (per-frame class weights), where nothing needs
    exchanging: this is for callers that shard ONE batch (bench.py --mode parent --batch B under data parallelism treats every rank's batch
    as its own reference batch instead, and says so)."""
    lab = (label >= 0.5)
    t = torch.stack([lab.sum().double(), torch.tensor(float(label.numel()), device=label.device, dtype=torch.float64),
                     torch.tensor(float(label.shape[0]), device=label.device, dtype=torch.float64)])
    if comm is not None and comm.world > 1:
        comm.all_reduce(t)                       # float64 sum (osvos_comm_allreduce_f64): exact for any count
    elif comm is None and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, group=group)
    return t[0], t[1], t[2]


def cbce_with_counts(output, label, n_pos, n_total, n_images, size_average=False, batch_average=True):
    """class_balanced_cross_entropy_loss (osvos_layers.py:19-48) of this rank's SHARD of a batch, weighted and normalised with the GLOBAL counts
    of ``global_class_counts``: summed over the ranks it is the reference loss of the whole batch (same operation order per element; the class
    weights stay float32 quotients like the reference's).  CUDA tensors take the HIP loss kernel with the counts as arguments
    (``osvos_cbce_step_ex``: no count sweep, weights and divisors from the global numbers; differentiable through ``CBCECountsFunction``); CPU
    tensors (the gloo tests) the plain torch expression below."""
    if output.is_cuda:
        from .autograd import CBCECountsFunction
        mode = 0 if size_average else (1 if batch_average else 2)
        cnt = torch.stack([n_pos.reshape(()).float(), n_total.reshape(()).float(), n_images.reshape(()).float()]).to(output.device)
        return CBCECountsFunction.apply(output, label, cnt, mode)
    labels = (label >= 0.5).float()
    n_pos32, n_tot32 = n_pos.float(), n_total.float()
    w_pos, w_neg = (n_tot32 - n_pos32) / n_tot32, n_pos32 / n_tot32
    g = (output >= 0).float()
    val = output * (labels - g) - torch.log(1 + torch.exp(output - 2 * output * g))
    final = w_pos * (-(labels * val)).sum() + w_neg * (-((1.0 - labels) * val)).sum()
    if size_average:
        final = final / n_total.to(final.dtype)
    elif batch_average:
        final = final / n_images.to(final.dtype)
    return final


def shard_indices(n_items, rank, world_size):
    """Frames / sequences of a job handled by `rank`: r, r+W, r+2W, ...."""
    return list(range(rank, n_items, world_size))
