"""Shared pieces of the two entry scripts: the SGD parameter groups and the restated inner loops of
the reference's train_online.py / train_parent.py, written as functions so the same code serves the
scripts, the benchmark and the tests.  Data-parallel operation (one process per GPU, gradient
all-reduce once per optimizer step) is layered on with ``GradientAllReducer``."""
from __future__ import division

import os

import torch
import torch.optim as optim

import numpy as np

from .layers.osvos_layers import (class_balanced_cross_entropy_loss, class_balanced_cross_entropy_loss_step,
                                  class_balanced_cross_entropy_loss_step_multi)
from .parallel import GradientAllReducer


def make_sgd(net, mode, lr=1e-8, wd=0.0002, momentum=0.9, fused=None):
    """Parameter groups of train_online.py:79-88 (mode 'online', score_dsn not optimised) or
    train_parent.py:87-103 (mode 'parent').  ``fused`` (default: on, OSVOS_FUSED_SGD=0 turns it off) selects
    ``FusedSGD`` -- one launch for all tensors, same numbers and state_dict -- over ``torch.optim.SGD``."""
    if fused is None:
        fused = os.environ.get("OSVOS_FUSED_SGD", "1") != "0"
    groups = [
        {'params': [pr[1] for pr in net.stages.named_parameters() if 'weight' in pr[0]], 'weight_decay': wd, 'initial_lr': lr},
        {'params': [pr[1] for pr in net.stages.named_parameters() if 'bias' in pr[0]], 'lr': 2 * lr, 'initial_lr': 2 * lr},
        {'params': [pr[1] for pr in net.side_prep.named_parameters() if 'weight' in pr[0]], 'weight_decay': wd, 'initial_lr': lr},
        {'params': [pr[1] for pr in net.side_prep.named_parameters() if 'bias' in pr[0]], 'lr': 2 * lr, 'initial_lr': 2 * lr},
    ]
    if mode == 'parent':
        groups += [
            {'params': [pr[1] for pr in net.score_dsn.named_parameters() if 'weight' in pr[0]], 'lr': lr / 10,
             'weight_decay': wd, 'initial_lr': lr / 10},
            {'params': [pr[1] for pr in net.score_dsn.named_parameters() if 'bias' in pr[0]], 'lr': 2 * lr / 10,
             'initial_lr': 2 * lr / 10},
        ]
    groups += [
        {'params': [pr[1] for pr in net.upscale.named_parameters() if 'weight' in pr[0]], 'lr': 0, 'initial_lr': 0},
        {'params': [pr[1] for pr in net.upscale_.named_parameters() if 'weight' in pr[0]], 'lr': 0, 'initial_lr': 0},
        {'params': net.fuse.weight, 'lr': lr / 100, 'initial_lr': lr / 100, 'weight_decay': wd},
        {'params': net.fuse.bias, 'lr': 2 * lr / 100, 'initial_lr': 2 * lr / 100},
    ]
    if fused:
        from .optim import FusedSGD
        return FusedSGD(groups, lr=lr, momentum=momentum)
    return optim.SGD(groups, lr=lr, momentum=momentum)


class TrainLoop(object):
    """One micro-batch = forward, loss, ``loss /= nAveGrad``, backward; optimizer step every
    ``nAveGrad`` micro-batches (train_online.py:116-149, train_parent.py:132-172).  The running loss
    is kept on the device and only read back when the caller asks for it (the reference's
    ``loss.item()`` every iteration is a logging artefact that stalls the GPU)."""

    def __init__(self, net, optimizer, mode='online', n_ave_grad=5, n_epochs=240, reducer=None, local_ave=None, loss_fn=None,
                 max_steps=None):
        """``n_ave_grad``: the loss divisor = micro-batches per optimizer step summed over ALL ranks (the reference's nAveGrad).
        ``local_ave``: micro-batches THIS rank contributes to a step (default: all of them); the step -- all-reduce, SGD update,
        one-memset zeroing of the flat gradient arena -- fires after that many local backwards.  The counter persists across
        epochs like the reference's ``aveGrad`` (train_parent.py:128,163-172).  ``loss_fn``: the class-balanced BCE (default: the
        HIP kernel; the CPU tests of the data-parallel bookkeeping pass the oracle's).  ``max_steps``: the number of COMPLETE
        optimizer-step windows of the whole run (``StepSchedule.total_steps``): a rank whose share of the trailing partial window
        happens to reach ``local_ave`` must not step (and must not enter a collective the other ranks never enter) -- like the
        reference, whose ``aveGrad`` simply never reaches nAveGrad again, the tail's gradients are computed and dropped."""
        self.net, self.opt, self.mode = net, optimizer, mode
        self.n_ave_grad, self.n_epochs = n_ave_grad, n_epochs
        self.local_ave = int(local_ave) if local_ave else n_ave_grad
        self.loss_fn = loss_fn or class_balanced_cross_entropy_loss
        # the HIP loss hands back (loss, scaled gradient) in one call and keeps the running loss on the device: no autograd scalar chain
        # (ones / div / add / scale launches) between the loss kernel and the head's backward.  OSVOS_FUSED_LOSS_STEP=0: the plain chain.
        self.fused_loss = loss_fn is None and os.environ.get("OSVOS_FUSED_LOSS_STEP", "1") != "0"
        self.reducer = reducer
        self.max_steps = max_steps
        self.ave = 0
        self.steps = 0
        if hasattr(net, 'set_inplace_grad_accumulation'):
            net.set_inplace_grad_accumulation(True)      # every backward of this loop is loss.backward()
        if hasattr(net, 'set_deferred_backward_join') and os.environ.get("OSVOS_DEFER_JOIN", "0") == "1":
            net.set_deferred_backward_join(True)         # the weight-gradient tail of micro-batch k runs under the forward of k + 1
        self._dev = next(net.parameters()).device
        self._nloss = 5 if mode == 'parent' else 1
        self._running = {}                                # epoch -> [device scalars]: a step window may hold frames of two epochs
        self.counts = {}                                  # epoch -> micro-batches this rank ran

    def _acc(self, epoch):
        r = self._running.get(epoch)
        if r is None:
            r = self._running[epoch] = [torch.zeros((), device=self._dev) for _ in range(self._nloss)]
            self.counts[epoch] = 0
        return r

    def micro_batch(self, inputs, gts, epoch=0):
        """One micro-batch.  Returns ``(loss, stepped)``: ``loss`` is the PLAIN loss of the fused head (detached 0-dim tensor; not divided by
        nAveGrad, side heads not mixed in) on every code path -- what the reference prints per iteration; sums over an epoch come from
        ``pop_running``.  ``stepped``: this micro-batch closed an optimizer-step window."""
        outputs = self.net.forward(inputs)
        running = self._acc(epoch)
        self.counts[epoch] += 1
        if self.fused_loss and outputs[-1].is_cuda:
            return self._micro_batch_fused(outputs, gts, running, epoch)
        if self.mode == 'online':
            loss = self.loss_fn(outputs[-1], gts, size_average=False)
            running[0] += loss.detach()
            plain = loss.detach().clone()
        else:
            losses = [self.loss_fn(o, gts, size_average=False) for o in outputs]
            for r, l in zip(running, losses):
                r += l.detach()
            plain = losses[-1].detach().clone()
            loss = (1 - epoch / self.n_epochs) * sum(losses[:-1]) + losses[-1]
        loss /= self.n_ave_grad
        will_step = (self.ave + 1) % self.local_ave == 0 and (self.max_steps is None or self.steps < self.max_steps)
        if self.reducer is not None and will_step:
            self.reducer.arm()          # last micro-batch of the step: reduce each gradient group as soon as its backward is done
        loss.backward()
        return plain, self._after_backward(will_step)

    def window_batch(self, inputs, gts, epoch=0):
        """A whole accumulation window in ONE forward / backward (round 4): ``inputs`` holds the ``local_ave`` frames this rank contributes to
        the optimizer step as a batch; each image is its own reference micro-batch -- class weights from its own label, loss divided by
        nAveGrad (``per_image`` mode of the loss kernel) -- so the accumulated gradient is the one ``local_ave`` ``micro_batch`` calls leave,
        up to fp32 summation order (train_online.py:116-149 / train_parent.py:132-172; the micro-batches of a window share the weights, nothing
        couples the images of a batch in this network).  conv5_x sees M = 5 x 1620 pixels instead of 1620, every launch of the step is issued
        once instead of nAveGrad times.  Returns ``(sum of the window's plain fused losses, stepped)``.
        ONE ``epoch`` applies to the whole window (its statistics bucket and, in parent mode, the side-head weight 1 - epoch / nEpochs):
        a parent-mode window that straddles an epoch boundary differs from ``micro_batch`` in the side weight of the frames past the boundary --
        callers that need the reference's per-frame epoch there run such a window through ``micro_batch`` (train_online.py has one epoch
        counter per window by construction: nAveGrad consecutive iterations of one sequence)."""
        n = int(inputs.shape[0])
        if n != self.local_ave or self.ave != 0:
            raise RuntimeError("window_batch: needs exactly local_ave = %d frames at the start of a window (got %d, %d accumulated)" % (self.local_ave, n, self.ave))
        if not inputs.is_cuda:
            raise RuntimeError("window_batch: CUDA tensors only (the per-image loss mode lives in the HIP kernel)")
        outputs = self.net.forward(inputs)
        running = self._acc(epoch)
        self.counts[epoch] += n
        inv = np.float32(1.0) / np.float32(self.n_ave_grad)
        if self.mode == 'online':
            heads, scales = [outputs[-1]], [inv]
        else:
            side = np.float32(inv * np.float32(1 - epoch / self.n_epochs))
            heads, scales = list(outputs), [side] * (len(outputs) - 1) + [inv]
        losses, grads = class_balanced_cross_entropy_loss_step_multi(heads, gts, size_average=False, grad_scales=[float(s) for s in scales],
                                                                     running=list(running), per_image=True)
        will_step = self.max_steps is None or self.steps < self.max_steps
        if self.reducer is not None and will_step:
            self.reducer.arm()
        torch.autograd.backward(heads, grads)
        self.ave = self.local_ave - 1
        stepped = self._after_backward(will_step)
        if not will_step:
            self.ave = 0      # past max_steps the window's gradients are computed and dropped, like micro_batch's trailing windows: the next call starts a window
        return losses[-1], stepped

    def _micro_batch_fused(self, outputs, gts, running, epoch):
        """The same micro-batch with the upstream gradients of ``loss /= nAveGrad; loss.backward()`` (train_online.py:140-141;
        train_parent.py:147,163-164: side heads weighted by 1 - epoch / nEpochs) folded into the loss kernel -- float32 products formed
        the way autograd forms them (1 / nAveGrad, then times the float32 side weight)."""
        inv = np.float32(1.0) / np.float32(self.n_ave_grad)
        if self.mode == 'online':
            heads, scales = [outputs[-1]], [inv]
        else:
            side = np.float32(inv * np.float32(1 - epoch / self.n_epochs))
            heads, scales = list(outputs), [side] * (len(outputs) - 1) + [inv]
        if len(heads) > 1 and os.environ.get('OSVOS_CBCE_MULTI', '1') != '0':      # the parent loop's five losses: one library call (class counts formed once, one sweep)
            losses, grads = class_balanced_cross_entropy_loss_step_multi(heads, gts, size_average=False, grad_scales=[float(s) for s in scales],
                                                                         running=list(running))
            loss = losses[-1]
        else:
            grads, loss = [], None
            for o, s, r in zip(heads, scales, running):
                loss, g = class_balanced_cross_entropy_loss_step(o, gts, size_average=False, grad_scale=float(s), running=r)
                grads.append(g)
        will_step = (self.ave + 1) % self.local_ave == 0 and (self.max_steps is None or self.steps < self.max_steps)
        if self.reducer is not None and will_step:
            self.reducer.arm()
        torch.autograd.backward(heads, grads)
        return loss.detach(), self._after_backward(will_step)       # (the fused head's plain loss, like the other path)

    def _after_backward(self, will_step):
        self.ave += 1
        stepped = False
        if will_step:
            if hasattr(self.net, 'join_backward'):
                self.net.join_backward()           # (deferred join: the gradients are complete on the main stream from here on)
            if self.reducer is not None:
                self.reducer.all_reduce()
            self.opt.step()
            if self.reducer is not None:
                self.reducer.zero_grads()          # the flat gradient arena stays attached: one memset
            else:
                self.opt.zero_grad()
            self.ave = 0
            self.steps += 1
            stepped = True
        return stepped

    def finish(self):
        """Join anything a deferred-join backward left running (call before reading gradients outside an optimizer step)."""
        if hasattr(self.net, 'join_backward'):
            self.net.join_backward()

    def pop_running(self, epoch=None):
        """Running loss sums (host floats) of `epoch` (default: everything accumulated so far) and forget them."""
        keys = list(self._running) if epoch is None else ([epoch] if epoch in self._running else [])
        vals = [0.0] * self._nloss
        for k in keys:
            for i, r in enumerate(self._running.pop(k)):
                vals[i] += float(r.item())
        return vals

    def pop_count(self, epoch):
        return self.counts.pop(epoch, 0)


class StepSchedule(object):
    """Where the optimizer steps of a run fall in the global micro-batch stream, and with them the only places at which every rank
    of a data-parallel run is known to stand at the same point of its collective sequence.

    The stream g = first_epoch * n_items, ... is cut into windows of nAveGrad iterations ACROSS epoch boundaries (the reference's
    ``aveGrad`` counter persists: train_parent.py:128,163-172); window k ends with the k-th gradient all-reduce.  2079 DAVIS frames
    and nAveGrad 10 never line up, so anything exchanged "at the end of an epoch" would be entered by the ranks at different
    positions between two gradient collectives (round-2 deadlock).  Per-epoch statistics are therefore exchanged right after the
    gradient collective of the window that holds the epoch's LAST iteration (``closing_step``) -- every rank has finished its share
    of the epoch by then -- and epochs that end in the trailing partial window are exchanged after the last epoch."""

    def __init__(self, n_items, n_ave_grad, first_epoch, n_epochs, start_iteration=None):
        """``start_iteration``: global index g of the first iteration of THIS run (default: ``first_epoch * n_items``, also what a
        reference-style resume does).  An exact resume (train_parent.py --save-optimizer) starts at the iteration after the optimizer step at
        which its checkpoint was taken -- in general somewhere inside an epoch -- so that the windows of the resumed run are the windows of the
        uninterrupted one."""
        self.n_items, self.n_ave_grad = int(n_items), int(n_ave_grad)
        self.first_epoch, self.n_epochs = int(first_epoch), int(n_epochs)
        self.start = int(start_iteration) if start_iteration is not None else self.first_epoch * self.n_items
        self.total_iterations = max(0, self.n_epochs * self.n_items - self.start)
        self.total_steps = self.total_iterations // self.n_ave_grad          # complete windows: what EVERY rank steps, no more

    def closing_step(self, epoch):
        """Index (within this run) of the optimizer step whose window holds the last iteration of `epoch`; None when that is the trailing
        partial window (no step, no gradient collective: such epochs are closed after the training loop)."""
        last = (epoch + 1) * self.n_items - 1 - self.start
        if last < 0:
            return None
        k = last // self.n_ave_grad
        return k if k < self.total_steps else None

    def next_iteration(self, steps_done):
        """global index of the first iteration after `steps_done` optimizer steps of this run"""
        return self.start + int(steps_done) * self.n_ave_grad

    def closed_by(self, steps_done, pending):
        """The epochs of `pending` (ascending) whose statistics may be exchanged once `steps_done` optimizer steps are complete."""
        out = []
        for e in pending:
            k = self.closing_step(e)
            if k is not None and k < steps_done:
                out.append(e)
        return out


def check_world_divides(n_ave_grad, world):
    """Data-parallel accumulation reproduces the single-process gradient only when every rank contributes the same number
    of micro-batches to every optimizer step, i.e. when the world size divides nAveGrad.  Anything else would
    silently train on a different effective batch; refuse it and say what works."""
    if world < 1 or n_ave_grad % world != 0:
        up = -(-n_ave_grad // world) * world
        raise ValueError("nAveGrad = %d is not a multiple of the world size %d: each optimizer step would sum %d micro-batches scaled by "
                         "1/%d.  Use --n-ave-grad %d (or a world size among %s)."
                         % (n_ave_grad, world, (n_ave_grad // world) * world, n_ave_grad, up,
                            [w for w in range(1, n_ave_grad + 1) if n_ave_grad % w == 0]))
    return n_ave_grad // world


def epoch_plan(n_items, epoch, n_ave_grad, rank=0, world=1, seed=0, shuffle=True):
    """Which frames THIS rank runs in `epoch`, in order: a list of (dataset index, global iteration number).

    Every rank derives the SAME permutation of range(n_items) for an epoch (generator seeded with (seed, epoch)), so across
    ranks each frame is visited exactly once per epoch and nothing is decoded twice.  The global micro-batch stream
    g = epoch * n_items + position is cut into optimizer steps of n_ave_grad consecutive iterations -- across epoch
    boundaries, like the reference's persistent ``aveGrad`` counter (train_parent.py:128,163-172) -- and inside a step rank r
    takes slots r, r + W, ...: every rank contributes n_ave_grad / W micro-batches to every step, all ranks enter the same
    number of all-reduces, and W x (n_ave_grad / W) micro-batches of per-frame class-balanced losses sum to the
    single-process gradient of the same stream (up to summation order)."""
    check_world_divides(n_ave_grad, world)
    if shuffle:
        g = torch.Generator().manual_seed(int(seed) * 1000003 + int(epoch))
        perm = torch.randperm(n_items, generator=g).tolist()
    else:
        perm = list(range(n_items))
    g0 = epoch * n_items
    # slot (g mod n_ave_grad) mod W == g mod W because W divides n_ave_grad: a plain round-robin over the global stream, so
    # ANY n_ave_grad consecutive iterations (also a window that starts at a resume point) hold n_ave_grad / W of every rank
    return [(idx, g0 + j) for j, idx in enumerate(perm) if (g0 + j) % world == rank]


_ABI_COMM = None      # the process's AbiCommunicator (OSVOS_DP_BACKEND=abi), created by init_distributed, used by make_reducer


def init_distributed(collectives=True):
    """One process per GPU (torchrun / torch.distributed.run).  Returns (rank, world, device).

    With more than one rank and `collectives` the RCCL communicator is brought up HERE, before the process allocates or launches anything on
    the GPU: a communicator initialised after the first device activity leaves every later step ~30 % slower on the GPU side (measured with
    one rank through both backends: 232 -> 162 frames/s on the headline loop; profiles/r03_dp_backends.txt).  torch.distributed creates its
    communicator eagerly when init_process_group is given `device_id`; the C-ABI communicator (OSVOS_DP_BACKEND=abi) is created right away
    and NO torch.distributed process group is opened then (one RCCL instance per process; gradients and epoch statistics use osvos_comm_*).
    `collectives=False` (train_online.py: every rank fine-tunes its own sequences, nothing is exchanged) opens nothing."""
    global _ABI_COMM
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if torch.cuda.is_available():
        torch.cuda.set_device(local)
        device = torch.device('cuda', local)
    else:
        device = torch.device('cpu')
    force = os.environ.get('OSVOS_DP_FORCE', '0') == '1'      # one rank, communicator and gradient all-reduce anyway: single-GPU run of the DP path
    if (world > 1 or (force and device.type == 'cuda')) and collectives:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if device.type != 'cuda':
            dist.init_process_group('gloo')
        elif os.environ.get('OSVOS_DP_BACKEND', 'torch') == 'abi':
            from .parallel import AbiCommunicator
            _ABI_COMM = AbiCommunicator(rank, world, device)
        else:
            dist.init_process_group('nccl', device_id=device, rank=rank, world_size=world)
            dist.barrier()          # (first collective: RCCL's own lazy initialisation happens now, not inside the first optimizer step)
    return rank, world, device


def make_reducer(net, world, average=False):
    """The gradient exchange of the data-parallel loop: torch.distributed ('nccl' = RCCL) by default, RCCL through the library's own C ABI
    with OSVOS_DP_BACKEND=abi (osvos_comm_*: no process group at all; the chunked overlap is cheap there).  Either communicator was brought up by
    init_distributed, before the network existed."""
    force = os.environ.get('OSVOS_DP_FORCE', '0') == '1' and next(net.parameters()).is_cuda
    if world <= 1 and not force:
        return None
    comm = _ABI_COMM          # OSVOS_DP_BACKEND=abi: created by init_distributed before anything touched the device
    return GradientAllReducer(net, average=average, comm=comm, always=force)
