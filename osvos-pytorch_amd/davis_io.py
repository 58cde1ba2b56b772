"""DAVIS-2016 frame access for the device-side input pipeline.

The reference decodes with ``cv2.imread`` inside ``DAVIS2016.__getitem__`` (dataloaders/davis_2016.py:88-112), converts to
float32, subtracts the mean and augments with OpenCV on DataLoader workers (custom_transforms.py:21-52,87-121; train_online.py:92-97,
train_parent.py:106-113).  Here the host only DECODES: ``DavisFrames`` lists the same files in the same order
(davis_2016.py:36-63) and returns the raw uint8 BGR frame + uint8 label (decoded with Pillow -- OpenCV is not part of this
image; both sit on libjpeg, the JPEG bit-exactness against cv2 is unpinned here), ``DevicePrefetcher`` decodes them on a host thread into a ring of
pinned staging buffers a few frames ahead and the consumer copies them to the GPU on its own stream, and
``osvos_pytorch_amd.augment.DeviceAugment`` does mean / flip / warp / CHW float32 in one kernel.
"""
from __future__ import annotations

import os
import queue
import threading

import numpy as np
import torch


class DavisFrames(object):
    """File lists of reference dataloaders/davis_2016.py:36-63 (``train_seqs.txt`` / ``val_seqs.txt`` under db_root_dir, or one
    sequence: first frame only when ``train``, every frame with the first annotation when not)."""

    def __init__(self, train=True, db_root_dir=None, seq_name=None):
        self.train, self.db_root_dir, self.seq_name = train, db_root_dir, seq_name
        if seq_name is None:
            img_list, labels = [], []
            with open(os.path.join(db_root_dir, ('train_seqs' if train else 'val_seqs') + '.txt')) as f:
                seqs = [s.strip() for s in f.readlines() if s.strip()]
            for seq in seqs:
                images = sorted(os.listdir(os.path.join(db_root_dir, 'JPEGImages/480p/', seq)))
                img_list.extend(os.path.join('JPEGImages/480p/', seq, x) for x in images)
                lab = sorted(os.listdir(os.path.join(db_root_dir, 'Annotations/480p/', seq)))
                labels.extend(os.path.join('Annotations/480p/', seq, x) for x in lab)
        else:
            names_img = sorted(os.listdir(os.path.join(db_root_dir, 'JPEGImages/480p/', str(seq_name))))
            img_list = [os.path.join('JPEGImages/480p/', str(seq_name), x) for x in names_img]
            name_label = sorted(os.listdir(os.path.join(db_root_dir, 'Annotations/480p/', str(seq_name))))
            labels = [os.path.join('Annotations/480p/', str(seq_name), name_label[0])] + [None] * (len(names_img) - 1)
            if train:
                img_list, labels = [img_list[0]], [labels[0]]
        if len(labels) != len(img_list):
            raise ValueError("DAVIS lists disagree: %d frames, %d annotations" % (len(img_list), len(labels)))
        self.img_list, self.labels = img_list, labels

    def __len__(self):
        return len(self.img_list)

    def fname(self, idx):
        return os.path.join(str(self.seq_name), "%05d" % idx) if self.seq_name is not None else self.img_list[idx]

    def __getitem__(self, idx):
        """(uint8 [H,W,3] BGR, uint8 [H,W] label or None): what cv2.imread(path) / cv2.imread(path, 0) hand the reference."""
        from PIL import Image
        with Image.open(os.path.join(self.db_root_dir, self.img_list[idx])) as im:
            img = np.ascontiguousarray(np.asarray(im.convert('RGB'))[:, :, ::-1])
        lab = None
        if self.labels[idx] is not None:
            with Image.open(os.path.join(self.db_root_dir, self.labels[idx])) as im:
                lab = np.ascontiguousarray(np.asarray(im.convert('L')))
        return img, lab


class ArrayFrames(object):
    """In-memory stand-in for ``DavisFrames`` (synthetic runs, tests): a list of (uint8 [H,W,3], uint8 [H,W] | None)."""

    def __init__(self, frames):
        self.frames = list(frames)

    def __len__(self):
        return len(self.frames)

    def fname(self, idx):
        return "%05d" % idx

    def __getitem__(self, idx):
        return self.frames[idx]


_POOL = {}        # (image shape, label shape) -> free pinned (image, label) staging pairs, kept for the life of the process: pinning host memory costs
                  # milliseconds per buffer and train_parent.py builds one prefetcher per epoch (profiles/r04_scripts_e2e.txt: the feeder stall of
                  # round 4).  A pair belongs to exactly ONE prefetcher between _take and _give: two live prefetchers never share a buffer, and a
                  # prefetcher gives its pairs back only after its producer has stopped and its in-flight copies have completed (close()).
_POOL_LOCK = threading.Lock()


def _take(img_shape, lab_shape):
    key = (tuple(img_shape), tuple(lab_shape) if lab_shape is not None else None)
    with _POOL_LOCK:
        free = _POOL.setdefault(key, [])
        if free:
            return key, free.pop()
    return key, (torch.empty(img_shape, dtype=torch.uint8).pin_memory(),
                 torch.empty(lab_shape, dtype=torch.uint8).pin_memory() if lab_shape is not None else None)


def _give(key, pair):
    with _POOL_LOCK:
        _POOL.setdefault(key, []).append(pair)


class DevicePrefetcher(object):
    """Iterate ``(index, uint8 CUDA frame [H,W,3], uint8 CUDA label [H,W] | None)`` over ``indices`` of ``frames``.

    A host thread decodes ``depth`` frames ahead into a ring of pinned staging buffers -- CPU work only.  The CONSUMER (the thread that also
    launches the training step) enqueues the two H2D copies of a frame on ITS OWN current stream, in line with the step that will read
    them: 1.6 MB per 854x480 frame, ~40 us of a 4.2 ms step.  A staging slot goes back to the producer only after the copies that read it have
    completed (an event on the same stream), which also keeps the host at most depth + 2 frames ahead of the GPU.

    How it got here (round 4, profiles/r04_scripts_e2e.txt; train_parent.py --device-augment, frames/s, fp32x3 / bf16, next to bench.py's
    239 / 555-594 for the same loop on a resident frame):
      1. copies, events and event waits issued by the PRODUCER thread on a copy stream: the autograd engine's thread launches ~100 kernels
         per backward through ctypes at the same time and the backward call took 8-12 ms of host time instead of 1.4 ............. - / 81
      2. no HIP call from the producer; the consumer issues the copies on a separate copy stream, 8 hardware queues ................ 154 / 310
      3. the same on 4 hardware queues (GPU_MAX_HW_QUEUES; with 5 or more the copy stream gets a queue of its own and every step
         stretches from 4.3 to 6.2 ms) .......................................................................................... 216 / 492
      4. no copy stream at all (this form) ........................................................................................ 239 / 580"""

    def __init__(self, frames, indices, device, depth=3):
        self.frames, self.indices, self.device, self.depth = frames, list(indices), torch.device(device), max(1, int(depth))
        if self.device.type != 'cuda':
            raise RuntimeError("DevicePrefetcher feeds the GPU input pipeline; it needs a CUDA (ROCm) device")
        self.q = queue.Queue(maxsize=self.depth)
        self.free = queue.Queue()                        # slots the consumer has finished copying out of
        for k in range(self.depth + 2):
            self.free.put(k)
        self.slots = [None] * (self.depth + 2)           # (pool key, pinned (img, lab) pair) per slot, taken from the pool at the first frame's size
        self._stop, self._closed, self._inflight = threading.Event(), False, []
        # the producer starts with the FIRST iteration, not here: a running thread holds a reference to its owner, so a prefetcher that
        # was built but never iterated could not be collected -- its daemon thread polled the queue every 50 ms for ever and kept its pinned buffers
        self.thread = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def _pinned(self, k, img, lab):
        cur = self.slots[k]
        want = (tuple(img.shape), tuple(lab.shape) if lab is not None else None)
        if cur is None or cur[0] != want:
            if cur is not None:
                _give(*cur)                              # (only reached for slots whose copies have completed: the slot came off self.free)
            cur = self.slots[k] = _take(img.shape, lab.shape if lab is not None else None)
        return cur[1]

    def _produce(self):
        try:
            for idx in self.indices:
                img, lab = self.frames[idx]
                k = None
                while k is None:                         # a slot whose previous copy has completed
                    if self._stop.is_set():
                        return
                    try:
                        k = self.free.get(timeout=0.05)
                    except queue.Empty:
                        pass
                pi, pl = self._pinned(k, img, lab)
                np.copyto(pi.numpy(), img)               # (numpy releases the GIL for the copy; also for the strided views synthetic frames are)
                if lab is not None:
                    np.copyto(pl.numpy(), lab)
                if not self._put((idx, k, lab is not None)):
                    return
            self._put(None)
        except BaseException as e:          # surface decode errors in the consumer instead of hanging it
            self._put(e)

    def _put(self, item):
        while not self._stop.is_set():
            try:
                self.q.put(item, timeout=0.05)
                return True
            except queue.Full:
                pass
        return False

    def close(self, join_timeout=5.0):
        """Stop the producer, wait for the copies that still read staging buffers, hand the buffers back to the pool.  Runs when iteration
        ends -- normally, by an exception, or because the consumer abandoned the iterator (generator close / garbage collection); idempotent."""
        if self._closed:
            return
        self._closed = True
        self._stop.set()
        if self.thread is not None:
            self.thread.join(timeout=join_timeout)
        for _, e0 in self._inflight:
            e0.synchronize()
        self._inflight = []
        if self.thread is None or not self.thread.is_alive():      # (a producer stuck in a decode keeps its buffers: never hand out memory a thread may still write)
            for k, cur in enumerate(self.slots):
                if cur is not None:
                    _give(*cur)
                    self.slots[k] = None

    def __del__(self):
        try:
            self.close(join_timeout=0.2)     # (garbage collection / interpreter shutdown must not wait five seconds for a decode)
        except Exception:
            pass

    def __iter__(self):
        if self._closed:
            raise RuntimeError("DevicePrefetcher: already closed (one pass per prefetcher)")
        if self.thread is None:
            self.thread = threading.Thread(target=self._produce, daemon=True)
            self.thread.start()
        inflight = self._inflight            # (slot, event) of copies not known to be complete yet
        try:
            while True:
                item = self.q.get()
                if item is None:
                    return
                if isinstance(item, BaseException):
                    raise item
                idx, k, has_lab = item
                pi, pl = self.slots[k][1]
                di = pi.to(self.device, non_blocking=True)        # (enqueued on the consumer's current stream of self.device)
                dl = pl.to(self.device, non_blocking=True) if has_lab else None
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(self.device))      # ... and so is the event: self.device's stream, not the current device's
                inflight.append((k, ev))
                while inflight and (inflight[0][1].query() or len(inflight) > self.depth):
                    k0, e0 = inflight.pop(0)
                    if not e0.query():
                        e0.synchronize()
                    self.free.put(k0)
                yield idx, di, dl
        finally:
            self.close()
