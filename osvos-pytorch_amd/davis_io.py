"""DAVIS-2016 frame access for the device-side input pipeline.

The reference decodes with ``cv2.imread`` inside ``DAVIS2016.__getitem__`` (dataloaders/davis_2016.py:88-112), converts to
float32, subtracts the mean and augments with OpenCV on DataLoader workers (custom_transforms.py:21-52,87-121; train_online.py:92-97,
train_parent.py:106-113).  Here the host only DECODES: ``DavisFrames`` lists the same files in the same order
(davis_2016.py:36-63) and returns the raw uint8 BGR frame + uint8 label (decoded with Pillow -- OpenCV is not part of this
image; both sit on libjpeg, the JPEG bit-exactness against cv2 is unpinned here), ``DevicePrefetcher`` moves them through
pinned staging buffers to the GPU on a copy stream a few frames ahead of the consumer, and
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


class DevicePrefetcher(object):
    """Iterate ``(index, uint8 CUDA frame [H,W,3], uint8 CUDA label [H,W] | None)`` over ``indices`` of ``frames``.

    A host thread decodes ``depth`` frames ahead into a ring of pinned staging buffers; the H2D copies run on their own stream
    and the consumer's stream waits on the copy event only -- decode, PCIe and the training step overlap.  A staging slot is
    re-used only after the copy that read it has completed (event), so the ring needs no extra synchronisation."""

    def __init__(self, frames, indices, device, depth=3):
        self.frames, self.indices, self.device, self.depth = frames, list(indices), torch.device(device), max(1, int(depth))
        if self.device.type != 'cuda':
            raise RuntimeError("DevicePrefetcher feeds the GPU input pipeline; it needs a CUDA (ROCm) device")
        self.copy_stream = torch.cuda.Stream(device=self.device)
        self.q = queue.Queue(maxsize=self.depth)
        self.slots = [None] * (self.depth + 1)           # pinned (img, lab) buffers, allocated at the first frame's size
        self.slot_free = [None] * (self.depth + 1)       # event: the H2D copy out of this slot is done
        self.thread = threading.Thread(target=self._produce, daemon=True)
        self.thread.start()

    def _pinned(self, k, img, lab):
        cur = self.slots[k]
        if cur is None or cur[0].shape != img.shape or (lab is not None and (cur[1] is None or cur[1].shape != lab.shape)):
            cur = (torch.empty(img.shape, dtype=torch.uint8).pin_memory(),
                   torch.empty(lab.shape, dtype=torch.uint8).pin_memory() if lab is not None else None)
            self.slots[k] = cur
        return cur

    def _produce(self):
        try:
            for n, idx in enumerate(self.indices):
                img, lab = self.frames[idx]
                k = n % (self.depth + 1)
                if self.slot_free[k] is not None:
                    self.slot_free[k].synchronize()
                pi, pl = self._pinned(k, img, lab)
                pi.copy_(torch.from_numpy(img))
                if lab is not None:
                    pl.copy_(torch.from_numpy(lab))
                with torch.cuda.stream(self.copy_stream):
                    di = pi.to(self.device, non_blocking=True)
                    dl = pl.to(self.device, non_blocking=True) if lab is not None else None
                    ev = torch.cuda.Event()
                    ev.record(self.copy_stream)
                self.slot_free[k] = ev
                self.q.put((idx, di, dl, ev))
            self.q.put(None)
        except BaseException as e:          # surface decode errors in the consumer instead of hanging it
            self.q.put(e)

    def __iter__(self):
        while True:
            item = self.q.get()
            if item is None:
                return
            if isinstance(item, BaseException):
                raise item
            idx, di, dl, ev = item
            torch.cuda.current_stream(self.device).wait_event(ev)
            di.record_stream(torch.cuda.current_stream(self.device))
            if dl is not None:
                dl.record_stream(torch.cuda.current_stream(self.device))
            yield idx, di, dl
