"""Device-side input feed: what sits between the dataset and ``model.forward``.

In the reference the LayoutLMv3 processor resizes / rescales / normalises every page on the host (PIL + numpy,
EE/data/RVL_CDIP.py:246-262), ``DataCollatorWithPadding(padding="max_length")`` pads on the host (EE/utils.py:93-98) and the
finished float tensors (602 KB per page) cross PCIe one batch at a time (EE/utils.py:173).  Here the raw uint8 page and the
ragged token ids cross PCIe (a greyscale 1000x762 page is 0.76 MB, usually less than its float tensor... and an "L" page
needs no RGB expansion), and resize + normalise + padding run on the GPU (ee_preprocess_images / ee_collate_pad) on a side
stream.  ``DeviceFeeder`` double-buffers this through two PINNED host staging slots: a batch is packed into one slot, crosses
PCIe as ONE asynchronous copy and is preprocessed on the side stream while the caller's stream runs the model on the previous
batch.  ``preprocess_images`` / ``collate_pad`` are the one-shot forms of the same kernels (pageable source, synchronous copy).
"""
from __future__ import annotations

import ctypes as C
from typing import Any, Dict, Iterable, Iterator, List, Optional, Sequence

import numpy as np

from . import capi
from .engine import _require_torch_cuda, torch

_DESC = np.dtype([("offset", np.int64), ("h", np.int32), ("w", np.int32), ("c", np.int32), ("pad", np.int32)])
MAX_RATIO = 31


def _stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def pack_images(images: Sequence[np.ndarray]):
    """Host side: concatenate uint8 pages ((H,W) greyscale or (H,W,3) RGB) + descriptor records."""
    desc = np.zeros(len(images), dtype=_DESC)
    chunks, off = [], 0
    for i, im in enumerate(images):
        a = np.ascontiguousarray(im, dtype=np.uint8)
        if a.ndim == 3 and a.shape[2] == 1:
            a = a[:, :, 0]
        if a.ndim not in (2, 3) or (a.ndim == 3 and a.shape[2] != 3):
            raise ValueError("images must be (H,W) or (H,W,3) uint8")
        desc[i] = (off, a.shape[0], a.shape[1], 1 if a.ndim == 2 else 3, 0)
        chunks.append(a.reshape(-1))
        off += a.size
        off = (off + 15) & ~15                      # keep every image 16-byte aligned
        pad = off - (desc[i]["offset"] + a.size)
        if pad:
            chunks.append(np.zeros(pad, np.uint8))
    return np.concatenate(chunks) if chunks else np.zeros(0, np.uint8), desc


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _preprocess_on_device(d_img, d_desc, B: int, size: int, max_h: int, dev, return_u8: bool = False):
    """ee_preprocess_images on buffers that are already on the device (current stream)."""
    lib = capi.load()
    ws_bytes = lib.ee_preprocess_workspace_bytes(B, size, max_h)
    ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
    out = torch.empty((B, 3, size, size), dtype=torch.float32, device=dev)
    u8 = torch.empty((B, size, size, 3), dtype=torch.uint8, device=dev) if return_u8 else None
    with torch.cuda.device(dev):
        capi.check(lib.ee_preprocess_images(_p(d_img), _p(d_desc), B, size, max_h, _p(ws), ws_bytes, _p(out), _p(u8), _stream_ptr()),
                   None, "ee_preprocess_images")
    return (out, u8) if return_u8 else out


def _collate_on_device(d_ids, d_bx, d_off, B: int, T: int, pad_id: int, dev):
    """ee_collate_pad on buffers that are already on the device (current stream)."""
    lib = capi.load()
    o_ids = torch.empty((B, T), dtype=torch.int64, device=dev)
    o_am = torch.empty((B, T), dtype=torch.int64, device=dev)
    o_bb = torch.empty((B, T, 4), dtype=torch.int64, device=dev)
    with torch.cuda.device(dev):
        capi.check(lib.ee_collate_pad(_p(d_ids), _p(d_bx), _p(d_off), B, T, pad_id, _p(o_ids), _p(o_am), _p(o_bb), _stream_ptr()), None,
                   "ee_collate_pad")
    return o_ids, o_am, o_bb


def _check_sizes(desc, size):
    if max(int(desc["h"].max()), int(desc["w"].max())) > MAX_RATIO * size:
        raise ValueError(f"image side / {size} must be <= {MAX_RATIO}")


def preprocess_images(images: Sequence[np.ndarray], size: int = 224, device=None, return_u8: bool = False):
    """uint8 pages -> pixel_values (B,3,size,size) float32 on the device, identical to the HF/PIL pipeline.  One-shot form:
    the packed pages are copied from pageable memory (a synchronous copy); ``DeviceFeeder`` is the pipelined form."""
    dev = _require_torch_cuda(device)
    buf, desc = pack_images(images)
    _check_sizes(desc, size)
    d_img = torch.from_numpy(buf).to(dev)
    d_desc = torch.from_numpy(desc.view(np.uint8).reshape(-1)).to(dev)
    return _preprocess_on_device(d_img, d_desc, len(images), size, int(desc["h"].max()), dev, return_u8)


def _flatten_tokens(input_ids, bboxes):
    lens = np.array([len(x) for x in input_ids], dtype=np.int64)
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    ids = np.concatenate([np.asarray(x, dtype=np.int64).reshape(-1) for x in input_ids]) if offs[-1] else np.zeros(1, np.int64)
    bx = np.concatenate([np.asarray(b, dtype=np.int64).reshape(-1, 4) for b in bboxes]) if offs[-1] else np.zeros((1, 4), np.int64)
    if bx.shape[0] != ids.shape[0]:
        raise ValueError("one box per token id")
    return ids, bx, offs


def collate_pad(input_ids: Sequence[Sequence[int]], bboxes: Sequence[Any], max_length: int = 512, pad_id: int = 1, device=None):
    """Ragged token ids / boxes -> (input_ids, attention_mask, bbox) of shape (B,T) / (B,T,4), padded on the device
    (one-shot form, synchronous copies from pageable memory)."""
    dev = _require_torch_cuda(device)
    ids, bx, offs = _flatten_tokens(input_ids, bboxes)
    d_ids, d_bx, d_off = (torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (ids, bx, offs))
    return _collate_on_device(d_ids, d_bx, d_off, len(input_ids), max_length, pad_id, dev)


class _Slot:
    """One staging slot: a pinned host buffer, its device twin and the event of the last copy out of the host buffer."""

    def __init__(self, dev, stream):
        self.dev, self.stream = dev, stream
        self.host = None          # pinned uint8 tensor
        self.devbuf = None
        self.copied = None        # torch.cuda.Event: the H2D copy that last read ``host``

    def reserve(self, nbytes: int):
        if self.copied is not None:
            self.copied.synchronize()              # the previous batch of this slot has left the pinned buffer
        if self.host is None or self.host.numel() < nbytes:
            cap = int(nbytes * 1.25) + 4096
            self.host = torch.empty((cap,), dtype=torch.uint8, pin_memory=True)
            with torch.cuda.stream(self.stream):          # owned by the side stream: only its kernels read it
                self.devbuf = torch.empty((cap,), dtype=torch.uint8, device=self.dev)
        return self.host.numpy()


class DeviceFeeder:
    """Double-buffered feed: ``for batch in DeviceFeeder(samples, batch_size): model.early_exit(**batch)``.

    ``samples`` yields dicts with ``image`` (uint8 (H,W) or (H,W,3)), ``input_ids`` (ids incl. <s> ... </s>), ``bbox`` ((n,4))
    and optionally ``labels``.  Batch i+1 is packed into one of two pinned host slots ([pages | descriptors | offsets | ids |
    boxes | labels], one contiguous region), crosses PCIe as ONE ``non_blocking`` copy on a side stream and is resized /
    normalised / padded there, while the caller's stream works on batch i; the yielded tensors are safe to use on the
    caller's current stream.  Replaces the reference's per-batch host preprocessing + ``.to(device)`` loop
    (EE/utils.py:93-98, 169-173)."""

    def __init__(self, samples: Iterable[Dict[str, Any]], batch_size: int, size: int = 224, max_length: int = 512,
                 pad_id: int = 1, device=None, workers: int = 4):
        self.samples, self.bs, self.size, self.T, self.pad_id = samples, batch_size, size, max_length, pad_id
        self.dev = _require_torch_cuda(device)
        self.stream = torch.cuda.Stream(device=self.dev)
        # page copies into the pinned slot are plain memcpy (numpy releases the GIL): a few threads keep the host side of
        # the pipeline ahead of the GPU (0.76 MB per RVL-CDIP page)
        self.pool = None
        if workers > 1:
            from concurrent.futures import ThreadPoolExecutor
            self.pool = ThreadPoolExecutor(max_workers=workers)
        self.workers = max(1, workers)
        self.slots = [_Slot(self.dev, self.stream), _Slot(self.dev, self.stream)]
        self.n_staged = 0
        self.bytes_h2d = 0

    def _stage(self, chunk: List[Dict[str, Any]]):
        B = len(chunk)
        imgs = []
        for s in chunk:
            a = np.asarray(s["image"])
            if a.dtype != np.uint8:
                raise ValueError("images must be uint8")
            if a.ndim == 3 and a.shape[2] == 1:
                a = a[:, :, 0]
            if a.ndim not in (2, 3) or (a.ndim == 3 and a.shape[2] != 3):
                raise ValueError("images must be (H,W) or (H,W,3) uint8")
            imgs.append(a)
        ids, bx, offs = _flatten_tokens([s["input_ids"] for s in chunk], [s["bbox"] for s in chunk])
        labels = np.array([int(s["labels"]) for s in chunk], dtype=np.int64) if "labels" in chunk[0] else None
        # region layout (every region 16-byte aligned)
        al = lambda n: (n + 15) & ~15
        desc = np.zeros(B, dtype=_DESC)
        off = 0
        for i, a in enumerate(imgs):
            desc[i] = (off, a.shape[0], a.shape[1], 1 if a.ndim == 2 else 3, 0)
            off = al(off + a.size)
        _check_sizes(desc, self.size)
        o_desc = off
        o_off = al(o_desc + desc.nbytes)
        o_ids = al(o_off + offs.nbytes)
        o_bx = al(o_ids + ids.nbytes)
        o_lab = al(o_bx + bx.nbytes)
        total = al(o_lab + (labels.nbytes if labels is not None else 0))
        slot = self.slots[self.n_staged & 1]
        self.n_staged += 1
        host = slot.reserve(total)
        def copy_pages(lo, hi):
            for i in range(lo, hi):
                a = imgs[i]
                o = int(desc[i]["offset"])
                np.copyto(host[o:o + a.size].reshape(a.shape), a)
        if self.pool is not None and B >= 2 * self.workers:
            step = (B + self.workers - 1) // self.workers
            for f in [self.pool.submit(copy_pages, lo, min(B, lo + step)) for lo in range(0, B, step)]:
                f.result()
        else:
            copy_pages(0, B)
        host[o_desc:o_desc + desc.nbytes] = desc.view(np.uint8).reshape(-1)
        host[o_off:o_off + offs.nbytes] = offs.view(np.uint8)
        host[o_ids:o_ids + ids.nbytes] = ids.view(np.uint8)
        host[o_bx:o_bx + bx.nbytes] = np.ascontiguousarray(bx).view(np.uint8).reshape(-1)
        if labels is not None:
            host[o_lab:o_lab + labels.nbytes] = labels.view(np.uint8)
        self.bytes_h2d += total
        with torch.cuda.stream(self.stream):
            d = slot.devbuf[:total]
            d.copy_(slot.host[:total], non_blocking=True)                 # pinned -> device, asynchronous
            slot.copied = torch.cuda.Event()
            slot.copied.record(self.stream)
            px = _preprocess_on_device(d[:o_desc], d[o_desc:o_desc + desc.nbytes], B, self.size, int(desc["h"].max()), self.dev)
            t_ids, t_am, t_bb = _collate_on_device(d[o_ids:o_ids + ids.nbytes], d[o_bx:o_bx + bx.nbytes], d[o_off:o_off + offs.nbytes],
                                                   B, self.T, self.pad_id, self.dev)
            batch = {"input_ids": t_ids, "attention_mask": t_am, "bbox": t_bb, "pixel_values": px}
            if labels is not None:
                batch["labels"] = d[o_lab:o_lab + labels.nbytes].view(torch.int64).clone()
            ev = torch.cuda.Event()
            ev.record(self.stream)
        return batch, ev

    def __iter__(self) -> Iterator[Dict[str, Any]]:
        it = iter(self.samples)

        def take():
            chunk = []
            for s in it:
                chunk.append(s)
                if len(chunk) == self.bs:
                    break
            return chunk

        chunk = take()
        pending = self._stage(chunk) if chunk else None
        while pending is not None:
            batch, ev = pending
            nxt = take()
            pending = self._stage(nxt) if nxt else None      # enqueue batch i+1 before handing out batch i
            torch.cuda.current_stream(self.dev).wait_event(ev)
            for t in batch.values():
                t.record_stream(torch.cuda.current_stream(self.dev))
            yield batch
