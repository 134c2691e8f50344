"""Device-side input feed: what sits between the dataset and ``model.forward``.

In the reference the LayoutLMv3 processor resizes / rescales / normalises every page on the host (PIL + numpy,
EE/data/RVL_CDIP.py:246-262), ``DataCollatorWithPadding(padding="max_length")`` pads on the host (EE/utils.py:93-98) and the
finished float tensors (602 KB per page) cross PCIe one batch at a time (EE/utils.py:173).  Here the raw uint8 page and the
ragged token ids cross PCIe (a greyscale 1000x762 page is 0.76 MB, usually less than its float tensor... and an "L" page
needs no RGB expansion), and resize + normalise + padding run on the GPU (ee_preprocess_images / ee_collate_pad) on a side
stream, double-buffered through pinned memory so that the copy and the preprocessing of batch i+1 overlap the model on batch i.
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


def preprocess_images(images: Sequence[np.ndarray], size: int = 224, device=None, return_u8: bool = False):
    """uint8 pages -> pixel_values (B,3,size,size) float32 on the device, identical to the HF/PIL pipeline."""
    lib = capi.load()
    dev = _require_torch_cuda(device)
    buf, desc = pack_images(images)
    B = len(images)
    max_h = int(desc["h"].max())
    if max(desc["h"].max(), desc["w"].max()) > MAX_RATIO * size:
        raise ValueError(f"image side / {size} must be <= {MAX_RATIO}")
    d_img = torch.from_numpy(buf).to(dev)
    d_desc = torch.from_numpy(desc.view(np.uint8).reshape(-1)).to(dev)
    ws_bytes = lib.ee_preprocess_workspace_bytes(B, size, max_h)
    ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
    out = torch.empty((B, 3, size, size), dtype=torch.float32, device=dev)
    u8 = torch.empty((B, size, size, 3), dtype=torch.uint8, device=dev) if return_u8 else None
    p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    with torch.cuda.device(dev):
        capi.check(lib.ee_preprocess_images(p(d_img), p(d_desc), B, size, max_h, p(ws), ws_bytes, p(out), p(u8), _stream_ptr()),
                   None, "ee_preprocess_images")
    return (out, u8) if return_u8 else out


def collate_pad(input_ids: Sequence[Sequence[int]], bboxes: Sequence[Any], max_length: int = 512, pad_id: int = 1, device=None):
    """Ragged token ids / boxes -> (input_ids, attention_mask, bbox) of shape (B,T) / (B,T,4), padded on the device."""
    lib = capi.load()
    dev = _require_torch_cuda(device)
    B = len(input_ids)
    lens = np.array([len(x) for x in input_ids], dtype=np.int64)
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    ids = np.concatenate([np.asarray(x, dtype=np.int64).reshape(-1) for x in input_ids]) if offs[-1] else np.zeros(1, np.int64)
    bx = np.concatenate([np.asarray(b, dtype=np.int64).reshape(-1, 4) for b in bboxes]) if offs[-1] else np.zeros((1, 4), np.int64)
    if bx.shape[0] != ids.shape[0]:
        raise ValueError("one box per token id")
    d_ids, d_bx, d_off = (torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (ids, bx, offs))
    T = max_length
    o_ids = torch.empty((B, T), dtype=torch.int64, device=dev)
    o_am = torch.empty((B, T), dtype=torch.int64, device=dev)
    o_bb = torch.empty((B, T, 4), dtype=torch.int64, device=dev)
    p = lambda t: C.c_void_p(t.data_ptr())
    with torch.cuda.device(dev):
        capi.check(lib.ee_collate_pad(p(d_ids), p(d_bx), p(d_off), B, T, pad_id, p(o_ids), p(o_am), p(o_bb), _stream_ptr()), None,
                   "ee_collate_pad")
    return o_ids, o_am, o_bb


class DeviceFeeder:
    """Double-buffered feed: ``for batch in DeviceFeeder(samples, batch_size): model.early_exit(**batch)``.

    ``samples`` yields dicts with ``image`` (uint8 (H,W) or (H,W,3)), ``input_ids`` (ids incl. <s> ... </s>), ``bbox`` ((n,4))
    and optionally ``labels``.  Batch i+1 is packed into pinned memory, copied and preprocessed on a side stream while the
    caller's stream works on batch i; the yielded tensors are safe to use on the caller's current stream."""

    def __init__(self, samples: Iterable[Dict[str, Any]], batch_size: int, size: int = 224, max_length: int = 512,
                 pad_id: int = 1, device=None):
        self.samples, self.bs, self.size, self.T, self.pad_id = samples, batch_size, size, max_length, pad_id
        self.dev = _require_torch_cuda(device)
        self.stream = torch.cuda.Stream(device=self.dev)

    def _stage(self, chunk: List[Dict[str, Any]]):
        with torch.cuda.stream(self.stream):
            px = preprocess_images([s["image"] for s in chunk], self.size, self.dev)
            ids, am, bb = collate_pad([s["input_ids"] for s in chunk], [s["bbox"] for s in chunk], self.T, self.pad_id, self.dev)
            batch = {"input_ids": ids, "attention_mask": am, "bbox": bb, "pixel_values": px}
            if "labels" in chunk[0]:
                batch["labels"] = torch.as_tensor([int(s["labels"]) for s in chunk], dtype=torch.int64).to(self.dev, non_blocking=True)
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
