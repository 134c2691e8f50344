"""Data-parallel sharding of the path over the GPUs of one node.

Every document is independent in the forward pass and in the policy (EE/policy.py:28-45 loops per sample; thresholds
and temperatures are precomputed inputs), so the path shards with NO data-path collective: one process per GPU, each
owning a round-robin slice of the documents (interleaved so that expected exit depth is balanced across ranks), the
weights replicated.  The single collective is one all-gather of the per-document results
``[logits (K) f32 | exit_layer i32 | confidence f32]`` (carried as int32 words) = 4 (K + 2) bytes per document at the end (RCCL over xGMI on GPUs: backend "nccl"; "gloo" in CPU tests).
The reference has no counterpart (its ``--data-parallel`` flag, EE/configs.py:116-121, is never read).
"""
from __future__ import annotations

from typing import Callable, Optional

import numpy as np

from .engine import torch


def shard_indices(n_docs: int, rank: int, world: int) -> np.ndarray:
    """Documents of ``rank``: rank, rank + world, rank + 2*world, ..."""
    return np.arange(rank, n_docs, world, dtype=np.int64)


def shard_size(n_docs: int, rank: int, world: int) -> int:
    return (n_docs - rank + world - 1) // world if n_docs > rank else 0


def pack_results(logits, exit_layer, confidence):
    """(n, K) float32, (n,) int32, (n,) float32 -> (n, K+2) rows of 4-byte words for the ONE all-gather of the north star.  The transport
    dtype of the whole row is INT32 (round 6): logits and confidence travel as their float32 bit patterns (views, not
    conversions), the exit index as the int32 it is.  Integer copies, concatenations and collectives are never flushed, canonicalised or
    rounded (round 5 carried the index as a float32 bit pattern, where small indices are denormals and negative ones NaN payloads that any
    float arithmetic would have destroyed), and float arithmetic on a packed row is now a visible dtype mistake.  The gathered contract is
    (f32 logits, i32 exit_layer, f32 confidence), which ``unpack_results`` returns with those dtypes."""
    lg = logits.to(torch.float32).contiguous().view(torch.int32)
    cf = confidence.to(torch.float32).contiguous().view(torch.int32)
    return torch.cat([lg, exit_layer.to(torch.int32).unsqueeze(1), cf.unsqueeze(1)], dim=1)


def unpack_results(rows):
    """(n, K+2) packed int32 rows -> (logits float32 (n,K), exit_layer int32 (n,), confidence float32 (n,))."""
    if rows.dtype != torch.int32:
        raise TypeError(f"packed result rows are int32 words (pack_results), got {rows.dtype}")
    K = rows.shape[1] - 2
    return rows[:, :K].contiguous().view(torch.float32), rows[:, K].contiguous(), rows[:, K + 1].contiguous().view(torch.float32)


def _collective_device(t, group=None):
    """Where a collective's payload must live: RCCL ("nccl") moves device memory over xGMI; "gloo" (CPU tests, or two ranks
    rehearsing the flow on one GPU) moves host memory."""
    import torch.distributed as dist
    return t.device if dist.get_backend(group) == "nccl" else torch.device("cpu")


def all_gather_results(local_rows, n_docs: int, rank: int, world: int, group=None, always_collective: bool = False):
    """One all-gather of every rank's ``(n_local, C)`` rows; returns ``(n_docs, C)`` in original document order.
    Shards may differ by one row: each rank pads to the largest shard so a single fixed-size collective suffices.
    ``always_collective``: issue the collective even at world size 1 (tests/test_gpu_api.py runs the job's exact RCCL calls on the one
    GPU the build pool offers)."""
    import torch.distributed as dist
    if world == 1 and not always_collective:
        return local_rows
    n_max = shard_size(n_docs, 0, world)
    C = local_rows.shape[1]
    if local_rows.shape[0] != shard_size(n_docs, rank, world):
        raise ValueError(f"rank {rank} holds {local_rows.shape[0]} rows, expected {shard_size(n_docs, rank, world)}")
    cdev = _collective_device(local_rows, group)
    buf = torch.zeros((n_max, C), dtype=local_rows.dtype, device=cdev)
    buf[:local_rows.shape[0]] = local_rows.to(cdev)
    out = torch.empty((world * n_max, C), dtype=local_rows.dtype, device=cdev)
    dist.all_gather_into_tensor(out, buf, group=group)
    # rank r, local row i  <->  document r + i * world
    res = torch.empty((n_docs, C), dtype=local_rows.dtype, device=cdev)
    for r in range(world):
        n_r = shard_size(n_docs, r, world)
        if n_r:
            res[r::world][:n_r] = out[r * n_max:r * n_max + n_r]
    return res.to(local_rows.device)


def run_sharded(run_local: Callable, n_docs: int, rank: int, world: int, group=None):
    """``run_local(doc_indices) -> (n_local, K+2) rows`` on this rank's shard, then the one all-gather."""
    idx = shard_indices(n_docs, rank, world)
    rows = run_local(idx)
    return all_gather_results(rows, n_docs, rank, world, group)


def broadcast_array(a: np.ndarray, src: int = 0, device=None, group=None) -> np.ndarray:
    """Every rank gets rank ``src``'s array (thresholds / temperatures are inputs of the path, EE/policy.py:12-24: every
    shard must test its documents against the same values)."""
    import torch.distributed as dist
    t = torch.from_numpy(np.array(a, copy=True, order="C"))      # a copy: the collective writes into its buffer, the caller's array stays
    if dist.get_backend(group) == "nccl":
        t = t.to(device)
    dist.broadcast(t, src, group=group)
    return t.cpu().numpy()


def max_over_ranks(v: float, device=None, group=None) -> float:
    import torch.distributed as dist
    t = torch.tensor([v], dtype=torch.float64)
    if dist.get_backend(group) == "nccl":
        t = t.to(device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())
