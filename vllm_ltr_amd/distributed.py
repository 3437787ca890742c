"""Request-level data parallelism of the ranker over the GPUs of one node.

The reference runs the predictor tensor-parallel over the backbone's TP group
(vllm/engine/llm_engine.py:237): two all-reduces of [T, H] per layer
(vllm/model_executor/layers/linear.py:577), an embedding all-reduce
(layers/vocab_parallel_embedding.py:105) and a logits gather
(layers/logits_processor.py:67).  Requests are independent (causal attention inside a
prompt only), so here the *batch* is sharded instead: predictor weights are replicated,
every rank scores a contiguous, token-balanced slice of the unscored requests, and ONE
collective - an all-gather of f32 scores (RCCL over xGMI; <= 32 KiB per rank at a 64k
queue) - gives every rank the full score vector; each rank then runs the same
deterministic rank step.  One process per GPU, ``torch.distributed`` (backend "nccl" =
RCCL on ROCm; "gloo" in the CPU tests).

All ranks derive the shard map from the same ``cu_seqlens`` and take the
shard-or-not decision from the same N, so the collective can never be mismatched
(SURVEY.md section 5 'failure detection').
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np
import torch


def shard_bounds(cu_seqlens: np.ndarray, world: int) -> List[Tuple[int, int]]:
    """Contiguous request ranges [r0, r1) per rank with ~T/world tokens each (greedy
    prefix split on the cumulative token counts).  Every request lands in exactly one
    shard; shards may be empty when there are fewer requests than ranks."""
    cu = np.asarray(cu_seqlens, dtype=np.int64)
    n = cu.shape[0] - 1
    T = int(cu[-1])
    cuts = [0]
    for r in range(1, world):
        target = (T * r) // world
        # first request whose START is >= target keeps shards contiguous and balanced
        idx = int(np.searchsorted(cu[:-1], target, side="left"))
        cuts.append(min(max(idx, cuts[-1]), n))
    cuts.append(n)
    return [(cuts[i], cuts[i + 1]) for i in range(world)]


def gather_scores(local: torch.Tensor, counts: Sequence[int], group=None) -> torch.Tensor:
    """All-gather variable-length f32 score shards: pad to the longest shard, one
    ``all_gather_into_tensor``, drop the padding.  ``counts[r]`` = shard length of rank r
    (known to every rank)."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    assert len(counts) == world
    mx = max(max(counts), 1)
    buf = torch.zeros(mx, dtype=torch.float32, device=local.device)
    buf[:local.numel()] = local
    out = torch.empty(world * mx, dtype=torch.float32, device=local.device)
    dist.all_gather_into_tensor(out, buf, group=group)
    out = out.view(world, mx)
    return torch.cat([out[r, :counts[r]] for r in range(world)])


class ShardedScorer:
    """SPMD wrapper: every rank calls :meth:`score` with the SAME (ids, cu_seqlens) and
    gets the full score vector back.

    score_fn(ids_shard, cu_shard) -> f32 tensor [n_shard] on this rank's device (the
    HIP predictor in production: ``lambda i, c: scorer.score_device(...)``).
    min_requests_to_shard: below it the launch + collective latency outweighs the split
    (north_star: 'only when the queue exceeds a single GPU's batch'); rank 0 scores alone
    and broadcasts.
    """

    def __init__(self, score_fn: Callable[[np.ndarray, np.ndarray], torch.Tensor], device,
                 group=None, min_requests_to_shard: int = 1024):
        import torch.distributed as dist
        self.dist = dist
        self.score_fn = score_fn
        self.device = torch.device(device)
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.min_requests_to_shard = int(min_requests_to_shard)

    def score(self, ids: np.ndarray, cu_seqlens: np.ndarray) -> torch.Tensor:
        cu = np.asarray(cu_seqlens, dtype=np.int64)
        n = cu.shape[0] - 1
        if n <= 0:
            return torch.zeros(0, dtype=torch.float32, device=self.device)
        if self.world == 1:
            return self.score_fn(ids, cu.astype(np.int32))
        if n < self.min_requests_to_shard:            # same decision on every rank (same n)
            if self.rank == 0:
                s = self.score_fn(ids, cu.astype(np.int32)).to(self.device, torch.float32)
            else:
                s = torch.empty(n, dtype=torch.float32, device=self.device)
            self.dist.broadcast(s, src=self.dist.get_global_rank(self.group, 0) if self.group else 0,
                                group=self.group)
            return s
        bounds = shard_bounds(cu, self.world)
        r0, r1 = bounds[self.rank]
        if r1 > r0:
            ids_s = np.asarray(ids)[cu[r0]:cu[r1]]
            cu_s = (cu[r0:r1 + 1] - cu[r0]).astype(np.int32)
            local = self.score_fn(ids_s, cu_s).to(self.device, torch.float32)
        else:
            local = torch.zeros(0, dtype=torch.float32, device=self.device)
        return gather_scores(local, [b - a for a, b in bounds], self.group)
