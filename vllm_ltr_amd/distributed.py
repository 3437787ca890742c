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


def gather_scores(local: torch.Tensor, counts: Sequence[int], group=None,
                  out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """All-gather variable-length f32 score shards: pad to the longest shard, one
    ``all_gather_into_tensor``, drop the padding.  ``counts[r]`` = shard length of rank r
    (known to every rank).  ``out`` f32 [sum(counts)]: the compaction writes straight into it (the queue's
    score slots) instead of a fresh tensor that would have to be copied there."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    assert len(counts) == world
    mx = max(max(counts), 1)
    buf = torch.zeros(mx, dtype=torch.float32, device=local.device)
    buf[:local.numel()] = local
    if local.is_cuda and dist.get_backend(group) != "nccl":
        # gloo has no CUDA all-gather: stage through the host.  Only the one-device dry run of the N-rank path
        # (tests, LTR_BENCH_BACKEND=gloo) comes here; production is RCCL ("nccl") on device buffers.
        out_h = torch.empty(world * mx, dtype=torch.float32)
        dist.all_gather_into_tensor(out_h, buf.cpu(), group=group)
        out_g = out_h.to(local.device)
    else:
        out_g = torch.empty(world * mx, dtype=torch.float32, device=local.device)
        dist.all_gather_into_tensor(out_g, buf, group=group)
    g = out_g.view(world, mx)
    parts = [g[r, :counts[r]] for r in range(world)]
    if out is not None:
        return torch.cat(parts, out=out)
    return torch.cat(parts)


class ShardedScorer:
    """SPMD wrapper: every rank calls :meth:`score` / :meth:`score_device` with the SAME batch and
    gets the full score vector back.

    scorer: the HIP predictor (:class:`~vllm_ltr_amd.scorer.HipOPTScorer`; anything with
    ``score_device(ids_dev, cu_dev, cu_host)``), or a plain ``score_fn(ids_shard, cu_shard) -> f32
    tensor [n_shard]`` over host arrays (the CPU tests put the oracle there).
    min_requests_to_shard: below it the launch + collective latency outweighs the split
    (north_star: 'only when the queue exceeds a single GPU's batch'); rank 0 scores alone
    and broadcasts.
    """

    def __init__(self, scorer, device, group=None, min_requests_to_shard: int = 1024):
        import torch.distributed as dist
        self.dist = dist
        self.scorer = scorer if hasattr(scorer, "score_device") else None
        self.score_fn: Optional[Callable[[np.ndarray, np.ndarray], torch.Tensor]] = None if self.scorer else scorer
        self.device = torch.device(device)
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.min_requests_to_shard = int(min_requests_to_shard)

    def any_rank(self, flag: bool) -> bool:
        """True on every rank iff ``flag`` is true on at least one (one 4-byte all-reduce): lets all ranks of an SPMD
        call agree on an error before any of them raises."""
        if self.world == 1:
            return bool(flag)
        on_dev = self.dist.get_backend(self.group) == "nccl"
        t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=self.device if on_dev else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX, group=self.group)
        return bool(int(t.item()))

    # ---- the collective part (same on both entry points)
    def _exchange(self, n: int, sharded: bool, bounds, local_fn, whole_fn, out=None) -> torch.Tensor:
        if self.world == 1:
            return whole_fn(out)
        if not sharded:                                  # same decision on every rank (same n)
            if self.rank == 0:
                s = whole_fn(out).to(self.device, torch.float32)
            else:
                s = out if out is not None else torch.empty(n, dtype=torch.float32, device=self.device)
            src = self.dist.get_global_rank(self.group, 0) if self.group else 0
            if s.is_cuda and self.dist.get_backend(self.group) != "nccl":     # one-device dry run (see gather_scores)
                h = s.cpu()
                self.dist.broadcast(h, src=src, group=self.group)
                s.copy_(h)
            else:
                self.dist.broadcast(s, src=src, group=self.group)
            return s
        r0, r1 = bounds[self.rank]
        if r1 > r0:
            local = local_fn(r0, r1).to(self.device, torch.float32)
        else:
            local = torch.zeros(0, dtype=torch.float32, device=self.device)
        return gather_scores(local, [b - a for a, b in bounds], self.group, out=out)

    def _local_host(self, ids, cu):
        if self.scorer is not None:
            ids_d = torch.from_numpy(np.ascontiguousarray(ids, dtype=np.int64)).to(self.device)
            cu32 = np.ascontiguousarray(cu, dtype=np.int32)
            return self.scorer.score_device(ids_d, torch.from_numpy(cu32).to(self.device), cu32)
        return self.score_fn(ids, cu.astype(np.int32))

    def score(self, ids: np.ndarray, cu_seqlens: np.ndarray) -> torch.Tensor:
        """Host arrays in (identical on every rank); each rank uploads and scores only its shard."""
        cu = np.asarray(cu_seqlens, dtype=np.int64)
        n = cu.shape[0] - 1
        if n <= 0:
            return torch.zeros(0, dtype=torch.float32, device=self.device)
        sharded = n >= self.min_requests_to_shard
        bounds = shard_bounds(cu, self.world) if sharded and self.world > 1 else None
        ids = np.asarray(ids)
        return self._exchange(n, sharded, bounds,
                              lambda r0, r1: self._local_host(ids[cu[r0]:cu[r1]], cu[r0:r1 + 1] - cu[r0]),
                              lambda out=None: self._local_host(ids, cu))

    def score_device(self, ids_dev: torch.Tensor, cu_dev: torch.Tensor, cu_host: np.ndarray,
                     out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """The whole batch is resident on every rank's device (ids int64 [T], cu int32 [N+1] + host mirror);
        each rank scores its slice of it in place - no copies, one all-gather of the scores.  ``out`` f32 [N] on the
        device: where the gathered scores go (e.g. the queue's score slots)."""
        assert self.scorer is not None, "score_device needs a device scorer"
        cu = np.asarray(cu_host, dtype=np.int64)
        n = cu.shape[0] - 1
        if n <= 0:
            return torch.zeros(0, dtype=torch.float32, device=self.device)
        sharded = n >= self.min_requests_to_shard
        bounds = shard_bounds(cu, self.world) if sharded and self.world > 1 else None

        def local(r0, r1):
            t0 = int(cu[r0])
            cu_s = (cu[r0:r1 + 1] - t0).astype(np.int32)
            cu_d = cu_dev[r0:r1 + 1] - t0 if t0 else cu_dev[r0:r1 + 1]
            return self.scorer.score_device(ids_dev[t0:int(cu[r1])], cu_d.contiguous(), cu_s)
        return self._exchange(n, sharded, bounds, local,
                              lambda out=None: self.scorer.score_device(ids_dev, cu_dev, np.ascontiguousarray(cu_host, np.int32),
                                                                        out=out), out=out)
