"""CPU, world_size 2, gloo: the sharded (N > 1 GPU) ranking path - shard map, the score
all-gather and the rank step on the gathered queue - with the oracle standing in for the
per-rank HIP predictor (the collective logic is backend-independent)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from vllm_ltr_amd.distributed import shard_bounds


def test_shard_bounds_properties():
    r = np.random.RandomState(0)
    for n, world in [(1, 2), (2, 2), (5, 8), (100, 2), (1000, 8), (8192, 8), (7, 1)]:
        lens = r.randint(1, 300, n)
        cu = np.concatenate([[0], np.cumsum(lens)])
        b = shard_bounds(cu, world)
        assert len(b) == world and b[0][0] == 0 and b[-1][1] == n
        assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))       # contiguous, disjoint, complete
        assert all(x[0] <= x[1] for x in b)
        if n >= 16 * world:                                                  # token balance
            tok = [cu[y] - cu[x] for x, y in b]
            assert max(tok) - min(tok) <= 2 * lens.max()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n, min_shard, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(1)
        from oracle import rank_step as rs
        from oracle.opt_scorer import OracleOPTScorer
        from util import synthetic_batch
        from vllm_ltr_amd.distributed import ShardedScorer
        from vllm_ltr_amd.opt_spec import OPTSpec, seeded_checkpoint
        spec = OPTSpec.tiny_pre_ln()
        orc = OracleOPTScorer(spec, seeded_checkpoint(spec, 3))
        lens = np.random.RandomState(1).randint(1, 40, n).tolist()
        ids, cu = synthetic_batch(spec, lens, 2)                 # identical on every rank (SPMD)
        calls = []

        def score_fn(i, c):
            calls.append(len(c) - 1)
            return torch.from_numpy(orc.score(i, c))
        sh = ShardedScorer(score_fn, "cpu", min_requests_to_shard=min_shard)
        scores = sh.score(ids, cu).numpy()
        full = orc.score(ids, cu)
        # every rank ends with the full vector, identical to the unsharded scores
        ok = bool(np.abs(scores - full).max() < 1e-6)
        # and therefore the same permutation from the deterministic rank step
        pri = np.zeros(n, np.int32); idle = np.zeros(n, np.int32); runs = np.zeros(n, np.int32)
        perm = rs.rank_step_np(scores, pri, idle, runs, 3, 2)
        gathered = [None] * world
        dist.all_gather_object(gathered, (perm.tolist(), calls))
        same_perm = all(g[0] == gathered[0][0] for g in gathered)
        q.put((rank, ok, same_perm, calls))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n,min_shard", [(64, 16), (7, 16), (3, 1)])
def test_two_rank_gloo_sharded_scoring(n, min_shard):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, min_shard, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res.sort()
    for rank, ok, same_perm, calls in res:
        assert ok and same_perm, (rank, ok, same_perm)
    if n >= min_shard:          # sharded: both ranks scored a proper part
        assert sum(r[3][0] for r in res) == n and all(0 < r[3][0] < n for r in res)
    else:                       # below the threshold only rank 0 scores, then broadcasts
        assert res[0][3] == [n] and res[1][3] == []
