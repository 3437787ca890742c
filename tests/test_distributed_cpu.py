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


def _worker_timeout(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import time
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from vllm_ltr_amd.distributed import PeerTimeout, ShardedScorer
    sh = ShardedScorer(lambda i, c: torch.zeros(len(c) - 1), "cpu", min_requests_to_shard=1, timeout_s=2.0)
    # both ranks agree on a status code (MAX over the ranks): rank 1 reports 2 (range), rank 0 nothing
    code = sh.agree_status(2 if rank == 1 else 0)
    if rank == 1:                      # ... then rank 1 "dies": it never enters the scoring call's collective
        q.put((rank, code, "absent"))
        q.close(); q.join_thread()     # flush before the hard exit below
        time.sleep(8)
        os._exit(0)
    t0 = time.time()
    try:
        sh.score(np.arange(8, dtype=np.int64), np.array([0, 4, 8], np.int32))
        q.put((rank, code, "returned"))
    except PeerTimeout as e:
        q.put((rank, code, f"PeerTimeout after {time.time() - t0:.1f}s"))
    q.close(); q.join_thread()
    os._exit(0)                        # (no clean teardown with a dead peer)


def test_dead_peer_raises_within_the_timeout_and_status_codes_agree():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_timeout, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
    assert res[0][1] == 2 and res[1][1] == 2               # the code of the failing rank reached both
    assert res[0][2].startswith("PeerTimeout") and res[1][2] == "absent", res


def test_shard_decision_is_a_token_rule_by_default():
    """north_star: shard 'only when the queue exceeds a single GPU's batch' = more tokens than one 196,608-token pass."""
    from vllm_ltr_amd.distributed import ONE_PASS_TOKENS, ShardedScorer

    class G:                            # a 4-rank group without processes: only the decision logic is used
        pass
    sh = ShardedScorer.__new__(ShardedScorer)
    sh.world, sh.min_requests_to_shard, sh.min_tokens_to_shard = 4, None, ONE_PASS_TOKENS
    assert not sh.shards(2000, ONE_PASS_TOKENS) and sh.shards(2000, ONE_PASS_TOKENS + 1) and not sh.shards(0, 10**9)
    sh.min_requests_to_shard = 64
    assert sh.shards(64, 100) and not sh.shards(63, 10**9)
    sh.world = 1
    assert not sh.shards(10**6, 10**9)
