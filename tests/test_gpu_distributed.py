"""-m gpu: the N-rank ranking path with the REAL HIP predictor, two processes sharing the one device of the
GPU box (gloo for the control plane; production uses RCCL with one device per rank): shard map, per-rank scoring
of its slice, score all-gather, identical rank step on every rank - gathered scores must equal the
single-process scores (to the 2e-6 a score may move between the GEMM kernels launch_gemm picks for a shard and for the whole
queue; every rank holds the same gathered bits)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _same(a, b):
    """A shard is a smaller batch than the whole queue, and launch_gemm picks its GEMM kernel (tile, split-K) per launch
    from the row count: a request's score may move by f32 rounding between the two (<= 2e-6, the bound
    tests/test_gpu_small_batches.py holds the regimes to); every rank sees the SAME gathered bits."""
    return a.shape == b.shape and float(np.abs(a - b).max()) <= 2e-6 * max(1.0, float(np.abs(b).max()))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker_125m(rank, world, port, q):
    """BASELINE config 4 in miniature: the OPT-125m predictor, a ShareGPT-profile queue, request-sharded over the ranks."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from util import bench_lengths
        from vllm_ltr_amd.distributed import ShardedScorer, shard_bounds
        from vllm_ltr_amd.opt_spec import OPTSpec, seeded_checkpoint
        from vllm_ltr_amd.rank import DeviceQueue
        from vllm_ltr_amd.scorer import HipOPTScorer
        dev = torch.device("cuda:0")
        spec = OPTSpec.opt_125m()
        sc = HipOPTScorer(spec, seeded_checkpoint(spec, 0), "cuda:0", "f16")
        n = 3000
        lens = bench_lengths(n, seed=7)
        g = torch.Generator().manual_seed(7)
        ids = torch.randint(4, spec.vocab_size, (int(lens.sum()),), generator=g, dtype=torch.int64).numpy()
        cu = np.zeros(n + 1, np.int32); np.cumsum(lens, out=cu[1:]); ids[cu[:-1]] = 2
        ids_d, cu_d = torch.from_numpy(ids).to(dev), torch.from_numpy(cu).to(dev)
        single = sc.score_device(ids_d, cu_d, cu).cpu().numpy() if rank == 0 else None
        got = ShardedScorer(sc, dev, min_requests_to_shard=1024).score_device(ids_d, cu_d, cu)
        b = shard_bounds(cu, world)
        tok = [int(cu[y] - cu[x]) for x, y in b]
        queue = DeviceQueue(dev, starv=200, period=10, capacity=n)
        queue.append(got)
        need = torch.from_numpy(lens.astype(np.int32)).to(dev)
        perm, n_sel, _, _ = queue.step(need, torch.ones(n, dtype=torch.int32, device=dev), 2048, 256)
        gathered = [None] * world
        dist.all_gather_object(gathered, (perm.cpu().numpy().tolist(), int(n_sel.item())))
        same = all(x == gathered[0] for x in gathered)
        ok = True if single is None else bool(_same(got.cpu().numpy(), single))
        q.put((rank, ok, same, tok))
    finally:
        dist.destroy_process_group()


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from collections import deque
        from util import FakeSeqGroup, bench_lengths, synthetic_batch
        from vllm_ltr_amd.distributed import ShardedScorer, shard_bounds
        from vllm_ltr_amd.opt_spec import OPTSpec, seeded_checkpoint
        from vllm_ltr_amd.plugin import MI355XRanker
        from vllm_ltr_amd.rank import DeviceQueue
        from vllm_ltr_amd.scorer import HipOPTScorer
        dev = torch.device("cuda:0")
        spec = OPTSpec.tiny_pre_ln()
        sc = HipOPTScorer(spec, seeded_checkpoint(spec, 3), "cuda:0", "f16")
        n = 700
        lens = bench_lengths(n, seed=5, mu=24.0).clip(1, 150)
        ids, cu = synthetic_batch(spec, lens.tolist(), 6)              # identical on every rank (SPMD)
        ids_d, cu_d = torch.from_numpy(ids).to(dev), torch.from_numpy(cu).to(dev)
        single = sc.score_device(ids_d, cu_d, cu).cpu().numpy()        # what one process computes alone
        sh = ShardedScorer(sc, dev, min_requests_to_shard=64)
        got_dev = sh.score_device(ids_d, cu_d, cu)                     # device-resident batch (bench)
        got_host = sh.score(ids, cu)                                   # host batch (each rank uploads its shard only)
        small = sh.score_device(ids_d[:cu[10]], cu_d[:11], cu[:11])    # below the threshold: rank 0 scores, broadcast
        ok = (_same(got_dev.cpu().numpy(), single) and _same(got_host.cpu().numpy(), single)
              and _same(small.cpu().numpy(), single[:10])
              and np.array_equal(got_dev.cpu().numpy(), got_host.cpu().numpy()))     # the same shards: the same bits
        b = shard_bounds(cu, world)
        # the rank step on the gathered queue is deterministic: same permutation everywhere
        queue = DeviceQueue(dev, starv=3, period=2, capacity=n)
        queue.append(got_dev)
        perm = queue.rank().cpu().numpy().tolist()
        # the plug-in with group=: obtain_aux_scores shards the same way
        ranker = MI355XRanker(sc, "opt-xxx-starv3-period2", max_length=160, group=dist.group.WORLD, min_requests_to_shard=64)
        groups = [FakeSeqGroup(str(i), ids[cu[i]:cu[i + 1]].tolist()) for i in range(n)]

        class Sched:
            pass
        s = Sched()
        s.waiting, s.running, s.swapped = deque(groups), deque(), deque()
        ranker.install(s)
        order = [g.request_id for g in s._get_ordered_requests()]
        plug_ok = _same(np.array([g.aux_model_score for g in groups], np.float32), single)
        gathered = [None] * world
        dist.all_gather_object(gathered, (perm, order))
        same = all(g == gathered[0] for g in gathered)
        q.put((rank, ok, plug_ok, same, b[rank], order[:5] == [str(i) for i in perm[:5]]))
    finally:
        dist.destroy_process_group()


def _worker_driver_mode(rank, world, port, q):
    """The vllm-ltr engine's shape: ONLY the driver (rank 1 here) has requests, a scheduler and a ranker that is asked
    for scores; rank 0 builds the same ranker and sits in serve()."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from collections import deque
        from util import FakeSeqGroup, bench_lengths, synthetic_batch
        from vllm_ltr_amd.opt_spec import OPTSpec, seeded_checkpoint
        from vllm_ltr_amd.plugin import MI355XRanker
        from vllm_ltr_amd.scorer import HipOPTScorer
        spec = OPTSpec.tiny_pre_ln()
        ckpt = seeded_checkpoint(spec, 3)
        driver = 1
        if rank != driver:
            sc = HipOPTScorer(spec, ckpt, "cuda:0", "f16")
            ranker = MI355XRanker(sc, "opt-xxx-starv3-period2", max_length=160, group=dist.group.WORLD, driver_rank=driver,
                                  min_requests_to_shard=64, collective_timeout_s=120.0)
            served = ranker.serve()
            q.put((rank, "worker", served, ranker.scorer.ln_fold))
            return
        n = 700
        lens = bench_lengths(n, seed=5, mu=24.0).clip(1, 150)
        ids, cu = synthetic_batch(spec, lens.tolist(), 6)
        sc = HipOPTScorer(spec, ckpt, "cuda:0", "f16")
        dev = torch.device("cuda:0")
        single = sc.score_device(torch.from_numpy(ids).to(dev), torch.from_numpy(cu).to(dev), cu).cpu().numpy()
        ranker = MI355XRanker(sc, "opt-xxx-starv3-period2", max_length=160, group=dist.group.WORLD, driver_rank=driver,
                              min_requests_to_shard=64, collective_timeout_s=120.0, prescore=True)
        try:
            groups = [FakeSeqGroup(str(i), ids[cu[i]:cu[i + 1]].tolist()) for i in range(n)]

            class Sched:
                pass
            s = Sched()
            s.waiting, s.running, s.swapped = deque(groups[:600]), deque(), deque()
            ranker.install(s)
            order = [g.request_id for g in s._get_ordered_requests()]              # 600 arrivals: sharded, the worker takes part
            collective_calls = ranker._sharded.calls_served
            for g in groups[600:610]:                                              # a few arrivals: scored at arrival / below the
                ranker.add_request(g)                                              # threshold - the worker never hears of them
                s.waiting.append(g)
            order2 = [g.request_id for g in s._get_ordered_requests()]
            small_calls = ranker._sharded.calls_served - collective_calls
            for g in groups[610:]:
                s.waiting.append(g)
            order3 = [g.request_id for g in s._get_ordered_requests()]             # 90 more: sharded again
            got = np.array([g.aux_model_score for g in groups], np.float32)
        finally:
            ranker.close()              # (whatever happens above: the worker's serve() loop must end)
        q.put((rank, "driver", _same(got, single), collective_calls, small_calls, ranker._sharded.calls_served,
               sorted(order3) == sorted(g.request_id for g in groups) and len(order) == 600 and len(order2) == 610))
    finally:
        dist.destroy_process_group()


def test_driver_workers_mode_two_processes_one_device():
    """`MI355XRanker(group=, driver_rank=)`: only the driver calls obtain_aux_scores (VERDICT r4 missing #2;
    ray_gpu_executor.py:440-523, worker.py:234-236, model_runner.py:760-807); the other process serves shards it is SENT
    (header broadcast + one scatter), and a step with a handful of arrivals involves no worker at all.  Scores equal to the
    one-process scores (2e-6: the GEMM kernel a shard's row count selects)."""
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_driver_mode, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    drv = next(r for r in res if r[1] == "driver")
    wrk = next(r for r in res if r[1] == "worker")
    assert drv[2], "driver-mode scores differ from the single-process scores"
    assert drv[3] == 1 and drv[4] == 0 and drv[5] == 2 and drv[6]          # two sharded calls, the small step stayed local
    assert wrk[2] == 2 and wrk[3] is True


def test_two_process_sharded_hip_scoring_on_one_device():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    res.sort()
    for rank, ok, plug_ok, same, bounds, head in res:
        assert ok, f"rank {rank}: gathered scores differ from the single-process scores"
        assert plug_ok, f"rank {rank}: plug-in scores differ"
        assert same and head, f"rank {rank}: ranks disagree on the permutation"
        assert 0 < bounds[1] - bounds[0] < 700
    assert res[0][4][1] == res[1][4][0]          # contiguous shards


def test_config4_shape_two_ranks_opt125m():
    """OPT-125m, 3,000-request ShareGPT-profile queue sharded over two ranks (one device): token-balanced shards, gathered
    scores equal to rank 0's single-process scores (2e-6, `_same`), identical rank step + budget selection on both ranks."""
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_125m, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=900) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank, ok, same, tok in res:
        assert ok and same, (rank, ok, same)
        assert abs(tok[0] - tok[1]) <= 2048          # token-balanced within one maximal request


# ---------------------------------------------------------------------------------------------------------------
# BASELINE config 4 at its size: OPT-125m, the 65,536-request / 5,734,532-token synthetic queue, sharded over
# EIGHT ranks (processes sharing the one device of the box; gloo control plane), score all-gather.
# ---------------------------------------------------------------------------------------------------------------
def _worker_config4(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import datetime
    import hashlib
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(minutes=40))
    try:
        from test_gpu_full_configs import _oracle_scores, _passes, _queue
        from oracle import rank_step as rs
        from vllm_ltr_amd.distributed import ShardedScorer, shard_bounds
        from vllm_ltr_amd.opt_spec import OPTSpec, seeded_checkpoint
        from vllm_ltr_amd.rank import DeviceQueue
        from vllm_ltr_amd.scorer import HipOPTScorer
        dev = torch.device("cuda:0")
        spec = OPTSpec.opt_125m()
        ckpt = seeded_checkpoint(spec, 0)
        sc = HipOPTScorer(spec, ckpt, "cuda:0", "f16")
        n = 65536
        ids, cu, lens = _queue(spec, n, 64.0)                         # bench.py's generator, seed 0 (identical on every rank)
        T = int(cu[-1])
        ids_d, cu_d = torch.from_numpy(ids).to(dev), torch.from_numpy(cu).to(dev)
        queue = DeviceQueue(dev, starv=200, period=10, capacity=n)
        queue.append(torch.zeros(n))
        got = ShardedScorer(sc, dev, min_requests_to_shard=1024).score_device(ids_d, cu_d, cu, out=queue._score[:n])
        assert got.data_ptr() == queue._score.data_ptr()               # gathered straight into the slot array
        need = torch.from_numpy(lens.astype(np.int32)).to(dev)
        perm, n_sel, ran, _ = queue.step(need, torch.ones(n, dtype=torch.int32, device=dev), 2048, 256)
        perm_h = perm.cpu().numpy()
        digest = hashlib.sha256(perm_h.tobytes() + ran.cpu().numpy().tobytes()).hexdigest()
        gathered = [None] * world
        dist.all_gather_object(gathered, (digest, int(n_sel.item())))
        same = all(x == gathered[0] for x in gathered)
        bounds = shard_bounds(cu, world)
        tok = [int(cu[y] - cu[x]) for x, y in bounds]
        res = dict(rank=rank, same=same, T=T, tok=tok, n_sel=int(n_sel.item()))
        if rank == 0:
            scores = got.cpu().numpy()
            single = sc.score_device(ids_d, cu_d, cu).cpu().numpy()   # what ONE process computes for the whole queue
            res["bit_identical"] = bool(_same(scores, single))
            # the oracle on the first and last request of every pass of every shard (+ a few random ones)
            sample = set()
            n_pass = 0
            for r0, r1 in bounds:
                cu_s = cu[r0:r1 + 1] - cu[r0]
                for p0, p1 in _passes(cu_s, 196608):
                    sample.update((r0 + p0, r0 + p1 - 1)); n_pass += 1
            sample.update(np.random.RandomState(4).randint(0, n, 16).tolist())
            sample = np.array(sorted(sample))
            want = _oracle_scores(spec, ckpt, ids, cu, sample)
            res["oracle_err"] = float(np.abs(want - scores[sample]).max())
            res["oracle_n"], res["n_pass"] = int(len(sample)), n_pass
            # the 64k permutation against the literal promote/demote + stable sort of the oracle
            z = np.zeros(n, np.int32)
            res["perm_ok"] = bool(np.array_equal(perm_h, rs.rank_step_np(scores, z.copy(), z.copy(), z.copy(), 200, 10)))
            wn, _ = rs.budget_walk(lens[perm_h], np.ones(n, np.int32), 2048, 256)
            res["budget_ok"] = wn == int(n_sel.item())
            # ... and against the REFERENCE's own run of this queue (tests/golden/config4_opt125m_65536.npz: its Scheduler + fp32
            # predictor on all 65,536 requests, oracle/make_config1_golden.py --config 4full): every gathered score, the HIP sort
            # of the reference's scores, the end-to-end order up to fp32 near-ties, the selected prefix
            zf = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "config4_opt125m_65536.npz"), allow_pickle=False)
            assert hashlib.sha256(np.ascontiguousarray(ids.astype(np.int32)).tobytes()).digest() == zf["ids_sha256"].tobytes()
            assert hashlib.sha256(np.ascontiguousarray(cu.astype(np.int32)).tobytes()).digest() == zf["cu_sha256"].tobytes()
            ref, want = zf["ref_score"], zf["a_order"][0].astype(np.int64)
            err = np.abs(scores.astype(np.float64) - ref.astype(np.float64))
            res["ref_err"], res["ref_rms"] = float(err.max()), float(np.sqrt((err ** 2).mean()))
            q2 = DeviceQueue(dev, starv=200, period=10, capacity=n)
            q2.append(torch.from_numpy(ref))
            perm_ref, n_sel_ref, ran_ref, _ = q2.step(need, torch.ones(n, dtype=torch.int32, device=dev), 2048, 256,
                                                      chunkable=torch.ones(n, dtype=torch.uint8, device=dev))   # (chunked prefill, as the run)
            res["ref_sort_ok"] = bool(np.array_equal(perm_ref.cpu().numpy(), want))
            res["ref_ran_ok"] = sorted(want[:int(n_sel_ref.item())].tolist()) == zf["a_ran"].tolist()
            # an inversion (i before j in the reference's order, j before i in ours) whose reference gap exceeds G exists iff, for
            # some j, a request more than G above it in the reference's scores sits behind it in our order: prefix maximum
            pos = np.empty(n, np.int64); pos[perm_h] = np.arange(n)
            p = pos[want]
            sref = ref.astype(np.float64)[want]                            # non-increasing
            G = 2.0 * res["ref_err"]
            k = np.searchsorted(-sref, -(sref + G), side="left")          # requests [0, k[j]) are more than G above request j
            cm = np.maximum.accumulate(p)
            bad = (k > 0) & (cm[np.maximum(k - 1, 0)] > p)
            res["ref_order_ok"] = not bool(bad.any())
            res["ref_moved"] = int((perm_h != want).sum())
        dist.barrier()
        q.put(res)
    finally:
        dist.destroy_process_group()


def test_config4_opt125m_64k_queue_sharded_over_8_ranks_full_size():
    """BASELINE config 4 at FULL size on one device: 65,536 requests / 5,734,532 tokens through ShardedScorer with 8
    ranks (LTR_TEST_CONFIG4_WORLD overrides): gathered scores equal to the single-process call (2e-6, `_same`), an oracle sample
    (1e-4) with the first and last request of every pass of every shard, the same 64k permutation + budget selection on
    every rank, equal to the oracle's literal sort."""
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    world = int(os.environ.get("LTR_TEST_CONFIG4_WORLD", "8"))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_config4, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=2400) for _ in range(world)]
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    res.sort(key=lambda r: r["rank"])
    r0 = res[0]
    assert r0["T"] == 5734532                                        # the 64k queue of bench.py (seed 0)
    assert all(r["same"] for r in res), "ranks disagree on the permutation / selection"
    assert r0["bit_identical"], "gathered scores differ from the single-process scores"
    print(f"config 4 (OPT-125m, 65536 requests, {r0['T']} tokens, {world} ranks x {r0['n_pass'] // world} passes): oracle "
          f"sample of {r0['oracle_n']} requests, max|d| = {r0['oracle_err']:.3e}; shard tokens {r0['tok']}")
    assert r0["oracle_n"] >= 64 and r0["oracle_err"] <= 1e-4
    assert r0["perm_ok"] and r0["budget_ok"]
    print(f"config 4 against the reference's own run of the 64k queue: max|d| = {r0['ref_err']:.3e} (rms {r0['ref_rms']:.3e}) over all "
          f"65,536 scores; {r0['ref_moved']} positions of the end-to-end order differ, none by more than an fp32 near-tie")
    assert r0["ref_err"] <= 1e-4 and r0["ref_sort_ok"] and r0["ref_ran_ok"] and r0["ref_order_ok"]
    assert max(r0["tok"]) - min(r0["tok"]) <= 2048                   # token-balanced within one maximal request


def test_bench_strong_scaling_mode_two_ranks_one_device():
    """`bench.py --queue-total` (fixed queue, strong scaling) through torch.distributed.run with two ranks on the one
    device (gloo): the line says strong, carries the queue it was given, and two ranks finish the 4,096-request queue."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, LTR_BENCH_ONE_DEVICE="1", LTR_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--queue-total", "4096", "--no-cpu-baseline", "--no-unfused", "--steady-new", "16"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["scaling"] == "strong" and out["n_gpus"] == 2 and out["config"]["queue_total"] == 4096
    assert out["value"] > 0 and out["roofline"]["frac"] > 0 and out["p50_steady_new_latency_ms"]["16"] > 0
    assert 0 < out["config"]["tokens_rank0_shard"] < out["config"]["tokens_total"]


def test_rccl_backend_collectives_single_rank():
    """The production backend ("nccl" = RCCL) on the one GPU of the box.  RCCL refuses two ranks on one device
    ("Duplicate GPU detected"), so the multi-rank tests above run the control plane over gloo; this one runs every
    collective the N-rank path issues - the padded score all-gather of `gather_scores` (device buffers, compaction into
    `out`), the int32 MAX all-reduce of `ShardedScorer.any_rank`, the f64 MAX all-reduce of the bench clock, the
    not-sharded broadcast, the barrier - through RCCL itself with `init_process_group("nccl", device_id=...)` at world
    size 1: the API usage (dtypes, device tensors, in-place outputs) is what a first 8-GPU run would otherwise meet cold."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "diag", "rccl_probe.py"), "1"], capture_output=True,
                       text=True, timeout=600, cwd=root, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0, (r.stdout + r.stderr)[-2000:]
    assert "rank 0/1: gather [0.0, 1.0, 2.0, 3.0, 4.0] max-rank 0 max-f64 1.5 bcast 1.0" in r.stdout, r.stdout[-1000:]
    assert "ok world 1" in r.stdout


def _bench_line(args, env_extra, timeout=1500):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **env_extra)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, capture_output=True, text=True,
                       timeout=timeout, env=env, cwd=root)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    return [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_plain_bench_gpus_2_launches_its_own_ranks():
    """`python bench.py --gpus 2` run PLAINLY - the form the driver uses - must become the 2-rank job by itself
    (re-exec under torch.distributed.run), here as a one-device dry run over gloo; the line names the ranks, the backend
    and every rank's token share; `--scale-table` adds north_star's fixed-queue table as one extra line."""
    lines = _bench_line(["--gpus", "2", "--steps", "1", "--warmup", "1", "--queue", "2048", "--no-cpu-baseline", "--no-unfused",
                         "--no-strong", "--no-scale-points", "--no-class-head", "--steady-new", "0"],
                        dict(LTR_BENCH_ONE_DEVICE="1", LTR_BENCH_BACKEND="gloo"))
    out = lines[-1]
    assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2 and out["backend"] == "gloo" and out["scaling"] == "weak"
    cfg = out["config"]
    assert cfg["queue_total"] == 4096 and cfg["sharded"] and len(cfg["tokens_shard"]) == 2
    assert sum(cfg["tokens_shard"]) == cfg["tokens_total"] and abs(cfg["tokens_shard"][0] - cfg["tokens_shard"][1]) < 2048
    assert out["value"] > 0 and out["roofline"]["hw_frac"] == pytest.approx(2 * out["roofline"]["frac"])


def test_bench_driver_broadcast_two_ranks_one_device():
    """`bench.py --gpus 2 --driver-broadcast`: only rank 0 holds the queue, every timed step distributes the inputs (header +
    scatter) before the shards are scored; the line says so."""
    lines = _bench_line(["--gpus", "2", "--steps", "1", "--warmup", "1", "--queue", "2048", "--no-cpu-baseline", "--no-unfused",
                         "--no-strong", "--no-scale-points", "--no-class-head", "--steady-new", "0", "--driver-broadcast"],
                        dict(LTR_BENCH_ONE_DEVICE="1", LTR_BENCH_BACKEND="gloo"))
    out = lines[-1]
    assert out["n_gpus"] == 2 and out["config"]["sharded"] and out["input_distribution"].startswith("driver-broadcast")
    assert out["value"] > 0 and sum(out["config"]["tokens_shard"]) == out["config"]["tokens_total"]


def test_plain_bench_gpus_1_on_rccl_world_1_with_scale_table():
    """`python bench.py --gpus 1` with LTR_BENCH_BACKEND=nccl: the process group is RCCL itself at world size 1 (init,
    barrier, the clock's MAX all-reduce, teardown), plus the scale table."""
    lines = _bench_line(["--gpus", "1", "--steps", "1", "--warmup", "1", "--queue", "1024", "--no-cpu-baseline", "--no-unfused",
                         "--no-strong", "--no-class-head", "--steady-new", "1", "--scale-table"], dict(LTR_BENCH_BACKEND="nccl"))
    table, out = lines[0], lines[-1]
    assert out["rccl_ranks"] == 1 and out["backend"] == "nccl" and out["value"] > 0
    r = out["roofline"]
    assert r["compulsory_bytes"] > 0 and r["mfma_only_floor_ms"] > 0 and "traffic_note" in r
    assert table["kind"] == "scale_table" and [p["queue_total"] for p in table["points"]] == [256, 1024, 2048, 4096, 8192, 16384, 32768, 65536]
    assert all(p["calls_per_s"] > 0 and 0 < p["mfma_frac"] < 1 and 0 < p["hbm_frac_compulsory"] < 1 for p in table["points"])


# ---- round 6: a world-8 dry run of the driver / workers mode on one device that takes BOTH payload forms ------
def _worker_world8_payload_forms(rank, world, port, q, form):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if form == "scatter_fails":
        os.environ["LTR_DIST_SCATTER_FAIL"] = "1"
    elif form == "broadcast":
        os.environ["LTR_DIST_PAYLOAD"] = "broadcast"
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from util import bench_lengths, synthetic_batch
        from vllm_ltr_amd.distributed import ShardedScorer
        from vllm_ltr_amd.opt_spec import OPTSpec, seeded_checkpoint
        from vllm_ltr_amd.scorer import HipOPTScorer
        spec = OPTSpec.tiny_pre_ln()
        sc = HipOPTScorer(spec, seeded_checkpoint(spec, 3), "cuda:0", "f16")
        dev = torch.device("cuda:0")
        sh = ShardedScorer(sc, dev, min_requests_to_shard=64, timeout_s=300.0, driver_rank=0, control=True)
        if rank != 0:
            q.put((rank, "worker", sh.serve(), sh.distribution))
            return
        res = {"header_channel": sh.header_channel}
        try:
            for name, n, seed in (("first", 900, 5), ("second", 333, 6)):
                lens = bench_lengths(n, seed=seed, mu=24.0).clip(1, 150)
                ids, cu = synthetic_batch(spec, lens.tolist(), seed)
                ids_d, cu_d = torch.from_numpy(ids).to(dev), torch.from_numpy(cu).to(dev)
                single = sc.score_device(ids_d, cu_d, cu).cpu().numpy()
                got = sh.score_from_driver(ids_d, cu_d, cu).cpu().numpy()
                code = sh.agree_status(0) if sh.last_call_collective else -1
                res[name] = (float(np.abs(got - single).max() / max(1.0, np.abs(single).max())), code, sh.distribution)
        finally:
            sh.stop_workers()
        q.put((rank, "driver", res, sh.distribution_note))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("form", ["scatter", "scatter_fails", "broadcast"])
def test_world8_driver_mode_takes_both_payload_forms(form):
    """Eight processes on the one device (gloo data group + the gloo header side channel): the driver's batch reaches the seven
    workers by ONE scatter of per-rank shards, or - when `dist.scatter` raises (simulated) / LTR_DIST_PAYLOAD=broadcast - by one
    broadcast of the whole payload; either way the gathered scores equal the one-process scores (2e-6) and every worker served
    both calls."""
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_world8_payload_forms, args=(r, world, port, q, form)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=900) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    drv = next(r for r in res if r[1] == "driver")
    out = drv[2]
    want = "scatter" if form == "scatter" else "broadcast"
    assert out["header_channel"] == "gloo side channel"
    for name in ("first", "second"):
        err, code, dist_form = out[name]
        assert err <= 2e-6 and code == 0 and dist_form == want, (name, out[name])
    assert all(r[2] == 2 and r[3] == want for r in res if r[1] == "worker")
    if form == "scatter_fails":
        assert "simulated" in drv[3]


def test_bench_preflight_and_payload_fallback_in_the_line():
    """`bench.py --gpus 2 --driver-broadcast` with a failing scatter: the run goes on over the broadcast form and the line says so;
    every N-rank line carries the preflight record (backend, every rank's device, a 4-byte all-gather round trip)."""
    lines = _bench_line(["--gpus", "2", "--steps", "1", "--warmup", "1", "--queue", "2048", "--no-cpu-baseline", "--no-unfused",
                         "--no-strong", "--no-scale-points", "--no-class-head", "--no-config3", "--steady-new", "0", "--driver-broadcast"],
                        dict(LTR_BENCH_ONE_DEVICE="1", LTR_BENCH_BACKEND="gloo", LTR_DIST_SCATTER_FAIL="1"))
    out = lines[-1]
    assert "ONE broadcast of the whole" in out["input_distribution"] and "simulated" in out["input_distribution"]
    pf = out["preflight"]
    assert pf["ok"] and pf["ranks"] == 2 and len(pf["devices"]) == 2 and pf["all_gather_4B_us_p50"] > 0
    assert out["value"] > 0
