#!/usr/bin/env python3
"""Headline benchmark of the ranking hot path (BASELINE.json):

    requests ranked/sec + p50 rank latency, 8k waiting queue, OPT-125m predictor.

    python bench.py [--gpus N --steps K --warmup W]          (N > 1: re-executes itself under torch.distributed.run)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one COLD ranker call on the synthetic queue (every request unscored):
predictor forward over all prompts (ltr_score) -> [N>1: RCCL all-gather of the score
shards] -> ltr_queue_step = starvation promote/demote + priority sort, budget-walk
prefix, aging (two launches).  Inputs (token ids, cu_seqlens, queue state) are resident in
HBM before the timed region.  N>1 runs the PRODUCT's sharding logic
(vllm_ltr_amd.distributed.ShardedScorer): one global queue known to every rank,
token-balanced contiguous shards from the shared cu_seqlens, each rank scores its shard, one
all-gather of f32 scores, and every rank runs the same deterministic rank step on the whole
gathered queue.

* default (weak scaling): N x 8k requests (BASELINE config 2 at N = 1, config 4 - the 64k queue - at N = 8);
* ``--queue-total Q`` (strong scaling): a FIXED queue of Q requests at every N - north_star's "1k-64k-request queues
  at 1/2/4/8 GPUs"; the default run also carries the Q = 65,536 point of that table in ``strong_scaling``;
* ``--scale-table`` (alias ``--sweep``): one extra JSON line (printed first) with north_star's table at this N - the cold
  call on FIXED queues of 256, 1k ... 64k requests: calls/s, requests/s, fraction of the MFMA and of the HBM roof;
* ``--trace burst|gamma``: BASELINE config 5, ranker side - replays an arrival trace through the scheduler plug-in
  (vllm_ltr_amd.replay) and prints the ranker's latency distribution per scheduler step instead of the headline line.

The timed region runs with the library's event profiler OFF; the per-kernel breakdown and ``roofline`` come from a
second pass of the same K steps with it on (``profiled_ms_per_step`` next to ``ms_per_step``).  Rank 0 prints one JSON
line (contract in the task statement) with ``roofline`` for the dominant kernel and ``cpu_baseline`` (the oracle = CPU
restatement of the reference, timed on this host on a bounded sample).
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from vllm_ltr_amd.opt_spec import OPTSpec, seeded_checkpoint  # noqa: E402

PEAK_F16_MFMA_TFLOPS = 2500.0     # MI355X dense fp16/bf16 MFMA (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0             # HBM3E spec


PROFILES = {"sharegpt": 64.0, "lmsys": 128.0}     # median prompt length of the lognormal profile
# sources whose text decides what the PMC passes under profiles/ measured (profiles/make_traffic.py stamps their hash)
PMC_SOURCES = ("vllm_ltr_amd/csrc/ltr_gemm.hip", "vllm_ltr_amd/csrc/ltr_api.hip")


def kernel_sources_sha16() -> str:
    h = hashlib.sha256()
    for rel in PMC_SOURCES:
        with open(os.path.join(ROOT, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def synthetic_queue(spec: OPTSpec, n: int, seed: int, profile: str = "sharegpt"):
    """BASELINE.md section 4 / SURVEY.md 8d: lengths clip(rint(exp(N(ln 64, 0.8))), 4, 1024)
    ("LMSYS-like", config 3: ln 128); ids [2] + randint(4, vocab)."""
    rs = np.random.RandomState(seed)
    lens = np.clip(np.rint(np.exp(rs.normal(np.log(PROFILES[profile]), 0.8, n))), 4, 1024).astype(np.int64)
    g = torch.Generator().manual_seed(seed)
    T = int(lens.sum())
    ids = torch.randint(4, spec.vocab_size, (T,), generator=g, dtype=torch.int64).numpy()
    cu = np.zeros(n + 1, np.int32)
    np.cumsum(lens, out=cu[1:])
    ids[cu[:-1]] = 2
    return ids, cu, lens


def model_flops(spec: OPTSpec, lens: np.ndarray):
    """SURVEY.md 8d algorithmic FLOPs of the decoder stack: Nl*[2L(4H^2+2HF) + 2L^2 H]."""
    H, F, Nl = spec.hidden_size, spec.ffn_dim, spec.num_hidden_layers
    L = lens.astype(np.float64)
    lin = Nl * 2.0 * (4 * H * H + 2 * H * F) * L.sum()
    if spec.has_proj:
        lin += 4.0 * spec.word_embed_proj_dim * H * L.sum()
    att = Nl * 2.0 * H * (L * L).sum()
    return lin, att


def compulsory_bytes(spec: OPTSpec, tokens: int, passes: int = 1) -> float:
    """HBM bytes the GEMM launches of one scoring call MUST move in the split-fp16 layout of DESIGN.md 3 (every activation
    between launches is 4 B per element: fp16 hi + lo planes, or f32).  Per token and layer: QKV reads its operand (4H) and
    writes q|k|v (12H); out_proj reads the attention output (4H) and the residual row (4H), writes the row (4H) and the
    next GEMM's operand (4H); fc1 reads that (4H), writes ReLU(fc1) (4F); fc2 reads it (4F) and the residual (4H), writes
    the row (4H) and the next operand (4H): 48 H + 8 F bytes (61,440 at the OPT-125m shape).  The pruned last layer
    moves the K|V projection only (4H in, 8H out).  Weights: once per pass."""
    H, F, Nl = spec.hidden_size, spec.ffn_dim, spec.num_hidden_layers
    per_tok = (48.0 * H + 8.0 * F) * (Nl - 1) + 12.0 * H
    if spec.has_proj:
        per_tok += 4.0 * spec.word_embed_proj_dim + 8.0 * H
    w = 2.0 * Nl * (4.0 * H * H + 2.0 * H * F)
    return tokens * per_tok + passes * w


def cpu_baseline(spec, ckpt, ids, cu, starv, period, budget_s: float = 12.0):
    """The oracle (CPU restatement of the reference path, oracle/) timed on this host's
    cores on a bounded prefix of the same workload: fp32 torch predictor packed <= 2048
    tokens per forward like the AUX engine (config.py:578-586) + the literal Python
    promote/demote + sorted() + aging."""
    from oracle import rank_step as rs
    from oracle.opt_scorer import OracleOPTScorer
    # threads actually used: the affinity mask, capped - beyond ~32 threads the small
    # per-forward GEMMs ([<=2048, 768] x [768, 3072]) stop scaling and oversubscribed
    # hosts collapse (measured: 256 threads on the GPU box = 13 s per request)
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, 32))
    torch.set_num_threads(cores)
    orc = OracleOPTScorer(spec, ckpt)
    orc.score_packed(ids[:cu[2]], cu[:3])                      # warm-up (thread pool, allocator)
    n0 = 8
    t = time.perf_counter()
    orc.score_packed(ids[:cu[n0]], cu[:n0 + 1])
    dt = time.perf_counter() - t
    n = int(min(len(cu) - 1, max(n0, n0 * budget_s / max(dt, 1e-3))))
    t = time.perf_counter()
    scores = orc.score_packed(ids[:cu[n]], cu[:n + 1])
    t_score = time.perf_counter() - t
    reqs = [rs.Req(str(i), float(s)) for i, s in enumerate(scores)]
    t = time.perf_counter()
    order = rs.opt_order(reqs, starv, period)
    rs.age_update(reqs, order[:256])
    t_rank = time.perf_counter() - t
    # what more threads give: the same 8-request sample at 32 / 64 / 128 threads (~1 s each)
    sweep = {}
    for th in (32, 64, 128):
        if th > avail:
            break
        torch.set_num_threads(th)
        orc.score_packed(ids[:cu[2]], cu[:3])
        t = time.perf_counter()
        orc.score_packed(ids[:cu[n0]], cu[:n0 + 1])
        sweep[str(th)] = n0 / (time.perf_counter() - t)
    torch.set_num_threads(cores)
    return dict(value=n / (t_score + t_rank), unit="requests/s", cores=cores, threads=cores, host_cores=os.cpu_count(),
                host_cores_usable=avail, kind="port",
                threads_sweep=dict(requests_per_s=sweep, sample=f"first {n0} requests ({int(cu[n0])} tokens), one packed forward"),
                sample=f"first {n} requests of the same queue ({int(cu[n])} tokens): oracle fp32 torch "
                       f"forward packed<=2048 tok ({t_score:.2f}s) + literal Python rank/age ({t_rank*1e3:.2f}ms)")


class ColdCall:
    """One global queue of ``n_total`` requests, resident on every rank, and its cold ranker call."""

    def __init__(self, spec, scorer, dev, dist, world, rank, n_total, profile, min_shard_tokens, starv, period, seed=0,
                 timeout_s=None, driver_mode=False, sharded=None):
        from vllm_ltr_amd.distributed import ShardedScorer, shard_bounds
        from vllm_ltr_amd.rank import DeviceQueue
        self.scorer, self.dev, self.dist, self.world, self.rank, self.n_total = scorer, dev, dist, world, rank, n_total
        self.ids, self.cu, self.lens = synthetic_queue(spec, n_total, seed=seed, profile=profile)
        # --driver-broadcast: only rank 0 (the rank that would own the scheduler) holds the queue on its device; the other
        # ranks receive their shard INSIDE every step (header broadcast + one scatter, distributed.py) and never see the rest
        self.driver_mode = bool(driver_mode and world > 1)
        self.passive = self.driver_mode and rank != 0
        self.ids_d = None if self.passive else torch.from_numpy(self.ids).to(dev)
        self.cu_d = None if self.passive else torch.from_numpy(self.cu).to(dev)
        # (one wrapper for the whole run when the caller has one: its header side channel is a process group of its own)
        self.sharded = sharded if sharded is not None else (
            ShardedScorer(scorer, dev, min_tokens_to_shard=min_shard_tokens, timeout_s=timeout_s, driver_rank=0,
                          control="auto" if self.driver_mode else None) if world > 1 else None)
        self.is_sharded = bool(self.sharded is not None and self.sharded.shards(n_total, int(self.cu[-1])))
        self.bounds = shard_bounds(self.cu, world) if self.is_sharded else [(0, n_total)] + [(n_total, n_total)] * (world - 1)
        self.r0, self.r1 = self.bounds[rank]
        self.tokens_shard = [int(self.cu[b] - self.cu[a]) for a, b in self.bounds]        # every rank's share (same on all ranks)
        self.queue = DeviceQueue(dev, starv=starv, period=period, capacity=n_total)
        self.queue.append(torch.zeros(n_total))
        self.need_tokens = torch.from_numpy(self.lens.astype(np.int32)).to(dev)
        self.need_seqs = torch.ones(n_total, dtype=torch.int32, device=dev)
        self.perm = torch.empty(n_total, dtype=torch.int32, device=dev)

    def rank_part(self):
        # promote/demote + sort, budget-walk prefix, aging: ltr_queue_step, two launches
        self.queue.step(self.need_tokens, self.need_seqs, 2048, 256, perm_out=self.perm)

    def step(self):
        out = self.queue._score[:self.n_total]
        if self.driver_mode:
            if self.passive:
                if self.is_sharded:                     # (an unsharded call never reaches a worker)
                    self.sharded.serve_once()
                return
            self.sharded.score_from_driver(self.ids_d, self.cu_d, self.cu, out=out)
            if self.sharded.last_call_collective:
                self.sharded.agree_status(0)            # the status agreement every collective call ends with
            self.rank_part()
            return
        if self.sharded is not None:
            # the one exchange step of the path: RCCL all-gather of f32 score shards over xGMI, compacted straight
            # into the queue's score slots
            self.sharded.score_device(self.ids_d, self.cu_d, self.cu, out=out)
        else:
            self.scorer.score_device(self.ids_d, self.cu_d, self.cu, out=out)
        self.rank_part()

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()
        torch.cuda.synchronize()

    def timed(self, steps, warmup):
        """W untimed + exactly K timed steps between barrier + synchronize; returns (max-over-ranks seconds, sorted
        per-step ms from events on this rank)."""
        for _ in range(warmup):
            self.step()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        self.barrier()
        t0 = time.perf_counter()
        for k in range(steps):
            ev[k][0].record()
            self.step()
            ev[k][1].record()
        self.barrier()
        elapsed = time.perf_counter() - t0
        on_dev = self.dist is None or self.dist.get_backend() == "nccl"
        t = torch.tensor([elapsed], dtype=torch.float64, device=self.dev if on_dev else "cpu")
        if self.dist is not None:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item()), sorted(a.elapsed_time(b) for a, b in ev)

    def release(self):
        for a in ("ids_d", "cu_d", "queue", "need_tokens", "need_seqs", "perm"):
            if hasattr(self, a):
                delattr(self, a)


def run_trace(args, spec, ckpt, dev):
    """BASELINE config 5, ranker side: k arrivals -> obtain_aux_scores(k) + order + budget walk + age per scheduler step."""
    from vllm_ltr_amd.plugin import MI355XRanker
    from vllm_ltr_amd.replay import replay, summarize, synthetic_trace
    from vllm_ltr_amd.scorer import HipOPTScorer
    scorer = HipOPTScorer(spec, ckpt, str(dev), args.weight_dtype)
    ranker = MI355XRanker(scorer, f"opt-xxx-starv{args.starv}-period{args.period}", max_length=1024, prescore=args.prescore,
                          prescore_graphs=not args.no_prescore_graphs)
    reqs = synthetic_trace(spec.vocab_size, args.trace_requests, args.trace, args.trace_rate, args.trace_cv, seed=0,
                           prompt_median=PROFILES[args.profile])
    # warm the kernels / allocator on a throw-away ranker-sized call
    warm = synthetic_trace(spec.vocab_size, 64, "burst", seed=7)
    ranker.obtain_aux_scores(warm)
    ranker.warm_prescore_graphs()
    res = replay(ranker, reqs, backbone_ms=args.trace_backbone_ms)
    s = summarize(res)
    out = {"metric": "ranker latency per scheduler step under an arrival trace (obtain_aux_scores + order + budget walk + aging "
                     "through MI355XRanker.install)",
           "value": s["ranker_ms_all"]["p50"], "unit": "ms", "higher_is_better": False, "n_gpus": 1,
           "data": "synthetic (seeded random-init OPT checkpoint, lognormal prompt / output lengths)",
           "config": {"workload": f"OPT-{args.model} predictor, {args.trace} trace of {args.trace_requests} requests"
                                  + (f" at {args.trace_rate} req/s, cv {args.trace_cv}" if args.trace == "gamma" else " at t = 0")
                                  + f", stand-in backbone step {args.trace_backbone_ms} ms, budget 2048 tokens / 256 seqs",
                      "starv": args.starv, "period": args.period,
                      "prescore": bool(args.prescore)},
           "trace": s, "ranker_metrics": ranker.metrics()}
    print(json.dumps(out))


def run_train(args, spec, ckpt, dev):
    """the fine-tuning step (train/trainer.py:137-165: forward, listMLE, backward, Adam) on a slate of
    ``--train-slate`` prompts of the bench length profile (train.sh: --batch-size 32)."""
    from vllm_ltr_amd.trainer import HipPredictorTrainer
    n = args.train_slate
    ids, cu, lens = synthetic_queue(spec, n, seed=0, profile=args.profile)
    labels = np.random.RandomState(1).permutation(n).astype(np.float32)      # (neuralNDCG: 2^label gains, labels must stay < 128)
    if args.train_loss == "neuralNDCG":
        labels = labels % 83                                                 # --label-group-size 100 (train.sh): labels 0..82
    sh = np.random.RandomState(2).permutation(n)

    def timed(precision):
        tr = HipPredictorTrainer(spec, ckpt, str(dev), lr=2e-5, weight_decay=0.01, loss=args.train_loss, precision=precision)
        for _ in range(args.warmup):
            tr.step(ids, cu, labels, shuffle=sh)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            loss = tr.step(ids, cu, labels, shuffle=sh)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.steps
        tr.close()
        return dt, loss
    # round 2's arithmetic for comparison: every GEMM on the exact-f32 MFMA, f32 VALU attention
    dt32 = timed("f32")[0] if args.train_precision == "both" else None
    dt, loss = timed("split")
    lin, att = model_flops(spec, lens)
    out = {"metric": f"fine-tuning step of the predictor (forward + {args.train_loss} + backward + Adam), tokens/s",
           "value": float(cu[-1]) / dt, "unit": "tokens/s", "higher_is_better": True, "n_gpus": 1, "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": dt * 1e3,
           "dtype": "f32 master weights / activations / gradients; GEMMs split-fp16 x split-fp16 MFMA (4 passes), f32 accumulate",
           "exact_f32_ms_per_step": dt32 * 1e3 if dt32 else None, "speedup_vs_exact_f32": dt32 / dt if dt32 else None,
           "data": "synthetic (seeded random-init OPT checkpoint, lognormal prompt lengths)",
           "config": {"workload": f"OPT-{args.model} predictor, slate of {n} prompts ({int(cu[-1])} tokens), {args.train_loss}, Adam"},
           "algorithmic_tflops": 3.0 * (lin + att) / dt / 1e12, "loss": loss}
    print(json.dumps(out))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--queue", type=int, default=8192, help="requests per GPU (weak scaling)")
    ap.add_argument("--queue-total", type=int, default=0,
                    help="fixed total queue at every N (strong scaling; overrides --queue)")
    ap.add_argument("--model", default="125m", choices=["125m", "350m"])
    ap.add_argument("--profile", default="sharegpt", choices=sorted(PROFILES),
                    help="prompt-length profile: sharegpt = ln 64 (BASELINE config 2), lmsys = ln 128 (config 3 with --model 350m)")
    ap.add_argument("--min-shard-tokens", type=int, default=196608,
                    help="ShardedScorer.min_tokens_to_shard: a call is sharded over the ranks only when it holds more tokens "
                         "than one pass of one GPU (north_star: 'only when the queue exceeds a single GPU's batch')")
    ap.add_argument("--collective-timeout", type=float, default=600.0,
                    help="seconds a rank waits for its peers in a collective before it raises (a dead peer must not hang the node)")
    ap.add_argument("--weight-dtype", default="f16", choices=["f16", "f32", "f16-1pass"],
                    help="f16 (default): fp16 weights x (hi+lo) fp16 activations - the 1e-4 contract, the headline; f32: exact f32 "
                         "MFMA; f16-1pass: ONE fp16 MFMA pass per product = the reference's own fp16 GPU arithmetic (~2e-3 from "
                         "the fp32 scores): a second, separately labelled number, never the headline")
    ap.add_argument("--driver-broadcast", action="store_true",
                    help="N > 1: only rank 0 holds the queue (the rank that owns the scheduler in a vllm-ltr engine); every "
                         "timed step includes the distribution of the inputs to the other ranks (header + per-rank scatter)")
    ap.add_argument("--no-scale-points", action="store_true", help="skip the 1k / 4k / 16k points of north_star's table (scale_table)")
    ap.add_argument("--chunk-tokens", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-unfused", action="store_true",
                    help="skip the extra forward with separate LayerNorm launches (roofline.unfused): keeps a rocprofv3 "
                         "trace / PMC pass of this command to the launches of the timed configuration")
    ap.add_argument("--no-profile-pass", action="store_true",
                    help="skip the second, event-profiled pass (roofline / kernels become null): for rocprofv3 runs")
    ap.add_argument("--starv", type=int, default=200)
    ap.add_argument("--period", type=int, default=10)
    ap.add_argument("--steady-new", default="1,16,64,256",
                    help="steady calls to time outside the timed region: k new requests scored + the whole queue re-ranked "
                         "(SURVEY 8d 'steady'); comma list, '0' = none")
    ap.add_argument("--no-strong", action="store_true", help="skip the 65,536-request strong-scaling point")
    ap.add_argument("--no-host-inclusive", action="store_true", help="skip the host_inclusive block (profiling passes: keeps the launch counts of the run to the timed steps)")
    ap.add_argument("--no-config3", action="store_true", help="skip the BASELINE config 3 block (OPT-350m, 8k lmsys queue) of the default run")
    ap.add_argument("--no-class-head", action="store_true",
                    help="skip the class-mode head measurement (8,192 requests x 8,192 labels, kernels.class_head)")
    ap.add_argument("--sweep", "--scale-table", dest="sweep", action="store_true",
                    help="extra JSON line (north_star's table at this N): cold call on fixed queues of 256, 1k ... 64k requests")
    ap.add_argument("--trace", default=None, choices=["burst", "gamma"], help="config 5 ranker-side trace replay")
    ap.add_argument("--train", action="store_true", help="time the fine-tuning step instead (SURVEY 8f-4)")
    ap.add_argument("--train-slate", type=int, default=32)
    ap.add_argument("--train-loss", default="listMLE", choices=["listMLE", "neuralNDCG"], help="trainer.py --loss (ranking losses)")
    ap.add_argument("--train-precision", default="both", choices=["both", "split"],
                    help="both: also time the exact-f32 path for comparison; split: the product path only (profiling)")
    ap.add_argument("--prescore", action="store_true", help="trace replay: score requests when they arrive (MI355XRanker(prescore=True))")
    ap.add_argument("--no-prescore-graphs", action="store_true", help="with --prescore: launch every forward eagerly (no captured graphs)")
    ap.add_argument("--trace-requests", type=int, default=2000)
    ap.add_argument("--trace-rate", type=float, default=16.0)
    ap.add_argument("--trace-cv", type=float, default=1.0)
    ap.add_argument("--trace-backbone-ms", type=float, default=25.0)
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` run plainly: become the N-rank job (one process per GPU) through torch's own launcher,
        # same arguments; 127.0.0.1 rendezvous on a free port.  The torch.distributed.run form keeps working (it sets
        # WORLD_SIZE, so this branch is skipped there).
        import socket
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.execv(sys.executable, [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
                                  "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)]
                 + sys.argv[1:])
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started {world} ranks (WORLD_SIZE): pass --gpus {world}")
    # LTR_BENCH_ONE_DEVICE=1 + LTR_BENCH_BACKEND=gloo: dry run of the N-rank path on a 1-GPU box (test hook)
    if os.environ.get("LTR_BENCH_ONE_DEVICE") == "1":
        local_rank = 0
    if local_rank >= torch.cuda.device_count():
        raise SystemExit(f"rank {rank}: local rank {local_rank} but only {torch.cuda.device_count()} GPU(s) visible "
                         "(one process per GPU; LTR_BENCH_ONE_DEVICE=1 LTR_BENCH_BACKEND=gloo for a one-device dry run)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist, backend = None, None
    # LTR_BENCH_BACKEND names the collective backend ("nccl" = RCCL on ROCm, the default); when it is set, the process
    # group is created at world size 1 too, so that a plain `--gpus 1` run exercises RCCL's init / teardown
    if world > 1 or os.environ.get("LTR_BENCH_BACKEND"):
        import datetime
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = os.environ.get("LTR_BENCH_BACKEND", "nccl")          # "nccl" is RCCL on ROCm
        if "MASTER_ADDR" not in os.environ:                              # world 1 without a launcher
            os.environ["MASTER_ADDR"] = "127.0.0.1"
            os.environ.setdefault("MASTER_PORT", "29533")
            os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        tmo = datetime.timedelta(seconds=max(args.collective_timeout, 1.0))
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, timeout=tmo)
        else:
            dist.init_process_group(backend, timeout=tmo)
        backend = dist.get_backend()

    from vllm_ltr_amd.scorer import HipOPTScorer

    # ---- preflight of the N-rank run (the first real RCCL run should say what it found before it times
    # anything): library version, every rank's device, one 4-byte all-gather round trip.  Rank 0 prints it on stderr at once
    # (so that it survives a later hang) and the final line carries it.
    preflight = None
    if dist is not None:
        pf = {"backend": backend, "ranks": world}
        try:
            pf["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version()) if backend == "nccl" else None
        except Exception as e:      # noqa: BLE001
            pf["rccl_version"] = f"unavailable ({type(e).__name__})"
        try:
            props = torch.cuda.get_device_properties(dev)
            mine = dict(rank=rank, local_rank=local_rank, device=props.name, cus=props.multi_processor_count,
                        pci=getattr(props, "pci_bus_id", None), hbm_gb=round(props.total_memory / 2**30))
            allr = [None] * world
            dist.all_gather_object(allr, mine)
            pf["devices"] = allr
            on = backend == "nccl"
            a = torch.zeros(1, dtype=torch.float32, device=dev if on else "cpu")
            g = torch.zeros(world, dtype=torch.float32, device=dev if on else "cpu")
            ts = []
            for _ in range(25):
                if on:
                    torch.cuda.synchronize()
                t0 = time.perf_counter()
                dist.all_gather_into_tensor(g, a)
                if on:
                    torch.cuda.synchronize()
                ts.append((time.perf_counter() - t0) * 1e6)
            pf["all_gather_4B_us_p50"] = sorted(ts[5:])[len(ts[5:]) // 2]
            pf["ok"] = True
        except Exception as e:      # noqa: BLE001 - say it and go on: the timed run will fail where the collective is needed
            pf["ok"] = False
            pf["error"] = f"{type(e).__name__}: {e}"[:300]
        preflight = pf
        if rank == 0:
            print(json.dumps({"kind": "preflight", **pf}), file=sys.stderr, flush=True)

    spec = OPTSpec.opt_125m() if args.model == "125m" else OPTSpec.opt_350m()
    ckpt = seeded_checkpoint(spec, 0)
    if args.trace:
        return run_trace(args, spec, ckpt, dev)
    if args.train:
        return run_train(args, spec, ckpt, dev)
    scorer = HipOPTScorer(spec, ckpt, str(dev), args.weight_dtype, chunk_tokens=args.chunk_tokens)

    strong = args.queue_total > 0
    n_total = args.queue_total if strong else args.queue * world
    n_local = n_total // world if strong else args.queue
    shared = None
    if world > 1:
        from vllm_ltr_amd.distributed import ShardedScorer
        shared = ShardedScorer(scorer, dev, min_tokens_to_shard=args.min_shard_tokens, timeout_s=args.collective_timeout, driver_rank=0,
                               control="auto" if args.driver_broadcast else None)
    mk = lambda n: ColdCall(spec, scorer, dev, dist, world, rank, n, args.profile, args.min_shard_tokens, args.starv, args.period,
                            timeout_s=args.collective_timeout, driver_mode=args.driver_broadcast, sharded=shared)
    f16 = args.weight_dtype in ("f16", "f16-1pass")        # both run the fp16-weight kernels
    one_pass = args.weight_dtype == "f16-1pass"

    if args.sweep:
        # north_star: "ranker calls/sec on synthetic 1k-64k-request queues reported at 1/2/4/8 GPUs as absolute numbers and as
        # fraction of the ... roofline": FIXED queues at this N (strong scaling), one cold call = one ranker call
        pts = []
        for n in (256, 1024, 2048, 4096, 8192, 16384, 32768, 65536):
            c = mk(n)
            k = 3 if n <= 16384 else 2
            el, _ = c.timed(k, 1)
            lin, att = model_flops(spec, c.lens)
            T = int(c.cu[-1])
            byt = compulsory_bytes(spec, T, passes=max(1, -(-max(c.tokens_shard) // 196608)))
            pts.append(dict(queue_total=n, tokens_total=T, ms_per_call=el / k * 1e3, calls_per_s=k / el, requests_per_s=n * k / el,
                            sharded=c.is_sharded, tokens_shard=c.tokens_shard,
                            mfma_frac=(lin + att) / (el / k) / 1e12 / (PEAK_F16_MFMA_TFLOPS * world),
                            hbm_frac_compulsory=byt / (el / k) / 1e9 / (PEAK_HBM_GBS * world)))
            c.release(); del c
        if rank == 0:
            print(json.dumps({"kind": "scale_table", "n_gpus": world, "rccl_ranks": world, "backend": backend, "model": args.model,
                              "profile": args.profile, "scaling": "strong (fixed queue at every N)",
                              "roofs": {"mfma_tflops_per_gpu": PEAK_F16_MFMA_TFLOPS, "hbm_gbs_per_gpu": PEAK_HBM_GBS,
                                        "note": "mfma_frac = algorithmic FLOPs (SURVEY 8d; the 2-pass split not counted) / time / "
                                                "(N x 2.5 PF); hbm_frac_compulsory = bytes the split-fp16 layout must move (DESIGN 3) "
                                                "/ time / (N x 8 TB/s)"},
                              "points": pts}))

    call = mk(n_total)
    ids, cu, lens = call.ids, call.cu, call.lens
    my_r0, my_r1 = call.r0, call.r1
    call_tokens_shard, call_sharded = call.tokens_shard, call.is_sharded
    for _ in range(args.warmup):
        call.step()
    call.barrier()
    # The GEMM launches of the default build also carry the LayerNorm work (LayerNorm fold, ltr_gemm.hip), so their
    # FLOP rate is not comparable with a plain GEMM's.  For the record, time the same call once more on a second handle
    # with the fold off (LTR_F_NO_LN_FOLD at ltr_create): `roofline.unfused` below.  Outside the timed region.
    unfused = None
    if rank == 0 and f16 and not args.no_unfused and os.environ.get("LTR_NO_LN_FOLD") != "1":
        sc2 = HipOPTScorer(spec, ckpt, str(dev), args.weight_dtype, chunk_tokens=args.chunk_tokens, ln_fold=False)
        tmp = torch.empty(my_r1 - my_r0, dtype=torch.float32, device=dev)
        cu_loc = np.ascontiguousarray(cu[my_r0:my_r1 + 1] - cu[my_r0]).astype(np.int32)
        ids_loc, cu_loc_d = call.ids_d[int(cu[my_r0]):int(cu[my_r1])], torch.from_numpy(cu_loc).to(dev)
        sc2.score_device(ids_loc, cu_loc_d, cu_loc, out=tmp)
        sc2.profile(True); sc2.profile_read(reset=True)
        for _ in range(2):
            sc2.score_device(ids_loc, cu_loc_d, cu_loc, out=tmp)
        pu = sc2.profile_read(reset=True)
        unfused = dict(gemm_ms=pu["gemm"]["ms"] / 2, ln_ms=pu["ln"]["ms"] / 2,
                       gemm_tflops=pu["gemm"]["work"] / (pu["gemm"]["ms"] * 1e-3) / 1e12,
                       frac=pu["gemm"]["work"] / (pu["gemm"]["ms"] * 1e-3) / 1e12 / PEAK_F16_MFMA_TFLOPS)
        sc2.close(); del sc2, tmp
    # ---- the timed region: exactly K steps, event profiler OFF
    scorer.profile(False)
    elapsed, step_ms = call.timed(args.steps, 0)
    # ---- second pass of the same K steps with the library's event profiler on: per-class kernel time (HIP events on
    # the launch stream around every launch) for `roofline` and `kernels`
    prof, prof_elapsed = None, None
    if not args.no_profile_pass:
        scorer.profile(True)
        scorer.profile_read(reset=True)
        prof_elapsed, _ = call.timed(args.steps, 0)
        prof = scorer.profile_read(reset=True)
        scorer.profile(False)
    # rank-only latency (steady call with nothing new to score), measured outside the timed region
    rk = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
    for a, b in rk:
        a.record()
        call.rank_part()
        b.record()
    torch.cuda.synchronize()
    rank_ms = sorted(a.elapsed_time(b) for a, b in rk)
    # steady calls with k new requests: score the first k of the local queue on THIS GPU, then
    # promote/demote + sort + budget prefix + aging over the whole queue - what a scheduler step with k arrivals pays
    steady = {}
    for k_new in [] if call.passive else [int(x) for x in args.steady_new.split(",") if x.strip() and int(x) > 0]:
        k_new = min(k_new, n_total)
        cu_k = np.ascontiguousarray(cu[:k_new + 1])
        ids_k, cu_k_d = call.ids_d[:int(cu_k[-1])], call.cu_d[:k_new + 1]
        sk = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(12)]
        for a, b in sk:
            a.record()
            scorer.score_device(ids_k, cu_k_d, cu_k, out=call.queue._score[:k_new])
            call.rank_part()
            b.record()
        torch.cuda.synchronize()
        ms = sorted(a.elapsed_time(b) for a, b in sk[2:])
        steady[str(k_new)] = ms[len(ms) // 2]
    # the same cold call with the boundary's HOST side inside the clock (the plug-in hands over host token ids and takes host
    # scores / order back: pinned H2D of ids + cu_seqlens, the call, pinned D2H of the scores and the order).  Reported beside
    # `value`, never as it (N = 1, outside the timed region).
    host_incl = None
    if world == 1 and not args.sweep and not args.no_host_inclusive:
        ids_h, cu_h = torch.from_numpy(ids).pin_memory(), torch.from_numpy(cu).pin_memory()
        sc_h = torch.empty(n_total, dtype=torch.float32).pin_memory()
        pm_h = torch.empty(n_total, dtype=torch.int32).pin_memory()
        hs = []
        for it in range(4):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ids_x, cu_x = ids_h.to(dev, non_blocking=True), cu_h.to(dev, non_blocking=True)
            scorer.score_device(ids_x, cu_x, cu, out=call.queue._score[:n_total])
            call.rank_part()
            sc_h.copy_(call.queue._score[:n_total], non_blocking=True)
            pm_h.copy_(call.perm, non_blocking=True)
            torch.cuda.synchronize()
            hs.append(time.perf_counter() - t0)
        hs = sorted(hs[1:])
        host_incl = {"value": n_total / hs[len(hs) // 2], "unit": "requests/s", "ms_per_step": hs[len(hs) // 2] * 1e3,
                     "h2d_bytes": int(ids_h.numel() * ids_h.element_size() + cu_h.numel() * cu_h.element_size()),
                     "d2h_bytes": int(n_total * 8),
                     "what": "pinned H2D of token ids + cu_seqlens, the cold call, pinned D2H of scores + order; wall clock, median of 3"}
        del ids_h, cu_h, sc_h, pm_h
    call.barrier()

    # ---- north_star's strong-scaling table, the 64k point: the SAME fixed 65,536-request queue at every N
    strong_pt = None
    if not args.no_strong and not strong and not args.sweep:
        call.release()
        c64 = mk(65536)
        el, _ = c64.timed(2, 1)
        strong_pt = dict(queue_total=65536, tokens_total=int(c64.cu[-1]), n_gpus=world, ms_per_call=el / 2 * 1e3,
                         requests_per_s=65536 * 2 / el, scaling="strong", tokens_shard=c64.tokens_shard)
        c64.release(); del c64

    # ---- three more points of north_star's table (fixed queues of 1k / 4k / 16k requests at this N; the 8k point is the
    # headline, the 64k point `strong_scaling`): calls/s at five queue sizes in the driver's own record
    scale_pts = None
    if not args.no_scale_points and not strong and not args.sweep:
        if strong_pt is None:
            call.release()
        scale_pts = []
        for n_q in (1024, 4096, 16384):
            c = mk(n_q)
            el, _ = c.timed(2, 1)
            lin_q, att_q = model_flops(spec, c.lens)
            byt_q = compulsory_bytes(spec, int(c.cu[-1]), passes=max(1, -(-max(c.tokens_shard) // 196608)))
            scale_pts.append(dict(queue_total=n_q, tokens_total=int(c.cu[-1]), ms_per_call=el / 2 * 1e3, calls_per_s=2 / el,
                                  requests_per_s=n_q * 2 / el, sharded=c.is_sharded,
                                  mfma_frac=(lin_q + att_q) / (el / 2) / 1e12 / (PEAK_F16_MFMA_TFLOPS * world),
                                  hbm_frac_compulsory=byt_q / (el / 2) / 1e9 / (PEAK_HBM_GBS * world)))
            c.release(); del c

    # ---- class-mode head at the reference's largest bucket count (train/train.sh: 8,192 labels; opt.py:389-397): the
    # head of an 8,192-request call = final LayerNorm of the last-token rows + [8192, De] x [8192, De]^T logits on the
    # split-fp16 MFMA kernel + row argmax.  One-token prompts keep the forward in front of it negligible.
    class_head = None
    if rank == 0 and world == 1 and not args.no_class_head and args.weight_dtype == "f16" and not args.sweep:
        spec_c = OPTSpec.opt_125m(8192) if args.model == "125m" else OPTSpec.opt_350m(8192)
        ck = dict(ckpt)
        ck["score.weight"] = (0.05 * np.random.RandomState(1).standard_normal((8192, spec.word_embed_proj_dim))).astype(np.float16)
        sc_c = HipOPTScorer(spec_c, ck, str(dev), "f16")
        n_c = 8192
        ids_c = torch.full((n_c,), 2, dtype=torch.int64, device=dev)
        cu_c = np.arange(n_c + 1, dtype=np.int32)
        cu_c_d = torch.from_numpy(cu_c).to(dev)
        out_c = torch.empty(n_c, device=dev)
        for _ in range(2):
            sc_c.score_device(ids_c, cu_c_d, cu_c, out=out_c)
        sc_c.profile(True); sc_c.profile_read(reset=True)
        for _ in range(5):
            sc_c.score_device(ids_c, cu_c_d, cu_c, out=out_c)
        pc = sc_c.profile_read(reset=True)["pool"]
        ms_c = pc["ms"] / 5
        fl = 2.0 * n_c * 8192 * spec.word_embed_proj_dim
        class_head = dict(requests=n_c, num_labels=8192, ms=ms_c, launches=pc["launches"] // 5,
                          logits_tflops=fl / (ms_c * 1e-3) / 1e12,
                          note="LayerNorm / project_out of the last-token rows + logits GEMM + argmax, per call")
        sc_c.close(); del sc_c, ids_c, out_c

    # ---- BASELINE config 3 in the driver's own record: OPT-350m predictor, 8,192-request LMSYS-like queue, one
    # GPU - one warm-up, two timed cold calls (profiler off), one more with the event profiler for the kernel classes.  Only in
    # the default run (the headline workload on one GPU); `--model 350m --profile lmsys` is the full-length run of the same thing.
    config3 = None
    if (rank == 0 and world == 1 and not args.no_config3 and args.weight_dtype == "f16" and not args.sweep and not strong
            and (args.model, args.profile, args.queue) == ("125m", "sharegpt", 8192)):
        call.release()
        spec3 = OPTSpec.opt_350m()
        ck3 = seeded_checkpoint(spec3, 0)
        sc3 = HipOPTScorer(spec3, ck3, str(dev), "f16")
        c3 = ColdCall(spec3, sc3, dev, None, 1, 0, 8192, "lmsys", args.min_shard_tokens, args.starv, args.period,
                      timeout_s=args.collective_timeout)
        sc3.profile(False)
        el3, _ = c3.timed(2, 1)
        sc3.profile(True); sc3.profile_read(reset=True)
        c3.timed(1, 0)
        p3 = sc3.profile_read(reset=True)
        sc3.profile(False)
        T3 = int(c3.cu[-1])
        lin3, att3 = model_flops(spec3, c3.lens)
        g3 = p3["gemm"]
        tf3 = g3["work"] / (g3["ms"] * 1e-3) / 1e12 if g3["ms"] > 0 else 0.0
        comp3 = compulsory_bytes(spec3, T3, max(1, -(-T3 // 196608)))
        traffic3, tsrc3 = None, None
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "gemm_traffic.350m_lmsys_8192.json")))
            tsrc3 = tj.get("source")
            if tsrc3 and tsrc3.get("kernel_sha16") == kernel_sources_sha16() and tsrc3.get("workload") == "350m/lmsys/8192":
                traffic3 = tj.get("hbm_bytes_per_launch")
        except Exception:
            pass
        config3 = {
            "workload": "OPT-350m predictor, 8192-request synthetic queue (lmsys length profile), cold ranker call, 1 GPU (BASELINE configs[2])",
            "tokens_total": T3, "steps": 2, "warmup": 1, "value": 8192 * 2 / el3, "unit": "requests/s", "ms_per_step": el3 / 2 * 1e3,
            "model_tflop_per_step": (lin3 + att3) / 1e12,
            "roofline": {"bound": "mfma", "kernel": "gemm_f16s_kernel", "achieved": tf3, "peak": PEAK_F16_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": tf3 / PEAK_F16_MFMA_TFLOPS, "hw_frac": 2 * tf3 / PEAK_F16_MFMA_TFLOPS, "launches_per_step": g3["launches"],
                         "avg_launch_ms": g3["ms"] / max(g3["launches"], 1), "traffic": traffic3, "compulsory_bytes": comp3,
                         "traffic_ratio": traffic3 * g3["launches"] / comp3 if traffic3 else None,
                         "traffic_source": tsrc3 if traffic3 is not None else {"stale": True, "file": tsrc3,
                                                                               "current_kernel_sha16": kernel_sources_sha16()},
                         "measured": "one more cold call with HIP events around every launch (profiler off in the two timed calls)"},
            "kernels": {k: dict(ms_per_step=v["ms"], launches_per_step=v["launches"]) for k, v in p3.items() if v["launches"]},
        }
        c3.release(); sc3.close(); del c3, sc3, ck3

    if rank == 0:
        lin, att = model_flops(spec, lens[my_r0:my_r1])        # this rank's shard: what its profiler timed
        kernels, roof = {}, None
        if prof is not None:
            gemm = prof["gemm"]
            gemm_tflops = gemm["work"] / (gemm["ms"] * 1e-3) / 1e12 if gemm["ms"] > 0 else 0.0
            for k, v in prof.items():
                if v["launches"] == 0:
                    continue
                rate = v["work"] / (v["ms"] * 1e-3)
                if k in ("gemm", "attn", "gemm_small"):
                    kernels[k] = dict(ms_per_step=v["ms"] / args.steps, launches_per_step=v["launches"] // args.steps,
                                      tflops=rate / 1e12)
                else:
                    kernels[k] = dict(ms_per_step=v["ms"] / args.steps, launches_per_step=v["launches"] // args.steps,
                                      gbs=rate / 1e9, frac_hbm=rate / 1e9 / PEAK_HBM_GBS)
            if "embed" in kernels:
                # SURVEY.md 8d counts 4616 B per token for the gather (a 2-byte activation row out); this kernel writes an
                # f32 residual row (6152 B, the `gbs` above).  The fraction on SURVEY's own bytes:
                w = 2.0 if f16 else 4.0
                survey_b = 8.0 + (spec.word_embed_proj_dim + spec.hidden_size) * w + spec.hidden_size * w
                ours_b = 8.0 + (spec.word_embed_proj_dim + spec.hidden_size) * w + spec.hidden_size * 4.0
                kernels["embed"]["frac_hbm_survey_bytes"] = kernels["embed"]["frac_hbm"] * survey_b / ours_b
            # PMC artefacts (separate rocprofv3 --pmc passes, diag/refresh_profiles.sh): used only when they were taken
            # on THESE kernel sources - the files carry the tag and the source hash of their pass
            sha = kernel_sources_sha16()
            traffic, traffic_src, pmc = None, None, None
            # (taken on THESE kernel sources and on THIS workload: the headline workload's passes live in gemm_traffic.json /
            # gemm_pmc.json, another workload's - BASELINE config 3 - in gemm_traffic.<model>_<profile>_<n>.json)
            wl = f"{args.model}/{args.profile}/{n_local}"
            suf = "" if wl == "125m/sharegpt/8192" else "." + wl.replace("/", "_")
            tpath, ppath = os.path.join(ROOT, "profiles", f"gemm_traffic{suf}.json"), os.path.join(ROOT, "profiles", f"gemm_pmc{suf}.json")
            try:
                tj = json.load(open(tpath))
                traffic_src = tj.get("source")
                if traffic_src and traffic_src.get("kernel_sha16") == sha and traffic_src.get("workload", "125m/sharegpt/8192") == wl:
                    traffic = tj.get("hbm_bytes_per_launch")
            except Exception:
                pass
            try:
                pj = json.load(open(ppath))
                if (pj.get("source") or {}).get("kernel_sha16") == sha and (pj.get("source") or {}).get("workload", "125m/sharegpt/8192") == wl:
                    pmc = pj.get("gemm_f16s_kernel")
            except Exception:
                pass
            avg_launch_s = gemm["ms"] / max(gemm["launches"], 1) * 1e-3
            peak = PEAK_F16_MFMA_TFLOPS if f16 else 157.3
            launches_per_step = gemm["launches"] // max(args.steps, 1)
            passes = max(1, -(-int(cu[my_r1] - cu[my_r0]) // 196608))
            comp = compulsory_bytes(spec, int(cu[my_r1] - cu[my_r0]), passes) if args.weight_dtype == "f16" else None
            n_pass = 2 if args.weight_dtype == "f16" else 1              # fp16 MFMA passes per product (lo, hi)
            roof = {"bound": "mfma", "kernel": "gemm_f16s_kernel" if f16 else "gemm_f32_kernel",
                    "achieved": gemm_tflops, "peak": peak, "unit": "TFLOP/s", "frac": gemm_tflops / peak,
                    # `frac` is ALGORITHMIC (2 M N K per product).  The split path issues every product twice (lo x W, hi x W):
                    # the matrix cores run at hw_frac of their fp16 peak
                    "mfma_passes_per_product": n_pass, "hw_tflops": gemm_tflops * n_pass, "hw_frac": gemm_tflops * n_pass / peak,
                    # the same MFMA instruction stream alone, fed from registers with real operands, under the 1.4 kW package
                    # cap (1.98 PFLOP/s of 16x16x32 fp16 MFMA at the throttled clock, profiles/r01_power_probe.txt): the time the
                    # GEMMs of one step would take if the operand stream, LDS reads and epilogues cost nothing
                    "mfma_only_floor_ms": gemm["work"] / max(args.steps, 1) * n_pass / 1.98e15 * 1e3 if f16 else None,
                    "traffic": traffic,
                    # bytes the GEMM launches of one call MUST move in this layout (compulsory_bytes()) against
                    # what the PMC passes counted at the fabric for the same launches
                    "compulsory_bytes": comp, "compulsory_bytes_per_launch": comp / launches_per_step if comp and launches_per_step else None,
                    "traffic_ratio": traffic * launches_per_step / comp if traffic and comp else None,
                    "traffic_note": "fabric-side (TCC_EA requests, calibrated): Infinity-Cache hits included - an upper bound on DRAM traffic",
                    "traffic_source": traffic_src if traffic is not None else
                    {"stale": True, "reason": "profiles/gemm_traffic.json was not taken on the current kernel sources and workload",
                     "file": traffic_src, "current_kernel_sha16": sha},
                    # launches of THIS kernel only (the 128 x 256-tile one: the rocprofv3 row of the same name); the compact
                    # last-token rows of each pass run on the small-batch kernels, timed apart as kernels.gemm_small
                    "launches_per_step": launches_per_step,
                    "avg_launch_ms": avg_launch_s * 1e3,
                    # the same kernel against the OTHER roof: PMC bytes per launch / live launch time, over 8 TB/s.
                    # The split-fp16 design moves 4 B per activation element between launches, so the MFMA-bound
                    # kernel is also a heavy HBM client (DESIGN.md 4.1 "bytes")
                    "hbm_gbs": traffic / avg_launch_s / 1e9 if traffic else None,
                    "hbm_frac": traffic / avg_launch_s / (PEAK_HBM_GBS * 1e9) if traffic else None,
                    "measured": "second pass of the same K steps with HIP events around every launch (profiler off in the timed region)",
                    "note": "the GEMM launches also carry the LayerNorm work of the layer (LayerNorm fold: operand + row "
                            "statistics in the producer epilogue, normalisation in the consumer epilogue); "
                            "`unfused` = the same forward with separate LayerNorm launches (LTR_F_NO_LN_FOLD)",
                    "unfused": unfused,
                    # MFMA pipe utilisation / L2 hit rate of the kernel from the last PMC passes (profiles/gemm_pmc.json)
                    "pmc": pmc}
        # steady rank step (nothing new to score): 37 B per request per step algorithmic
        rk_us = rank_ms[len(rank_ms) // 2] * 1e3
        if class_head:
            kernels["class_head"] = class_head
        kernels["rank_step"] = dict(us_per_step=rk_us, launches_per_step=2 if n_total <= 12288 else 9,
                                    gbs=37.0 * n_total / (rk_us * 1e-6) / 1e9,
                                    frac_hbm=37.0 * n_total / (rk_us * 1e-6) / 1e9 / PEAK_HBM_GBS,
                                    note="latency-bound: two dependent launches over a few hundred KB")
        out = {
            "metric": "requests ranked/sec (cold call: OPT predictor forward + priority sort/aging)",
            "value": n_total * args.steps / elapsed,
            "unit": "requests/s",
            "n_gpus": world, "rccl_ranks": dist.get_world_size() if dist is not None else 1, "backend": backend,
            "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "profiled_ms_per_step": prof_elapsed / args.steps * 1e3 if prof_elapsed else None,
            "p50_rank_latency_ms": step_ms[len(step_ms) // 2],
            "p50_steady_rank_latency_ms": rank_ms[len(rank_ms) // 2],
            # steady call with k new requests: score k + re-rank the whole queue (one GPU; SURVEY 8d "steady")
            "p50_steady_new_latency_ms": steady or None,
            "host_inclusive": host_incl,
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
            "dtype": "f16 weights x (hi+lo) f16 activations, f32 accumulate" if f16 else "f32",
            "data": "synthetic (seeded random-init OPT checkpoint, lognormal prompt lengths, random token ids)",
            "config": {"workload": (f"OPT-{args.model} predictor, fixed {n_total}-request synthetic queue ({args.profile} length "
                                    f"profile), cold ranker call" if strong else
                                    f"OPT-{args.model} predictor, {n_local} synthetic queue per GPU ({n_total} total, "
                                    f"{args.profile} length profile), cold ranker call"),
                       "queue_per_gpu": n_local, "queue_total": n_total, "tokens_total": int(cu[-1]),
                       "tokens_rank0_shard": int(cu[my_r1] - cu[my_r0]), "tokens_shard": call_tokens_shard,
                       "sharded": call_sharded, "min_shard_tokens": args.min_shard_tokens, "starv": args.starv, "period": args.period,
                       "parallelism": f"request-sharded dp{world}" + (", token-balanced shards of one global queue, "
                                                                       "RCCL all-gather of scores" if world > 1 else "")},
            "roofline": roof,
            "kernels": kernels,
            "strong_scaling": strong_pt,
            # north_star: "ranker calls/sec on synthetic 1k-64k-request queues": cold calls on FIXED queues at this N
            # (with the headline = the 8k point and strong_scaling = the 64k point: five queue sizes)
            "scale_table": scale_pts,
            "config3": config3,
            "model_tflop_per_step": (lin + att) / 1e12,
            "input_distribution": ((f"driver-broadcast: only rank 0 holds the queue; header ({shared.header_channel}) + "
                                    + ("one per-rank scatter of (cu_seqlens slice, token ids)" if shared.distribution == "scatter" else
                                       f"ONE broadcast of the whole (cu_seqlens, token ids) payload [fallback: {shared.distribution_note}]")
                                    + " inside every timed step") if args.driver_broadcast and world > 1
                                   else "resident: every rank holds the whole queue before the timed region (SPMD)"),
            "preflight": preflight,
        }
        if one_pass:
            out["metric"] = ("requests ranked/sec, ONE fp16 MFMA pass per product (the reference's fp16 GPU arithmetic, ~2e-3 from "
                             "the fp32 predictor: NOT the 1e-4 headline metric)")
            out["dtype"] = "f16 weights x f16 activations (one MFMA pass), f32 accumulate"
        if not args.no_cpu_baseline and world == 1:          # rank 0 at N = 1 only (the other ranks would idle)
            out["cpu_baseline"] = cpu_baseline(spec, ckpt, ids, cu, args.starv, args.period)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
