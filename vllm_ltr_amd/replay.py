"""Ranker-side replay of an arrival trace (BASELINE config 5, ranker's share only).

The reference measures head-of-line latency end to end: requests arrive either all at t = 0
(``benchmarks/burst-*.sh`` -> ``benchmark_throughput_original.py``) or with gamma-distributed
inter-arrival times (``benchmarks/benchmark_serving_real.py:159-176``: ``shape = 1 / cv^2``,
``scale = cv^2 / request_rate``), the engine loop calls ``Scheduler.schedule()`` once per iteration
(``vllm/entrypoints/llm.py:232-241``) and ``output.HOL = time_in_queue`` (``llm.py:237``,
``vllm/sequence.py:486``: first scheduled time - arrival time).  The backbone (Llama-3-8B + paged KV
cache) is out of scope here; what IS on the ranking path is what every one of those iterations pays
inside ``schedule()``: k arrivals -> ``obtain_aux_scores(k)``, the promote/demote + sort over
``waiting + running + swapped``, the budget walk and the aging.  This module replays exactly that through
:class:`~vllm_ltr_amd.plugin.MI355XRanker` behind ``install()`` on duck-typed request objects, with a
stand-in for the rest of the iteration (a fixed virtual ``backbone_ms`` per step; prefill chunks and one
decode token per scheduled request per step), and records the ranker's wall time per step.

Virtual clock: ``t += ranker wall time of the step + backbone_ms``; a request joins ``waiting`` in the
first step whose start time is >= its arrival time.  ``ranker_delay`` of a request = the ranker wall
time accumulated over the steps between its arrival and its first scheduling = the ranker-induced part
of its head-of-line time ``HOL = first_scheduled - arrival``.  The ranker's time of a step = its
``_schedule()`` - what blocks the engine loop; the arrival hooks (``add_request``) are timed separately
(``arrival_hook_ms``): in the reference's serving path they run on the event-loop thread while the backbone step
runs in the executor thread (async_llm_engine.py: ``step_async`` -> ``execute_model_async``), i.e. beside the step.
``ranker_ms_*_incl_hooks`` adds them to the step for the synchronous engine (entrypoints/llm.py), where they are not.

Scoring at arrival (``MI355XRanker(prescore=True)``): a request that arrives at virtual time ``a`` during the
backbone step is handed to ``add_request`` there and the scheduler step starts at ``t >= a``; the forward has
``t - a`` of the engine's time to finish.  The replay gives it that lead in REAL time: the gap from each arrival to the
next one / to the start of the step, each capped at ``lead_cap_ms`` (6 ms of a 25-ms backbone step: a few one-request
forwards; a longer gap changes nothing, a shorter one is honoured exactly - the replay never grants more overlap than
the trace has).
"""
from __future__ import annotations

import time
from collections import deque
from types import SimpleNamespace
from typing import Callable, Dict, List, Optional, Sequence

import numpy as np
import torch


class ReplayRequest:
    """The SequenceGroup fields the ranking path touches (vllm/sequence.py:426-465,
    vllm/core/scheduler.py:372-374) plus the replay's own bookkeeping."""

    def __init__(self, request_id: str, prompt_token_ids: Sequence[int], arrival: float, output_len: int):
        self.request_id = request_id
        self.prompt_token_ids = list(prompt_token_ids)
        self.prompt = None
        self.aux_model_score = None
        self.pri = self.idle = self.runs = 0                 # scheduler.py:372-374
        self.arrival = float(arrival)
        self.output_len = int(output_len)
        self.prefill_left = len(self.prompt_token_ids)
        self.decode_left = int(output_len)
        self.first_scheduled: Optional[float] = None
        self.ranker_delay = 0.0

    def need_aux_model_score(self):
        return self.aux_model_score is None

    def set_aux_model_score(self, s):
        self.aux_model_score = s


def arrival_times(n: int, kind: str, request_rate: float = 16.0, cv: float = 1.0, seed: int = 0) -> np.ndarray:
    """``burst``: everything at t = 0 (burst-*.sh).  ``gamma``: benchmark_serving_real.py:159-176."""
    if kind == "burst" or request_rate == float("inf"):
        return np.zeros(n)
    if kind != "gamma":
        raise ValueError(f"trace kind {kind!r}: burst | gamma")
    rs = np.random.RandomState(seed)
    shape, scale = 1.0 / (cv * cv), cv * cv / request_rate
    iv = rs.gamma(shape, scale, n)
    iv[0] = 0.0                                              # the first request is sent immediately (:166-168)
    return np.cumsum(iv)


def synthetic_trace(vocab_size: int, n: int, kind: str, request_rate: float = 16.0, cv: float = 1.0, seed: int = 0,
                    prompt_median: float = 64.0, output_median: float = 128.0, max_prompt: int = 1024) -> List[ReplayRequest]:
    """Prompts with the BASELINE length profile (lognormal, clip [4, 1024]: the benchmark's filter,
    benchmark_serving_real.py:141-147) and lognormal output lengths."""
    rs = np.random.RandomState(seed)
    pl = np.clip(np.rint(np.exp(rs.normal(np.log(prompt_median), 0.8, n))), 4, max_prompt).astype(np.int64)
    ol = np.clip(np.rint(np.exp(rs.normal(np.log(output_median), 1.0, n))), 4, 2048).astype(np.int64)
    at = arrival_times(n, kind, request_rate, cv, seed + 1)
    reqs = []
    for i in range(n):
        ids = rs.randint(4, vocab_size, int(pl[i])).tolist()
        ids[0] = 2
        reqs.append(ReplayRequest(str(i), ids, at[i], int(ol[i])))
    return reqs


class ReplayScheduler:
    """The three deques and the loop shape of ``Scheduler._general_schedule`` (scheduler.py:1101-1373) without the
    block manager: order (through the installed ranker), budget walk over the order with chunked prefill
    (:1137-1211), hand the step's ``scheduled_seq_groups`` back.  ``install()`` wraps ``_schedule`` with the aging."""

    def __init__(self, max_num_batched_tokens: int = 2048, max_num_seqs: int = 256):
        self.waiting, self.running, self.swapped = deque(), deque(), deque()
        self.max_num_batched_tokens, self.max_num_seqs = max_num_batched_tokens, max_num_seqs
        self.last_order: list = []
        self.aux_model = None

    def _general_schedule(self):
        self._update_priority()                              # :1103
        order = self._get_ordered_requests()                 # :1105
        self.last_order = order
        tokens = seqs = 0
        sel = []
        for g in order:                                      # :1137-1211 (every group has one sequence: chunkable)
            need = g.prefill_left if g.prefill_left > 0 else 1
            n = min(need, self.max_num_batched_tokens - tokens)
            if n == 0 or seqs + 1 > self.max_num_seqs:
                break
            tokens += n
            seqs += 1
            sel.append((g, n))
        chosen = {id(g) for g, _ in sel}
        # queue moves of the step: selected requests run; running requests that were not selected are preempted by
        # swapping (the reference's priority swap, :1376-1452); everything else stays where it is
        self.swapped = deque(g for g in list(self.swapped) + list(self.running) if id(g) not in chosen)
        self.waiting = deque(g for g in self.waiting if id(g) not in chosen)
        self.running = deque(g for g, _ in sel)
        return SimpleNamespace(scheduled_seq_groups=[SimpleNamespace(seq_group=g, token_chunk_size=n) for g, n in sel])


def replay(ranker, requests: List[ReplayRequest], backbone_ms: float = 25.0, max_num_batched_tokens: int = 2048,
           max_num_seqs: int = 256, max_steps: int = 1 << 20, on_step: Optional[Callable] = None,
           before_step: Optional[Callable] = None, lead_cap_ms: float = 6.0) -> Dict:
    """Run the trace to completion.  Returns per-step ranker wall times (ms), the number of arrivals scored per step,
    the queue length per step and per-request HOL / ranker_delay (s).  ``on_step(step, scheduler, ran)`` is called
    after every step, ``before_step(step, scheduler)`` right before its ``_schedule()`` with the arrivals already in
    ``waiting`` (tests replay the literal reference expressions on the same deques and compare ``last_order``)."""
    sched = ReplayScheduler(max_num_batched_tokens, max_num_seqs)
    ranker.install(sched)
    pending = deque(sorted(requests, key=lambda r: r.arrival))
    t = 0.0
    step_ms, step_new, step_queue, step_hook_ms = [], [], [], []
    finished = 0
    prescore = bool(getattr(ranker, "prescore", False))
    n = len(requests)
    dev = ranker.device
    for step in range(max_steps):
        if finished == n:
            break
        if not (sched.waiting or sched.running or sched.swapped) and pending and pending[0].arrival > t:
            t = pending[0].arrival                           # idle engine: jump to the next arrival
        k = 0
        hook = 0.0
        lead_cap = lead_cap_ms * 1e-3
        while pending and pending[0].arrival <= t:
            r = pending.popleft()
            h0 = time.perf_counter()
            ranker.add_request(r)                            # arrival-time hook (tokenise / truncate once; prescore: launch)
            hook += time.perf_counter() - h0
            sched.waiting.append(r)
            k += 1
            if prescore:
                # the engine time between this arrival and the next one / the start of the step, in real time (each gap capped)
                nxt = pending[0].arrival if pending and pending[0].arrival <= t else t
                lead = min(max(nxt - r.arrival, 0.0), lead_cap)
                if lead > 0:
                    end = h0 + lead                          # the lead counts from the ARRIVAL: the hook's own time is part of it
                    while time.perf_counter() < end:         # (sleep() overshoots by tens of microseconds)
                        pass
        qlen = len(sched.waiting) + len(sched.running) + len(sched.swapped)
        unscheduled = [g for g in sched.waiting if g.first_scheduled is None]
        if before_step is not None:
            before_step(step, sched)
        if not prescore:
            torch.cuda.synchronize(dev)                      # (prescore: the forward of the arrivals may still be running - that is the point)
        t0 = time.perf_counter()
        ret = sched._schedule()                              # obtain_aux_scores(k) + order + budget walk + aging
        torch.cuda.current_stream(dev).synchronize()
        dt = time.perf_counter() - t0
        step_ms.append(dt * 1e3); step_new.append(k); step_queue.append(qlen); step_hook_ms.append(hook * 1e3)
        ran = [x.seq_group for x in ret.scheduled_seq_groups]
        t_sched = t + dt
        for g in unscheduled:                                # the step's ranker time precedes any scheduling it decides
            g.ranker_delay += dt
        for x in ret.scheduled_seq_groups:
            g = x.seq_group
            if g.first_scheduled is None:
                g.first_scheduled = t_sched
            if g.prefill_left > 0:
                g.prefill_left -= x.token_chunk_size
            else:
                g.decode_left -= 1
        if on_step is not None:
            on_step(step, sched, ran)
        done = [g for g in sched.running if g.prefill_left <= 0 and g.decode_left <= 0]
        if done:
            gone = {id(g) for g in done}
            sched.running = deque(g for g in sched.running if id(g) not in gone)
            finished += len(done)
        t = t_sched + backbone_ms * 1e-3
    hol = np.array([r.first_scheduled - r.arrival for r in requests if r.first_scheduled is not None])
    delay = np.array([r.ranker_delay for r in requests if r.first_scheduled is not None])
    return dict(step_ms=np.asarray(step_ms), step_new=np.asarray(step_new), step_queue=np.asarray(step_queue),
                step_hook_ms=np.asarray(step_hook_ms),
                hol_s=hol, ranker_delay_s=delay, finished=finished, virtual_seconds=t)


def summarize(res: Dict) -> Dict:
    """p50 / p95 / p99 of the ranker's wall time per scheduler step (all steps, steps that scored arrivals, steady
    steps) and of the ranker-induced head-of-line delay."""
    pct = lambda a, q: float(np.percentile(a, q)) if len(a) else None
    ms, new = res["step_ms"], res["step_new"]
    out = {"steps": int(len(ms)), "finished": int(res["finished"]), "virtual_seconds": float(res["virtual_seconds"]),
           "max_queue": int(res["step_queue"].max()) if len(ms) else 0,
           "arrivals_per_scoring_step_mean": float(new[new > 0].mean()) if (new > 0).any() else 0.0}
    for name, sel in (("all", np.ones(len(ms), bool)), ("with_arrivals", new > 0), ("steady", new == 0)):
        a = ms[sel]
        out[f"ranker_ms_{name}"] = dict(n=int(sel.sum()), p50=pct(a, 50), p95=pct(a, 95), p99=pct(a, 99),
                                        max=float(a.max()) if len(a) else None)
    if "step_hook_ms" in res:
        hk = res["step_hook_ms"]
        out["arrival_hook_ms_per_scoring_step"] = dict(p50=pct(hk[new > 0], 50), p99=pct(hk[new > 0], 99))
        a = (ms + hk)[new > 0]
        out["ranker_ms_with_arrivals_incl_hooks"] = dict(n=int((new > 0).sum()), p50=pct(a, 50), p95=pct(a, 95), p99=pct(a, 99))
    d, h = res["ranker_delay_s"], res["hol_s"]
    out["ranker_hol_delay_ms"] = dict(p50=pct(d * 1e3, 50), p95=pct(d * 1e3, 95), p99=pct(d * 1e3, 99))
    out["hol_ms"] = dict(p50=pct(h * 1e3, 50), p95=pct(h * 1e3, 95), p99=pct(h * 1e3, 99))
    out["ranker_share_of_hol"] = float(d.sum() / h.sum()) if len(h) and h.sum() > 0 else None
    return out
