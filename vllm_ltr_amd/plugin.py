"""Drop-in for the reference scheduler's predictor plug-in surface.

What the reference wires (all paths relative to the reference checkout):

* ``LLMEngine.__init__`` builds an ``AUXLLM`` when a predictor config is given and assigns
  it to ``scheduler.aux_model`` (vllm/engine/llm_engine.py:224-242);
* ``Scheduler._get_opt_ordered_requests`` (vllm/core/scheduler.py:969-1000) calls
  ``self.aux_model.obtain_aux_scores(need_aux_scores)``, then promotes/demotes starved
  requests and stable-sorts ``waiting+running+swapped`` by ``(pri, -aux_model_score)``;
* after the budget walk, ``_general_schedule`` ages ``idle/runs`` (:1358-1365).

:class:`MI355XRanker` provides the same surface on MI355X:

* ``obtain_aux_scores(seq_groups)`` - same name/contract as ``AUXLLM.obtain_aux_scores``
  (vllm/entrypoints/aux_llm.py:125-126 -> vllm/engine/aux_llm_engine.py:332-412): sets
  ``aux_model_score`` on every group via ``set_aux_model_score`` and returns the scores;
* ``ordered_requests(scheduler)`` - the body of ``_get_opt_ordered_requests`` (and the
  ``tpt/rtpt/ropt/xpt`` variants) with the promote/demote + sort done by ``ltr_rank_step``;
* ``age(all_pri, running_this_step)`` - the aging loop done by ``ltr_age_update``;
* ``install(scheduler)`` - assigns ``scheduler.aux_model`` and rebinds
  ``scheduler._schedule / _get_ordered_requests`` exactly where ``scheduler.py:319-329`` binds them.

**Device-resident queue** (SURVEY.md 7).  The ranking state of a request - score, ``pri``, ``idle``,
``runs`` - lives in a slot of a :class:`~vllm_ltr_amd.rank.DeviceQueue` from the first time the ranker
sees the request until it leaves the scheduler's deques.  Per scheduler step the host only (1) lists the
deques, (2) reads the slot number off each request (one C-level pass), (3) uploads that ``members`` array
(4 B per request) and (4) builds the output list from the returned permutation; ``age`` uploads the slots
of the <= ``max_num_seqs`` requests that ran.  Nothing in the reference outside scheduler.py:984-993 and
:1358-1365 reads ``pri/idle/runs``, so the host attributes are refreshed only on request
(:meth:`sync_host`, or ``mirror_host=True`` to write them back every step like the reference does).

Requests are duck-typed on the fields the reference touches: ``request_id``,
``aux_model_score`` / ``need_aux_model_score()`` / ``set_aux_model_score()``
(vllm/sequence.py:429,461-465), ``pri / idle / runs`` (scheduler.py:372-374), and the
prompt (``prompt`` text or ``prompt_token_ids``).
"""
from __future__ import annotations

import collections
import itertools
import operator
import os
import time
from typing import Callable, Iterable, List, Optional, Sequence

import numpy as np
import torch

from . import _lib
from .config_predictor import PrefillPredictorConfig
from .host_pipeline import InputStager, PinnedI32 as _PinnedI32, cached_token_ids
from .opt_spec import checkpoint_weight_dtype, load_hf_checkpoint
from .prescore import PrescoreMixin
from .rank import DeviceQueue, RankWorkspace, budget_prefix, rank_step, reserve_select
from .schedule_type import ScheduleType, parse_schedule_type
from .scorer import HipOPTScorer

_ranker_ids = itertools.count()


def _string_rank(request_ids: Sequence[str]) -> np.ndarray:
    """Rank of each request_id under Python string comparison (the ``tpt`` tiebreak,
    scheduler.py:948: ``key=(-score, req.request_id)``)."""
    order = sorted(range(len(request_ids)), key=request_ids.__getitem__)
    rank = np.empty(len(request_ids), np.int32)
    rank[order] = np.arange(len(request_ids), dtype=np.int32)
    return rank


class MI355XRanker(PrescoreMixin):

    def __init__(self, scorer: HipOPTScorer, schedule_type: str = "opt", max_length: int = 2048,
                 tokenize: Optional[Callable[[str], List[int]]] = None, mtype: str = "rank",
                 xpt_distribution=None, group=None, min_requests_to_shard: Optional[int] = None,
                 min_tokens_to_shard: Optional[int] = None, collective_timeout_s: Optional[float] = None,
                 mirror_host: bool = False, prescore: bool = False, prescore_graphs: bool = True,
                 driver_rank: Optional[int] = None):
        """
        scorer      the HBM-resident predictor
        schedule_type  the reference's schedule string (``opt-...-starv<S>-period<P>``; an ``xpt{path}...``
                    string names its score -> expected-length table like scheduler.py:312)
        max_length  prompt truncation, ``PrefillModelConfig.max_length`` = the AUX engine's
                    ``max_model_len`` (aux_llm_engine.py:365-369, llm_engine.py:236)
        tokenize    text -> predictor token ids.  The reference re-tokenises the prompt
                    TEXT with the predictor's (OPT) tokenizer (aux_llm_engine.py:341,365-369);
                    pass that tokenizer's ``encode`` here.  None: use the request's
                    ``prompt_token_ids`` (valid when backbone and predictor share a
                    tokenizer, and for synthetic workloads).
        group       a ``torch.distributed`` process group: ``obtain_aux_scores`` then shards the batch over the
                    ranks of the group (every rank must make the same call, SPMD) with one all-gather of the
                    scores (RCCL over xGMI); the reference instead runs the predictor tensor-parallel over the
                    backbone's TP group (llm_engine.py:237).  None: this GPU scores alone.
        driver_rank  None: SPMD - every rank of ``group`` makes the same ``obtain_aux_scores`` call with the same batch.
                    An int: the vllm-ltr engine's shape - only THAT rank of the group has a scheduler and calls
                    ``obtain_aux_scores``; the other ranks build the same ranker and sit in :meth:`serve` (what
                    ``execute_aux_method`` is to the reference's workers, worker_base.py:151-166): the driver sends each
                    worker its shard of the batch (header broadcast + one scatter, distributed.py), all score, one
                    all-gather; calls below the shard threshold involve no worker.  :meth:`close` ends the workers' loops.
        min_tokens_to_shard  shard a scoring call only when it holds more tokens than this (default: one pass of one GPU,
                    196,608 - north_star: "only when the queue exceeds a single GPU's batch"); ``min_requests_to_shard``
                    replaces it with a rule on the request count.  ``collective_timeout_s`` bounds every collective of a
                    sharded call: a dead peer raises :class:`~vllm_ltr_amd.distributed.PeerTimeout` on the surviving ranks.
        mirror_host write ``pri/idle/runs`` back to the request objects every step (what the reference's loops
                    do; costs a D2H copy + a Python loop per step).  Default: device-resident only.
        prescore    score a request when it ARRIVES (``add_request``, where ``Scheduler.add_seq_group`` runs) instead of
                    inside the scheduler step: the forward is launched asynchronously on a second stream while the engine
                    is busy with the backbone step, and ``obtain_aux_scores`` - which blocks the engine loop
                    (SURVEY.md 8b "Threading") - only collects what is already there.  Arrivals that come while a
                    forward is in flight are batched into the next one.  Same kernels, same scores (up to the batch a
                    request is scored in: the <= 2e-6 of DESIGN.md 4.1b); not with ``group=`` (every rank must make the
                    same collective calls).  Off by default: the reference scores at the step.  ``prescore_graphs``: a lone
                    arrival's forward is replayed from a captured graph (one per 64-token bucket; ~30 us of host time in
                    the hook instead of the ~0.3 ms its launches take one by one); False: always launch eagerly.
        """
        self.scorer = scorer
        self.device = scorer.device
        self.st: ScheduleType = parse_schedule_type(schedule_type)
        self.max_length = int(max_length)
        self.tokenize = tokenize
        self.mtype = mtype
        # xpt policy: (key, value) score -> expected-length table, scheduler.py:312 (torch.load of the {path})
        self.xpt_distribution = xpt_distribution
        if self.xpt_distribution is None and self.st.policy == "xpt" and self.st.table_path:
            self.xpt_distribution = torch.load(self.st.table_path)
        if mtype == "rank" and scorer.spec.num_labels != 1:
            raise ValueError("mtype 'rank' needs num_labels == 1 (prefill_predictor.py:35-36)")
        self.mirror_host = bool(mirror_host)
        self._ws = RankWorkspace(self.device)
        self._stager = InputStager(self.device)
        self._sharded = None
        if group is not None:
            from .distributed import ShardedScorer
            self._sharded = ShardedScorer(self.scorer, self.device, group=group,
                                          min_requests_to_shard=min_requests_to_shard,
                                          min_tokens_to_shard=min_tokens_to_shard, timeout_s=collective_timeout_s,
                                          driver_rank=0 if driver_rank is None else driver_rank)
        self.driver_rank = driver_rank if group is not None else None
        starv, period = (self.st.starv, self.st.period) if self.st.policy == "opt" else (-1, 0)
        self.queue = DeviceQueue(self.device, starv=starv, period=period, capacity=1 << 13)
        # the slot number lives on the request object under a per-ranker attribute name (two rankers - e.g. in a
        # test - must not read each other's slots); attrgetter keeps the per-step pass over the queue in C
        self._slot_attr = f"_ltr_slot{next(_ranker_ids)}"
        self._get_slot = operator.attrgetter(self._slot_attr)
        self._members = _PinnedI32(self.device)
        self._perm = _PinnedI32(self.device)
        self._ran = _PinnedI32(self.device, 1 << 10)
        self._n_members = 0
        self._members_dev: Optional[torch.Tensor] = None
        self._live_slots = 0
        self._aged = True                    # False between an order() and the age() that belongs to it
        self._last_reqs: Sequence = ()
        self._last_perm_dev: Optional[torch.Tensor] = None
        self._order_ran = False
        self.stats = dict(aux_calls=0, requests_scored=0, rank_calls=0, score_seconds=0.0, rank_seconds=0.0)
        # wall time of the last 1,000 scoring / ordering calls ("calls, requests scored, ms/call")
        self._score_ms: collections.deque = collections.deque(maxlen=1000)
        self._rank_ms: collections.deque = collections.deque(maxlen=1000)
        self._init_prescore(prescore, prescore_graphs, group, driver_rank)

    def __del__(self):
        # the prescore scratch this ranker parked on the (possibly shared) scorer
        try:
            for k in (self._pre_ws_key, self._pre_ws_key + "-graph"):
                self.scorer._ws_by_key.pop(k, None)
        except Exception:      # noqa: BLE001 - interpreter shutdown / partially constructed object
            pass

    # ---- construction from the reference's config objects ------------------------------
    @classmethod
    def from_predictor_config(cls, cfg, schedule_type: str, device: str = "cuda:0", tokenize=None,
                              weight_dtype: str = "auto", **kw) -> "MI355XRanker":
        """``cfg``: a :class:`PrefillPredictorConfig` or a path to its JSON
        (``--prefill-predictor-model-config``, arg_utils.py:346-359).  Loads the HF
        checkpoint at ``cfg.model.path`` like llm_engine.py:228-240 does for the AUXLLM.
        Extra keywords (``xpt_distribution``, ``group``, ...) go to the constructor."""
        if isinstance(cfg, (str, os.PathLike)):
            cfg = PrefillPredictorConfig.from_json(cfg)
        from .trainer import refuse_activation
        refuse_activation(cfg.model.activation, "MI355XRanker.from_predictor_config")
        spec, ckpt = load_hf_checkpoint(cfg.model.path)
        if weight_dtype == "auto":       # fp16 checkpoint (trainer.py:215) -> split-fp16 path; fp32 -> exact f32 path
            weight_dtype = checkpoint_weight_dtype(ckpt)
        scorer = HipOPTScorer(spec, ckpt, device=device, weight_dtype=weight_dtype)
        return cls(scorer, schedule_type, max_length=cfg.model.max_length, tokenize=tokenize,
                   mtype=cfg.model.mtype, **kw)

    # ---- slots ------------------------------------------------------------------------------
    def _assign_slots(self, reqs: Sequence, being_scored: bool = False) -> None:
        """Give every request of ``reqs`` that has none a slot.  A request that already carries a score (set by
        someone else, or before this ranker was installed) gets it - and its counters - uploaded."""
        attr = self._slot_attr
        fresh = [r for r in reqs if not hasattr(r, attr)]
        if not fresh:
            return
        if not being_scored:
            for r in fresh:
                if getattr(r, "aux_model_score", None) is None:
                    # the reference fails here too: sorted() would negate None (scheduler.py:996)
                    raise TypeError(f"request {r.request_id} reaches the ordering without an aux_model_score")
        slots = self.queue.alloc_slots(len(fresh))
        have = []
        for r, s in zip(fresh, slots):
            setattr(r, attr, int(s))
            if getattr(r, "aux_model_score", None) is not None:
                have.append((int(s), float(r.aux_model_score), int(getattr(r, "pri", 0)), int(getattr(r, "idle", 0)),
                             int(getattr(r, "runs", 0))))
        self._live_slots += len(fresh)
        if have:     # adopted mid-flight: carry score and counters over
            a = np.asarray(have, np.float64)
            idx = torch.from_numpy(a[:, 0].astype(np.int64)).to(self.device)
            self.queue._score.index_copy_(0, idx, torch.from_numpy(a[:, 1].astype(np.float32)).to(self.device))
            for col, t in ((2, self.queue._pri), (3, self.queue._idle), (4, self.queue._runs)):
                t.index_copy_(0, idx, torch.from_numpy(a[:, col].astype(np.int32)).to(self.device))

    def _collect_members(self, reqs: Sequence) -> int:
        """Slots of ``reqs`` into the pinned members buffer (one C-level pass); returns n."""
        n = len(reqs)
        self._members.ensure(n)
        try:
            self._members.np[:n] = np.fromiter(map(self._get_slot, reqs), np.int32, n)
        except AttributeError:           # requests the ranker has not seen yet (adopted mid-flight, direct order() calls)
            self._assign_slots(reqs)
            self._members.np[:n] = np.fromiter(map(self._get_slot, reqs), np.int32, n)
        return n

    def _gc_slots(self, n_members: int) -> None:
        """Requests that left the deques (finished / aborted) never come back: reclaim their slots once the
        live count has drifted well past the queue size."""
        if self._live_slots <= 2 * n_members + 1024:
            return
        used = np.zeros(self.queue.n, bool)
        used[self._members.np[:n_members]] = True
        used[np.asarray(self.queue._free, np.int64)] = True
        dead = np.nonzero(~used)[0]
        self.queue.free_slots(dead.tolist())
        self._live_slots -= len(dead)

    # ---- AUXLLM.obtain_aux_scores -------------------------------------------------------
    def add_request(self, sg) -> None:
        """Optional arrival-time hook (where ``Scheduler.add_seq_group`` runs, scheduler.py:368-376):
        tokenise / truncate the prompt once so the scoring call only packs cached arrays.  With ``prescore=True`` the
        request's forward is also started here, asynchronously (see the constructor)."""
        cached_token_ids(sg, self.tokenize, self.max_length)
        if self.prescore and sg.need_aux_model_score() and getattr(sg, "_ltr_pre", None) is None:
            t0 = time.perf_counter()
            self._pre_pending.append(sg)
            self._prescore_pump()
            self.stats["arrival_hook_seconds"] += time.perf_counter() - t0

    def obtain_aux_scores(self, seq_groups) -> List[float]:
        seq_groups = list(seq_groups)
        if not seq_groups:
            return []
        t0 = time.perf_counter()
        for sg in seq_groups:
            assert sg.need_aux_model_score()               # aux_llm_engine.py:409
        out: List[Optional[float]] = [None] * len(seq_groups)
        todo = list(range(len(seq_groups)))
        if self.prescore:
            todo = self._collect_prescored(seq_groups, out)
        if todo:
            batch = [seq_groups[i] for i in todo]
            for i, v in zip(todo, self._score_now(batch)):
                out[i] = v
        try:
            self._check_status()                           # out-of-vocabulary ids raise, like F.embedding
        except _lib.LtrError as e:
            if e.code != _lib.LTR_E_RANGE:
                raise
            # The residual stream of this checkpoint left the fp16 range of the LayerNorm-fold operand somewhere in this
            # call (or in a forward started at arrival): its scores are invalid.  The reference would carry on (its fp16
            # path overflows only at |x| > 65504) and an exception here ends the engine (llm_engine.py:569): re-score the
            # whole batch on a handle that feeds the GEMMs the bounded LayerNorm output, keep using that handle.
            out = self._rescore_unfolded(seq_groups)
        for sg, s in zip(seq_groups, out):
            sg.set_aux_model_score(s)                      # aux_llm_engine.py:408-410
        dt = time.perf_counter() - t0
        self.stats["aux_calls"] += 1
        self.stats["requests_scored"] += len(seq_groups)
        self.stats["score_seconds"] += dt
        self._score_ms.append(dt * 1e3)
        return out

    def _score_now(self, batch) -> List[float]:
        """One forward over ``batch`` (this GPU alone, or sharded over the group): scores into the requests' device slots,
        host values returned."""
        arrays = [cached_token_ids(sg, self.tokenize, self.max_length) for sg in batch]
        ids_dev, cu_dev, cu_host = self._stager.stage(arrays)          # pinned pack + async H2D
        if self._sharded is not None and self.driver_rank is not None:
            scores_dev = self._sharded.score_from_driver(ids_dev, cu_dev, cu_host)     # the workers are in serve()
        elif self._sharded is not None:
            scores_dev = self._sharded.score_device(ids_dev, cu_dev, cu_host)
        else:
            scores_dev = self.scorer.score_device(ids_dev, cu_dev, cu_host)
        # the scores stay on the device in the requests' slots; the host copy is only for the
        # reference-visible ``aux_model_score`` attribute
        self._assign_slots(batch, being_scored=True)
        slots = torch.from_numpy(np.fromiter(map(self._get_slot, batch), np.int64, len(batch))).to(self.device)
        self.queue.set_scores(slots, scores_dev)
        return self._stager.fetch_scores(scores_dev).tolist()            # opt.py:408 .tolist()

    def _use_unfolded_twin(self) -> None:
        """Switch this ranker (and its sharded wrapper) to a handle created with ``LTR_F_NO_LN_FOLD`` - for the rest of the
        process: a checkpoint that overflowed the folded operand once will do so again."""
        if not getattr(self.scorer, "ln_fold", False):
            return
        twin = self.scorer.unfolded_twin()
        # (the folded handle may live on in its creator's hands; its default scratch and THIS ranker's prescore scratch need not)
        if self.prescore:
            self._pre_stream.synchronize()
        self.scorer.release_workspaces((self._pre_ws_key, self._pre_ws_key + "-graph"))
        self.scorer = twin
        if self._sharded is not None:
            self._sharded.scorer = twin
        if self.prescore:
            # forwards started at arrival ran on the folded handle: any of them may be the one that overflowed (the flag does
            # not say which), so none of their scores is trusted - the requests fall back to being scored in their step, on
            # the twin; the captured graphs replay the folded handle's launches and are dropped as well
            self._pre_stream.synchronize()
            for rec in self._pre_inflight:
                self._prescore_retire(rec)
            self._pre_inflight.clear()
            self._pre_static = None

    def _rescore_unfolded(self, seq_groups) -> List[float]:
        if not getattr(self.scorer, "ln_fold", False):
            raise _lib.LtrError("ltr_score: LTR_E_RANGE on a handle that does not fold its LayerNorms", _lib.LTR_E_RANGE)
        self._use_unfolded_twin()
        if self._sharded is not None:
            self._sharded.unfolded = True                   # driver mode: the next header tells the workers to switch too
        self.stats["range_fallbacks"] += 1
        out = self._score_now(seq_groups)
        self._check_status()                                # (a second failure - e.g. a bad token id - raises)
        return out

    # ---- driver / workers mode -------------------------------------------------------------------------------------
    def serve(self) -> int:
        """Worker ranks of ``MI355XRanker(group=, driver_rank=)``: score the shards the driver sends until it closes.
        Returns the number of sharded calls served."""
        if self._sharded is None or self.driver_rank is None or self._sharded.rank == self.driver_rank:
            raise RuntimeError("serve() is for the non-driver ranks of MI355XRanker(group=, driver_rank=)")
        return self._sharded.serve()

    def close(self) -> None:
        """Driver: end the workers' :meth:`serve` loops (no-op otherwise)."""
        if self._sharded is not None and self.driver_rank is not None and self._sharded.rank == self.driver_rank:
            self._sharded.stop_workers()

    def metrics(self) -> dict:
        """Counters of the ranking path: calls, requests scored, mean ms per call since construction and
        p50 / p99 over the last 1,000 calls - for ``obtain_aux_scores`` (score) and the ordering step (rank)."""
        def pct(d):
            if not d:
                return dict(n=0, p50_ms=None, p99_ms=None)
            a = np.sort(np.fromiter(d, np.float64, len(d)))
            return dict(n=len(a), p50_ms=float(a[len(a) // 2]), p99_ms=float(a[min(len(a) - 1, int(0.99 * len(a)))]))
        st = self.stats
        return dict(score=dict(calls=st["aux_calls"], requests=st["requests_scored"],
                               mean_ms_per_call=st["score_seconds"] * 1e3 / st["aux_calls"] if st["aux_calls"] else None,
                               last=pct(self._score_ms)),
                    rank=dict(calls=st["rank_calls"],
                              mean_ms_per_call=st["rank_seconds"] * 1e3 / st["rank_calls"] if st["rank_calls"] else None,
                              last=pct(self._rank_ms)),
                    live_slots=self._live_slots, queue_length=self._n_members,
                    sharded=self._sharded is not None, range_fallbacks=st["range_fallbacks"],
                    prescore=dict(enabled=self.prescore, launches=st["prescore_launches"],
                                  graph_replays=st["prescore_graph_replays"], requests=st["prescored_requests"],
                                  orphans=st["prescore_orphans"], inflight=len(self._pre_inflight),
                                  wait_ms_total=st["prescore_wait_seconds"] * 1e3,
                                  arrival_hook_ms_total=st["arrival_hook_seconds"] * 1e3))

    def _check_status(self) -> None:
        """Raise where the reference's F.embedding raises (a token id outside the vocabulary), or when the residual
        stream left the fp16 range of the LayerNorm-fold operand (LTR_E_RANGE).  With ``group=`` the flag lives on the
        rank whose shard met the condition: the ranks agree on the status CODE first (one 4-byte MAX all-reduce), so
        that EVERY rank raises - a rank that carried on alone would enter the next collective without its peers - and
        every rank names the right condition."""
        err = None
        try:
            self.scorer.check_status()
        except _lib.LtrError as e:
            err = e
        if self._sharded is not None and (self.driver_rank is None or self._sharded.last_call_collective):
            self._sharded.last_call_collective = False      # (driver mode: one agreement per call the workers took part in)
            mine = 0 if err is None else (2 if err.code == _lib.LTR_E_RANGE else 1)
            code = self._sharded.agree_status(mine)
            if code and err is None:
                if code == 2:
                    err = _lib.LtrError("ltr_score: on another rank of the group the residual stream left the fp16 range of "
                                        "the LayerNorm-fold operand; the scores of this call are invalid on every rank "
                                        "(every rank re-scores on its unfolded twin handle)", _lib.LTR_E_RANGE)
                else:
                    err = _lib.LtrError("ltr_score: another rank of the group met a token id outside the predictor's "
                                        "vocabulary (F.embedding raises on it, vocab_parallel_embedding.py:95-106); the "
                                        "scores of this call are invalid on every rank", _lib.LTR_E_INVAL)
        if err is not None:
            raise err

    # ---- Scheduler._get_*_ordered_requests ------------------------------------------------
    def order(self, reqs: Sequence, policy: Optional[str] = None, want_list: bool = True) -> Optional[list]:
        """Promote/demote (when starvation control is on) and order ``reqs`` (already the
        concatenation waiting+running+swapped, all scored); scheduler.py:984-998.  The counters are
        updated in the device-resident slots (see :meth:`sync_host`).

        ``want_list=False``: leave the permutation on the device and return None - for a caller that goes straight
        to :meth:`plan_step` ``(ordered=None, ...)``, which brings the permutation back together with the selection
        in ONE copy (no synchronisation here)."""
        n = len(reqs)
        self._aged = False           # install(): the wrapped _schedule ages the device slots if nobody calls age()
        self._last_reqs, self._last_perm_dev, self._order_ran = reqs, None, True
        if n == 0:
            self._n_members = 0
            return []
        t0 = time.perf_counter()
        policy = policy or self.st.policy
        n = self._collect_members(reqs)
        members = self._members.upload(n)
        self._members_dev, self._n_members = members, n
        q = self.queue
        self._perm.ensure(n)
        perm_dev = self._perm.dev[:n]
        if policy == "opt":
            rank_step(q._score, q._pri, q._idle, q._runs, q.starv, q.period, self._ws, out=perm_dev, members=members)
        elif policy == "xpt":                                                    # scheduler.py:910-933
            keys = np.fromiter((self._xpt_key(r) for r in reqs), np.float32, n)
            rank_step(torch.from_numpy(keys).to(self.device), None, None, None, -1, 0, self._ws, ascending=True,
                      out=perm_dev)
        else:
            tiebreak = None
            if policy in ("tpt", "rtpt"):                                        # scheduler.py:948,961
                tiebreak = torch.from_numpy(_string_rank([r.request_id for r in reqs])).to(self.device)
            rank_step(q._score, None, None, None, -1, 0, self._ws, tiebreak=tiebreak,
                      ascending=policy in ("ropt", "rtpt"), out=perm_dev, members=members)   # scheduler.py:961,1015
        self._last_perm_dev = perm_dev
        if not want_list:
            self.stats["rank_calls"] += 1
            self.stats["rank_seconds"] += time.perf_counter() - t0
            self._rank_ms.append((time.perf_counter() - t0) * 1e3)
            return None
        host = self._perm.host[:n]
        host.copy_(perm_dev, non_blocking=True)
        torch.cuda.current_stream(self.device).synchronize()
        perm = self._perm.np[:n].tolist()
        out = list(operator.itemgetter(*perm)(reqs)) if n > 1 else [reqs[perm[0]]]
        if self.mirror_host and policy == "opt" and q.starv != -1:
            self.sync_host(reqs)
        self.stats["rank_calls"] += 1
        self.stats["rank_seconds"] += time.perf_counter() - t0
        self._rank_ms.append((time.perf_counter() - t0) * 1e3)
        return out

    def _xpt_key(self, req) -> float:
        """expected_length(score) - output_len, the SRTF key of scheduler.py:920-933.  The table
        lookup is cached on the request like the reference does (``req.expected_length``)."""
        if not hasattr(req, "expected_length"):
            key, value = self.xpt_distribution
            score = round(-req.aux_model_score, 2)
            req.expected_length = -10000
            for kid in range(len(key) - 1, -1, -1):
                if score >= key[kid]:
                    req.expected_length = value[kid]
                    break
        if hasattr(req, "seqs_dict"):
            out_len = req.seqs_dict[next(iter(req.seqs_dict))].data.get_output_len()
        else:
            out_len = getattr(req, "output_len", 0)
        return float(req.expected_length - out_len)

    def ordered_requests(self, scheduler, policy: Optional[str] = None) -> list:
        """scheduler.py:969-1000 (and :936-948, :951-961, :1005-1015 for tpt/rtpt/ropt; ``policy`` "ltr" / "constraint":
        :1020-1052 - the same descending sort without starvation control, unbound in the reference's own string table;
        "constraint" also keeps ``scheduler.records``, the sorted ranking scores of everything scored so far, :1031-1033)."""
        waiting = scheduler.waiting
        # unscored requests = the arrivals since the last call: a suffix of the waiting deque
        # (add_seq_group appends, scheduler.py:376; preempted requests are pushed to the FRONT and are scored
        # already).  Walking back from the tail until the first scored request replaces the reference's scan
        # of the whole deque (:971-975) without changing the set.
        # (A scheduler that breaks this invariant fails loudly in order(): a request without a score and without
        # a slot raises, as the reference's sorted() would on -None.)
        need = []
        for r in reversed(waiting):
            if not r.need_aux_model_score():
                break
            need.append(r)
        need.reverse()
        if need:
            timed = int(os.environ.get("OPT_TIME", 0))                           # scheduler.py:977-982
            t0 = time.time()
            ret = scheduler.aux_model.obtain_aux_scores(need)
            if timed:
                print("OPT-TIME: ", time.time() - t0)
            if policy == "constraint":                                           # scheduler.py:1031-1033
                scheduler.records = sorted(list(getattr(scheduler, "records", [])) + [-x for x in ret])
        reqs = list(waiting) + list(scheduler.running) + list(scheduler.swapped)
        try:
            out = self.order(reqs, policy)
        except TypeError:
            # an unscored request that is NOT in the tail of `waiting` (a scheduler that inserts arrivals elsewhere):
            # do what the reference does - scan the whole deque (:971-975) - and order again.  Unscored requests in
            # `running` / `swapped` still fail, like the reference's sorted() on -None.
            need = [r for r in waiting if r.need_aux_model_score()]
            if not need:
                raise
            scheduler.aux_model.obtain_aux_scores(need)
            out = self.order(reqs, policy)
        self._gc_slots(len(reqs))        # `reqs` is every live request here (not so for direct order() calls on a subset)
        return out

    # ---- aging loop of _general_schedule ----------------------------------------------------
    def age(self, all_pri: Sequence, running_this_step: Iterable) -> None:
        """scheduler.py:1358-1365, computed by ltr_age_update on the device-resident slots: only the slots of
        ``running_this_step`` (<= max_num_seqs) are uploaded; the members of the step are the ones
        :meth:`order` ranked (``all_pri`` is the same set, scheduler.py:1337-1338)."""
        self._aged = True
        n = len(all_pri)
        if n == 0:
            return
        if n != self._n_members or self._members_dev is None:
            n = self._collect_members(all_pri)
            self._members_dev, self._n_members = self._members.upload(n), n
        ran = running_this_step if isinstance(running_this_step, (list, tuple)) else list(running_this_step)
        k = len(ran)
        self._ran.ensure(max(k, 1))
        if k:
            a = np.fromiter(map(self._get_slot, ran), np.int32, k)
            a.sort()
            self._ran.np[:k] = a
        q = self.queue
        q.age(members=self._members_dev, ran_slots=self._ran.upload(k))
        if self.mirror_host:
            self.sync_host(all_pri)
        if self.prescore:
            self._prescore_pump(force=True)                # whatever was held back: the next step is a backbone step away

    def sync_host(self, reqs: Sequence) -> None:
        """Refresh ``pri / idle / runs`` of the request objects from their device slots."""
        n = len(reqs)
        if n == 0:
            return
        idx = torch.from_numpy(np.fromiter(map(self._get_slot, reqs), np.int64, n)).to(self.device)
        q = self.queue
        st = torch.stack([q._pri[idx], q._idle[idx], q._runs[idx]]).cpu().numpy()
        for i, r in enumerate(reqs):
            r.pri = int(st[0, i]); r.idle = int(st[1, i]); r.runs = int(st[2, i])

    # ---- front half of _general_schedule: budget walk + eviction choice -----------------------
    def plan_step(self, ordered: Sequence, new_tokens: Sequence[int], new_seqs: Sequence[int], token_budget: int,
                  max_num_seqs: int, blocks: Optional[dict] = None, chunkable: Optional[Sequence[int]] = None) -> dict:
        """What ``_general_schedule`` decides between the sort and the block-table updates
        (scheduler.py:1137-1218): the prefix of ``ordered`` the budget walk selects with the tokens
        granted to each request, and - when ``blocks`` describes the KV-block state - the requests
        ``reserve_free_blocks`` (:1376-1452) evicts.  Both run on the device (``ltr_budget_prefix``,
        ``ltr_reserve_select``); one D2H copy brings the decisions back.

        ordered       the ranked list (``ordered_requests``)
        new_tokens    per element: un-chunked ``_get_num_new_tokens`` (:1878-1881)
        new_seqs      per element: ``get_max_num_running_seqs()``
        chunkable     per element: 1 iff the group has ONE sequence in the walked status (``len(seqs) == 1``,
                      :1884; a WAITING prompt with best_of > 1 is chunkable).  None: ``new_seqs <= 1``.
        blocks        None, or dict(state=, phys=, logical=, nrun=, nswap=, free=, watermark=) with
                      per-element sequences and the block manager's two scalars

        Returns dict(selected=[...], granted=[...], swap_out=[...], put_back=[...], execute=[...]):
        ``swap_out`` = unselected running requests to swap out, lowest priority first;
        ``put_back`` = selected requests dropped from the selection, last selected first;
        ``execute``  = ``execute_pinned_requests``."""
        dev = self.device
        i32 = lambda a: torch.from_numpy(np.asarray(a, np.int32)).to(dev)
        if ordered is None:
            # straight from order(..., want_list=False): the permutation is still on the device and every per-element
            # sequence is indexed by CONCATENATION position (waiting+running+swapped, the list order() was given);
            # the ranked list comes back with the decisions in the one copy below
            reqs, perm = self._last_reqs, self._last_perm_dev
            n = len(reqs)
            if perm is None and (n or not self._order_ran):    # (an idle step - order([]) - has no permutation and needs none)
                raise ValueError("plan_step(ordered=None) needs a preceding order(..., want_list=False)")
        else:
            n = len(ordered)
            perm = torch.arange(n, dtype=torch.int32, device=dev) if n else None   # `ordered` is already in rank order
        if n == 0:
            return dict(selected=[], granted=[], swap_out=[], put_back=[], execute=[], ordered=[])
        nt, nq = i32(new_tokens), i32(new_seqs)
        ck = None if chunkable is None else torch.from_numpy(np.asarray(chunkable, np.uint8)).to(dev)
        n_sel, _, granted = budget_prefix(perm, nt, nq, token_budget, max_num_seqs, want_ran=False, chunkable=ck)
        parts = [n_sel, granted]
        if blocks is not None:
            state = torch.from_numpy(np.asarray(blocks["state"], np.uint8)).to(dev)
            action, n_exec, _ = reserve_select(perm, n_sel, state, i32(blocks["phys"]), i32(blocks["logical"]),
                                               i32(blocks["nrun"]), i32(blocks["nswap"]),
                                               int(blocks["free"]) - int(blocks["watermark"]), new_seqs=nq)
            parts += [n_exec, action.to(torch.int32)]
        if ordered is None:
            parts.append(perm)
        host = torch.cat(parts).cpu().numpy()                       # the one D2H copy of the step
        k = int(host[0])
        if ordered is None:
            p = host[-n:]
            ordered = list(operator.itemgetter(*p.tolist())(reqs)) if n > 1 else [reqs[int(p[0])]]
            by_rank = lambda a: a[p]                                # per-request outputs are indexed by position
        else:
            by_rank = lambda a: a
        g = by_rank(host[1:1 + n])
        if blocks is None:
            sel = list(ordered[:k])
            return dict(selected=sel, granted=g[:k].tolist(), swap_out=[], put_back=[], execute=sel, ordered=ordered)
        ke = int(host[1 + n])
        act = by_rank(host[2 + n:2 + 2 * n])
        sel = list(ordered[:k])
        swap_out = [ordered[i] for i in range(n - 1, k - 1, -1) if act[i] == 1]
        put_back = [ordered[i] for i in range(k - 1, -1, -1) if act[i] in (2, 3)]
        return dict(selected=sel, granted=g[:k].tolist(), swap_out=swap_out, put_back=put_back,
                    execute=sel[:ke], ordered=ordered)

    # ---- wiring ------------------------------------------------------------------------------
    def install(self, scheduler) -> None:
        """Put this ranker where llm_engine.py:228-242 puts the AUXLLM and where
        scheduler.py:319-329 binds the schedule / ordering functions of a score-ordered policy."""
        policy = self.st.policy
        if not self.st.need_score:
            raise ValueError(f"schedule type {self.st.raw!r} orders without a predictor score (scheduler.py:292-311): "
                             "nothing for the ranker to do - keep the scheduler's own ordering function")
        if policy == "xpt" and self.xpt_distribution is None:
            self.xpt_distribution = getattr(scheduler, "distribution", None)     # what scheduler.py:312 loaded
            if self.xpt_distribution is None:
                raise ValueError("schedule type xpt needs its score -> length table: name it in the string "
                                 "(xpt{path}..., scheduler.py:312) or pass xpt_distribution=(key, value)")
        scheduler.aux_model = self
        scheduler.need_score = True
        scheduler.starv = self.st.starv
        if self.st.starv != -1:
            scheduler.period = self.st.period
        if hasattr(scheduler, "_general_schedule"):
            # scheduler.py:313,320,326 bind _schedule = _general_schedule.  The starvation counters live in the device
            # slots, so the aging of :1358-1365 must reach them: a scheduler patched with INTEGRATION.md hunk (b)
            # calls self.age(all_pri, running_this_step) itself; on an UNPATCHED reference scheduler (its loop ages the
            # host attributes only) the wrapper below does it from the step's outputs - install() alone is enough.
            inner = scheduler._general_schedule

            def _schedule_and_age():
                ret = inner()
                if not self._aged:
                    all_pri = list(scheduler.swapped) + list(scheduler.running) + list(scheduler.waiting)   # :1337
                    self.age(all_pri, [r.seq_group for r in getattr(ret, "scheduled_seq_groups", ())])
                return ret
            _schedule_and_age.__wrapped__ = inner
            scheduler._schedule = _schedule_and_age
        # _general_schedule starts with self._update_priority() (:1103); every score-ordered policy binds a no-op
        # there (:935,950,965,1002,1017) - a scheduler built as `fcfs` and given a ranker afterwards has none
        scheduler._update_priority = lambda: None
        scheduler._get_ordered_requests = lambda: self.ordered_requests(scheduler, policy)
