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
  ``tpt/rtpt/ropt`` variants) with the promote/demote + sort done by ``ltr_rank_step``;
* ``age(all_pri, running_this_step)`` - the aging loop done by ``ltr_age_update``;
* ``install(scheduler)`` - assigns ``scheduler.aux_model`` and rebinds
  ``scheduler._get_ordered_requests`` exactly where ``scheduler.py:325-329`` binds them.

Requests are duck-typed on the fields the reference touches: ``request_id``,
``aux_model_score`` / ``need_aux_model_score()`` / ``set_aux_model_score()``
(vllm/sequence.py:429,461-465), ``pri / idle / runs`` (scheduler.py:372-374), and the
prompt (``prompt`` text or ``prompt_token_ids``).
"""
from __future__ import annotations

import os
import time
from typing import Callable, Iterable, List, Optional, Sequence

import numpy as np
import torch

from . import _lib
from .config_predictor import PrefillPredictorConfig
from .host_pipeline import InputStager, cached_token_ids
from .opt_spec import OPTSpec, load_hf_checkpoint
from .rank import RankWorkspace, age_update, budget_prefix, rank_step, reserve_select
from .schedule_type import ScheduleType, parse_schedule_type
from .scorer import HipOPTScorer


def _string_rank(request_ids: Sequence[str]) -> np.ndarray:
    """Rank of each request_id under Python string comparison (the ``tpt`` tiebreak,
    scheduler.py:948: ``key=(-score, req.request_id)``)."""
    order = sorted(range(len(request_ids)), key=request_ids.__getitem__)
    rank = np.empty(len(request_ids), np.int32)
    rank[order] = np.arange(len(request_ids), dtype=np.int32)
    return rank


class MI355XRanker:
    def __init__(self, scorer: HipOPTScorer, schedule_type: str = "opt", max_length: int = 2048,
                 tokenize: Optional[Callable[[str], List[int]]] = None, mtype: str = "rank",
                 xpt_distribution=None):
        """
        scorer      the HBM-resident predictor
        schedule_type  the reference's schedule string (``opt-...-starv<S>-period<P>``)
        max_length  prompt truncation, ``PrefillModelConfig.max_length`` = the AUX engine's
                    ``max_model_len`` (aux_llm_engine.py:365-369, llm_engine.py:236)
        tokenize    text -> predictor token ids.  The reference re-tokenises the prompt
                    TEXT with the predictor's (OPT) tokenizer (aux_llm_engine.py:341,365-369);
                    pass that tokenizer's ``encode`` here.  None: use the request's
                    ``prompt_token_ids`` (valid when backbone and predictor share a
                    tokenizer, and for synthetic workloads).
        """
        self.scorer = scorer
        self.device = scorer.device
        self.st: ScheduleType = parse_schedule_type(schedule_type)
        self.max_length = int(max_length)
        self.tokenize = tokenize
        self.mtype = mtype
        # xpt policy: (key, value) score -> expected-length table, scheduler.py:312 (torch.load(...))
        self.xpt_distribution = xpt_distribution
        if mtype == "rank" and scorer.spec.num_labels != 1:
            raise ValueError("mtype 'rank' needs num_labels == 1 (prefill_predictor.py:35-36)")
        self._ws = RankWorkspace(self.device)
        self._stager = InputStager(self.device)
        self.stats = dict(aux_calls=0, requests_scored=0, rank_calls=0, score_seconds=0.0, rank_seconds=0.0)

    # ---- construction from the reference's config objects ------------------------------
    @classmethod
    def from_predictor_config(cls, cfg, schedule_type: str, device: str = "cuda:0", tokenize=None,
                              weight_dtype: str = "f16") -> "MI355XRanker":
        """``cfg``: a :class:`PrefillPredictorConfig` or a path to its JSON
        (``--prefill-predictor-model-config``, arg_utils.py:346-359).  Loads the HF
        checkpoint at ``cfg.model.path`` like llm_engine.py:228-240 does for the AUXLLM."""
        if isinstance(cfg, (str, os.PathLike)):
            cfg = PrefillPredictorConfig.from_json(cfg)
        spec, ckpt = load_hf_checkpoint(cfg.model.path)
        scorer = HipOPTScorer(spec, ckpt, device=device, weight_dtype=weight_dtype)
        return cls(scorer, schedule_type, max_length=cfg.model.max_length, tokenize=tokenize,
                   mtype=cfg.model.mtype)

    # ---- AUXLLM.obtain_aux_scores -------------------------------------------------------
    def add_request(self, sg) -> None:
        """Optional arrival-time hook (where ``Scheduler.add_seq_group`` runs, scheduler.py:368-376):
        tokenise / truncate the prompt once so the scoring call only packs cached arrays."""
        cached_token_ids(sg, self.tokenize, self.max_length)

    def obtain_aux_scores(self, seq_groups) -> List[float]:
        seq_groups = list(seq_groups)
        if not seq_groups:
            return []
        t0 = time.perf_counter()
        for sg in seq_groups:
            assert sg.need_aux_model_score()               # aux_llm_engine.py:409
        arrays = [cached_token_ids(sg, self.tokenize, self.max_length) for sg in seq_groups]
        ids_dev, cu_dev, cu_host = self._stager.stage(arrays)          # pinned pack + async H2D
        scores = self._stager.fetch_scores(self.scorer.score_device(ids_dev, cu_dev, cu_host))
        out = scores.tolist()                              # opt.py:408 .tolist()
        for sg, s in zip(seq_groups, out):
            sg.set_aux_model_score(s)                      # aux_llm_engine.py:408-410
        self.stats["aux_calls"] += 1
        self.stats["requests_scored"] += len(seq_groups)
        self.stats["score_seconds"] += time.perf_counter() - t0
        return out

    # ---- Scheduler._get_*_ordered_requests ------------------------------------------------
    def order(self, reqs: Sequence, policy: Optional[str] = None) -> list:
        """Promote/demote (when starvation control is on) and order ``reqs`` (already the
        concatenation waiting+running+swapped, all scored).  Mutates ``pri/idle/runs`` on the
        request objects like scheduler.py:986-993 and returns the new list."""
        n = len(reqs)
        if n == 0:
            return []
        t0 = time.perf_counter()
        policy = policy or self.st.policy
        starv, period = (self.st.starv, self.st.period) if policy == "opt" else (-1, 0)
        if policy == "xpt":                                                      # scheduler.py:910-933
            keys = np.fromiter((self._xpt_key(r) for r in reqs), np.float32, n)
            score = torch.from_numpy(keys).to(self.device)
        else:
            score = torch.from_numpy(np.fromiter((r.aux_model_score for r in reqs), np.float32, n)).to(self.device)
        pri = idle = runs = None
        if starv != -1:
            st = np.empty((3, n), np.int32)
            for i, r in enumerate(reqs):
                st[0, i] = r.pri; st[1, i] = r.idle; st[2, i] = r.runs
            dev = torch.from_numpy(st).to(self.device)
            pri, idle, runs = dev[0], dev[1], dev[2]
        tiebreak = None
        ascending = policy in ("ropt", "rtpt", "xpt")                            # scheduler.py:933,961,1015
        if policy in ("tpt", "rtpt"):                                            # scheduler.py:948,961
            tiebreak = torch.from_numpy(_string_rank([r.request_id for r in reqs])).to(self.device)
        perm = rank_step(score, pri, idle, runs, starv, period, self._ws, tiebreak=tiebreak, ascending=ascending)
        perm_h = perm.cpu().numpy()
        if starv != -1:
            st = dev.cpu().numpy()
            for i, r in enumerate(reqs):
                r.pri = int(st[0, i]); r.idle = int(st[1, i]); r.runs = int(st[2, i])
        self.stats["rank_calls"] += 1
        self.stats["rank_seconds"] += time.perf_counter() - t0
        return [reqs[i] for i in perm_h]

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
        """scheduler.py:969-1000 (and :936-948, :951-961, :1005-1015 for tpt/rtpt/ropt)."""
        need = [r for r in scheduler.waiting if r.need_aux_model_score()]
        if need:
            timed = int(os.environ.get("OPT_TIME", 0))                           # scheduler.py:977-982
            t0 = time.time()
            scheduler.aux_model.obtain_aux_scores(need)
            if timed:
                print("OPT-TIME: ", time.time() - t0)
        reqs = list(scheduler.waiting) + list(scheduler.running) + list(scheduler.swapped)
        return self.order(reqs, policy)

    # ---- aging loop of _general_schedule ----------------------------------------------------
    def age(self, all_pri: Sequence, running_this_step: Iterable) -> None:
        """scheduler.py:1358-1365 on the request objects, computed by ltr_age_update."""
        n = len(all_pri)
        if n == 0:
            return
        ran_ids = {id(r) for r in running_this_step}
        st = np.empty((3, n), np.int32)
        ran = np.zeros(n, np.uint8)
        for i, r in enumerate(all_pri):
            st[0, i] = r.pri; st[1, i] = r.idle; st[2, i] = r.runs
            ran[i] = id(r) in ran_ids
        dev = torch.from_numpy(st).to(self.device)
        age_update(torch.from_numpy(ran).to(self.device), dev[0], dev[1], dev[2])
        st = dev.cpu().numpy()
        for i, r in enumerate(all_pri):
            r.pri = int(st[0, i]); r.idle = int(st[1, i]); r.runs = int(st[2, i])

    # ---- front half of _general_schedule: budget walk + eviction choice -----------------------
    def plan_step(self, ordered: Sequence, new_tokens: Sequence[int], new_seqs: Sequence[int], token_budget: int,
                  max_num_seqs: int, blocks: Optional[dict] = None) -> dict:
        """What ``_general_schedule`` decides between the sort and the block-table updates
        (scheduler.py:1137-1218): the prefix of ``ordered`` the budget walk selects with the tokens
        granted to each request, and - when ``blocks`` describes the KV-block state - the requests
        ``reserve_free_blocks`` (:1376-1452) evicts.  Both run on the device (``ltr_budget_prefix``,
        ``ltr_reserve_select``); one D2H copy brings the decisions back.

        ordered       the ranked list (``ordered_requests``)
        new_tokens    per element: un-chunked ``_get_num_new_tokens`` (:1878-1881)
        new_seqs      per element: ``get_max_num_running_seqs()``
        blocks        None, or dict(state=, phys=, logical=, nrun=, nswap=, free=, watermark=) with
                      per-element sequences and the block manager's two scalars

        Returns dict(selected=[...], granted=[...], swap_out=[...], put_back=[...], execute=[...]):
        ``swap_out`` = unselected running requests to swap out, lowest priority first;
        ``put_back`` = selected requests dropped from the selection, last selected first;
        ``execute``  = ``execute_pinned_requests``."""
        n = len(ordered)
        if n == 0:
            return dict(selected=[], granted=[], swap_out=[], put_back=[], execute=[])
        dev = self.device
        i32 = lambda a: torch.from_numpy(np.asarray(a, np.int32)).to(dev)
        perm = torch.arange(n, dtype=torch.int32, device=dev)       # `ordered` is already in rank order
        nt, nq = i32(new_tokens), i32(new_seqs)
        n_sel, _, granted = budget_prefix(perm, nt, nq, token_budget, max_num_seqs, want_ran=False)
        if blocks is None:
            k = int(n_sel.item())
            g = granted[:k].cpu().numpy()
            sel = list(ordered[:k])
            return dict(selected=sel, granted=g.tolist(), swap_out=[], put_back=[], execute=sel)
        state = torch.from_numpy(np.asarray(blocks["state"], np.uint8)).to(dev)
        action, n_exec, _ = reserve_select(perm, n_sel, state, i32(blocks["phys"]), i32(blocks["logical"]),
                                           i32(blocks["nrun"]), i32(blocks["nswap"]),
                                           int(blocks["free"]) - int(blocks["watermark"]), new_seqs=nq)
        host = torch.cat([n_sel, n_exec, granted, action.to(torch.int32)]).cpu().numpy()
        k, ke = int(host[0]), int(host[1])
        g, act = host[2:2 + n], host[2 + n:]
        sel = list(ordered[:k])
        swap_out = [ordered[i] for i in range(n - 1, k - 1, -1) if act[i] == 1]
        put_back = [ordered[i] for i in range(k - 1, -1, -1) if act[i] in (2, 3)]
        return dict(selected=sel, granted=g[:k].tolist(), swap_out=swap_out, put_back=put_back,
                    execute=sel[:ke])

    # ---- wiring ------------------------------------------------------------------------------
    def install(self, scheduler) -> None:
        """Put this ranker where llm_engine.py:228-242 puts the AUXLLM and where
        scheduler.py:325-329 binds the ordering function."""
        scheduler.aux_model = self
        scheduler.need_score = True
        scheduler.starv = self.st.starv
        if self.st.starv != -1:
            scheduler.period = self.st.period
        policy = self.st.policy if self.st.policy in ("opt", "tpt", "xpt") else "opt"
        if policy == "xpt" and self.xpt_distribution is None:
            raise ValueError("schedule type xpt needs xpt_distribution=(key, value) (scheduler.py:312)")
        scheduler._get_ordered_requests = lambda: self.ordered_requests(scheduler, policy)
