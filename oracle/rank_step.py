"""ORACLE (test infrastructure, not product code) - CPU restatement of the
reference scheduler's per-step ranking: starvation promote/demote, the stable
priority sort, and the post-schedule aging of the ``idle/runs`` counters.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module.

Reference (relative to /root/reference):
* schedule-type grammar ``...starv<S>-period<P>``: vllm/core/scheduler.py:269-275
* per-request state ``idle = runs = pri = 0`` on arrival: scheduler.py:372-374
* promote / demote + ``sorted(..., key=(pri, -score))`` / ``key=-score``:
  scheduler.py:984-998 (``_get_opt_ordered_requests``)
* other orderings that share the kernel: ``tpt`` ``(-score, request_id)`` :948,
  ``rtpt`` ``(score, request_id)`` :961, ``ropt`` ``score`` :1015
* aging after the budget walk: scheduler.py:1337,1358-1365

Pinning: ``oracle/make_golden.py`` drives the *reference's own*
``Scheduler._get_opt_ordered_requests`` / ``Scheduler.schedule`` (imported from
/root/reference in the build container) and stores orders + counters in
``tests/golden/rank_*.npz``; ``tests/test_oracle_golden.py`` replays them here.

Two forms are kept: the literal per-object Python (what the reference executes,
used as the CPU baseline) and a NumPy form for 64k+ queues; tests check they
agree.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np


class Req:
    """The fields of ``SequenceGroup`` the ranking touches
    (sequence.py:426-433, scheduler.py:372-374)."""

    def __init__(self, request_id: str, score: Optional[float] = None):
        self.request_id = request_id
        self.aux_model_score = score
        self.pri = 0
        self.idle = 0
        self.runs = 0

    def need_aux_model_score(self) -> bool:          # sequence.py:461-462
        return self.aux_model_score is None

    def set_aux_model_score(self, s) -> None:        # sequence.py:464-465
        self.aux_model_score = s


def parse_starvation(schedule_type: str) -> Tuple[int, int]:
    """scheduler.py:269-275 (same slicing arithmetic). Returns (starv, period);
    starv == -1 means the feature is off."""
    starv, period = -1, 0
    if "starv" in schedule_type:
        starv = int(schedule_type[schedule_type.find("starv") + len("starv"):
                                  schedule_type.find("period") - 1])
        period = int(schedule_type[schedule_type.find("period") + len("period"):])
    return starv, period


# ---- literal per-object form --------------------------------------------------
def opt_order(reqs: Sequence[Req], starv: int, period: int) -> List[Req]:
    """scheduler.py:984-998 over ``list(waiting)+list(running)+list(swapped)``."""
    if starv != -1:
        for r in reqs:
            if r.idle >= starv:
                r.pri = -1
                r.idle = 0
                r.runs = period
            elif r.pri == -1 and r.runs <= 0:
                r.pri = 0
        return list(sorted(reqs, key=lambda req: (req.pri, -req.aux_model_score)))
    return list(sorted(reqs, key=lambda req: -req.aux_model_score))


def tpt_order(reqs: Sequence[Req]) -> List[Req]:      # scheduler.py:948
    return list(sorted(reqs, key=lambda req: (-req.aux_model_score, req.request_id)))


def rtpt_order(reqs: Sequence[Req]) -> List[Req]:     # scheduler.py:961
    return list(sorted(reqs, key=lambda req: (req.aux_model_score, req.request_id)))


def ropt_order(reqs: Sequence[Req]) -> List[Req]:     # scheduler.py:1015
    return list(sorted(reqs, key=lambda req: req.aux_model_score))


def xpt_expected_length(aux_model_score: float, key: Sequence[float], value: Sequence[float]) -> float:
    """scheduler.py:920-931: expected output length looked up from the score->length table
    (``self.distribution = (key, value)``, loaded at :312)."""
    score = round(-aux_model_score, 2)
    expected_length = -10000
    for kid in range(len(key) - 1, -1, -1):
        if score >= key[kid]:
            expected_length = value[kid]
            break
    return expected_length


def xpt_order(reqs, key, value, output_len) -> list:
    """scheduler.py:910-933 (``_get_xpt_ordered_requests``): SRTF on the table's expected length;
    ``output_len(req)`` = tokens generated so far (``seq.data.get_output_len()``)."""
    for req in reqs:
        if not hasattr(req, "expected_length"):
            req.expected_length = xpt_expected_length(req.aux_model_score, key, value)
    return list(sorted(reqs, key=lambda req: req.expected_length - output_len(req)))


def age_update(all_pri: Sequence[Req], running_this_step: Sequence[Req]) -> None:
    """scheduler.py:1358-1365 (membership is by object identity/equality there;
    a set of ids gives the same answer in O(N))."""
    ran = {id(r) for r in running_this_step}
    for seq in all_pri:
        if id(seq) in ran:
            if seq.pri == -1:
                seq.runs -= 1
            seq.idle = 0
        else:
            seq.idle += 1


# ---- NumPy array form ---------------------------------------------------------
def promote_demote_np(pri, idle, runs, starv: int, period: int) -> None:
    """In-place scheduler.py:986-993 on int32 arrays."""
    if starv == -1:
        return
    promote = idle >= starv
    demote = (~promote) & (pri == -1) & (runs <= 0)
    pri[promote] = -1
    idle[promote] = 0
    runs[promote] = period
    pri[demote] = 0


def order_np(score, pri, tiebreak=None, use_pri: bool = True, ascending: bool = False):
    """Permutation ``perm`` with ``perm[k]`` = input index of the k-th request.
    Keys, most significant first: ``pri`` (if used), ``-score`` (or ``score``),
    ``tiebreak`` (default: input index = Python's stable sort).  ``-0.0 == 0.0``
    compare equal exactly as Python floats do."""
    score = np.asarray(score, np.float32)
    n = score.shape[0]
    k = score.astype(np.float64)
    k = k if ascending else -k
    k = k + 0.0                                     # -0.0 + 0.0 -> +0.0
    tb = np.arange(n, dtype=np.int64) if tiebreak is None else np.asarray(tiebreak, np.int64)
    keys = (tb, k) + ((np.asarray(pri, np.int64),) if use_pri else ())
    return np.lexsort(keys).astype(np.int32)


def rank_step_np(score, pri, idle, runs, starv: int, period: int, tiebreak=None):
    """promote/demote (in place) followed by the (pri, -score) order -
    the array form of :func:`opt_order`."""
    promote_demote_np(pri, idle, runs, starv, period)
    return order_np(score, pri, tiebreak, use_pri=(starv != -1))


def age_update_np(ran, pri, idle, runs) -> None:
    """In-place scheduler.py:1358-1365 on arrays; ``ran`` is a bool/uint8 mask."""
    ran = np.asarray(ran).astype(bool)
    runs[ran & (pri == -1)] -= 1
    idle[ran] = 0
    idle[~ran] += 1


def string_rank(request_ids: Sequence[str]) -> np.ndarray:
    """Tiebreak key for the ``tpt`` orderings: rank of ``request_id`` under
    Python *string* comparison (``"10" < "9"``), scheduler.py:948."""
    order = sorted(range(len(request_ids)), key=lambda i: request_ids[i])
    rank = np.empty(len(request_ids), np.int64)
    rank[order] = np.arange(len(request_ids))
    return rank


# ---- budget walk (next row in scope) --------------------------------
def budget_walk(order_need_tokens, order_need_seqs, token_budget: int, max_num_seqs: int, order_chunkable=None):
    """Literal restatement of the selection loop of Scheduler._general_schedule
    (scheduler.py:1137-1211) for one ranked order: per request, in order,
    ``num_new_tokens = min(need, budget.remaining_token_budget())`` when the group has a
    single sequence IN THE WALKED STATUS (``len(seqs) == 1`` in _get_num_new_tokens, :1867-1888,
    enable_chunking=True at :1128) - ``order_chunkable``; a WAITING prompt with best_of > 1 has one
    sequence but ``new_seqs = best_of`` (sequence.py:500-504), so the flag is NOT ``new_seqs == 1``
    (None keeps that approximation for callers that only have single-sequence groups);
    ``break`` if ``num_new_tokens == 0 or not budget.can_schedule(...)`` (:51-55).
    Returns (n_selected, granted tokens per position)."""
    used_tokens = 0
    used_seqs = 0
    granted = []
    if order_chunkable is None:
        order_chunkable = [nseq <= 1 for nseq in order_need_seqs]
    for need, nseq, chunk in zip(order_need_tokens, order_need_seqs, order_chunkable):
        n = int(need)
        if chunk:
            n = min(n, token_budget - used_tokens)
        if n == 0 or not (used_tokens + n <= token_budget and used_seqs + nseq <= max_num_seqs):
            break
        used_tokens += n
        used_seqs += int(nseq)
        granted.append(n)
    return len(granted), granted


# ---------------------------------------------------------------------------------------------
# Victim selection of Scheduler.reserve_free_blocks (scheduler.py:1376-1452), next row 8f-1.
# ---------------------------------------------------------------------------------------------
def reserve_select(perm, n_selected, state, phys, logical, nrun, nswap, need):
    """Which requests reserve_free_blocks evicts, restated on arrays.

    perm        request indices in rank order; perm[:n_selected] is the budget walk's selection
                (``pinned_requests``), the rest ``priority_requests``
    state       per request: 0 waiting, 1 has RUNNING seqs, 2 has SWAPPED seqs
    phys/logical/nrun/nswap   len(_get_physical_blocks), len(logical_token_blocks),
                num_seqs(RUNNING), num_seqs(SWAPPED) per request
    need        num_blocks_needed - free GPU blocks + watermark (:1384-1388)

    Returns (action uint8 per request, n_exec): action 1 = unselected running request swapped out
    (:1400-1420, walked from the lowest priority), 2 = selected running request put back and
    preempted, 3 = selected swapped / waiting request put back (:1422-1447); n_exec = selected
    requests that still execute.
    """
    n = len(state)
    action = np.zeros(n, np.uint8)
    n_exec = int(n_selected)
    if need <= 0:
        return action, n_exec
    for i in reversed(list(perm[n_selected:])):          # :1400 reversed(priority_requests)
        if need <= 0:
            break
        if state[i] == 1:                                # has RUNNING seqs
            need -= int(phys[i])
            action[i] = 1
    if need > 0:                                         # :1422
        sel = list(perm[:n_selected])
        while need > 0 and sel:
            i = sel.pop(-1)
            if state[i] == 1:
                need -= int(nrun[i]) + int(phys[i])
                action[i] = 2
            elif state[i] == 2:
                need -= int(phys[i]) + int(nswap[i])
                action[i] = 3
            else:
                need -= int(logical[i])
                action[i] = 3
        n_exec = len(sel)
    return action, n_exec


def reserve_select_np(perm, n_selected, state, phys, logical, nrun, nswap, need):
    """The same as one reversed exclusive prefix sum (what the HIP kernel computes)."""
    perm = np.asarray(perm, np.int64)
    n = len(state)
    action = np.zeros(n, np.uint8)
    if need <= 0 or len(perm) == 0:
        return action, int(n_selected)
    pos = np.arange(len(perm))
    st = np.asarray(state)[perm]
    sel = pos < n_selected
    w_unsel = np.where(st == 1, np.asarray(phys)[perm], 0)
    w_sel = np.where(st == 1, np.asarray(nrun)[perm] + np.asarray(phys)[perm],
                     np.where(st == 2, np.asarray(phys)[perm] + np.asarray(nswap)[perm], np.asarray(logical)[perm]))
    w = np.where(sel, w_sel, w_unsel).astype(np.int64)
    rev = w[::-1]
    excl = (np.cumsum(rev) - rev)[::-1]                   # blocks freed by everything behind this position
    hit = excl < need
    a = np.where(sel, np.where(st == 1, 2, 3), np.where(st == 1, 1, 0))
    a = np.where(hit, a, 0)
    action[perm] = a.astype(np.uint8)
    return action, int(n_selected - int((hit & sel).sum()))
