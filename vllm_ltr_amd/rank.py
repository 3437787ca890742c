"""Array-level binding of the rank kernels (ltr_rank_step / ltr_age_update / ltr_queue_step /
ltr_budget_prefix / ltr_reserve_select): starvation promote/demote + stable priority sort +
budget-walk selection + aging on device-resident ``score / pri / idle / runs`` slot arrays.

Reference semantics: vllm/core/scheduler.py:984-998 (order), :1358-1365 (aging),
:1137-1211 (budget walk prefix), :1376-1452 (eviction choice).

Every call runs with the device of its tensors made current (the handle-less C entry points
launch on the current device) and on torch's current stream of that device.
"""
from __future__ import annotations

from typing import Optional, Sequence

import numpy as np
import torch

from . import _lib


def _stream(device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def _p(t: Optional[torch.Tensor]):
    return t.data_ptr() if t is not None else None


class RankWorkspace:
    """Reusable scratch for ltr_rank_step, grown on demand (only touched above 12,288 requests)."""

    def __init__(self, device):
        self.device = torch.device(device)
        self.lib = _lib.load()
        self._buf: Optional[torch.Tensor] = None

    def get(self, N: int) -> torch.Tensor:
        need = int(self.lib.ltr_workspace_bytes(None, _lib.LTR_WS_RANK, N, 0))
        if self._buf is None or self._buf.numel() < need:
            self._buf = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._buf


def _flags(starv: int, ascending: bool, use_pri: Optional[bool]) -> int:
    flags = _lib.LTR_RANK_ASCENDING if ascending else 0
    if use_pri or (use_pri is None and starv != -1):
        flags |= _lib.LTR_RANK_USE_PRI
    return flags


def rank_step(scores: torch.Tensor, pri: Optional[torch.Tensor], idle: Optional[torch.Tensor],
              runs: Optional[torch.Tensor], starv: int, period: int, ws: RankWorkspace,
              tiebreak: Optional[torch.Tensor] = None, ascending: bool = False, use_pri: Optional[bool] = None,
              out: Optional[torch.Tensor] = None, members: Optional[torch.Tensor] = None) -> torch.Tensor:
    """In-place promote/demote on (pri, idle, runs) and the ranked permutation (int32 [N], ``perm[k]`` =
    position of the k-th request).  ``members`` int32 [N]: slot of each queued request in the order
    list(waiting)+list(running)+list(swapped); None: the arrays are that concatenation.  Asynchronous."""
    assert scores.dtype == torch.float32 and scores.is_cuda
    N = members.numel() if members is not None else scores.numel()
    for t in (pri, idle, runs):
        assert t is None or (t.dtype == torch.int32 and t.numel() == scores.numel() and t.is_cuda)
    if members is not None:
        assert members.dtype == torch.int32 and members.is_cuda
    if tiebreak is not None:
        assert tiebreak.dtype in (torch.int32, torch.uint32) and tiebreak.numel() == N
    if out is None:
        out = torch.empty(N, dtype=torch.int32, device=scores.device)
    if N == 0:
        return out
    buf = ws.get(N)
    with torch.cuda.device(scores.device):
        _lib.check(ws.lib.ltr_rank_step(scores.data_ptr(), _p(pri), _p(idle), _p(runs), _p(tiebreak), _p(members), N,
                                        int(starv), int(period), _flags(starv, ascending, use_pri), out.data_ptr(),
                                        buf.data_ptr(), buf.numel(), _stream(scores.device)), "ltr_rank_step")
    return out


def age_update(ran: Optional[torch.Tensor], pri: torch.Tensor, idle: torch.Tensor, runs: torch.Tensor,
               members: Optional[torch.Tensor] = None, ran_slots: Optional[torch.Tensor] = None) -> None:
    """In-place scheduler.py:1358-1365 over the queued requests.  ``ran`` uint8 [N] by position, or
    ``ran_slots`` int32 ascending = slots of ``running_this_step``."""
    N = members.numel() if members is not None else pri.numel()
    if N == 0:
        return
    if ran is not None:
        assert ran.dtype == torch.uint8 and ran.numel() == N
    else:
        assert ran_slots is not None and ran_slots.dtype == torch.int32
    lib = _lib.load()
    with torch.cuda.device(pri.device):
        _lib.check(lib.ltr_age_update(_p(ran), _p(ran_slots), 0 if ran_slots is None else ran_slots.numel(),
                                      pri.data_ptr(), idle.data_ptr(), runs.data_ptr(), _p(members), N,
                                      _stream(pri.device)), "ltr_age_update")


def budget_prefix(perm: torch.Tensor, new_tokens: torch.Tensor, new_seqs: torch.Tensor, token_budget: int,
                  max_num_seqs: int, want_ran: bool = True, want_granted: bool = True,
                  chunkable: Optional[torch.Tensor] = None):
    """Selection of the budget walk over the ranked order (scheduler.py:1137-1211).
    ``chunkable`` uint8 [n_req]: 1 iff the group has exactly one sequence in the walked status
    (scheduler.py:1884; a WAITING prompt with best_of > 1 is chunkable although new_seqs > 1);
    None: ``new_seqs <= 1``.
    Returns (n_selected int32[1] device tensor, ran uint8[N] | None, granted int32[N] | None)."""
    N = perm.numel()
    n_req = new_tokens.numel()            # perm holds request indices < n_req (N <= n_req are queued)
    dev = perm.device
    # when every request is queued the kernel writes every output element: no zero-fill launches
    alloc = torch.empty if n_req == N else torch.zeros
    n_sel = torch.empty(1, dtype=torch.int32, device=dev)
    ran = alloc(n_req, dtype=torch.uint8, device=dev) if want_ran else None
    granted = alloc(n_req, dtype=torch.int32, device=dev) if want_granted else None
    lib = _lib.load()
    if chunkable is not None:
        assert chunkable.dtype == torch.uint8 and chunkable.numel() == n_req
    with torch.cuda.device(dev):
        _lib.check(lib.ltr_budget_prefix(perm.data_ptr(), new_tokens.data_ptr(), new_seqs.data_ptr(), _p(chunkable), N,
                                         int(token_budget), int(max_num_seqs), n_sel.data_ptr(), _p(ran), _p(granted),
                                         _stream(dev)), "ltr_budget_prefix")
    return n_sel, ran, granted


def reserve_select(perm: torch.Tensor, n_selected: torch.Tensor, state: torch.Tensor, phys: torch.Tensor,
                   logical: torch.Tensor, nrun: torch.Tensor, nswap: torch.Tensor, need: int,
                   new_seqs: Optional[torch.Tensor] = None):
    """Victim selection of ``Scheduler.reserve_free_blocks`` (scheduler.py:1376-1452) over the ranked
    order.  ``n_selected``: int32[1] device tensor (``budget_prefix``'s first result).
    ``new_seqs`` None: ``need`` = num_blocks_needed - free GPU blocks + watermark;
    ``new_seqs`` given: ``need`` = free GPU blocks - watermark and the blocks the selection requires
    are accumulated on the device.  Returns (action uint8[n_req], n_exec int32[1], blocks_required int32[1])."""
    N = perm.numel()
    n_req = state.numel()
    dev = perm.device
    action = (torch.empty if n_req == N else torch.zeros)(n_req, dtype=torch.uint8, device=dev)
    n_exec = torch.empty(1, dtype=torch.int32, device=dev)
    req = torch.empty(1, dtype=torch.int32, device=dev)
    lib = _lib.load()
    with torch.cuda.device(dev):
        _lib.check(lib.ltr_reserve_select(perm.data_ptr(), n_selected.data_ptr(), state.data_ptr(), phys.data_ptr(),
                                          logical.data_ptr(), nrun.data_ptr(), nswap.data_ptr(), _p(new_seqs), N,
                                          int(need), action.data_ptr(), n_exec.data_ptr(), req.data_ptr(),
                                          _stream(dev)), "ltr_reserve_select")
    return action, n_exec, req


class DeviceQueue:
    """Device-resident ranking state of the scheduler queue: ``score``, ``pri``, ``idle``,
    ``runs`` per SLOT.  A request owns a slot from arrival to completion; new slots start with
    ``idle = runs = pri = 0`` (scheduler.py:372-374).  Two ways to use it:

    * dense: ``append(scores)`` hands out slots ``n, n+1, ...``; ``rank()`` / ``age(ran)`` / ``step()``
      operate on slots ``[0, n)`` in slot order (bench, smoke);
    * sparse (behind the scheduler plug-in): ``alloc_slots`` / ``free_slots`` manage a free list and every
      call passes ``members`` = the slots in the order list(waiting)+list(running)+list(swapped).
    """

    def __init__(self, device, starv: int = -1, period: int = 0, capacity: int = 1024):
        self.device = torch.device(device)
        self.starv, self.period = int(starv), int(period)
        self.n = 0                          # high-water mark of slots handed out
        self._free: list = []
        self._out = None
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self._alloc(capacity)
        self.ws = RankWorkspace(self.device)

    def _alloc(self, cap: int):
        new = [torch.zeros(cap, dtype=dt, device=self.device)
               for dt in (torch.float32, torch.int32, torch.int32, torch.int32)]
        if self.n:
            for t, o in zip(new, (self._score, self._pri, self._idle, self._runs)):
                t[:self.n].copy_(o[:self.n])
        self.cap = cap
        self._score, self._pri, self._idle, self._runs = new

    score = property(lambda self: self._score[:self.n])
    pri = property(lambda self: self._pri[:self.n])
    idle = property(lambda self: self._idle[:self.n])
    runs = property(lambda self: self._runs[:self.n])

    # ---- dense use
    def append(self, scores: torch.Tensor) -> None:
        k = scores.numel()
        if self.n + k > self.cap:
            self._alloc(max(2 * self.cap, self.n + k))
        self._score[self.n:self.n + k] = scores.to(self.device, torch.float32)
        self._pri[self.n:self.n + k] = 0
        self._idle[self.n:self.n + k] = 0
        self._runs[self.n:self.n + k] = 0
        self.n += k

    def remove(self, keep_mask: torch.Tensor) -> None:
        """Compact after requests finish (``keep_mask`` bool [n]); dense use only."""
        idx = torch.nonzero(keep_mask.to(self.device), as_tuple=False).flatten()
        k = idx.numel()
        for t in (self._score, self._pri, self._idle, self._runs):
            t[:k] = t[:self.n][idx]
        self.n = k

    # ---- sparse use
    def alloc_slots(self, k: int) -> np.ndarray:
        """k slot numbers (int32, host) with zeroed counters; reuses freed slots first."""
        take = min(k, len(self._free))
        got = [self._free.pop() for _ in range(take)]
        fresh = k - take
        if fresh:
            if self.n + fresh > self.cap:
                self._alloc(max(2 * self.cap, self.n + fresh))
            got.extend(range(self.n, self.n + fresh))      # fresh slots are zero already (torch.zeros / never used)
            self.n += fresh
        slots = np.asarray(got, np.int32)
        if take:
            idx = torch.from_numpy(slots[:take].astype(np.int64)).to(self.device)
            for t in (self._pri, self._idle, self._runs):
                t.index_fill_(0, idx, 0)
        return slots

    def free_slots(self, slots: Sequence[int]) -> None:
        self._free.extend(int(s) for s in slots)

    def set_scores(self, slots_dev: torch.Tensor, scores: torch.Tensor) -> None:
        """``slots_dev`` int64 [k] on the device."""
        self._score.index_copy_(0, slots_dev, scores.to(self.device, torch.float32))

    # ---- steps
    def _view(self, members):
        if members is None:
            return self.score, self.pri, self.idle, self.runs
        return self._score, self._pri, self._idle, self._runs

    def rank(self, out: Optional[torch.Tensor] = None, members: Optional[torch.Tensor] = None) -> torch.Tensor:
        s, p, i, r = self._view(members)
        return rank_step(s, p, i, r, self.starv, self.period, self.ws, out=out, members=members)

    def age(self, ran: Optional[torch.Tensor] = None, members: Optional[torch.Tensor] = None,
            ran_slots: Optional[torch.Tensor] = None) -> None:
        _, p, i, r = self._view(members)
        age_update(ran, p, i, r, members=members, ran_slots=ran_slots)

    def step(self, new_tokens: torch.Tensor, new_seqs: torch.Tensor, token_budget: int, max_num_seqs: int,
             members: Optional[torch.Tensor] = None, chunkable: Optional[torch.Tensor] = None,
             perm_out: Optional[torch.Tensor] = None, want_ran: bool = True, want_granted: bool = False):
        """One steady scheduler step in two launches (``ltr_queue_step``): promote/demote + rank, budget-walk
        selection, aging with the selection as ``ran``.  ``new_tokens / new_seqs / chunkable`` are indexed by
        position.  Returns (perm int32 [N], n_selected int32 [1], ran uint8 [N] | None, granted int32 [N] | None)."""
        s, p, i, r = self._view(members)
        N = members.numel() if members is not None else self.n
        dev = self.device
        # outputs live in buffers owned by the queue (valid until the next step): a steady step is two ~10 us
        # launches, three torch.empty calls would cost as much again on the host
        ob = self._out
        if ob is None or ob[0].numel() < N:
            cap = max(N, 2 * (ob[0].numel() if ob else 0))
            ob = self._out = (torch.empty(cap, dtype=torch.int32, device=dev), torch.empty(1, dtype=torch.int32, device=dev),
                              torch.empty(cap, dtype=torch.uint8, device=dev), torch.empty(cap, dtype=torch.int32, device=dev))
        if perm_out is None:
            perm_out = ob[0][:N]
        n_sel = ob[1]
        ran = ob[2][:N] if want_ran else None
        granted = ob[3][:N] if want_granted else None
        buf = self.ws.get(N)

        def call():
            _lib.check(self.ws.lib.ltr_queue_step(s.data_ptr(), p.data_ptr(), i.data_ptr(), r.data_ptr(), None,
                                                  _p(members), N, self.starv, self.period,
                                                  _flags(self.starv, False, None), new_tokens.data_ptr(),
                                                  new_seqs.data_ptr(), _p(chunkable), int(token_budget),
                                                  int(max_num_seqs), perm_out.data_ptr(), n_sel.data_ptr(), _p(ran),
                                                  _p(granted), buf.data_ptr(), buf.numel(), _stream(dev)),
                       "ltr_queue_step")
        if torch.cuda.current_device() == dev.index:     # the usual case: skip the context manager (a few us per step)
            call()
        else:
            with torch.cuda.device(dev):
                call()
        return perm_out, n_sel, ran, granted
