"""Array-level binding of the rank kernels (ltr_rank_step / ltr_age_update /
ltr_budget_prefix): starvation promote/demote + stable priority sort + aging on
device-resident ``score / pri / idle / runs`` arrays.

Reference semantics: vllm/core/scheduler.py:984-998 (order), :1358-1365 (aging),
:1137-1211 (budget walk prefix).
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from . import _lib


def _stream(device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


class RankWorkspace:
    """Reusable scratch for ltr_rank_step, grown on demand."""

    def __init__(self, device):
        self.device = torch.device(device)
        self.lib = _lib.load()
        self._buf: Optional[torch.Tensor] = None

    def get(self, N: int) -> torch.Tensor:
        need = int(self.lib.ltr_workspace_bytes(None, _lib.LTR_WS_RANK, N, 0))
        if self._buf is None or self._buf.numel() < need:
            self._buf = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._buf


def rank_step(scores: torch.Tensor, pri: Optional[torch.Tensor], idle: Optional[torch.Tensor],
              runs: Optional[torch.Tensor], starv: int, period: int, ws: RankWorkspace,
              tiebreak: Optional[torch.Tensor] = None, ascending: bool = False, use_pri: Optional[bool] = None,
              out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """In-place promote/demote on (pri, idle, runs) and the ranked permutation
    (int32 [N], ``perm[k]`` = input index of the k-th request).  Asynchronous."""
    N = scores.numel()
    assert scores.dtype == torch.float32 and scores.is_cuda
    for t in (pri, idle, runs):
        assert t is None or (t.dtype == torch.int32 and t.numel() == N and t.is_cuda)
    if tiebreak is not None:
        assert tiebreak.dtype in (torch.int32, torch.uint32) and tiebreak.numel() == N
    if out is None:
        out = torch.empty(N, dtype=torch.int32, device=scores.device)
    if N == 0:
        return out
    flags = (_lib.LTR_RANK_ASCENDING if ascending else 0)
    if use_pri or (use_pri is None and starv != -1):
        flags |= _lib.LTR_RANK_USE_PRI
    buf = ws.get(N)
    p = lambda t: t.data_ptr() if t is not None else None
    _lib.check(ws.lib.ltr_rank_step(scores.data_ptr(), p(pri), p(idle), p(runs), p(tiebreak), N, int(starv),
                                    int(period), flags, out.data_ptr(), buf.data_ptr(), buf.numel(),
                                    _stream(scores.device)), "ltr_rank_step")
    return out


def age_update(ran: torch.Tensor, pri: torch.Tensor, idle: torch.Tensor, runs: torch.Tensor) -> None:
    """In-place scheduler.py:1358-1365; ``ran`` uint8 [N]."""
    N = pri.numel()
    if N == 0:
        return
    assert ran.dtype == torch.uint8 and ran.numel() == N
    lib = _lib.load()
    _lib.check(lib.ltr_age_update(ran.data_ptr(), pri.data_ptr(), idle.data_ptr(), runs.data_ptr(), N,
                                  _stream(pri.device)), "ltr_age_update")


def budget_prefix(perm: torch.Tensor, new_tokens: torch.Tensor, new_seqs: torch.Tensor, token_budget: int,
                  max_num_seqs: int, want_ran: bool = True, want_granted: bool = True):
    """Selection of the budget walk over the ranked order (scheduler.py:1137-1211).
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
    _lib.check(lib.ltr_budget_prefix(perm.data_ptr(), new_tokens.data_ptr(), new_seqs.data_ptr(), N,
                                     int(token_budget), int(max_num_seqs), n_sel.data_ptr(),
                                     ran.data_ptr() if ran is not None else None,
                                     granted.data_ptr() if granted is not None else None, _stream(dev)),
               "ltr_budget_prefix")
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
    _lib.check(lib.ltr_reserve_select(perm.data_ptr(), n_selected.data_ptr(), state.data_ptr(), phys.data_ptr(),
                                      logical.data_ptr(), nrun.data_ptr(), nswap.data_ptr(),
                                      new_seqs.data_ptr() if new_seqs is not None else None, N, int(need),
                                      action.data_ptr(), n_exec.data_ptr(), req.data_ptr(), _stream(dev)),
               "ltr_reserve_select")
    return action, n_exec, req


class DeviceQueue:
    """Device-resident ranking state of the scheduler queue: ``score``, ``pri``,
    ``idle``, ``runs`` per queued request (slot order = the order of
    ``list(waiting)+list(running)+list(swapped)``).  New requests start with
    ``idle = runs = pri = 0`` (scheduler.py:372-374)."""

    def __init__(self, device, starv: int = -1, period: int = 0, capacity: int = 1024):
        self.device = torch.device(device)
        self.starv, self.period = int(starv), int(period)
        self.n = 0
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
        """Compact after requests finish (``keep_mask`` bool [n])."""
        idx = torch.nonzero(keep_mask.to(self.device), as_tuple=False).flatten()
        k = idx.numel()
        for t in (self._score, self._pri, self._idle, self._runs):
            t[:k] = t[:self.n][idx]
        self.n = k

    def rank(self, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        return rank_step(self.score, self.pri, self.idle, self.runs, self.starv, self.period, self.ws, out=out)

    def age(self, ran: torch.Tensor) -> None:
        age_update(ran, self.pri, self.idle, self.runs)
