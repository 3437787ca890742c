"""Host input pipeline of the ranker call.

The reference re-tokenises every prompt inside ``obtain_aux_scores`` and builds a
``Sequence``/``SequenceGroup`` per request for the AUX engine (aux_llm_engine.py:340-394), then
``ModelRunner._prepare_prompt`` flattens Python lists into ``torch.tensor(list)``
(model_runner.py:224-422).  Here:

* the predictor token ids of a request are produced ONCE (at ``add_request`` time or on first use),
  truncated to ``max_length`` and cached on the request as an int64 array (``_ltr_ids``);
* a call packs the cached arrays with one ``np.concatenate`` straight into a reusable pinned
  staging buffer (ids int64 [T] | cu_seqlens int32 [N+1]) and issues two asynchronous H2D copies
  on the caller's stream, so the copy overlaps the previous kernels and no pageable bounce buffer
  is involved; scores come back through a pinned buffer as well.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np
import torch


def cached_token_ids(sg, tokenize: Optional[Callable[[str], List[int]]], max_length: int) -> np.ndarray:
    """int64 predictor token ids of a request, computed once and kept on the request object
    (aux_llm_engine.py:341,365-369: prompt text -> predictor tokenizer -> truncate)."""
    ids = getattr(sg, "_ltr_ids", None)
    if ids is None:
        if tokenize is not None:
            prompt = getattr(sg, "prompt", None)
            if prompt is None:
                prompt = next(iter(sg.seqs_dict.values())).prompt
            raw = tokenize(prompt)
        else:
            raw = sg.prompt_token_ids
        ids = np.asarray(raw[:max_length], dtype=np.int64)
        if ids.size == 0:
            raise ValueError(f"request {sg.request_id}: empty prompt cannot be scored")
        try:
            sg._ltr_ids = ids
        except AttributeError:        # objects with __slots__: recompute next time
            pass
    return ids


class InputStager:
    """Reusable pinned staging + device buffers for (token ids, cu_seqlens, scores)."""

    def __init__(self, device, capacity_tokens: int = 1 << 16, capacity_requests: int = 1 << 10):
        self.device = torch.device(device)
        self._pin = self.device.type == "cuda"
        self._ids_h = self._ids_d = self._cu_h = self._cu_d = self._sc_h = None
        self._grow(capacity_tokens, capacity_requests)

    def _host(self, n, dtype):
        t = torch.empty(n, dtype=dtype)
        return t.pin_memory() if self._pin else t

    def _grow(self, T: int, N: int):
        if self._ids_h is None or self._ids_h.numel() < T:
            cap = max(T, 2 * (self._ids_h.numel() if self._ids_h is not None else 0))
            self._ids_h = self._host(cap, torch.int64)
            self._ids_d = torch.empty(cap, dtype=torch.int64, device=self.device)
        if self._cu_h is None or self._cu_h.numel() < N + 1:
            cap = max(N + 1, 2 * (self._cu_h.numel() if self._cu_h is not None else 0))
            self._cu_h = self._host(cap, torch.int32)
            self._cu_d = torch.empty(cap, dtype=torch.int32, device=self.device)
            self._sc_h = self._host(cap, torch.float32)

    def stage(self, token_arrays: Sequence[np.ndarray]) -> Tuple[torch.Tensor, torch.Tensor, np.ndarray]:
        """Pack + asynchronous H2D.  Returns (ids_dev [T], cu_dev [N+1], cu_host [N+1] view of the
        pinned buffer - valid until the next ``stage``)."""
        N = len(token_arrays)
        lens = np.fromiter((a.shape[0] for a in token_arrays), np.int64, N)
        T = int(lens.sum())
        self._grow(T, N)
        cu = self._cu_h.numpy()[:N + 1]
        cu[0] = 0
        np.cumsum(lens, out=cu[1:])
        if N:
            np.concatenate(token_arrays, out=self._ids_h.numpy()[:T])
        ids_d, cu_d = self._ids_d[:T], self._cu_d[:N + 1]
        ids_d.copy_(self._ids_h[:T], non_blocking=True)
        cu_d.copy_(self._cu_h[:N + 1], non_blocking=True)
        return ids_d, cu_d, cu

    def fetch_scores(self, scores_dev: torch.Tensor) -> np.ndarray:
        """D2H of the scores through the pinned buffer; synchronises the current stream."""
        n = scores_dev.numel()
        self._sc_h[:n].copy_(scores_dev, non_blocking=True)
        if self.device.type == "cuda":
            torch.cuda.current_stream(self.device).synchronize()
        return self._sc_h.numpy()[:n]


class PinnedI32:
    """Growable pinned int32 staging buffer + device twin (members / ran slots / permutation)."""

    def __init__(self, device, cap: int = 1 << 13):
        self.device = device
        self._evt = None
        self._grow(cap)

    def _grow(self, cap: int):
        self.cap = cap
        self.host = torch.empty(cap, dtype=torch.int32).pin_memory()
        self.np = self.host.numpy()
        self.dev = torch.empty(cap, dtype=torch.int32, device=self.device)
        self._evt = None

    def ensure(self, n: int):
        """Call before writing ``self.np``: grows the buffers and waits until the previous asynchronous upload has
        read the pinned buffer (normally long done)."""
        if self._evt is not None:
            self._evt.synchronize()
            self._evt = None
        if n > self.cap:
            self._grow(max(n, 2 * self.cap))

    def upload(self, n: int) -> torch.Tensor:
        d = self.dev[:n]
        d.copy_(self.host[:n], non_blocking=True)
        self._evt = torch.cuda.Event()
        self._evt.record(torch.cuda.current_stream(self.device))
        return d
