"""Scoring at arrival for :class:`~vllm_ltr_amd.plugin.MI355XRanker` (``prescore=True``; off by default - the reference scores at the
step, scheduler.py:971-982).

``add_request`` (where ``Scheduler.add_seq_group`` runs, scheduler.py:368-376) starts the request's predictor forward
asynchronously on a side stream; the scheduler step that first needs the score collects it (``_collect_prescored``) instead of
running the forward inside the step.  Arrivals closer together than one launch's host time share a forward; a LONE arrival
replays a captured graph (one per 64-token bucket, dummy-padded): host ~0.2 ms instead of 0.44 / 0.69 ms of eager launches.
Split out of plugin.py in round 6; the state lives on the ranker (``self._pre_*``), this mixin holds the
methods.
"""
from __future__ import annotations

import collections
import time
from typing import List, Optional

import numpy as np
import torch

from .host_pipeline import InputStager, PinnedI32 as _PinnedI32, cached_token_ids


class PrescoreMixin:
    PRESCORE_WINDOW_S = 3e-4     # prescore: arrivals closer together than the host time of one launch share a forward
    PRESCORE_BURST_S = 5e-3      # prescore: launches inside this window count as one burst (growing batches)
    PRESCORE_GRAPH_BUCKET = 64   # prescore: one captured graph per this many tokens of (prompt + >= 1 dummy token)
    PRESCORE_ORPHAN_S = 30.0     # prescore: a finished batch nobody collected for this long (aborted requests) is dropped
    PRESCORE_MAX_STAGERS = 16    # prescore: pinned staging sets kept for reuse (more in flight: allocated, then freed)

    def _init_prescore(self, prescore: bool, prescore_graphs: bool, group, driver_rank) -> None:
        # ---- asynchronous scoring at arrival (prescore=True)
        if prescore and group is not None and driver_rank is None:
            raise ValueError("prescore=True does not combine with an SPMD group=: the ranks of a sharded call must make the "
                             "same collective calls, and arrivals are not synchronised across ranks (with driver_rank= the "
                             "arrival-time forwards are the driver's own and the combination is fine)")
        self.prescore = bool(prescore)
        self.prescore_graphs = bool(prescore_graphs)
        self._pre_static: Optional[dict] = None
        self._pre_ws_key = f"prescore-{id(self):x}"      # scoring scratch of THIS ranker's prescore stream (rankers may share a scorer)
        self._pre_stream = torch.cuda.Stream(self.device) if self.prescore else None
        self._pre_pending: list = []                     # arrivals not yet launched
        self._pre_inflight: collections.deque = collections.deque()    # launched batches, oldest first
        self._pre_free_stagers: list = []
        self._pre_up = _PinnedI32(self.device, 1 << 8) if self.prescore else None
        self._pre_recent: collections.deque = collections.deque()     # issue times of the launches of the last PRESCORE_BURST_S
        self.stats.update(prescore_launches=0, prescore_graph_replays=0, prescored_requests=0, prescore_wait_seconds=0.0,
                          arrival_hook_seconds=0.0, prescore_orphans=0, range_fallbacks=0)


    def _prescore_pump(self, force: bool = False) -> None:
        """Launch a forward over the pending arrivals - a lone arrival always starts at once - unless
        * the previous launch was issued less than ``PRESCORE_WINDOW_S`` ago (arrivals closer together than the host time of
          a launch share a forward), or
        * this is a burst: from the third launch inside ``PRESCORE_BURST_S`` on, a launch needs twice as many pending
          arrivals as the one before (2, 4, 8, ...: a burst of N arrivals costs ~log2 N launches however slow the host
          is, instead of one-request forwards issued back to back), or
        * two launched batches are still unfinished (one running, one queued behind it).
        What is held back goes with the next pump - every ``add_request``; the end of every scheduler step (``age``), which
        forces it - or is scored by ``obtain_aux_scores`` itself when the step asks for it first."""
        if not self._pre_pending:
            return
        now = time.perf_counter()
        recent = self._pre_recent
        while recent and now - recent[0] > self.PRESCORE_BURST_S:
            recent.popleft()
        if not force:
            if recent and now - recent[-1] < self.PRESCORE_WINDOW_S:
                return
            if len(recent) >= 2 and len(self._pre_pending) < min(1 << (len(recent) - 1), 4096):
                return
            if len(self._pre_inflight) >= 2 and not self._pre_inflight[-1]["event"].query() \
                    and not self._pre_inflight[-2]["event"].query():
                return
        recent.append(now)
        batch, self._pre_pending = self._pre_pending, []
        arrays = [cached_token_ids(sg, self.tokenize, self.max_length) for sg in batch]
        stager = self._pre_free_stagers.pop() if self._pre_free_stagers else InputStager(self.device, 1 << 12, 1 << 6)
        with torch.cuda.stream(self._pre_stream):
            graph = self._prescore_graph(arrays[0], stager) if len(batch) == 1 and self.prescore_graphs else None
            if graph is not None:
                graph.replay()                             # ~30 us of host time instead of ~85 launches
                scores_dev = self._pre_static["out"]
            else:
                ids_dev, cu_dev, cu_host = stager.stage(arrays)
                scores_dev = self.scorer.score_device(ids_dev, cu_dev, cu_host, workspace_key=self._pre_ws_key)
            scores_host = stager._sc_h[:len(batch)]
            scores_host.copy_(scores_dev[:len(batch)], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self._pre_stream)
        if graph is not None:
            self.stats["prescore_graph_replays"] += 1
        rec = dict(reqs=batch, scores_host=scores_host, event=ev, stager=stager, left=len(batch), t=now)
        for i, sg in enumerate(batch):
            sg._ltr_pre = (rec, i)
        self._pre_inflight.append(rec)
        self.stats["prescore_launches"] += 1
        if len(self._pre_inflight) > 8:
            self._prescore_sweep(now)

    def _prescore_graph(self, ids: np.ndarray, stager) -> Optional["torch.cuda.CUDAGraph"]:
        """A one-request forward as a captured graph (the common prescore launch: a lone arrival).  The request is padded
        to a bucket of ``PRESCORE_GRAPH_BUCKET`` tokens by a second, dummy request (requests do not see each other: the
        score of the first is what it is alone, up to the <= 2e-6 of the batch it is scored in), so that one graph per bucket
        serves every prompt length in it.  The graphs share one set of static device buffers and one workspace - every
        replay runs on the prescore stream, in order; the inputs reach them through this launch's own pinned staging.
        Called on the prescore stream; stages the inputs and returns the graph to replay, or None (bucket not capturable:
        the eager path takes over)."""
        B = self.PRESCORE_GRAPH_BUCKET
        L = int(ids.shape[0])
        Tp = (L + 1 + B - 1) // B * B                      # room for the real request and a dummy of >= 1 token
        st = self._pre_static
        if st is None:
            cap = (self.max_length + 1 + B - 1) // B * B
            st = self._pre_static = dict(cap=cap, ids=torch.empty(cap, dtype=torch.int64, device=self.device),
                                         cu=torch.empty(3, dtype=torch.int32, device=self.device),
                                         out=torch.empty(2, dtype=torch.float32, device=self.device), graphs={})
            self.scorer._workspace(2, cap, self._pre_ws_key + "-graph")    # full size now: the graphs hold its address
        if Tp > st["cap"]:
            return None
        g = st["graphs"].get(Tp)
        if g is False:
            return None
        stager._grow(Tp, 2)
        h_ids, h_cu = stager._ids_h.numpy(), stager._cu_h.numpy()
        h_ids[:L] = ids
        h_ids[L] = 2                                       # the dummy: BOS + filler
        h_ids[L + 1:Tp] = 4
        h_cu[0], h_cu[1], h_cu[2] = 0, L, Tp
        ids_d, cu_d = st["ids"][:Tp], st["cu"]
        ids_d.copy_(stager._ids_h[:Tp], non_blocking=True)
        cu_d.copy_(stager._cu_h[:3], non_blocking=True)
        if g is None:
            try:
                cu_host = np.array([0, L, Tp], np.int32)
                # once outside a capture (one-time initialisation inside the library), then captured
                self.scorer.score_device(ids_d, cu_d, cu_host, out=st["out"], workspace_key=self._pre_ws_key + "-graph")
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=self._pre_stream):
                    self.scorer.score_device(ids_d, cu_d, cu_host, out=st["out"], workspace_key=self._pre_ws_key + "-graph")
            except Exception:      # noqa: BLE001 - a runtime that cannot capture: this bucket stays eager
                st["graphs"][Tp] = False
                return None
            st["graphs"][Tp] = g
        return g

    def warm_prescore_graphs(self) -> int:
        """Capture the graph of every bucket now (otherwise each is captured by the first arrival that needs it, a few
        milliseconds of host time).  Returns the number of graphs."""
        if not (self.prescore and self.prescore_graphs):
            return 0
        B = self.PRESCORE_GRAPH_BUCKET
        stager = InputStager(self.device, 1 << 12, 1 << 6)
        while len(self._pre_free_stagers) < 4:               # pinned allocations are slow: not in the first arrivals' hooks
            self._pre_free_stagers.append(InputStager(self.device, 1 << 12, 1 << 6))
        with torch.cuda.stream(self._pre_stream):
            cap = (self.max_length + 1 + B - 1) // B * B
            for Tp in range(B, cap + 1, B):                  # the longest prompt of every bucket
                self._prescore_graph(np.full(min(Tp - 1, self.max_length), 4, np.int64), stager)
            self._pre_stream.synchronize()
        return sum(1 for g in self._pre_static["graphs"].values() if g)

    def _prescore_retire(self, rec) -> None:
        for sg in rec["reqs"]:
            if getattr(sg, "_ltr_pre", None) is not None and sg._ltr_pre[0] is rec:
                sg._ltr_pre = None
        rec["reqs"] = ()                                    # (no reference cycle request -> record -> request left behind)
        if len(self._pre_free_stagers) < self.PRESCORE_MAX_STAGERS:
            self._pre_free_stagers.append(rec["stager"])
        rec["stager"] = None

    def _prescore_sweep(self, now: Optional[float] = None) -> None:
        """Retire every launched batch that is done with - all its requests collected or aborted - WHEREVER it sits in the
        list (one aborted request at the head must not keep every later batch and its pinned staging alive), and batches
        whose forward finished ``PRESCORE_ORPHAN_S`` ago without anybody asking for the scores (requests aborted without
        :meth:`abort_request`): should such a request still show up in a scheduler step, the step scores it itself."""
        now = time.perf_counter() if now is None else now
        keep = collections.deque()
        for rec in self._pre_inflight:
            if rec["left"] <= 0 and rec["event"].query():
                self._prescore_retire(rec)
            elif now - rec["t"] > self.PRESCORE_ORPHAN_S and rec["event"].query():
                self.stats["prescore_orphans"] += rec["left"]
                self._prescore_retire(rec)
            else:
                keep.append(rec)
        self._pre_inflight = keep

    def abort_request(self, sg) -> None:
        """Optional hook where the engine aborts a request (``Scheduler.abort_seq_group``, scheduler.py:378-409): forget the
        score started for it at arrival.  Without it the record is dropped by age (``PRESCORE_ORPHAN_S``)."""
        if not self.prescore:
            return
        pre = getattr(sg, "_ltr_pre", None)
        if pre is not None:
            pre[0]["left"] -= 1
            sg._ltr_pre = None
        if self._pre_pending:
            self._pre_pending = [g for g in self._pre_pending if g is not sg]

    def _collect_prescored(self, seq_groups, out) -> list:
        """Scores of the requests whose forward was started at arrival: into their device slots (the ordering reads them
        there) and into ``out``.  Returns the positions that still need a forward (arrivals never handed to
        ``add_request``, or still waiting for a launch)."""
        pending = {id(sg) for sg in self._pre_pending}
        if pending:                                        # not launched yet: they go into this step's own batch
            ask = {id(sg) for sg in seq_groups}
            self._pre_pending = [sg for sg in self._pre_pending if id(sg) not in ask]
        todo, by_rec = [], {}
        for i, sg in enumerate(seq_groups):
            pre = getattr(sg, "_ltr_pre", None)
            if pre is None:
                todo.append(i)
            else:
                by_rec.setdefault(id(pre[0]), (pre[0], []))[1].append((i, pre[1], sg))
        n_pre = sum(len(items) for _, items in by_rec.values())
        if n_pre:
            # one pinned upload for all of them: [slots | score bits] as int32 (the host values are the D2H copies the
            # forwards left behind; the slots then hold exactly what the request objects get)
            self._pre_up.ensure(2 * n_pre)
            buf = self._pre_up.np
            buf_f = buf.view(np.float32)
            k = 0
            for rec, items in by_rec.values():
                t0 = time.perf_counter()
                rec["event"].synchronize()                 # normally long done: the forward ran during the backbone step
                self.stats["prescore_wait_seconds"] += time.perf_counter() - t0
                self._assign_slots([sg for _, _, sg in items], being_scored=True)
                host = rec["scores_host"].numpy()
                for i, j, sg in items:
                    buf[k] = self._get_slot(sg)
                    buf_f[n_pre + k] = host[j]
                    out[i] = float(host[j])
                    sg._ltr_pre = None
                    k += 1
                rec["left"] -= len(items)
            up = self._pre_up.upload(2 * n_pre)
            self.queue.set_scores(up[:n_pre].to(torch.int64), up[n_pre:].view(torch.float32))
            self.stats["prescored_requests"] += n_pre
        # retire the batches that are done with (their staging buffers go back to the pool)
        self._prescore_sweep()
        return todo

