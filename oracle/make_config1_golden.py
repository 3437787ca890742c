#!/usr/bin/env python3
"""BASELINE config 1 ("OPT-125m predictor, 256-request queue, CPU reference scheduler path") run by the REFERENCE
ITSELF, end to end: the reference's own ``vllm.core.scheduler.Scheduler`` (chunked prefill, ``opt-xxx-starv200-period10``)
with the reference's own fp32 ``OPTForSequenceClassification`` standing where ``llm_engine.py:228-242`` puts the AUXLLM.

    PYTHONPATH=/root/reference PYTHONDONTWRITEBYTECODE=1 python oracle/make_config1_golden.py

Build container only (the GPU box has no /root/reference).  Writes ``tests/golden/config1_opt125m_256.npz``: the
queue (ids, cu_seqlens - ``bench.synthetic_queue(spec, 256, seed 0)``: T = 23,078), the 256 scores the reference
predictor produced, and per scheduler step of three runs the three deques, the order ``_get_ordered_requests`` returned,
the ``ran`` set, the budget-walk inputs and the starvation counters after the aging loop:

* run ``a``: config 1 as BASELINE.json words it (starv 200 / period 10, 2,048-token / 256-sequence budget), 64 requests
  at step 0 and 8 arrivals per step after that, 36 steps; every arrival batch scored by the reference predictor;
* run ``b``: the same scores under ``starv6-period2`` with a tight budget, so that promotions / demotions fire
  (scheduler.py:984-993) - the scores come from run a's reference predictor calls;
* run ``c`` = run a again with the PRODUCT's ``MI355XRanker.install()`` wiring on the same real ``Scheduler``
  (``plugin.py``; device pieces replaced by recording doubles - there is no GPU here): pins the attribute surface
  ``install()`` touches (scheduler.py:290-331,1101-1105,1337-1365) and must reproduce run a's orders and counters.

``--config 3`` writes ``tests/golden/config3_opt350m_128.npz`` instead: BASELINE config 3's predictor (OPT-350m: post-LN blocks,
project_in / project_out, 24 layers) on 128 requests of the LMSYS-like length profile (``synthetic_queue(spec, 128, seed 0,
"lmsys")``), runs ``a`` (starv 200 / period 10) and ``b`` (starv 6 / period 2, tight budget) as above - the second model
family through the reference's own Scheduler + predictor.

``--config xpt`` writes ``tests/golden/config1_xpt.npz``: config 1's queue and reference scores under ``xpt{table}`` (score ->
expected length through a (key, value) table on ``round(-score, 2)``, order by ``expected_length - output_len``,
scheduler.py:910-933), 60 steps, with the output lengths the key saw at every step.

``--config tpt`` writes ``tests/golden/config1_tpt_class82.npz``: config 1's queue (the ids live in ``config1_opt125m_256.npz``)
scheduled by the reference under ``tpt`` - the class-mode predictor (``OPTSpec.opt_125m(82)``: ``compute_logits`` returns
``float(argmax)``, opt.py:394-395) and the order ``(-score, request_id)`` with STRING request ids (scheduler.py:938-948): 82
labels over 256 requests, i.e. ties everywhere, broken by ``"10" < "9"``.  Also the reference's top-2 logit gap per request.

``--config outlier`` writes ``tests/golden/outlier_opt125m_64.npz``: the reference's fp32 predictor on a checkpoint with the
structure of TRAINED OPT weights (``opt_spec.structured_checkpoint``: 5x init scale, two massive embedding channels at
+40 / -55, LayerNorm gains in [0.2, 3]) - 64 requests incl. L = 1, 2 and 1024 - and the order of one cold scheduler step
(``opt``, no starvation) the reference's own Scheduler returns for those scores.  ``--config outlier350``: the same for the
OPT-350m shape (post-LN blocks, project_in / project_out; 48 requests) -> ``tests/golden/outlier_opt350m_48.npz``.

``--config 2`` writes ``tests/golden/config2_opt125m_8192.npz``: BASELINE config 2 at FULL size (8,192 requests, 708,977 tokens) as one
cold step of the reference's Scheduler + fp32 predictor: all 8,192 reference scores and the order (see ``main_config2``);
``--config 3full`` the same for BASELINE config 3 (OPT-350m, 8,192 LMSYS-like requests) -> ``tests/golden/config3_opt350m_8192.npz``;
``--config 4full`` for BASELINE config 4's queue (OPT-125m, 65,536 requests, 5,734,532 tokens: scores + order only) ->
``tests/golden/config4_opt125m_65536.npz`` (~40 min on 8 cores).

Nothing of the reference is copied: the fixture holds inputs and what the reference computed.
"""
from __future__ import annotations

import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import make_golden as mg  # noqa: E402  (import shims + scheduler helpers; asserts vllm is /root/reference)
import torch  # noqa: E402

from bench import synthetic_queue  # noqa: E402
from vllm_ltr_amd.opt_spec import OPTSpec, seeded_checkpoint  # noqa: E402

GOLD = mg.GOLD
N_REQ = 256


class RefPredictor:
    """The reference's own OPTForSequenceClassification (fp32, TorchSDPA backend), driven like
    ModelRunner.execute_model drives it for a prefill batch (model_runner.py:827-877), FCFS packs of <= 2048 tokens
    per forward like the AUX engine's scheduler (config.py:578-586)."""

    def __init__(self, spec, ckpt):
        from transformers import OPTConfig
        from vllm.model_executor.models.opt import OPTForSequenceClassification as REF
        self.cfg = OPTConfig(**spec.to_hf_config_kwargs())
        torch.manual_seed(0)
        self.ref = REF(self.cfg).eval().float()
        self.ref.load_weights([(k, torch.from_numpy(v.astype(np.float32))) for k, v in ckpt.items()])
        self.calls = []          # request ids per obtain_aux_scores call
        self.seconds = 0.0
        self.top2_gap = {}       # class mode: request -> (top-1 logit) - (top-2 logit) of the reference's own logits
        self._rows_ids = None

    def _forward(self, rows):
        from vllm.attention.backends.torch_sdpa import TorchSDPAMetadata
        from vllm.model_executor.sampling_metadata import SamplingMetadata
        lens = [len(r) for r in rows]
        N, T = len(lens), sum(lens)
        ids = torch.from_numpy(np.concatenate(rows).astype(np.int64))
        pos = torch.cat([torch.arange(L) for L in lens]).long()
        sel = torch.as_tensor(np.cumsum(lens).astype(np.int64) - 1)
        md = TorchSDPAMetadata(context_lens=None, max_context_len=None, block_tables=torch.tensor([]),
                               num_prefills=N, num_prefill_tokens=T, num_decode_tokens=0, prefill_metadata=object(),
                               decode_metadata=None, slot_mapping=torch.zeros(T, dtype=torch.long), kv_cache_dtype="auto",
                               need_score=False, selected_token_indices=sel, is_prompt=True, prompt_lens=lens)
        sm = SamplingMetadata(seq_groups=[], seq_data={}, prompt_lens=lens, selected_token_indices=sel,
                              categorized_sample_indices=None, generators=None, perform_sampling=False)
        with torch.no_grad():
            hs, _ = self.ref(ids, pos, [None] * self.cfg.num_hidden_layers, md)
            if self.cfg.num_labels > 1 and self._rows_ids is not None:
                # the logits the reference's argmax sees (logits_processor.py:61-79: last-token rows x score.weight^T)
                raw = self.ref.logits_processor(self.ref.score.weight, hs, sm)
                top = torch.topk(raw.float(), 2, dim=-1).values
                for rid, g in zip(self._rows_ids, (top[:, 0] - top[:, 1]).tolist()):
                    self.top2_gap[rid] = g
            return self.ref.compute_logits(hs, sm)[:, 0].float().tolist()          # opt.py:399-409 .tolist()

    def obtain_aux_scores(self, sgs):                      # AUXLLM.obtain_aux_scores (aux_llm.py:125-126)
        t0 = time.time()
        self.calls.append([sg.request_id for sg in sgs])
        pack, pack_ids, tok, out = [], [], 0, []
        for sg in sgs:
            assert sg.need_aux_model_score()               # aux_llm_engine.py:409
            row = np.asarray(sg.prompt_token_ids, np.int64)
            if pack and tok + len(row) > 2048:
                self._rows_ids = pack_ids
                out += self._forward(pack)
                pack, pack_ids, tok = [], [], 0
            pack.append(row)
            pack_ids.append(sg.request_id)
            tok += len(row)
        if pack:
            self._rows_ids = pack_ids
            out += self._forward(pack)
        for sg, s in zip(sgs, out):
            sg.set_aux_model_score(s)                      # aux_llm_engine.py:408-410
        self.seconds += time.time() - t0


def mk_sg(rid: int, token_ids, block_size=16):
    from vllm import SamplingParams
    from vllm.sequence import Sequence, SequenceGroup
    seq = Sequence(rid, "p", [int(t) for t in token_ids], block_size)
    return SequenceGroup(str(rid), [seq], SamplingParams(max_tokens=10**6, ignore_eos=True), time.time())


def run(tag, schedule_type, aux, ids, cu, arrive_at, steps, max_tokens, max_seqs, out, install=None):
    """Multi-step run of the reference's Scheduler.schedule(); records what a replay needs (see module docstring).
    ``install``: callable(scheduler) that wires a ranker INSTEAD of assigning aux_model directly."""
    from vllm.sequence import Logprob, SequenceStatus
    n = len(cu) - 1
    s = mg._mk_scheduler(schedule_type, max_tokens, max_seqs, blocks=8192)
    if install is None:
        s.aux_model = aux
    else:
        install(s)
    sgs = [mk_sg(i, ids[cu[i]:cu[i + 1]]) for i in range(n)]
    captured = {}
    inner = s._get_ordered_requests

    def spy():
        captured["deques"] = (len(s.waiting), len(s.running), len(s.swapped))
        captured["concat"] = [int(g.request_id) for g in list(s.waiting) + list(s.running) + list(s.swapped)]
        o = inner()
        captured["order"] = [int(g.request_id) for g in o]
        return o
    s._get_ordered_requests = spy
    rec = {k: [] for k in ("concat", "order", "deques", "ran", "states", "need_tokens", "need_seqs", "chunkable", "granted", "out_len")}
    for step in range(steps):
        for i in np.nonzero(arrive_at == step)[0]:
            s.add_seq_group(sgs[i])
        nd = np.zeros(n, np.int32); nq = np.zeros(n, np.int32); ck = np.zeros(n, np.uint8); ol = np.zeros(n, np.int32)
        for dq in (s.waiting, s.running, s.swapped):
            for g in dq:                                       # what the xpt key subtracts (scheduler.py:933)
                ol[int(g.request_id)] = g.seqs_dict[next(iter(g.seqs_dict))].data.get_output_len()
        for dq, status in ((s.waiting, SequenceStatus.WAITING), (s.running, SequenceStatus.RUNNING),
                           (s.swapped, SequenceStatus.SWAPPED)):
            for g in dq:
                seqs = g.get_seqs(status=status)
                nd[int(g.request_id)] = sum(q.get_num_new_tokens() for q in seqs)     # scheduler.py:1878-1881
                nq[int(g.request_id)] = g.get_max_num_running_seqs()
                ck[int(g.request_id)] = len(seqs) == 1                                # :1884
        metas, o = s.schedule()
        gr = np.zeros(n, np.int32)
        ran = np.zeros(n, np.uint8)
        for x, meta in zip(o.scheduled_seq_groups, metas):
            g = x.seq_group
            ran[int(g.request_id)] = 1
            gr[int(g.request_id)] = meta.token_chunk_size * len(g.get_seqs(status=SequenceStatus.RUNNING))
            g.update_num_computed_tokens(meta.token_chunk_size)
            if not g.is_prefill():
                for seq in g.get_seqs(status=SequenceStatus.RUNNING):
                    seq.append_token_id(1, {1: Logprob(0.0)})
        c = np.full(n, -1, np.int32); c[:len(captured["concat"])] = captured["concat"]
        od = np.full(n, -1, np.int32); od[:len(captured["order"])] = captured["order"]
        st = np.zeros((n, 3), np.int32)
        for g in sgs:
            if hasattr(g, "pri"):
                st[int(g.request_id)] = (g.pri, g.idle, g.runs)
        for k, v in (("concat", c), ("order", od), ("deques", np.asarray(captured["deques"], np.int32)), ("ran", ran),
                     ("states", st), ("need_tokens", nd), ("need_seqs", nq), ("chunkable", ck), ("granted", gr), ("out_len", ol)):
            rec[k].append(v)
    for k, v in rec.items():
        out[f"{tag}_{k}"] = np.stack(v)
    starv, period = s.starv, getattr(s, "period", 0)
    out[f"{tag}_starv"], out[f"{tag}_period"] = np.int64(starv), np.int64(period)
    out[f"{tag}_arrive_at"] = arrive_at.astype(np.int32)
    out[f"{tag}_token_budget"], out[f"{tag}_max_num_seqs"] = np.int64(max_tokens), np.int64(max_seqs)
    promoted = int((np.stack(rec["states"])[:, :, 0] == -1).any(axis=0).sum())
    chunked = int(((np.stack(rec["granted"]) > 0) & (np.stack(rec["granted"]) < np.stack(rec["need_tokens"]))).sum())
    print(f"run {tag}: {schedule_type}, {steps} steps, budget {max_tokens}/{max_seqs}: requests ever promoted {promoted}, "
          f"chunked grants {chunked}, first order {rec['order'][0][:8].tolist()}")
    return s, sgs, rec


class ScoreTable:
    """Stands where the AUXLLM stands with scores a reference predictor call produced earlier (run a)."""

    def __init__(self, table):
        self.table, self.calls = table, []

    def obtain_aux_scores(self, sgs):
        self.calls.append([sg.request_id for sg in sgs])
        for sg in sgs:
            assert sg.need_aux_model_score()
            sg.set_aux_model_score(self.table[sg.request_id])


def recording_ranker(schedule_type, aux, log):
    """The PRODUCT's MI355XRanker with its device pieces replaced by recording doubles (no GPU in the build
    container): install(), ordered_requests() and the wrapped _schedule are plugin.py's own code; scoring goes to the
    reference predictor / score table, the order to the oracle's literal restatement on the HOST attributes (which
    the reference's own aging loop, scheduler.py:1358-1365, keeps up to date)."""
    from oracle import rank_step as rs
    from vllm_ltr_amd.plugin import MI355XRanker
    from vllm_ltr_amd.schedule_type import parse_schedule_type

    class RecordingRanker(MI355XRanker):
        def __init__(self):
            self.st = parse_schedule_type(schedule_type)
            self.xpt_distribution = None
            self._aged = True

        def obtain_aux_scores(self, seq_groups):
            log.append(["obtain_aux_scores", len(seq_groups)])
            aux.obtain_aux_scores(seq_groups)
            return [sg.aux_model_score for sg in seq_groups]

        def order(self, reqs, policy=None, want_list=True):
            self._aged = False
            log.append(["order", len(reqs)])
            return rs.opt_order(reqs, self.st.starv, self.st.period)

        def age(self, all_pri, running_this_step):
            self._aged = True
            log.append(["age", len(all_pri), len(list(running_this_step))])

        def _gc_slots(self, n):
            pass
    return RecordingRanker()


def main_config3():
    """OPT-350m, 128 LMSYS-like requests: runs a and b through the reference's own Scheduler + predictor."""
    mg._init_dist()
    torch.set_num_threads(os.cpu_count())
    n = 128
    spec = OPTSpec.opt_350m()
    ckpt = seeded_checkpoint(spec, 0)
    ids, cu, lens = synthetic_queue(spec, n, seed=0, profile="lmsys")
    out = dict(ids=ids.astype(np.int32), cu_seqlens=cu, seed=np.int64(0))
    arrive_a = np.zeros(n, np.int32)
    arrive_a[32:] = 1 + (np.arange(n - 32) // 8)               # 32 at step 0, then 8 per step (steps 1..12)
    pred = RefPredictor(spec, ckpt)
    t0 = time.time()
    s_a, sgs_a, rec_a = run("a", "opt-xxx-starv200-period10", pred, ids, cu, arrive_a, 40, 2048, 256, out)
    scores = np.array([g.aux_model_score for g in sgs_a], np.float64)
    assert np.isfinite(scores).all() and len(pred.calls) == 13 and sum(map(len, pred.calls)) == n
    out["ref_score"] = scores.astype(np.float32)
    assert np.array_equal(out["ref_score"].astype(np.float64), scores)
    out["a_aux_calls"] = np.array([len(c) for c in pred.calls], np.int32)
    print(f"config 3 run a: T = {int(cu[-1])}, {len(pred.calls)} predictor calls, {pred.seconds:.1f} s in the reference "
          f"predictor ({n / pred.seconds:.1f} req/s on {os.cpu_count()} threads), total {time.time()-t0:.1f} s; score range "
          f"[{scores.min():.4f}, {scores.max():.4f}], smallest gap between sorted scores {np.diff(np.sort(scores)).min():.3e}")
    table = {str(i): float(out["ref_score"][i]) for i in range(n)}
    arrive_b = np.sort(np.random.RandomState(5).randint(0, 16, n)).astype(np.int32)
    run("b", "opt-xxx-starv6-period2", ScoreTable(table), ids, cu, arrive_b, 48, 768, 16, out)
    path = os.path.join(GOLD, "config3_opt350m_128.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {os.path.getsize(path) / 1024:.0f} KB")


def main_config2(which="2"):
    """(``which`` = "3full": BASELINE config 3 at full size instead - OPT-350m, 8,192 LMSYS-like requests, 1,407,401 tokens - into
    ``tests/golden/config3_opt350m_8192.npz``; ~35 min on 8 cores.)
    BASELINE config 2 at FULL size - the headline workload: OPT-125m, 8,192 requests, 708,977 tokens - as ONE cold step of the
    reference's own Scheduler (``opt-xxx-starv200-period10``): every request is scored by the reference's fp32 predictor in the AUX
    engine's FCFS packs (one ``obtain_aux_scores`` call), then ordered by ``_get_ordered_requests``.  The fixture holds the 8,192
    reference scores, the order, the budget-walk inputs / grants of that step and a SHA-256 of the token ids (the queue itself is
    ``bench.synthetic_queue(spec, 8192, seed 0)``: regenerated by the test, not stored).  ~25 min on 8 cores."""
    import hashlib
    mg._init_dist()
    torch.set_num_threads(os.cpu_count())
    n = 65536 if which == "4full" else 8192
    full3 = which == "3full"
    spec = OPTSpec.opt_350m() if full3 else OPTSpec.opt_125m()
    ckpt = seeded_checkpoint(spec, 0)
    ids, cu, lens = synthetic_queue(spec, n, seed=0, profile="lmsys") if full3 else synthetic_queue(spec, n, seed=0)
    assert int(cu[-1]) == (1407401 if full3 else (5734532 if n == 65536 else 708977)), int(cu[-1])   # BASELINE.json configs[2] / [3] / [1]
    out = dict(cu_seqlens=cu, seed=np.int64(0),
               ids_sha256=np.frombuffer(hashlib.sha256(np.ascontiguousarray(ids.astype(np.int32)).tobytes()).digest(), np.uint8))
    pred = RefPredictor(spec, ckpt)
    t0 = time.time()
    s_a, sgs_a, rec_a = run("a", "opt-xxx-starv200-period10", pred, ids, cu, np.zeros(n, np.int32), 1, 2048, 256, out)
    scores = np.array([g.aux_model_score for g in sgs_a], np.float64)
    assert np.isfinite(scores).all() and len(pred.calls) == 1 and len(pred.calls[0]) == n
    out["ref_score"] = scores.astype(np.float32)
    assert np.array_equal(out["ref_score"].astype(np.float64), scores)
    gaps = np.diff(np.sort(scores))
    print(f"config {'3' if full3 else ('4' if n == 65536 else '2')} (full size): T = {int(cu[-1])}, one predictor call of {n} requests, {pred.seconds:.1f} s in the reference "
          f"predictor ({n / pred.seconds:.1f} req/s on {os.cpu_count()} threads), total {time.time()-t0:.1f} s; score range "
          f"[{scores.min():.4f}, {scores.max():.4f}]; gaps between sorted scores: min {gaps.min():.3e}, "
          f"{int((gaps < 2e-6).sum())} below 2e-6, {int((gaps == 0).sum())} exact ties")
    if n == 65536:      # config 4: the scores and the order only (the per-request walk inputs are the prompt lengths)
        out = {k: v for k, v in out.items() if k in ("seed", "ids_sha256", "ref_score", "a_order", "a_ran", "a_starv", "a_period",
                                                     "a_token_budget", "a_max_num_seqs")}
        out["a_ran"] = np.nonzero(out["a_ran"][0])[0].astype(np.int32)
        out["cu_sha256"] = np.frombuffer(hashlib.sha256(np.ascontiguousarray(cu.astype(np.int32)).tobytes()).digest(), np.uint8)
    path = os.path.join(GOLD, "config4_opt125m_65536.npz" if n == 65536 else
                        ("config3_opt350m_8192.npz" if full3 else "config2_opt125m_8192.npz"))
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {os.path.getsize(path) / 1024:.0f} KB")


def main_tpt():
    """Config 1's queue under `tpt`: class-mode predictor (82 labels), order (-score, request_id) on string ids."""
    mg._init_dist()
    torch.set_num_threads(os.cpu_count())
    spec = OPTSpec.opt_125m(82)
    ckpt = seeded_checkpoint(spec, 0)
    ids, cu, lens = synthetic_queue(spec, N_REQ, seed=0)
    z1 = np.load(os.path.join(GOLD, "config1_opt125m_256.npz"), allow_pickle=False)
    assert np.array_equal(z1["ids"], ids.astype(np.int32)) and np.array_equal(z1["cu_seqlens"], cu)   # the same queue
    out = dict(seed=np.int64(0), num_labels=np.int64(82))
    arrive_a = np.zeros(N_REQ, np.int32)
    arrive_a[64:] = 1 + (np.arange(N_REQ - 64) // 8)
    pred = RefPredictor(spec, ckpt)
    s_a, sgs_a, rec_a = run("a", "tpt-xxx", pred, ids, cu, arrive_a, 36, 2048, 256, out)
    scores = np.array([g.aux_model_score for g in sgs_a], np.float64)
    assert (scores == np.rint(scores)).all() and scores.min() >= 0 and scores.max() <= 81
    out["ref_score"] = scores.astype(np.float32)
    out["ref_top2_gap"] = np.array([pred.top2_gap[str(i)] for i in range(N_REQ)], np.float32)
    out["a_aux_calls"] = np.array([len(c) for c in pred.calls], np.int32)
    labels, counts = np.unique(scores, return_counts=True)
    print(f"tpt run a: {len(pred.calls)} predictor calls, {len(labels)} distinct labels over {N_REQ} requests (largest tie group "
          f"{counts.max()}), smallest top-2 logit gap {out['ref_top2_gap'].min():.3e}; first order {rec_a['order'][0][:10].tolist()}")
    path = os.path.join(GOLD, "config1_tpt_class82.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {os.path.getsize(path) / 1024:.0f} KB")


def main_xpt():
    """Config 1's queue and reference scores under `xpt{table}`: score -> expected length by table lookup on
    round(-score, 2), order by expected_length - output_len (scheduler.py:910-933) - an order that moves every step."""
    mg._init_dist()
    z1 = np.load(os.path.join(GOLD, "config1_opt125m_256.npz"), allow_pickle=False)
    ids, cu, ref = z1["ids"].astype(np.int64), z1["cu_seqlens"], z1["ref_score"]
    lo, hi = float((-ref).min()), float((-ref).max())
    key = np.round(np.linspace(lo - 0.02, hi + 0.02, 48), 2).tolist()                   # ascending thresholds on round(-score, 2)
    value = (24 + 3 * np.arange(48) + np.random.RandomState(3).randint(0, 9, 48)).tolist()   # expected output lengths: several per step count
    path_t = "/tmp/xpt_table_config1.pt"
    torch.save((key, value), path_t)
    table = {str(i): float(ref[i]) for i in range(N_REQ)}
    out = dict(seed=np.int64(0), xpt_key=np.asarray(key, np.float64), xpt_value=np.asarray(value, np.int64))
    arrive_a = np.zeros(N_REQ, np.int32)
    arrive_a[64:] = 1 + (np.arange(N_REQ - 64) // 8)
    s_a, sgs_a, rec_a = run("a", "xpt{" + path_t + "}-xxx", ScoreTable(table), ids, cu, arrive_a, 60, 2048, 256, out)
    exp = np.array([g.expected_length for g in sgs_a], np.int64)
    out["ref_expected_length"] = exp
    # how far every score is from a rounding boundary of round(-score, 2): scores closer than the predictor's error could
    # land in the neighbouring table row
    x = -ref.astype(np.float64) * 100.0
    out["ref_round_margin"] = (np.abs(x - np.floor(x) - 0.5) / 100.0).astype(np.float32)
    orders = np.stack(rec_a["order"])
    moved = sum(not np.array_equal(orders[i][orders[i] >= 0][:16], orders[i + 1][orders[i + 1] >= 0][:16]) for i in range(len(orders) - 1))
    print(f"xpt run a: {len(set(exp.tolist()))} distinct expected lengths, head of the order changes in {moved} of {len(orders) - 1} "
          f"steps, smallest rounding margin {out['ref_round_margin'].min():.3e}")
    path = os.path.join(GOLD, "config1_xpt.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {os.path.getsize(path) / 1024:.0f} KB")


def main_outlier(family="125m"):
    """OPT-125m (or, ``--config outlier350``, OPT-350m: post-LN blocks, project_in / project_out) shape, structured
    checkpoint, 64 (48) requests: reference fp32 scores + the reference Scheduler's cold order."""
    from vllm_ltr_amd.opt_spec import structured_checkpoint
    mg._init_dist()
    torch.set_num_threads(os.cpu_count())
    n = 64 if family == "125m" else 48
    spec = OPTSpec.opt_125m() if family == "125m" else OPTSpec.opt_350m()
    from vllm_ltr_amd.opt_spec import STRUCTURED_350M
    ckpt = structured_checkpoint(spec, 0) if family == "125m" else structured_checkpoint(spec, 0, **STRUCTURED_350M)
    rs = np.random.RandomState(11)
    lens = np.clip(np.rint(np.exp(rs.normal(np.log(64), 0.9, n))), 3, 600).astype(np.int64)
    lens[0], lens[1], lens[2], lens[3] = 1, 2, 1024, 700          # the shortest prompts, the benchmark's longest, a long one
    g = torch.Generator().manual_seed(11)
    T = int(lens.sum())
    ids = torch.randint(4, spec.vocab_size, (T,), generator=g, dtype=torch.int64).numpy()
    cu = np.zeros(n + 1, np.int32)
    np.cumsum(lens, out=cu[1:])
    ids[cu[:-1]] = 2
    out = dict(ids=ids.astype(np.int32), cu_seqlens=cu, seed=np.int64(0))
    pred = RefPredictor(spec, ckpt)
    arrive = np.zeros(n, np.int32)
    s_a, sgs_a, rec_a = run("a", "opt-xxx", pred, ids, cu, arrive, 1, 2048, 256, out)
    scores = np.array([g_.aux_model_score for g_ in sgs_a], np.float64)
    assert np.isfinite(scores).all()
    out["ref_score"] = scores.astype(np.float32)
    assert np.array_equal(out["ref_score"].astype(np.float64), scores)
    keep = {k: v for k, v in out.items() if k in ("ids", "cu_seqlens", "seed", "ref_score", "a_order", "a_concat")}
    # the size of the residual stream the reference saw (hidden state before the final LayerNorm, first and last request)
    print(f"outlier: T = {T}, score range [{scores.min():.4f}, {scores.max():.4f}], smallest gap between sorted scores "
          f"{np.diff(np.sort(scores)).min():.3e}; {pred.seconds:.1f} s in the reference predictor; cold order head "
          f"{rec_a['order'][0][:8].tolist()}")
    path = os.path.join(GOLD, "outlier_opt125m_64.npz" if family == "125m" else "outlier_opt350m_48.npz")
    np.savez_compressed(path, **keep)
    print(f"wrote {path}: {os.path.getsize(path) / 1024:.0f} KB")


def main():
    if "--config" in sys.argv and sys.argv[sys.argv.index("--config") + 1] == "outlier":
        return main_outlier()
    if "--config" in sys.argv and sys.argv[sys.argv.index("--config") + 1] == "outlier350":
        return main_outlier("350m")
    if "--config" in sys.argv and sys.argv[sys.argv.index("--config") + 1] == "xpt":
        return main_xpt()
    if "--config" in sys.argv and sys.argv[sys.argv.index("--config") + 1] == "3":
        return main_config3()
    if "--config" in sys.argv and sys.argv[sys.argv.index("--config") + 1] in ("2", "3full", "4full"):
        return main_config2(sys.argv[sys.argv.index("--config") + 1])
    if "--config" in sys.argv and sys.argv[sys.argv.index("--config") + 1] == "tpt":
        return main_tpt()
    mg._init_dist()
    torch.set_num_threads(os.cpu_count())
    spec = OPTSpec.opt_125m()
    ckpt = seeded_checkpoint(spec, 0)
    ids, cu, lens = synthetic_queue(spec, N_REQ, seed=0)
    assert int(cu[-1]) == 23078, int(cu[-1])                   # BASELINE.md section 3 / SURVEY 8d
    out = dict(ids=ids.astype(np.int32), cu_seqlens=cu, seed=np.int64(0))

    # ---- run a: config 1, every arrival batch scored by the reference's own predictor
    arrive_a = np.zeros(N_REQ, np.int32)
    arrive_a[64:] = 1 + (np.arange(N_REQ - 64) // 8)           # 64 at step 0, then 8 per step (steps 1..24)
    pred = RefPredictor(spec, ckpt)
    t0 = time.time()
    s_a, sgs_a, rec_a = run("a", "opt-xxx-starv200-period10", pred, ids, cu, arrive_a, 36, 2048, 256, out)
    scores = np.array([g.aux_model_score for g in sgs_a], np.float64)
    assert np.isfinite(scores).all() and len(pred.calls) == 25 and sum(map(len, pred.calls)) == N_REQ
    out["ref_score"] = scores.astype(np.float32)
    assert np.array_equal(out["ref_score"].astype(np.float64), scores)      # fp32 values widened by .tolist()
    out["a_aux_calls"] = np.array([len(c) for c in pred.calls], np.int32)
    print(f"run a: {len(pred.calls)} predictor calls, {pred.seconds:.1f} s in the reference predictor "
          f"({N_REQ / pred.seconds:.1f} req/s on {os.cpu_count()} threads), total {time.time()-t0:.1f} s; "
          f"score range [{scores.min():.4f}, {scores.max():.4f}], smallest gap between sorted scores "
          f"{np.diff(np.sort(scores)).min():.3e}")

    # ---- run b: the same scores, starvation control biting (tight budget, early promotions)
    table = {str(i): float(out["ref_score"][i]) for i in range(N_REQ)}
    arrive_b = np.sort(np.random.RandomState(5).randint(0, 20, N_REQ)).astype(np.int32)
    run("b", "opt-xxx-starv6-period2", ScoreTable(table), ids, cu, arrive_b, 48, 512, 24, out)

    # ---- run c: run a through the product's install() wiring on the same real Scheduler
    log = []
    surface = {}

    def install(s):
        for name in ("_general_schedule", "_schedule", "_update_priority", "_get_ordered_requests", "waiting", "running",
                     "swapped", "need_score", "starv", "period"):
            surface[name] = type(getattr(s, name)).__name__               # all exist on the real object BEFORE install
        assert not hasattr(s, "aux_model")                     # a plain attribute the engine assigns after construction (llm_engine.py:228-242)
        rk = recording_ranker("opt-xxx-starv200-period10", ScoreTable(table), log)
        rk.install(s)
        assert s.aux_model is rk and s._schedule.__wrapped__ == s._general_schedule
    s_c, sgs_c, rec_c = run("c", "opt-xxx-starv200-period10", None, ids, cu, arrive_a, 36, 2048, 256, out, install=install)
    o = s_c._schedule()                                        # the wrapper's view of the step outputs (plugin.py install)
    surface["SchedulerOutputs.scheduled_seq_groups"] = type(o.scheduled_seq_groups).__name__
    if o.scheduled_seq_groups:
        surface["scheduled_seq_groups[i].seq_group"] = type(o.scheduled_seq_groups[0].seq_group).__name__
    for k in ("order", "concat", "ran", "states", "granted"):
        assert np.array_equal(np.stack(rec_c[k]), np.stack(rec_a[k])), k      # the wiring changes nothing
    # one order + one age per step, an obtain_aux_scores exactly in the steps with arrivals
    kinds = [e[0] for e in log]
    assert kinds.count("order") == 37 and kinds.count("age") == 37 and kinds.count("obtain_aux_scores") == 25
    for k in [k for k in out if k.startswith("c_")]:
        del out[k]                                             # identical to run a: not stored twice
    out["surface"] = np.array(json.dumps(dict(attributes=surface, calls=log[:12])))
    print("install() surface on the real Scheduler:", surface)
    path = os.path.join(GOLD, "config1_opt125m_256.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {os.path.getsize(path) / 1024:.0f} KB")


if __name__ == "__main__":
    main()
