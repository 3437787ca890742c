#!/usr/bin/env python3
"""Generate tests/golden/* by running the REFERENCE ITSELF (imported read-only
from /root/reference) and HF transformers on seeded inputs.

Runs only in the build container (the GPU box has no /root/reference):

    PYTHONPATH=/root/reference PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden.py [--big]

Nothing of the reference is copied: the fixtures hold inputs and the outputs the
reference computed (scores, orders, counters).  Checkpoints are NOT stored; they
are regenerated from ``vllm_ltr_amd.opt_spec.seeded_checkpoint(spec, seed)``.

Import shims: a stub ``cpuinfo`` module
(vllm/usage/usage_lib.py:13), ``selector.is_cpu -> True`` so ``Attention`` picks
the TorchSDPA backend (vllm/attention/selector.py:50-60), and a 1-rank gloo
process group (vllm/distributed/parallel_state.py:52-77).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")

sys.modules["cpuinfo"] = types.ModuleType("cpuinfo")
import torch  # noqa: E402
import vllm  # noqa: E402  (the reference, via PYTHONPATH)
import vllm.attention.selector as _sel  # noqa: E402

_sel.is_cpu = lambda: True
from vllm.distributed import (ensure_model_parallel_initialized,  # noqa: E402
                              init_distributed_environment)

from vllm_ltr_amd.opt_spec import OPTSpec, seeded_checkpoint  # noqa: E402

assert vllm.__file__.startswith("/root/reference"), vllm.__file__


def _init_dist():
    init_distributed_environment(1, 0, "tcp://127.0.0.1:29577", 0, backend="gloo")
    ensure_model_parallel_initialized(1, 1)


# --------------------------------------------------------------------------------
# predictor scores
# --------------------------------------------------------------------------------
def make_inputs(spec: OPTSpec, lens, seed):
    rs = np.random.RandomState(seed)
    ids = []
    for L in lens:
        row = rs.randint(4, spec.vocab_size, size=L)
        row[0] = 2                                   # BOS first, like the OPT tokenizer
        ids.append(row)
    ids = np.concatenate(ids).astype(np.int64)
    cu = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    return ids, cu


def ref_scores(spec: OPTSpec, ckpt, ids, cu):
    """Drive vllm.model_executor.models.opt.OPTForSequenceClassification exactly as
    ModelRunner.execute_model does for a prefill batch (model_runner.py:827-877)."""
    from transformers import OPTConfig
    from vllm.attention.backends.torch_sdpa import TorchSDPAMetadata
    from vllm.model_executor.models.opt import OPTForSequenceClassification as REF
    from vllm.model_executor.sampling_metadata import SamplingMetadata
    cfg = OPTConfig(**spec.to_hf_config_kwargs())
    torch.manual_seed(0)
    ref = REF(cfg).eval().float()
    ref.load_weights([(k, torch.from_numpy(v.astype(np.float32))) for k, v in ckpt.items()])
    lens = np.diff(cu).tolist()
    N, T = len(lens), int(cu[-1])
    pos = torch.cat([torch.arange(L) for L in lens]).long()
    sel = torch.as_tensor(cu[1:].astype(np.int64) - 1)
    md = TorchSDPAMetadata(context_lens=None, max_context_len=None, block_tables=torch.tensor([]),
                           num_prefills=N, num_prefill_tokens=T, num_decode_tokens=0,
                           prefill_metadata=object(), decode_metadata=None,
                           slot_mapping=torch.zeros(T, dtype=torch.long), kv_cache_dtype="auto",
                           need_score=False, selected_token_indices=sel, is_prompt=True,
                           prompt_lens=lens)
    sm = SamplingMetadata(seq_groups=[], seq_data={}, prompt_lens=lens, selected_token_indices=sel,
                          categorized_sample_indices=None, generators=None, perform_sampling=False)
    with torch.no_grad():
        hs, _ = ref(torch.from_numpy(ids), pos, [None] * cfg.num_hidden_layers, md)
        logits = ref.compute_logits(hs, sm)           # opt.py:389-397 (argmax'd in class mode)
        last_hidden = hs.index_select(0, sel)
    return logits[:, 0].float().numpy(), last_hidden.float().numpy()


def hf_logits(spec: OPTSpec, ckpt, ids, cu):
    """HF OPTForSequenceClassification on the right-padded batch - the model that
    prefill_predictor.py:25,50-53 wraps (the 'CPU reference predictor')."""
    from transformers import OPTConfig, OPTForSequenceClassification as HF
    cfg = OPTConfig(**spec.to_hf_config_kwargs())
    cfg.pad_token_id = 1
    hf = HF(cfg).eval().float()
    sd = {k: torch.from_numpy(v.astype(np.float32)) for k, v in ckpt.items()}
    missing, unexpected = hf.load_state_dict(sd, strict=False)
    missing = [m for m in missing if "lm_head" not in m]
    assert not missing and not unexpected, (missing, unexpected)
    lens = np.diff(cu).tolist()
    out = []
    with torch.no_grad():
        for i, L in enumerate(lens):                  # one by one: no padding effects at all
            x = torch.from_numpy(ids[cu[i]:cu[i + 1]])[None]
            out.append(hf(input_ids=x, attention_mask=torch.ones_like(x)).logits[0])
    return torch.stack(out).float().numpy()


def score_fixture(name, spec, lens, seed):
    t0 = time.time()
    ckpt = seeded_checkpoint(spec, seed)
    ids, cu = make_inputs(spec, lens, seed + 1)
    ref, last_hidden = ref_scores(spec, ckpt, ids, cu)
    hfl = hf_logits(spec, ckpt, ids, cu)
    if spec.num_labels == 1:
        d = float(np.abs(hfl[:, 0] - ref).max())
    else:
        # class mode: the two implementations may pick different labels only where the two largest logits are closer than
        # their own f32 noise (with hundreds of labels such near-ties do occur); everywhere else the argmax must agree
        # (the reference's LogitsProcessor also cuts the logits at vocab_size columns, logits_processor.py:68-70, before
        # opt.py:395 takes the argmax: with fewer vocabulary entries than labels only the first vocab_size labels compete)
        hfe = hfl[:, :min(spec.num_labels, spec.vocab_size)]
        top2 = np.sort(hfe, axis=-1)[:, -2:]
        clear = (top2[:, 1] - top2[:, 0]) > 1e-5
        print(f"{name}: {int((~clear).sum())} of {len(lens)} requests have a top-2 gap <= 1e-5")
        d = float(np.abs(hfe.argmax(-1).astype(np.float32) - ref)[clear].max()) if clear.any() else 0.0
    print(f"{name}: N={len(lens)} T={int(cu[-1])} ref-vs-HF max|d|={d:.3e}  ({time.time()-t0:.1f}s)")
    assert d < 5e-5, d
    np.savez_compressed(os.path.join(GOLD, f"score_{name}.npz"),
                        spec=np.array(list(spec.to_hf_config_kwargs().items()), dtype=object).astype(str),
                        seed=np.int64(seed), ids=ids, cu_seqlens=cu, ref_score=ref.astype(np.float32),
                        hf_logits=hfl.astype(np.float32),
                        ref_last_hidden=last_hidden[:, :min(64, last_hidden.shape[1])].astype(np.float32))


# --------------------------------------------------------------------------------
# scheduler ranking
# --------------------------------------------------------------------------------
class _StubAux:
    """Stands where llm_engine.py:228-242 puts the AUXLLM: sets a given score."""

    def __init__(self, table):
        self.table = table
        self.calls = []

    def obtain_aux_scores(self, sgs):
        self.calls.append([sg.request_id for sg in sgs])
        for sg in sgs:
            assert sg.need_aux_model_score()
            sg.set_aux_model_score(self.table[sg.request_id])


def _mk_scheduler(schedule_type, max_tokens, max_seqs, blocks=4096, block_size=16, cpu_blocks=None):
    from vllm.config import CacheConfig, SchedulerConfig
    from vllm.core.scheduler import Scheduler
    sc = SchedulerConfig(max_tokens, max_seqs, 2048, enable_chunked_prefill=True,
                         schedule_type=schedule_type)
    cc = CacheConfig(block_size, 1.0, 1, "auto")
    cc.num_gpu_blocks = blocks
    cc.num_cpu_blocks = blocks if cpu_blocks is None else cpu_blocks
    return Scheduler(sc, cc, None)


def _mk_sg(rid: str, plen: int, block_size=16, best_of: int = 1):
    from vllm import SamplingParams
    from vllm.sequence import Sequence, SequenceGroup
    seq = Sequence(int(rid), "p", list(range(plen)), block_size)
    return SequenceGroup(rid, [seq], SamplingParams(n=best_of, best_of=best_of, max_tokens=10**6, ignore_eos=True),
                         time.time())


def order_fixture():
    """Standalone calls of the reference's _get_{opt,tpt,rtpt,ropt}_ordered_requests
    (scheduler.py:936-1016) on engineered score vectors, with requests spread over
    waiting / running / swapped and pre-set (pri, idle, runs)."""
    rs = np.random.RandomState(7)
    cases = []

    def scores_engineered(n):
        s = rs.standard_normal(n).astype(np.float32)
        s = s.astype(np.float16).astype(np.float32)           # fp16-valued like GPU scores
        s[rs.randint(0, n, n // 4)] = s[rs.randint(0, n, n // 4)]   # many exact ties
        s[rs.randint(0, n, max(1, n // 16))] = 0.0
        s[rs.randint(0, n, max(1, n // 16))] = -0.0
        return s

    for ci, (n, starv, period) in enumerate([(5, -1, 0), (64, -1, 0), (64, 3, 2), (257, 5, 3),
                                             (1000, 200, 10), (33, 0, 1), (128, 1, 1)]):
        st = "opt-xxx" + (f"-starv{starv}-period{period}" if starv != -1 else "")
        s = _mk_scheduler(st, 4096, 256)
        sc = scores_engineered(n)
        if ci == 0:
            sc = np.array([0.5, 0.5, -0.0, 0.0, 1.0], np.float32)
        ids = [str(i) for i in range(n)]
        sgs = [_mk_sg(r, 4) for r in ids]
        table = {r: float(x) for r, x in zip(ids, sc)}
        s.aux_model = _StubAux(table)
        # queue membership: first chunk waiting (unscored), then running, then swapped (scored)
        n_w = n - 2 * (n // 3)
        where = np.array([0] * n_w + [1] * (n // 3) + [2] * (n // 3))
        pri0 = np.zeros(n, np.int32); idle0 = np.zeros(n, np.int32); runs0 = np.zeros(n, np.int32)
        if starv != -1:
            pri0 = -(rs.rand(n) < 0.3).astype(np.int32)
            idle0 = rs.randint(0, max(2 * starv, 2), n).astype(np.int32)
            runs0 = rs.randint(-1, period + 1, n).astype(np.int32)
        for i, sg in enumerate(sgs):
            s.add_seq_group(sg)                                  # sets idle=runs=pri=0, :372-374
        for i, sg in enumerate(sgs):
            sg.pri, sg.idle, sg.runs = int(pri0[i]), int(idle0[i]), int(runs0[i])
            if where[i] != 0:
                s.waiting.remove(sg)
                sg.set_aux_model_score(table[sg.request_id])
                (s.running if where[i] == 1 else s.swapped).append(sg)
        order = [sg.request_id for sg in s._get_opt_ordered_requests()]
        post = np.array([[sg.pri, sg.idle, sg.runs] for sg in sgs], np.int32)
        extra = {}
        if starv == -1:
            extra["tpt"] = np.array([int(g.request_id) for g in s._get_tpt_ordered_requests()], np.int32)
            extra["rtpt"] = np.array([int(g.request_id) for g in s._get_rtpt_ordered_requests()], np.int32)
            extra["ropt"] = np.array([int(g.request_id) for g in s._get_ropt_ordered_requests()], np.int32)
        cases.append(dict(score=sc, where=where.astype(np.int8), starv=starv, period=period,
                          pri0=pri0, idle0=idle0, runs0=runs0,
                          order=np.array([int(r) for r in order], np.int32), post=post,
                          aux_calls=len(s.aux_model.calls), **extra))
        print(f"order case {ci}: n={n} starv={starv} period={period} head={order[:6]}")
    flat = {}
    for ci, c in enumerate(cases):
        for k, v in c.items():
            flat[f"c{ci}_{k}"] = np.asarray(v)
    flat["n_cases"] = np.int64(len(cases))
    np.savez_compressed(os.path.join(GOLD, "rank_order.npz"), **flat)


def steps_fixture():
    """Multi-step runs of the reference's full Scheduler.schedule() (= _general_schedule,
    scheduler.py:1101-1373) with starvation control; per step we record the concatenation
    list(waiting)+list(running)+list(swapped) the order is computed from (scheduler.py:985), the
    order returned by _get_ordered_requests, what the budget walk sees per request (un-chunked new
    tokens, new sequences, whether _get_num_new_tokens may chunk it: exactly one sequence in the
    walked status, :1884), which requests ran with the tokens granted, and (pri, idle, runs) after
    the aging loop (:1358-1365).  The last case holds best_of = 2 requests: a WAITING prompt has ONE
    sequence but get_max_num_running_seqs() = 2 (sequence.py:500-504), so it is chunked although
    new_seqs > 1; after its prefill the sequence is forked (what the engine's output processor does),
    so its decode steps carry two RUNNING sequences and are not chunked."""
    from vllm.sequence import Logprob, SequenceStatus
    runs_out = {}
    cases = [(10, 3, 2, 4, 64, 12, (4, 4), 0.0), (48, 4, 2, 6, 96, 24, (8, 8), 0.0), (200, 6, 3, 16, 256, 30, (5, 5), 0.0),
             (36, 5, 2, 8, 64, 40, (20, 150), 0.5)]
    for fi, (n, starv, period, max_seqs, max_tokens, steps, (plo, phi), frac_bo2) in enumerate(cases):
        rs = np.random.RandomState(100 + fi)
        st = f"opt-xxx-starv{starv}-period{period}"
        s = _mk_scheduler(st, max_tokens, max_seqs)
        sc = (np.arange(n) % 5).astype(np.float32) if fi == 0 else \
            rs.standard_normal(n).astype(np.float16).astype(np.float32)
        ids = [str(i) for i in range(n)]
        table = {r: float(x) for r, x in zip(ids, sc)}
        s.aux_model = _StubAux(table)
        plens = rs.randint(plo, phi + 1, n)
        bo = np.where(rs.rand(n) < frac_bo2, 2, 1)
        sgs = [_mk_sg(r, int(plens[i]), best_of=int(bo[i])) for i, r in enumerate(ids)]
        arrive_at = np.zeros(n, np.int32) if fi == 0 else np.sort(rs.randint(0, steps // 2, n)).astype(np.int32)
        orders, rans, states, present, needs, nseqs, grants, concats, chunkables = [], [], [], [], [], [], [], [], []
        captured = {}
        inner = s._get_ordered_requests
        next_seq_id = [10**6]

        def spy():
            captured["concat"] = [g.request_id for g in list(s.waiting) + list(s.running) + list(s.swapped)]
            o = inner()
            captured["order"] = [g.request_id for g in o]
            return o
        s._get_ordered_requests = spy
        for step in range(steps):
            for i in np.nonzero(arrive_at == step)[0]:
                s.add_seq_group(sgs[i])
            # what the budget walk will see: un-chunked new tokens / new sequences per queued request
            # (_get_num_new_tokens before the min() with the remaining budget, scheduler.py:1878-1881)
            nd = np.zeros(n, np.int32); nq = np.zeros(n, np.int32); ck = np.zeros(n, np.uint8)
            for dq, status in ((s.waiting, SequenceStatus.WAITING), (s.running, SequenceStatus.RUNNING),
                               (s.swapped, SequenceStatus.SWAPPED)):
                for g in dq:
                    seqs = g.get_seqs(status=status)
                    nd[int(g.request_id)] = sum(q.get_num_new_tokens() for q in seqs)
                    nq[int(g.request_id)] = g.get_max_num_running_seqs()
                    ck[int(g.request_id)] = len(seqs) == 1
            needs.append(nd); nseqs.append(nq); chunkables.append(ck)
            metas, out = s.schedule()
            ran = [x.seq_group.request_id for x in out.scheduled_seq_groups]
            gr = np.zeros(n, np.int32)
            for x, meta in zip(out.scheduled_seq_groups, metas):
                # tokens charged to the budget for this group (the walk's num_new_tokens, scheduler.py:1159);
                # = token_chunk_size x running sequences
                gr[int(x.seq_group.request_id)] = meta.token_chunk_size * len(x.seq_group.get_seqs(status=SequenceStatus.RUNNING))
            grants.append(gr)
            for x, meta in zip(out.scheduled_seq_groups, metas):
                g = x.seq_group
                was_prefill = g.is_prefill()
                g.update_num_computed_tokens(meta.token_chunk_size)
                if not g.is_prefill():
                    for seq in g.get_seqs(status=SequenceStatus.RUNNING):
                        seq.append_token_id(1, {1: Logprob(0.0)})
                    if was_prefill and g.sampling_params.best_of > g.num_seqs():
                        # the engine forks the prompt sequence into best_of samples after the prefill
                        # (llm_engine.py _process_sequence_group_outputs -> scheduler.fork_seq)
                        parent = g.get_seqs(status=SequenceStatus.RUNNING)[0]
                        child = parent.fork(next_seq_id[0]); next_seq_id[0] += 1
                        g.add(child)
                        s.fork_seq(parent, child)
            alive = [g.request_id for g in list(s.waiting) + list(s.running) + list(s.swapped)]
            o = np.full(n, -1, np.int32); o[:len(captured["order"])] = [int(r) for r in captured["order"]]
            orders.append(o)
            c = np.full(n, -1, np.int32); c[:len(captured["concat"])] = [int(r) for r in captured["concat"]]
            concats.append(c)
            r = np.zeros(n, np.uint8); r[[int(x) for x in ran]] = 1
            rans.append(r)
            p = np.zeros(n, np.uint8); p[[int(x) for x in alive]] = 1
            present.append(p)
            stt = np.zeros((n, 3), np.int32)
            for g in sgs:
                if hasattr(g, "pri"):
                    stt[int(g.request_id)] = (g.pri, g.idle, g.runs)
            states.append(stt)
        nch = int(sum(((np.stack(grants)[t] > 0) & (np.stack(grants)[t] < np.stack(needs)[t])).sum() for t in range(steps)))
        nbo = int(sum(((np.stack(nseqs)[t] > 1) & (np.stack(grants)[t] > 0)).sum() for t in range(steps)))
        print(f"steps case {fi}: n={n} first orders {orders[0][:6]} ... step3 {orders[min(3, steps-1)][:8]}; "
              f"chunked grants {nch}, grants to new_seqs>1 groups {nbo}")
        runs_out.update({f"f{fi}_score": sc, f"f{fi}_starv": np.int64(starv), f"f{fi}_period": np.int64(period),
                         f"f{fi}_arrive_at": arrive_at, f"f{fi}_orders": np.stack(orders), f"f{fi}_concat": np.stack(concats),
                         f"f{fi}_ran": np.stack(rans), f"f{fi}_present": np.stack(present),
                         f"f{fi}_states": np.stack(states), f"f{fi}_need_tokens": np.stack(needs),
                         f"f{fi}_need_seqs": np.stack(nseqs), f"f{fi}_chunkable": np.stack(chunkables),
                         f"f{fi}_granted": np.stack(grants), f"f{fi}_token_budget": np.int64(max_tokens),
                         f"f{fi}_max_num_seqs": np.int64(max_seqs)})
    runs_out["n_cases"] = np.int64(len(cases))
    np.savez_compressed(os.path.join(GOLD, "rank_steps.npz"), **runs_out)


def reserve_fixture():
    """Calls of the reference's Scheduler.reserve_free_blocks (scheduler.py:1376-1452) recorded
    inside full schedule() runs under KV-block pressure: inputs as the vectorised form sees them
    (rank order, selected prefix, per-request block counts / states, blocks to free) and the
    requests it evicted (unselected running requests from the low-priority end, then selected
    requests put back from the end of the selection)."""
    from vllm.sequence import Logprob, SequenceStatus
    out = {}
    ncase = 0
    for fi, (n, blocks, max_seqs, max_tokens, steps, lo, hi) in enumerate(
            [(40, 40, 8, 128, 40, 20, 90), (120, 96, 16, 256, 60, 10, 140), (64, 24, 6, 96, 50, 30, 60)]):
        rs = np.random.RandomState(300 + fi)
        s = _mk_scheduler("opt-xxx-starv6-period3", max_tokens, max_seqs, blocks=blocks, cpu_blocks=8192)
        sc = rs.standard_normal(n).astype(np.float16).astype(np.float32)
        ids = [str(i) for i in range(n)]
        s.aux_model = _StubAux({r: float(x) for r, x in zip(ids, sc)})
        sgs = [_mk_sg(r, int(rs.randint(lo, hi))) for r in ids]
        arrive_at = np.sort(rs.randint(0, steps // 2, n))
        calls = []
        inner = s.reserve_free_blocks

        def spy(num_blocks_needed, pinned, priority, remaining_running, final_budget):
            bm = s.block_manager
            order = [g.request_id for g in pinned] + [g.request_id for g in priority]
            rec = dict(perm=np.array([int(r) for r in order], np.int32), n_selected=len(pinned),
                       need=int(num_blocks_needed - bm.gpu_allocator.get_num_free_blocks() + bm.watermark_blocks),
                       state=np.zeros(n, np.uint8), phys=np.zeros(n, np.int32), logical=np.zeros(n, np.int32),
                       nrun=np.zeros(n, np.int32), nswap=np.zeros(n, np.int32))
            for g in list(pinned) + list(priority):
                i = int(g.request_id)
                nr, nsw = g.num_seqs(status=SequenceStatus.RUNNING), g.num_seqs(status=SequenceStatus.SWAPPED)
                rec["state"][i] = 1 if nr else (2 if nsw else 0)
                rec["nrun"][i], rec["nswap"][i] = nr, nsw
                rec["logical"][i] = len(g.get_seqs()[0].logical_token_blocks)
                if nr or nsw:
                    rec["phys"][i] = len(bm._get_physical_blocks(g))
            res = inner(num_blocks_needed, pinned, priority, remaining_running, final_budget)
            _, exe, preempted, swapped_out = res[0], res[1], res[2], res[3]
            act = np.zeros(n, np.uint8)
            pinned_ids = {g.request_id for g in pinned}
            exe_ids = {g.request_id for g in exe}
            for g in list(preempted) + list(swapped_out):
                act[int(g.request_id)] = 2 if g.request_id in pinned_ids else 1
            for r in pinned_ids - exe_ids:
                if act[int(r)] == 0:
                    act[int(r)] = 3
            rec["action"] = act
            rec["n_exec"] = len(exe)
            calls.append(rec)
            return res
        s.reserve_free_blocks = spy
        for step in range(steps):
            for i in np.nonzero(arrive_at == step)[0]:
                s.add_seq_group(sgs[i])
            metas, o = s.schedule()
            for x, meta in zip(o.scheduled_seq_groups, metas):
                x.seq_group.update_num_computed_tokens(meta.token_chunk_size)
                if not x.seq_group.is_prefill():
                    for seq in x.seq_group.get_seqs():
                        seq.append_token_id(1, {1: Logprob(0.0)})
        pressured = [c for c in calls if c["need"] > 0]
        keep = pressured[:24] + [c for c in calls if c["need"] <= 0][:4]
        print(f"reserve case {fi}: {len(calls)} calls, {len(pressured)} under pressure, "
              f"actions {[int((c['action'] == k).sum()) for k in (1, 2, 3) for c in pressured[:1]]}, "
              f"total evictions {sum(int((c['action'] > 0).sum()) for c in pressured)}")
        for c in keep:
            for k, v in c.items():
                out[f"c{ncase}_{k}"] = np.asarray(v)
            out[f"c{ncase}_n"] = np.int64(n)
            ncase += 1
    out["n_calls"] = np.int64(ncase)
    np.savez_compressed(os.path.join(GOLD, "reserve_calls.npz"), **out)


def listmle_fixture():
    """The reference's listMLE (train/allrank/models/losses/listMLE.py) and its autograd gradient on seeded
    slates, with torch.randperm replaced by a recorded permutation."""
    # Load the reference's listMLE.py itself (by path).  Its package __init__ chain pulls torchvision / gcsfs,
    # absent here, for code the loss never touches, so the two constants it imports are provided through
    # stand-in parent modules with the values of allrank/models/losses/__init__.py:17 and
    # allrank/data/dataset_loading.py:31.
    import importlib.util
    for name in ("allrank", "allrank.data", "allrank.data.dataset_loading", "allrank.models", "allrank.models.losses"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["allrank.data.dataset_loading"].PADDED_Y_VALUE = -1
    sys.modules["allrank.models.losses"].DEFAULT_EPS = 1e-10
    spec_ = importlib.util.spec_from_file_location("ref_listMLE", "/root/reference/train/allrank/models/losses/listMLE.py")
    mod = importlib.util.module_from_spec(spec_)
    spec_.loader.exec_module(mod)
    out = {}
    # labels are distinct inside a slate: with ties the reference's result depends on the tie order of
    # torch.sort (unstable; implementation specific), which is why it shuffles first
    cases = [("single_slate", 1, 24, False, False, 2.0), ("pad3", 3, 40, False, True, 2.0), ("long", 2, 300, False, False, 2.0),
             ("all_pad_tail", 2, 17, False, True, 2.0), ("batch64", 64, 32, False, True, 2.0),
             ("wide_spread", 4, 48, False, True, 14.0)]
    for ci, (name, B, S, ties, pad, scale) in enumerate(cases):
        rs = np.random.RandomState(1000 + 7 * ci)      # fixed per case: the fixture regenerates bit-identically
        pred = rs.standard_normal((B, S)).astype(np.float32) * scale
        true = (rs.randint(0, 6, (B, S)) if ties else np.stack([rs.permutation(S) for _ in range(B)])).astype(np.float32)
        if pad:
            for b in range(B):
                true[b, rs.randint(S // 2, S):] = -1
        perm = rs.permutation(S)
        real = torch.randperm
        torch.randperm = lambda n, *a, **k: torch.from_numpy(perm.copy())
        try:
            yp = torch.tensor(pred, requires_grad=True)
            loss = mod.listMLE(yp, torch.tensor(true))
            loss.backward()
        finally:
            torch.randperm = real
        out[f"{name}_pred"], out[f"{name}_true"], out[f"{name}_perm"] = pred, true, perm.astype(np.int32)
        out[f"{name}_loss"], out[f"{name}_grad"] = np.float32(loss.item()), yp.grad.numpy()
        print(f"listMLE {name}: B={B} S={S} loss={loss.item():.6f}")
    out["names"] = np.array([c[0] for c in cases])
    np.savez_compressed(os.path.join(GOLD, "listmle.npz"), **out)


def config_fixture():
    """Round-trip the shipped predictor configs through the reference's
    PrefillPredictorConfig.from_json (config_predictor.py:136-147) and record the
    parsed fields; also the starv/period parse of scheduler.py:269-275."""
    import json
    from vllm.config_predictor import PrefillPredictorConfig
    out = {}
    cfgdir = "/root/reference/train/configs"
    for fn in sorted(os.listdir(cfgdir)):
        c = PrefillPredictorConfig.from_json(os.path.join(cfgdir, fn))
        with open(os.path.join(cfgdir, fn)) as f:
            raw = json.load(f)
        out[fn] = dict(input=raw, parsed=dict(c.model.__dict__))
    sts = {}
    for st in ["opt-xxx-starv200-period10", "opt-starv3-period2", "opt-125m-sharegpt-starv256-period32",
               "opt", "opt-class-starv0-period1"]:
        s = _mk_scheduler(st, 64, 4)
        sts[st] = dict(starv=s.starv, period=getattr(s, "period", 0), need_score=s.need_score)
    with open(os.path.join(GOLD, "config_cases.json"), "w") as f:
        json.dump(dict(predictor_configs=out, schedule_types=sts), f, indent=1, sort_keys=True)
    print("config cases:", list(out), sts)


def ltr_head_fixture():
    """The reference's own predictor_model (vllm/model_executor/predictor.py:128-145) with seeded
    weights loaded through load_state_dict, scored on random hidden states."""
    import copy
    from vllm.model_executor.predictor import predictor_model
    from oracle.ltr_head import seeded_head_weights
    cases = [
        ("small_relu", 256, dict(sizes=[128, 64], input_norm=True, activation="ReLU", dropout=0.1), dict(d_output=1, output_activation=None), 37),
        ("wide_tanh_sum", 4096, dict(sizes=[1024], input_norm=True, activation="Tanh", dropout=None), dict(d_output=4, output_activation="Sigmoid"), 41),
        ("no_fc", 512, None, dict(d_output=1, output_activation=None), 43),
        ("gelu_nonorm", 768, dict(sizes=[96, 200], input_norm=False, activation="GELU", dropout=None), dict(d_output=3, output_activation="Tanh"), 47),
    ]
    for name, nf, fc, post, seed in cases:
        sd = seeded_head_weights(nf, fc["sizes"] if fc else None, bool(fc and fc["input_norm"]), post["d_output"], seed)
        m = predictor_model(fc_model=copy.deepcopy(fc), post_model=dict(post), n_features=nf, pred_layer_idx=31).eval().float()
        m.load_state_dict({k: torch.from_numpy(v.astype(np.float32)) for k, v in sd.items()})
        x = (np.random.RandomState(seed + 1).standard_normal((40, nf)) * 1.5).astype(np.float32)
        with torch.no_grad():
            y = m.score(torch.from_numpy(x))
        y = y.reshape(40, -1)[:, 0].float().numpy() if post["d_output"] == 1 else y.float().numpy()
        np.savez_compressed(os.path.join(GOLD, f"ltr_head_{name}.npz"), x=x, score=y.astype(np.float32),
                            n_features=np.int64(nf), seed=np.int64(seed),
                            cfg=np.array(json.dumps(dict(fc_model=fc, post_model=post))))
        print(f"ltr_head {name}: n_features={nf} score[:3]={y[:3]}")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--big", action="store_true", help="also the true-shape 125m / 350m cases")
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    os.makedirs(GOLD, exist_ok=True)
    _init_dist()
    torch.set_num_threads(os.cpu_count())
    if args.only in ("", "config"):
        config_fixture()
    if args.only in ("", "order"):
        order_fixture()
    if args.only in ("", "steps"):
        steps_fixture()
    if args.only in ("", "head"):
        ltr_head_fixture()
    if args.only in ("", "reserve"):
        reserve_fixture()
    if args.only in ("", "listmle"):
        listmle_fixture()
    if args.only in ("", "score"):
        edge = [1, 2, 4, 5, 63, 64, 65, 100, 3, 128, 17, 1, 31, 32, 33, 150]
        score_fixture("tiny_pre_ln", OPTSpec.tiny_pre_ln(), edge, 11)
        score_fixture("tiny_post_ln", OPTSpec.tiny_post_ln(), edge, 12)
        score_fixture("tiny_pre_ln_class10", OPTSpec.tiny_pre_ln(10), edge, 13)
        score_fixture("tiny_post_ln_class7", OPTSpec.tiny_post_ln(7), edge, 14)
    if args.only in ("", "score", "class"):
        # class-mode heads at the reference's bucket counts (train/train.sh:19-44: buckets 100 / 10 / 1 over 8192 ->
        # 82 / 820 / 8192 labels; benchmarks/*.sh tpt-class82/820/8192-xxx): argmax over num_labels (opt.py:394-395)
        edge2 = [1, 2, 5, 63, 64, 65, 100, 3, 128, 17, 31, 33, 150, 7, 90, 44, 12, 77, 140, 9]
        score_fixture("tiny_pre_ln_class82", OPTSpec.tiny_pre_ln(82), edge2, 15)
        score_fixture("tiny_post_ln_class820", OPTSpec.tiny_post_ln(820), edge2, 16)      # vocab 512 < 820 labels: the cut
        import dataclasses
        score_fixture("tiny_post_ln_v1024_class820", dataclasses.replace(OPTSpec.tiny_post_ln(820), vocab_size=1024), edge2, 17)
    if args.big or args.only == "class":
        score_fixture("opt125m_class8192", OPTSpec.opt_125m(8192), [1, 5, 64, 65, 200, 33, 90, 17, 129, 300], 23)
    if args.big:
        big = [1, 2, 4, 5, 63, 64, 65, 200, 1024, 2048, 90, 33]
        score_fixture("opt125m", OPTSpec.opt_125m(), big, 21)
        score_fixture("opt350m", OPTSpec.opt_350m(), big[:9] + [300], 22)
