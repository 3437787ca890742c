"""-m gpu: the HIP predictor forward (through the C ABI) against the CPU oracle and the
vectors the reference itself produced (tests/golden/score_*.npz).

Tolerance: BASELINE.json north_star - scores within 1e-4 of the fp32 CPU predictor.
"""
import os

import numpy as np
import pytest
import torch

from oracle.opt_scorer import OracleOPTScorer
from util import GOLDEN, bench_lengths, spec_from_npz, synthetic_batch
from vllm_ltr_amd.opt_spec import OPTSpec, seeded_checkpoint

pytestmark = pytest.mark.gpu
TOL = 1e-4          # north_star tolerance on scores
TOL_HIDDEN = 2e-4   # on O(1..10) hidden-state entries


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return "cuda:0"


def _scorer(spec, ckpt, dev, mode, **kw):
    from vllm_ltr_amd.scorer import HipOPTScorer
    return HipOPTScorer(spec, ckpt, device=dev, weight_dtype=mode, **kw)


TINY = ["tiny_pre_ln", "tiny_post_ln", "tiny_pre_ln_class10", "tiny_post_ln_class7", "tiny_pre_ln_class82",
        "tiny_post_ln_class820", "tiny_post_ln_v1024_class820"]


@pytest.mark.parametrize("mode", ["f16", "f32"])
@pytest.mark.parametrize("name", TINY)
def test_golden_tiny(dev, name, mode):
    z = np.load(os.path.join(GOLDEN, f"score_{name}.npz"))
    spec = spec_from_npz(z)
    ckpt = seeded_checkpoint(spec, int(z["seed"]))
    sc = _scorer(spec, ckpt, dev, mode)
    got, logits = sc.score(z["ids"], z["cu_seqlens"], return_logits=True)
    if spec.num_labels == 1:
        err = np.abs(got - z["ref_score"]).max()
        print(f"{name}/{mode}: max|score - reference| = {err:.3e}")
        assert err <= TOL
        np.testing.assert_allclose(logits[:, 0], z["hf_logits"][:, 0], atol=TOL, rtol=0)
    else:
        np.testing.assert_allclose(logits, z["hf_logits"], atol=TOL, rtol=0)
        # class mode: float(argmax) (opt.py:394-395); a flipped argmax is only legitimate
        # when the top two logits are closer than the tolerance
        # (over the labels that survive the reference's vocab_size cut, logits_processor.py:68-70)
        top2 = np.sort(z["hf_logits"][:, :min(spec.num_labels, spec.vocab_size)], -1)[:, -2:]
        safe = (top2[:, 1] - top2[:, 0]) > 2 * TOL
        assert safe.sum() >= len(got) // 2
        assert (got[safe] == z["ref_score"][safe]).all()
        assert (got == np.floor(got)).all() and got.min() >= 0 and got.max() < min(spec.num_labels, spec.vocab_size)


def test_class_mode_head_at_8192_labels_true_width(dev):
    """The reference's largest class head (train/train.sh: bucket size 1 over 8192 -> 8,192 labels; tpt-class8192-xxx in
    benchmarks/): OPT-125m width, logits against HF at 1e-4, the label against the reference where the top-2 gap exceeds
    2e-4, and torch.argmax's FIRST-maximum rule on exact ties (two identical rows of score.weight)."""
    z = np.load(os.path.join(GOLDEN, "score_opt125m_class8192.npz"))
    spec = spec_from_npz(z)
    assert spec.num_labels == 8192
    ckpt = seeded_checkpoint(spec, int(z["seed"]))
    sc = _scorer(spec, ckpt, dev, "f16")
    got, logits = sc.score(z["ids"], z["cu_seqlens"], return_logits=True)
    err = np.abs(logits - z["hf_logits"]).max()
    top2 = np.sort(z["hf_logits"], -1)[:, -2:]
    safe = (top2[:, 1] - top2[:, 0]) > 2 * TOL
    print(f"class head, 8192 labels: max|logit - HF| = {err:.3e}; {int(safe.sum())} of {len(got)} requests with a clear top-2 gap")
    assert err <= TOL and safe.sum() >= len(got) - 2
    assert (got[safe] == z["ref_score"][safe]).all()
    assert (got == logits.argmax(-1)).all()                  # the label IS the argmax of the logits handed back
    # exact ties: copy the winning row of score.weight of request 0 to a LATER and an EARLIER label
    w = ckpt["score.weight"].copy()
    win = int(got[0])
    lo, hi = (win - 5) % 8192, (win + 7) % 8192
    w[lo] = w[win]; w[hi] = w[win]
    ck2 = dict(ckpt); ck2["score.weight"] = w
    got2 = _scorer(spec, ck2, dev, "f16").score(z["ids"], z["cu_seqlens"])
    assert got2[0] == min(win, lo, hi)


def test_class_head_logits_in_row_blocks():
    """The GEMM head keeps the padded class logits for a BLOCK of rows at a time (row windows of the label GEMM + argmax
    per block; 64 MB at most), so the scoring workspace does not grow with requests x labels.  In a fresh process with
    blocks of 128 rows (LTR_HEAD_BLOCK_ROWS, read once): 300 one-token-to-40-token requests, 820 labels on the tiny post-LN
    model - labels and logits identical to the one-block run of this process, and the workspace of an 8,192-request call at
    8,192 labels holds 64 MB of logits instead of 268."""
    import subprocess
    import sys
    code = r"""
import sys, numpy as np
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
from util import synthetic_batch
from vllm_ltr_amd.opt_spec import OPTSpec, seeded_checkpoint
from vllm_ltr_amd.scorer import HipOPTScorer
spec = OPTSpec.tiny_post_ln(820)
sc = HipOPTScorer(spec, seeded_checkpoint(spec, 16), 'cuda:0', 'f16')
lens = np.random.RandomState(0).randint(1, 41, 300).tolist()
ids, cu = synthetic_batch(spec, lens, 3)
s, l = sc.score(ids, cu, return_logits=True)
np.save(sys.argv[1], np.concatenate([s[:, None], l], 1))
"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for blk in ("0", "128"):
        path = os.path.join(root, "gpurun_out", f"_head_blocks_{blk}.npy")
        os.makedirs(os.path.dirname(path), exist_ok=True)
        r = subprocess.run([sys.executable, "-c", code, path], capture_output=True, text=True, timeout=600, cwd=root,
                           env=dict(os.environ, LTR_HEAD_BLOCK_ROWS=blk))
        assert r.returncode == 0, (r.stdout + r.stderr)[-2000:]
        outs.append(np.load(path))
        os.remove(path)
    assert np.array_equal(outs[0], outs[1]) and np.isfinite(outs[0]).all()
    assert (outs[0][:, 0] == outs[0][:, 1:1 + 512].argmax(-1)).all()     # (vocabulary 512 < 820 labels: logits_processor.py:68-70)
    from vllm_ltr_amd import _lib
    from vllm_ltr_amd.scorer import HipOPTScorer
    spec = OPTSpec.opt_125m(8192)
    sc = HipOPTScorer(spec, seeded_checkpoint(spec, 0), "cuda:0", "f16")
    small = int(sc.lib.ltr_workspace_bytes(sc._h, _lib.LTR_WS_SCORE, 8192, 8192))
    print(f"workspace of an 8,192-request / 8,192-token call at 8,192 labels: {small / 1e6:.0f} MB")
    # 253 MB of per-token buffers + 64 MB of logits + ... = 455 MB (round 3: + 268 MB of logits)
    assert small < 500e6, small


@pytest.mark.parametrize("name,mode", [("opt125m", "f16"), ("opt125m", "f32"), ("opt350m", "f16")])
def test_golden_true_shape(dev, name, mode):
    path = os.path.join(GOLDEN, f"score_{name}.npz")
    assert os.path.exists(path), "true-shape fixture missing: run oracle/make_golden.py --big"
    z = np.load(path)
    spec = spec_from_npz(z)
    ckpt = seeded_checkpoint(spec, int(z["seed"]))
    sc = _scorer(spec, ckpt, dev, mode)
    got = sc.score(z["ids"], z["cu_seqlens"])
    err = np.abs(got - z["ref_score"]).max()
    print(f"{name}/{mode}: N={len(got)} T={int(z['cu_seqlens'][-1])} max|score - reference| = {err:.3e}")
    assert err <= TOL


@pytest.mark.parametrize("mode", ["f16", "f32"])
@pytest.mark.parametrize("mk", [OPTSpec.tiny_pre_ln, OPTSpec.tiny_post_ln])
def test_per_layer_hidden_vs_oracle(dev, mk, mode):
    spec = mk()
    ckpt = seeded_checkpoint(spec, 5)
    ids, cu = synthetic_batch(spec, [7, 1, 64, 65, 2, 130, 33], 9)
    orc = OracleOPTScorer(spec, ckpt)
    sc = _scorer(spec, ckpt, dev, mode)
    for nl in range(spec.num_hidden_layers + 1):
        want = orc.hidden(ids, cu, n_layers=nl).numpy()
        got = sc.hidden(ids, cu, n_layers=nl)
        err = np.abs(got - want).max()
        assert err <= TOL_HIDDEN, f"layers={nl}: {err}"


@pytest.mark.parametrize("mode", ["f16", "f32"])
def test_embed_gather_kernel_alone(dev, mode):
    spec = OPTSpec.tiny_pre_ln()
    ckpt = seeded_checkpoint(spec, 3)
    ids, cu = synthetic_batch(spec, [5, 1, 100, 64, 3], 4)
    sc = _scorer(spec, ckpt, dev, mode)
    T, N = int(cu[-1]), len(cu) - 1
    out = torch.empty(T, spec.hidden_size, device=dev)
    sc.embed_gather_device(torch.from_numpy(ids).to(dev), torch.from_numpy(cu).to(dev), N, T, out)
    orc = OracleOPTScorer(spec, ckpt)
    want = orc.hidden(ids, cu, n_layers=0).numpy()
    assert np.array_equal(out.cpu().numpy(), want)      # gather + one f32 add: bit-exact


@pytest.mark.parametrize("mode", ["f16", "f32"])
@pytest.mark.parametrize("mk", [OPTSpec.tiny_pre_ln, OPTSpec.tiny_post_ln, lambda: OPTSpec.tiny_post_ln(5)])
def test_pool_head_kernel_alone(dev, mk, mode):
    spec = mk()
    ckpt = seeded_checkpoint(spec, 3)
    lens = [5, 1, 100, 64, 3]
    cu = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    T, N = int(cu[-1]), len(lens)
    h = torch.randn(T, spec.hidden_size, generator=torch.Generator().manual_seed(1)) * 2.0
    sc = _scorer(spec, ckpt, dev, mode)
    scores = torch.empty(N, device=dev)
    logits = torch.empty(N, spec.num_labels, device=dev)
    sc.pool_head_device(h.to(dev), torch.from_numpy(cu).to(dev), N, scores, logits)
    orc = OracleOPTScorer(spec, ckpt)
    want = orc.pool_head(h, torch.as_tensor(cu[1:].astype(np.int64) - 1)).numpy()
    np.testing.assert_allclose(logits.cpu().numpy(), want, atol=2e-5, rtol=0)
    if spec.num_labels == 1:
        np.testing.assert_allclose(scores.cpu().numpy(), want[:, 0], atol=2e-5, rtol=0)
    else:
        assert (scores.cpu().numpy() == want.argmax(-1)).all()


@pytest.mark.parametrize("mk", [OPTSpec.tiny_pre_ln, OPTSpec.tiny_post_ln])
def test_last_layer_pruning_is_invisible(dev, mk, monkeypatch):
    """ltr_score carries only the last-token rows through the last layer: Q, the attention output, out_proj, the
    LayerNorms and the MLP for n_req rows, K | V for every token; ltr_forward_hidden runs every row.  With the
    last-query attention switched off (LTR_NO_LASTQ=1) only per-token maps are pruned and pooling the full hidden
    states gives the same scores to f32 rounding (5e-7; round 3: bit for bit, when every batch size ran the same GEMM
    arithmetic); with it (default) the last query's softmax runs in f32 on the VALU instead of the split-fp16 MFMA
    passes - f32-grade agreement."""
    spec = mk()
    ckpt = seeded_checkpoint(spec, 6)
    ids, cu = synthetic_batch(spec, [9, 1, 64, 65, 2, 130, 33, 128, 31, 7, 8, 150], 10)
    sc = _scorer(spec, ckpt, dev, "f16")
    pruned = sc.score(ids, cu)
    h = torch.from_numpy(sc.hidden(ids, cu, n_layers=-1)).to(dev)
    full = torch.empty(len(cu) - 1, device=dev)
    sc.pool_head_device(h, torch.from_numpy(cu).to(dev), len(cu) - 1, full)
    full = full.cpu().numpy()
    np.testing.assert_allclose(pruned, full, atol=2e-6, rtol=0)
    monkeypatch.setenv("LTR_NO_LASTQ", "1")
    sc2 = _scorer(spec, ckpt, dev, "f16")
    # the same maths with different rounding - f32-grade agreement, not bit identity: the compact rows of the pruned last
    # layer (12 rows) and the full forward (700 rows) pick different GEMM kernels / split-K (launch_gemm chooses per launch),
    # and for post-LN blocks the compact rows run with explicit LayerNorm launches where the full forward rebuilds the
    # LayerNorm'd residuals in the GEMM epilogues
    np.testing.assert_allclose(sc2.score(ids, cu), full, atol=5e-7, rtol=0)


def test_chunking_is_invisible(dev):
    """Scores do not depend on how the batch is cut into passes - beyond the 2e-6 by which a score may move
    between the GEMM kernels the pass size selects (small passes run the narrow outputs with split-K: another summation
    order; tests/test_gpu_small_batches.py) - and a given cut is deterministic."""
    spec = OPTSpec.tiny_pre_ln()
    ckpt = seeded_checkpoint(spec, 8)
    lens = bench_lengths(300, seed=1, mu=24.0).clip(1, 150)
    ids, cu = synthetic_batch(spec, lens.tolist(), 2)
    sc = _scorer(spec, ckpt, dev, "f16")
    whole = sc.score(ids, cu)
    sc.set_chunk_tokens(200)          # many request-aligned chunks
    parts = sc.score(ids, cu)
    assert np.array_equal(parts, sc.score(ids, cu))
    assert np.abs(whole - parts).max() <= 2e-6 * max(1.0, float(np.abs(whole).max()))
    orc = OracleOPTScorer(spec, ckpt)
    assert np.abs(whole - orc.score(ids, cu)).max() <= TOL


def test_edge_inputs(dev):
    from vllm_ltr_amd._lib import LtrError
    spec = OPTSpec.tiny_pre_ln()
    ckpt = seeded_checkpoint(spec, 8)
    sc = _scorer(spec, ckpt, dev, "f16")
    assert sc.score(np.zeros(0, np.int64), np.zeros(1, np.int32)).shape == (0,)     # empty batch
    orc = OracleOPTScorer(spec, ckpt)
    ids, cu = synthetic_batch(spec, [1], 1)                                         # single 1-token request
    assert abs(sc.score(ids, cu)[0] - orc.score(ids, cu)[0]) <= TOL
    L = spec.max_position_embeddings                                                # maximum length
    ids, cu = synthetic_batch(spec, [L, 1, L], 2)
    assert np.abs(sc.score(ids, cu) - orc.score(ids, cu)).max() <= TOL
    # only one-token requests: n_req == T, the compact last-token buffers of the last layer are as large as the pass
    for mk in (OPTSpec.tiny_pre_ln, OPTSpec.tiny_post_ln):
        sp2 = mk()
        ck2 = seeded_checkpoint(sp2, 9)
        ids1, cu1 = synthetic_batch(sp2, [1] * 300, 4)
        s2 = _scorer(sp2, ck2, dev, "f16")
        want = OracleOPTScorer(sp2, ck2).score(ids1, cu1)
        assert np.abs(s2.score(ids1, cu1) - want).max() <= TOL
        s2.set_chunk_tokens(64)                                                      # ... and across several passes
        assert np.abs(s2.score(ids1, cu1) - want).max() <= TOL
    with pytest.raises(LtrError):                                                   # over-long prompt
        i2, c2 = synthetic_batch(spec, [L + 1], 3)
        sc.score(i2, c2)
    with pytest.raises(LtrError):                                                   # empty request
        sc.score(np.array([2, 5], np.int64), np.array([0, 0, 2], np.int32))


def test_bench_profile_properties_125m(dev):
    """At the BASELINE profile (true 125m shape) with sizes the oracle cannot finish
    quickly: size-independent properties - permutation invariance and determinism -
    plus an oracle spot check on a sub-sample."""
    spec = OPTSpec.opt_125m()
    ckpt = seeded_checkpoint(spec, 0)
    n = 512
    lens = bench_lengths(n, seed=0)
    ids, cu = synthetic_batch(spec, lens.tolist(), 0)
    sc = _scorer(spec, ckpt, dev, "f16")
    s1 = sc.score(ids, cu)
    s2 = sc.score(ids, cu)
    assert np.array_equal(s1, s2)                                   # deterministic
    perm = np.random.RandomState(1).permutation(n)                  # request order does not matter
    ids_p = np.concatenate([ids[cu[i]:cu[i + 1]] for i in perm])
    cu_p = np.concatenate([[0], np.cumsum(lens[perm])]).astype(np.int32)
    s3 = sc.score(ids_p, cu_p)
    assert np.abs(s3 - s1[perm]).max() <= 2e-5
    sub = np.arange(0, n, 37)[:12]                                  # oracle on a sub-sample
    orc = OracleOPTScorer(spec, ckpt)
    ids_s = np.concatenate([ids[cu[i]:cu[i + 1]] for i in sub])
    cu_s = np.concatenate([[0], np.cumsum(lens[sub])]).astype(np.int32)
    err = np.abs(orc.score(ids_s, cu_s) - s1[sub]).max()
    print(f"125m bench-profile spot check: max|d| = {err:.3e}")
    assert err <= TOL


def test_plugin_surface(dev):
    """obtain_aux_scores / ordered_requests / age on SequenceGroup-like objects behind ``install`` against the
    literal reference expressions, over 24 scheduler steps with arrivals, requests moving between the three
    deques and departures; the ranking state lives in device slots (``sync_host`` pulls it for the comparison)."""
    from collections import deque
    from oracle import rank_step as rs
    from util import FakeSeqGroup
    from vllm_ltr_amd.plugin import MI355XRanker

    spec = OPTSpec.tiny_pre_ln()
    ckpt = seeded_checkpoint(spec, 4)
    sc = _scorer(spec, ckpt, dev, "f16")
    ranker = MI355XRanker(sc, "opt-xxx-starv3-period2", max_length=100)
    r = np.random.RandomState(0)
    mk = lambda i: FakeSeqGroup(str(i), [2] + r.randint(4, spec.vocab_size, r.randint(1, 140)).tolist())
    groups = [mk(i) for i in range(40)]

    class Sched:
        def _general_schedule(self):
            pass
    s = Sched()
    s.waiting, s.running, s.swapped = deque(groups), deque(), deque()
    ranker.install(s)
    assert s._schedule.__wrapped__.__func__ is Sched._general_schedule and s.aux_model is ranker and s.starv == 3 and s.period == 2
    order = s._get_ordered_requests()
    assert all(g.aux_model_score is not None for g in groups)
    orc = OracleOPTScorer(spec, ckpt)
    for g in groups[:8]:                                   # truncation to max_length (aux_llm_engine.py:365-369)
        ids = np.array(g.prompt_token_ids[:100], np.int64)
        want = orc.score(ids, np.array([0, len(ids)], np.int32))[0]
        assert abs(g.aux_model_score - want) <= TOL
    # same order as the literal reference expression on the same scores
    mirror = {g.request_id: rs.Req(g.request_id, g.aux_model_score) for g in groups}
    concat = lambda: list(s.waiting) + list(s.running) + list(s.swapped)
    lit = rs.opt_order([mirror[g.request_id] for g in concat()], 3, 2)
    assert [g.request_id for g in order] == [m.request_id for m in lit]
    next_id = 40
    for step in range(24):
        ran = order[:5]
        # what a scheduler does with the selection: waiting -> running; now and then a running request is swapped
        # out, a swapped one comes back, and an old running request finishes
        for g in ran:
            if g in s.waiting:
                s.waiting.remove(g); s.running.append(g)
            elif g in s.swapped:
                s.swapped.remove(g); s.running.append(g)
        if step % 3 == 1 and len(s.running) > 3:
            g = s.running.popleft(); s.swapped.append(g)
        all_pri = list(s.swapped) + list(s.running) + list(s.waiting)          # scheduler.py:1337
        ranker.age(all_pri, ran)
        ran_ids = {g.request_id for g in ran}
        rs.age_update([mirror[g.request_id] for g in all_pri], [mirror[g.request_id] for g in all_pri if g.request_id in ran_ids])
        ranker.sync_host(all_pri)
        assert [(g.pri, g.idle, g.runs) for g in all_pri] == [(mirror[g.request_id].pri, mirror[g.request_id].idle, mirror[g.request_id].runs) for g in all_pri]
        if step % 4 == 2 and len(s.running) > 2:                                # a departure (finished request)
            s.running.popleft()
        for _ in range(int(r.randint(0, 4))):                                   # arrivals
            g = mk(next_id); next_id += 1
            s.waiting.append(g); groups.append(g)
        order = s._get_ordered_requests()
        for g in concat():
            if g.request_id not in mirror:
                mirror[g.request_id] = rs.Req(g.request_id, g.aux_model_score)
        lit = rs.opt_order([mirror[g.request_id] for g in concat()], 3, 2)
        assert [g.request_id for g in order] == [m.request_id for m in lit], step
        ranker.sync_host(concat())
        assert all((g.pri, g.idle, g.runs) == (mirror[g.request_id].pri, mirror[g.request_id].idle, mirror[g.request_id].runs) for g in concat())
    assert ranker.stats["requests_scored"] == len(groups)     # every request scored exactly once (sequence.py:461-465)
    # a request that reaches the ordering unscored fails loudly (the reference: -None in sorted())
    s.running.append(mk(10**6))
    with pytest.raises(TypeError):
        s._get_ordered_requests()
    # schedule types that need no score are refused, xpt needs its table
    with pytest.raises(ValueError):
        MI355XRanker(sc, "fifo").install(Sched())
    with pytest.raises(ValueError):
        MI355XRanker(sc, "xpt-nofile").install(Sched())


def test_install_alone_keeps_the_starvation_state_machine_running(dev):
    """An UNPATCHED reference scheduler (no INTEGRATION.md hunk (b): its aging loop, scheduler.py:1358-1365, touches the
    host attributes only) behind install(): the wrapped ``_schedule`` ages the device slots from the step's outputs, so
    ``idle >= starv`` promotions fire exactly where the reference's do.  Also: a scheduler constructed without any
    ``_update_priority`` (``fcfs``) works (scheduler.py:1103), arrivals inserted in the MIDDLE of ``waiting`` are
    scored by the fall-back full scan (:971-975), and ``plan_step(ordered=None)`` - permutation kept on the device -
    equals ``plan_step(ordered_list)``."""
    from collections import deque
    from types import SimpleNamespace
    from oracle import rank_step as rs
    from util import FakeSeqGroup
    from vllm_ltr_amd.plugin import MI355XRanker

    spec = OPTSpec.tiny_pre_ln()
    sc = _scorer(spec, seeded_checkpoint(spec, 4), dev, "f16")
    ranker = MI355XRanker(sc, "opt-xxx-starv3-period2", max_length=100)
    r = np.random.RandomState(3)
    mk = lambda i: FakeSeqGroup(str(i), [2] + r.randint(4, spec.vocab_size, r.randint(1, 40)).tolist())

    class Sched:                                             # the shape of Scheduler._general_schedule, host aging only
        def _general_schedule(self):
            self._update_priority()                          # :1103
            order = self._get_ordered_requests()
            ran = order[:4]
            for g in ran:
                if g in self.waiting:
                    self.waiting.remove(g); self.running.append(g)
            all_pri = list(self.swapped) + list(self.running) + list(self.waiting)
            for g in all_pri:                                # the reference's own loop: host attributes
                if g in ran:
                    if g.pri == -1:
                        g.runs -= 1
                    g.idle = 0
                else:
                    g.idle += 1
            self.last_order = order
            return SimpleNamespace(scheduled_seq_groups=[SimpleNamespace(seq_group=g) for g in ran])
    s = Sched()
    groups = [mk(i) for i in range(24)]
    s.waiting, s.running, s.swapped = deque(groups), deque(), deque()
    assert not hasattr(s, "_update_priority")
    ranker.install(s)
    mirror = {}
    promoted = 0
    for step in range(12):
        if step == 5:                                        # an arrival that is NOT a suffix of `waiting`
            g = mk(100); groups.append(g)
            s.waiting.insert(1, g)
        concat = list(s.waiting) + list(s.running) + list(s.swapped)
        ret = s._schedule()
        for g in concat:
            mirror.setdefault(g.request_id, rs.Req(g.request_id, g.aux_model_score))
        lit = rs.opt_order([mirror[g.request_id] for g in concat], 3, 2)
        assert [g.request_id for g in s.last_order] == [m.request_id for m in lit], step
        promoted += sum(m.pri == -1 for m in lit)
        ran_ids = {x.seq_group.request_id for x in ret.scheduled_seq_groups}
        all_pri = list(s.swapped) + list(s.running) + list(s.waiting)
        rs.age_update([mirror[g.request_id] for g in all_pri], [mirror[g.request_id] for g in all_pri if g.request_id in ran_ids])
        ranker.sync_host(all_pri)                            # device slots == the literal state machine
        assert [(g.pri, g.idle, g.runs) for g in all_pri] == \
            [(mirror[g.request_id].pri, mirror[g.request_id].idle, mirror[g.request_id].runs) for g in all_pri], step
    assert promoted > 0                                      # the starvation path was exercised
    # plan_step with the permutation left on the device == plan_step on the ordered list
    reqs = list(s.waiting) + list(s.running) + list(s.swapped)
    n = len(reqs)
    nt = r.randint(1, 50, n).astype(np.int32); nq = np.ones(n, np.int32)
    order = ranker.order(reqs)
    pos = {id(g): i for i, g in enumerate(reqs)}
    want = ranker.plan_step(order, [nt[pos[id(g)]] for g in order], [nq[pos[id(g)]] for g in order], 120, 6)
    assert ranker.order(reqs, want_list=False) is None
    got = ranker.plan_step(None, nt, nq, 120, 6)
    assert [g.request_id for g in got["ordered"]] == [g.request_id for g in order]
    assert [g.request_id for g in got["selected"]] == [g.request_id for g in want["selected"]] and got["granted"] == want["granted"]
    assert len(got["selected"]) > 0


@pytest.mark.timing
def test_plugin_step_time_at_8k(dev):
    """ordered_requests() + age() through the plug-in on 8,192 request objects: the device-resident queue keeps
    the per-step host work to one C-level pass over the queue (round 1 marshalled every counter of every object:
    ~14 ms; the reference's own promote/demote + sorted() + aging loops: ~5 ms)."""
    import time
    from collections import deque
    from oracle import rank_step as rs
    from util import FakeSeqGroup
    from vllm_ltr_amd.plugin import MI355XRanker

    spec = OPTSpec.tiny_pre_ln()
    sc = _scorer(spec, seeded_checkpoint(spec, 4), dev, "f16")
    ranker = MI355XRanker(sc, "opt-xxx-starv200-period10", max_length=100)
    r = np.random.RandomState(1)
    n = 8192
    groups = [FakeSeqGroup(str(i), [2] + r.randint(4, spec.vocab_size, 3).tolist()) for i in range(n)]

    class Sched:
        pass
    s = Sched()
    s.waiting, s.running, s.swapped = deque(groups), deque(), deque()
    ranker.install(s)
    order = s._get_ordered_requests()                      # cold: scores everything
    mirror = [rs.Req(g.request_id, g.aux_model_score) for g in groups]
    t_order, t_age, t_ref = [], [], []
    for step in range(12):
        ran = order[:256]
        all_pri = list(s.swapped) + list(s.running) + list(s.waiting)
        t0 = time.perf_counter(); ranker.age(all_pri, ran); torch.cuda.synchronize(); t_age.append(time.perf_counter() - t0)
        t0 = time.perf_counter(); order = s._get_ordered_requests(); t_order.append(time.perf_counter() - t0)
        ran_ids = {g.request_id for g in ran}
        t0 = time.perf_counter()
        rs.age_update(mirror, [m for m in mirror if m.request_id in ran_ids])
        lit = rs.opt_order(mirror, 200, 10)
        t_ref.append(time.perf_counter() - t0)
        assert [g.request_id for g in order] == [m.request_id for m in lit]
    med = lambda a: sorted(a)[len(a) // 2]
    total = med(t_order) + med(t_age)
    print(f"plug-in step at {n} objects: ordered_requests {med(t_order)*1e3:.3f} ms + age {med(t_age)*1e3:.3f} ms = "
          f"{total*1e3:.3f} ms; literal reference loops on this host {med(t_ref)*1e3:.3f} ms")
    # (no wall-clock assertion: the order above is the parity property; the times are a report, `bench.py` measures them)


def test_plugin_tpt_and_xpt_orders(dev):
    """tpt (class-mode score, string request-id tiebreak, scheduler.py:948) and xpt (expected-length
    table + SRTF key, scheduler.py:910-933) through the plug-in against the literal expressions."""
    from oracle import rank_step as rs
    from util import FakeSeqGroup
    from vllm_ltr_amd.plugin import MI355XRanker

    spec = OPTSpec.tiny_pre_ln(10)                          # class mode: many exact score ties
    ckpt = seeded_checkpoint(spec, 9)
    sc = _scorer(spec, ckpt, dev, "f16")
    r = np.random.RandomState(3)
    groups = [FakeSeqGroup(str(i), [2] + r.randint(4, spec.vocab_size, r.randint(1, 60)).tolist()) for i in range(120)]
    ranker = MI355XRanker(sc, "tpt-class", max_length=100, mtype="class")
    ranker.obtain_aux_scores(groups)
    got = ranker.order(groups, "tpt")
    mirror = [rs.Req(g.request_id, g.aux_model_score) for g in groups]
    assert [g.request_id for g in got] == [m.request_id for m in rs.tpt_order(mirror)]
    assert len({g.aux_model_score for g in groups}) < len(groups)      # ties were exercised

    key = [-3.0, -1.0, 0.0, 1.0, 2.5]
    value = [900, 400, 150, 60, 20]
    xr = MI355XRanker(sc, "opt", max_length=100, mtype="class", xpt_distribution=(key, value))
    for g in groups:
        g.output_len = int(r.randint(0, 50))
    got = xr.order(groups, "xpt")
    mirror = [rs.Req(g.request_id, g.aux_model_score) for g in groups]
    for m, g in zip(mirror, groups):
        m.output_len = g.output_len
    want = rs.xpt_order(mirror, key, value, lambda q: q.output_len)
    assert [g.request_id for g in got] == [m.request_id for m in want]
    # `ltr` / `constraint` (scheduler.py:1020-1052; unbound in the reference's string table): sorted(key=-score), ties by
    # position; `constraint` keeps scheduler.records = the sorted ranking scores (-score) of everything scored so far
    from collections import deque
    lr = MI355XRanker(sc, "opt-xxx-starv3-period2", max_length=100, mtype="class")   # (the starvation part does not apply)
    fresh = [FakeSeqGroup(g.request_id, g.prompt_token_ids) for g in groups]

    class Sched:
        pass
    s2 = Sched()
    s2.waiting, s2.running, s2.swapped, s2.aux_model = deque(fresh[:30]), deque(), deque(), lr
    got = lr.ordered_requests(s2, "constraint")
    lit = sorted(list(s2.waiting), key=lambda q: -q.aux_model_score)
    assert [g.request_id for g in got] == [g.request_id for g in lit]
    assert s2.records == sorted(-g.aux_model_score for g in fresh[:30])
    s2.waiting.extend(fresh[30:])
    got = lr.ordered_requests(s2, "ltr")
    assert [g.request_id for g in got] == [g.request_id for g in sorted(list(s2.waiting), key=lambda q: -q.aux_model_score)]
    assert len(s2.records) == 30


@pytest.mark.gpu
@pytest.mark.parametrize("variant", ["tiny_pre_ln", "tiny_post_ln"])
def test_outlier_activations(dev, variant):
    """Trained OPT checkpoints carry a few 'massive activation' channels; the hi|lo split must keep its
    f32-grade relative accuracy when the residual stream spans several orders of magnitude (and stay
    finite: hi is an fp16).  Weights 5x the init scale plus two embedding channels at ~60x."""
    spec = OPTSpec.tiny_pre_ln() if variant == "tiny_pre_ln" else OPTSpec.tiny_post_ln()
    ckpt = seeded_checkpoint(spec, 31, std=0.1, qk_std=0.15)
    r = np.random.RandomState(5)
    for name in ("model.decoder.embed_tokens.weight", "model.decoder.embed_positions.weight"):
        w = ckpt[name].astype(np.float32)
        w[:, [3, 17]] += 60.0 * r.standard_normal((w.shape[0], 2)).astype(np.float32) * 0.05 + np.array([40.0, -55.0], np.float32)
        ckpt[name] = w.astype(np.float16)
    ids, cu = synthetic_batch(spec, [1, 7, 33, 64, 129, 150], seed=9)
    sc = _scorer(spec, ckpt, dev, "f16")
    got = sc.score(ids, cu)
    want = OracleOPTScorer(spec, ckpt, dtype=torch.float64).score(ids, cu)
    scale = max(1.0, float(np.abs(want).max()))
    rel = float(np.abs(got - want).max()) / scale
    print(f"{variant}: score range {np.abs(want).max():.2f}, max rel err {rel:.2e}")
    assert np.isfinite(got).all() and rel <= 2e-5


def test_layernorm_fold_operand_overflow_is_reported(dev, monkeypatch):
    """The LayerNorm-fold operand is split(x * gamma * 16) of the UN-normalised residual stream: beyond |x gamma| ~ 4094
    its fp16 hi plane is inf.  That must not pass silently as NaN scores: ltr_status reports it (LTR_E_RANGE), and the
    same checkpoint scores correctly on a handle without the fold (LTR_NO_LN_FOLD=1: bounded LayerNorm output)."""
    from vllm_ltr_amd._lib import LtrError
    spec = OPTSpec.tiny_pre_ln()
    ckpt = seeded_checkpoint(spec, 31)
    w = ckpt["model.decoder.embed_positions.weight"].astype(np.float32)
    w[:, 5] += 8000.0                                       # one massive channel, exact in fp16
    ckpt["model.decoder.embed_positions.weight"] = w.astype(np.float16)
    ids, cu = synthetic_batch(spec, [1, 7, 33, 64], seed=9)
    sc = _scorer(spec, ckpt, dev, "f16")
    with pytest.raises(LtrError, match="fp16 range"):
        sc.score(ids, cu)
    near = dict(ckpt)                                       # just inside the range: no flag, f32-grade scores
    w2 = seeded_checkpoint(spec, 31)["model.decoder.embed_positions.weight"].astype(np.float32)
    w2[:, 5] += 2500.0
    near["model.decoder.embed_positions.weight"] = w2.astype(np.float16)
    want_near = OracleOPTScorer(spec, near, dtype=torch.float64).score(ids, cu)
    got_near = _scorer(spec, near, dev, "f16").score(ids, cu)
    assert np.isfinite(got_near).all() and np.abs(got_near - want_near).max() <= 1e-4 * max(1.0, np.abs(want_near).max())
    want = OracleOPTScorer(spec, ckpt, dtype=torch.float64).score(ids, cu)
    tol = 1e-4 * max(1.0, np.abs(want).max())
    # the handle flag (what a caller passes): LTR_F_NO_LN_FOLD through HipOPTScorer(ln_fold=False)
    from vllm_ltr_amd.scorer import HipOPTScorer
    got = HipOPTScorer(spec, ckpt, str(dev), "f16", ln_fold=False).score(ids, cu)
    assert np.isfinite(got).all() and np.abs(got - want).max() <= tol
    # the serving plug-in does not raise through schedule() -> step() (llm_engine.py:569 would end the engine): it re-scores
    # the batch on the unfolded twin handle, counts the fallback and stays on the twin
    from util import FakeSeqGroup
    from vllm_ltr_amd.plugin import MI355XRanker
    for prescore in (False, True):
        rk = MI355XRanker(_scorer(spec, ckpt, dev, "f16"), "opt", max_length=150, prescore=prescore)
        groups = [FakeSeqGroup(str(i), ids[cu[i]:cu[i + 1]].tolist()) for i in range(len(cu) - 1)]
        if prescore:
            for g in groups[:2]:
                rk.add_request(g)
        got = np.array(rk.obtain_aux_scores(groups))
        assert np.abs(got - want).max() <= tol, (prescore, got, want)
        assert rk.metrics()["range_fallbacks"] == 1 and rk.scorer.ln_fold is False
        assert [int(g.request_id) for g in rk.order(groups)] == sorted(range(len(groups)), key=lambda i: -got[i])
        more = [FakeSeqGroup(f"m{i}", ids[cu[i]:cu[i + 1]].tolist()) for i in range(len(cu) - 1)]
        if prescore:
            rk.add_request(more[0])
        got2 = np.array(rk.obtain_aux_scores(more))            # later calls run on the twin: no second fallback
        assert np.abs(got2 - want).max() <= tol and rk.metrics()["range_fallbacks"] == 1
    monkeypatch.setenv("LTR_NO_LN_FOLD", "1")                  # the environment switch (diag) still works
    got = _scorer(spec, ckpt, dev, "f16").score(ids, cu)
    assert np.isfinite(got).all() and np.abs(got - want).max() <= tol


@pytest.mark.parametrize("variant,expected,mode", [("st", "expected_class2", "f16"), ("sharded", "expected_class2", "f16"),
                                                   ("rank1", "expected_rank1", "f16"), ("fp32", "expected_rank1_fp32", "f32")])
def test_hf_written_checkpoints_through_the_hip_scorer(dev, variant, expected, mode):
    """Directories HF itself wrote (tests/golden/hf_tiny, oracle/make_hf_fixture.py) -> load_hf_checkpoint -> HIP
    forward: logits within 1e-4 of what HF computed (a14)."""
    from vllm_ltr_amd._lib import LtrError
    from vllm_ltr_amd.opt_spec import checkpoint_weight_dtype, load_hf_checkpoint
    path = os.path.join(GOLDEN, "hf_tiny", variant)
    spec, ckpt = load_hf_checkpoint(path)
    assert checkpoint_weight_dtype(ckpt) == mode
    z = np.load(os.path.join(GOLDEN, "hf_tiny", expected + ".npz"))
    sc = _scorer(spec, ckpt, dev, mode)
    got, logits = sc.score(z["ids"], z["cu_seqlens"], return_logits=True)
    np.testing.assert_allclose(logits, z["logits"], atol=TOL, rtol=0)
    if spec.num_labels == 1:
        np.testing.assert_allclose(got, z["logits"][:, 0], atol=TOL, rtol=0)
    if mode == "f32":
        with pytest.raises(LtrError):                       # an fp32 checkpoint is never silently rounded to fp16
            _scorer(spec, ckpt, dev, "f16")


def test_out_of_vocabulary_token_id_is_reported(dev):
    """F.embedding raises on an id outside the table (vocab_parallel_embedding.py:95-106); the asynchronous HIP path
    flags it and the first status check / score read raises."""
    from vllm_ltr_amd._lib import LtrError
    spec = OPTSpec.tiny_pre_ln()
    sc = _scorer(spec, seeded_checkpoint(spec, 8), dev, "f16")
    ids, cu = synthetic_batch(spec, [5, 9, 3], 1)
    good = sc.score(ids, cu)
    for bad in (spec.vocab_size, -1, 2 ** 40):
        ids2 = ids.copy(); ids2[7] = bad
        with pytest.raises(LtrError, match="token id"):
            sc.score(ids2, cu)
    assert np.array_equal(sc.score(ids, cu), good)          # the flag is cleared by the failed check


def test_max_len_contract(dev):
    import ctypes as C
    from vllm_ltr_amd import _lib
    spec = OPTSpec.tiny_pre_ln()
    sc = _scorer(spec, seeded_checkpoint(spec, 8), dev, "f16")
    ids, cu = synthetic_batch(spec, [5, 9, 3], 1)
    ids_d, cu_d = torch.from_numpy(ids).to(dev), torch.from_numpy(cu).to(dev)
    out = torch.empty(3, device=dev)
    ws = sc._workspace(3, int(cu[-1]))
    rc = sc.lib.ltr_score(sc._h, ids_d.data_ptr(), cu_d.data_ptr(), cu.ctypes.data, 3, int(cu[-1]), 8, out.data_ptr(), None,
                          ws.data_ptr(), ws.numel(), sc._stream())
    assert rc == -22 and b"max_len" in sc.lib.ltr_last_error()      # a 9-token request against max_len = 8


def test_handle_of_another_device_is_usable(dev):
    """The handle remembers its device: calls work whatever device is current.  One-GPU boxes can only
    check that the guard leaves the current device alone."""
    spec = OPTSpec.tiny_pre_ln()
    sc = _scorer(spec, seeded_checkpoint(spec, 8), "cuda", "f16")     # no index: resolved to the current device
    assert sc.device.index == torch.cuda.current_device()
    ids, cu = synthetic_batch(spec, [5, 9, 3], 1)
    before = torch.cuda.current_device()
    sc.score(ids, cu)
    assert torch.cuda.current_device() == before
    if torch.cuda.device_count() > 1:
        sc1 = _scorer(spec, seeded_checkpoint(spec, 8), "cuda:1", "f16")
        assert np.array_equal(sc1.score(ids, cu), sc.score(ids, cu)) and torch.cuda.current_device() == before


def test_plugin_slot_reclamation_mirror_and_adoption(dev):
    """Long-running queue behind the plug-in: thousands of arrivals and departures through a small live set so that slots
    are reclaimed and reused (the counters of a reused slot must restart at zero, scheduler.py:372-374); mirror_host=True
    writes pri / idle / runs back every step like the reference's loops; requests scored before the ranker saw them are
    adopted with their host state."""
    from collections import deque
    from oracle import rank_step as rs
    from util import FakeSeqGroup
    from vllm_ltr_amd.plugin import MI355XRanker

    spec = OPTSpec.tiny_pre_ln()
    sc = _scorer(spec, seeded_checkpoint(spec, 4), dev, "f16")
    ranker = MI355XRanker(sc, "opt-xxx-starv2-period1", max_length=100, mirror_host=True)
    r = np.random.RandomState(5)
    nid = [0]

    def mk():
        nid[0] += 1
        return FakeSeqGroup(str(nid[0]), [2] + r.randint(4, spec.vocab_size, r.randint(1, 12)).tolist())

    class Sched:
        pass
    s = Sched()
    s.waiting, s.running, s.swapped = deque(mk() for _ in range(30)), deque(), deque()
    # three requests that some other component scored and aged already
    for g, (p, i, u) in zip(list(s.waiting)[:3], [(-1, 0, 1), (0, 5, 0), (0, 1, 0)]):
        g.aux_model_score, g.pri, g.idle, g.runs = float(r.standard_normal()), p, i, u
    ranker.install(s)
    mirror = {}
    peak_slots = 0
    for step in range(260):
        order = s._get_ordered_requests()
        reqs = list(s.waiting) + list(s.running) + list(s.swapped)
        for g in reqs:
            if g.request_id not in mirror:
                mirror[g.request_id] = rs.Req(g.request_id, g.aux_model_score)
        if step == 0:                                        # adopted host state (before this step's promote/demote)
            for g, (p, i, u) in zip(reqs[:3], [(-1, 0, 1), (0, 5, 0), (0, 1, 0)]):
                mirror[g.request_id].pri, mirror[g.request_id].idle, mirror[g.request_id].runs = p, i, u
        lit = rs.opt_order([mirror[g.request_id] for g in reqs], 2, 1)
        assert [g.request_id for g in order] == [m.request_id for m in lit], step
        assert all((g.pri, g.idle, g.runs) == (mirror[g.request_id].pri, mirror[g.request_id].idle, mirror[g.request_id].runs)
                   for g in reqs), step                      # mirror_host: the objects follow the device state
        ran = order[:6]
        for g in ran:
            if g in s.waiting:
                s.waiting.remove(g); s.running.append(g)
        all_pri = list(s.swapped) + list(s.running) + list(s.waiting)
        ranker.age(all_pri, ran)
        ran_ids = {g.request_id for g in ran}
        rs.age_update([mirror[g.request_id] for g in all_pri], [mirror[g.request_id] for g in all_pri if g.request_id in ran_ids])
        assert all((g.pri, g.idle, g.runs) == (mirror[g.request_id].pri, mirror[g.request_id].idle, mirror[g.request_id].runs)
                   for g in all_pri), step
        for _ in range(5):                                   # five departures and five arrivals per step: the live set stays ~30
            if s.running:
                s.running.popleft()
        for _ in range(5):
            s.waiting.append(mk())
        peak_slots = max(peak_slots, ranker.queue.n)
    assert nid[0] > 1300 and peak_slots < 1300              # slots were reclaimed and reused (one per live request + slack)
    assert nid[0] - 10 <= ranker.stats["requests_scored"] <= nid[0] - 3      # everything but the 3 adopted (and the last arrivals)


@pytest.mark.parametrize("variant", ["tiny_pre_ln", "tiny_post_ln"])
def test_layernorm_fold_on_and_off_agree(dev, monkeypatch, variant):
    """The LayerNorm fold (default for fp16 models: operand + row statistics from the producing GEMM's epilogue,
    normalisation in the consuming GEMM's epilogue; post-LN blocks additionally rebuild the LayerNorm'd RESIDUAL in the
    epilogue of the GEMM that adds it) against the same forward with separate LayerNorm launches (LTR_NO_LN_FOLD=1, read
    by ltr_create), both against the oracle - including rows with a large mean / std ratio, where the fold's
    `acc - mean c` cancels."""
    spec = OPTSpec.tiny_pre_ln() if variant == "tiny_pre_ln" else OPTSpec.tiny_post_ln()
    ckpt = seeded_checkpoint(spec, 12)
    w = ckpt["model.decoder.embed_positions.weight"].astype(np.float32)
    w += 0.5                                              # every residual row gets mean ~ 0.5 against std ~ 0.03
    ckpt["model.decoder.embed_positions.weight"] = w.astype(np.float16)
    ids, cu = synthetic_batch(spec, [1, 2, 17, 64, 65, 128, 150, 33], 13)
    folded = _scorer(spec, ckpt, dev, "f16")
    monkeypatch.setenv("LTR_NO_LN_FOLD", "1")
    plain = _scorer(spec, ckpt, dev, "f16")
    monkeypatch.delenv("LTR_NO_LN_FOLD")
    a, b = folded.score(ids, cu), plain.score(ids, cu)
    want = OracleOPTScorer(spec, ckpt, dtype=torch.float64).score(ids, cu)
    ea, eb = float(np.abs(a - want).max()), float(np.abs(b - want).max())
    print(f"LayerNorm fold: max|d| vs f64 oracle {ea:.2e} (folded), {eb:.2e} (separate launches)")
    assert ea <= 2e-5 and eb <= 2e-5
    folded.profile(True); folded.profile_read(True); folded.score(ids, cu)
    plain.profile(True); plain.profile_read(True); plain.score(ids, cu)
    # pre-LN: layer 0's first LayerNorm + the LayerNorm of the n_req last-token rows in front of the last layer's Q GEMM
    # (last-query pruning).  Post-LN: the three LayerNorms of the compact rows of the pruned last layer.
    nf, npl = folded.profile_read()["ln"]["launches"], plain.profile_read()["ln"]["launches"]
    if variant == "tiny_pre_ln":
        assert nf == 2 and npl == 2 * spec.num_hidden_layers + 1
    else:
        assert nf == 3 and npl == 2 * spec.num_hidden_layers
    # per-layer hidden states through the fold (the test hook stops after k layers: the last layer run materialises
    # its LayerNorm) against the oracle
    orc = OracleOPTScorer(spec, ckpt, dtype=torch.float64)
    for k in range(1, spec.num_hidden_layers + 1):
        got = folded.hidden(ids, cu, n_layers=k)
        wantk = orc.hidden(ids, cu, n_layers=k).numpy()
        assert np.abs(got - wantk).max() <= 2e-4 * max(1.0, np.abs(wantk).max()), k


# tiny_*: the OPT-125m and the OPT-350m block structures; opt350m: config 5's predictor itself at its true shape (2,000 requests)
@pytest.mark.parametrize("family,prescore", [("tiny_pre_ln", False), ("tiny_post_ln", False), ("opt350m", False),
                                             ("tiny_pre_ln", True), ("opt350m", True)])
@pytest.mark.parametrize("kind", ["burst", "gamma"])
@pytest.mark.timing
def test_config5_ranker_side_trace_replay(dev, kind, family, prescore):
    """BASELINE config 5, the ranker's share: a burst (everything at t = 0, benchmarks/burst-*.sh) and a gamma arrival
    process (benchmark_serving_real.py:159-176) replayed through MI355XRanker.install() on an (unpatched) scheduler
    loop - per step k arrivals -> obtain_aux_scores(k) + order + aging.  EVERY step's order is compared with the
    literal reference expressions (promote/demote + stable sorted, scheduler.py:984-998; aging :1358-1365) replayed
    on the same deques; every request finishes; latency percentiles come out of the summary.  ``prescore``: the same with
    the requests scored when they ARRIVE (asynchronously, ``MI355XRanker(prescore=True)``): same checks, and the scheduler
    steps that admit arrivals no longer wait for a forward."""
    from oracle import rank_step as rs
    from vllm_ltr_amd.plugin import MI355XRanker
    from vllm_ltr_amd.replay import replay, summarize, synthetic_trace
    true_shape = family == "opt350m"
    spec = OPTSpec.opt_350m() if true_shape else getattr(OPTSpec, family)()
    ckpt = seeded_checkpoint(spec, 0 if true_shape else 4)
    sc = _scorer(spec, ckpt, dev, "f16")
    n_req = 2000 if true_shape else 400
    if true_shape:       # bench.py --trace --model 350m: 2,048-token / 256-sequence budget, prompts of median 64
        ranker = MI355XRanker(sc, "opt-xxx-starv20-period3", max_length=1024, prescore=prescore)
        reqs = synthetic_trace(spec.vocab_size, n_req, kind, request_rate=64.0, cv=1.0, seed=0, prompt_median=64.0,
                               output_median=24.0)
        budget = dict(backbone_ms=25.0, max_num_batched_tokens=2048, max_num_seqs=256)
    else:
        ranker = MI355XRanker(sc, "opt-xxx-starv20-period3", max_length=150, prescore=prescore)
        reqs = synthetic_trace(spec.vocab_size, n_req, kind, request_rate=200.0, cv=2.0, seed=1, prompt_median=24.0,
                               output_median=12.0, max_prompt=140)
        budget = dict(backbone_ms=5.0, max_num_batched_tokens=256, max_num_seqs=16)
    mirror, state = {}, {}

    def before(step, s):
        state["concat"] = list(s.waiting) + list(s.running) + list(s.swapped)

    def after(step, s, ran):
        concat = state["concat"]
        for g in concat:
            mirror.setdefault(g.request_id, rs.Req(g.request_id, g.aux_model_score))
        lit = rs.opt_order([mirror[g.request_id] for g in concat], 20, 3)
        assert [g.request_id for g in s.last_order] == [m.request_id for m in lit], step
        ran_ids = {g.request_id for g in ran}
        all_pri = list(s.swapped) + list(s.running) + list(s.waiting)
        rs.age_update([mirror[g.request_id] for g in all_pri], [mirror[g.request_id] for g in all_pri if g.request_id in ran_ids])
        state["promoted"] = state.get("promoted", 0) + sum(m.pri == -1 for m in lit)

    res = replay(ranker, reqs, before_step=before, on_step=after, **budget)
    s = summarize(res)
    print(f"{family} {kind}{' prescore' if prescore else ''}: {s['steps']} steps, max queue {s['max_queue']}, ranker p50/p95/p99 = "
          f"{s['ranker_ms_all']['p50']:.3f} / {s['ranker_ms_all']['p95']:.3f} / {s['ranker_ms_all']['p99']:.3f} ms, steps with arrivals p50 "
          f"{s['ranker_ms_with_arrivals']['p50']:.3f} ms, ranker share of HOL {s['ranker_share_of_hol']:.3f}")
    if prescore:
        m = ranker.metrics()["prescore"]
        print("prescore:", m)
        assert m["requests"] + 0 <= n_req and m["requests"] >= (n_req // 2 if kind == "gamma" else 1)
        if kind == "burst":
            assert m["launches"] <= 16                         # a burst of 400 / 2,000 arrivals: ~log2 N growing batches, not one launch each
    assert s["finished"] == n_req and all(r.aux_model_score is not None for r in reqs)
    assert ranker.stats["requests_scored"] == n_req            # every request scored exactly once
    if true_shape:       # the scores the replay ran on, against the oracle (first / last arrivals and a few in between)
        pick = [0, 1, n_req // 3, n_req // 2, n_req - 2, n_req - 1]
        orc = OracleOPTScorer(spec, ckpt)
        for i in pick:
            ids_i = np.asarray(reqs[i].prompt_token_ids, np.int64)
            want = orc.score(ids_i, np.array([0, len(ids_i)], np.int32))[0]
            assert abs(reqs[i].aux_model_score - want) <= TOL, (i, reqs[i].aux_model_score, want)
    if not (true_shape and kind == "gamma"):                   # (64 req/s against a 256-sequence budget never starves a request)
        assert state["promoted"] > 0                           # starvation promotions happened and matched
    if kind == "burst":
        assert s["ranker_ms_with_arrivals"]["n"] == 1          # one scoring step for the whole burst
    else:
        assert s["ranker_ms_with_arrivals"]["n"] > 20
    assert s["ranker_ms_steady"]["n"] > 0             # (latencies are reported by `bench.py --trace`, never asserted here)


@pytest.mark.parametrize("name,hidden,ffn,heads,pre_ln,embed", [("1.3b", 2048, 8192, 32, True, 2048),
                                                                  ("350m-wide", 1536, 6144, 24, False, 768)])
def test_wider_opt_shapes_two_layers(dev, name, hidden, ffn, heads, pre_ln, embed):
    """The OPT family beyond the two predictors of the benchmark (the reference takes any HF OPT checkpoint through
    `OPTForSequenceClassification`, opt.py:349-397): two layers at the width of OPT-1.3b (H = 2048, 32 heads, FFN 8192:
    32 statistics pieces per row in the LayerNorm fold, K = 8192 in fc2) and a post-LN / project_in-out shape wider than
    OPT-350m, small vocabulary.  Oracle on every request, one-request-at-a-time against the batch (small- vs large-tile
    kernels), fold on / off."""
    spec = OPTSpec(vocab_size=4096, hidden_size=hidden, ffn_dim=ffn, num_hidden_layers=2, num_attention_heads=heads,
                   word_embed_proj_dim=embed, max_position_embeddings=512, do_layer_norm_before=pre_ln, num_labels=1)
    ckpt = seeded_checkpoint(spec, 3)
    lens = [1, 2, 17, 31, 32, 33, 64, 100, 128, 129, 200, 300, 5, 77, 256, 450]
    ids, cu = synthetic_batch(spec, lens, 5)
    sc = _scorer(spec, ckpt, dev, "f16")
    got = sc.score(ids, cu)
    want = OracleOPTScorer(spec, ckpt).score(ids, cu)
    err = np.abs(got - want).max()
    print(f"OPT {name} shape, 2 layers: max|score - oracle| = {err:.3e}")
    assert np.isfinite(got).all() and err <= TOL
    for i in (0, 5, 11, 15):                  # alone (small-batch kernels, split-K) against inside the batch: f32 rounding
        one = sc.score(ids[cu[i]:cu[i + 1]], np.array([0, lens[i]], np.int32))            # (K = 8,192 sums: a few 1e-6)
        assert abs(one[0] - got[i]) <= 1e-5 and abs(one[0] - want[i]) <= TOL, (i, one[0], got[i])
    os.environ["LTR_NO_LN_FOLD"] = "1"
    try:
        plain = _scorer(spec, ckpt, dev, "f16")
    finally:
        del os.environ["LTR_NO_LN_FOLD"]
    assert np.abs(plain.score(ids, cu) - got).max() <= 2e-5


@pytest.mark.parametrize("model", ["125m", "350m"])
def test_row_statistics_combined_once_per_launch_are_invisible(dev, model, monkeypatch):
    """LayerNorm fold, large passes (>= 8,192 rows): the (mean, M2) pieces a producer GEMM wrote are combined ONCE by
    `row_stats_combine_kernel` into (mean, rstd) per row, and the consuming tiles (QKV / fc1 consumers, the post-LN blocks'
    LayerNorm'd residual) load one float2 per row instead of gathering 12-16 pieces in front of their first barrier.  Same
    function on the same inputs: the scores must be BIT-identical to the in-tile combine (LTR_STATS_COMB_MIN switches it off)."""
    from util import bench_lengths
    spec = OPTSpec.opt_125m() if model == "125m" else OPTSpec.opt_350m()
    sc = _scorer(spec, seeded_checkpoint(spec, 2), dev, "f16")
    lens = bench_lengths(150, seed=4, mu=80.0)
    ids, cu = synthetic_batch(spec, lens.tolist(), 8)
    assert int(cu[-1]) >= 8192 + 4096                               # ONE pass of well over 8,192 rows
    monkeypatch.setenv("LTR_STATS_COMB_MIN", "1000000000")
    base = sc.score(ids, cu)
    monkeypatch.setenv("LTR_STATS_COMB_MIN", "8192")
    comb = sc.score(ids, cu)
    monkeypatch.setenv("LTR_STATS_COMB_MIN", "1")                   # (every pass, also the compact last-token rows)
    comb1 = sc.score(ids, cu)
    assert np.array_equal(base, comb) and np.array_equal(base, comb1)
    want = OracleOPTScorer(spec, seeded_checkpoint(spec, 2)).score(*_first(ids, cu, 6))
    assert np.abs(comb[:6] - want).max() <= TOL


def _first(ids, cu, n):
    return ids[:cu[n]], cu[:n + 1]


@pytest.mark.parametrize("variant", ["st", "sharded", "bin", "rank1", "fp32"])
def test_verify_checkpoint_tool_on_hf_written_directories(variant):
    """tests/tools/verify_checkpoint.py - the one command for `LLM-ltr/OPT-Predictors` (README.md:26) once somebody has network
    access: an HF directory through load_hf_checkpoint, 64 prompts through the oracle and through MI355XRanker, max|d| / discordant
    pairs / range_fallbacks printed, exit code 0 inside north_star's 1e-4."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "tools", "verify_checkpoint.py"),
                        os.path.join(GOLDEN, "hf_tiny", variant), "-n", "64"], capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, (r.stdout + r.stderr)[-2000:]
    rec = json.loads(r.stdout[r.stdout.index("{"):])
    assert rec["verdict"] == "PASS" and rec["max_abs_err"] <= 1e-4 and rec["range_fallbacks"] == 0
    assert rec["tokens"] > 64 and "synthetic" in rec["prompts"]
