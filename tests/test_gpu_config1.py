"""-m gpu: BASELINE config 1 (OPT-125m predictor, 256-request queue, T = 23,078) and config 3's predictor (OPT-350m - post-LN
blocks, project_in / project_out - on 128 LMSYS-like requests, T = 25,532) END TO END against the reference's own runs.

``tests/golden/config1_opt125m_256.npz`` (oracle/make_config1_golden.py) was recorded from the reference's own
``Scheduler`` (scheduler.py:969-1000,1101-1373) with the reference's own fp32 ``OPTForSequenceClassification`` standing
where the AUXLLM stands (aux_llm_engine.py:398-410).  Here the same steps go through ``MI355XRanker.install()``:

* scores: the HIP predictor on all 256 prompts within 1e-4 of the reference predictor's (north_star);
* order, given the reference's scores: every step of both runs bit-identical (ties by position in the concatenation);
* order END TO END - HIP scores -> HIP sort, the order a live scheduler would see: the discordant pairs against the
  reference's order are counted per step, and every one of them must be a near-tie of the reference's fp32 scores
  (gap <= 2 x the measured score error): "permutation-identical" holds wherever fp32 itself decides the order.
"""
import os
from collections import deque
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from util import GOLDEN, FakeSeqGroup, discordant_pairs
from vllm_ltr_amd.opt_spec import OPTSpec, seeded_checkpoint

pytestmark = pytest.mark.gpu
TOL = 1e-4


CASES = {"config1": ("config1_opt125m_256.npz", OPTSpec.opt_125m), "config3": ("config3_opt350m_128.npz", OPTSpec.opt_350m)}


@pytest.fixture(scope="module", params=list(CASES))
def case(request):
    from vllm_ltr_amd.scorer import HipOPTScorer
    name, mk = CASES[request.param]
    z = np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    spec = mk()
    return request.param, z, HipOPTScorer(spec, seeded_checkpoint(spec, int(z["seed"])), "cuda:0", "f16")


@pytest.fixture(scope="module")
def z(case):
    return case[1]


@pytest.fixture(scope="module")
def scorer(case):
    return case[2]


class ReplayScheduler:
    """The attribute surface of the reference's Scheduler that install() touches (pinned on the real object by
    oracle/make_config1_golden.py run c), replaying a RECORDED run: the three deques of every step and the set that ran
    come from the recording, the order comes from the ranker."""

    def __init__(self, z, tag, groups):
        self.z, self.tag, self.groups = z, tag, groups
        self.waiting, self.running, self.swapped = deque(), deque(), deque()
        self.need_score, self.starv, self.period = False, -1, 0
        self.step = 0
        self.orders = []
        self._schedule = self._general_schedule

    def load_step(self, step):
        z, tag = self.z, self.tag
        concat = z[f"{tag}_concat"][step]
        concat = concat[concat >= 0]
        nw, nr, ns = (int(x) for x in z[f"{tag}_deques"][step])
        assert nw + nr + ns == len(concat)
        g = [self.groups[int(i)] for i in concat]
        self.waiting, self.running, self.swapped = deque(g[:nw]), deque(g[nw:nw + nr]), deque(g[nw + nr:])
        self.step = step

    def _update_priority(self):
        raise AssertionError("install() must rebind _update_priority (scheduler.py:935,1002: a no-op for score-ordered policies)")

    def _get_ordered_requests(self):
        raise AssertionError("install() must rebind _get_ordered_requests")

    def _general_schedule(self):                      # the shape of scheduler.py:1101-1373 around the ordering
        self._update_priority()
        order = self._get_ordered_requests()
        self.orders.append([int(g.request_id) for g in order])
        ran = np.nonzero(self.z[f"{self.tag}_ran"][self.step])[0]
        return SimpleNamespace(scheduled_seq_groups=[SimpleNamespace(seq_group=self.groups[int(i)]) for i in ran])


def _groups(z):
    ids, cu = z["ids"].astype(np.int64), z["cu_seqlens"]
    return [FakeSeqGroup(str(i), ids[cu[i]:cu[i + 1]].tolist()) for i in range(len(cu) - 1)]


def _replay(z, tag, scorer, preset_scores):
    from vllm_ltr_amd.plugin import MI355XRanker
    starv, period = int(z[f"{tag}_starv"]), int(z[f"{tag}_period"])
    groups = _groups(z)
    if preset_scores is not None:
        for g, s in zip(groups, preset_scores):
            g.set_aux_model_score(float(s))
    ranker = MI355XRanker(scorer, f"opt-xxx-starv{starv}-period{period}", max_length=2048)
    s = ReplayScheduler(z, tag, groups)
    ranker.install(s)
    assert s.aux_model is ranker and s.need_score and s.starv == starv and s.period == period
    steps = z[f"{tag}_order"].shape[0]
    for step in range(steps):
        s.load_step(step)
        s._schedule()
        alive = list(s.waiting) + list(s.running) + list(s.swapped)
        ranker.sync_host(alive)
        st = z[f"{tag}_states"][step]
        assert all((g.pri, g.idle, g.runs) == tuple(st[int(g.request_id)]) for g in alive), (tag, step)
    return s.orders, groups, ranker


def test_config1_hip_scores_vs_reference_predictor(z, scorer):
    got = scorer.score(z["ids"].astype(np.int64), z["cu_seqlens"])
    err = np.abs(got - z["ref_score"])
    print(f"HIP predictor vs the reference's fp32 predictor over {len(got)} requests ({int(z['cu_seqlens'][-1]):,} tokens): "
          f"max|d| = {err.max():.3e}, rms {np.sqrt((err ** 2).mean()):.3e}")
    assert err.max() <= TOL


@pytest.mark.parametrize("tag", ["a", "b"])
def test_config1_order_bit_identical_given_reference_scores(z, scorer, tag):
    orders, _, ranker = _replay(z, tag, scorer, z["ref_score"])
    for step, got in enumerate(orders):
        want = z[f"{tag}_order"][step]
        assert got == want[want >= 0].tolist(), (tag, step)
    assert ranker.stats["aux_calls"] == 0                      # every score came with the request (adoption path)


@pytest.mark.parametrize("tag", ["a", "b"])
def test_config1_end_to_end_order_hip_scores_hip_sort(z, scorer, tag):
    orders, groups, ranker = _replay(z, tag, scorer, None)
    ref = z["ref_score"]
    hip = np.array([g.aux_model_score for g in groups], np.float64)
    err = float(np.abs(hip - ref).max())
    assert err <= TOL
    if tag == "a":                                             # one predictor call per arrival batch, like the reference's run
        assert ranker.stats["aux_calls"] == len(z["a_aux_calls"]) and ranker.stats["requests_scored"] == len(ref)
    n_pairs = n_disc = n_steps_diff = 0
    distinct = set()
    worst = 0.0
    for step, got in enumerate(orders):
        want = z[f"{tag}_order"][step]
        want = want[want >= 0].tolist()
        n = len(want)
        n_pairs += n * (n - 1) // 2
        if got == want:
            continue
        n_steps_diff += 1
        # discordant pairs are only meaningful inside one priority class; a pair the two orders rank differently must be
        # an fp32 near-tie: |reference score gap| <= 2 x the measured score error
        d = discordant_pairs(want, got, ref)
        n_disc += len(d)
        for a, b, gap in d:
            worst = max(worst, gap)
            distinct.add((min(a, b), max(a, b)))
            assert gap <= 2 * err, (tag, step, a, b, gap, err)
    print(f"run {tag}: END-TO-END order (HIP scores -> HIP sort) vs the reference's order over {len(orders)} steps: "
          f"{n_disc} discordant pairs of {n_pairs} ({n_steps_diff} steps differ; {len(distinct)} distinct request pairs), largest reference-score gap among them "
          f"{worst:.3e}; max|score error| {err:.3e}; closest pair of reference scores {np.diff(np.sort(ref)).min():.3e}")
    # (a near-tie is the same two requests step after step while both wait: config 3 has one pair 2.7e-6 apart)
    assert len(distinct) <= max(1, len(ref) // 100)


def test_one_pass_mode_reports_its_distance_from_the_reference(case):
    """`weight_dtype="f16-1pass"` (LTR_F_ONE_PASS): ONE fp16 MFMA pass per product in the GEMMs - the arithmetic of the
    reference's own fp16 GPU predictor (vllm/config.py:906-943, trainer.py:213-216), opt-in and OUTSIDE the 1e-4 contract.
    Against the reference's fp32 scores of the recorded run: the distance is reported (expected ~2e-3), and the cold order
    a scheduler would see (sort by -score) is compared pair by pair.  Asserted: finite, deterministic, an order of
    magnitude beyond the default mode's error (so the flag really reached the kernels) and well inside 2e-2."""
    from vllm_ltr_amd.scorer import HipOPTScorer
    name, z, two_pass = case
    spec = two_pass.spec
    sc = HipOPTScorer(spec, seeded_checkpoint(spec, int(z["seed"])), "cuda:0", "f16-1pass")
    ids, cu = z["ids"].astype(np.int64), z["cu_seqlens"]
    got = sc.score(ids, cu)
    ref = z["ref_score"]
    err = np.abs(got - ref)
    err2 = np.abs(two_pass.score(ids, cu) - ref)
    order_ref = np.argsort(-ref.astype(np.float64), kind="stable")
    order_got = np.argsort(-got.astype(np.float64), kind="stable")
    d = discordant_pairs(order_ref.tolist(), order_got.tolist(), ref)
    n = len(ref)
    print(f"{name}: one fp16 pass vs the reference's fp32 predictor over {n} requests: max|d| = {err.max():.3e}, rms "
          f"{np.sqrt((err ** 2).mean()):.3e} (two passes: max {err2.max():.3e}); cold order: {len(d)} discordant pairs of "
          f"{n * (n - 1) // 2}, largest reference-score gap among them {max((g for _, _, g in d), default=0.0):.3e}")
    assert np.isfinite(got).all() and np.array_equal(got, sc.score(ids, cu))
    assert 10 * err2.max() < err.max() <= 2e-2
    assert all(g <= 2 * err.max() for _, _, g in d)           # only pairs closer than the arithmetic's own error change places


def test_config1_tpt_class_head_end_to_end():
    """Config 1's queue under `tpt` (tests/golden/config1_tpt_class82.npz: the reference's own Scheduler with its class-mode
    predictor, 82 labels): the HIP class head (GEMM + first-maximum argmax) must give the reference's LABEL for every one
    of the 256 requests (its closest top-2 logits are 2.5e-3 apart, 250x the score error), and then the order through
    install() - (-score, request_id) on string ids, 219 requests in one tie group - is the reference's at every step, bit for
    bit, end to end."""
    from vllm_ltr_amd.plugin import MI355XRanker
    from vllm_ltr_amd.scorer import HipOPTScorer
    z = np.load(os.path.join(GOLDEN, "config1_tpt_class82.npz"), allow_pickle=False)
    q = np.load(os.path.join(GOLDEN, "config1_opt125m_256.npz"), allow_pickle=False)
    spec = OPTSpec.opt_125m(int(z["num_labels"]))
    sc = HipOPTScorer(spec, seeded_checkpoint(spec, int(z["seed"])), "cuda:0", "f16")
    labels = sc.score(q["ids"].astype(np.int64), q["cu_seqlens"])
    assert np.array_equal(labels, z["ref_score"]), np.nonzero(labels != z["ref_score"])[0][:8]
    groups = _groups(q)
    ranker = MI355XRanker(sc, "tpt-xxx", max_length=2048, mtype="class")
    zz = {k: z[k] for k in z.files}
    zz.update(a_starv=np.int64(-1), a_period=np.int64(0))
    s = ReplayScheduler(zz, "a", groups)
    ranker.install(s)
    for step in range(z["a_order"].shape[0]):
        s.load_step(step)
        s._schedule()
        want = z["a_order"][step]
        assert s.orders[-1] == want[want >= 0].tolist(), step
    assert ranker.stats["aux_calls"] == len(z["a_aux_calls"])
    assert [int(g.aux_model_score) for g in groups] == z["ref_score"].astype(int).tolist()


def test_config1_xpt_end_to_end():
    """Config 1's queue under `xpt{table}` (tests/golden/config1_xpt.npz: the reference's own Scheduler, its table lookup on
    round(-score, 2) and the SRTF key expected_length - output_len, scheduler.py:910-933, 60 steps): HIP scores -> the
    plug-in's xpt order through install().  The lookup quantises the score to two decimals: every HIP score must fall in the
    reference's table row (the closest reference score is 1.5e-5 from a rounding boundary, twice the score error), and then
    the order is the reference's at every step, bit for bit."""
    from vllm_ltr_amd.plugin import MI355XRanker
    from vllm_ltr_amd.scorer import HipOPTScorer
    z = np.load(os.path.join(GOLDEN, "config1_xpt.npz"), allow_pickle=False)
    q = np.load(os.path.join(GOLDEN, "config1_opt125m_256.npz"), allow_pickle=False)
    spec = OPTSpec.opt_125m()
    sc = HipOPTScorer(spec, seeded_checkpoint(spec, int(z["seed"])), "cuda:0", "f16")
    groups = _groups(q)
    ranker = MI355XRanker(sc, "xpt{/nonexistent/table.pt}-xxx", max_length=2048,
                          xpt_distribution=(z["xpt_key"].tolist(), z["xpt_value"].tolist()))
    zz = {k: z[k] for k in z.files}
    zz.update(a_starv=np.int64(-1), a_period=np.int64(0))
    s = ReplayScheduler(zz, "a", groups)
    ranker.install(s)
    for step in range(z["a_order"].shape[0]):
        s.load_step(step)
        for g, n in zip(groups, z["a_out_len"][step]):
            g.output_len = int(n)
        s._schedule()
        want = z["a_order"][step]
        assert s.orders[-1] == want[want >= 0].tolist(), step
    hip = np.array([g.aux_model_score for g in groups], np.float64)
    err = float(np.abs(hip - q["ref_score"]).max())
    assert err < float(z["ref_round_margin"].min()), (err, float(z["ref_round_margin"].min()))
    assert [g.expected_length for g in groups] == z["ref_expected_length"].tolist()


@pytest.mark.parametrize("name,tag", [("config1_opt125m_256.npz", "a"), ("config1_opt125m_256.npz", "b"), ("config3_opt350m_128.npz", "a"),
                                      ("config3_opt350m_128.npz", "b"), ("config1_tpt_class82.npz", "a"), ("config1_xpt.npz", "a")])
def test_budget_walk_on_the_end_to_end_runs(name, tag):
    """ltr_budget_prefix on every step of the reference's end-to-end runs (the order its scheduler saw, the per-request
    needs): the selected prefix and the granted chunk sizes of the reference's schedule()."""
    from vllm_ltr_amd.rank import budget_prefix
    dev = torch.device("cuda:0")
    z = np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    B, S = int(z[f"{tag}_token_budget"]), int(z[f"{tag}_max_num_seqs"])
    for step in range(z[f"{tag}_order"].shape[0]):
        o = z[f"{tag}_order"][step]
        o = o[o >= 0].astype(np.int32)
        if len(o) == 0:
            continue
        t = lambda k, dt: torch.from_numpy(z[f"{tag}_{k}"][step].astype(dt)).to(dev)
        nsel, ran, granted = budget_prefix(torch.from_numpy(o).to(dev), t("need_tokens", np.int32), t("need_seqs", np.int32), B, S,
                                           chunkable=t("chunkable", np.uint8))
        n = int(nsel.item())
        assert sorted(o[:n].tolist()) == np.nonzero(z[f"{tag}_ran"][step])[0].tolist(), step
        assert granted.cpu().numpy()[o[:n]].tolist() == z[f"{tag}_granted"][step][o[:n]].tolist(), step
