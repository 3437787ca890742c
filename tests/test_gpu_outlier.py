"""-m gpu: checkpoints with the structure of TRAINED OPT weights, scored by the reference itself (VERDICT r4 missing #4).

Every other reference-run fixture uses N(0, 0.02) weights - benign LayerNorm statistics, scores within +-2.  Trained OPT
predictors carry massive-activation channels and LayerNorm gains spread over an order of magnitude.
``tests/golden/outlier_opt125m_64.npz`` / ``outlier_opt350m_48.npz`` (oracle/make_config1_golden.py --config outlier /
outlier350) hold what the reference's own fp32 ``OPTForSequenceClassification`` (opt.py:362-444) computed for
``opt_spec.structured_checkpoint`` - two embedding channels at +40 / -55, LayerNorm gains in [0.2, 3], the pre-LN family also
with weights at 5x the init scale - on 64 / 48 requests incl. L = 1, 2 and 1024, and the order its Scheduler returned."""
import os
from collections import deque

import numpy as np
import pytest

from util import GOLDEN, FakeSeqGroup, discordant_pairs
from vllm_ltr_amd.opt_spec import OPTSpec, STRUCTURED_350M, structured_checkpoint

pytestmark = pytest.mark.gpu

FAMILIES = {"125m": ("outlier_opt125m_64.npz", OPTSpec.opt_125m, {}), "350m": ("outlier_opt350m_48.npz", OPTSpec.opt_350m, STRUCTURED_350M)}


@pytest.fixture(scope="module", params=list(FAMILIES))
def case(request):
    name, mk, kw = FAMILIES[request.param]
    z = np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    spec = mk()
    return request.param, z, spec, structured_checkpoint(spec, int(z["seed"]), **kw)


# (f32 against the reference's fp32: the reference's own scores are 4.6e-6 from an f64 evaluation of the same checkpoint -
# oracle/opt_scorer.py in float64 - and the exact-f32 HIP path is a different summation order: two fp32 evaluations of these
# checkpoints differ by ~1e-5 (measured 1.14e-5), so the f32 bound is 2e-5; the split-fp16 production path holds 1e-4.)
@pytest.mark.parametrize("mode,tol", [("f16", 1e-4), ("f32", 2e-5)])
def test_structured_checkpoint_scores_vs_the_reference(case, mode, tol):
    from vllm_ltr_amd.scorer import HipOPTScorer
    family, z, spec, ckpt = case
    if mode == "f32":
        ckpt = {k: v.astype(np.float32) for k, v in ckpt.items()}
    sc = HipOPTScorer(spec, ckpt, "cuda:0", mode)
    ids, cu, ref = z["ids"].astype(np.int64), z["cu_seqlens"], z["ref_score"]
    got = sc.score(ids, cu)                                     # (check_status inside: the folded operand stayed in range)
    err = np.abs(got - ref)
    bound = tol * max(1.0, float(np.abs(ref).max()))
    print(f"OPT-{family} {mode}: HIP vs the reference's fp32 predictor on the structured checkpoint, {len(ref)} requests "
          f"({int(cu[-1])} tokens): max|d| = {err.max():.3e}, rms {np.sqrt((err ** 2).mean()):.3e} (bound {bound:.1e}); "
          f"scores in [{ref.min():.4f}, {ref.max():.4f}]")
    assert np.isfinite(got).all() and err.max() <= bound
    # each request scored ALONE (the small-batch kernels, one lane): the same contract
    for i in (0, 1, 2, 3, len(ref) - 1):
        one = sc.score(ids[cu[i]:cu[i + 1]], np.array([0, cu[i + 1] - cu[i]], np.int32))[0]
        assert abs(one - ref[i]) <= bound, (i, one, ref[i])


def test_structured_checkpoint_order_end_to_end(case):
    """HIP scores -> HIP sort through install() against the order the reference's Scheduler returned for ITS scores: the pairs
    the two orders rank differently must be fp32 near-ties (reference-score gap <= 2 x the measured score error; the closest
    pairs of reference scores are 6.6e-5 / 3.6e-5 apart)."""
    from vllm_ltr_amd.plugin import MI355XRanker
    from vllm_ltr_amd.scorer import HipOPTScorer
    family, z, spec, ckpt = case
    sc = HipOPTScorer(spec, ckpt, "cuda:0", "f16")
    ids, cu, ref = z["ids"].astype(np.int64), z["cu_seqlens"], z["ref_score"]
    groups = [FakeSeqGroup(str(i), ids[cu[i]:cu[i + 1]].tolist()) for i in range(len(ref))]
    rk = MI355XRanker(sc, "opt-xxx", max_length=2048)

    class Sched:
        pass
    s = Sched()
    s.waiting, s.running, s.swapped = deque(groups), deque(), deque()
    rk.install(s)
    got = [int(g.request_id) for g in s._get_ordered_requests()]
    hip = np.array([g.aux_model_score for g in groups], np.float64)
    err = float(np.abs(hip - ref).max())
    want = z["a_order"][0]
    want = want[want >= 0].tolist()
    d = discordant_pairs(want, got, ref)
    print(f"OPT-{family} structured checkpoint, cold step: {len(d)} discordant pairs of {len(ref) * (len(ref) - 1) // 2}, largest "
          f"reference-score gap among them {max((g for _, _, g in d), default=0.0):.3e}; max|score error| {err:.3e}; range fallbacks "
          f"{rk.metrics()['range_fallbacks']}")
    assert err <= 1e-4 and all(gap <= 2 * err for _, _, gap in d)


def test_structured_checkpoint_one_pass_report(case):
    """The opt-in one-pass mode (the reference's fp16 GPU arithmetic) on the structured checkpoints: finite - the plain-fp16
    operands and the folded LayerNorm operand stay in range, or the plug-in falls back - and its distance from the fp32 scores
    is reported (no contract: outside 1e-4 by design)."""
    from vllm_ltr_amd.plugin import MI355XRanker
    from vllm_ltr_amd.scorer import HipOPTScorer
    family, z, spec, ckpt = case
    ids, cu, ref = z["ids"].astype(np.int64), z["cu_seqlens"], z["ref_score"]
    rk = MI355XRanker(HipOPTScorer(spec, ckpt, "cuda:0", "f16-1pass"), "opt-xxx", max_length=2048)
    groups = [FakeSeqGroup(str(i), ids[cu[i]:cu[i + 1]].tolist()) for i in range(len(ref))]
    got = np.array(rk.obtain_aux_scores(groups))
    err = np.abs(got - ref)
    print(f"OPT-{family} structured checkpoint, one fp16 pass: max|d| = {err.max():.3e}, rms {np.sqrt((err ** 2).mean()):.3e}; "
          f"range fallbacks {rk.metrics()['range_fallbacks']}")
    assert np.isfinite(got).all() and err.max() <= 5e-2
