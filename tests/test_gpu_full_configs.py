"""-m gpu: the BASELINE configurations at FULL size through the C ABI.

config 2: OPT-125m predictor, 8,192-request synthetic queue (ln 64 lengths, T = 708,977 tokens): the exact call
          `bench.py` times - four passes of <= 196,608 tokens.
config 3: OPT-350m predictor, 8,192-request "LMSYS-like" queue (ln 128 lengths).

The oracle cannot score 700k+ tokens in test time, so parity at this size is established through
size-independent properties - determinism, invariance of every score to how the batch is cut into passes
(1 pass / 4 / many), the last-layer pruning against the unpruned forward - plus an oracle check (1e-4, the
north_star tolerance) on >= 32 requests sampled across ALL passes, including the first and last request of
each pass (where the multi-pass offsets of ltr_api.hip's forward_chunk are exercised)."""
import numpy as np
import pytest
import torch

from oracle.opt_scorer import OracleOPTScorer
from util import bench_lengths
from vllm_ltr_amd.opt_spec import OPTSpec, seeded_checkpoint

pytestmark = pytest.mark.gpu
TOL = 1e-4
DEFAULT_CHUNK = 196608


def _queue(spec, n, mu):
    """bench.py's synthetic_queue (same generator, same seed 0)."""
    lens = bench_lengths(n, seed=0, mu=mu)
    g = torch.Generator().manual_seed(0)
    ids = torch.randint(4, spec.vocab_size, (int(lens.sum()),), generator=g, dtype=torch.int64).numpy()
    cu = np.zeros(n + 1, np.int32)
    np.cumsum(lens, out=cu[1:])
    ids[cu[:-1]] = 2
    return ids, cu, lens


def _passes(cu, cap):
    """Request ranges of the passes ltr_score cuts (request-aligned, <= cap tokens; ltr_api.hip run_forward)."""
    out, r0, n = [], 0, len(cu) - 1
    while r0 < n:
        r1 = r0
        while r1 < n and int(cu[r1 + 1]) - int(cu[r0]) <= cap:
            r1 += 1
        out.append((r0, r1))
        r0 = r1
    return out


def _sample(passes, n, k_extra, seed):
    idx = set()
    for r0, r1 in passes:
        idx.update((r0, r1 - 1))
    r = np.random.RandomState(seed)
    idx.update(r.randint(0, n, k_extra).tolist())
    return np.array(sorted(idx))


def _oracle_scores(spec, ckpt, ids, cu, sample):
    lens = np.diff(cu)[sample]
    ids_s = np.concatenate([ids[cu[i]:cu[i + 1]] for i in sample])
    cu_s = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    torch.set_num_threads(min(32, torch.get_num_threads()))
    return OracleOPTScorer(spec, ckpt).score_packed(ids_s, cu_s, max_tokens=2048)


def test_config2_opt125m_8k_queue_full_size():
    from vllm_ltr_amd.scorer import HipOPTScorer
    dev = "cuda:0"
    spec = OPTSpec.opt_125m()
    ckpt = seeded_checkpoint(spec, 0)
    ids, cu, lens = _queue(spec, 8192, 64.0)
    assert int(cu[-1]) == 708977                                    # the bench workload
    sc = HipOPTScorer(spec, ckpt, dev, "f16")
    ids_d, cu_d = torch.from_numpy(ids).to(dev), torch.from_numpy(cu).to(dev)
    s4 = sc.score_device(ids_d, cu_d, cu).cpu().numpy()             # default: 4 passes
    passes = _passes(cu, DEFAULT_CHUNK)
    assert len(passes) == 4
    assert np.isfinite(s4).all()
    assert np.array_equal(sc.score_device(ids_d, cu_d, cu).cpu().numpy(), s4)          # deterministic
    # how the queue is cut into passes: the token rows run the same kernels at every cut (bit-identical there); the compact
    # last-token rows of a pass (745 ... 8,192 of them) may change GEMM kernel with the pass size: <= 2e-6
    sc.set_chunk_tokens(65536)                                                          # 11 passes
    assert len(_passes(cu, 65536)) == 11
    assert np.abs(sc.score_device(ids_d, cu_d, cu).cpu().numpy() - s4).max() <= 2e-6
    sc.set_chunk_tokens(1 << 20)                                                        # ONE 709k-token pass
    assert np.abs(sc.score_device(ids_d, cu_d, cu).cpu().numpy() - s4).max() <= 2e-6
    sc.set_chunk_tokens(0)
    # last-layer pruning (ltr_score carries only the last-token rows through the tail of layer 12) against the
    # unpruned forward of a whole pass, H = 768: pool the full hidden states of pass 2 (non-zero offsets)
    r0, r1 = passes[1]
    t0, t1 = int(cu[r0]), int(cu[r1])
    cu_p = (cu[r0:r1 + 1] - t0).astype(np.int32)
    h = torch.from_numpy(sc.hidden(ids[t0:t1], cu_p, n_layers=-1)).to(dev)
    full = torch.empty(r1 - r0, device=dev)
    sc.pool_head_device(h, torch.from_numpy(cu_p).to(dev), r1 - r0, full)
    d_prune = float(np.abs(full.cpu().numpy() - s4[r0:r1]).max())   # last query: f32 VALU attention vs split-fp16 MFMA
    print(f"config 2: pruned last layer vs full forward of pass 2, max|d| = {d_prune:.3e}")
    assert d_prune <= 5e-6
    del h
    # the oracle on requests from every pass, pass boundaries included
    sample = _sample(passes, 8192, 28, seed=1)
    assert len(sample) >= 32
    want = _oracle_scores(spec, ckpt, ids, cu, sample)
    err = float(np.abs(want - s4[sample]).max())
    print(f"config 2 (OPT-125m, 8192 requests, {int(cu[-1])} tokens, 4 passes): oracle sample of {len(sample)} "
          f"requests, max|d| = {err:.3e}")
    assert err <= TOL
    # END-TO-END order at the 8k queue (north_star: "ranked order ... permutation-identical"): the HIP sort of the HIP
    # scores against the order of fp32 scores.  With ~5e-6 of score error over 8,192 scores spread over ~4 units, SOME
    # adjacent pairs must swap; the claim that can hold - and is checked - is that every swapped pair is an fp32
    # near-tie.  Oracle sample of 256 requests chosen where swaps are most likely: the 64 closest ADJACENT pairs of the
    # HIP order, plus 128 random requests.
    from util import discordant_pairs
    from vllm_ltr_amd.rank import RankWorkspace, rank_step
    sd = torch.from_numpy(s4).to(dev)
    perm = rank_step(sd, None, None, None, -1, 0, RankWorkspace(dev)).cpu().numpy()      # the order the scheduler sees
    assert np.array_equal(perm, np.argsort(-s4.astype(np.float64), kind="stable"))           # HIP sort == stable sort of its scores
    gaps = -np.diff(s4[perm].astype(np.float64))
    closest = np.argsort(gaps, kind="stable")[:64]
    pick = set(perm[closest].tolist()) | set(perm[closest + 1].tolist())
    r = np.random.RandomState(4)
    while len(pick) < 256:
        pick.add(int(r.randint(0, 8192)))
    pick = np.array(sorted(pick))
    orc = _oracle_scores(spec, ckpt, ids, cu, pick)
    err2 = float(np.abs(orc - s4[pick]).max())
    assert err2 <= TOL
    score_of = dict(zip(pick.tolist(), orc.tolist()))
    order_hip = [int(i) for i in perm if int(i) in score_of]                                 # the HIP order restricted to the sample
    order_orc = [int(pick[i]) for i in np.argsort(-orc.astype(np.float64), kind="stable")]
    d = discordant_pairs(order_orc, order_hip, score_of)
    worst = max((g for _, _, g in d), default=0.0)
    print(f"config 2 END-TO-END order: oracle sample of {len(pick)} requests (the 64 closest adjacent pairs of the HIP order, "
          f"gaps {gaps[closest].min():.2e} ... {gaps[closest].max():.2e}, + random): max|d| = {err2:.3e}; {len(d)} discordant pairs "
          f"of {len(pick) * (len(pick) - 1) // 2}, largest oracle-score gap among them {worst:.3e}")
    assert all(g <= 2 * err2 for _, _, g in d), d[:4]


def test_config3_opt350m_8k_lmsys_like_queue_full_size():
    from vllm_ltr_amd.scorer import HipOPTScorer
    dev = "cuda:0"
    spec = OPTSpec.opt_350m()
    ckpt = seeded_checkpoint(spec, 0)
    ids, cu, lens = _queue(spec, 8192, 128.0)                       # ln 64 -> ln 128
    T = int(cu[-1])
    sc = HipOPTScorer(spec, ckpt, dev, "f16")
    ids_d, cu_d = torch.from_numpy(ids).to(dev), torch.from_numpy(cu).to(dev)
    s = sc.score_device(ids_d, cu_d, cu).cpu().numpy()
    passes = _passes(cu, DEFAULT_CHUNK)
    assert len(passes) >= 6 and np.isfinite(s).all()
    assert np.array_equal(sc.score_device(ids_d, cu_d, cu).cpu().numpy(), s)           # deterministic
    sc.set_chunk_tokens(98304)                                                          # twice as many passes
    assert np.abs(sc.score_device(ids_d, cu_d, cu).cpu().numpy() - s).max() <= 2e-6
    sc.set_chunk_tokens(0)
    # permutation of the requests: the score of a request does not depend on its neighbours
    perm = np.random.RandomState(2).permutation(8192)
    ids_p = np.concatenate([ids[cu[i]:cu[i + 1]] for i in perm])
    cu_p = np.concatenate([[0], np.cumsum(lens[perm])]).astype(np.int32)
    sp = sc.score_device(torch.from_numpy(ids_p).to(dev), torch.from_numpy(cu_p).to(dev), cu_p).cpu().numpy()
    assert np.abs(sp - s[perm]).max() <= 2e-6             # (the passes cut elsewhere: other kernels for their compact rows)
    sample = _sample(passes, 8192, 16, seed=3)
    assert len(sample) >= 28
    want = _oracle_scores(spec, ckpt, ids, cu, sample)
    err = float(np.abs(want - s[sample]).max())
    print(f"config 3 (OPT-350m, 8192 requests, {T} tokens, {len(passes)} passes): oracle sample of {len(sample)} "
          f"requests, max|d| = {err:.3e}")
    assert err <= TOL


def _inversions_worst_gap(order_ref, order_got, score):
    """(number of pairs the two orders rank differently, largest |score gap| among them) - numpy, for 8k-request orders."""
    pos = np.empty(len(order_got), np.int64)
    pos[np.asarray(order_got, np.int64)] = np.arange(len(order_got))
    ref = np.asarray(order_ref, np.int64)
    p = pos[ref]                                   # position in `got` of the i-th request of `ref`
    sc = np.asarray(score, np.float64)[ref]
    n_inv, worst = 0, 0.0
    for i0 in range(0, len(ref), 512):
        blk = p[i0:i0 + 512, None] > p[None, :]    # [i, j]: j sits before i in `got` ...
        blk &= np.arange(len(ref))[None, :] > np.arange(i0, min(i0 + 512, len(ref)))[:, None]     # ... but after it in `ref`
        if blk.any():
            n_inv += int(blk.sum())
            gaps = np.abs(sc[i0:i0 + 512, None] - sc[None, :])
            worst = max(worst, float(gaps[blk].max()))
    return n_inv, worst


FULL_RUNS = {"config2": ("config2_opt125m_8192.npz", OPTSpec.opt_125m, 64.0, 708977),
             "config3": ("config3_opt350m_8192.npz", OPTSpec.opt_350m, 128.0, 1407401)}


@pytest.mark.parametrize("mode", ["f16", "f32", "f16-1pass"])
@pytest.mark.parametrize("which", list(FULL_RUNS))
def test_full_queue_against_the_reference_run(which, mode):
    """BASELINE configs 2 and 3 at FULL size against the REFERENCE itself (tests/golden/config2_opt125m_8192.npz /
    config3_opt350m_8192.npz, written by oracle/make_config1_golden.py --config 2 / 3full: one cold step of the reference's
    Scheduler with its fp32 predictor on all 8,192 requests): every HIP score within the north-star tolerance of the
    reference's, the HIP sort of the REFERENCE's scores bit-identical to the reference's order, and the end-to-end order
    (HIP scores -> HIP sort) differing from it only in fp32 near-ties.  ``mode``: the product's split-fp16 arithmetic, and the
    exact-f32 mode (f32 weights, f32 MFMA / VALU kernels); ``f16-1pass`` (LTR_F_ONE_PASS: the reference's own fp16 GPU arithmetic,
    opt-in, OUTSIDE the 1e-4 contract) only reports its distance and how much of the order it moves."""
    import hashlib
    import os
    from vllm_ltr_amd.rank import RankWorkspace, budget_prefix, rank_step
    from vllm_ltr_amd.scorer import HipOPTScorer
    name, mk_spec, mu, tokens = FULL_RUNS[which]
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name), allow_pickle=False)
    dev = torch.device("cuda:0")
    spec = mk_spec()
    ids, cu, lens = _queue(spec, 8192, mu)
    assert int(cu[-1]) == tokens and np.array_equal(cu, z["cu_seqlens"])
    assert hashlib.sha256(np.ascontiguousarray(ids.astype(np.int32)).tobytes()).digest() == z["ids_sha256"].tobytes()
    ref = z["ref_score"]
    want = z["a_order"][0]
    assert (want >= 0).all() and sorted(want.tolist()) == list(range(8192))
    sc = HipOPTScorer(spec, seeded_checkpoint(spec, int(z["seed"])), "cuda:0", mode)
    hip = sc.score(ids, cu)
    err = np.abs(hip.astype(np.float64) - ref.astype(np.float64))
    print(f"{which} [{mode}], all 8,192 requests ({int(cu[-1]):,} tokens) against the reference's fp32 predictor: max|d| = {err.max():.3e}, "
          f"rms {np.sqrt((err ** 2).mean()):.3e}")
    assert err.max() <= (2e-2 if mode == "f16-1pass" else TOL)
    ws = RankWorkspace(dev)
    perm_ref = rank_step(torch.from_numpy(ref).to(dev), None, None, None, -1, 0, ws).cpu().numpy()
    assert perm_ref.tolist() == want.tolist()                       # the reference's scores through the HIP sort: its order, bit for bit
    perm = rank_step(torch.from_numpy(hip).to(dev), None, None, None, -1, 0, ws).cpu().numpy()
    n_inv, worst = _inversions_worst_gap(want, perm, ref)
    print(f"{which} [{mode}] END-TO-END order of the 8,192-request queue against the reference's: {n_inv} discordant pairs of "
          f"{8192 * 8191 // 2}, largest reference-score gap among them {worst:.3e} (score error {err.max():.3e}; closest pair of "
          f"reference scores {np.diff(np.sort(ref.astype(np.float64))).min():.3e})")
    assert worst <= 2 * float(err.max())
    # the budget walk of that step from the reference's order and needs: its selected prefix and grants
    B, S = int(z["a_token_budget"]), int(z["a_max_num_seqs"])
    t = lambda k, dt: torch.from_numpy(z[f"a_{k}"][0].astype(dt)).to(dev)
    nsel, ran, granted = budget_prefix(torch.from_numpy(want.astype(np.int32)).to(dev), t("need_tokens", np.int32), t("need_seqs", np.int32),
                                       B, S, chunkable=t("chunkable", np.uint8))
    n = int(nsel.item())
    assert sorted(want[:n].tolist()) == np.nonzero(z["a_ran"][0])[0].tolist()
    assert granted.cpu().numpy()[want[:n]].tolist() == z["a_granted"][0][want[:n]].tolist()
