"""Randomised parity soak (not collected by pytest): HIP kernels vs the oracle on many random shapes.
    python tests/diag/fuzz_gpu.py [seconds]        (LTR_FUZZ_SEED=<n> replays a run; the seed is printed)"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import rank_step as rs
from oracle.opt_scorer import OracleOPTScorer
from util import synthetic_batch
from vllm_ltr_amd.opt_spec import OPTSpec, seeded_checkpoint
from vllm_ltr_amd.rank import RankWorkspace, age_update, budget_prefix, rank_step, reserve_select
from vllm_ltr_amd.scorer import HipOPTScorer

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
dev = "cuda:0"
seed = int(os.environ.get("LTR_FUZZ_SEED", int(time.time()) % 100000))
print(f"fuzz: LTR_FUZZ_SEED={seed}", flush=True)          # a red run replays with it
r = np.random.RandomState(seed)
t = lambda a: torch.from_numpy(a).to(dev)
t0 = time.time(); n_rank = n_res = n_bud = n_score = 0
ws = RankWorkspace(torch.device(dev))
specs = [(OPTSpec.tiny_pre_ln(), 3), (OPTSpec.tiny_post_ln(), 4)]
scorers = [(s, seeded_checkpoint(s, sd)) for s, sd in specs]
scorers = [(s, c, OracleOPTScorer(s, c)) for s, c in scorers]
worst = 0.0
while time.time() - t0 < budget:
    # ---- rank step with heavy ties / specials
    n = int(r.choice([1, 2, 63, 64, 65, 1000, 4096, 12287, 12288, 12289, 20000, int(r.randint(1, 70000))]))
    kind = r.randint(0, 4)
    sc = r.standard_normal(n).astype(np.float32)
    if kind == 1: sc = np.round(sc * 2) / 2
    if kind == 2: sc[r.randint(0, n, max(1, n // 10))] = r.choice([np.inf, -np.inf, 0.0, -0.0], max(1, n // 10))
    if kind == 3: sc[:] = 1.0
    starv = int(r.choice([-1, 0, 1, 5, 200])); period = int(r.randint(1, 12))
    pri = r.choice([-1, 0], n).astype(np.int32); idle = r.randint(0, 8, n).astype(np.int32); runs = r.randint(0, 4, n).astype(np.int32)
    want = rs.rank_step_np(sc, pri.copy(), idle.copy(), runs.copy(), starv, period)
    p_d, i_d, r_d = t(pri.copy()), t(idle.copy()), t(runs.copy())
    got = rank_step(t(sc), p_d if starv != -1 else None, i_d if starv != -1 else None, r_d if starv != -1 else None,
                    starv, period, ws).cpu().numpy()
    assert (got == want).all(), ("rank", n, kind, starv, period)
    n_rank += 1
    # ---- budget prefix + reserve select
    perm = r.permutation(n).astype(np.int32)
    need = r.randint(0 if r.rand() < 0.2 else 1, 64, n).astype(np.int32); seqs = r.randint(1, 3, n).astype(np.int32)
    B, S = int(r.randint(1, 5000)), int(r.randint(1, 400))
    nsel, ran, granted = budget_prefix(t(perm), t(need), t(seqs), B, S)
    wn, wg = rs.budget_walk(need[perm], seqs[perm], B, S)
    assert int(nsel.item()) == wn and granted.cpu().numpy()[perm[:wn]].tolist() == wg, ("budget", n, B, S)
    n_bud += 1
    state = r.randint(0, 3, n).astype(np.uint8); phys = r.randint(0, 20, n).astype(np.int32); logical = r.randint(1, 20, n).astype(np.int32)
    nrun = (state == 1).astype(np.int32) * r.randint(1, 3, n).astype(np.int32); nswap = (state == 2).astype(np.int32)
    needb = int(r.randint(-5, 1 + 3 * n))
    act, nexec, _ = reserve_select(t(perm), nsel, t(state), t(phys), t(logical), t(nrun), t(nswap), needb)
    wa, wne = rs.reserve_select(perm, wn, state, phys, logical, nrun, nswap, needb)
    assert act.cpu().numpy().tolist() == wa.tolist() and int(nexec.item()) == wne, ("reserve", n, needb)
    n_res += 1
    # ---- scorer on random ragged batches with tiny chunks (pass boundaries)
    if n_rank % 4 == 0:
        spec, ck, orc = scorers[r.randint(0, len(scorers))]
        lens = r.randint(1, 150, r.randint(1, 40 if r.rand() < 0.5 else 160)).tolist()
        ids, cu = synthetic_batch(spec, lens, int(r.randint(0, 10**6)))
        hs = HipOPTScorer(spec, ck, device=dev, weight_dtype="f16", chunk_tokens=int(r.choice([0, 160, 256, 1000])))
        got = hs.score(ids, cu); want = orc.score(ids, cu)
        err = float(np.abs(got - want).max()); worst = max(worst, err)
        assert err <= 1e-4, ("score", lens, err)
        n_score += 1
print(f"fuzz ok: {n_rank} rank steps, {n_bud} budget walks, {n_res} eviction choices, {n_score} scoring calls "
      f"(worst |d| {worst:.2e}) in {time.time() - t0:.0f} s")
