"""Randomised soak of scoring at arrival (not collected by pytest): random arrival patterns (lone, clustered, bursts, requests
that skip the hook), random step boundaries; every score against the oracle, every request scored once.
    python tests/diag/fuzz_prescore.py [seconds]        (LTR_FUZZ_SEED=<n> replays a run; the seed is printed)"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle.opt_scorer import OracleOPTScorer
from util import FakeSeqGroup, synthetic_batch
from vllm_ltr_amd.opt_spec import OPTSpec, seeded_checkpoint
from vllm_ltr_amd.plugin import MI355XRanker
from vllm_ltr_amd.scorer import HipOPTScorer

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(os.environ.get("LTR_FUZZ_SEED", int(time.time()) % 100000))
print(f"prescore fuzz: LTR_FUZZ_SEED={seed}", flush=True)          # a red run replays with it
r = np.random.RandomState(seed)
t0 = time.time()
n_req = n_steps = n_aborted = 0
worst = 0.0
models = []
for spec, sd in ((OPTSpec.tiny_pre_ln(), 3), (OPTSpec.tiny_post_ln(), 4)):
    ck = seeded_checkpoint(spec, sd)
    models.append((spec, HipOPTScorer(spec, ck, "cuda:0", "f16"), OracleOPTScorer(spec, ck)))
while time.time() - t0 < budget:
    spec, sc, orc = models[r.randint(0, 2)]
    rk = MI355XRanker(sc, "opt-xxx-starv5-period2", max_length=150, prescore=True, prescore_graphs=bool(r.randint(0, 2)))
    if r.rand() < 0.5:
        rk.warm_prescore_graphs()
    queue, rid = [], 0
    for step in range(int(r.randint(3, 25))):
        k = int(r.choice([0, 1, 1, 2, 3, 8, 60]))
        lens = r.randint(1, 151, k).tolist()
        ids, cu = synthetic_batch(spec, lens, int(r.randint(0, 10**6))) if k else (np.zeros(0, np.int64), np.zeros(1, np.int32))
        new = [FakeSeqGroup(str(rid + i), ids[cu[i]:cu[i + 1]].tolist()) for i in range(k)]
        rid += k
        for g in new:
            if r.rand() < 0.85:
                rk.add_request(g)
            if r.rand() < 0.5:
                time.sleep(float(r.choice([0.0, 0.0002, 0.001, 0.003])))
        # some arrivals are ABORTED before any scheduler step sees them (Scheduler.abort_seq_group): with the hook, or
        # silently (their record is dropped by the sweep: forced here by ageing the records)
        keep = np.ones(k, bool)
        if k and r.rand() < 0.3:
            keep = r.rand(k) > 0.3
            for g, kp in zip(new, keep):
                if not kp and r.rand() < 0.5:
                    rk.abort_request(g)
            n_aborted += int((~keep).sum())
            if r.rand() < 0.5:
                for rec in rk._pre_inflight:
                    rec["t"] -= 2 * rk.PRESCORE_ORPHAN_S
        alive = [g for g, kp in zip(new, keep) if kp]
        if alive:
            got = np.array(rk.obtain_aux_scores(alive))
            want = orc.score(ids, cu)[keep]
            err = float(np.abs(got - want).max()); worst = max(worst, err)
            assert err <= 1e-4, (step, lens, err)
            n_req += len(alive)
        new = alive
        queue += new
        order = rk.order(queue)
        assert sorted(int(g.request_id) for g in order) == sorted(int(g.request_id) for g in queue)
        ran = order[: int(r.randint(0, 4))]
        rk.age(queue, ran)
        if r.rand() < 0.3 and queue:
            queue = queue[int(r.randint(0, len(queue))):]        # some requests finish
        n_steps += 1
    torch.cuda.synchronize()
    rk._prescore_sweep(time.perf_counter() + 2 * rk.PRESCORE_ORPHAN_S)
    assert len(rk._pre_inflight) == 0, len(rk._pre_inflight)          # nothing is kept alive by an aborted request
    assert len(rk._pre_free_stagers) <= rk.PRESCORE_MAX_STAGERS
print(f"prescore fuzz ok: {n_req} requests ({n_aborted} more aborted before their step) over {n_steps} scheduler steps (worst |score - oracle| {worst:.2e}) in {time.time() - t0:.0f} s")
