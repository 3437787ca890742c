"""diag: soak of the scoring forward under concurrency (not collected by pytest; tests/test_gpu_small_batches.py runs it).  Random
batches of 2 ... 96 requests, both model families at true width, three ways against a one-lane handle on an idle stream:
  * the lanes handle (two halves on two streams) on the default stream,
  * the lanes handle on a SIDE stream with its own scratch (what ``MI355XRanker(prescore=True)`` does at ``add_request``),
  * the one-lane handle itself while the GPU is busy,
equal to the few 1e-6 of the batch-size regimes (bound 8e-6), each deterministic over repeats - half of the time beside an
unrelated stream of LIBRARY GEMMs (fp16 and bf16 ``torch.matmul``: what a serving engine's backbone runs beside the ranker, and
the co-runner that exposes the packed-f32 hazard of profiles/r06_rln_fault.txt).
    LTR_FUZZ_SEED=<n> python tests/diag/lanes_stress.py [seconds]        (the seed is printed; a red run replays with it)"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from util import bench_lengths, synthetic_batch  # noqa: E402
from vllm_ltr_amd.opt_spec import OPTSpec, seeded_checkpoint  # noqa: E402
from vllm_ltr_amd.scorer import HipOPTScorer  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(os.environ.get("LTR_FUZZ_SEED", int(time.time()) % 100000))
print(f"lanes stress: LTR_FUZZ_SEED={seed}", flush=True)
r = np.random.RandomState(seed)
dev = torch.device("cuda:0")
models = []
for spec in (OPTSpec.opt_125m(), OPTSpec.opt_350m()):
    ck = seeded_checkpoint(spec, 0)
    models.append((spec, HipOPTScorer(spec, ck, "cuda:0", "f16"), HipOPTScorer(spec, ck, "cuda:0", "f16", lanes=False)))
noise_stream = torch.cuda.Stream()
side_stream = torch.cuda.Stream()
a16 = torch.randn(2048, 2048, device=dev, dtype=torch.float16)
b16 = torch.randn(2048, 2048, device=dev, dtype=torch.bfloat16)
t0 = time.time()
n_calls = n_lane = n_side = 0
worst = 0.0
it = 0
while time.time() - t0 < budget:
    it += 1
    spec, two, one = models[r.randint(0, 2)]
    k = int(r.choice([2, 3, 4, 5, 6, 7, 8, 10, 12, 16, 24, 32, 48, 64, 96]))
    lens = bench_lengths(k, seed=int(r.randint(0, 10**6)), mu=float(r.choice([40.0, 64.0, 128.0, 300.0])))
    ids, cu = synthetic_batch(spec, lens.tolist(), int(r.randint(0, 10**6)))
    noisy = r.rand() < 0.5
    on_side = r.rand() < 0.35
    want = one.score(ids, cu)                       # idle device
    if noisy:
        with torch.cuda.stream(noise_stream):
            for _ in range(25):
                a16 = (a16 @ a16).clamp_(-1, 1)
                b16 = (b16 @ b16).clamp_(-1, 1)
    before = two.lane_calls()
    if on_side:                                     # the prescore path: another stream, scratch of its own
        ids_d, cu_d = torch.from_numpy(ids).to(dev), torch.from_numpy(cu).to(dev)
        side_stream.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side_stream):
            outs = [two.score_device(ids_d, cu_d, cu, workspace_key="stress-side").clone() for _ in range(3)]
        side_stream.synchronize()
        got = [o.cpu().numpy() for o in outs]
        n_side += 3
    else:
        got = [two.score(ids, cu) for _ in range(3)]
    used = two.lane_calls() - before
    busy = one.score(ids, cu) if noisy else want    # the one-lane handle beside the library GEMMs
    scale = max(1.0, float(np.abs(want).max()))
    err = max(float(np.abs(g - want).max()) for g in got)
    worst = max(worst, err / scale)
    ctx = (seed, it, spec.hidden_size, k, int(cu[-1]), used, noisy, on_side)
    assert err <= 8e-6 * scale, ctx + (err,)
    assert all(np.array_equal(got[0], g) for g in got[1:]), ctx + ("not deterministic",)
    assert np.array_equal(busy, want), ctx + ("one-lane call differs beside a busy GPU", float(np.abs(busy - want).max()))
    n_calls += 3
    n_lane += used
    noise_stream.synchronize()
print(f"lanes stress ok: seed {seed}, {n_calls} calls ({n_lane} on two lanes, {n_side} on a side stream), worst |two lanes - one lane| / scale "
      f"{worst:.2e}, {time.time() - t0:.0f} s")
