"""diag: soak of the two-lane forward under concurrency (not collected by pytest).  Random batches of 2 ... 96 requests, both
model families at true width, the lanes handle (two halves on two streams) against a one-lane handle of the same checkpoint:
equal to the few 1e-6 of the batch-size regimes (bound 8e-6), the lanes handle deterministic over repeats - with and without an unrelated
stream that keeps the GPU busy (a serving engine's backbone kernels run beside the ranker's).
    python tests/diag/lanes_stress.py [seconds]"""
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
r = np.random.RandomState(int(time.time()) % 100000)
dev = torch.device("cuda:0")
models = []
for spec in (OPTSpec.opt_125m(), OPTSpec.opt_350m()):
    ck = seeded_checkpoint(spec, 0)
    models.append((spec, HipOPTScorer(spec, ck, "cuda:0", "f16"), HipOPTScorer(spec, ck, "cuda:0", "f16", lanes=False)))
noise_stream = torch.cuda.Stream()
a = torch.randn(2048, 2048, device=dev, dtype=torch.float16)
t0 = time.time()
n_calls = n_lane = 0
worst = 0.0
while time.time() - t0 < budget:
    spec, two, one = models[r.randint(0, 2)]
    k = int(r.choice([2, 3, 4, 5, 6, 7, 8, 10, 12, 16, 24, 32, 48, 64, 96]))
    lens = bench_lengths(k, seed=int(r.randint(0, 10**6)), mu=float(r.choice([40.0, 64.0, 128.0, 300.0])))
    ids, cu = synthetic_batch(spec, lens.tolist(), int(r.randint(0, 10**6)))
    noisy = r.rand() < 0.5
    if noisy:
        with torch.cuda.stream(noise_stream):
            for _ in range(40):
                a = (a @ a).clamp_(-1, 1)
    want = one.score(ids, cu)
    before = two.lane_calls()
    got = [two.score(ids, cu) for _ in range(3)]
    used = two.lane_calls() - before
    scale = max(1.0, float(np.abs(want).max()))
    err = max(float(np.abs(g - want).max()) for g in got)
    worst = max(worst, err / scale)
    assert err <= 8e-6 * scale, (spec.hidden_size, k, int(cu[-1]), used, noisy, err)
    assert all(np.array_equal(got[0], g) for g in got[1:]), (spec.hidden_size, k, int(cu[-1]), used, "not deterministic")
    n_calls += 3
    n_lane += used
    noise_stream.synchronize()
print(f"lanes stress ok: {n_calls} calls ({n_lane} on two lanes), worst |two lanes - one lane| / scale {worst:.2e}, {time.time() - t0:.0f} s")
