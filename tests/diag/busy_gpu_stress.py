"""diag: soak of the scoring forward beside a busy GPU (not collected by pytest; tests/test_gpu_small_batches.py runs it).  Random
batches of 2 ... 96 requests, both model families at true width: the call on the default stream, and on a SIDE stream with its own
scratch (what ``MI355XRanker(prescore=True)`` does at ``add_request``), three times each, beside an unrelated stream of LIBRARY
GEMMs (fp16 and bf16 ``torch.matmul``: what a serving engine's backbone runs beside the ranker, and the co-runner that exposes the
packed-f32 hazard of profiles/r06_rln_fault.txt) - every result BIT-identical to the same call on an idle device.
    LTR_FUZZ_SEED=<n> python tests/diag/busy_gpu_stress.py [seconds]        (the seed is printed; a red run replays with it)"""
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
print(f"busy-GPU stress: LTR_FUZZ_SEED={seed}", flush=True)
r = np.random.RandomState(seed)
dev = torch.device("cuda:0")
models = []
for spec in (OPTSpec.opt_125m(), OPTSpec.opt_350m()):
    models.append((spec, HipOPTScorer(spec, seeded_checkpoint(spec, 0), "cuda:0", "f16")))
noise_stream = torch.cuda.Stream()
side_stream = torch.cuda.Stream()
a16 = torch.randn(2048, 2048, device=dev, dtype=torch.float16)
b16 = torch.randn(2048, 2048, device=dev, dtype=torch.bfloat16)
t0 = time.time()
n_calls = n_side = n_noisy = 0
it = 0
while time.time() - t0 < budget:
    it += 1
    spec, sc = models[r.randint(0, 2)]
    k = int(r.choice([1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 16, 24, 32, 48, 64, 96]))
    lens = bench_lengths(k, seed=int(r.randint(0, 10**6)), mu=float(r.choice([40.0, 64.0, 128.0, 300.0])))
    ids, cu = synthetic_batch(spec, lens.tolist(), int(r.randint(0, 10**6)))
    noisy = r.rand() < 0.7
    on_side = r.rand() < 0.4
    want = sc.score(ids, cu)                        # idle device
    if noisy:
        with torch.cuda.stream(noise_stream):
            for _ in range(25):
                a16 = (a16 @ a16).clamp_(-1, 1)
                b16 = (b16 @ b16).clamp_(-1, 1)
    if on_side:                                     # the prescore path: another stream, scratch of its own
        ids_d, cu_d = torch.from_numpy(ids).to(dev), torch.from_numpy(cu).to(dev)
        side_stream.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side_stream):
            outs = [sc.score_device(ids_d, cu_d, cu, workspace_key="stress-side").clone() for _ in range(3)]
        side_stream.synchronize()
        got = [o.cpu().numpy() for o in outs]
        n_side += 3
    else:
        got = [sc.score(ids, cu) for _ in range(3)]
    ctx = (seed, it, spec.hidden_size, k, int(cu[-1]), noisy, on_side)
    for g in got:
        assert np.array_equal(g, want), ctx + ("differs from the idle call", float(np.abs(g - want).max()))
    n_calls += 3
    n_noisy += 3 * noisy
    noise_stream.synchronize()
print(f"busy-GPU stress ok: seed {seed}, {n_calls} calls ({n_noisy} beside library GEMMs, {n_side} on a side stream), all bit-identical "
      f"to the idle call, {time.time() - t0:.0f} s")
