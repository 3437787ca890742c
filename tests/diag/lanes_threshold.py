"""diag: where do two lanes start to pay?  One process, two handles (lanes on - probed - and LTR_F_NO_LANES), the steady call
of k arrivals (score k + re-rank the 8k queue) alternately, median of 21, for a fine grid of k around the lower end of the
lane range (1,200 tokens).   python tests/diag/lanes_threshold.py [125m|350m]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from util import bench_lengths, synthetic_batch  # noqa: E402
from vllm_ltr_amd.opt_spec import OPTSpec, seeded_checkpoint  # noqa: E402
from vllm_ltr_amd.rank import DeviceQueue  # noqa: E402
from vllm_ltr_amd.scorer import HipOPTScorer  # noqa: E402

model = sys.argv[1] if len(sys.argv) > 1 else "125m"
spec = OPTSpec.opt_125m() if model == "125m" else OPTSpec.opt_350m()
ckpt = seeded_checkpoint(spec, 0)
dev = torch.device("cuda:0")
os.environ["LTR_LANES"] = "2"                 # every call with two requests may use the lanes: the rule under test is OURS
two = HipOPTScorer(spec, ckpt, "cuda:0", "f16")
one = HipOPTScorer(spec, ckpt, "cuda:0", "f16", lanes=False)
print(model, "lane probe:", two.lane_probe())
n = 8192
queue = DeviceQueue(dev, starv=200, period=10, capacity=n)
queue.append(torch.randn(n))
need = torch.full((n,), 64, dtype=torch.int32, device=dev)
ones = torch.ones(n, dtype=torch.int32, device=dev)


def timed(sc, ids_d, cu_d, cu, k, reps=21):
    for _ in range(3):
        sc.score_device(ids_d, cu_d, cu, out=queue._score[:k])
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); sc.score_device(ids_d, cu_d, cu, out=queue._score[:k]); queue.step(need, ones, 2048, 256); b.record()
    torch.cuda.synchronize()
    return sorted(a.elapsed_time(b) for a, b in ev)[reps // 2]


for seed in (0, 5):
    for k in (4, 6, 8, 10, 12, 14, 16, 18, 20, 22, 24, 28, 32, 40, 48):
        lens = bench_lengths(max(k, 256), seed=seed)[:k]
        ids, cu = synthetic_batch(spec, lens.tolist(), 1)
        ids_d, cu_d = torch.from_numpy(ids).to(dev), torch.from_numpy(cu).to(dev)
        a = [timed(one, ids_d, cu_d, cu, k), timed(two, ids_d, cu_d, cu, k), timed(one, ids_d, cu_d, cu, k), timed(two, ids_d, cu_d, cu, k)]
        o, t = min(a[0], a[2]), min(a[1], a[3])
        print(f"{model} seed {seed} k = {k:3d} tokens = {int(cu[-1]):6d}: one lane {o:.3f} ms, two lanes {t:.3f} ms ({(t / o - 1) * 100:+.1f} %)", flush=True)
