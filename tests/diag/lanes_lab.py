"""diag: wall time of ltr_score on the first k requests of the bench queue, for the lab of the two-lane forward
(ltr_api.hip run_forward).  LTR_LANES=0|2 python tests/diag/lanes_lab.py <125m|350m> <out.npz> [k ...]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from util import bench_lengths, synthetic_batch  # noqa: E402
from vllm_ltr_amd.opt_spec import OPTSpec, seeded_checkpoint  # noqa: E402
from vllm_ltr_amd.scorer import HipOPTScorer  # noqa: E402

model = sys.argv[1]
spec = OPTSpec.opt_125m() if model == "125m" else OPTSpec.opt_350m()
sc = HipOPTScorer(spec, seeded_checkpoint(spec, 0), "cuda:0", "f16")
dev = torch.device("cuda:0")
ks = [int(x) for x in sys.argv[3:]] or [2, 4, 8, 16, 32, 64, 128, 256, 512]
res, scores = {}, {}
for k in ks:
    lens = bench_lengths(max(k, 256), seed=0)[:k]
    ids, cu = synthetic_batch(spec, lens.tolist(), 1)
    ids_d, cu_d = torch.from_numpy(ids).to(dev), torch.from_numpy(cu).to(dev)
    out = torch.empty(k, device=dev)
    for _ in range(3):
        sc.score_device(ids_d, cu_d, cu, out=out)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(15)]
    for a, b in ev:
        a.record(); sc.score_device(ids_d, cu_d, cu, out=out); b.record()
    torch.cuda.synchronize()
    wall = sorted(a.elapsed_time(b) for a, b in ev)[7]
    res[k] = (int(cu[-1]), wall)
    scores[f"k{k}"] = out.cpu().numpy()
print(f"{model} LTR_LANES={os.environ.get('LTR_LANES', '-')}: " + "  ".join(f"k={k} T={t}: {w*1e3:.0f} us" for k, (t, w) in res.items()))
np.savez(sys.argv[2], **scores)
