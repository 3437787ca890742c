"""diag: where a small scoring call spends its time (per kernel class, from the library's event profiler) next to
the wall time of the call.  python tests/diag/small_call_profile.py [k ...]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from util import bench_lengths, synthetic_batch  # noqa: E402
from vllm_ltr_amd.opt_spec import OPTSpec, seeded_checkpoint  # noqa: E402
from vllm_ltr_amd.scorer import HipOPTScorer  # noqa: E402

spec = OPTSpec.opt_125m()
sc = HipOPTScorer(spec, seeded_checkpoint(spec, 0), "cuda:0", "f16")
dev = torch.device("cuda:0")
ks = [int(x) for x in sys.argv[1:]] or [1, 16, 64, 256]
for k in ks:
    lens = bench_lengths(max(k, 256), seed=0)[:k]
    ids, cu = synthetic_batch(spec, lens.tolist(), 1)
    ids_d, cu_d = torch.from_numpy(ids).to(dev), torch.from_numpy(cu).to(dev)
    out = torch.empty(k, device=dev)
    for _ in range(3):
        sc.score_device(ids_d, cu_d, cu, out=out)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
    for a, b in ev:
        a.record(); sc.score_device(ids_d, cu_d, cu, out=out); b.record()
    torch.cuda.synchronize()
    wall = sorted(a.elapsed_time(b) for a, b in ev)[5]
    sc.profile(True); sc.profile_read(reset=True)
    for _ in range(5):
        sc.score_device(ids_d, cu_d, cu, out=out)
    p = sc.profile_read(reset=True)
    sc.profile(False)
    parts = {kk: (round(v["ms"] / 5 * 1e3, 1), v["launches"] // 5) for kk, v in p.items() if v["launches"]}
    tot = sum(v[0] for v in parts.values())
    print(f"k={k} tokens={int(cu[-1])}: wall {wall*1e3:.0f} us; kernel classes (us, launches): {parts}; sum {tot:.0f} us")
