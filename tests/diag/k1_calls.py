"""diag: n back-to-back scoring calls of the first k requests of the bench queue (for a rocprofv3 kernel trace of the call:
profiles/k1_timeline.py prints the last call kernel by kernel).   python tests/diag/k1_calls.py [k] [n] [model]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from util import bench_lengths, synthetic_batch  # noqa: E402
from vllm_ltr_amd.opt_spec import OPTSpec, seeded_checkpoint  # noqa: E402
from vllm_ltr_amd.scorer import HipOPTScorer  # noqa: E402

k = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n = int(sys.argv[2]) if len(sys.argv) > 2 else 18
spec = OPTSpec.opt_350m() if len(sys.argv) > 3 and sys.argv[3] == "350m" else OPTSpec.opt_125m()
sc = HipOPTScorer(spec, seeded_checkpoint(spec, 0), "cuda:0", "f16")
dev = torch.device("cuda:0")
lens = bench_lengths(max(k, 256), seed=0)[:k]
ids, cu = synthetic_batch(spec, lens.tolist(), 1)
ids_d, cu_d = torch.from_numpy(ids).to(dev), torch.from_numpy(cu).to(dev)
out = torch.empty(k, device=dev)
for _ in range(n):
    sc.score_device(ids_d, cu_d, cu, out=out)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record(); sc.score_device(ids_d, cu_d, cu, out=out); b.record()
torch.cuda.synchronize()
print(f"k={k} tokens={int(cu[-1])}: one more call between events {a.elapsed_time(b) * 1e3:.0f} us")
