"""diag: is a small scoring call bound by the host (launch issue) or by the GPU?  Host time of ltr_score (the ctypes call
returns when everything is enqueued) next to the event-timed GPU span of the same call.
python tests/diag/host_issue_time.py [125m|350m] [k ...]"""
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

model = sys.argv[1] if len(sys.argv) > 1 else "125m"
spec = OPTSpec.opt_125m() if model == "125m" else OPTSpec.opt_350m()
sc = HipOPTScorer(spec, seeded_checkpoint(spec, 0), "cuda:0", "f16")
dev = torch.device("cuda:0")
for k in [int(x) for x in sys.argv[2:]] or [1, 4, 16, 64]:
    lens = bench_lengths(max(k, 256), seed=0)[:k]
    ids, cu = synthetic_batch(spec, lens.tolist(), 1)
    ids_d, cu_d = torch.from_numpy(ids).to(dev), torch.from_numpy(cu).to(dev)
    out = torch.empty(k, device=dev)
    for _ in range(5):
        sc.score_device(ids_d, cu_d, cu, out=out)
    torch.cuda.synchronize()
    host, gpu = [], []
    for _ in range(30):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record()
        t0 = time.perf_counter()
        sc.score_device(ids_d, cu_d, cu, out=out)
        t1 = time.perf_counter()
        b.record()
        torch.cuda.synchronize()
        host.append((t1 - t0) * 1e6)
        gpu.append(a.elapsed_time(b) * 1e3)
    # the same with the GPU kept busy in front of the call: the host runs ahead, the GPU span is what the kernels need
    busy = torch.empty(64 << 20, device=dev)
    gpu2 = []
    for _ in range(30):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        for _ in range(40):
            busy.mul_(1.0001)                      # ~2 ms of queued work
        a.record()
        sc.score_device(ids_d, cu_d, cu, out=out)
        b.record()
        torch.cuda.synchronize()
        gpu2.append(a.elapsed_time(b) * 1e3)
    med = lambda v: sorted(v)[len(v) // 2]
    print(f"{model} k={k} T={int(cu[-1])}: host issue {med(host):.0f} us, GPU span {med(gpu):.0f} us, "
          f"GPU span behind 2 ms of queued work (host ahead) {med(gpu2):.0f} us")
