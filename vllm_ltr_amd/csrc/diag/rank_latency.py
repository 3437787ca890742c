"""Diagnostic: latency of the steady scheduler step (ltr_queue_step) per queue size, HIP events over 200 calls.
    LTR_RANK_MODE=0|1 LTR_RANK_NW=4|8|16 python vllm_ltr_amd/csrc/diag/rank_latency.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))
from vllm_ltr_amd.rank import DeviceQueue  # noqa: E402

dev = torch.device("cuda:0")
out = []
for n in (1024, 4096, 8192, 12288, 16384, 65536):
    r = np.random.RandomState(0)
    q = DeviceQueue(dev, starv=200, period=10, capacity=n)
    q.append(torch.from_numpy(r.standard_normal(n).astype(np.float16).astype(np.float32)))
    need = torch.from_numpy(r.randint(4, 300, n).astype(np.int32)).to(dev)
    seqs = torch.ones(n, dtype=torch.int32, device=dev)
    perm = torch.empty(n, dtype=torch.int32, device=dev)
    for _ in range(20):
        q.step(need, seqs, 2048, 256, perm_out=perm)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(200)]
    for a, b in ev:
        a.record(); q.step(need, seqs, 2048, 256, perm_out=perm); b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in ev)
    out.append(f"{n}:{t[len(t)//2]*1e3:.1f}us")
print(f"mode={os.environ.get('LTR_RANK_MODE','0')} nw={os.environ.get('LTR_RANK_NW','16')}  " + "  ".join(out))
