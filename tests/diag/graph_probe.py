"""diag: does a captured graph of a small scoring call run faster on the GPU than its launches issued one by one?
python tests/diag/graph_probe.py [k ...]"""
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

spec = OPTSpec.opt_125m()
sc = HipOPTScorer(spec, seeded_checkpoint(spec, 0), "cuda:0", "f16")
dev = torch.device("cuda:0")
for k in [int(x) for x in sys.argv[1:]] or [1, 4, 16]:
    lens = bench_lengths(max(k, 256), seed=0)[:k]
    ids, cu = synthetic_batch(spec, lens.tolist(), 1)
    ids_d, cu_d = torch.from_numpy(ids).to(dev), torch.from_numpy(cu).to(dev)
    out = torch.empty(k, device=dev)
    s = torch.cuda.Stream(dev)
    with torch.cuda.stream(s):
        for _ in range(5):
            sc.score_device(ids_d, cu_d, cu, out=out)
    torch.cuda.synchronize()
    want = out.clone()

    def timed(fn, n=30):
        ts = []
        for _ in range(n):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            with torch.cuda.stream(s):
                a.record(s); fn(); b.record(s)
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) * 1e3)
        return sorted(ts)[len(ts) // 2]
    eager = timed(lambda: sc.score_device(ids_d, cu_d, cu, out=out))
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(g, stream=s):
            sc.score_device(ids_d, cu_d, cu, out=out)
        out.zero_()
        g.replay(); torch.cuda.synchronize()
        ok = bool(torch.equal(out, want))
        t0 = time.perf_counter()
        for _ in range(50):
            g.replay()
        host = (time.perf_counter() - t0) / 50 * 1e6
        torch.cuda.synchronize()
        graph = timed(lambda: g.replay())
        print(f"k={k} T={int(cu[-1])}: eager GPU span {eager:.0f} us; graph replay GPU span {graph:.0f} us (host {host:.0f} us per replay), same scores: {ok}")
    except Exception as e:      # noqa: BLE001
        print(f"k={k}: capture failed: {type(e).__name__}: {str(e)[:300]}")
