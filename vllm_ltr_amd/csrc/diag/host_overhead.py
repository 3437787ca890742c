"""Diagnostic: wall time of MI355XRanker.obtain_aux_scores (host pipeline + GPU) against the GPU
time of the same cold call, 8k-request BASELINE queue."""
import sys, time
import numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from bench import synthetic_queue
from util import FakeSeqGroup
from vllm_ltr_amd.opt_spec import OPTSpec, seeded_checkpoint
from vllm_ltr_amd.plugin import MI355XRanker
from vllm_ltr_amd.scorer import HipOPTScorer
spec = OPTSpec.opt_125m()
sc = HipOPTScorer(spec, seeded_checkpoint(spec, 0), device="cuda:0")
ranker = MI355XRanker(sc, "opt-xxx-starv200-period10", max_length=2048)
ids, cu, lens = synthetic_queue(spec, 8192, 0)
def groups():
    return [FakeSeqGroup(str(i), ids[cu[i]:cu[i + 1]].tolist()) for i in range(len(lens))]
ranker.obtain_aux_scores(groups())            # warm-up
for label, pre in (("tokenised inside the call", False), ("ids cached at add_request", True)):
    g = groups()
    if pre:
        for x in g:
            ranker.add_request(x)
    torch.cuda.synchronize(); t = time.perf_counter()
    ranker.obtain_aux_scores(g)
    dt = time.perf_counter() - t
    print(f"obtain_aux_scores, {label}: {dt * 1e3:.1f} ms")
ids_d = torch.from_numpy(ids).cuda(); cu_d = torch.from_numpy(cu).cuda()
torch.cuda.synchronize(); t = time.perf_counter()
sc.score_device(ids_d, cu_d, cu); torch.cuda.synchronize()
print(f"ltr_score with inputs resident in HBM: {(time.perf_counter() - t) * 1e3:.1f} ms")
