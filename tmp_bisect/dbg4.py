import os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from vllm_ltr_amd.opt_spec import OPTSpec, seeded_checkpoint
from vllm_ltr_amd.scorer import HipOPTScorer
z = np.load("tests/golden/config3_opt350m_128.npz")
spec = OPTSpec.opt_350m(); ck = seeded_checkpoint(spec, int(z["seed"]))
ids, cu, ref = z["ids"].astype(np.int64), z["cu_seqlens"], z["ref_score"]
two = HipOPTScorer(spec, ck, "cuda:0", "f16")
side = torch.cuda.Stream()
for n in (5, 8):
    i, c = ids[:cu[n]], cu[:n + 1]
    errs = [float(np.abs(two.score(i, c) - ref[:n]).max()) for _ in range(4)]
    with torch.cuda.stream(side):
        errs2 = [float(np.abs(two.score(i, c) - ref[:n]).max()) for _ in range(4)]
    print(f"n={n} T={int(c[-1])} null-stream errs {['%.1e' % e for e in errs]}  side-stream errs {['%.1e' % e for e in errs2]}", flush=True)
