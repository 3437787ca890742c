import os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from vllm_ltr_amd.opt_spec import OPTSpec, seeded_checkpoint
from vllm_ltr_amd.scorer import HipOPTScorer
z = np.load("tests/golden/config3_opt350m_128.npz")
spec = OPTSpec.opt_350m(); ck = seeded_checkpoint(spec, int(z["seed"]))
ids, cu, ref = z["ids"].astype(np.int64), z["cu_seqlens"], z["ref_score"]
two = HipOPTScorer(spec, ck, "cuda:0", "f16")
one = HipOPTScorer(spec, ck, "cuda:0", "f16", lanes=False)
print("probe", two.lane_probe())
for n in (8, 16, 24, 32, 48, 64, 128):
    i, c = ids[:cu[n]], cu[:n + 1]
    b = two.lane_calls()
    s2 = two.score(i, c); used = two.lane_calls() - b
    s1 = one.score(i, c)
    print(f"n={n} T={int(c[-1])} lanes_used={used} |two-ref|={np.abs(s2-ref[:n]).max():.2e} |one-ref|={np.abs(s1-ref[:n]).max():.2e} worst idx {int(np.abs(s2-ref[:n]).argmax())}", flush=True)
    ws = int(two.lib.ltr_workspace_bytes(two._h, 0, n, int(c[-1])))
    print("   ws bytes", ws)
