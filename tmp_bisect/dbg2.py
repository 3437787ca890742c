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
def sub(a, b):
    return ids[cu[a]:cu[b]], (cu[a:b + 1] - cu[a]).astype(np.int32)
for a, b in ((0, 4), (4, 8), (0, 3), (0, 5), (0, 6), (0, 7), (0, 8), (8, 16), (0, 2), (0, 1)):
    i, c = sub(a, b)
    s1 = one.score(i, c); bb = two.lane_calls(); s2 = two.score(i, c); used = two.lane_calls() - bb
    print(f"[{a},{b}) T={int(c[-1])} one-lane err {np.abs(s1-ref[a:b]).max():.2e}  lanes-handle err {np.abs(s2-ref[a:b]).max():.2e} (two lanes used {used})", flush=True)
for rep in range(3):
    i, c = sub(0, 8)
    print("repeat", np.abs(two.score(i, c) - ref[:8]).round(6).tolist())
