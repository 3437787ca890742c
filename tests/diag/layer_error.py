"""Diagnostic: per-layer max|hidden - oracle| at the true 125m shape, F16 and F32 modes."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from oracle.opt_scorer import OracleOPTScorer
from vllm_ltr_amd.opt_spec import OPTSpec, seeded_checkpoint
from vllm_ltr_amd.scorer import HipOPTScorer

spec = OPTSpec.opt_125m()
ckpt = seeded_checkpoint(spec, 21)
rs = np.random.RandomState(22)
lens = [1, 2, 4, 5, 63, 64, 65, 200, 1024, 90, 33, 300]
ids = np.concatenate([np.r_[2, rs.randint(4, spec.vocab_size, L - 1)] for L in lens]).astype(np.int64)
cu = np.r_[0, np.cumsum(lens)].astype(np.int32)
torch.set_num_threads(16)
orc64 = OracleOPTScorer(spec, ckpt, dtype=torch.float64)
layers = [0, 1, 2, 4, 8, 12]
want = {nl: orc64.hidden(ids, cu, n_layers=nl).numpy() for nl in layers}
orc32 = OracleOPTScorer(spec, ckpt)
print("layers     " + "  ".join(f"{nl:9d}" for nl in layers))
print("oracle f32 " + "  ".join(f"{np.abs(orc32.hidden(ids, cu, n_layers=nl).numpy() - want[nl]).max():9.2e}" for nl in layers))
for mode, env in (("f16", "0"), ("f16+valu-attn", "1"), ("f32", "0")):
    os.environ["LTR_DEBUG_ATTN_VALU"] = env
    sc = HipOPTScorer(spec, ckpt, "cuda:0", mode.split("+")[0])
    errs = []
    for nl in layers:
        got = sc.hidden(ids, cu, n_layers=nl)
        e = np.abs(got - want[nl])
        errs.append(e.max())
    print(f"{mode:10s} " + "  ".join(f"{e:9.2e}" for e in errs))
    if False:
        got = sc.hidden(ids, cu, n_layers=1); e = np.abs(got - want[1]).max(1)
        worst = np.argsort(-e)[:8]
        print("  worst rows after 1 layer:", [(int(t), int(np.searchsorted(cu, t, 'right') - 1), float(e[t])) for t in worst])
