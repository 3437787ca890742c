import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dataclasses
from util import bench_lengths, synthetic_batch
from oracle.opt_scorer import OracleOPTScorer
from vllm_ltr_amd.opt_spec import OPTSpec, seeded_checkpoint
from vllm_ltr_amd.scorer import HipOPTScorer
for name, spec, n, mu, clip in (("tiny_post big-kernel", OPTSpec.tiny_post_ln(), 300, 24.0, 150),
                                ("350m 3 layers small", dataclasses.replace(OPTSpec.opt_350m(), num_hidden_layers=3), 8, 64.0, 300),
                                ("350m 3 layers mid", dataclasses.replace(OPTSpec.opt_350m(), num_hidden_layers=3), 30, 64.0, 300),
                                ("350m 3 layers big", dataclasses.replace(OPTSpec.opt_350m(), num_hidden_layers=3), 80, 64.0, 300)):
    ckpt = seeded_checkpoint(spec, 5)
    lens = bench_lengths(n, seed=2, mu=mu).clip(1, clip)
    ids, cu = synthetic_batch(spec, lens.tolist(), 3)
    sc = HipOPTScorer(spec, ckpt, "cuda:0", "f16")
    orc = OracleOPTScorer(spec, ckpt)
    for k in range(1, spec.num_hidden_layers + 1):
        got = sc.hidden(ids, cu, n_layers=k)
        want = orc.hidden(ids, cu, n_layers=k).numpy()
        bad = ~np.isfinite(got)
        print(name, "T", int(cu[-1]), "layers", k, "nonfinite rows", int(bad.any(1).sum()), "max|d|", float(np.nanmax(np.abs(got - want))))
    try:
        s = sc.score(ids, cu)
        w = orc.score(ids, cu)
        print(name, "score finite", bool(np.isfinite(s).all()), "max|d|", float(np.nanmax(np.abs(s - w))))
    except Exception as e:
        print(name, "score raised", e)
