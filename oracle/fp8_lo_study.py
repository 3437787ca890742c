#!/usr/bin/env python3
"""CPU emulation (test infrastructure): what an fp8 `lo` pass would do to the scores.

The production GEMM computes  acc = hi.W + lo.W  on the fp16 MFMA (two passes).  VERDICT r1 4(d) asks whether the lo pass
could run on the 2x faster fp8 MFMA (v_mfma_scale_f32_16x16x128_f8f6f4): that needs lo AND W in fp8 (e4m3, optionally MX
block scales of 32 along K).  This script emulates exactly that arithmetic in torch on the true-shape golden inputs
(tests/golden/score_opt125m.npz / score_opt350m.npz) and prints max|score - reference| per variant:

    python oracle/fp8_lo_study.py            # ~1 min on 8 cores

    variants: which GEMMs use the fp8 lo pass (all / fc1+fc2 / fc2 only / none), plain e4m3 vs MX (per-32 power-of-two scales)
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle.opt_scorer import OracleOPTScorer, LN_EPS  # noqa: E402
from util import spec_from_npz  # noqa: E402
from vllm_ltr_amd.opt_spec import seeded_checkpoint  # noqa: E402

E4M3_MAX = 448.0


def q8(x: torch.Tensor, mx: bool) -> torch.Tensor:
    """Round to fp8 e4m3 (optionally with a power-of-two scale per block of 32 along the last axis, as MX does)."""
    if not mx:
        return x.to(torch.float8_e4m3fn).to(torch.float32)
    K = x.shape[-1]
    xb = x.reshape(*x.shape[:-1], K // 32, 32)
    amax = xb.abs().amax(-1, keepdim=True).clamp_min(1e-30)
    scale = torch.exp2(torch.floor(torch.log2(E4M3_MAX / amax)))
    return ((xb * scale).to(torch.float8_e4m3fn).to(torch.float32) / scale).reshape(x.shape)


def split_linear(x, w, b, fp8_lo: bool, mx: bool):
    """hi.W + lo.W with hi = fp16(x), lo = fp16(x - hi); fp8_lo: the lo pass sees q8(lo) and q8(W)."""
    hi = x.to(torch.float16).to(torch.float32)
    lo = (x - hi).to(torch.float16).to(torch.float32)
    if fp8_lo:
        out = F.linear(hi, w) + F.linear(q8(lo, mx), q8(w, mx))
    else:
        out = F.linear(hi, w) + F.linear(lo, w)
    return out + b if b is not None else out


def forward(orc: OracleOPTScorer, ids, cu, use, mx):
    s = orc.spec
    H = s.hidden_size
    lens = np.diff(cu).tolist()
    ids_t = torch.as_tensor(ids, dtype=torch.long)
    pos = torch.cat([torch.arange(L, dtype=torch.long) for L in lens])
    h = orc.embed(ids_t, pos)
    for lw in orc.layers:
        res = h
        x = F.layer_norm(h, (H,), lw["ln1w"], lw["ln1b"], LN_EPS) if s.do_layer_norm_before else h
        x = split_linear(x, lw["wqkv"], lw["bqkv"], "qkv" in use, mx)
        x = orc.attention(x, lens)
        x = split_linear(x, lw["wo"], lw["bo"], "out" in use, mx)
        h = res + x
        if not s.do_layer_norm_before:
            h = F.layer_norm(h, (H,), lw["ln1w"], lw["ln1b"], LN_EPS)
        res = h
        x = F.layer_norm(h, (H,), lw["ln2w"], lw["ln2b"], LN_EPS) if s.do_layer_norm_before else h
        x = F.relu(split_linear(x, lw["w1"], lw["b1"], "fc1" in use, mx))
        x = split_linear(x, lw["w2"], lw["b2"], "fc2" in use, mx)
        h = res + x
        if not s.do_layer_norm_before:
            h = F.layer_norm(h, (H,), lw["ln2w"], lw["ln2b"], LN_EPS)
    return orc.pool_head(h, torch.as_tensor(cu[1:].astype(np.int64) - 1))[:, 0].numpy()


def tail_study(n_req: int):
    """Round 3: the 3.4e-5 / 6.0e-5 above are maxima over the 12 golden requests; the bar is 1e-4 on EVERY request of an
    8k queue.  Distribution of the error over the first n_req requests of the bench queue (OPT-125m, seed 0)."""
    from vllm_ltr_amd.opt_spec import OPTSpec
    sys.path.insert(0, ROOT)
    import bench
    spec = OPTSpec.opt_125m()
    ckpt = seeded_checkpoint(spec, 0)
    ids, cu, lens = bench.synthetic_queue(spec, 8192, 0)
    ids, cu = ids[:cu[n_req]], cu[:n_req + 1].astype(np.int64)
    orc = OracleOPTScorer(spec, ckpt)
    out = {}
    with torch.no_grad():
        for label, use in (("fp16 lo (production)", ()), ("fp8 lo in fc1 + fc2", ("fc1", "fc2")),
                           ("fp8 lo in all four", ("qkv", "out", "fc1", "fc2"))):
            got = np.concatenate([forward(orc, ids[cu[i]:cu[j]], cu[i:j + 1] - cu[i], use, True)
                                  for i, j in zip(range(0, n_req, 16), list(range(16, n_req, 16)) + [n_req])])
            out[label] = got
    ref = out["fp16 lo (production)"]
    for label, got in out.items():
        e = np.abs(got - ref)
        print(f"tail over {n_req} requests, {label:24s} vs the production arithmetic: max {e.max():.2e}  p99 {np.percentile(e, 99):.2e}  "
              f"p50 {np.percentile(e, 50):.2e}  rms {np.sqrt((e ** 2).mean()):.2e}")


def main():
    torch.set_num_threads(os.cpu_count())
    if len(sys.argv) > 2 and sys.argv[1] == "--tail":
        return tail_study(int(sys.argv[2]))
    for name in ("opt125m", "opt350m"):
        z = np.load(os.path.join(ROOT, "tests", "golden", f"score_{name}.npz"))
        spec = spec_from_npz(z)
        orc = OracleOPTScorer(spec, seeded_checkpoint(spec, int(z["seed"])))
        ref = z["ref_score"]
        with torch.no_grad():
            for label, use in (("fp16 lo everywhere (production arithmetic)", ()), ("fp8 lo in fc2", ("fc2",)),
                               ("fp8 lo in fc1 + fc2", ("fc1", "fc2")), ("fp8 lo in all four GEMMs", ("qkv", "out", "fc1", "fc2"))):
                for mx in ((False,) if not use else (False, True)):
                    got = forward(orc, z["ids"], z["cu_seqlens"].astype(np.int64), use, mx)
                    print(f"{name:8s} {label:42s} {'MX block scales' if mx else 'plain e4m3    '}  max|score - reference| = {np.abs(got - ref).max():.2e}")


if __name__ == "__main__":
    main()
