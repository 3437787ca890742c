"""diag: cost of the attention kernel per token as a function of the prompt length (uniform-length batches of ~196k tokens,
OPT-125m head layout).  python tests/diag/attn_by_length.py"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import dataclasses
from vllm_ltr_amd.opt_spec import OPTSpec, seeded_checkpoint
from vllm_ltr_amd.scorer import HipOPTScorer
spec = dataclasses.replace(OPTSpec.opt_125m(), num_hidden_layers=1)
sc = HipOPTScorer(spec, seeded_checkpoint(spec, 0), "cuda:0", "f16")
dev = torch.device("cuda:0")
H = spec.hidden_size
def run(lens, tag):
    lens = np.asarray(lens); T = int(lens.sum()); n = len(lens)
    cu = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    qkv = (torch.randn(2, T, 3 * H, device=dev) * torch.tensor([1.0, 1e-4], device=dev).view(2, 1, 1)).to(torch.float16)
    out = torch.empty(2, T, H, dtype=torch.float16, device=dev)
    cu_d = torch.from_numpy(cu).to(dev)
    for _ in range(3): sc.attention_device(qkv, cu_d, n, T, out)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
    for a, b in ev:
        a.record(); sc.attention_device(qkv, cu_d, n, T, out); b.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in ev)[5]
    print(f"{tag:28s} n={n:6d} T={T:7d}  {ms*1e3:8.1f} us  {ms*1e6/T:7.2f} ns/token  {ms*1e6/(n*12):7.2f} ns per (request, head)")
for L in (16, 32, 33, 64, 65, 96, 128, 129, 256, 512, 1024):
    run([L] * (196608 // L), f"uniform L={L}")
rs = np.random.RandomState(0)
lens = np.clip(np.rint(np.exp(rs.normal(np.log(64), 0.8, 8192))), 4, 1024).astype(np.int64)
cum = np.cumsum(lens); k = int(np.searchsorted(cum, 196608))
run(lens[:k], "bench profile, one pass")
short = lens[:k][lens[:k] <= 64]; long_ = lens[:k][lens[:k] > 64]
run(short, "  its requests <= 64 tokens")
run(long_, "  its requests  > 64 tokens")
