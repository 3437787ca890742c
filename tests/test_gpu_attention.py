"""-m gpu: the varlen causal attention kernels ALONE against the oracle's per-sequence
softmax(q k^T / 8) v (oracle/opt_scorer.py::attention = torch_sdpa.py:138-178 semantics), through the C ABI
(``ltr_attention``).  The scorer tests see attention only through whole layers; this pins the kernel itself at the lengths
where its tiling changes: 1, 31 / 32 / 33 (one key tile), 128 / 129 (one / two query blocks), 1024, 2048 (the position
table's limit), mixed in one ragged batch with non-trivial offsets."""
import numpy as np
import pytest
import torch

from oracle.opt_scorer import OracleOPTScorer
from vllm_ltr_amd.opt_spec import OPTSpec, seeded_checkpoint

pytestmark = pytest.mark.gpu

LENS = [1, 31, 32, 33, 128, 129, 1024, 2048, 5, 64, 65, 1, 300, 127]


def _qkv(T, H, seed, scale_q):
    r = np.random.RandomState(seed)
    x = r.standard_normal((T, 3 * H)).astype(np.float32)
    x[:, :H] *= scale_q                      # sharper or flatter softmax
    x[:, H:2 * H] *= 1.5
    return x


# split_kv: the 32-query split-K/V variant small passes run (each wave walks every fourth K/V tile as its own stream, the
# four online-softmax states merged through LDS) against the 128-query kernel of full-size passes
# split_kv = "wide": the 256-query / eight-wave workgroups (attn_f16s_kernel<8, ...>, LTR_ATTN_NW=8: one K/V tile in LDS serves
# 256 queries), with 255 / 256 / 257-token requests at its block edge
@pytest.mark.parametrize("model,mode,split_kv", [("125m", "f16", True), ("125m", "f16", False), ("350m", "f16", True),
                                                  ("350m", "f16", False), ("125m", "f32", False), ("125m", "f16", "wide"),
                                                  ("350m", "f16", "wide")])
@pytest.mark.parametrize("scale_q", [0.5, 4.0])
def test_attention_kernel_alone_vs_oracle(model, mode, split_kv, scale_q, monkeypatch):
    from vllm_ltr_amd.scorer import HipOPTScorer
    # the library reads the threshold per call (default 600 tokens): this test runs the variant up to 2,048
    monkeypatch.setenv("LTR_ATTN_SPLITKV_TOKENS", "2048")
    wide = split_kv == "wide"
    monkeypatch.setenv("LTR_ATTN_NW", "8" if wide else "4")
    split_kv = False if wide else split_kv
    spec = (OPTSpec.opt_125m() if model == "125m" else OPTSpec.opt_350m())
    spec1 = OPTSpec(**{**spec.__dict__, "num_hidden_layers": 1})          # the handle only supplies H / heads / mode
    ckpt = seeded_checkpoint(spec1, 0)
    if mode == "f32":
        ckpt = {k: v.astype(np.float32) for k, v in ckpt.items()}
    sc = HipOPTScorer(spec1, ckpt, "cuda:0", mode)
    H = spec.hidden_size
    lens = LENS if mode == "f16" else [1, 31, 32, 33, 128, 129, 300]       # the f32 VALU kernel is the slow cross-check path
    if split_kv:
        lens = [1, 31, 32, 33, 128, 129, 161, 1024, 300]                   # (LTR_ATTN_SPLITKV_TOKENS below: the split-K/V kernel)
    if wide:
        lens = LENS + [255, 256, 257, 513]
    T = int(np.sum(lens))
    cu = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    x = _qkv(T, H, 3, scale_q)
    want = OracleOPTScorer(spec1, ckpt, dtype=torch.float64).attention(torch.from_numpy(x).double(), lens).numpy()
    dev = torch.device("cuda:0")
    cu_d = torch.from_numpy(cu).to(dev)
    xt = torch.from_numpy(x).to(dev)
    if mode == "f16":
        hi = xt.to(torch.float16)
        lo = (xt - hi.float()).to(torch.float16)
        qkv = torch.stack([hi, lo]).contiguous()
        out = torch.full((2, T, H), float("nan"), dtype=torch.float16, device=dev)
        sc.attention_device(qkv, cu_d, len(lens), T, out, split_kv=split_kv)
        got = (out[0].double() + out[1].double()).cpu().numpy()
        # the kernel sees hi + lo (22 bits of x): compare with the oracle on exactly those inputs
        x_seen = (hi.double() + lo.double()).cpu().numpy()
        want = OracleOPTScorer(spec1, ckpt, dtype=torch.float64).attention(torch.from_numpy(x_seen), lens).numpy()
    else:
        out = torch.full((T, H), float("nan"), dtype=torch.float32, device=dev)
        sc.attention_device(xt, cu_d, len(lens), T, out)
        got = out.double().cpu().numpy()
    assert np.isfinite(got).all()
    err = np.abs(got - want)
    scale = np.abs(want).max()
    worst = int(err.max(axis=1).argmax())
    req = int(np.searchsorted(cu, worst, side="right") - 1)
    print(f"{model}/{mode}{' split-K/V' if split_kv else (' 256-query' if wide else '')} q-scale {scale_q}: T = {T}, max|out - oracle| = {err.max():.3e} (|out| <= {scale:.2f}) at row {worst} "
          f"= position {worst - cu[req]} of a {lens[req]}-token request")
    # outputs are convex combinations of v ~ N(0, 1): absolute tolerance.  f16 path: products of split operands
    # (dropped lo.lo terms 2^-22) + an fp16 hi|lo output; f32 path: plain f32 arithmetic
    assert err.max() <= 3e-6 * max(1.0, scale)


@pytest.mark.parametrize("heads,scale_q,dout_scale", [(12, 1.0, 1.0), (12, 4.0, 3e-7), (16, 0.5, 2e3)])
def test_attention_backward_alone_vs_f64_autograd(heads, scale_q, dout_scale):
    """The training step's attention backward ALONE (``ltr_train_attention``: split-fp16 MFMA forward + backward)
    against torch autograd in f64 on the same ragged batch - lengths around every tiling edge (1; 31 / 32 / 33: one key
    tile; 127 / 128 / 129: one or two blocks; 300; 1024), output gradients from 3e-7 (far below fp16's range: the kernel
    scales them) to 2e3.  Bar: 2e-5 of each gradient's largest entry, per (request, q / k / v)."""
    from vllm_ltr_amd.trainer import attention_forward_backward
    lens = [1, 31, 32, 33, 128, 129, 1024, 5, 64, 65, 1, 300, 127]
    H = 64 * heads
    T = int(np.sum(lens))
    cu = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    x = _qkv(T, H, 11, scale_q)
    r = np.random.RandomState(12)
    do = (r.standard_normal((T, H)) * dout_scale * np.exp(r.standard_normal((T, 1)))).astype(np.float32)   # rows of very different size
    dev = torch.device("cuda:0")
    out, dqkv = attention_forward_backward(torch.from_numpy(x).to(dev), torch.from_numpy(do).to(dev), torch.from_numpy(cu).to(dev), heads)
    out, dqkv = out.double().cpu(), dqkv.double().cpu()
    assert torch.isfinite(out).all() and torch.isfinite(dqkv).all()
    worst = 0.0
    for i, L in enumerate(lens):
        a, b = int(cu[i]), int(cu[i + 1])
        xx = torch.from_numpy(x[a:b]).double().requires_grad_(True)
        q, k, v = (xx[:, j * H:(j + 1) * H].reshape(L, heads, 64).transpose(0, 1) for j in range(3))
        s = (q @ k.transpose(1, 2)) * 0.125
        s = s.masked_fill(torch.triu(torch.ones(L, L, dtype=torch.bool), 1), float("-inf"))
        o = (torch.softmax(s, -1) @ v).transpose(0, 1).reshape(L, H)
        o.backward(torch.from_numpy(do[a:b]).double())
        assert (out[a:b] - o.detach()).abs().max() <= 2e-6 * max(1.0, float(o.detach().abs().max())), (i, L)
        for j, name in enumerate("qkv"):
            want = xx.grad[:, j * H:(j + 1) * H]
            got = dqkv[a:b, j * H:(j + 1) * H]
            # (a one-token prompt has dq = dk = 0 exactly; what the kernel leaves there is the rounding of dP - D:
            #  an absolute floor of 1e-6 |dO| |x|^2 next to the relative bar)
            floor = 1e-6 * float(np.abs(do[a:b]).max()) * float(np.abs(x[a:b]).max()) ** 2
            err = float((got - want).abs().max())
            rel = err / max(float(want.abs().max()), floor / 2e-5)
            worst = max(worst, rel)
            assert rel <= 2e-5, f"request {i} (L = {L}) d{name}: {rel:.2e} of its largest entry"
    print(f"attention backward, heads={heads} scale_q={scale_q} dout~{dout_scale:g}: worst max-relative error {worst:.2e}")
