"""ORACLE (test infrastructure, not product code) - CPU fp32 restatement of the
reference's predictor forward.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module; the product package (``vllm_ltr_amd``) never does.

What it restates (all citations relative to /root/reference):

* flat varlen inputs - ``input_tokens`` [T], positions 0..L_i-1, ``seq_start_loc``
  = cumsum(L), ``selected_token_indices`` = cu[i+1]-1
  (vllm/worker/model_runner.py:383-395, 592-593, 711-716)
* ``OPTDecoder.forward`` (vllm/model_executor/models/opt.py:233-263):
  token embedding (+ ``project_in``) + learned positions at ``pos + 2`` (:43-53)
* ``OPTDecoderLayer.forward`` (:145-176) pre-LN / post-LN block,
  ``OPTAttention`` (:92-102) with per-sequence causal softmax(QK^T * d^-0.5)V,
  which is what the prefill path with ``kv_cache=None`` computes
  (vllm/attention/backends/torch_sdpa.py:138-178)
* final LayerNorm / ``project_out`` (:259-262)
* ``compute_logits`` (:389-397) = ``index_select`` of the last token of every
  prompt (layers/logits_processor.py:74-79) times ``score.weight^T`` (no bias,
  :374); class mode = ``argmax`` over ``num_labels`` returned as float (:394-395)
* rank-mode hand-back ``logits[:, 0]`` (:408)

Pinning: ``oracle/make_golden.py`` runs the *reference itself* (imported from
/root/reference in the build container) and HF ``OPTForSequenceClassification``
on the same seeded checkpoints and stores their scores in ``tests/golden/``;
``tests/test_oracle_golden.py`` checks this restatement against those vectors.
The reference's own test-suite holds no vectors for this path.

Arithmetic: fp32 torch CPU ops; weights are the checkpoint's fp16 values widened
to fp32 (what loading an fp16 HF checkpoint with ``torch_dtype=float32`` does).
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F

LN_EPS = 1e-5  # nn.LayerNorm default, opt.py:131-133,165-167,222-224


class OracleOPTScorer:
    def __init__(self, spec, ckpt: Dict[str, np.ndarray], dtype=torch.float32):
        self.spec = spec
        self.dtype = dtype
        self.w = {k: torch.from_numpy(np.asarray(v).astype(np.float32)).to(dtype)
                  for k, v in ckpt.items()}
        # q,k,v stacked in that order, as opt.py:411-417 / linear.py:391-466 do
        self.layers = []
        for i in range(spec.num_hidden_layers):
            p = f"model.decoder.layers.{i}."
            g = lambda n: self.w[p + n]
            self.layers.append(dict(
                wqkv=torch.cat([g("self_attn.q_proj.weight"), g("self_attn.k_proj.weight"),
                                g("self_attn.v_proj.weight")], 0),
                bqkv=torch.cat([g("self_attn.q_proj.bias"), g("self_attn.k_proj.bias"),
                                g("self_attn.v_proj.bias")], 0),
                wo=g("self_attn.out_proj.weight"), bo=g("self_attn.out_proj.bias"),
                ln1w=g("self_attn_layer_norm.weight"), ln1b=g("self_attn_layer_norm.bias"),
                w1=g("fc1.weight"), b1=g("fc1.bias"), w2=g("fc2.weight"), b2=g("fc2.bias"),
                ln2w=g("final_layer_norm.weight"), ln2b=g("final_layer_norm.bias")))

    # -- pieces (individually addressable so kernels can be checked one by one) --
    def embed(self, ids: torch.Tensor, pos: torch.Tensor) -> torch.Tensor:
        """opt.py:241-245: h0 = P_in(E_tok[ids]) + E_pos[pos + 2]."""
        s = self.spec
        x = F.embedding(ids, self.w["model.decoder.embed_tokens.weight"])
        if s.has_proj:
            x = F.linear(x, self.w["model.decoder.project_in.weight"])
        return x + F.embedding(pos + s.POS_OFFSET, self.w["model.decoder.embed_positions.weight"])

    def attention(self, qkv: torch.Tensor, lens: Sequence[int]) -> torch.Tensor:
        """Per-sequence causal attention, heads of size head_dim, scale d^-0.5
        (opt.py:73,98-100; torch_sdpa.py:138-178)."""
        s = self.spec
        H, nh, d = s.hidden_size, s.num_attention_heads, s.head_dim
        q, k, v = qkv.split(H, dim=-1)
        out = torch.empty_like(q)
        start = 0
        for L in lens:
            sl = slice(start, start + L)
            qq = q[sl].view(L, nh, d).transpose(0, 1)
            kk = k[sl].view(L, nh, d).transpose(0, 1)
            vv = v[sl].view(L, nh, d).transpose(0, 1)
            att = torch.matmul(qq, kk.transpose(1, 2)) * (d ** -0.5)
            mask = torch.ones(L, L, dtype=torch.bool).tril_()
            att = att.masked_fill(~mask, float("-inf"))
            att = torch.softmax(att, dim=-1)
            out[sl] = torch.matmul(att, vv).transpose(0, 1).reshape(L, H)
            start += L
        return out

    def layer(self, h: torch.Tensor, lw: dict, lens: Sequence[int]) -> torch.Tensor:
        """opt.py:145-176."""
        s = self.spec
        H = s.hidden_size
        res = h
        x = F.layer_norm(h, (H,), lw["ln1w"], lw["ln1b"], LN_EPS) if s.do_layer_norm_before else h
        x = F.linear(x, lw["wqkv"], lw["bqkv"])
        x = self.attention(x, lens)
        x = F.linear(x, lw["wo"], lw["bo"])
        h = res + x
        if not s.do_layer_norm_before:
            h = F.layer_norm(h, (H,), lw["ln1w"], lw["ln1b"], LN_EPS)
        res = h
        x = F.layer_norm(h, (H,), lw["ln2w"], lw["ln2b"], LN_EPS) if s.do_layer_norm_before else h
        x = F.relu(F.linear(x, lw["w1"], lw["b1"]))
        x = F.linear(x, lw["w2"], lw["b2"])
        h = res + x
        if not s.do_layer_norm_before:
            h = F.layer_norm(h, (H,), lw["ln2w"], lw["ln2b"], LN_EPS)
        return h

    def pool_head(self, h: torch.Tensor, last_idx: torch.Tensor) -> torch.Tensor:
        """Final LN / project_out on the selected rows only (both are per-token
        maps, so this equals opt.py:259-262 over all T followed by
        logits_processor.py:74-79) then ``score.weight`` (opt.py:374,389-397).
        Returns logits [N, num_labels]."""
        s = self.spec
        x = h.index_select(0, last_idx)
        if s.has_final_ln:
            x = F.layer_norm(x, (s.hidden_size,), self.w["model.decoder.final_layer_norm.weight"],
                             self.w["model.decoder.final_layer_norm.bias"], LN_EPS)
        if s.has_proj:
            x = F.linear(x, self.w["model.decoder.project_out.weight"])
        return F.linear(x, self.w["score.weight"])

    # -- whole path ------------------------------------------------------------
    @torch.no_grad()
    def hidden(self, ids: np.ndarray, cu_seqlens: np.ndarray,
               n_layers: Optional[int] = None) -> torch.Tensor:
        lens = np.diff(np.asarray(cu_seqlens)).astype(np.int64).tolist()
        ids_t = torch.as_tensor(np.asarray(ids), dtype=torch.long)
        pos = torch.cat([torch.arange(L, dtype=torch.long) for L in lens]) if lens else \
            torch.zeros(0, dtype=torch.long)
        h = self.embed(ids_t, pos)
        for lw in self.layers[:n_layers]:
            h = self.layer(h, lw, lens)
        return h

    @torch.no_grad()
    def logits(self, ids: np.ndarray, cu_seqlens: np.ndarray) -> torch.Tensor:
        cu = np.asarray(cu_seqlens).astype(np.int64)
        if len(cu) <= 1:
            return torch.zeros(0, self.spec.num_labels, dtype=self.dtype)
        h = self.hidden(ids, cu)
        return self.pool_head(h, torch.as_tensor(cu[1:] - 1, dtype=torch.long))

    @torch.no_grad()
    def score(self, ids: np.ndarray, cu_seqlens: np.ndarray) -> np.ndarray:
        """The value the scheduler sees as ``aux_model_score``: rank mode
        ``logits[:, 0]`` (opt.py:408); class mode ``float(argmax)`` (:394-395)."""
        lg = self.logits(ids, cu_seqlens)
        if self.spec.num_labels > 1:
            # LogitsProcessor._get_logits cuts the logits at vocab_size columns ("remove paddings in vocab",
            # layers/logits_processor.py:68-70; LogitsProcessor(config.vocab_size), opt.py:375) before opt.py:395 takes
            # the argmax over [:num_labels]: with fewer vocabulary entries than labels only the first vocab_size compete
            lg = lg[:, :min(self.spec.num_labels, self.spec.vocab_size)].argmax(dim=-1, keepdim=True).float()
        return lg[:, 0].float().numpy().astype(np.float32)

    def score_packed(self, ids: np.ndarray, cu_seqlens: np.ndarray,
                     max_tokens: int = 2048) -> np.ndarray:
        """Same scores, but packed into FCFS batches of <= max_tokens like the AUX
        engine does (scheduler.py:763-855, config.py:578-586) - bounds the CPU
        memory of the LxL attention and is how the CPU baseline is timed."""
        cu = np.asarray(cu_seqlens).astype(np.int64)
        N = len(cu) - 1
        out = np.zeros(N, np.float32)
        i = 0
        while i < N:
            j = i + 1
            while j < N and cu[j + 1] - cu[i] <= max_tokens:
                j += 1
            out[i:j] = self.score(np.asarray(ids)[cu[i]:cu[j]], cu[i:j + 1] - cu[i])
            i = j
        return out
