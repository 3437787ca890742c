"""ORACLE (test infrastructure, not product code) - CPU restatement of ONE optimisation step of the reference's
predictor fine-tuning loop, ``train/trainer.py:122-165``:

    optimizer = torch.optim.Adam(predictor.model.parameters(), lr=args.lr, weight_decay=args.wc)      # :122
    outputs = predictor(input_ids, attention_mask)                                                   # :146
    loss = loss_func(outputs.view(1, -1), labels)      # listMLE / mse: the batch is ONE slate        # :157
    loss = loss_func(logits, labels.view(...))         # crossentropy over num_labels classes        # :154-155
    loss.backward(); optimizer.step(); optimizer.zero_grad()                                          # :161-165

The forward is :class:`oracle.opt_scorer.OracleOPTScorer`'s arithmetic (pinned to the reference's own
``OPTForSequenceClassification`` and to HF's); here its tensors are autograd leaves named like the HF checkpoint,
the backward is torch autograd and the update is ``torch.optim.Adam`` itself (L2 weight decay added to the
gradient, bias-corrected moments), so nothing of the optimisation is re-derived by hand.  The ListMLE loss restates
``train/allrank/models/losses/listMLE.py:23-54`` in torch with the shuffle permutation as an input (pinned to the
reference's function by tests/golden/listmle.npz).  Dropout (HF OPT ``dropout = 0.1`` in train mode) draws from
torch's global RNG in the reference and is therefore not reproducible across implementations; the oracle - like the
parity fixtures - runs with dropout 0.

Pinning: ``oracle/make_train_golden.py`` runs HF ``OPTForSequenceClassification`` (fp32, dropout 0) + the
reference's listMLE + ``torch.optim.Adam`` for a few steps and stores losses and updated tensors in
``tests/golden/train_steps_*.npz``; ``tests/test_train_step.py`` replays them here.
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F

from oracle.neuralndcg import neuralndcg_torch
from oracle.opt_scorer import OracleOPTScorer

DEFAULT_EPS = 1e-10      # allrank/models/losses/__init__.py:17
PADDED_Y_VALUE = -1      # allrank/data/dataset_loading.py:31


def listmle_torch(y_pred: torch.Tensor, y_true: torch.Tensor, shuffle: torch.Tensor, eps: float = DEFAULT_EPS,
                  pad: float = PADDED_Y_VALUE) -> torch.Tensor:
    """listMLE.py:23-54 with ``random_indices`` = ``shuffle``; ties keep the shuffled order (stable sort)."""
    ps, ts = y_pred[:, shuffle], y_true[:, shuffle]                                  # :33-35
    ts_sorted, idx = torch.sort(ts, descending=True, dim=-1, stable=True)            # :37
    mask = ts_sorted == pad                                                          # :39
    p = torch.gather(ps, 1, idx)                                                     # :41
    p = p.masked_fill(mask, float("-inf"))                                           # :42
    m, _ = p.max(dim=1, keepdim=True)                                                # :44
    q = p - m                                                                        # :46
    c = torch.cumsum(q.exp().flip(dims=[1]), dim=1).flip(dims=[1])                   # :48
    obs = torch.log(c + eps) - q                                                     # :50
    obs = obs.masked_fill(mask, 0.0)                                                 # :52
    return torch.mean(torch.sum(obs, dim=1))                                         # :54


class OracleTrainer:
    def __init__(self, spec, ckpt: Dict[str, np.ndarray], lr: float = 2e-5, weight_decay: float = 0.01,
                 betas=(0.9, 0.999), eps: float = 1e-8, loss: str = "listMLE", dtype=torch.float32):
        self.spec = spec
        self.loss = loss
        self.dtype = dtype
        self.params = {k: torch.tensor(np.asarray(v).astype(np.float32), dtype=dtype, requires_grad=True)
                       for k, v in ckpt.items()}
        self.opt = torch.optim.Adam(list(self.params.values()), lr=lr, weight_decay=weight_decay, betas=betas, eps=eps)

    def _model(self) -> OracleOPTScorer:
        orc = OracleOPTScorer.__new__(OracleOPTScorer)
        orc.spec, orc.dtype, orc.w = self.spec, self.dtype, self.params
        orc.layers = []
        for i in range(self.spec.num_hidden_layers):
            p = f"model.decoder.layers.{i}."
            g = lambda n: self.params[p + n]
            orc.layers.append(dict(
                wqkv=torch.cat([g("self_attn.q_proj.weight"), g("self_attn.k_proj.weight"), g("self_attn.v_proj.weight")], 0),
                bqkv=torch.cat([g("self_attn.q_proj.bias"), g("self_attn.k_proj.bias"), g("self_attn.v_proj.bias")], 0),
                wo=g("self_attn.out_proj.weight"), bo=g("self_attn.out_proj.bias"),
                ln1w=g("self_attn_layer_norm.weight"), ln1b=g("self_attn_layer_norm.bias"),
                w1=g("fc1.weight"), b1=g("fc1.bias"), w2=g("fc2.weight"), b2=g("fc2.bias"),
                ln2w=g("final_layer_norm.weight"), ln2b=g("final_layer_norm.bias")))
        return orc

    def logits(self, ids: np.ndarray, cu: np.ndarray) -> torch.Tensor:
        orc = self._model()
        cu = np.asarray(cu).astype(np.int64)
        lens = np.diff(cu).tolist()
        ids_t = torch.as_tensor(np.asarray(ids), dtype=torch.long)
        pos = torch.cat([torch.arange(L, dtype=torch.long) for L in lens])
        h = orc.embed(ids_t, pos)
        for lw in orc.layers:
            h = orc.layer(h, lw, lens)
        return orc.pool_head(h, torch.as_tensor(cu[1:] - 1, dtype=torch.long))

    def loss_of(self, logits: torch.Tensor, labels: np.ndarray, shuffle: Optional[np.ndarray]) -> torch.Tensor:
        y = torch.as_tensor(np.asarray(labels))
        if self.loss == "listMLE":                    # trainer.py:157: the batch is one slate
            return listmle_torch(logits.view(1, -1), y.to(self.dtype).view(1, -1), torch.as_tensor(np.asarray(shuffle), dtype=torch.long))
        if self.loss == "neuralNDCG":                 # trainer.py:127-128,157: every keyword of neuralNDCG at its default
            return neuralndcg_torch(logits.view(1, -1), y.to(self.dtype).view(1, -1))
        if self.loss == "mse":                        # trainer.py:130,157
            return F.mse_loss(logits.view(1, -1), y.to(self.dtype).view(1, -1))
        if self.loss == "crossentropy":               # trainer.py:132,152-155
            return F.cross_entropy(logits.view(-1, self.spec.num_labels), y.long().view(-1))
        raise ValueError(self.loss)

    def step(self, ids, cu, labels, shuffle=None, apply: bool = True):
        """One optimisation step; returns (loss, logits before the update, {name: grad})."""
        self.opt.zero_grad()
        logits = self.logits(ids, cu)
        loss = self.loss_of(logits, labels, shuffle)
        loss.backward()
        grads = {k: (v.grad.detach().clone() if v.grad is not None else torch.zeros_like(v)) for k, v in self.params.items()}
        if apply:
            self.opt.step()
        return float(loss.item()), logits.detach().numpy().astype(np.float32), grads

    def state(self) -> Dict[str, np.ndarray]:
        return {k: v.detach().numpy().astype(np.float32) for k, v in self.params.items()}
