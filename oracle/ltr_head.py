"""ORACLE (test infrastructure, not product code) - CPU restatement of the reference's
hidden-state learning-to-rank head.

Reference: vllm/model_executor/predictor.py - ``FCModel`` (:10-43: optional input LayerNorm, then
``activation(Linear(x))`` per layer, dropout is identity at inference), ``OutputLayer`` (:91-125:
``activation(w_1(x))``, ``score`` sums over ``d_output`` when it is > 1), ``LTRModel.score``
(:78-89) and the factory ``predictor_model`` (:128-145).  The reference hooks it at
``OPTDecoder.forward`` / ``LlamaModel.forward`` on the hidden states of layer ``pred_layer_idx`` at
the selected tokens (opt.py:250-255) and loads it in model_loader/loader.py:234-241; no schedule
type consumes it.

Pinned by ``tests/golden/ltr_head_*.npz`` (``oracle/make_golden.py`` runs the reference's own
``predictor_model`` on seeded weights).
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F

_ACT = {None: lambda x: x, "Identity": lambda x: x, "ReLU": F.relu, "Sigmoid": torch.sigmoid,
        "Tanh": torch.tanh, "GELU": F.gelu, "SiLU": F.silu}


def seeded_head_weights(n_features: int, fc_sizes, input_norm: bool, d_output: int, seed: int) -> Dict[str, np.ndarray]:
    """State dict (reference parameter names) from NumPy RandomState, fp16-rounded values."""
    rs = np.random.RandomState(seed)
    sd = {}
    sizes = [n_features] + list(fc_sizes or [])
    if fc_sizes is not None:
        if input_norm:
            sd["input_layer.input_norm.weight"] = 1.0 + 0.1 * rs.standard_normal(n_features)
            sd["input_layer.input_norm.bias"] = 0.05 * rs.standard_normal(n_features)
        for i, (a, b) in enumerate(zip(sizes[:-1], sizes[1:])):
            sd[f"input_layer.layers.{i}.weight"] = rs.standard_normal((b, a)) * (2.0 / (a + b)) ** 0.5
            sd[f"input_layer.layers.{i}.bias"] = 0.05 * rs.standard_normal(b)
    d_model = sizes[-1]
    sd["output_layer.w_1.weight"] = rs.standard_normal((d_output, d_model)) * (2.0 / (d_model + d_output)) ** 0.5
    sd["output_layer.w_1.bias"] = 0.05 * rs.standard_normal(d_output)
    return {k: v.astype(np.float32).astype(np.float16) for k, v in sd.items()}


class OracleLTRHead:
    def __init__(self, n_features: int, fc_model: Optional[dict], post_model: dict, state: Dict[str, np.ndarray],
                 dtype=torch.float32):
        self.fc = fc_model
        self.post = post_model
        self.w = {k: torch.from_numpy(np.asarray(v).astype(np.float32)).to(dtype) for k, v in state.items()}
        self.n_features = n_features
        self.dtype = dtype

    @torch.no_grad()
    def score(self, x: np.ndarray) -> np.ndarray:
        h = torch.from_numpy(np.asarray(x, np.float32)).to(self.dtype)
        if self.fc is not None:                                                  # FCModel.forward, :33-43
            if self.fc.get("input_norm"):
                h = F.layer_norm(h, (self.n_features,), self.w["input_layer.input_norm.weight"],
                                 self.w["input_layer.input_norm.bias"], 1e-5)
            act = _ACT[self.fc.get("activation")]
            i = 0
            while f"input_layer.layers.{i}.weight" in self.w:
                h = act(F.linear(h, self.w[f"input_layer.layers.{i}.weight"], self.w[f"input_layer.layers.{i}.bias"]))
                i += 1
        out = _ACT[self.post.get("output_activation")](                         # OutputLayer.forward, :108-114
            F.linear(h, self.w["output_layer.w_1.weight"], self.w["output_layer.w_1.bias"]))
        if self.post["d_output"] > 1:                                            # OutputLayer.score, :116-125
            return out.sum(-1).float().numpy()
        return out[:, 0].float().numpy()
