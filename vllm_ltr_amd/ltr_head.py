"""Hidden-state learning-to-rank head on the device (``ltr_head_*`` of libltr_hip.so).

Mirrors ``vllm/model_executor/predictor.py`` (``predictor_model`` / ``LTRModel.score``) and the
``PredictorConfig`` JSON (``vllm/config_predictor.py:78-117``); the state dict uses the reference's
parameter names (``input_layer.input_norm.*``, ``input_layer.layers.<i>.*``, ``output_layer.w_1.*``).
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import numpy as np
import torch

from . import _lib


class HipLTRHead:
    def __init__(self, n_features: int, fc_model: Optional[dict], post_model: dict, state: Dict[str, np.ndarray],
                 device: str = "cuda:0", weight_dtype: str = "f16"):
        if not torch.cuda.is_available():
            raise _lib.LtrError("HipLTRHead needs a ROCm GPU (no CPU fallback on the product path)")
        self.lib = _lib.load()
        self.device = torch.device(device)
        wt = torch.float16 if weight_dtype == "f16" else torch.float32
        f32 = lambda a: torch.from_numpy(np.asarray(a).astype(np.float32)).to(self.device).contiguous()
        mat = lambda a: torch.from_numpy(np.asarray(a).astype(np.float32)).to(self.device, wt).contiguous()
        sizes = list(fc_model["sizes"]) if fc_model else []
        norm = bool(fc_model and fc_model.get("input_norm"))
        t = [f32(state["input_layer.input_norm.weight"]) if norm else None,
             f32(state["input_layer.input_norm.bias"]) if norm else None]
        for i in range(len(sizes)):
            t += [mat(state[f"input_layer.layers.{i}.weight"]), f32(state[f"input_layer.layers.{i}.bias"])]
        t += [mat(state["output_layer.w_1.weight"]), f32(state["output_layer.w_1.bias"])]
        self._tensors = t
        desc = _lib.HeadDesc()
        desc.n_features, desc.n_fc = n_features, len(sizes)
        for i, s in enumerate(sizes):
            desc.fc_sizes[i] = s
        desc.input_norm = 1 if norm else 0
        desc.activation = _lib.ACTIVATIONS[fc_model.get("activation") if fc_model else None]
        desc.d_output = post_model["d_output"]
        desc.output_activation = _lib.ACTIVATIONS[post_model.get("output_activation")]
        desc.weight_dtype = _lib.LTR_W_F16 if weight_dtype == "f16" else _lib.LTR_W_F32
        ptrs = (C.c_void_p * len(t))(*[(x.data_ptr() if x is not None else None) for x in t])
        self._h = C.c_void_p()
        _lib.check(self.lib.ltr_head_create(C.byref(desc), ptrs, len(t), C.byref(self._h)), "ltr_head_create")

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self.lib.ltr_head_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def score(self, hidden: torch.Tensor, row_index: Optional[torch.Tensor] = None) -> torch.Tensor:
        """hidden f32 [rows, n_features] on the device; row_index int32 [N] or None."""
        assert hidden.dtype == torch.float32 and hidden.is_cuda and hidden.is_contiguous()
        n = hidden.shape[0] if row_index is None else row_index.numel()
        out = torch.empty(n, dtype=torch.float32, device=self.device)
        _lib.check(self.lib.ltr_head_score(self._h, hidden.data_ptr(),
                                           row_index.data_ptr() if row_index is not None else None, n,
                                           out.data_ptr(), torch.cuda.current_stream(self.device).cuda_stream),
                   "ltr_head_score")
        return out
