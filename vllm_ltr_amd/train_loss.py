"""The reference's ranking losses on the device: ListMLE (``ltr_listmle``) and NeuralNDCG (``ltr_neuralndcg``) of libltr_hip.so.

Mirrors ``train/allrank/models/losses/listMLE.py`` (the ``loss_func`` of ``train/trainer.py:125-150``):
``listmle(y_pred, y_true)`` returns the loss and, on request, its gradient w.r.t. ``y_pred``; the random
shuffle of listMLE.py:33 is drawn here (or passed in) so a call is reproducible.  ``neuralndcg(y_pred, y_true)`` mirrors
``train/allrank/models/losses/neuralNDCG.py:27-87`` (``--loss neuralNDCG``, trainer.py:127-128) the same way."""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import _lib

DEFAULT_EPS = 1e-10      # allrank/models/losses/__init__.py:17
PADDED_Y_VALUE = -1      # allrank/data/dataset_loading.py:31


def listmle(y_pred: torch.Tensor, y_true: torch.Tensor, shuffle: Optional[torch.Tensor] = None,
            eps: float = DEFAULT_EPS, padded_value_indicator: float = PADDED_Y_VALUE,
            with_grad: bool = True) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """y_pred, y_true: f32 [B, S] on the GPU.  Returns (loss f32[1] on the device, grad f32 [B, S] | None)."""
    if not y_pred.is_cuda:
        raise _lib.LtrError("listmle needs device tensors (no CPU fallback on the product path)")
    assert y_pred.shape == y_true.shape and y_pred.dim() == 2
    B, S = y_pred.shape
    dev = y_pred.device
    yp = y_pred.detach().to(torch.float32).contiguous()
    yt = y_true.to(device=dev, dtype=torch.float32).contiguous()
    if shuffle is None:
        shuffle = torch.randperm(S, device=dev)                      # listMLE.py:33
    sh = shuffle.to(device=dev, dtype=torch.int32).contiguous()
    loss = torch.empty(1, dtype=torch.float32, device=dev)
    rows = torch.empty(max(B, 1), dtype=torch.float32, device=dev)
    grad = torch.empty(B, S, dtype=torch.float32, device=dev) if with_grad else None
    lib = _lib.load()
    _lib.check(lib.ltr_listmle(yp.data_ptr(), yt.data_ptr(), sh.data_ptr(), B, S, float(eps),
                               float(padded_value_indicator), loss.data_ptr(), rows.data_ptr(),
                               grad.data_ptr() if grad is not None else None,
                               torch.cuda.current_stream(dev).cuda_stream), "ltr_listmle")
    return loss, grad


def neuralndcg(y_pred: torch.Tensor, y_true: torch.Tensor, padded_value_indicator: float = PADDED_Y_VALUE,
               temperature: float = 1.0, k: Optional[int] = None, stochastic: bool = False,
               with_grad: bool = True) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """neuralNDCG.py:27-87 with ``powered_relevancies=True`` and the deterministic NeuralSort (what trainer.py:157 runs:
    every keyword at its default).  y_pred, y_true: f32 [B, S] on the GPU, 2 <= S <= 1024.
    Returns (loss f32[1] on the device, grad f32 [B, S] | None)."""
    if stochastic:
        raise NotImplementedError("neuralndcg: the stochastic variant (Gumbel-perturbed NeuralSort, neuralNDCG.py:51-53) is not "
                                  "built - train/trainer.py never selects it")
    if not y_pred.is_cuda:
        raise _lib.LtrError("neuralndcg needs device tensors (no CPU fallback on the product path)")
    assert y_pred.shape == y_true.shape and y_pred.dim() == 2
    B, S = y_pred.shape
    dev = y_pred.device
    yp = y_pred.detach().to(torch.float32).contiguous()
    yt = y_true.to(device=dev, dtype=torch.float32).contiguous()
    lib = _lib.load()
    loss = torch.empty(1, dtype=torch.float32, device=dev)
    rows = torch.empty(max(B, 1), dtype=torch.float32, device=dev)
    grad = torch.empty(B, S, dtype=torch.float32, device=dev) if with_grad else None
    nbytes = int(lib.ltr_neuralndcg_workspace_bytes(B, S))
    ws = torch.empty(max(nbytes, 4), dtype=torch.uint8, device=dev)
    _lib.check(lib.ltr_neuralndcg(yp.data_ptr(), yt.data_ptr(), B, S, float(temperature), int(k or 0),
                                  float(padded_value_indicator), loss.data_ptr(), rows.data_ptr(),
                                  grad.data_ptr() if grad is not None else None, ws.data_ptr(), nbytes,
                                  torch.cuda.current_stream(dev).cuda_stream), "ltr_neuralndcg")
    return loss, grad
