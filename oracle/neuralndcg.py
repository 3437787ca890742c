"""CPU restatement of the reference's NeuralNDCG training loss (TEST INFRASTRUCTURE ONLY).

Reference: ``train/allrank/models/losses/neuralNDCG.py:27-87`` (deterministic variant: what ``train/trainer.py:127-128,157``
runs, ``loss_func(outputs.view(1, -1), labels)`` with every keyword at its default) on top of
``train/allrank/models/losses/loss_utils.py:24-83`` (``deterministic_neural_sort``, ``sinkhorn_scaling``) and
``train/allrank/models/metrics.py:89-135`` (``dcg`` for the ideal DCG).  In the reference's order:

  1. NeuralSort (loss_utils.py:50-83).  With mask = (y_true == pad), s the predictions, n the slate length and m the number
     of padded items:  Asum_c = sum_k |s_c - s_k| over unpadded (c, k);  scaling_r = (n - m + 1) - 2 (r + 1) for r < n - m,
     0 after;  logits[r, c] = s_c * scaling_r - Asum_c  (s_c = 0 where padded),  -inf where exactly one of item r / item c
     is padded, 1 where both are;  P = softmax_c(logits / tau).     (row r = rank, column c = item; the row mask is the
     ITEM mask at index r, loss_utils.py:79 - it equals "rank beyond the valid items" only for trailing padding)
  2. Sinkhorn scaling (loss_utils.py:24-47, tol 1e-6, at most 50 rounds): entries with one padded index := 0, with both := 1;
     a round divides by the column sums, then by the row sums, both clamped at 1e-10; stop after the first round in which
     every row sum and every column sum OF THE WHOLE BATCH is within tol of 1.  Then the padded entries := 0 again.
  3. NDCG (neuralNDCG.py:60-87): gains 2^y - 1 (0 where padded), approximately sorted by P, discounted by 1 / log2(rank + 2),
     cut at k, divided by (ideal DCG@k + 1e-10); slates whose ideal DCG is 0 count as 0 and do not enter the mean;
     loss = -mean.  (All ideal DCGs zero: the reference returns a constant 0.)

The arithmetic is torch's, in the dtype of ``y_pred`` (the reference: float32); gradients come from autograd of this
restatement.  Pinned against the reference's own function - value and autograd gradient - by tests/golden/neuralndcg.npz
(oracle/make_neuralndcg_golden.py imports the reference in the build container).
"""
import math

import numpy as np
import torch

DEFAULT_EPS = 1e-10      # allrank/models/losses/__init__.py:17
PADDED_Y_VALUE = -1      # allrank/data/dataset_loading.py:31
SINKHORN_TOL = 1e-6      # neuralNDCG.py:57
SINKHORN_ROUNDS = 50     # neuralNDCG.py:57


def neural_sort(s: torch.Tensor, pad: torch.Tensor, tau: float) -> torch.Tensor:
    """s [B, n] scores, pad [B, n] bool -> P [B, n, n] (rank x item), loss_utils.py:50-83."""
    B, n = s.shape
    live = ~pad
    both = pad[:, :, None] & pad[:, None, :]
    either = pad[:, :, None] | pad[:, None, :]
    diff = (s[:, :, None] - s[:, None, :]).abs()
    asum = torch.where(either, torch.zeros_like(diff), diff).sum(dim=2)                      # [B, item]
    n_live = live.sum(dim=1, keepdim=True)                                                   # n - m
    r = torch.arange(n, device=s.device)[None, :]
    scaling = torch.where(r < n_live, (n_live + 1 - 2 * (r + 1)).to(s.dtype), torch.zeros((), dtype=s.dtype))
    s0 = torch.where(pad, torch.zeros_like(s), s)
    logits = scaling[:, :, None] * s0[:, None, :] - asum[:, None, :]                         # [B, rank, item]
    logits = torch.where(either, torch.full_like(logits, -math.inf), logits)
    logits = torch.where(both, torch.ones_like(logits), logits)
    return torch.softmax(logits / tau, dim=-1)


def sinkhorn(mat: torch.Tensor, pad: torch.Tensor, tol: float = SINKHORN_TOL, rounds: int = SINKHORN_ROUNDS):
    """loss_utils.py:24-47.  Returns (matrix, rounds run)."""
    both = pad[:, :, None] & pad[:, None, :]
    either = pad[:, :, None] | pad[:, None, :]
    mat = torch.where(either, torch.zeros_like(mat), mat)
    mat = torch.where(both, torch.ones_like(mat), mat)
    done = 0
    for _ in range(rounds):
        mat = mat / mat.sum(dim=1, keepdim=True).clamp(min=DEFAULT_EPS)
        mat = mat / mat.sum(dim=2, keepdim=True).clamp(min=DEFAULT_EPS)
        done += 1
        with torch.no_grad():
            worst = max(float((mat.sum(dim=2) - 1).abs().max()), float((mat.sum(dim=1) - 1).abs().max()))
        if worst < tol:
            break
    return torch.where(either, torch.zeros_like(mat), mat), done


def ideal_dcg(y_true: torch.Tensor, pad: torch.Tensor, k: int) -> torch.Tensor:
    """metrics.py:99-135 with y_pred = y_true: labels sorted descending (padded ones count as label 0, last)."""
    lab = torch.where(pad, torch.zeros_like(y_true), y_true)
    key = torch.where(pad, torch.full_like(y_true, -math.inf), y_true)
    order = key.sort(descending=True, dim=-1).indices
    top = torch.gather(lab, 1, order)
    disc = 1.0 / torch.log2(torch.arange(y_true.shape[1], dtype=torch.float32) + 2.0)
    return ((torch.pow(2, top) - 1) * disc.to(y_true.dtype))[:, :k].sum(dim=1)


def neuralndcg_torch(y_pred: torch.Tensor, y_true: torch.Tensor, pad_value: float = PADDED_Y_VALUE, tau: float = 1.0, k=None,
                     info: dict = None) -> torch.Tensor:
    """y_pred, y_true [B, n] -> scalar loss (differentiable in y_pred)."""
    B, n = y_true.shape
    k = n if k is None else min(int(k), n)
    pad = y_true == pad_value
    P, done = sinkhorn(neural_sort(y_pred, pad, tau), pad)
    if info is not None:
        info["rounds"] = done
    gain = torch.pow(2.0, torch.where(pad, torch.zeros_like(y_true), y_true)) - 1.0
    approx_sorted = (P * gain[:, None, :]).sum(dim=2)                                        # [B, rank]
    disc = (1.0 / torch.log2(torch.arange(n, dtype=torch.float32) + 2.0)).to(y_pred.dtype)
    dcg_hat = (approx_sorted * disc)[:, :k].sum(dim=1)
    idcg = ideal_dcg(y_true, pad, k)
    dead = idcg == 0
    if bool(dead.all()):
        return torch.zeros((), dtype=y_pred.dtype)
    ndcg = torch.where(dead, torch.zeros_like(dcg_hat), dcg_hat / (idcg + DEFAULT_EPS))
    return -ndcg.sum() / (~dead).sum()


def neuralndcg(y_pred, y_true, pad_value: float = PADDED_Y_VALUE, tau: float = 1.0, k=None, dtype=torch.float64):
    """numpy in, (loss, grad [B, n], sinkhorn rounds) out; arithmetic in `dtype`."""
    p = torch.tensor(np.asarray(y_pred), dtype=dtype, requires_grad=True)
    t = torch.tensor(np.asarray(y_true), dtype=dtype)
    info = {}
    loss = neuralndcg_torch(p, t, pad_value, tau, k, info)
    if loss.requires_grad:
        loss.backward()
        g = p.grad.numpy()
    else:
        g = np.zeros(p.shape)
    return float(loss.detach()), g, info.get("rounds", 0)
