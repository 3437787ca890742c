"""CPU restatement of the reference's ListMLE training loss (TEST INFRASTRUCTURE ONLY).

Reference: ``train/allrank/models/losses/listMLE.py:23-54`` as used by ``train/trainer.py:125-150``
(``loss_func(outputs.view(1, -1), labels)``): shuffle the slate, sort by the true label (descending),
mask padded items (``y_true == -1``), ``loss = mean_b sum_i [log(sum_{j>=i} exp(p_j - max) + eps) - (p_i - max)]``.

Tie rule: the reference shuffles "for randomised tie resolution" and then calls ``torch.sort``, whose
order among equal labels is implementation specific (torch 2.10 CPU: neither stable nor reversed), so
with ties its value is one random sample by design.  Here ties keep the shuffled order (a stable
descending sort); the golden fixtures use distinct labels (+ padding), where the reference is
deterministic.  The shuffle permutation is an INPUT (``random_indices`` of listMLE.py:33).
"""
import numpy as np

DEFAULT_EPS = 1e-10      # allrank/models/losses/__init__.py:17
PADDED_Y_VALUE = -1      # allrank/data/dataset_loading.py:31


def listmle(y_pred, y_true, shuffle, eps=DEFAULT_EPS, pad=PADDED_Y_VALUE, with_grad=True):
    """y_pred, y_true: [B, S]; shuffle: permutation of range(S).  Returns (loss, grad [B, S] | None)
    in float64 arithmetic."""
    y_pred = np.asarray(y_pred, np.float64)
    y_true = np.asarray(y_true, np.float64)
    B, S = y_pred.shape
    shuffle = np.asarray(shuffle, np.int64)
    ps, ts = y_pred[:, shuffle], y_true[:, shuffle]                       # listMLE.py:33-35
    loss = 0.0
    grad = np.zeros((B, S), np.float64) if with_grad else None
    for b in range(B):
        order = np.argsort(-ts[b], kind="stable")                        # :37 sort(descending=True)
        t, p = ts[b, order], ps[b, order].copy()
        mask = t == pad                                                  # :39
        p[mask] = -np.inf                                                # :42
        m = p.max()                                                      # :44
        with np.errstate(invalid="ignore"):                              # a fully padded slate: -inf - -inf, masked below
            q = p - m
            e = np.where(mask, 0.0, np.exp(q))
        c = np.cumsum(e[::-1])[::-1]                                     # :48 reversed cumsum
        obs = np.log(c + eps) - q                                        # :50
        obs[mask] = 0.0
        loss += obs.sum()
        if with_grad:
            w = np.where(mask, 0.0, 1.0 / (c + eps))
            g = e * np.cumsum(w) - np.where(mask, 0.0, 1.0)              # d/dp_j of sum_i obs_i with the max held fixed
            g[mask] = 0.0
            # ... plus the path through max_pred_values (:44-46), which autograd routes to the arg-max item:
            # d/dm [log(C_i + eps) - (p_i - m)] = 1 - C_i / (C_i + eps) = eps / (C_i + eps).  Negligible while
            # C_i >> eps; ~1 per trailing item once exp(p_i - max) < eps (score spread above ~23).
            if not mask.all():
                g[int(np.argmax(p))] += float((np.where(mask, 0.0, eps / (c + eps))).sum())
            gb = np.zeros(S)
            gb[order] = g                                                # back to shuffled positions
            grad[b, shuffle] = gb / B                                    # and to the original columns
    return loss / B, grad
