#!/usr/bin/env python3
"""Generate tests/golden/neuralndcg.npz by running the REFERENCE's own neuralNDCG (value + autograd gradient, float32 as
the reference computes it) on seeded slates.  Runs only in the build container (it loads
/root/reference/train/allrank/models/losses/{neuralNDCG,loss_utils}.py and models/metrics.py by path):

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_neuralndcg_golden.py

The package modules those files import and that would drag in the rest of allRank (data loading, GCS helpers, logging) are
replaced by the three things they are used for here: PADDED_Y_VALUE = -1 (allrank/data/dataset_loading.py:31),
DEFAULT_EPS = 1e-10 (allrank/models/losses/__init__.py:17) and get_torch_device() (allrank/models/model_utils.py:29-34:
cuda:0 if there is one, else cpu - this container has none).  Stored per case: y_pred, y_true, loss, grad.  No reference
source is stored."""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
REF = "/root/reference/train/allrank/models"


def load_reference_neuralndcg():
    for name in ("allrank", "allrank.data", "allrank.data.dataset_loading", "allrank.models", "allrank.models.losses",
                 "allrank.models.model_utils"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["allrank.data.dataset_loading"].PADDED_Y_VALUE = -1
    sys.modules["allrank.models.losses"].DEFAULT_EPS = 1e-10
    sys.modules["allrank.models.model_utils"].get_torch_device = lambda: torch.device("cpu")

    def by_path(mod_name, path):
        sp = importlib.util.spec_from_file_location(mod_name, path)
        mod = importlib.util.module_from_spec(sp)
        sys.modules[mod_name] = mod
        sp.loader.exec_module(mod)
        return mod
    by_path("allrank.models.losses.loss_utils", f"{REF}/losses/loss_utils.py")
    by_path("allrank.models.metrics", f"{REF}/metrics.py")
    return by_path("allrank.models.losses.neuralNDCG", f"{REF}/losses/neuralNDCG.py").neuralNDCG


def cases():
    rs = np.random.RandomState(20260929)
    out = {}

    def lens_labels(n, group):       # trainer.py:50-52 with label_max_length 8192
        lens = np.clip(np.rint(np.exp(rs.normal(np.log(300), 1.0, n))), 1, 9000).astype(np.int64)
        return (8192 // group - np.minimum(8192, lens) // group).astype(np.float32)

    out["trainer_b32_group100"] = (rs.normal(0, 1, (1, 32)), lens_labels(32, 100)[None])
    out["trainer_b32_group820"] = (rs.normal(0, 0.3, (1, 32)), lens_labels(32, 820)[None])
    out["trainer_b16_group100"] = (rs.normal(2, 1.5, (1, 16)), lens_labels(16, 100)[None])
    out["small_labels_n8"] = (rs.normal(0, 1, (1, 8)), rs.randint(0, 5, (1, 8)).astype(np.float32))
    t = rs.randint(0, 4, (2, 12)).astype(np.float32); t[0, 9:] = -1; t[1, 5:] = -1
    out["trailing_padding"] = (rs.normal(0, 1, (2, 12)), t)
    t = rs.randint(0, 4, (1, 10)).astype(np.float32); t[0, [2, 6]] = -1
    out["padding_in_the_middle"] = (rs.normal(0, 1, (1, 10)), t)
    t = rs.randint(0, 6, (3, 20)).astype(np.float32); t[1] = 0.0; t[2, 15:] = -1
    out["three_slates_one_without_gain"] = (rs.normal(0, 1, (3, 20)) * np.array([[0.2], [1.0], [3.0]]), t)
    p = rs.normal(0, 1, (1, 12)); p[0, 3] = p[0, 7]; p[0, 1] = p[0, 10] = p[0, 11]
    out["tied_scores"] = (p, rs.randint(0, 5, (1, 12)).astype(np.float32))
    out["sharp_scores"] = (rs.normal(0, 8, (1, 16)), rs.randint(0, 8, (1, 16)).astype(np.float32))
    # (a slate of ONE item is not a case: the reference raises IndexError at loss_utils.py:70, mask.squeeze(-1).sum(dim=1))
    out["all_padded_and_live"] = (rs.normal(0, 1, (2, 6)), np.array([[-1] * 6, [2, 0, 1, -1, -1, -1]], np.float32))
    out["n64_group100"] = (rs.normal(0, 1, (1, 64)), lens_labels(64, 100)[None])
    out["n100_small"] = (rs.normal(0, 2, (2, 100)), rs.randint(0, 10, (2, 100)).astype(np.float32))
    out["label_overflow_group10"] = (rs.normal(0, 1, (1, 8)), np.array([[800, 3, 2, 40, 127, 128, 5, 0]], np.float32))
    out["no_gain_at_all"] = (rs.normal(0, 1, (2, 5)), np.zeros((2, 5), np.float32))
    return out


def main():
    ref = load_reference_neuralndcg()
    store = {}
    for name, (p, t) in cases().items():
        yp = torch.tensor(np.asarray(p, np.float32), requires_grad=True)
        yt = torch.tensor(np.asarray(t, np.float32))
        loss = ref(yp, yt)
        if loss.requires_grad:
            loss.backward()
            g = yp.grad.numpy()
        else:                         # neuralNDCG.py:83-84: a constant 0 when no slate has gain
            g = np.zeros_like(p, dtype=np.float32)
        store[name + "/y_pred"] = np.asarray(p, np.float32)
        store[name + "/y_true"] = np.asarray(t, np.float32)
        store[name + "/loss"] = np.float32(loss.item())
        store[name + "/grad"] = g.astype(np.float32)
        print(f"{name:34s} loss {loss.item(): .6f}  max|grad| {np.abs(g).max():.3e}")
    np.savez_compressed(os.path.join(GOLD, "neuralndcg.npz"), **store)


if __name__ == "__main__":
    main()
