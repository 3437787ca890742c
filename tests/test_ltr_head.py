"""Hidden-state LTR head: oracle vs the reference's vectors (CPU) and the HIP
kernel vs both (GPU)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle.ltr_head import OracleLTRHead, seeded_head_weights
from util import GOLDEN

CASES = ["small_relu", "wide_tanh_sum", "no_fc", "gelu_nonorm"]


def _load(name):
    z = np.load(os.path.join(GOLDEN, f"ltr_head_{name}.npz"))
    cfg = json.loads(str(z["cfg"]))
    nf, fc, post = int(z["n_features"]), cfg["fc_model"], cfg["post_model"]
    sd = seeded_head_weights(nf, fc["sizes"] if fc else None, bool(fc and fc["input_norm"]), post["d_output"],
                             int(z["seed"]))
    return z, nf, fc, post, sd


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_reference(name):
    z, nf, fc, post, sd = _load(name)
    got = OracleLTRHead(nf, fc, post, sd).score(z["x"])
    np.testing.assert_allclose(got, z["score"], atol=1e-6, rtol=0)


def test_predictor_config_mirror(tmp_path):
    from vllm_ltr_amd.config_predictor import PredictorConfig
    raw = {"model": {"fc_model": {"sizes": [64], "input_norm": True, "activation": "ReLU", "dropout": 0.0},
                     "transformer": None, "post_model": {"d_output": 1, "output_activation": None},
                     "path": "/x/head.pt", "n_features": 4096, "pred_layer_idx": 31}}
    p = tmp_path / "cfg.json"
    p.write_text(json.dumps(raw))
    c = PredictorConfig.from_json(str(p))
    assert c.model.n_features == 4096 and c.model.fc_model["sizes"] == [64] and c.model.pred_layer_idx == 31


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["f16", "f32"])
@pytest.mark.parametrize("name", CASES)
def test_hip_head_matches_reference(name, mode):
    from vllm_ltr_amd.ltr_head import HipLTRHead
    z, nf, fc, post, sd = _load(name)
    head = HipLTRHead(nf, fc, post, sd, "cuda:0", mode)
    x = torch.from_numpy(z["x"]).cuda()
    got = head.score(x).cpu().numpy()
    err = np.abs(got - z["score"]).max()
    print(f"ltr_head {name}/{mode}: max|d| = {err:.2e}")
    assert err <= 1e-4
    idx = torch.tensor([5, 0, 39, 7], dtype=torch.int32, device="cuda")        # index_select path
    np.testing.assert_allclose(head.score(x, idx).cpu().numpy(), z["score"][[5, 0, 39, 7]], atol=1e-4, rtol=0)
