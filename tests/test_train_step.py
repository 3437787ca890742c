"""Predictor fine-tuning step (train/trainer.py:122-165).

CPU: the oracle (oracle/train_step.py: the scorer's arithmetic under torch autograd + torch.optim.Adam) against steps
computed by the pieces the reference uses - HF OPTForSequenceClassification, the reference's own listMLE, torch Adam
(tests/golden/train_steps_*.npz, oracle/make_train_golden.py).
GPU (-m gpu): the HIP training step (ltr_train_step through the C ABI) against both."""
import os

import numpy as np
import pytest
import torch

from oracle.train_step import OracleTrainer, listmle_torch
from util import GOLDEN, spec_from_npz
from vllm_ltr_amd.opt_spec import seeded_checkpoint

CASES = ["pre_ln_listmle", "post_ln_listmle", "pre_ln_class5_ce", "post_ln_mse"]


def sample_index(size: int, k: int = 2048) -> np.ndarray:
    return np.arange(size) if size <= k else np.linspace(0, size - 1, k).astype(np.int64)


def _noise_driven(name: str, wd: float) -> bool:
    """The gradient of k_proj.bias is mathematically zero (softmax is invariant to a constant added to every score of
    a row), so without weight decay Adam's m / (sqrt(v) + eps) turns fp32 rounding noise into +-lr steps: its
    trajectory is not comparable between implementations.  (With weight decay the L2 term dominates.)"""
    return wd == 0.0 and name.endswith("k_proj.bias")


def _load(case):
    z = np.load(os.path.join(GOLDEN, f"train_steps_{case}.npz"))
    return z, spec_from_npz(z), str(z["loss_name"]), float(z["lr"]), float(z["weight_decay"]), int(z["n_steps"])


def test_torch_listmle_restatement_matches_reference_fixture():
    z = np.load(os.path.join(GOLDEN, "listmle.npz"))
    for n in [str(x) for x in z["names"]]:
        yp = torch.tensor(z[f"{n}_pred"], requires_grad=True)
        loss = listmle_torch(yp, torch.tensor(z[f"{n}_true"]), torch.as_tensor(z[f"{n}_perm"].astype(np.int64)))
        loss.backward()
        assert abs(loss.item() - float(z[f"{n}_loss"])) <= 2e-6 * abs(float(z[f"{n}_loss"])), n
        np.testing.assert_allclose(yp.grad.numpy(), z[f"{n}_grad"], atol=2e-6, rtol=1e-5)


@pytest.mark.parametrize("case", CASES)
def test_oracle_trainer_reproduces_hf_plus_adam_steps(case):
    z, spec, loss_name, lr, wd, n_steps = _load(case)
    tr = OracleTrainer(spec, seeded_checkpoint(spec, int(z["seed"])), lr=lr, weight_decay=wd, loss=loss_name)
    for st in range(n_steps):
        loss, logits, grads = tr.step(z[f"s{st}_ids"], z[f"s{st}_cu"], z[f"s{st}_labels"], z[f"s{st}_shuffle"])
        assert abs(loss - float(z[f"s{st}_loss"])) <= 1e-5 * max(1.0, abs(float(z[f"s{st}_loss"]))), (case, st)
        np.testing.assert_allclose(logits, z[f"s{st}_logits"], atol=2e-5, rtol=0)
        if st == 0:
            for key in [k for k in z.files if k.startswith("grad0::")]:
                name = key[len("grad0::"):]
                g = grads[name].numpy()
                np.testing.assert_allclose(g, z[key], atol=2e-6 + 2e-5 * np.abs(z[key]).max(), rtol=0, err_msg=name)
    final = tr.state()
    for key in [k for k in z.files if k.startswith("final::")]:
        name = key[len("final::"):]
        if _noise_driven(name, wd):
            continue
        v = final[name].ravel()
        # three Adam steps at lr 1e-3 move every weight by ~3e-3; the update direction m / sqrt(v) amplifies fp32
        # gradient noise where |g| is tiny, so compare with a tolerance of 1 % of one step
        np.testing.assert_allclose(v[sample_index(v.size)], z[key], atol=lr * 0.02 * n_steps, rtol=0, err_msg=name)
        s = z[f"finalsum::{name}"]
        assert abs(v.astype(np.float64).sum() - s[0]) <= lr * 0.02 * n_steps * v.size
