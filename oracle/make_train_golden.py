#!/usr/bin/env python3
"""Generate tests/golden/train_steps_*.npz: a few optimisation steps of the reference's fine-tuning recipe
(train/trainer.py:122-165) computed by the pieces the reference itself uses - HF ``OPTForSequenceClassification`` in
fp32 (dropout 0 so that the run is reproducible), the reference's own listMLE / neuralNDCG (loaded by path from
/root/reference/train/allrank, its ``torch.randperm`` replaced by a recorded permutation), ``torch.optim.Adam``.
Runs only in the build container:

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_train_golden.py

Stored per case: the spec, the checkpoint seed, per step (ids, cu_seqlens, labels, shuffle), the loss, the logits
before the update, and - after the last step - every parameter tensor of small models / a fixed sample of entries of
each tensor; plus the gradient of the first step for a few tensors.  No reference source is stored."""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")

from vllm_ltr_amd.opt_spec import OPTSpec, seeded_checkpoint  # noqa: E402


def load_reference_listmle():
    for name in ("allrank", "allrank.data", "allrank.data.dataset_loading", "allrank.models", "allrank.models.losses"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["allrank.data.dataset_loading"].PADDED_Y_VALUE = -1      # allrank/data/dataset_loading.py:31
    sys.modules["allrank.models.losses"].DEFAULT_EPS = 1e-10             # allrank/models/losses/__init__.py:17
    sp = importlib.util.spec_from_file_location("ref_listMLE", "/root/reference/train/allrank/models/losses/listMLE.py")
    mod = importlib.util.module_from_spec(sp)
    sp.loader.exec_module(mod)
    return mod.listMLE


def hf_model(spec: OPTSpec, ckpt):
    from transformers import OPTConfig, OPTForSequenceClassification
    cfg = OPTConfig(**spec.to_hf_config_kwargs(), dropout=0.0, attention_dropout=0.0, pad_token_id=1)
    m = OPTForSequenceClassification(cfg).float()
    sd = {k: torch.from_numpy(v.astype(np.float32)) for k, v in ckpt.items()}
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not [x for x in missing if "lm_head" not in x] and not unexpected, (missing, unexpected)
    return m.train()


def sample_index(size: int, k: int = 2048) -> np.ndarray:
    """Evenly spaced flat indices (all of them for tensors up to k entries)."""
    return np.arange(size) if size <= k else np.linspace(0, size - 1, k).astype(np.int64)


def batch(spec, lens, seed):
    rs = np.random.RandomState(seed)
    ids = np.concatenate([np.r_[2, rs.randint(4, spec.vocab_size, L - 1)] for L in lens]).astype(np.int64)
    cu = np.r_[0, np.cumsum(lens)].astype(np.int32)
    return ids, cu


def run_case(name, spec, seed, loss_name, steps, lens_list, lr, wd):
    listMLE = load_reference_listmle()
    if loss_name == "neuralNDCG":
        from oracle.make_neuralndcg_golden import load_reference_neuralndcg
        neuralNDCG = load_reference_neuralndcg()
    ckpt = seeded_checkpoint(spec, seed)
    m = hf_model(spec, ckpt)
    opt = torch.optim.Adam(m.parameters(), lr=lr, weight_decay=wd)                  # trainer.py:122
    opt.zero_grad()
    out = dict(spec=np.array(list(spec.to_hf_config_kwargs().items()), dtype=object).astype(str), seed=np.int64(seed),
               loss_name=np.array(loss_name), lr=np.float64(lr), weight_decay=np.float64(wd), n_steps=np.int64(steps))
    rs = np.random.RandomState(seed + 100)
    for st in range(steps):
        lens = lens_list[st % len(lens_list)]
        ids, cu = batch(spec, lens, seed * 10 + st)
        n = len(lens)
        # right-padded batch like the trainer's tokenizer(padding=True) call (trainer.py:141-144)
        L = max(lens)
        inp = torch.full((n, L), 1, dtype=torch.long)
        att = torch.zeros((n, L), dtype=torch.long)
        for i in range(n):
            inp[i, :lens[i]] = torch.from_numpy(ids[cu[i]:cu[i + 1]]); att[i, :lens[i]] = 1
        logits = m(input_ids=inp, attention_mask=att).logits                        # PredModel.forward, prefill_predictor.py:76-79
        if loss_name == "listMLE":
            labels = rs.permutation(n).astype(np.float32)                           # distinct labels (see listmle fixture)
            perm = rs.permutation(n)
            real = torch.randperm
            torch.randperm = lambda k, *a, **kw: torch.from_numpy(perm.copy())
            try:
                loss = listMLE(logits.view(1, -1), torch.from_numpy(labels).view(1, -1))     # trainer.py:157
            finally:
                torch.randperm = real
        elif loss_name == "neuralNDCG":
            labels = rs.randint(0, 12, n).astype(np.float32)                        # bucketed lengths (trainer.py:50-52), with ties
            perm = np.arange(n)
            loss = neuralNDCG(logits.view(1, -1), torch.from_numpy(labels).view(1, -1))      # trainer.py:127-128,157
        elif loss_name == "crossentropy":
            labels = rs.randint(0, spec.num_labels, n).astype(np.int64)
            perm = np.arange(n)
            loss = torch.nn.CrossEntropyLoss()(logits.view(-1, spec.num_labels), torch.from_numpy(labels))   # trainer.py:152-155
        else:
            labels = rs.standard_normal(n).astype(np.float32)
            perm = np.arange(n)
            loss = torch.nn.MSELoss()(logits.view(1, -1), torch.from_numpy(labels).view(1, -1))
        loss.backward()                                                             # trainer.py:161
        if st == 0:
            for k, p in m.named_parameters():
                if any(t in k for t in ("score.weight", "layers.0.fc1.weight", "layers.0.self_attn.q_proj.bias",
                                        "embed_positions", "layers.1.self_attn_layer_norm.weight", "project_in",
                                        "layers.0.self_attn.out_proj.weight", "decoder.final_layer_norm.bias")):
                    out[f"grad0::{k}"] = p.grad.detach().numpy().copy()
        opt.step(); opt.zero_grad()                                                 # trainer.py:163-165
        out[f"s{st}_ids"], out[f"s{st}_cu"] = ids, cu
        out[f"s{st}_labels"], out[f"s{st}_shuffle"] = labels, perm.astype(np.int32)
        out[f"s{st}_loss"] = np.float64(loss.item())
        out[f"s{st}_logits"] = logits.detach().numpy().astype(np.float32)
        print(f"{name} step {st}: N={n} T={int(cu[-1])} loss={loss.item():.6f}")
    for k, p in m.named_parameters():
        if "lm_head" in k:
            continue
        v = p.detach().numpy().astype(np.float32).ravel()
        idx = sample_index(v.size)
        out[f"final::{k}"] = v[idx]                                                  # a fixed sample of the entries ...
        out[f"finalsum::{k}"] = np.array([v.astype(np.float64).sum(), (v.astype(np.float64) ** 2).sum()])   # ... + two moments of all
    np.savez_compressed(os.path.join(GOLD, f"train_steps_{name}.npz"), **out)


if __name__ == "__main__":
    torch.manual_seed(0)
    torch.set_num_threads(8)
    a = [[5, 1, 33, 64, 7, 2, 65, 20], [3, 17, 9, 40, 12, 6, 1, 28, 11, 4]]
    # lr far above the trainer's default (2e-5) so that three steps move the weights well above fp32 noise
    run_case("pre_ln_listmle", OPTSpec.tiny_pre_ln(), 41, "listMLE", 3, a, 1e-3, 0.01)
    run_case("post_ln_listmle", OPTSpec.tiny_post_ln(), 42, "listMLE", 3, a, 1e-3, 0.01)
    run_case("pre_ln_class5_ce", OPTSpec.tiny_pre_ln(5), 43, "crossentropy", 2, a, 1e-3, 0.0)
    run_case("post_ln_mse", OPTSpec.tiny_post_ln(), 44, "mse", 2, a, 5e-4, 0.01)
    run_case("pre_ln_neuralndcg", OPTSpec.tiny_pre_ln(), 45, "neuralNDCG", 3, a, 1e-3, 0.01)
