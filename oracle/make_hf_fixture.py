#!/usr/bin/env python3
"""Generate tests/golden/hf_tiny/*: checkpoint DIRECTORIES as HuggingFace itself writes them for a tiny seeded
``OPTForSequenceClassification`` - what ``train/trainer.py:213-216`` (``predictor.model.half().save_pretrained``)
produces and what the reference's loader consumes (``vllm/model_executor/models/opt.py:411-444``) - plus the
logits HF computes on fixed inputs.  Test infrastructure; runs in the build container (transformers 5.15):

    python oracle/make_hf_fixture.py

Variants (same 2-label weights in the first three):
  st/       ``.half().save_pretrained()``: config.json + model.safetensors.  With the default 2 labels HF omits
            ``num_labels`` / ``id2label`` from config.json - the loader must size the head from ``score.weight``.
  sharded/  ``save_pretrained(max_shard_size=...)``: model-0000x-of-0000y.safetensors + model.safetensors.index.json
  bin/      ``pytorch_model.bin`` (``torch.save(state_dict)``, what transformers 4.40 - the reference's pin - wrote),
            with the legacy ``decoder.*`` key prefix and a stray ``lm_head.weight`` (opt.py:424-427 handles both)
  rank1/    a 1-label (rank mode) model, post-LN with project_in/out (350m style), fp16 safetensors
  fp32/     the rank1 model saved without ``.half()``
"""
import json
import os
import shutil
import sys

import numpy as np
import torch
from transformers import OPTConfig, OPTForSequenceClassification

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden", "hf_tiny")


def inputs(vocab, seed):
    rs = np.random.RandomState(seed)
    lens = [1, 2, 7, 33, 64, 5]
    ids = np.concatenate([np.r_[2, rs.randint(4, vocab, L - 1)] for L in lens]).astype(np.int64)
    cu = np.r_[0, np.cumsum(lens)].astype(np.int32)
    return ids, cu


def hf_logits(model, ids, cu):
    model = model.float().eval()
    out = []
    with torch.no_grad():
        for i in range(len(cu) - 1):
            x = torch.from_numpy(ids[cu[i]:cu[i + 1]])[None]
            out.append(model(input_ids=x, attention_mask=torch.ones_like(x)).logits[0])
    return torch.stack(out).numpy().astype(np.float32)


def main():
    shutil.rmtree(OUT, ignore_errors=True)
    os.makedirs(OUT)
    torch.manual_seed(0)
    cfg = OPTConfig(vocab_size=128, hidden_size=64, ffn_dim=128, num_hidden_layers=2, num_attention_heads=1,
                    word_embed_proj_dim=64, max_position_embeddings=64, do_layer_norm_before=True, num_labels=2,
                    pad_token_id=1)
    m = OPTForSequenceClassification(cfg)
    with torch.no_grad():                      # non-trivial LayerNorm affine terms and biases
        for n, p in m.named_parameters():
            if "layer_norm" in n or n.endswith(".bias"):
                p.add_(0.05 * torch.randn_like(p))
    m = m.half()
    m.save_pretrained(os.path.join(OUT, "st"))
    m.save_pretrained(os.path.join(OUT, "sharded"), max_shard_size="60KB")
    os.makedirs(os.path.join(OUT, "bin"))
    shutil.copy(os.path.join(OUT, "st", "config.json"), os.path.join(OUT, "bin", "config.json"))
    sd = {(k[len("model."):] if k.startswith("model.decoder.") else k): v for k, v in m.state_dict().items()}
    sd["lm_head.weight"] = m.state_dict()["model.decoder.embed_tokens.weight"].clone()
    torch.save(sd, os.path.join(OUT, "bin", "pytorch_model.bin"))
    ids, cu = inputs(128, 1)
    np.savez(os.path.join(OUT, "expected_class2.npz"), ids=ids, cu_seqlens=cu, logits=hf_logits(m, ids, cu))

    torch.manual_seed(1)
    cfg1 = OPTConfig(vocab_size=128, hidden_size=64, ffn_dim=128, num_hidden_layers=3, num_attention_heads=1,
                     word_embed_proj_dim=32, max_position_embeddings=64, do_layer_norm_before=False, num_labels=1,
                     pad_token_id=1)
    m1 = OPTForSequenceClassification(cfg1)
    with torch.no_grad():
        for n, p in m1.named_parameters():
            if "layer_norm" in n or n.endswith(".bias"):
                p.add_(0.05 * torch.randn_like(p))
    m1.save_pretrained(os.path.join(OUT, "fp32"))
    ids1, cu1 = inputs(128, 2)
    np.savez(os.path.join(OUT, "expected_rank1_fp32.npz"), ids=ids1, cu_seqlens=cu1, logits=hf_logits(m1, ids1, cu1))
    m1 = m1.half()
    m1.save_pretrained(os.path.join(OUT, "rank1"))
    np.savez(os.path.join(OUT, "expected_rank1.npz"), ids=ids1, cu_seqlens=cu1, logits=hf_logits(m1, ids1, cu1))
    for d in sorted(os.listdir(OUT)):
        p = os.path.join(OUT, d)
        print(d, sorted(os.listdir(p)) if os.path.isdir(p) else os.path.getsize(p))
    print(json.load(open(os.path.join(OUT, "st", "config.json"))).get("num_labels", "config.json of st/: no num_labels key"))


if __name__ == "__main__":
    sys.exit(main())
