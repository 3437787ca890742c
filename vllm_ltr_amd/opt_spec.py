"""Predictor model description + deterministic seeded checkpoints.

The predictor the `opt-*` schedule types use is an HF-format
``OPTForSequenceClassification`` checkpoint saved in fp16
(reference: train/trainer.py:213-216 ``predictor.model.half().save_pretrained``;
vLLM-side module vllm/model_executor/models/opt.py:362-444).  No trained
checkpoints or tokenizer files exist offline, so tests/bench use *seeded*
checkpoints at the true 125m / 350m shapes.  They are generated from
``numpy.random.RandomState`` (bit-stable across machines) and rounded to fp16,
i.e. they are exactly what such a ``.half()`` checkpoint holds; the CPU oracle
widens the same values to fp32 and the HIP path consumes them as fp16.

Tensor names follow the HF checkpoint (``model.decoder.*`` / ``score.weight``),
which is what opt.py:411-444 ``load_weights`` consumes.
"""
from __future__ import annotations

import dataclasses
import json
import os
from typing import Dict, Iterable, Tuple

import numpy as np


@dataclasses.dataclass(frozen=True)
class OPTSpec:
    """Shape of an OPT sequence-classification predictor (HF ``OPTConfig`` fields)."""
    vocab_size: int = 50272
    hidden_size: int = 768
    ffn_dim: int = 3072
    num_hidden_layers: int = 12
    num_attention_heads: int = 12
    word_embed_proj_dim: int = 768
    max_position_embeddings: int = 2048
    do_layer_norm_before: bool = True
    num_labels: int = 1
    # opt.py:43-53 - learned positions are looked up at position + 2
    POS_OFFSET = 2

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_attention_heads

    @property
    def has_proj(self) -> bool:            # opt.py:205-219
        return self.word_embed_proj_dim != self.hidden_size

    @property
    def has_final_ln(self) -> bool:        # opt.py:221-226
        return self.do_layer_norm_before

    @staticmethod
    def opt_125m(num_labels: int = 1) -> "OPTSpec":
        return OPTSpec(num_labels=num_labels)

    @staticmethod
    def opt_350m(num_labels: int = 1) -> "OPTSpec":
        return OPTSpec(hidden_size=1024, ffn_dim=4096, num_hidden_layers=24,
                       num_attention_heads=16, word_embed_proj_dim=512,
                       do_layer_norm_before=False, num_labels=num_labels)

    @staticmethod
    def tiny_pre_ln(num_labels: int = 1) -> "OPTSpec":
        """125m-style (pre-LN, final LN, De == H) at CI size."""
        return OPTSpec(vocab_size=512, hidden_size=128, ffn_dim=512, num_hidden_layers=2,
                       num_attention_heads=2, word_embed_proj_dim=128,
                       max_position_embeddings=160, do_layer_norm_before=True,
                       num_labels=num_labels)

    @staticmethod
    def tiny_post_ln(num_labels: int = 1) -> "OPTSpec":
        """350m-style (post-LN, project_in/out, no final LN) at CI size."""
        return OPTSpec(vocab_size=512, hidden_size=128, ffn_dim=512, num_hidden_layers=3,
                       num_attention_heads=2, word_embed_proj_dim=64,
                       max_position_embeddings=160, do_layer_norm_before=False,
                       num_labels=num_labels)

    @staticmethod
    def from_hf_config(cfg: dict) -> "OPTSpec":
        return OPTSpec(
            vocab_size=cfg["vocab_size"], hidden_size=cfg["hidden_size"],
            ffn_dim=cfg["ffn_dim"], num_hidden_layers=cfg["num_hidden_layers"],
            num_attention_heads=cfg["num_attention_heads"],
            word_embed_proj_dim=cfg.get("word_embed_proj_dim", cfg["hidden_size"]),
            max_position_embeddings=cfg.get("max_position_embeddings", 2048),
            do_layer_norm_before=cfg.get("do_layer_norm_before", True),
            num_labels=cfg.get("num_labels", len(cfg.get("id2label", {0: 0}))))

    def to_hf_config_kwargs(self) -> dict:
        return dict(vocab_size=self.vocab_size, hidden_size=self.hidden_size,
                    ffn_dim=self.ffn_dim, num_hidden_layers=self.num_hidden_layers,
                    num_attention_heads=self.num_attention_heads,
                    word_embed_proj_dim=self.word_embed_proj_dim,
                    max_position_embeddings=self.max_position_embeddings,
                    do_layer_norm_before=self.do_layer_norm_before,
                    num_labels=self.num_labels)


def tensor_shapes(spec: OPTSpec) -> Iterable[Tuple[str, Tuple[int, ...]]]:
    """HF checkpoint tensor names and shapes, in a fixed order."""
    H, F, De = spec.hidden_size, spec.ffn_dim, spec.word_embed_proj_dim
    yield "model.decoder.embed_tokens.weight", (spec.vocab_size, De)
    yield "model.decoder.embed_positions.weight", (spec.max_position_embeddings + spec.POS_OFFSET, H)
    if spec.has_proj:
        yield "model.decoder.project_in.weight", (H, De)
        yield "model.decoder.project_out.weight", (De, H)
    for i in range(spec.num_hidden_layers):
        p = f"model.decoder.layers.{i}."
        for proj in ("q_proj", "k_proj", "v_proj", "out_proj"):
            yield p + f"self_attn.{proj}.weight", (H, H)
            yield p + f"self_attn.{proj}.bias", (H,)
        yield p + "self_attn_layer_norm.weight", (H,)
        yield p + "self_attn_layer_norm.bias", (H,)
        yield p + "fc1.weight", (F, H)
        yield p + "fc1.bias", (F,)
        yield p + "fc2.weight", (H, F)
        yield p + "fc2.bias", (H,)
        yield p + "final_layer_norm.weight", (H,)
        yield p + "final_layer_norm.bias", (H,)
    if spec.has_final_ln:
        yield "model.decoder.final_layer_norm.weight", (H,)
        yield "model.decoder.final_layer_norm.bias", (H,)
    yield "score.weight", (spec.num_labels, De)


def seeded_checkpoint(spec: OPTSpec, seed: int = 0, std: float = 0.02,
                      qk_std: float = 0.06) -> Dict[str, np.ndarray]:
    """Deterministic fp16 checkpoint at ``spec``'s shapes.

    Weights ~ N(0, std) (HF ``init_std`` = 0.02); q/k projections use a larger
    std so attention is not near-uniform (keeps the parity tests sensitive to
    masking / softmax mistakes); biases and LayerNorm affine terms are non-trivial
    so every epilogue term is exercised.  Values are rounded to fp16.
    """
    rs = np.random.RandomState(seed)
    out: Dict[str, np.ndarray] = {}
    for name, shape in tensor_shapes(spec):
        if name.endswith("layer_norm.weight"):
            w = 1.0 + 0.1 * rs.standard_normal(shape)
        elif name.endswith("layer_norm.bias"):
            w = 0.05 * rs.standard_normal(shape)
        elif ".q_proj.weight" in name or ".k_proj.weight" in name:
            w = qk_std * rs.standard_normal(shape)
        elif name == "score.weight":
            w = 0.05 * rs.standard_normal(shape)
        else:
            w = std * rs.standard_normal(shape)
        out[name] = w.astype(np.float32).astype(np.float16)
    return out


def load_hf_checkpoint(path: str) -> Tuple[OPTSpec, Dict[str, np.ndarray]]:
    """Read an HF ``OPTForSequenceClassification`` directory (config.json +
    model.safetensors or pytorch_model.bin), the format train/trainer.py:213-216
    writes and model_loader/loader.py:114-243 reads.  Names are normalised the
    way opt.py:424-427 does (``decoder.*`` -> ``model.decoder.*``; ``lm_head``
    skipped).  Returned arrays are fp16."""
    with open(os.path.join(path, "config.json")) as f:
        spec = OPTSpec.from_hf_config(json.load(f))
    tensors: Dict[str, np.ndarray] = {}
    st = os.path.join(path, "model.safetensors")
    if os.path.exists(st):
        from safetensors.numpy import load_file
        raw = load_file(st)
    else:
        import torch
        sd = torch.load(os.path.join(path, "pytorch_model.bin"), map_location="cpu",
                        weights_only=True)
        raw = {k: v.float().numpy() for k, v in sd.items()}
    for name, arr in raw.items():
        if "lm_head.weight" in name:
            continue
        if name.startswith("decoder."):
            name = "model." + name
        tensors[name] = np.asarray(arr).astype(np.float16)
    want = dict(tensor_shapes(spec))
    missing = [n for n in want if n not in tensors]
    if missing:
        raise KeyError(f"checkpoint at {path} lacks tensors: {missing[:4]}...")
    for n, shp in want.items():
        if tuple(tensors[n].shape) != tuple(shp):
            raise ValueError(f"{n}: shape {tensors[n].shape} != expected {shp}")
    return spec, {n: tensors[n] for n in want}


def save_hf_checkpoint(path: str, spec: OPTSpec, ckpt: Dict[str, np.ndarray]) -> None:
    """Write ``ckpt`` as an HF directory (safetensors, fp16) - the inverse of
    :func:`load_hf_checkpoint`; used by tests for the loader round trip."""
    from safetensors.numpy import save_file
    os.makedirs(path, exist_ok=True)
    cfg = dict(spec.to_hf_config_kwargs(), model_type="opt",
               architectures=["OPTForSequenceClassification"], torch_dtype="float16")
    with open(os.path.join(path, "config.json"), "w") as f:
        json.dump(cfg, f)
    save_file({k: np.ascontiguousarray(v) for k, v in ckpt.items()},
              os.path.join(path, "model.safetensors"))
