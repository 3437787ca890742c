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
    def from_hf_config(cfg: dict, num_labels: int = 0) -> "OPTSpec":
        """``num_labels``: rows of ``score.weight`` when known.  ``save_pretrained`` OMITS ``num_labels`` and
        ``id2label`` from config.json when they equal HF's default of two labels, so without the checkpoint the
        fallback is ``num_labels`` -> ``len(id2label)`` -> 2 (PretrainedConfig's default)."""
        if not num_labels:
            num_labels = cfg.get("num_labels") or (len(cfg["id2label"]) if cfg.get("id2label") else 2)
        return OPTSpec(
            vocab_size=cfg["vocab_size"], hidden_size=cfg["hidden_size"],
            ffn_dim=cfg["ffn_dim"], num_hidden_layers=cfg["num_hidden_layers"],
            num_attention_heads=cfg["num_attention_heads"],
            word_embed_proj_dim=cfg.get("word_embed_proj_dim", cfg["hidden_size"]),
            max_position_embeddings=cfg.get("max_position_embeddings", 2048),
            do_layer_norm_before=cfg.get("do_layer_norm_before", True),
            num_labels=int(num_labels))

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


# parameters of the OPT-350m structured fixture (see structured_checkpoint)
STRUCTURED_350M = dict(outlier=(40.0, -55.0), std=0.02, qk_std=0.06)


def structured_checkpoint(spec: OPTSpec, seed: int = 0, outlier=(40.0, -55.0), std: float = 0.1,
                          qk_std: float = 0.15) -> Dict[str, np.ndarray]:
    """A seeded fp16 checkpoint with the STRUCTURE trained OPT predictors have and N(0, 0.02) weights lack: weights at 5x the
    init scale (q / k at 7.5x: peaked attention), two "massive activation" channels in the token and position embeddings
    (offsets +40 and -55 with a spread of 3), LayerNorm gains spread over [0.2, 3] - the residual stream spans three orders
    of magnitude and the LayerNorm statistics are dominated by two channels.  Used by the reference-run fixtures
    ``tests/golden/outlier_opt125m_64.npz`` / ``outlier_opt350m_48.npz`` (oracle/make_config1_golden.py --config outlier /
    outlier350; the post-LN family keeps the init scale of the weights - ``STRUCTURED_350M`` - because a deep post-LN stack with
    3-5x weights forgets its input: the 48 reference scores came out within 2e-4 of each other) and their GPU tests."""
    out = seeded_checkpoint(spec, seed, std=std, qk_std=qk_std)
    rs = np.random.RandomState(seed + 1000)
    for name in ("model.decoder.embed_tokens.weight", "model.decoder.embed_positions.weight"):
        w = out[name].astype(np.float32)
        ch = [3, 17] if w.shape[1] > 17 else [0, 1]
        w[:, ch] += 3.0 * rs.standard_normal((w.shape[0], 2)).astype(np.float32) + np.array(outlier, np.float32)
        out[name] = w.astype(np.float16)
    for name in list(out):
        if name.endswith("layer_norm.weight"):
            out[name] = rs.uniform(0.2, 3.0, out[name].shape).astype(np.float32).astype(np.float16)
    return out


def _read_hf_tensors(path: str) -> Dict[str, "np.ndarray"]:
    """Every tensor of an HF checkpoint directory, as torch writes them: ``model.safetensors``, sharded
    ``model-0000x-of-0000y.safetensors`` + ``model.safetensors.index.json``, ``pytorch_model.bin``, or sharded
    ``pytorch_model-*.bin`` + ``pytorch_model.bin.index.json`` (model_loader/weight_utils.py of the reference
    iterates the same file set)."""
    import torch
    files = []
    for index, single in (("model.safetensors.index.json", "model.safetensors"),
                          ("pytorch_model.bin.index.json", "pytorch_model.bin")):
        if os.path.exists(os.path.join(path, index)):
            with open(os.path.join(path, index)) as f:
                files = sorted(set(json.load(f)["weight_map"].values()))
            break
        if os.path.exists(os.path.join(path, single)):
            files = [single]
            break
    if not files:
        raise FileNotFoundError(f"{path}: no model.safetensors / pytorch_model.bin (or their index.json) found")
    raw = {}
    for fn in files:
        fp = os.path.join(path, fn)
        if fn.endswith(".safetensors"):
            from safetensors.torch import load_file
            part = load_file(fp)
        else:
            part = torch.load(fp, map_location="cpu", weights_only=True)
        for k, v in part.items():
            if k in raw:
                raise ValueError(f"{path}: tensor {k} appears in more than one shard")
            raw[k] = v
    out = {}
    for k, v in raw.items():
        if v.dtype not in (torch.float16, torch.float32):
            # the f16 path relies on the weights being EXACT in fp16 and the f32 path on exact f32: a silent cast of
            # bf16 / f64 would change the scores
            raise ValueError(f"{path}: tensor {k} has dtype {v.dtype}; only fp16 (trainer.py:213-216 saves .half()) "
                             "and fp32 checkpoints are supported - convert explicitly")
        out[k] = v.contiguous().numpy()
    return out


def load_hf_checkpoint(path: str) -> Tuple[OPTSpec, Dict[str, np.ndarray]]:
    """Read an HF ``OPTForSequenceClassification`` directory - the format train/trainer.py:213-216 writes and
    model_loader/loader.py:114-243 reads: config.json + (sharded) safetensors or pytorch_model.bin.  Names are
    normalised the way opt.py:424-427 does (``decoder.*`` -> ``model.decoder.*``; ``lm_head.weight`` skipped).
    Returned arrays keep the checkpoint's dtype (fp16 or fp32; anything else is refused)."""
    with open(os.path.join(path, "config.json")) as f:
        cfg = json.load(f)
    tensors: Dict[str, np.ndarray] = {}
    for name, arr in _read_hf_tensors(path).items():
        if "lm_head.weight" in name:                     # opt.py:421-422
            continue
        if name.startswith("decoder."):                  # opt.py:424-425
            name = "model." + name
        tensors[name] = arr
    if "score.weight" not in tensors:
        raise KeyError(f"checkpoint at {path} has no score.weight: not an OPTForSequenceClassification (opt.py:374)")
    spec = OPTSpec.from_hf_config(cfg, num_labels=tensors["score.weight"].shape[0])
    stated = cfg.get("num_labels") or (len(cfg["id2label"]) if cfg.get("id2label") else None)
    if stated is not None and stated != spec.num_labels:
        raise ValueError(f"{path}: config.json states {stated} labels, score.weight has {spec.num_labels} rows")
    want = dict(tensor_shapes(spec))
    missing = [n for n in want if n not in tensors]
    if missing:
        raise KeyError(f"checkpoint at {path} lacks tensors: {missing[:4]}...")
    for n, shp in want.items():
        if tuple(tensors[n].shape) != tuple(shp):
            raise ValueError(f"{n}: shape {tensors[n].shape} != expected {shp}")
    return spec, {n: tensors[n] for n in want}


def checkpoint_weight_dtype(ckpt: Dict[str, np.ndarray]) -> str:
    """"f16" when every matrix / table of the checkpoint is fp16 (the ``.half()`` checkpoints of the reference's
    trainer), else "f32" - the scorer mode that keeps the weights exact."""
    mats = [v for k, v in ckpt.items() if np.asarray(v).ndim == 2]
    return "f16" if all(np.asarray(v).dtype == np.float16 for v in mats) else "f32"


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
