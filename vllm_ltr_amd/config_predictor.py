"""Predictor JSON config - same schema, names and behaviour as the reference's
``vllm/config_predictor.py`` (``PrefillModelConfig`` :39-47,
``PrefillPredictorConfig.from_json/from_dict/to_json`` :136-154) so the files under
``train/configs/*.txt`` and the ``usage_config.json`` written by
``train/trainer.py:203-216`` load unchanged.  Written with dataclasses (attrs is not a
dependency of this package); unknown keys raise ``TypeError`` like attrs does.
"""
from __future__ import annotations

import dataclasses
import json
from typing import Optional


@dataclasses.dataclass
class PrefillModelConfig:
    pred_model: str                 # HF id of the base predictor, also its tokenizer
    num_labels: int                 # 1 for mtype "rank"; bucket count for "class"
    mtype: str                      # "rank" | "class"
    activation: Optional[str]       # torch.nn activation name applied to rank logits, or None
    path: str = ""                  # fine-tuned checkpoint dir (HF format)
    max_length: int = 1024          # prompt truncation (aux_llm_engine.py:365-369)
    max_batch_size: int = 512


@dataclasses.dataclass
class PrefillPredictorConfig:
    model: PrefillModelConfig

    @classmethod
    def from_json(cls, config_path):
        with open(config_path) as config_file:
            config = json.load(config_file)
            return PrefillPredictorConfig.from_dict(config)

    @classmethod
    def from_dict(cls, config):
        config = dict(config)
        config["model"] = PrefillModelConfig(**config["model"])
        return cls(**config)

    @classmethod
    def to_json(cls, config, config_path):
        content = {"model": dict(config.model.__dict__)}
        with open(config_path, "w") as outfile:
            json.dump(content, outfile)


# ---- hidden-state predictor head (vllm/config_predictor.py:17-38, 78-117) ---------------------
@dataclasses.dataclass
class FCConfig:
    sizes: list
    input_norm: bool
    activation: Optional[str]
    dropout: Optional[float]


@dataclasses.dataclass
class PostModelConfig:
    d_output: int
    output_activation: Optional[str]


@dataclasses.dataclass
class ModelConfig:
    fc_model: Optional[dict]          # kept as the raw dict like the reference (it is **-expanded later)
    transformer: Optional[dict]
    post_model: dict
    path: str = ""
    n_features: int = 4096
    pred_layer_idx: int = 31


@dataclasses.dataclass
class PredictorConfig:
    model: ModelConfig

    @classmethod
    def from_json(cls, config_path):
        with open(config_path) as config_file:
            return PredictorConfig.from_dict(json.load(config_file))

    @classmethod
    def from_dict(cls, config):
        config = dict(config)
        config["model"] = ModelConfig(**config["model"])
        return cls(**config)
