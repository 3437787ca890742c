"""CPU-only: host logic, config/schedule-string mirrors, checkpoint IO, and that the
C-ABI library loads and exports every symbol include/ltr_hip.h declares."""
import json
import os
import re

import numpy as np
import pytest

from conftest import GOLDEN, ROOT
from vllm_ltr_amd import _lib
from vllm_ltr_amd.config_predictor import PrefillPredictorConfig
from vllm_ltr_amd.opt_spec import (OPTSpec, load_hf_checkpoint, save_hf_checkpoint, seeded_checkpoint,
                                   tensor_shapes)
from vllm_ltr_amd.schedule_type import parse_schedule_type


def _built():
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()


def test_library_exports_every_declared_symbol():
    _built()
    lib = _lib.load()
    header = open(os.path.join(ROOT, "include", "ltr_hip.h")).read()
    declared = set(re.findall(r"\b(ltr_[a-z_]+)\s*\(", header))
    declared -= {"ltr_model_desc"}
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.ltr_abi_version() == _lib.ABI_VERSION


def test_library_argument_errors_without_gpu():
    """Argument validation paths never touch the device."""
    import ctypes as C
    _built()
    lib = _lib.load()
    h = C.c_void_p()
    desc = _lib.ModelDesc(512, 100, 512, 2, 2, 128, 162, 1, 1, _lib.LTR_W_F16)     # head size != 64
    ptrs = (C.c_void_p * 31)()
    assert lib.ltr_create(C.byref(desc), ptrs, 31, None, C.byref(h)) == -22
    assert b"head size" in lib.ltr_last_error()
    desc = _lib.ModelDesc(512, 128, 512, 2, 2, 128, 162, 1, 1, _lib.LTR_W_F16)
    assert lib.ltr_create(C.byref(desc), ptrs, 30, None, C.byref(h)) == -22               # wrong pointer count
    assert lib.ltr_create(C.byref(desc), ptrs, 31, None, C.byref(h)) == -22               # NULL weights
    assert lib.ltr_workspace_bytes(None, _lib.LTR_WS_RANK, 8192, 0) >= 8192 * 12
    with pytest.raises(_lib.LtrError):
        _lib.check(-22, "x")


def test_product_path_never_imports_oracle():
    pkg = os.path.join(ROOT, "vllm_ltr_amd")
    for dp, _, fns in os.walk(pkg):
        for fn in fns:
            if fn.endswith(".py"):
                src = open(os.path.join(dp, fn)).read()
                assert "import oracle" not in src and "from oracle" not in src, fn


def test_scorer_refuses_to_run_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from vllm_ltr_amd.scorer import HipOPTScorer
    spec = OPTSpec.tiny_pre_ln()
    with pytest.raises(_lib.LtrError):
        HipOPTScorer(spec, seeded_checkpoint(spec, 0))


def test_predictor_config_matches_reference_parse():
    cases = json.load(open(os.path.join(GOLDEN, "config_cases.json")))["predictor_configs"]
    assert cases
    for fn, c in cases.items():
        cfg = PrefillPredictorConfig.from_dict(c["input"])
        assert dict(cfg.model.__dict__) == c["parsed"], fn


def test_predictor_config_roundtrip(tmp_path):
    cfg = PrefillPredictorConfig.from_dict({"model": {"pred_model": "facebook/opt-125m", "num_labels": 1,
                                                      "mtype": "rank", "activation": None,
                                                      "path": "/x/finetuned", "max_length": 2048,
                                                      "max_batch_size": 1000}})
    p = tmp_path / "usage_config.json"
    PrefillPredictorConfig.to_json(cfg, str(p))
    again = PrefillPredictorConfig.from_json(str(p))
    assert again == cfg
    with pytest.raises(TypeError):
        PrefillPredictorConfig.from_dict({"model": {"pred_model": "x", "num_labels": 1, "mtype": "rank",
                                                    "activation": None, "bogus": 1}})


def test_schedule_type_matches_reference_parse():
    cases = json.load(open(os.path.join(GOLDEN, "config_cases.json")))["schedule_types"]
    for st, want in cases.items():
        got = parse_schedule_type(st)
        assert got.starv == want["starv"] and got.need_score == want["need_score"]
        if got.starv != -1:
            assert got.period == want["period"]
        assert got.policy == "opt"
    assert parse_schedule_type("tpt-abc").policy == "tpt"
    assert not parse_schedule_type("fifo").need_score
    with pytest.raises(AssertionError):
        parse_schedule_type("bogus")
    with pytest.raises(ValueError):            # the reference's slicing fails the same way
        parse_schedule_type("opt-starv5")


def test_seeded_checkpoint_is_deterministic_and_fp16():
    spec = OPTSpec.tiny_post_ln(3)
    a, b = seeded_checkpoint(spec, 7), seeded_checkpoint(spec, 7)
    assert list(a) == [n for n, _ in tensor_shapes(spec)]
    for k in a:
        assert a[k].dtype == np.float16 and np.array_equal(a[k], b[k])
    assert not np.array_equal(a["score.weight"], seeded_checkpoint(spec, 8)["score.weight"])
    n125 = sum(int(np.prod(s)) for _, s in tensor_shapes(OPTSpec.opt_125m()))
    assert n125 == 125_239_296 + 768            # params (+ score head)
    n350 = sum(int(np.prod(s)) for _, s in tensor_shapes(OPTSpec.opt_350m()))
    assert n350 == 331_196_416 + 512


def test_hf_checkpoint_roundtrip(tmp_path):
    spec = OPTSpec.tiny_post_ln(3)
    ck = seeded_checkpoint(spec, 1)
    save_hf_checkpoint(str(tmp_path), spec, ck)
    spec2, ck2 = load_hf_checkpoint(str(tmp_path))
    assert spec2 == spec
    for k in ck:
        assert np.array_equal(ck[k], ck2[k])


def test_pack_layout():
    from vllm_ltr_amd.scorer import HipOPTScorer
    ids, cu = HipOPTScorer.pack([[2, 5, 6], [2], [2, 9]])
    assert ids.tolist() == [2, 5, 6, 2, 2, 9] and cu.tolist() == [0, 3, 4, 6]
    assert ids.dtype == np.int64 and cu.dtype == np.int32
    ids, cu = HipOPTScorer.pack([])
    assert ids.shape == (0,) and cu.tolist() == [0]


def test_input_stager_and_token_cache():
    """Host input pipeline: ids are produced once per request and cached; packing
    equals HipOPTScorer.pack; buffers grow and are reused."""
    from util import FakeSeqGroup
    from vllm_ltr_amd.host_pipeline import InputStager, cached_token_ids
    from vllm_ltr_amd.scorer import HipOPTScorer
    calls = []

    def tok(text):
        calls.append(text)
        return [2] + [ord(c) for c in text]
    g = FakeSeqGroup("a", [], prompt="hello world")
    a = cached_token_ids(g, tok, 6)
    b = cached_token_ids(g, tok, 6)
    assert a is b and calls == ["hello world"] and a.tolist() == [2] + [ord(c) for c in "hello"]
    with pytest.raises(ValueError):
        cached_token_ids(FakeSeqGroup("e", []), None, 10)
    st = InputStager("cpu", capacity_tokens=4, capacity_requests=1)
    r = np.random.RandomState(0)
    for n in (1, 7, 300):
        lists = [r.randint(0, 1000, r.randint(1, 40)).tolist() for _ in range(n)]
        ids_d, cu_d, cu_h = st.stage([np.asarray(x, np.int64) for x in lists])
        want_ids, want_cu = HipOPTScorer.pack(lists)
        assert ids_d.numpy().tolist() == want_ids.tolist() and cu_d.numpy().tolist() == want_cu.tolist()
        assert cu_h.tolist() == want_cu.tolist()
    ids_d, cu_d, cu_h = st.stage([])
    assert ids_d.numel() == 0 and cu_d.tolist() == [0]


HF_TINY = os.path.join(GOLDEN, "hf_tiny")


@pytest.mark.parametrize("variant,expected", [("st", "expected_class2"), ("sharded", "expected_class2"),
                                              ("bin", "expected_class2"), ("rank1", "expected_rank1"),
                                              ("fp32", "expected_rank1_fp32")])
def test_loader_reads_what_hf_writes(variant, expected):
    """Checkpoint directories written by HF itself (oracle/make_hf_fixture.py: ``.half().save_pretrained()`` single
    file / sharded + index, a transformers-4-style pytorch_model.bin with ``decoder.*`` names and a stray lm_head,
    an fp32 save) through load_hf_checkpoint; the oracle on the loaded weights reproduces HF's own logits."""
    from oracle.opt_scorer import OracleOPTScorer
    from vllm_ltr_amd.opt_spec import checkpoint_weight_dtype
    spec, ckpt = load_hf_checkpoint(os.path.join(HF_TINY, variant))
    z = np.load(os.path.join(HF_TINY, expected + ".npz"))
    assert spec.num_labels == z["logits"].shape[1]            # st/: config.json has NO num_labels (HF default 2)
    assert list(ckpt) == [n for n, _ in tensor_shapes(spec)]
    want_dtype = np.float32 if variant == "fp32" else np.float16
    assert all(v.dtype == want_dtype for v in ckpt.values())
    assert checkpoint_weight_dtype(ckpt) == ("f32" if variant == "fp32" else "f16")
    orc = OracleOPTScorer(spec, ckpt)
    h = orc.hidden(z["ids"], z["cu_seqlens"]).float()
    logits = orc.pool_head(h, __import__("torch").as_tensor(z["cu_seqlens"][1:].astype(np.int64) - 1)).numpy()
    np.testing.assert_allclose(logits, z["logits"], atol=2e-6, rtol=0)
    if variant in ("sharded", "bin"):                          # same weights as the single-file save
        _, base = load_hf_checkpoint(os.path.join(HF_TINY, "st"))
        assert all(np.array_equal(ckpt[k], base[k]) for k in base)


def test_loader_refuses_other_dtypes_and_broken_dirs(tmp_path):
    import torch
    from safetensors.torch import load_file, save_file
    src = os.path.join(HF_TINY, "rank1")
    sd = load_file(os.path.join(src, "model.safetensors"))
    d = tmp_path / "bf16"
    d.mkdir()
    (d / "config.json").write_text(open(os.path.join(src, "config.json")).read())
    save_file({k: v.to(torch.bfloat16) for k, v in sd.items()}, str(d / "model.safetensors"))
    with pytest.raises(ValueError, match="dtype"):
        load_hf_checkpoint(str(d))                              # no silent astype(float16)
    e = tmp_path / "empty"
    e.mkdir()
    (e / "config.json").write_text(open(os.path.join(src, "config.json")).read())
    with pytest.raises(FileNotFoundError):
        load_hf_checkpoint(str(e))
    m = tmp_path / "missing"
    m.mkdir()
    (m / "config.json").write_text(open(os.path.join(src, "config.json")).read())
    save_file({k: v for k, v in sd.items() if "layers.1.fc2" not in k}, str(m / "model.safetensors"))
    with pytest.raises(KeyError):
        load_hf_checkpoint(str(m))


def test_schedule_type_xpt_table_path():
    st = parse_schedule_type("xpt{/data/dist/table.pt}-whatever")           # scheduler.py:312 slicing
    assert st.policy == "xpt" and st.need_score and st.table_path == "/data/dist/table.pt"
    assert parse_schedule_type("opt-starv3-period2").table_path == ""


def test_gemm_kernels_do_not_spill():
    """The split-fp16 GEMM kernels must fit their register budget without scratch: a spilled LayerNorm-fold instance
    produced NaNs on the GPU (its inline-asm epilogue stores sat behind scratch reloads) - keep the
    compiler's own report at zero for every gemm_f16s instance."""
    import re
    import subprocess
    src = os.path.join(ROOT, "vllm_ltr_amd", "csrc", "ltr_gemm.hip")
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc",
                        "-Wno-unused-function", "-c", src, "-o", os.devnull, "-Rpass-analysis=kernel-resource-usage"],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    name, seen = None, 0
    for line in r.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
        m = re.search(r"ScratchSize \[bytes/lane\]: (\d+)", line)
        if m and name and "gemm_f16s" in name:
            seen += 1
            assert int(m.group(1)) == 0, f"{name} spills {m.group(1)} bytes per lane"
    assert seen >= 10


def test_non_null_activation_is_refused_by_name(tmp_path):
    """PrefillModelConfig.activation is applied to the rank logits by the reference's training / offline path
    (prefill_predictor.py:28-29,53,80); this path computes the raw logit, so a config that names an activation is
    refused instead of silently serving / training a different function.  Every shipped config says null."""
    from vllm_ltr_amd.config_predictor import PrefillModelConfig, PrefillPredictorConfig
    from vllm_ltr_amd.plugin import MI355XRanker
    from vllm_ltr_amd.trainer import HipPredictorTrainer, refuse_activation
    cfg = PrefillPredictorConfig(PrefillModelConfig("facebook/opt-125m", 1, "rank", "Sigmoid", path=str(tmp_path)))
    with pytest.raises(NotImplementedError, match="Sigmoid"):
        MI355XRanker.from_predictor_config(cfg, "opt-xxx-starv200-period10")
    p = tmp_path / "cfg.json"
    PrefillPredictorConfig.to_json(cfg, str(p))
    with pytest.raises(NotImplementedError, match="activation"):
        MI355XRanker.from_predictor_config(str(p), "opt")
    spec = OPTSpec.tiny_pre_ln()
    with pytest.raises(NotImplementedError, match="Tanh"):
        HipPredictorTrainer(spec, {}, activation="Tanh")
    refuse_activation(None, "x"); refuse_activation("Identity", "x")
    for fn, case in json.load(open(os.path.join(GOLDEN, "config_cases.json")))["predictor_configs"].items():
        assert case["parsed"]["activation"] is None, fn       # what the reference ships


def test_train_config_carries_the_precision():
    """The trainer's GEMM precision travels in ltr_train_config (per handle), not in the process environment; the field
    sits in what used to be padding in front of `seed` (include/ltr_hip.h)."""
    import ctypes as C
    from vllm_ltr_amd import _lib
    assert _lib.TrainConfig.precision.offset == 36 and _lib.TrainConfig.seed.offset == 40 and C.sizeof(_lib.TrainConfig) == 48
    assert _lib.TRAIN_PRECISIONS == {None: 0, "split": 1, "f32": 2}
    src = open(os.path.join(ROOT, "vllm_ltr_amd", "trainer.py")).read()
    assert "os.environ[" not in src                            # no process-wide mutation around ltr_train_create
