"""Device-side predictor: HF ``OPTForSequenceClassification`` weights resident in HBM,
scored through ``ltr_score`` (libltr_hip.so).

Mirrors what the reference's AUX engine does with the model
(vllm/engine/aux_llm_engine.py:332-412 -> vllm/worker/model_runner.py:827-877 ->
vllm/model_executor/models/opt.py:362-444) minus the second engine: one call scores a
flat varlen batch of token ids.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from .opt_spec import OPTSpec


class HipOPTScorer:
    """Owns the weight tensors (torch, device memory) and the ``ltr_handle``.

    weight_dtype: "f16" (production: fp16 weights x hi+lo split activations, f32
    accumulate), "f32" (exact f32 MFMA path) or "f16-1pass" (opt-in: ONE fp16 MFMA pass per
    product - the arithmetic of the reference's own fp16 GPU predictor, ~2e-3 from the fp32
    scores, outside the 1e-4 contract of the default; `LTR_F_ONE_PASS`).
    ln_fold=False: separate LayerNorm launches instead of the GEMM-epilogue fold
    (`LTR_F_NO_LN_FOLD`: the handle a caller falls back to on LTR_E_RANGE).
    """

    def __init__(self, spec: OPTSpec, ckpt: Dict[str, np.ndarray], device: str = "cuda:0",
                 weight_dtype: str = "f16", chunk_tokens: int = 0, ln_fold: bool = True):
        if not torch.cuda.is_available():
            raise _lib.LtrError("HipOPTScorer needs a ROCm GPU (no CPU fallback on the product path)")
        self.lib = _lib.load()
        self.spec = spec
        self.device = torch.device(device)
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        if weight_dtype not in ("f16", "f32", "f16-1pass"):
            raise _lib.LtrError(f"weight_dtype {weight_dtype!r}: expected 'f16', 'f32' or 'f16-1pass'")
        self.weight_dtype = weight_dtype
        self.one_pass = weight_dtype == "f16-1pass"
        self.ln_fold = bool(ln_fold)
        self._ckpt_ref = ckpt              # (a twin handle without the fold is built from the same arrays: unfolded_twin)
        weight_dtype = "f16" if self.one_pass else weight_dtype
        wt = torch.float16 if weight_dtype == "f16" else torch.float32
        self._tensors: List[Optional[torch.Tensor]] = []
        for name, arr in ckpt.items():
            if np.asarray(arr).dtype not in (np.float16, np.float32):
                raise _lib.LtrError(f"checkpoint tensor {name} has dtype {np.asarray(arr).dtype}: only fp16 / fp32 "
                                    "checkpoints are supported (trainer.py:213-216 saves .half())")
            if weight_dtype == "f16" and np.asarray(arr).dtype != np.float16 and name.endswith("weight") \
                    and "layer_norm" not in name:
                raise _lib.LtrError(f"{name} is fp32 but weight_dtype='f16' would round it: pass weight_dtype='f32' "
                                    "or a .half() checkpoint (the f16 path assumes the weights are EXACT in fp16)")

        def mat(name):        # matrices / tables in the weight dtype
            return torch.from_numpy(np.ascontiguousarray(ckpt[name])).to(self.device, wt).contiguous()

        def vec(name):        # biases / LN affine always f32
            return torch.from_numpy(np.ascontiguousarray(ckpt[name]).astype(np.float32)).to(self.device).contiguous()

        def cat(names, f):
            return torch.cat([f(n) for n in names], 0).contiguous()

        g: List[Optional[torch.Tensor]] = [
            mat("model.decoder.embed_tokens.weight"),
            mat("model.decoder.embed_positions.weight"),
            mat("model.decoder.project_in.weight") if spec.has_proj else None,
            mat("model.decoder.project_out.weight") if spec.has_proj else None,
            vec("model.decoder.final_layer_norm.weight") if spec.has_final_ln else None,
            vec("model.decoder.final_layer_norm.bias") if spec.has_final_ln else None,
            mat("score.weight"),
        ]
        for i in range(spec.num_hidden_layers):
            p = f"model.decoder.layers.{i}."
            qkv = [p + f"self_attn.{x}_proj" for x in "qkv"]       # stacked q,k,v (opt.py:411-417)
            g += [cat([n + ".weight" for n in qkv], mat), cat([n + ".bias" for n in qkv], vec),
                  mat(p + "self_attn.out_proj.weight"), vec(p + "self_attn.out_proj.bias"),
                  vec(p + "self_attn_layer_norm.weight"), vec(p + "self_attn_layer_norm.bias"),
                  mat(p + "fc1.weight"), vec(p + "fc1.bias"), mat(p + "fc2.weight"), vec(p + "fc2.bias"),
                  vec(p + "final_layer_norm.weight"), vec(p + "final_layer_norm.bias")]
        self._tensors = g
        ptrs = (C.c_void_p * len(g))(*[(t.data_ptr() if t is not None else None) for t in g])
        desc = _lib.ModelDesc(spec.vocab_size, spec.hidden_size, spec.ffn_dim, spec.num_hidden_layers,
                              spec.num_attention_heads, spec.word_embed_proj_dim,
                              spec.max_position_embeddings + spec.POS_OFFSET, spec.num_labels,
                              1 if spec.do_layer_norm_before else 0,
                              _lib.LTR_W_F16 if weight_dtype == "f16" else _lib.LTR_W_F32,
                              (0 if ln_fold else _lib.LTR_F_NO_LN_FOLD) |
                              (_lib.LTR_F_ONE_PASS if self.one_pass else 0))
        self._h = C.c_void_p()
        # the library packs the dense-layer weights on THIS stream (ordered after the uploads above, which
        # torch issued on the same stream) and makes self.device current for its own launches
        with torch.cuda.device(self.device):
            _lib.check(self.lib.ltr_create(C.byref(desc), ptrs, len(g), self._stream(), C.byref(self._h)), "ltr_create")
        self.chunk_tokens = 0
        if chunk_tokens:
            self.set_chunk_tokens(chunk_tokens)
        self._ws: Optional[torch.Tensor] = None
        self._ws_by_key: dict = {}

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self.lib.ltr_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_chunk_tokens(self, n: int):
        _lib.check(self.lib.ltr_set_chunk_tokens(self._h, int(n)), "ltr_set_chunk_tokens")
        self.chunk_tokens = int(n)
        self._ws = None

    def unfolded_twin(self) -> "HipOPTScorer":
        """A scorer of the same checkpoint whose GEMMs are fed by separate LayerNorm launches (`LTR_F_NO_LN_FOLD`):
        what `MI355XRanker` re-scores a batch on when this handle reports LTR_E_RANGE.  Shares the weight tensors
        (the library keeps pointers; only its packed GEMM images are per handle)."""
        twin = HipOPTScorer.__new__(HipOPTScorer)
        twin.lib, twin.spec, twin.device = self.lib, self.spec, self.device
        twin.weight_dtype, twin.one_pass, twin.ln_fold = self.weight_dtype, self.one_pass, False
        twin._ckpt_ref = self._ckpt_ref
        twin._tensors = self._tensors
        g = self._tensors
        ptrs = (C.c_void_p * len(g))(*[(t.data_ptr() if t is not None else None) for t in g])
        spec = self.spec
        desc = _lib.ModelDesc(spec.vocab_size, spec.hidden_size, spec.ffn_dim, spec.num_hidden_layers,
                              spec.num_attention_heads, spec.word_embed_proj_dim,
                              spec.max_position_embeddings + spec.POS_OFFSET, spec.num_labels,
                              1 if spec.do_layer_norm_before else 0,
                              _lib.LTR_W_F32 if self.weight_dtype == "f32" else _lib.LTR_W_F16,
                              _lib.LTR_F_NO_LN_FOLD | (_lib.LTR_F_ONE_PASS if self.one_pass else 0))
        twin._h = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(self.lib.ltr_create(C.byref(desc), ptrs, len(g), self._stream(), C.byref(twin._h)), "ltr_create")
        twin._ws, twin._ws_by_key = None, {}
        twin.chunk_tokens = 0
        if getattr(self, "chunk_tokens", 0):              # (a handle configured with set_chunk_tokens keeps its pass size)
            twin.set_chunk_tokens(self.chunk_tokens)
        return twin

    def release_workspaces(self, keys=()) -> None:
        """Drop the default-stream scoring scratch and the named side-stream scratches (re-grown on demand): what a ranker calls
        on the handle it has just replaced by its unfolded twin, so that the scratch is not held twice while somebody else still
        references the old handle.  (Other users' keyed scratches stay: they may be in flight on their own streams.)"""
        self._ws = None
        for k in keys:
            self._ws_by_key.pop(k, None)

    def profile(self, on: bool) -> None:
        _lib.check(self.lib.ltr_profile_enable(self._h, 1 if on else 0), "ltr_profile_enable")

    def profile_read(self, reset: bool = True) -> dict:
        """Per-kernel-class {ms, work, launches} since the last reset (synchronises)."""
        st = _lib.ProfileStats()
        _lib.check(self.lib.ltr_profile_read(self._h, C.byref(st), 1 if reset else 0), "ltr_profile_read")
        return {k: dict(ms=st.ms[i], work=st.work[i], launches=int(st.launches[i]))
                for i, k in enumerate(_lib.PROFILE_KINDS)}

    # ------------------------------------------------------------------ helpers
    def _workspace(self, N: int, T: int, key: Optional[str] = None) -> torch.Tensor:
        """The scoring workspace (grown on demand).  ``key``: a second, independent workspace for calls that run on ANOTHER
        stream at the same time as the default one's (the handle allows concurrent calls; they must not share scratch)."""
        need = int(self.lib.ltr_workspace_bytes(self._h, _lib.LTR_WS_SCORE, N, T))
        if key is not None:
            ws = self._ws_by_key.get(key)
            if ws is None or ws.numel() < need:
                self._ws_by_key[key] = None
                ws = self._ws_by_key[key] = torch.empty(need, dtype=torch.uint8, device=self.device)
            return ws
        if self._ws is None or self._ws.numel() < need:
            self._ws = None
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._ws

    def _stream(self) -> int:
        return torch.cuda.current_stream(self.device).cuda_stream

    def check_status(self) -> None:
        """Synchronises the current stream and raises if a forward since the last check met a token id outside
        the vocabulary (where the reference's F.embedding raises).  Call where scores are read."""
        _lib.check(self.lib.ltr_status(self._h, self._stream()), "ltr_status")

    @staticmethod
    def pack(token_lists: Sequence[Sequence[int]]) -> Tuple[np.ndarray, np.ndarray]:
        """Flat ids + cu_seqlens, the layout ModelRunner._prepare_prompt builds
        (model_runner.py:383-395, 711-716)."""
        lens = np.fromiter((len(t) for t in token_lists), np.int64, len(token_lists))
        cu = np.zeros(len(token_lists) + 1, np.int32)
        np.cumsum(lens, out=cu[1:])
        ids = np.empty(int(cu[-1]), np.int64)
        for i, t in enumerate(token_lists):
            ids[cu[i]:cu[i + 1]] = t
        return ids, cu

    # ------------------------------------------------------------------ calls
    def score_device(self, ids_dev: torch.Tensor, cu_dev: torch.Tensor, cu_host: np.ndarray,
                     logits_out: Optional[torch.Tensor] = None,
                     out: Optional[torch.Tensor] = None, workspace_key: Optional[str] = None) -> torch.Tensor:
        """Inputs already resident in HBM (int64 [T], int32 [N+1]); returns f32 [N] on
        the device.  ``cu_host`` is the host mirror of ``cu_dev``.  Asynchronous, on torch's current stream;
        ``workspace_key``: see :meth:`_workspace` (calls in flight on two streams)."""
        N = int(cu_host.shape[0]) - 1
        T = int(cu_host[-1]) if N >= 0 else 0
        if out is None:
            out = torch.empty(max(N, 0), dtype=torch.float32, device=self.device)
        if N <= 0:
            return out
        assert ids_dev.dtype == torch.int64 and cu_dev.dtype == torch.int32
        cu_host = np.ascontiguousarray(cu_host, dtype=np.int32)
        ws = self._workspace(N, T, workspace_key)
        max_len = int(np.diff(cu_host).max())
        # (the library makes the handle's device current itself; the torch context keeps the workspace
        # allocation and the stream lookup on the same device)
        _lib.check(self.lib.ltr_score(self._h, ids_dev.data_ptr(), cu_dev.data_ptr(), cu_host.ctypes.data, N, T,
                                      max_len, out.data_ptr(),
                                      logits_out.data_ptr() if logits_out is not None else None,
                                      ws.data_ptr(), ws.numel(), self._stream()), "ltr_score")
        return out

    def score(self, ids: np.ndarray, cu_seqlens: np.ndarray, return_logits: bool = False):
        """Host arrays in, host f32 scores out (synchronises)."""
        cu = np.ascontiguousarray(cu_seqlens, dtype=np.int32)
        N = cu.shape[0] - 1
        if N <= 0:
            z = np.zeros(0, np.float32)
            return (z, np.zeros((0, self.spec.num_labels), np.float32)) if return_logits else z
        ids_dev = torch.from_numpy(np.ascontiguousarray(ids, dtype=np.int64)).to(self.device)
        cu_dev = torch.from_numpy(cu).to(self.device)
        logits = torch.empty(N, self.spec.num_labels, dtype=torch.float32, device=self.device) \
            if return_logits else None
        s = self.score_device(ids_dev, cu_dev, cu, logits)
        self.check_status()
        s = s.cpu().numpy()
        return (s, logits.cpu().numpy()) if return_logits else s

    def score_lists(self, token_lists: Sequence[Sequence[int]]) -> np.ndarray:
        ids, cu = self.pack(token_lists)
        return self.score(ids, cu)

    def hidden(self, ids: np.ndarray, cu_seqlens: np.ndarray, n_layers: int = -1) -> np.ndarray:
        """f32 hidden states [T, H] after ``n_layers`` decoder layers (test hook)."""
        cu = np.ascontiguousarray(cu_seqlens, dtype=np.int32)
        N, T = cu.shape[0] - 1, int(cu[-1])
        ids_dev = torch.from_numpy(np.ascontiguousarray(ids, dtype=np.int64)).to(self.device)
        cu_dev = torch.from_numpy(cu).to(self.device)
        out = torch.empty(T, self.spec.hidden_size, dtype=torch.float32, device=self.device)
        ws = self._workspace(N, T)
        _lib.check(self.lib.ltr_forward_hidden(self._h, ids_dev.data_ptr(), cu_dev.data_ptr(), cu.ctypes.data, N, T,
                                               int(np.diff(cu).max()), n_layers, out.data_ptr(), ws.data_ptr(),
                                               ws.numel(), self._stream()), "ltr_forward_hidden")
        return out.cpu().numpy()

    def embed_gather_device(self, ids_dev: torch.Tensor, cu_dev: torch.Tensor, N: int, T: int,
                            hidden_out: torch.Tensor, tok_out: Optional[torch.Tensor] = None) -> None:
        _lib.check(self.lib.ltr_embed_gather(self._h, ids_dev.data_ptr(), cu_dev.data_ptr(), N, T,
                                             hidden_out.data_ptr(),
                                             tok_out.data_ptr() if tok_out is not None else None,
                                             self._stream()), "ltr_embed_gather")

    def attention_device(self, qkv: torch.Tensor, cu_dev: torch.Tensor, N: int, T: int, out: torch.Tensor,
                         split_kv: bool = True) -> None:
        """The varlen causal attention kernel alone (``ltr_attention``).  f16 mode: ``qkv`` fp16 [2, T, 3H] (hi | lo planes),
        ``out`` fp16 [2, T, H]; f32 mode: f32 [T, 3H] -> f32 [T, H].  ``split_kv``: hand over the larger work-list scratch
        (32-query blocks) with which passes of <= 600 tokens run the split-K/V variant, as they do inside ``ltr_score``;
        False: the minimal scratch of include/ltr_hip.h - always the 128-query kernel."""
        need = (N + 4) * 4 + (T // (32 if split_kv else 64) + N + 1) * 16
        ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        _lib.check(self.lib.ltr_attention(self._h, qkv.data_ptr(), cu_dev.data_ptr(), N, T, out.data_ptr(), ws.data_ptr(),
                                          ws.numel(), self._stream()), "ltr_attention")

    def pool_head_device(self, hidden: torch.Tensor, cu_dev: torch.Tensor, N: int, scores_out: torch.Tensor,
                         logits_out: Optional[torch.Tensor] = None) -> None:
        _lib.check(self.lib.ltr_pool_head(self._h, hidden.data_ptr(), cu_dev.data_ptr(), N, scores_out.data_ptr(),
                                          logits_out.data_ptr() if logits_out is not None else None,
                                          self._stream()), "ltr_pool_head")
