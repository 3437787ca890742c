"""Fine-tuning of the predictor on the device (``ltr_train_step`` of libltr_hip.so) - the host-side mirror of
the reference's training loop, ``train/trainer.py:85-216``:

    predictor = prefill_predictor_model(pred_model=..., num_labels=..., mtype=...)        # :99-101 fp32 master weights
    optimizer = torch.optim.Adam(predictor.model.parameters(), lr=args.lr, weight_decay=args.wc)     # :122
    for prompt, labels, origin_len in train_dataloader:                                            # :137
        outputs = predictor(input_ids, attention_mask)                                             # :146
        loss = loss_func(outputs.view(1, -1), labels)   # listMLE / mse; crossentropy over classes   # :150-157
        loss.backward(); optimizer.step(); optimizer.zero_grad()                                    # :161-165
    predictor.model.half().save_pretrained(finetuned_model_output_path)                              # :213-216

:class:`HipPredictorTrainer` keeps parameters, gradients and Adam moments in HBM (f32); ``step`` runs forward, loss,
backward and the Adam update as HIP kernels; ``save_pretrained`` writes the HF-format fp16 checkpoint + the
``usage_config.json`` the serving side loads (``MI355XRanker.from_predictor_config``).  Tokenisation is the caller's
(no tokenizer files offline): a slate is a list of token-id lists.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import _lib
from .config_predictor import PrefillModelConfig, PrefillPredictorConfig
from .opt_spec import OPTSpec, save_hf_checkpoint


def refuse_activation(activation, who: str) -> None:
    """``PrefillModelConfig.activation`` names a torch.nn activation the reference applies to the rank logits
    (prefill_predictor.py:28-29,53,80).  This path computes the raw ``score.weight`` logit (what opt.py:389-397 serves and
    every config under train/configs/ asks for: null); a config that names one would silently get a different function."""
    if activation not in (None, "Identity"):
        raise NotImplementedError(f"{who}: PrefillModelConfig.activation = {activation!r} is not supported - this path "
                                  "scores / trains the raw logit (activation null, as in every shipped config: "
                                  "train/configs/*.txt, trainer.py:203-216); prefill_predictor.py:28-29,80 would apply "
                                  f"torch.nn.{activation} to it")


def len2label(length: int, label_max_length: int = 8192, label_group_size: int = 1) -> int:
    """``RankingDataset.__len2label__`` (trainer.py:52-54): shorter generations get LARGER labels."""
    return label_max_length // label_group_size - min(label_max_length, length) // label_group_size


class HipPredictorTrainer:
    def __init__(self, spec: OPTSpec, ckpt: Dict[str, np.ndarray], device: str = "cuda:0", lr: float = 2e-5,
                 weight_decay: float = 0.01, betas=(0.9, 0.999), eps: float = 1e-8, loss: str = "listMLE",
                 dropout: float = 0.0, seed: int = 42, precision: Optional[str] = None, activation: Optional[str] = None):
        """Defaults are the reference's: ``--lr 2e-5 --wc 0.01`` (trainer.py:28-29), Adam's betas / eps, seed 42 (:86).
        ``dropout``: HF OPT trains with 0.1; 0 keeps the step reproducible against other implementations.
        ``precision``: how the dense layers multiply - "split": both operands as two fp16 terms on the fp16
        matrix cores (f32-grade products, f32 accumulation; the reference itself trains under fp16 autocast,
        trainer.py:147) and the scoring path's MFMA attention in the forward; "f32": the exact-f32 MFMA everywhere;
        None (default): "split" unless the process environment holds LTR_TRAIN_F32=1.  The choice
        travels in ``ltr_train_config.precision`` - per handle, no process-wide state.
        ``activation``: ``PrefillModelConfig.activation`` (prefill_predictor.py:28-29,80 applies it to the rank logits
        in the training forward).  Every shipped config says null; anything else is refused by name rather than trained
        as a different function."""
        if precision not in _lib.TRAIN_PRECISIONS:
            raise ValueError(f"precision {precision!r}: 'split', 'f32' or None")
        refuse_activation(activation, "HipPredictorTrainer")
        self.precision = precision or ("f32" if os.environ.get("LTR_TRAIN_F32", "")[:1] == "1" else "split")
        if not torch.cuda.is_available():
            raise _lib.LtrError("HipPredictorTrainer needs a ROCm GPU (no CPU fallback on the product path)")
        if loss not in _lib.LOSSES:
            raise ValueError(f"loss {loss!r}: one of {sorted(_lib.LOSSES)} (trainer.py:125-132)")
        self.lib = _lib.load()
        self.spec, self.loss = spec, loss
        self.device = torch.device(device)
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        f32 = lambda name: torch.from_numpy(np.ascontiguousarray(ckpt[name]).astype(np.float32)).to(self.device).contiguous()
        cat = lambda names: torch.cat([f32(n) for n in names], 0).contiguous()
        g: List[Optional[torch.Tensor]] = [
            f32("model.decoder.embed_tokens.weight"), f32("model.decoder.embed_positions.weight"),
            f32("model.decoder.project_in.weight") if spec.has_proj else None,
            f32("model.decoder.project_out.weight") if spec.has_proj else None,
            f32("model.decoder.final_layer_norm.weight") if spec.has_final_ln else None,
            f32("model.decoder.final_layer_norm.bias") if spec.has_final_ln else None,
            f32("score.weight")]
        for i in range(spec.num_hidden_layers):
            p = f"model.decoder.layers.{i}."
            qkv = [p + f"self_attn.{x}_proj" for x in "qkv"]
            g += [cat([n + ".weight" for n in qkv]), cat([n + ".bias" for n in qkv]),
                  f32(p + "self_attn.out_proj.weight"), f32(p + "self_attn.out_proj.bias"),
                  f32(p + "self_attn_layer_norm.weight"), f32(p + "self_attn_layer_norm.bias"),
                  f32(p + "fc1.weight"), f32(p + "fc1.bias"), f32(p + "fc2.weight"), f32(p + "fc2.bias"),
                  f32(p + "final_layer_norm.weight"), f32(p + "final_layer_norm.bias")]
        ptrs = (C.c_void_p * len(g))(*[(t.data_ptr() if t is not None else None) for t in g])
        desc = _lib.ModelDesc(spec.vocab_size, spec.hidden_size, spec.ffn_dim, spec.num_hidden_layers,
                              spec.num_attention_heads, spec.word_embed_proj_dim,
                              spec.max_position_embeddings + spec.POS_OFFSET, spec.num_labels,
                              1 if spec.do_layer_norm_before else 0, _lib.LTR_W_F32)
        cfg = _lib.TrainConfig(lr, betas[0], betas[1], eps, weight_decay, _lib.LOSSES[loss], 1e-10, -1.0, dropout,
                               _lib.TRAIN_PRECISIONS[precision], seed)
        self._h = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(self.lib.ltr_train_create(C.byref(desc), ptrs, len(g), C.byref(cfg), self._stream(), C.byref(self._h)),
                       "ltr_train_create")
        del g                                    # the library copied the weights
        self._n_weights = len(ptrs)
        self._ws: Optional[torch.Tensor] = None
        self.steps = 0

    def close(self):
        if getattr(self, "_h", None):
            self.lib.ltr_train_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _stream(self) -> int:
        return torch.cuda.current_stream(self.device).cuda_stream

    # ------------------------------------------------------------------ one step (trainer.py:137-165)
    def step(self, ids: np.ndarray, cu_seqlens: np.ndarray, labels: Sequence[float],
             shuffle: Optional[Sequence[int]] = None, apply_update: bool = True, return_logits: bool = False):
        """``ids`` int64 [T] / ``cu_seqlens`` int32 [N+1]: the slate of prompts (already truncated to max_length);
        ``labels`` [N] (neuralNDCG: every label below 128); ``shuffle``: listMLE's random permutation (listMLE.py:33; drawn here when None).
        Returns the loss (float) [and the logits [N, num_labels] before the update]."""
        cu = np.ascontiguousarray(cu_seqlens, dtype=np.int32)
        N, T = cu.shape[0] - 1, int(cu[-1])
        dev = self.device
        ids_d = torch.from_numpy(np.ascontiguousarray(ids, dtype=np.int64)).to(dev)
        cu_d = torch.from_numpy(cu).to(dev)
        lab = np.ascontiguousarray(labels, dtype=np.float32)
        if lab.size != N:
            raise ValueError(f"{lab.size} labels for a slate of {N} prompts")
        if self.loss == "crossentropy":
            # class indices: the reference asserts labels.max() < num_labels (trainer.py:151); the kernel indexes the
            # logits row with them, so a label outside [0, num_labels) must never reach it
            bad = (lab != np.floor(lab)) | (lab < 0) | (lab >= self.spec.num_labels)
            if bad.any():
                i = int(np.nonzero(bad)[0][0])
                raise ValueError(f"crossentropy label {lab[i]!r} of prompt {i} is not a class index in "
                                 f"[0, {self.spec.num_labels}) (len2label with a mismatched label_max_length / "
                                 "label_group_size produces such labels; trainer.py:50-52,151)")
        if self.loss == "neuralNDCG":
            # neuralNDCG.py:62-64 gains are 2^label - 1 in f32: from label 128 on they are inf and the reference's loss - and,
            # one Adam step later, every weight - is NaN.  ltr_neuralndcg computes exactly that; the trainer stops before it.
            if N < 2 or N > 1024:
                raise ValueError(f"neuralNDCG takes a slate of 2..1024 prompts, not {N} (one item: the reference raises "
                                 "IndexError, loss_utils.py:70)")
            if lab.max() >= 128:
                raise ValueError(f"neuralNDCG label {lab.max():.0f}: 2^label overflows f32 from 128 on and the loss is NaN, in "
                                 "the reference too (neuralNDCG.py:62-64) - bucket the lengths (--label-group-size >= 65 for "
                                 "label_max_length 8192, trainer.py:50-52)")
        lab_d = torch.from_numpy(lab).to(dev)
        sh_d = None
        if self.loss == "listMLE":
            sh = np.random.permutation(N) if shuffle is None else np.asarray(shuffle)
            sh_d = torch.from_numpy(np.ascontiguousarray(sh, dtype=np.int32)).to(dev)
        need = int(self.lib.ltr_train_workspace_bytes(self._h, N, T))
        if self._ws is None or self._ws.numel() < need:
            self._ws = None
            self._ws = torch.empty(need, dtype=torch.uint8, device=dev)
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        logits = torch.empty(N, self.spec.num_labels, dtype=torch.float32, device=dev) if return_logits else None
        _lib.check(self.lib.ltr_train_step(self._h, ids_d.data_ptr(), cu_d.data_ptr(), cu.ctypes.data, N, T, lab_d.data_ptr(),
                                           sh_d.data_ptr() if sh_d is not None else None, 1 if apply_update else 0,
                                           loss.data_ptr(), logits.data_ptr() if logits is not None else None,
                                           self._ws.data_ptr(), self._ws.numel(), self._stream()), "ltr_train_step")
        self.steps += 1
        out = float(loss.item())
        return (out, logits.cpu().numpy()) if return_logits else out

    def predict(self, ids: np.ndarray, cu_seqlens: np.ndarray) -> np.ndarray:
        """``predictor.model.eval()`` forward with the current f32 weights (trainer.py:171-190): logits [N, num_labels]."""
        cu = np.ascontiguousarray(cu_seqlens, dtype=np.int32)
        N, T = cu.shape[0] - 1, int(cu[-1])
        dev = self.device
        ids_d = torch.from_numpy(np.ascontiguousarray(ids, dtype=np.int64)).to(dev)
        cu_d = torch.from_numpy(cu).to(dev)
        need = int(self.lib.ltr_train_workspace_bytes(self._h, N, T))
        if self._ws is None or self._ws.numel() < need:
            self._ws = None
            self._ws = torch.empty(need, dtype=torch.uint8, device=dev)
        logits = torch.empty(N, self.spec.num_labels, dtype=torch.float32, device=dev)
        _lib.check(self.lib.ltr_train_step(self._h, ids_d.data_ptr(), cu_d.data_ptr(), cu.ctypes.data, N, T, None, None, -1,
                                           None, logits.data_ptr(), self._ws.data_ptr(), self._ws.numel(), self._stream()),
                   "ltr_train_step(eval)")
        return logits.cpu().numpy()

    def fit(self, train: Sequence, test: Sequence, epochs: int = 1, batch_size: int = 32, seed: int = 42,
            log=print, label_max_length: int = 8192, label_group_size: int = 1) -> List[dict]:
        """The loop of train/trainer.py:134-200.  ``train`` / ``test``: sequences of ``(token_ids, label)`` with
        ``label = len2label(output_length, ...)`` (RankingDataset, :50-66).  Per epoch: the examples in a fresh random
        order (``DataLoader(shuffle=True)``, :118) in slates of ``batch_size`` -> :meth:`step`; then the evaluation pass
        (:167-200): predictions of the test examples, Kendall's tau against their labels (``scipy.stats.kendalltau``,
        :195) and, for crossentropy, the accuracy (:198-199).  Returns one record per epoch.

        The reference evaluates against TWO labels of a test example: tau against ``RankingTestDataset``'s UNGROUPED
        label ``label_max_length - min(label_max_length, len)`` (:68-86, group size 1 whatever ``--label-group-size``
        says) and the accuracy against the TRAIN data set's grouped ``len2label(origin_len)`` (:191,198-199).  A test item
        ``(token_ids, label, origin_len)`` gets exactly that (``label`` = the ungrouped test label; the grouped one is
        rebuilt from ``origin_len`` with ``label_max_length`` / ``label_group_size``); a 2-tuple ``(token_ids, label)``
        uses the one label for both, which equals the reference only for ``label_group_size == 1``."""
        from scipy.stats import kendalltau
        from .scorer import HipOPTScorer
        rs = np.random.RandomState(seed)
        hist = []
        for epoch in range(epochs):
            order = rs.permutation(len(train))
            total, nb = 0.0, 0
            for b0 in range(0, len(order), batch_size):
                idx = order[b0:b0 + batch_size]
                ids, cu = HipOPTScorer.pack([train[i][0] for i in idx])
                total += self.step(ids, cu, np.asarray([train[i][1] for i in idx], np.float32),
                                   shuffle=rs.permutation(len(idx)) if self.loss == "listMLE" else None)
                nb += 1
            preds, truth, train_labels = [], [], []
            for b0 in range(0, len(test), batch_size):
                chunk = test[b0:b0 + batch_size]
                ids, cu = HipOPTScorer.pack([t[0] for t in chunk])
                out = self.predict(ids, cu)
                preds.extend(out.argmax(-1).tolist() if self.loss == "crossentropy" else out[:, 0].tolist())   # :186-189
                truth.extend(t[1] for t in chunk)                                                               # :190
                train_labels.extend(len2label(t[2], label_max_length, label_group_size) if len(t) > 2 else t[1]
                                    for t in chunk)                                                             # :191
            tau, pval = kendalltau(truth, preds) if len(truth) > 1 else (float("nan"), float("nan"))
            rec = dict(epoch=epoch + 1, loss=total / max(nb, 1), kendall_tau=float(tau), p_value=float(pval))
            if self.loss == "crossentropy":
                rec["acc"] = float((np.asarray(train_labels) == np.asarray(preds)).mean())                   # :198-199
            if log:
                log(f"Epoch {epoch + 1}, Loss: {rec['loss']}")                               # :165
                log(f"Kendall's Tau: {rec['kendall_tau']}, p-value: {rec['p_value']}")        # :196
            hist.append(rec)
        return hist

    def step_lists(self, token_lists: Sequence[Sequence[int]], labels, **kw):
        from .scorer import HipOPTScorer
        ids, cu = HipOPTScorer.pack(token_lists)
        return self.step(ids, cu, labels, **kw)

    # ------------------------------------------------------------------ parameters
    def _tensor(self, what: int, index: int) -> Optional[torch.Tensor]:
        cnt = C.c_size_t()
        _lib.check(self.lib.ltr_train_read(self._h, index, what, None, 0, C.byref(cnt), self._stream()), "ltr_train_read")
        if not cnt.value:
            return None
        out = torch.empty(cnt.value, dtype=torch.float32, device=self.device)
        _lib.check(self.lib.ltr_train_read(self._h, index, what, out.data_ptr(), out.numel(), C.byref(cnt), self._stream()),
                   "ltr_train_read")
        return out

    def _named(self, fn: int) -> Dict[str, np.ndarray]:
        """HF-named f32 arrays (q, k, v unstacked)."""
        s = self.spec
        H, F, De = s.hidden_size, s.ffn_dim, s.word_embed_proj_dim
        get = lambda i, shape: self._tensor(fn, i).view(*shape).cpu().numpy()
        out = {"model.decoder.embed_tokens.weight": get(0, (s.vocab_size, De)),
               "model.decoder.embed_positions.weight": get(1, (s.max_position_embeddings + s.POS_OFFSET, H))}
        if s.has_proj:
            out["model.decoder.project_in.weight"] = get(2, (H, De))
            out["model.decoder.project_out.weight"] = get(3, (De, H))
        if s.has_final_ln:
            out["model.decoder.final_layer_norm.weight"] = get(4, (H,))
            out["model.decoder.final_layer_norm.bias"] = get(5, (H,))
        out["score.weight"] = get(6, (s.num_labels, De))
        for i in range(s.num_hidden_layers):
            b = _lib.LTR_WT_GLOBAL_COUNT + i * _lib.LTR_WL_COUNT
            p = f"model.decoder.layers.{i}."
            wqkv, bqkv = get(b + 0, (3 * H, H)), get(b + 1, (3 * H,))
            for j, x in enumerate("qkv"):
                out[p + f"self_attn.{x}_proj.weight"] = wqkv[j * H:(j + 1) * H]
                out[p + f"self_attn.{x}_proj.bias"] = bqkv[j * H:(j + 1) * H]
            out[p + "self_attn.out_proj.weight"] = get(b + 2, (H, H)); out[p + "self_attn.out_proj.bias"] = get(b + 3, (H,))
            out[p + "self_attn_layer_norm.weight"] = get(b + 4, (H,)); out[p + "self_attn_layer_norm.bias"] = get(b + 5, (H,))
            out[p + "fc1.weight"] = get(b + 6, (F, H)); out[p + "fc1.bias"] = get(b + 7, (F,))
            out[p + "fc2.weight"] = get(b + 8, (H, F)); out[p + "fc2.bias"] = get(b + 9, (H,))
            out[p + "final_layer_norm.weight"] = get(b + 10, (H,)); out[p + "final_layer_norm.bias"] = get(b + 11, (H,))
        return out

    def state(self) -> Dict[str, np.ndarray]:
        """Current f32 master weights, HF names."""
        return self._named(0)

    def grads(self) -> Dict[str, np.ndarray]:
        """Gradients of the last step, HF names."""
        return self._named(1)

    # ------------------------------------------------------------------ trainer.py:203-216
    def save_pretrained(self, output_dir: str, config: Optional[PrefillPredictorConfig] = None) -> str:
        """Writes ``<output_dir>/finetuned`` (HF directory, weights ``.half()``, trainer.py:213-216) and
        ``<output_dir>/usage_config.json`` with ``model.path`` pointing at it (:203-211); returns the config path."""
        from .opt_spec import tensor_shapes
        fin = os.path.join(output_dir, "finetuned")
        st = self.state()
        save_hf_checkpoint(fin, self.spec, {k: st[k].astype(np.float16) for k, _ in tensor_shapes(self.spec)})
        if config is None:
            config = PrefillPredictorConfig(PrefillModelConfig(
                pred_model="facebook/opt", num_labels=self.spec.num_labels,
                mtype="rank" if self.spec.num_labels == 1 else "class", activation=None))
        config.model.path = str(fin)
        config.model.num_labels = self.spec.num_labels
        path = os.path.join(output_dir, "usage_config.json")
        PrefillPredictorConfig.to_json(config, path)
        return path


def attention_forward_backward(qkv: torch.Tensor, dout: torch.Tensor, cu_seqlens: torch.Tensor, num_heads: int):
    """The attention block of the training step alone (``ltr_train_attention``): ``qkv`` f32 [T, 3H], ``dout`` f32 [T, H] and
    ``cu_seqlens`` int32 [N + 1] on one ROCm device -> ``(out [T, H], dqkv [T, 3H])`` f32.  The kernels ``step()`` runs:
    split-fp16 MFMA forward with saved log-sum-exp rows, split-fp16 MFMA backward (what autograd does for OPTAttention
    under train/trainer.py:147-159)."""
    if not qkv.is_cuda:
        raise _lib.LtrError("attention_forward_backward needs a ROCm GPU (no CPU fallback on the product path)")
    lib = _lib.load()
    T, N = int(qkv.shape[0]), int(cu_seqlens.numel()) - 1
    H = int(qkv.shape[1]) // 3
    assert qkv.dtype == torch.float32 and dout.dtype == torch.float32 and cu_seqlens.dtype == torch.int32
    assert qkv.is_contiguous() and dout.is_contiguous() and dout.shape == (T, H) and H == 64 * num_heads
    out = torch.empty((T, H), dtype=torch.float32, device=qkv.device)
    dqkv = torch.empty((T, 3 * H), dtype=torch.float32, device=qkv.device)
    ws = torch.empty(int(lib.ltr_train_attention_workspace_bytes(num_heads, N, T)), dtype=torch.uint8, device=qkv.device)
    with torch.cuda.device(qkv.device):
        st = torch.cuda.current_stream().cuda_stream
        _lib.check(lib.ltr_train_attention(num_heads, qkv.data_ptr(), dout.data_ptr(), cu_seqlens.data_ptr(), N, T,
                                           out.data_ptr(), dqkv.data_ptr(), ws.data_ptr(), ws.numel(), st), "ltr_train_attention")
    return out, dqkv
