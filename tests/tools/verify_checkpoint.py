"""One command for the parity check that cannot be made offline: the reference's REAL predictors (README.md:26 `LLM-ltr/OPT-Predictors`,
an HF `OPTForSequenceClassification` directory) through the HIP path against the CPU oracle.

    python tests/tools/verify_checkpoint.py <hf_dir> [--tokenizer <dir>] [--prompts file.txt] [-n 256] [--max-length 2048]

Loads the directory with `opt_spec.load_hf_checkpoint` (config.json + safetensors / sharded / .bin; opt.py:411-444), takes N prompts
- lines of --prompts tokenised with the OPT tokenizer and truncated the way aux_llm_engine.py:365-369 does, or synthetic token ids
with ShareGPT-like lengths when no tokenizer is given -, scores them with the oracle (oracle/opt_scorer.py, fp32 on the host cores)
and with the HIP path exactly as the plug-in runs it (`MI355XRanker.obtain_aux_scores`: LayerNorm fold, LTR_E_RANGE fallback to the
unfolded twin), and prints max|d score|, the discordant pairs of the two orders with their score gaps, `range_fallbacks`.
Exit code 0 when max|d| <= 1e-4 (north_star's tolerance) and every discordant pair is a near-tie (gap <= 2 x max|d|).

This is TEST infrastructure (it lives under tests/ because it runs the oracle); nothing in the product imports it.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("hf_dir")
    ap.add_argument("--tokenizer", default=None, help="tokenizer directory (AutoTokenizer); without it: synthetic token ids")
    ap.add_argument("--prompts", default=None, help="text file, one prompt per line (needs --tokenizer)")
    ap.add_argument("-n", type=int, default=256)
    ap.add_argument("--max-length", type=int, default=0, help="truncate prompts to this many tokens (0: the model's positions)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--device", default="cuda:0")
    ap.add_argument("--tol", type=float, default=1e-4)
    args = ap.parse_args(argv)

    import torch
    from oracle.opt_scorer import OracleOPTScorer
    from util import FakeSeqGroup, bench_lengths
    from vllm_ltr_amd.opt_spec import checkpoint_weight_dtype, load_hf_checkpoint
    from vllm_ltr_amd.plugin import MI355XRanker
    from vllm_ltr_amd.scorer import HipOPTScorer

    spec, ckpt = load_hf_checkpoint(args.hf_dir)
    mode = checkpoint_weight_dtype(ckpt)
    max_pos = spec.max_position_embeddings
    max_len = min(args.max_length, max_pos) if args.max_length > 0 else max_pos
    if args.tokenizer:
        from transformers import AutoTokenizer
        tok = AutoTokenizer.from_pretrained(args.tokenizer)
        if args.prompts:
            texts = [ln.rstrip("\n") for ln in open(args.prompts) if ln.strip()][:args.n]
        else:
            r = np.random.RandomState(args.seed)
            words = ["explain", "the", "schedule", "of", "a", "ranking", "model", "please", "write", "code", "for", "sorting", "why", "is", "sky"]
            texts = [" ".join(r.choice(words, int(k))) for k in bench_lengths(args.n, seed=args.seed, mu=48.0)]
        toks = [tok(t, truncation=True, max_length=max_len)["input_ids"] for t in texts]            # aux_llm_engine.py:365-369
        source = f"{len(toks)} prompts tokenised with {args.tokenizer}"
    else:
        r = np.random.RandomState(args.seed)
        lens = np.clip(bench_lengths(args.n, seed=args.seed), 1, max_len)
        toks = [r.randint(4, spec.vocab_size, int(k)).tolist() for k in lens]
        source = f"{len(toks)} synthetic prompts (random token ids, ShareGPT-like lengths; no tokenizer given)"
    toks = [t for t in toks if len(t) > 0]
    n = len(toks)
    lens = np.array([len(t) for t in toks])
    cu = np.zeros(n + 1, np.int32)
    np.cumsum(lens, out=cu[1:])
    ids = np.concatenate([np.asarray(t, np.int64) for t in toks])

    t0 = time.time()
    want = OracleOPTScorer(spec, ckpt).score(ids, cu).astype(np.float64)
    t_or = time.time() - t0

    sc = HipOPTScorer(spec, ckpt, args.device, mode)
    ranker = MI355XRanker(sc, "opt-xxx", max_length=max_len, mtype="rank" if spec.num_labels == 1 else "class")
    groups = [FakeSeqGroup(str(i), t) for i, t in enumerate(toks)]
    t0 = time.time()
    got = np.asarray(ranker.obtain_aux_scores(groups), np.float64)
    torch.cuda.synchronize()
    t_hip = time.time() - t0
    fallbacks = ranker.metrics()["range_fallbacks"]

    err = np.abs(got - want)
    if spec.num_labels > 1:
        # class mode: the score is the predicted label (opt.py:394-395); the numeric comparison is on the logits, and a label may
        # differ only where the oracle's two best logits are closer than the logit error
        lg_o = OracleOPTScorer(spec, ckpt).logits(ids, cu).numpy().astype(np.float64)
        _, lg_h = sc.score(ids, cu, return_logits=True)
        lerr = np.abs(lg_h.astype(np.float64) - lg_o)
        top2 = np.sort(lg_o[:, :min(spec.num_labels, spec.vocab_size)], axis=1)[:, -2:]
        flips = np.nonzero(got != want)[0]
        bad_flips = [int(i) for i in flips if top2[i, 1] - top2[i, 0] > 2 * lerr.max()]
        err = np.where(got != want, 0.0, 0.0) + lerr.max(axis=1)
        class_info = dict(label_flips=int(len(flips)), label_flips_not_near_ties=bad_flips, max_logit_err=float(lerr.max()))
    else:
        class_info = None
    worst = int(err.argmax())
    # rank mode: the scheduler orders by score (descending where the policy says so): count the pairs the two score vectors order differently
    oo, og = np.argsort(-want, kind="stable"), np.argsort(-got, kind="stable")
    pos_g = np.empty(n, np.int64)
    pos_g[og] = np.arange(n)
    disc, worst_gap = 0, 0.0
    for a in range(n):
        ia = oo[a]
        later = oo[a + 1:]
        bad = later[pos_g[later] < pos_g[ia]]
        disc += len(bad)
        if len(bad):
            worst_gap = max(worst_gap, float(np.abs(want[ia] - want[bad]).max()))
    rec = dict(checkpoint=os.path.abspath(args.hf_dir), layers=spec.num_hidden_layers, hidden=spec.hidden_size, labels=spec.num_labels,
               pre_ln=bool(spec.do_layer_norm_before), weights=mode, prompts=source, tokens=int(cu[-1]),
               max_abs_err=float(err.max()), worst_request=worst, mean_abs_err=float(err.mean()), score_scale=float(np.abs(want).max()),
               discordant_pairs=int(disc), pairs=n * (n - 1) // 2, largest_oracle_gap_of_a_discordant_pair=worst_gap,
               range_fallbacks=int(fallbacks), oracle_seconds=round(t_or, 2), hip_seconds=round(t_hip, 3), tolerance=args.tol)
    if class_info:
        rec.update(class_info)
        ok = class_info["max_logit_err"] <= args.tol and not class_info["label_flips_not_near_ties"]
    else:
        ok = rec["max_abs_err"] <= args.tol and worst_gap <= 2 * max(rec["max_abs_err"], 1e-12) + 1e-12
    rec["verdict"] = "PASS" if ok else "FAIL"
    print(json.dumps(rec, indent=1))
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
