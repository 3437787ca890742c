"""Shared helpers for the test-suite."""
import os

import numpy as np

from vllm_ltr_amd.opt_spec import OPTSpec

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def spec_from_npz(z) -> OPTSpec:
    conv = {}
    for k, v in z["spec"]:
        conv[str(k)] = (v == "True") if v in ("True", "False") else int(v)
    return OPTSpec(**conv)


def synthetic_batch(spec: OPTSpec, lens, seed):
    rs = np.random.RandomState(seed)
    ids = []
    for L in lens:
        row = rs.randint(4, spec.vocab_size, size=L)
        row[0] = 2
        ids.append(row)
    ids = np.concatenate(ids).astype(np.int64) if len(lens) else np.zeros(0, np.int64)
    cu = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    return ids, cu


def bench_lengths(n, seed=0, mu=64.0):
    """BASELINE.md section 4 / SURVEY.md 8d length profile."""
    rs = np.random.RandomState(seed)
    return np.clip(np.rint(np.exp(rs.normal(np.log(mu), 0.8, n))), 4, 1024).astype(np.int64)


def discordant_pairs(order_ref, order_got, score_of):
    """Pairs of requests the two orders rank differently (Kendall distance), as (a, b, |score_of[a] - score_of[b]|)."""
    pos = {int(r): i for i, r in enumerate(order_got)}
    ref = [int(r) for r in order_ref]
    assert sorted(pos) == sorted(ref)
    out = []
    for i in range(len(ref)):
        pi = pos[ref[i]]
        for j in range(i + 1, len(ref)):
            if pos[ref[j]] < pi:
                out.append((ref[i], ref[j], abs(float(score_of[ref[i]]) - float(score_of[ref[j]]))))
    return out


class FakeSeqGroup:
    """The SequenceGroup fields the ranking path touches (vllm/sequence.py:426-465,
    vllm/core/scheduler.py:372-374)."""

    def __init__(self, request_id, prompt_token_ids, prompt=None):
        self.request_id = request_id
        self.prompt_token_ids = list(prompt_token_ids)
        self.prompt = prompt
        self.aux_model_score = None
        self.pri = 0
        self.idle = 0
        self.runs = 0

    def need_aux_model_score(self):
        return self.aux_model_score is None

    def set_aux_model_score(self, s):
        self.aux_model_score = s
