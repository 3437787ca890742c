"""diag: do the two lanes of a mid-sized scoring call (ltr_api.hip run_forward) run side by side in a process that looks like
a serving engine - dozens of handles and streams alive - and not only in bench.py's?  (k = 64 took 3.93 ms
inside the driver's pytest process, 2.91 in its bench, 2.97 on ONE lane.)

    python tests/diag/lanes_busy_process.py

Handles are created one after the other - what a test suite or an engine with several predictors does - first in a fresh
process, then after 32 more handles (each owns a lane stream) and 32 torch streams that have run work.  "unprobed" = the
round-4 behaviour (the first candidate stream is kept, `LTR_F_LANES_UNPROBED`), "probed" = the default (ltr_create keeps a
candidate only if a 150-us spin on it overlaps one on the caller's stream).  Per handle: `ltr_lane_probe` (pair ~ solo:
concurrent, ~ 2 x solo: the lane stream shares the caller's hardware queue) and the k = 16 / 64 steady call (+ re-rank of the
8k queue) against a one-lane handle, alternately."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from util import bench_lengths, synthetic_batch  # noqa: E402
from vllm_ltr_amd.opt_spec import OPTSpec, seeded_checkpoint  # noqa: E402
from vllm_ltr_amd.rank import DeviceQueue  # noqa: E402
from vllm_ltr_amd.scorer import HipOPTScorer  # noqa: E402

dev = torch.device("cuda:0")
spec = OPTSpec.opt_125m()
ckpt = seeded_checkpoint(spec, 0)
n = 8192
queue = DeviceQueue(dev, starv=200, period=10, capacity=n)
queue.append(torch.randn(n))
need = torch.full((n,), 64, dtype=torch.int32, device=dev)
ones = torch.ones(n, dtype=torch.int32, device=dev)
print(f"GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES', '(unset: runtime default 4)')}  "
      f"LTR_LANE_PROBE={os.environ.get('LTR_LANE_PROBE', '(unset: probe on)')}")


def timed(sc, k, reps=21):
    lens = bench_lengths(k, seed=0)
    ids, cu = synthetic_batch(spec, lens.tolist(), 1)
    ids_d, cu_d = torch.from_numpy(ids).to(dev), torch.from_numpy(cu).to(dev)
    for _ in range(3):
        sc.score_device(ids_d, cu_d, cu, out=queue._score[:k])
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record()
        sc.score_device(ids_d, cu_d, cu, out=queue._score[:k])
        queue.step(need, ones, 2048, 256)
        b.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in ev)
    return ms[len(ms) // 2]


def stage(name, lanes=True):
    two = HipOPTScorer(spec, ckpt, "cuda:0", "f16", lanes=lanes)
    tries, solo, pair = two.lane_probe()
    state = "no lane stream" if not tries else ("concurrent" if pair < 1.6 * solo else "ALIASED: serial")
    line = f"[{name}] lane stream: {tries} candidate(s) tried; spin solo {solo:.0f} us, pair {pair:.0f} us ({state})"
    for k in (16, 64):
        a = [timed(one, k), timed(two, k), timed(one, k), timed(two, k)]
        before = two.lane_calls()
        timed(two, k, reps=3)
        line += (f"; k = {k}: one lane {min(a[0], a[2]):.3f} ms, this handle {min(a[1], a[3]):.3f} ms "
                 f"({(min(a[1], a[3]) / min(a[0], a[2]) - 1) * 100:+.1f} %, two lanes used: {two.lane_calls() > before})")
    print(line, flush=True)
    return two


one = HipOPTScorer(spec, ckpt, "cuda:0", "f16", lanes=False)
keep = []
# Consecutive handles: the runtime maps each new stream to one of its GPU_MAX_HW_QUEUES hardware queues in turn, so one
# UNPROBED lane stream in every few lands on the queue of the caller's stream; a probed handle never keeps such a stream.
for i in range(6):
    keep.append(stage(f"unprobed handle {i}", lanes="unprobed"))
for i in range(6):
    keep.append(stage(f"probed handle {i}"))
tiny = OPTSpec.tiny_pre_ln()
tck = seeded_checkpoint(tiny, 1)
for i in range(32):
    keep.append(HipOPTScorer(tiny, tck, "cuda:0", "f16"))
streams = [torch.cuda.Stream() for _ in range(32)]
x = torch.zeros(1024, device=dev)
for st in streams:
    with torch.cuda.stream(st):
        x.add_(1.0)
torch.cuda.synchronize()
print("--- after 32 more handles and 32 torch streams that have run work", flush=True)
for i in range(4):
    keep.append(stage(f"busy process, unprobed handle {i}", lanes="unprobed"))
for i in range(4):
    keep.append(stage(f"busy process, probed handle {i}"))
