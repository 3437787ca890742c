"""diag: do the two lanes of a mid-sized scoring call (ltr_api.hip run_forward) run side by side in a process that looks like
a serving engine - dozens of handles and streams alive - and not only in bench.py's?  (k = 64 took 3.93 ms
inside the driver's pytest process, 2.91 in its bench, 2.97 on ONE lane.)

    [LTR_LANE_PROBE=0] python tests/diag/lanes_busy_process.py

Stage A: a fresh process.  Stage B: after 32 more handles (each owns a lane stream) and 32 torch streams that have run work.
In each stage a NEW pair of scorers (lanes on / off) is created - what test #185 did - and timed alternately at k = 16 / 64
arrivals (+ re-rank of the 8k queue); `ltr_lane_probe` reports how many candidate streams ltr_create went through and the
fork / join spin measurement (pair ~ solo: concurrent, ~ 2 x solo: the lane stream shares the caller's hardware queue).
LTR_LANE_PROBE=0 = the round-4 behaviour (first candidate kept unprobed)."""
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


def stage(name):
    two = HipOPTScorer(spec, ckpt, "cuda:0", "f16")
    one = HipOPTScorer(spec, ckpt, "cuda:0", "f16", lanes=False)
    tries, solo, pair = two.lane_probe()
    print(f"[{name}] lane stream: {tries} candidate(s) tried; spin solo {solo:.1f} us, pair {pair:.1f} us "
          f"({'concurrent' if tries and pair < 1.6 * solo else ('no lane stream' if not tries else 'ALIASED: serial')})")
    for k in (16, 64, 256):
        a = [timed(one, k), timed(two, k), timed(one, k), timed(two, k)]
        before = two.lane_calls()
        timed(two, k, reps=3)
        used = two.lane_calls() > before
        print(f"[{name}] k = {k:3d}: one lane {a[0]:.3f} / {a[2]:.3f} ms, lanes handle {a[1]:.3f} / {a[3]:.3f} ms "
              f"({(min(a[1], a[3]) / min(a[0], a[2]) - 1) * 100:+.1f} %; two lanes in use: {used})")
    return two, one


keep = [stage("A fresh process")]
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
keep.append(stage("B +32 handles +32 streams"))
