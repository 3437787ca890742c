"""Diagnostic: phase cycles of an attention workgroup on the BASELINE 8k queue.  Needs the library
built with  LTR_HIPCC_EXTRA=-DLTR_ATTN_TIMELINE python -m vllm_ltr_amd.csrc.build --force"""
import ctypes as C, sys
import torch
sys.path.insert(0, ".")
from bench import synthetic_queue
from vllm_ltr_amd import _lib
from vllm_ltr_amd.opt_spec import OPTSpec, seeded_checkpoint
from vllm_ltr_amd.scorer import HipOPTScorer
spec = OPTSpec.opt_125m()
sc = HipOPTScorer(spec, seeded_checkpoint(spec, 0), device="cuda:0")
ids, cu, lens = synthetic_queue(spec, 8192, 0)
sc.score(ids, cu)
lib = _lib.load()
buf = (C.c_ulonglong * 8)()
lib.ltr_debug_attn_timeline(buf, 1)
sc.score(ids, cu)
torch.cuda.synchronize()
lib.ltr_debug_attn_timeline(buf, 0)
n = buf[4]
print(f"workgroups {n}  tiles/workgroup {buf[5] / n:.2f}  cycles: setup {buf[0] / n:.0f}  "
      f"first tile wait {buf[1] / n:.0f}  tile loop {buf[2] / n:.0f}")
