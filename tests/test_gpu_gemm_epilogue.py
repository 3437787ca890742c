"""The F16 GEMM's LayerNorm-fold epilogues checked element by element (producer: f32 output, a' = split(x gamma 16)
operand planes, per-64-column (mean, M2); consumer: LN(x) W^T + b from a' and the statistics) against a host
computation in double - vllm_ltr_amd/csrc/diag/gemm_check.hip, built by build.py next to the library.  The scorer
tests see these paths only through 12-24 layers of model; this one pins the kernel itself (it is how a stale-register
store in one lane of 16 of an experimental epilogue was found).  The row counts cover the three tile
regimes of launch_gemm: small-tile (M <= 1024), mid-tile (<= 3072) and the 128 x 256 kernel."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "vllm_ltr_amd", "csrc")


@pytest.mark.parametrize("shape", [(200, 128, 128), (1000, 768, 768), (129, 1024, 4096), (513, 768, 3072), (2500, 768, 768), (6400, 128, 128)])
def test_layernorm_fold_epilogues_element_by_element(shape):
    from vllm_ltr_amd.csrc import build
    build.build()
    exe = os.path.join(CSRC, "build", "gemm_check")
    env = dict(os.environ, LD_LIBRARY_PATH=CSRC + os.pathsep + os.environ.get("LD_LIBRARY_PATH", ""))
    r = subprocess.run([exe, *map(str, shape)], capture_output=True, text=True, env=env, timeout=600)
    print(r.stdout)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "LNP: bad out 0, bad a' 0, bad stats 0" in r.stdout
    assert "LNC: bad 0 of" in r.stdout
