"""The F16 GEMM's LayerNorm-fold epilogues checked element by element (producer: f32 output, a' = split(x gamma 16)
operand planes, per-64-column (mean, M2); consumer: LN(x) W^T + b from a' and the statistics) against a host
computation in double - vllm_ltr_amd/csrc/diag/gemm_check.hip, built by build.py next to the library.  The scorer
tests see these paths only through 12-24 layers of model; this one pins the kernel itself (it is how a stale-register
store in one lane of 16 of an experimental epilogue was found).  The row counts cover the three tile
regimes of launch_gemm: small-tile, mid-tile and the 128 x 256 kernel - round 4: every tile configuration and K-part count
launch_gemm chooses or can be forced into, and row windows."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "vllm_ltr_amd", "csrc")


def _run(shape, env_extra=None, window=()):
    from vllm_ltr_amd.csrc import build
    build.build()
    exe = os.path.join(CSRC, "build", "gemm_check")
    env = dict(os.environ, LD_LIBRARY_PATH=CSRC + os.pathsep + os.environ.get("LD_LIBRARY_PATH", ""), **(env_extra or {}))
    r = subprocess.run([exe, *map(str, shape), *map(str, window)], capture_output=True, text=True, env=env, timeout=600)
    print(r.stdout)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "LNP: bad out 0, bad a' 0, bad stats 0" in r.stdout
    assert "LNC: bad 0 of" in r.stdout


# row counts across launch_gemm's choices (ltr_gemm.hip choose_small): 32 x 64 / 64 x 128 / 64 x 256 tiles, the narrow long-K
# shapes with 2 / 4 K parts + the reduce kernel, the 128 x 256 kernel with and without parts
# (the host reference is a scalar triple loop: widths of 256 keep the file to a minute; the true model widths run once each)
@pytest.mark.parametrize("shape", [(200, 128, 128), (1000, 768, 768), (129, 1024, 4096), (513, 256, 1024), (900, 256, 1024),
                                   (1400, 256, 1024), (2500, 256, 256), (3500, 256, 1024), (6400, 128, 128), (300, 768, 3072)])
def test_layernorm_fold_epilogues_element_by_element(shape):
    _run(shape)


# every kernel / split combination launch_gemm can be forced into (the lab's knobs), on one producer shape with a long K
@pytest.mark.parametrize("cfg,split", [(0, 1), (0, 2), (0, 4), (1, 1), (1, 4), (5, 1), (5, 2), (-1, 1), (-1, 2), (-1, 4)])
def test_forced_kernel_and_split_combinations(cfg, split):
    _run((700, 256, 1024), dict(LTR_GEMM_FORCE_CFG=str(cfg), LTR_GEMM_FORCE_SPLIT=str(split)))


# row windows (GemmArgs::row0 / ldm): the window's rows are right, every other row of every output is untouched - for
# the small-batch kernels, the split-K reduce and the 128 x 256 kernel; and launch_gemm's own use of them (tail rows)
@pytest.mark.parametrize("shape,window,env", [((1000, 256, 256), (128, 300), {}), ((2000, 256, 1024), (1536, 464), {}),
                                              ((9000, 256, 256), (3968, 5032), {}), ((9000, 256, 256), (0, 8960), {}),
                                              ((600, 256, 1024), (64, 520), {}),
                                              ((66000, 256, 256), (0, 66000), {"LTR_GEMM_TAIL": "1"})])
def test_row_windows(shape, window, env):
    _run(shape, env, window)
