"""Build libltr_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python -m vllm_ltr_amd.csrc.build [--force]

Objects and the shared library are written next to the sources (in-tree, git-ignored)
so the library travels with the repository snapshot to the GPU box.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ["ltr_api.hip", "ltr_rank.hip", "ltr_rows.hip", "ltr_gemm.hip", "ltr_attn.hip", "ltr_pool.hip", "ltr_head.hip", "ltr_train.hip", "ltr_ndcg.hip", "ltr_trainer.hip"]
HEADERS = ["ltr_internal.h", os.path.join("..", "..", "include", "ltr_hip.h")]
LIB = os.path.join(HERE, "libltr_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -amdgpu-kernarg-preload-count: gfx950's command processor delivers the leading kernel arguments (pointers, sizes) in SGPRs at
# wave launch, so a kernel's first address computation does not wait for an s_load of its kernarg segment - worth 0.14 us per
# launch on the 79-launch one-request call (-1.7 % at k = 1, nothing on the cold call; bit-identical scores:
# profiles/r06_kernarg_preload.txt).  LTR_NO_KERNARG_PRELOAD=1 builds without it (the A/B).
PRELOAD = [] if os.environ.get("LTR_NO_KERNARG_PRELOAD") == "1" else ["-mllvm", "-amdgpu-kernarg-preload-count=16"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-fno-gpu-rdc"] + PRELOAD + os.environ.get("LTR_HIPCC_EXTRA", "").split()


# per-file flags: the attention kernel is bound by per-wave latency (a few hundred dependent VALU / LDS / MFMA
# instructions per key tile) and gains 6 % from the ILP-maximising scheduler; the GEMM loses 3 % with it
PER_FILE_FLAGS = {"ltr_attn.hip": ["-mllvm", "-amdgpu-sched-strategy=max-ilp"]}


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(src: str, force: bool) -> str:
    obj = os.path.join(HERE, src.replace(".hip", ".o"))
    deps = [os.path.join(HERE, src)] + [os.path.join(HERE, h) for h in HEADERS]
    extra = os.environ.get("LTR_FLAGS_" + src.split(".")[0].upper(), "").split()     # experiments: LTR_FLAGS_LTR_GEMM="..."
    if force or extra or _stale(obj, deps):
        cmd = [HIPCC, *FLAGS, *PER_FILE_FLAGS.get(src, []), *extra, "-c", os.path.join(HERE, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
        if r.stderr.strip():
            sys.stderr.write(r.stderr)
    return obj


def build(force: bool = False) -> str:
    with ThreadPoolExecutor(max_workers=len(SOURCES)) as ex:
        objs = list(ex.map(lambda s: _compile(s, force), SOURCES))
    if force or _stale(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    # the build checks its own output: no packed-f32 instruction of the form MI355X mis-executes beside a library fp16 GEMM
    # (isa_lint.py, profiles/r06_rln_fault.txt)
    from . import isa_lint
    isa_lint.check([LIB])
    _build_tools(force)
    _build_standalone(force)
    return LIB


# element-level checker of the GEMM's LayerNorm-fold epilogues (diag/gemm_check.hip), run by the -m gpu tests
TOOLS = {"gemm_check": os.path.join("diag", "gemm_check.hip"),
         # times the four dense layers of a decoder layer at M rows through launch_gemm (A/B of tile / split-K / XCD-map choices)
         "gemm_bench": os.path.join("diag", "gemm_bench.hip"),
         # hardware probe: a VALU write into the data registers of a 16-byte store 0 / 1 / 2 wait states behind it (isa_lint's second rule)
         "store_hazard_probe": os.path.join("diag", "store_hazard_probe.hip")}
# reproducers of the round-5 concurrency fault (profiles/r06_rln_fault.txt); they compile ltr_gemm.hip into themselves and use
# rocBLAS as the co-running library GEMM: name -> (source, extra flags)
ROCBLAS = os.path.exists("/opt/rocm/lib/librocblas.so") and os.path.exists("/opt/rocm/include/rocblas/rocblas.h")
STANDALONE = {"pk_opsel_probe": (os.path.join("diag", "pk_opsel_probe.hip"), []),
              "rln_fault": (os.path.join("diag", "rln_fault.hip"), []),
              "rln_fault_r5": (os.path.join("diag", "rln_fault.hip"), ["-DLTR_RLN_FAULT_SHAPE"])}


def _build_tools(force: bool) -> None:
    out_dir = os.path.join(HERE, "build")
    os.makedirs(out_dir, exist_ok=True)
    for name, src in TOOLS.items():
        exe = os.path.join(out_dir, name)
        deps = [os.path.join(HERE, src), LIB] + [os.path.join(HERE, h) for h in HEADERS]
        if force or _stale(exe, deps):
            cmd = [HIPCC, "--offload-arch=gfx950", "-O2", "-std=c++17", "-pthread", "-Wno-unused-result", "-I" + HERE,
                   "-I" + os.path.join(HERE, "..", "..", "include"), os.path.join(HERE, src), "-L" + HERE, "-lltr_hip",
                   "-Wl,-rpath,$ORIGIN/..", "-o", exe]
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")




def _build_standalone(force: bool) -> None:
    out_dir = os.path.join(HERE, "build")
    os.makedirs(out_dir, exist_ok=True)
    if not ROCBLAS:
        return
    def one(item):
        name, (src, extra) = item
        exe = os.path.join(out_dir, name)
        deps = [os.path.join(HERE, src), os.path.join(HERE, "ltr_gemm.hip")] + [os.path.join(HERE, h) for h in HEADERS]
        if force or _stale(exe, deps):
            cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-result", "-Wno-unused-function", "-DWITH_ROCBLAS", *extra,
                   "-I" + HERE, "-I" + os.path.join(HERE, "..", "..", "include"), os.path.join(HERE, src), "-L/opt/rocm/lib", "-lrocblas",
                   "-Wl,-rpath,/opt/rocm/lib", "-o", exe]
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    with ThreadPoolExecutor(max_workers=len(STANDALONE)) as ex:
        list(ex.map(one, STANDALONE.items()))


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
