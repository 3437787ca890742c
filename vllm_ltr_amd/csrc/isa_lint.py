"""ISA lint of the built gfx950 code objects: no packed-f32 instruction may select the HIGH register of src1 for its lo lane.

Why (profiles/r06_rln_fault.txt): on MI355X, `v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32` with `op_sel[1] = 1, op_sel[0] = 0`
(`op_sel:[0,1]`, `op_sel:[0,1,x]`) return a wrong lo half - the src1 term reads as zero - in lanes 48-63 whenever a wave on the
same SIMD issues gfx950's K-doubled MFMA forms (v_mfma_f32_32x32x16_* / 16x16x32_*: a library fp16 / bf16 GEMM, or this
library's own GEMM and attention kernels on another stream); every other operand-select placement (`[1,0]`, `[1,1]`, any `op_sel_hi`, src2) computes
correctly in the same test (diag/pk_opsel_probe.hip: 4 of 4 runs, ~130,000 wrong values per 16 M; 0 for the other forms).
hipcc emits the form when the y component of a float2 that lives in ONE 64-bit register (a 64-bit load, a phi of float2)
feeds vectorised f32 math; round 5 met it as "scores off by 1e-2 whenever another stream has a kernel in flight" in one
kernel instance.  The hazard is invisible to the compiler, so the build checks its own output:

    python -m vllm_ltr_amd.csrc.isa_lint [file.so | file.o ...]      (default: the library and every object next to it)

`build.py` runs this after linking and refuses a library that contains the form.
"""
from __future__ import annotations

import os
import re
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
OBJDUMP = os.environ.get("LLVM_OBJDUMP", "/opt/rocm/lib/llvm/bin/llvm-objdump")

# packed-f32 VOP3P with op_sel of src0 clear and of src1 set (the lo lane of src1 comes from the high register of the pair)
_BAD = re.compile(r"\b(v_pk_(?:mul|add|fma)_f32)\b[^\n;]*\bop_sel:\[0,1(?:,[01])?\]")
_LABEL = re.compile(r"^[0-9a-f]+ <([^>]+)>:")


def device_code_objects(path: str, workdir: str) -> list[str]:
    """Extract the gfx950 code objects embedded in a host ELF (.o / .so); a bare code object is returned as it is."""
    with open(path, "rb") as f:
        head = f.read(20)
    if head[:4] == b"\x7fELF" and head[18:20] == b"\xe0\x00":          # e_machine 224: AMDGPU
        return [path]
    base = os.path.basename(path)
    local = os.path.join(workdir, base)
    if not os.path.exists(local):
        os.symlink(os.path.abspath(path), local)
    r = subprocess.run([OBJDUMP, "--offloading", base], cwd=workdir, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"llvm-objdump --offloading {path} failed:\n{r.stderr}")
    return sorted(os.path.join(workdir, f) for f in os.listdir(workdir) if f.startswith(base + ".") and "amdgcn" in f)


# Second rule (round 6, profiles/r06_store_policy.txt): a vector store of MORE THAN 64 BITS reads its data registers over several
# cycles; on gfx940-class hardware a VALU instruction must not write one of them within two wait states of the store (LLVM:
# GCNHazardRecognizer::createsVALUHazard, VALUWaitStates = 2 with gfx940 instructions).  The compiler keeps that distance behind the
# stores it emits itself - it cannot for a store inside an asm statement (epi_store16).  The rule flags every VALU write of a
# store's data registers closer than that, wherever the store came from.
_WIDE_STORE = re.compile(r"^\s*(?:global|flat|buffer|scratch)_store_dwordx[34]\s+(.*)$")
_VREG = re.compile(r"^v(\d+)$|^v\[(\d+):(\d+)\]$")
STORE_DATA_WAIT_STATES = 2


def _vrange(tok: str):
    m = _VREG.match(tok.strip())
    if not m:
        return None
    return (int(m.group(1)),) * 2 if m.group(1) is not None else (int(m.group(2)), int(m.group(3)))


def _store_data(args: str):
    """Data registers of a wide store: the widest VGPR range among its operands (address pairs are two registers wide)."""
    best = None
    for tok in args.split(","):
        r = _vrange(tok.split()[0] if tok.split() else "")
        if r and r[1] - r[0] >= 2 and (best is None or r[1] - r[0] > best[1] - best[0]):
            best = r
    return best


def _store_data_hazards(ins: list[str]) -> list[tuple[int, str]]:
    """ins: the instructions of one kernel (text before the // comment).  Returns (index of the store, description)."""
    out = []
    for i, line in enumerate(ins):
        m = _WIDE_STORE.match(line)
        if not m:
            continue
        data = _store_data(m.group(1))
        if data is None:
            continue
        ws, k = 0, i + 1
        while ws < STORE_DATA_WAIT_STATES and k < len(ins):
            parts = ins[k].split(None, 1)
            op, args = parts[0], (parts[1] if len(parts) > 1 else "")
            if op == "s_nop":
                ws += int(args.strip() or "0", 0) + 1
                k += 1
                continue
            if op.startswith("v_"):
                dst = _vrange(args.split(",")[0]) if args else None
                if dst and not (dst[1] < data[0] or dst[0] > data[1]):
                    out.append((i, f"{line.strip()}  <-  {ins[k].strip()}  ({ws} wait state(s) after the store)"))
                    break
            if op in ("s_branch", "s_endpgm", "s_setpc_b64") or op.startswith("s_cbranch"):
                break                     # (control flow: the fall-through / target start a new window; the compiler's stores are safe there)
            ws += 1
            k += 1
    return out


def lint_code_object(co: str) -> list[tuple[str, str]]:
    r = subprocess.run([OBJDUMP, "-d", co], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"llvm-objdump -d {co} failed:\n{r.stderr}")
    found, kernel, body = [], "?", []

    def flush():
        for _, desc in _store_data_hazards(body):
            found.append((kernel, "store-data hazard: " + desc))

    for line in r.stdout.splitlines():
        m = _LABEL.match(line)
        if m:
            flush()
            kernel, body = m.group(1), []
            continue
        text = line.split("//")[0].strip()
        if text:
            body.append(text)
        if _BAD.search(line):
            found.append((kernel, text))
    flush()
    return found


def lint(paths: list[str]) -> list[tuple[str, str, str]]:
    out = []
    with tempfile.TemporaryDirectory() as wd:
        for p in paths:
            cos = device_code_objects(p, wd)
            if not cos:
                raise RuntimeError(f"{p}: no gfx950 code object found")
            for co in cos:
                out += [(os.path.basename(p), k, ins) for k, ins in lint_code_object(co)]
    return out


def check(paths: list[str]) -> None:
    bad = lint(paths)
    hz = [b for b in bad if b[2].startswith("store-data hazard")]
    if hz:
        lines = "\n".join(f"  {f}: {k}: {ins}" for f, k, ins in hz[:20])
        raise RuntimeError(
            f"ISA lint: {len(hz)} VALU write(s) of the data registers of a > 64-bit store within {STORE_DATA_WAIT_STATES} wait states of it "
            f"(the store may pick up the new value; profiles/r06_store_policy.txt).  A store inside an asm statement must carry its own "
            f"wait states (`s_nop 1` behind it):\n{lines}")
    if bad:
        lines = "\n".join(f"  {f}: {k}: {ins}" for f, k, ins in bad[:20])
        raise RuntimeError(
            f"ISA lint: {len(bad)} packed-f32 instruction(s) select the high register of src1 for the lo lane (op_sel:[0,1...]) - wrong "
            f"results in lanes 48-63 beside a library fp16 GEMM on MI355X (profiles/r06_rln_fault.txt).  Keep the y component of a "
            f"64-bit float2 out of vectorised f32 math (pass it through `asm volatile(\"\" : \"+v\"(y))` first):\n{lines}")


if __name__ == "__main__":
    args = sys.argv[1:] or [os.path.join(HERE, "libltr_hip.so")]
    bad = lint(args)
    for f, k, ins in bad:
        print(f"{f}: {k}: {ins}")
    print(f"isa_lint: {len(bad)} finding(s) in {len(args)} file(s)")
    sys.exit(1 if bad else 0)
