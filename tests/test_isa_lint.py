"""The build's ISA lint (vllm_ltr_amd/csrc/isa_lint.py): the library contains no packed-f32 instruction of the operand-select form that
MI355X mis-executes beside a library fp16 / bf16 GEMM (profiles/r06_rln_fault.txt), and the lint does see the form when it is there.
CPU only: hipcc cross-compiles, llvm-objdump disassembles."""
import os
import subprocess

import pytest

from vllm_ltr_amd.csrc import build, isa_lint

HIPCC = build.HIPCC


def test_library_is_free_of_the_hazardous_form():
    lib = build.build()
    assert isa_lint.lint([lib]) == []


def test_lint_sees_a_planted_instruction(tmp_path):
    src = tmp_path / "planted.hip"
    src.write_text(r'''
#include <hip/hip_runtime.h>
typedef float v2f __attribute__((ext_vector_type(2)));
__global__ void planted(const v2f* a, const v2f* b, v2f* o) {
  v2f x = a[threadIdx.x], y = b[threadIdx.x], d, e, f;
  asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=&v"(d) : "v"(x), "v"(y));                 // hazardous
  asm volatile("v_pk_fma_f32 %0, %1, %2, %1 op_sel:[0,1,0]" : "=&v"(e) : "v"(x), "v"(y));           // hazardous
  asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0]" : "=&v"(f) : "v"(x), "v"(y));                 // fine
  o[threadIdx.x] = d + e + f;
}
''')
    obj = tmp_path / "planted.o"
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-c", str(src), "-o", str(obj)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    found = isa_lint.lint([str(obj)])
    assert len(found) == 2 and all("planted" in k for _, k, _ in found), found
    assert {ins.split()[0] for _, _, ins in found} == {"v_pk_mul_f32", "v_pk_fma_f32"}
    with pytest.raises(RuntimeError, match="op_sel"):
        isa_lint.check([str(obj)])


def test_round5_expression_is_what_the_lint_rejects(tmp_path):
    """ltr_gemm.hip built with -DLTR_RLN_FAULT_SHAPE (round 5's `ptr ? ptr[row] : combine(...)` in the split-K reduce kernel) contains
    the instruction the fault was traced to; the shipped expression does not."""
    obj = tmp_path / "gemm_r5.o"
    r = subprocess.run([HIPCC, *build.FLAGS, "-DLTR_RLN_FAULT_SHAPE", "-c", os.path.join(build.HERE, "ltr_gemm.hip"), "-o", str(obj)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    found = isa_lint.lint([str(obj)])
    assert found and all("splitk_epilogue_kernel" in k and "v_pk_mul_f32" in ins for _, k, ins in found), found


def test_lint_sees_a_valu_write_into_the_data_of_a_wide_store(tmp_path):
    """Second rule: a VALU write of a data register of a > 64-bit store within two wait states of it (the store reads its data over
    several cycles; the compiler keeps the distance behind its own stores, not behind one inside an asm statement)."""
    src = tmp_path / "planted_store.hip"
    src.write_text(r'''
#include <hip/hip_runtime.h>
__global__ void planted_bad(float* o) {
  asm volatile("global_store_dwordx4 %0, v[10:13], off nt\n\tv_mov_b32 v12, 0" ::"v"(o + threadIdx.x * 4) : "v10", "v11", "v12", "v13", "memory");
}
__global__ void planted_one_state(float* o) {
  asm volatile("global_store_dwordx4 %0, v[10:13], off\n\ts_nop 0\n\tv_add_f32 v10, v10, v10" ::"v"(o + threadIdx.x * 4) : "v10", "v11", "v12", "v13", "memory");
}
__global__ void planted_fine(float* o) {
  asm volatile("global_store_dwordx4 %0, v[10:13], off nt\n\ts_nop 1\n\tv_mov_b32 v12, 0" ::"v"(o + threadIdx.x * 4) : "v10", "v11", "v12", "v13", "memory");
  asm volatile("global_store_dwordx2 %0, v[10:11], off\n\tv_mov_b32 v10, 0" ::"v"(o + threadIdx.x * 4) : "v10", "v11", "memory");   // 64 bits: no hazard
  asm volatile("global_store_dwordx4 %0, v[10:13], off\n\tv_mov_b32 v14, 0" ::"v"(o + threadIdx.x * 4) : "v10", "v11", "v12", "v13", "v14", "memory");
}
''')
    obj = tmp_path / "planted_store.o"
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-c", str(src), "-o", str(obj)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    found = isa_lint.lint([str(obj)])
    assert len(found) == 2 and all(ins.startswith("store-data hazard") for _, _, ins in found), found
    assert sorted("bad" in k or "one_state" in k for _, k, _ in found) == [True, True] and not any("fine" in k for _, k, _ in found)
    with pytest.raises(RuntimeError, match="wait states"):
        isa_lint.check([str(obj)])


def test_stores_of_rounds_2_to_6_are_what_the_second_rule_rejects(tmp_path):
    """ltr_gemm.hip with its asm stores as rounds 2-6 shipped them (no wait states behind the store: -DLTR_EPI_POST="") has VALU writes
    into store data one instruction behind the store - what corrupted the a' planes the moment the code around the stores changed
    (profiles/r06_store_policy.txt); the shipped file has none."""
    obj = tmp_path / "gemm_nopost.o"
    r = subprocess.run([HIPCC, *build.FLAGS, '-DLTR_EPI_POST=""', "-c", os.path.join(build.HERE, "ltr_gemm.hip"), "-o", str(obj)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    found = isa_lint.lint([str(obj)])
    assert len(found) >= 8 and all(ins.startswith("store-data hazard") and " nt " in ins for _, _, ins in found), found[:3]
