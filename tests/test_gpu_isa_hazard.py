"""Round 5's concurrency-dependent wrong scores, root-caused in round 6 (profiles/r06_rln_fault.txt): a packed-f32 operand-select form
(`v_pk_{mul,add,fma}_f32 ... op_sel:[0,1...]`) computes a wrong lo half in lanes 48-63 on MI355X while a library fp16 / bf16 GEMM shares
the CU.  The standalone reproducers, on the hardware:
  * the split-K reduce kernel with the SHIPPED expression, on fixed inputs beside rocBLAS GEMMs: every run bit-identical to the idle run;
  * the same kernel with ROUND 5's expression: differs (reported, not asserted - it documents the hazard on this box);
  * one instruction per kernel, every operand-select placement: only the forms the build's ISA lint rejects may differ."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "vllm_ltr_amd", "csrc", "build")


def _run(name, *args, timeout=300):
    exe = os.path.join(BUILD, name)
    if not os.path.exists(exe):
        pytest.skip(f"{name} not built (rocBLAS missing)")
    return subprocess.run([exe, *args], capture_output=True, text=True, timeout=timeout, cwd=ROOT)


def test_reduce_kernel_beside_library_gemms_is_exact():
    r = _run("rln_fault", "20", "1383", "quick")
    assert r.returncode == 0 and "E5 0 of 20" in r.stdout, (r.stdout + r.stderr)[-2000:]
    print(r.stdout.strip().splitlines()[-1])


def test_round5_expression_still_shows_the_hazard():
    r = _run("rln_fault_r5", "20", "1383", "quick")
    last = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-500:]
    print(("hazard reproduced: " if r.returncode == 1 else "hazard NOT reproduced on this box: ") + last)
    assert r.returncode in (0, 1), (r.stdout + r.stderr)[-2000:]


def test_only_lint_rejected_forms_differ():
    r = _run("pk_opsel_probe", "2", "quick")
    assert r.returncode in (0, 1), (r.stdout + r.stderr)[-3000:]          # 2 = a form the library may contain differed
    print(r.stdout.strip().splitlines()[-1])


def test_wide_store_needs_two_wait_states_before_its_data_is_overwritten():
    """The second hazard the build lints for (isa_lint.py rule 2, profiles/r06_store_policy.txt), on the hardware: a VALU write into a
    data register of a 16-byte store lands in memory when it follows the store by 0 wait states (about a quarter of the overwritten
    words) or, rarely, by 1; never by 2 - the distance the library's asm epilogue stores now carry (`s_nop 1`).  The probe exits
    non-zero if a word is wrong behind two wait states; the 0-wait-state line documents that the hazard is real on this box."""
    r = _run("store_hazard_probe", timeout=300)
    assert r.returncode == 0 and "wrong words with two wait states: 0" in r.stdout, (r.stdout + r.stderr)[-2000:]
    first = r.stdout.strip().splitlines()[0]
    print(first)
    assert "after 0 wait state" in first and int(first.split(":")[1].split()[0]) > 0, first
