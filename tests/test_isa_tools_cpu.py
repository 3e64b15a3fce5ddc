"""The assembly checks that are part of every build (gcgcn_amd/csrc/Makefile -> tools/isa_mfma_hazard_check.py,
tools/isa_spill_check.py), pinned on hand-written gfx950 snippets: the construct that made one round-4 build non-deterministic
(an MFMA result read by a spill store four wait states after the MFMA on the branch path into a join) must
be reported, the same code with enough wait states must pass, and a spill slot reloaded under a wider EXEC mask than its only
store must be reported."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

import isa_mfma_hazard_check as H   # noqa: E402
import isa_spill_check as S         # noqa: E402


def _kernel(body, name="_ZN2gc4testEv"):
    return f"{name}: ; @{name}\n" + textwrap.dedent(body).replace("    ", "\t", 1).replace("\n    ", "\n\t") + "\n.Lfunc_end0:\n"


def _lines(src):
    # the tools read compiler output: instructions start with a tab, labels do not
    out = []
    for ln in textwrap.dedent(src).strip("\n").splitlines():
        ln = ln.strip()
        out.append(ln if ln.endswith(":") or ln.startswith(".L") else "\t" + ln)
    return out


BAD_JOIN = """
    v_mfma_f32_16x16x4_f32 v[16:19], v39, v43, v[16:19]
    s_cbranch_vccnz .LBB0_2
    ds_read_b128 v[40:43], v1 offset:37824
    s_waitcnt lgkmcnt(0)
    v_mfma_f32_16x16x4_f32 v[2:5], v36, v40, v[2:5]
    v_mfma_f32_16x16x4_f32 v[2:5], v37, v41, v[2:5]
    v_mfma_f32_16x16x4_f32 v[2:5], v38, v42, v[2:5]
    v_mfma_f32_16x16x4_f32 v[2:5], v39, v43, v[2:5]
    .LBB0_2:
    s_or_b64 exec, exec, s[82:83]
    s_nop {nop}
    scratch_store_dwordx4 off, v[16:19], off offset:32
    s_nop 5
    scratch_store_dwordx4 off, v[2:5], off offset:48
    s_endpgm
"""


def test_mfma_result_read_too_early_on_the_branch_path_is_found():
    """The round-4 sequence, instruction for instruction: `s_nop 1` tops the fall-through path (8 wait states) up to the 10 an
    8-pass fp32 MFMA needs; the direct branch has 4."""
    findings, n_mfma = H.analyse(_lines(BAD_JOIN.format(nop=1)), 0)
    assert n_mfma == 5
    assert len(findings) == 1
    consumer, producer, reg, have, need = findings[0]
    assert consumer.op == "scratch_store_dwordx4" and producer.op == "v_mfma_f32_16x16x4_f32"
    assert reg in (16, 17, 18, 19) and have == 4 and need == 10


def test_enough_wait_states_on_every_path_pass():
    findings, _ = H.analyse(_lines(BAD_JOIN.format(nop=7)), 0)
    assert findings == []


def test_back_to_back_accumulation_and_straight_line_nops_pass():
    """Same destination and SrcC back to back is interlocked by the hardware (0 wait states); a VALU read after `s_nop 9` has its
    10; one wait state short is reported."""
    ok = """
        v_mfma_f32_16x16x4_f32 v[0:3], v8, v9, v[0:3]
        v_mfma_f32_16x16x4_f32 v[0:3], v10, v11, v[0:3]
        s_nop 9
        v_add_f32_e32 v20, v0, v21
        s_endpgm
    """
    assert H.analyse(_lines(ok), 0)[0] == []
    short = ok.replace("s_nop 9", "s_nop 8")
    f = H.analyse(_lines(short), 0)[0]
    assert len(f) == 1 and f[0][3] == 9 and f[0][4] == 10
    # a 16-pass product (32x32x2) needs 18
    big = """
        v_mfma_f32_32x32x2_f32 v[0:15], v20, v21, v[0:15]
        s_nop 15
        s_nop 0
        global_store_dword v[30:31], v7, off
        s_endpgm
    """
    f = H.analyse(_lines(big), 0)[0]
    assert len(f) == 1 and f[0][3] == 17 and f[0][4] == 18
    assert H.analyse(_lines(big.replace("s_nop 0", "s_nop 1")), 0)[0] == []


def test_partial_srcc_overlap_needs_the_passes():
    """SrcC overlapping the previous destination without being identical: `passes` wait states."""
    src = """
        v_mfma_f32_16x16x4_f32 v[14:17], v6, v10, v[16:19]
        {gap}
        v_mfma_f32_16x16x4_f32 v[20:23], v7, v11, v[16:19]
        s_endpgm
    """
    f = H.analyse(_lines(src.format(gap="s_nop 2")), 0)[0]
    assert len(f) == 1 and f[0][4] == 8
    assert H.analyse(_lines(src.format(gap="s_nop 7")), 0)[0] == []


def test_accumulation_registers_are_tracked_like_vector_registers():
    """A product whose destination is an a-register tuple, read back by v_accvgpr_read one wait state short; the v-register of
    the same number is a different register."""
    src = """
        v_mfma_f32_16x16x4_f32 a[0:3], v8, v9, a[0:3]
        s_nop {n}
        v_accvgpr_read_b32 v20, a1
        v_add_f32_e32 v21, v1, v2
        s_endpgm
    """
    f = H.analyse(_lines(src.format(n=8)), 0)[0]
    assert len(f) == 1 and f[0][2] == 1001 and f[0][3] == 9 and f[0][4] == 10
    assert H.analyse(_lines(src.format(n=9)), 0)[0] == []


def test_spill_reloaded_under_a_wider_mask_than_its_store_is_found():
    bad = """
        v_mov_b32_e32 v1, 0
        s_and_saveexec_b64 s[0:1], vcc
        s_cbranch_execz .LBB0_2
        scratch_store_dword off, v1, off offset:16
        .LBB0_2:
        s_or_b64 exec, exec, s[0:1]
        scratch_load_dword v2, off, off offset:16
        s_endpgm
    """
    r = S.analyse(_lines(bad))
    assert r["spill_dwords"] == 1 and r["masked"] == 1 and len(r["hazards"]) == 1
    good = bad.replace("        v_mov_b32_e32 v1, 0\n", "        v_mov_b32_e32 v1, 0\n        scratch_store_dword off, v1, off offset:16\n")
    r = S.analyse(_lines(good))
    assert r["hazards"] == []
    # a reload inside the same masked region as its store is fine
    inside = """
        s_and_saveexec_b64 s[0:1], vcc
        s_cbranch_execz .LBB0_2
        scratch_store_dword off, v1, off offset:16
        scratch_load_dword v2, off, off offset:16
        .LBB0_2:
        s_or_b64 exec, exec, s[0:1]
        s_endpgm
    """
    assert S.analyse(_lines(inside))["hazards"] == []


def test_the_build_ran_the_checks_on_the_shipped_kernels():
    """If the library was built here, its assembly is under gcgcn_amd/csrc/isa/ and the stamp of a clean check exists; run the
    MFMA check once more on the widest chain unit as a smoke test of the command line (skipped where only the .so travelled)."""
    isa = os.path.join(ROOT, "gcgcn_amd", "csrc", "isa")
    unit = os.path.join(isa, "chain_t_u2.s")
    if not os.path.exists(unit):
        import pytest
        pytest.skip("no assembly here (the library was built elsewhere)")
    assert os.path.exists(os.path.join(isa, "check.stamp")), "the build did not finish its assembly checks"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_mfma_hazard_check.py"), unit], capture_output=True, text=True)
    assert p.returncode == 0, p.stdout[-2000:]
