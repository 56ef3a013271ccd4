"""The C-ABI library loads and exports every symbol include/mpcq.h declares; without a GPU the
product fails loudly instead of falling back to anything."""
import ctypes
import os
import re

import pytest

from mpc_quad_ros_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "mpcq.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(mpcq_[a-z0-9_]+)\s*\(", hdr)))


def test_header_and_loader_agree():
    assert declared_symbols() == sorted(n for n, _, _ in _lib.SYMBOLS)


def test_library_exports_every_declared_symbol():
    assert os.path.exists(_lib.DEFAULT_LIB), "build with: python -c 'import __graft_entry__ as g; g.build()'"
    lib = ctypes.CDLL(_lib.DEFAULT_LIB)
    for name in declared_symbols():
        assert hasattr(lib, name), name


def test_config_struct_matches_header_order():
    from mpc_quad_ros_amd.params import CConfig
    hdr = open(os.path.join(ROOT, "include", "mpcq.h")).read()
    body = hdr[hdr.index("typedef struct mpcq_config {") + len("typedef struct mpcq_config {"):hdr.index("} mpcq_config;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for decl in body.split(";"):
        parts = [p.strip() for p in decl.strip().split(",") if p.strip()]
        for part in parts:
            nm = re.findall(r"([A-Za-z_][A-Za-z0-9_]*)\s*(?:\[\d+\])?$", part)
            if nm:
                names.append(nm[0])
    assert names == [f[0] for f in CConfig._fields_]
    from mpc_quad_ros_amd.params import CTuning
    body = hdr[hdr.index("typedef struct mpcq_tuning {") + len("typedef struct mpcq_tuning {"):hdr.index("} mpcq_tuning;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = [re.findall(r"([A-Za-z_][A-Za-z0-9_]*)$", d.strip())[0] for d in body.split(";") if d.strip()]
    assert names == [f[0] for f in CTuning._fields_]


def test_code_objects_pass_the_exec_prologue_check():
    """Round 4 root-caused the code-generation-dependent wrong results of rounds 1-3 (tools/repro_codegen/README.md): ROCm 7.2's
    register allocator can place VGPR -> AGPR spill copies in front of the EXEC restore of a control-flow join, where they execute
    for no lane.  tools/check_exec_prologue.py finds that pattern in a code object: it must flag the committed excerpt of the
    failing kernel, and every kernel of the in-tree libmpcq.so must be clean (the Makefile enforces the same at build time)."""
    import subprocess
    import sys
    chk = os.path.join(ROOT, "tools", "check_exec_prologue.py")
    if not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump"):
        pytest.skip("no ROCm LLVM tools on this runner")
    bad = subprocess.run([sys.executable, chk, "--min-instructions=1", os.path.join(ROOT, "tools", "repro_codegen", "isa_excerpt_BB10_117.s")], capture_output=True, text=True)
    assert bad.returncode == 1 and "EXEC = 0" in bad.stdout and "v_accvgpr_write_b32" in bad.stdout, bad.stdout
    # the same misplacement behind an SGPR spill to scratch (-amdgpu-spill-sgpr-to-vgpr=0: EXEC saved, set to a constant, restored) and a
    # legitimate join (per-lane copy at the end of a divergent `if`, nothing but the EXEC restore behind a loop exit)
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        bad2, ok2 = os.path.join(tmp, "bad.s"), os.path.join(tmp, "ok.s")
        open(bad2, "w").write("_ZN4mpcq11step_kernel_synthetic_badEv:\n.LBB0_1:\n\ts_andn2_b64 exec, exec, s[18:19]\n\ts_cbranch_execnz .LBB0_1\n.LBB0_2:\n"
                              "\ts_mov_b64 s[4:5], exec\n\ts_mov_b64 exec, 3\n\tscratch_store_dword off, v0, off offset:8\n\tv_writelane_b32 v0, s34, 0\n"
                              "\ts_mov_b64 exec, s[4:5]\n\tv_accvgpr_write_b32 a1, v7\n\ts_or_b64 exec, exec, s[2:3]\n\ts_endpgm\n.Lfunc_end0:\n")
        open(ok2, "w").write("_ZN4mpcq11step_kernel_synthetic_okEv:\n.LBB1_1:\n\ts_andn2_b64 exec, exec, s[18:19]\n\ts_cbranch_execnz .LBB1_1\n.LBB1_2:\n"
                             "\tv_writelane_b32 v9, s34, 0\n\ts_or_b64 exec, exec, s[2:3]\n\tv_accvgpr_write_b32 a1, v7\n\ts_cbranch_execz .LBB1_4\n.LBB1_3:\n\tv_mov_b32_e32 v3, v4\n.LBB1_4:\n"
                             "\tv_mov_b32_e32 v5, v3\n\ts_or_b64 exec, exec, s[6:7]\n\ts_endpgm\n.Lfunc_end1:\n")
        r_bad = subprocess.run([sys.executable, chk, "--min-instructions=1", bad2], capture_output=True, text=True)
        r_ok = subprocess.run([sys.executable, chk, "--min-instructions=1", ok2], capture_output=True, text=True)
        assert r_bad.returncode == 1 and "v_accvgpr_write_b32 a1, v7" in r_bad.stdout, r_bad.stdout
        assert r_ok.returncode == 0, r_ok.stdout
        # the gate fails CLOSED (advisor finding of round 4): an input in which nothing could be verified -- an assembly file without a
        # kernel, a kernel of a handful of parsed lines, a host-only object, a missing file -- is exit status 2, not "clean"
        empty, host = os.path.join(tmp, "empty.s"), os.path.join(tmp, "host.o")
        open(empty, "w").write("\t.text\n")
        assert subprocess.run([sys.executable, chk, empty], capture_output=True, text=True).returncode == 2
        assert subprocess.run([sys.executable, chk, ok2], capture_output=True, text=True).returncode == 2            # 8 lines: below the default minimum
        assert subprocess.run([sys.executable, chk, os.path.join(tmp, "missing.o")], capture_output=True, text=True).returncode == 2
        src = os.path.join(tmp, "host.c"); open(src, "w").write("int f(int a) { return a + 1; }\n")
        if subprocess.run(["gcc", "-c", "-o", host, src]).returncode == 0:
            assert subprocess.run([sys.executable, chk, host], capture_output=True, text=True).returncode == 2
    # disassembly has no labels: a join that no branch targets is merged with the `if` body in front of it.  The body's own reload
    # (consumed inside the body) must pass, spill code directly in front of the restore must not
    import importlib.util
    spec = importlib.util.spec_from_file_location("check_exec_prologue", chk)
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    body = [("@0x10", None), (None, "v_accvgpr_read_b32 v29, a17"), (None, "global_store_dwordx4 v[28:29], v[22:25], off"), (None, "s_or_b64 exec, exec, s[24:25]")]
    assert mod.check(body, labelled=False) == [] and mod.check(body, labelled=True) != []
    late = [("@0x10", None), (None, "global_store_dwordx4 v[28:29], v[22:25], off"), (None, "v_accvgpr_write_b32 a1, v7"), (None, "v_writelane_b32 v9, s34, 0"),
            (None, "s_or_b64 exec, exec, s[24:25]")]
    assert mod.check(late, labelled=False) != []
    good = subprocess.run([sys.executable, chk, _lib.DEFAULT_LIB], capture_output=True, text=True)
    if good.returncode != 0:
        # Without labels the check over-approximates basic blocks (a join that no branch targets is merged with the `if` body in front of
        # it).  The build verifies every object it flags that way on the LABELLED assembly of the same compilation (csrc/cc_checked.sh) and
        # keeps what the unlabelled check said in <object>.labelled_clean: the linked library may be flagged in exactly those kernels.
        import glob
        flagged = set(re.findall(r": (void mpcq::\S.*?): \d+ block\(s\)", good.stdout))
        cleared = set()
        for f in glob.glob(os.path.join(ROOT, "mpc_quad_ros_amd", "csrc", "build", "*.labelled_clean")):
            cleared |= set(re.findall(r": (void mpcq::\S.*?): \d+ block\(s\)", open(f).read()))
        assert good.returncode == 1 and flagged and flagged <= cleared, (sorted(flagged - cleared), good.stdout[-2000:])
    # every step-kernel instance of the library was looked at, and really parsed: tens of thousands of instruction lines each
    lines = [int(n) for n in re.findall(r"step_kernel<.*?: clean \((\d+) lines\)", good.stdout)]
    assert len(lines) >= 20 and min(lines) > 5000, (len(lines), min(lines) if lines else None)


def test_no_gpu_fails_loudly():
    if os.path.exists("/dev/kfd"):
        pytest.skip("GPU present")
    from mpc_quad_ros_amd.engine import Engine
    from mpc_quad_ros_amd.params import EngineConfig
    with pytest.raises(_lib.MpcqError, match="no HIP device"):
        Engine(EngineConfig(batch=1, N=5))


def test_missing_library_fails_loudly(tmp_path):
    with pytest.raises(_lib.MpcqError, match="no CPU implementation"):
        _lib.load(str(tmp_path / "libmpcq.so"))


def test_toolchain_is_the_validated_one():
    """The kernels are validated with ROCm 7.2's hipcc (every instance at -O3; DESIGN.md section 3.5 records two
    code-generation observations with this compiler).  A different toolchain has to re-run the GPU suite, in particular
    test_kernel_variants_agree and tools/o3_discrepancy_probe.py, before its build is trusted."""
    import subprocess
    if not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc on this runner")
    out = subprocess.check_output(["/opt/rocm/bin/hipcc", "--version"]).decode()
    assert "HIP version: 7.2" in out, out.splitlines()[0]
