#!/usr/bin/env python3
"""Static check of gfx950 code objects for ONE miscompile pattern of ROCm 7.2's compiler (found in round 4, tools/repro_codegen/README.md):

    a control-flow join block that is entered with EXEC = 0 (the exit of a divergent loop:  s_andn2_b64 exec, exec, sX /
    s_cbranch_execnz LOOP  falls through with no lane active) or with a partial EXEC (the end of a divergent `if`) restores
    the saved mask with  s_or_b64 exec, exec, sY  -- and the register allocator has put EXEC-dependent spill code
    (v_accvgpr_write_b32 aN, vM  = VGPR -> AGPR spill, rematerialised v_mov_b32, scratch stores) IN FRONT of that restore,
    behind SGPR spills (v_writelane_b32) it had placed there first.  Those copies execute for no lane, or only for the
    lanes that took the branch: the spill slot keeps its old contents and a later reload delivers garbage.

The check disassembles every kernel and reports, per basic block, EXEC-dependent vector instructions that sit between the
start of the block and a leading  s_or_b64 exec, exec, ...  (no other EXEC write in between).  v_writelane / v_readlane /
v_readfirstlane and scalar instructions ignore EXEC and are fine there; an SGPR spill to scratch (-amdgpu-spill-sgpr-to-vgpr=0) that saves
EXEC, sets it to a constant and restores it is looked through.

usage: check_exec_prologue.py file.o|file.so|file.s [...]      exit status 1 if any kernel shows the pattern"""
import re
import subprocess
import sys
import tempfile
import os

LLVM = "/opt/rocm/lib/llvm/bin"
IGNORES_EXEC = ("v_writelane", "v_readlane", "v_readfirstlane", "s_", ";", "v_nop")


def code_objects(path):
    """gfx950 code objects inside a host object / shared library (clang offload bundle), or the file itself."""
    if path.endswith(".s"):
        return [("asm", path)]
    out = []
    tmp = tempfile.mkdtemp()
    data = open(path, "rb").read()
    # host objects and libraries carry one or more __CLANG_OFFLOAD_BUNDLE__ blobs in .hip_fatbin
    r = subprocess.run([f"{LLVM}/clang-offload-bundler", "--list", "--type=o", f"--input={path}"], capture_output=True, text=True)
    if r.returncode == 0 and "gfx950" in r.stdout:
        for t in r.stdout.split():
            if "gfx950" in t:
                o = os.path.join(tmp, "dev.co")
                subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--type=o", f"--targets={t}", f"--input={path}", f"--output={o}", "--unbundle"])
                out.append(("co", o))
        return out
    # a linked library: extract the fat binary section and split the bundles
    sec = os.path.join(tmp, "fatbin")
    if subprocess.run([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={sec}", path, os.path.join(tmp, "x")], capture_output=True).returncode == 0 and os.path.exists(sec):
        blob = open(sec, "rb").read()
        magic = b"__CLANG_OFFLOAD_BUNDLE__"
        starts = [m.start() for m in re.finditer(re.escape(magic), blob)]
        for k, s in enumerate(starts):
            piece = os.path.join(tmp, f"bundle{k}")
            open(piece, "wb").write(blob[s:starts[k + 1] if k + 1 < len(starts) else len(blob)])
            r = subprocess.run([f"{LLVM}/clang-offload-bundler", "--list", "--type=o", f"--input={piece}"], capture_output=True, text=True)
            for t in r.stdout.split():
                if "gfx950" in t:
                    o = os.path.join(tmp, f"dev{k}.co")
                    subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--type=o", f"--targets={t}", f"--input={piece}", f"--output={o}", "--unbundle"])
                    out.append(("co", o))
        return out
    return [("co", path)]


def kernels_from_asm(path):
    """{name: [(label-or-None, text)]} from compiler assembly output (-S)."""
    ks, cur, name = {}, None, None
    for line in open(path):
        m = re.match(r"(_Z\w+):", line)
        if m and "step_kernel" in m.group(1) or (m and "kernel" in m.group(1)):
            name, cur = m.group(1), []
            ks[name] = cur
            continue
        if cur is None:
            continue
        if line.startswith(".Lfunc_end"):
            cur = None
            continue
        s = line.strip()
        m = re.match(r"(\.LBB\d+_\d+):", s)
        if m:
            cur.append((m.group(1), None))
        elif s and not s.startswith((";", ".")):
            cur.append((None, s))
    return ks


def kernels_from_co(path):
    """{name: [(label-or-None, text)]} from llvm-objdump: a block starts at every branch target and behind every branch."""
    txt = subprocess.check_output([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", path], text=True)
    raw, cur = {}, None
    for line in txt.splitlines():
        m = re.match(r"[0-9a-f]+ <(\S+)>:", line)
        if m:
            cur = []
            raw[m.group(1)] = cur
            continue
        if cur is None:
            continue
        m = re.match(r"\s+(.*?)\s*// ([0-9A-F]+):", line)
        if m:
            cur.append((int(m.group(2), 16), m.group(1).strip()))
    ks = {}
    for name, seq in raw.items():
        starts = set()
        for k, (addr, t) in enumerate(seq):
            op = t.split()[0]
            if op.startswith(("s_cbranch", "s_branch")):
                imm = int(t.split()[-1])
                imm = imm - 65536 if imm >= 32768 else imm
                starts.add(addr + 4 + 4 * imm)
                if k + 1 < len(seq):
                    starts.add(seq[k + 1][0])
            elif op.startswith("s_setpc") and k + 1 < len(seq):
                starts.add(seq[k + 1][0])
        ins = []
        for addr, t in seq:
            if addr in starts:
                ins.append((f"@{addr:#x}", None))
            ins.append((None, t))
        ks[name] = ins
    return ks


SPILL_OPS = ("v_accvgpr_write", "v_accvgpr_read", "scratch_store", "scratch_load", "buffer_store", "buffer_load")


def check(ins, labelled=True):
    """ins: [(label, text)].  A block starts at a label.  Findings: (label, instructions, why).
    labelled = False (disassembly: block starts are the branch targets and the instructions behind branches): a join that no branch
    targets -- the compiler drops the skip branch of a short `if` -- is merged with the body in front of it, whose own reloads (consumed
    inside the body, under the body's EXEC: legitimate) must not count; there only the spill code DIRECTLY in front of the restore
    counts (nothing but EXEC-independent instructions in between), which is where the misplaced prologue sits.
    - the block is the fall-through of  s_cbranch_execnz  (exit of a divergent loop: EXEC = 0): EVERY EXEC-dependent
      instruction in front of the restore is lost;
    - otherwise (end of a divergent `if`: the lanes of the branch are active) per-lane copies are legitimate there, but spill
      code (VGPR <-> AGPR copies, scratch traffic) is not: it saves / restores a register for a subset of its lanes."""
    findings = []
    n = len(ins)
    for i in range(n):
        lab, _ = ins[i]
        if lab is None:
            continue
        prev = next((ins[k][1] for k in range(i - 1, -1, -1) if ins[k][0] is None), "")
        exec_zero = prev.startswith("s_cbranch_execnz")
        head = []
        saved, inside = None, False      # SGPR spills to scratch wrap themselves in  s_mov sX, exec / s_mov exec, imm / ... / s_mov exec, sX: EXEC is unchanged behind them
        for j in range(i + 1, n):
            if ins[j][0] is not None:
                break
            t = ins[j][1]
            op = t.split()[0]
            m = re.match(r"s_mov_b64 (s\[\d+:\d+\]), exec$", t)
            if m and not inside:
                saved = m.group(1)
                continue
            if saved and not inside and re.match(r"s_mov_b64 exec, (-?\d+|0x[0-9a-f]+)$", t):
                inside = True
                continue
            if inside:
                if t == f"s_mov_b64 exec, {saved}":
                    inside, saved = False, None
                continue
            if re.match(r"s_or_b64 exec, exec,", t):
                dep = [h for h in head if not h.startswith(IGNORES_EXEC)]
                if exec_zero:
                    bad = dep
                elif labelled:
                    bad = [h for h in dep if h.startswith(SPILL_OPS)]
                else:
                    bad = []
                    for h in reversed(dep):
                        if not h.startswith(SPILL_OPS):
                            break
                        bad.insert(0, h)
                if bad:
                    findings.append((lab, bad, "entered with EXEC = 0 (exit of a divergent loop)" if exec_zero else "spill code under the partial EXEC of a branch"))
                break
            if op.startswith(("s_", "v_cmpx")) and "exec" in t.split(",")[0]:      # any other EXEC write ends the head
                break
            if op.startswith(("s_cbranch", "s_branch", "s_setpc", "s_endpgm", "s_barrier")):
                break
            head.append(t)
    return findings


MIN_INSTRUCTIONS = 50      # a kernel of this library disassembles to tens of thousands of lines; a handful means the parse failed


def main(paths):
    """Exit status: 0 every kernel parsed and clean | 1 the pattern was found | 2 NOTHING COULD BE VERIFIED for some input (no gfx950
    code object, no kernel symbol, or a kernel with fewer than MIN_INSTRUCTIONS parsed lines: a changed disassembly format, another
    ARCH, a host-only object) -- the gate fails closed.  --expect-kernels N: additionally require at least N kernels per input."""
    quiet = "--quiet" in paths
    expect, min_ins = 1, MIN_INSTRUCTIONS
    args = [p for p in paths if p != "--quiet"]
    paths = []
    k = 0
    while k < len(args):
        if args[k] == "--expect-kernels":
            expect = int(args[k + 1]); k += 2
        elif args[k].startswith("--expect-kernels="):
            expect = int(args[k].split("=", 1)[1]); k += 1
        elif args[k].startswith("--llvm-bin="):             # the LLVM tools of the compiler in use (csrc/cc_checked.sh passes them)
            global LLVM
            LLVM = args[k].split("=", 1)[1]; k += 1
        elif args[k].startswith("--min-instructions="):      # (unit tests of the checker on hand-written snippets)
            min_ins = int(args[k].split("=", 1)[1]); k += 1
        else:
            paths.append(args[k]); k += 1
    if not paths:
        print("check_exec_prologue.py: no input", file=sys.stderr)
        return 2
    total, unverified = 0, 0
    for p in paths:
        nk = 0
        cos = code_objects(p) if os.path.exists(p) else []
        for kind, f in cos:
            ks = kernels_from_asm(f) if kind == "asm" else kernels_from_co(f)
            for name, ins in ks.items():
                nk += 1
                fnd = check(ins, labelled=(kind == "asm"))
                dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
                dem = re.sub(r"\(mpcq::DevModel.*", "", dem)
                if len(ins) < min_ins:
                    unverified += 1
                    print(f"{os.path.basename(p)}: {dem}: NOT VERIFIED: only {len(ins)} instruction line(s) parsed")
                if fnd:
                    total += len(fnd)
                    print(f"{os.path.basename(p)}: {dem}: {len(fnd)} block(s) with EXEC-dependent code in front of the EXEC restore")
                    for lab, bad, why in fnd[:6]:
                        print(f"    block {lab}: {why}: {len(bad)} instruction(s), e.g. {bad[0]} | {bad[-1]}")
                elif not quiet and len(ins) >= min_ins:
                    print(f"{os.path.basename(p)}: {dem}: clean ({len(ins)} lines)")
        if nk < expect:
            unverified += 1
            print(f"{os.path.basename(p)}: NOT VERIFIED: {nk} kernel(s) found in {len(cos)} gfx950 code object(s), expected at least {expect}")
    if total:
        return 1
    return 2 if unverified else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
