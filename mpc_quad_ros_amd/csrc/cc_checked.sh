#!/bin/sh
# cc_checked.sh <out.o> <source> <compiler and flags...>
# Compile one translation unit and run tools/check_exec_prologue.py on the result (the one miscompile pattern of this toolchain that
# has produced wrong code objects from this source: tools/repro_codegen/README.md).
#   1. compile to <out.o> (a compile error ends the script with the compiler's status), check the object.  Without labels the check
#      over-approximates basic blocks (a join that no branch targets -- the compiler removes the skip branch of a short `if` -- is merged
#      with the body in front of it, whose own reloads then look like spill code in front of the EXEC restore);
#   2. on a hit, emit the device assembly of the same compilation (labels = the compiler's own basic blocks) and check that: clean -> keep;
#   3. still flagged: recompile with the basic SGPR allocator (-mllvm -sgpr-regalloc=basic: SGPR spills at definitions and uses, not at
#      block tops -- the trigger is gone; 1-3 % slower), check again the same way, leave the marker <out.o>.fallback next to the object
#      (one file per object: nothing shared between the jobs of `make -j`);
#   4. still flagged, or the checker could not verify anything (its exit status 2: no gfx950 code object / no kernel parsed): fail.
out=$1; src=$2; shift 2
here=$(dirname "$0")
# the LLVM tools next to the compiler in use (HIPCC of the Makefile), not a hard-coded ROCm path
hipcc_real=$(readlink -f "$(command -v "$1" 2>/dev/null || echo "$1")")
llvm_bin="$(dirname "$hipcc_real")/../lib/llvm/bin"
[ -x "$llvm_bin/llvm-objdump" ] || llvm_bin=/opt/rocm/lib/llvm/bin
check="python3 $here/../../tools/check_exec_prologue.py --quiet --llvm-bin=$llvm_bin"
rm -f "$out.fallback"
# returns 0 clean | 1 flagged; exits on a compile error or an unverifiable object
checked_compile() {
  "$@" $extra -c -o "$out" "$src" || { rc=$?; echo "== $out: compilation failed" >&2; exit $rc; }
  $check "$out" > "$out.check" 2>&1; rc=$?
  if [ $rc -eq 0 ]; then rm -f "$out.check"; return 0; fi
  if [ $rc -ne 1 ]; then cat "$out.check" >&2; echo "== $out: check_exec_prologue.py could not verify the object (status $rc)" >&2; rm -f "$out"; exit 2; fi
  "$@" $extra -S --cuda-device-only -o "$out.s" "$src" 2>/dev/null || { echo "== $out: could not emit the assembly" >&2; exit 1; }
  $check "$out.s" > "$out.check" 2>&1; rc=$?
  if [ $rc -eq 0 ]; then
    echo "== $out: flagged on the object, clean on the labelled assembly (a join without a branch target)"; rm -f "$out.check" "$out.s"; return 0
  fi
  cat "$out.check"; rm -f "$out.s"
  [ $rc -eq 1 ] || { echo "== $out: check_exec_prologue.py could not verify the assembly (status $rc)" >&2; rm -f "$out"; exit 2; }
  return 1
}
extra=""
if checked_compile "$@"; then exit 0; fi
echo "== $out: flagged by check_exec_prologue.py, recompiling with -mllvm -sgpr-regalloc=basic"
extra="-mllvm -sgpr-regalloc=basic"
if checked_compile "$@"; then echo "-mllvm -sgpr-regalloc=basic" > "$out.fallback"; exit 0; fi
echo "== $out: still flagged with the fallback allocator" >&2
rm -f "$out"
exit 1
