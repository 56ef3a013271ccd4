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
#   4. still flagged, or the checker could not verify anything (its exit status 2: no gfx950 code object / no kernel parsed): fail
#      (round 6: one more level first, see 5. below).
out=$1; src=$2; shift 2
here=$(dirname "$0")
# the LLVM tools next to the compiler in use (HIPCC of the Makefile), not a hard-coded ROCm path
hipcc_real=$(readlink -f "$(command -v "$1" 2>/dev/null || echo "$1")")
llvm_bin="$(dirname "$hipcc_real")/../lib/llvm/bin"
[ -x "$llvm_bin/llvm-objdump" ] || llvm_bin=/opt/rocm/lib/llvm/bin
check="python3 $here/../../tools/check_exec_prologue.py --quiet --llvm-bin=$llvm_bin"
rm -f "$out.fallback"
# returns 0 clean (without labels, or on the labelled assembly: then <out.o>.labelled_clean keeps what the unlabelled check of the object
# said -- tests/test_abi.py checks the LINKED library, where there are no labels, and accepts exactly these kernels) | 1 flagged; exits on a
# compile error or an unverifiable object.
# (Round 6 tried to prefer a build that is clean without labels too: it took the headline lockstep object to the basic allocator and the
#  B = 8 192 rate from 16.0 to 14.0 M steps/s -- the allocator fallback is not a 1-3 % matter on every object.)
checked_compile() {
  rm -f "$out.labelled_clean"
  "$@" $extra -c -o "$out" "$src" || { rc=$?; echo "== $out: compilation failed" >&2; exit $rc; }
  $check "$out" > "$out.check" 2>&1; rc=$?
  if [ $rc -eq 0 ]; then rm -f "$out.check"; return 0; fi
  if [ $rc -ne 1 ]; then cat "$out.check" >&2; echo "== $out: check_exec_prologue.py could not verify the object (status $rc)" >&2; rm -f "$out"; exit 2; fi
  "$@" $extra -S --cuda-device-only -o "$out.s" "$src" 2>/dev/null || { echo "== $out: could not emit the assembly" >&2; exit 1; }
  cp "$out.check" "$out.unlabelled"
  $check "$out.s" > "$out.check" 2>&1; rc=$?
  if [ $rc -eq 0 ]; then
    echo "== $out: flagged on the object, clean on the labelled assembly (a join without a branch target)"
    mv "$out.unlabelled" "$out.labelled_clean"; rm -f "$out.check" "$out.s"; return 0
  fi
  cat "$out.check"; rm -f "$out.s" "$out.unlabelled"
  [ $rc -eq 1 ] || { echo "== $out: check_exec_prologue.py could not verify the assembly (status $rc)" >&2; rm -f "$out"; exit 2; }
  return 1
}
# one attempt: $1 = what to write into the .fallback marker ("" for the plain build), rest = compiler and flags
attempt() {
  note=$1; shift
  if checked_compile "$@"; then
    [ -n "$note" ] && echo "$note" > "$out.fallback"
    exit 0
  fi
}
extra=""
attempt "" "$@"
echo "== $out: not clean with the plain build, recompiling with -mllvm -sgpr-regalloc=basic"
extra="-mllvm -sgpr-regalloc=basic"
attempt "-mllvm -sgpr-regalloc=basic" "$@"
#   5. (round 6) still not clean and the unit was built with -mllvm -amdgpu-mfma-vgpr-form=1 (MFMA accumulators in architected VGPRs: the
#      Makefile's VGPRFORM, worth ~1 % on the instances it is for): once more without it -- accumulators in AccVGPRs shift the whole
#      allocation --, default allocator first, then the basic one.  Seen on the free-running (20, 10) object of the round-6 source.
case " $* " in
  *" -mllvm -amdgpu-mfma-vgpr-form=1 "*)
    echo "== $out: not clean with the fallback allocator, recompiling without -amdgpu-mfma-vgpr-form"
    n=$#; i=0
    while [ $i -lt $n ]; do      # rebuild "$@" without the flag pair
      a=$1; shift; i=$((i + 1))
      if [ "$a" = "-mllvm" ] && [ "$1" = "-amdgpu-mfma-vgpr-form=1" ]; then shift; i=$((i + 1)); continue; fi
      set -- "$@" "$a"
    done
    extra=""
    attempt "without -amdgpu-mfma-vgpr-form" "$@"
    extra="-mllvm -sgpr-regalloc=basic"
    attempt "without -amdgpu-mfma-vgpr-form, -mllvm -sgpr-regalloc=basic" "$@"
    ;;
esac
echo "== $out: flagged with every fallback" >&2
rm -f "$out"
exit 1
