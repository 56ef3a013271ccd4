#!/bin/sh
# cc_checked.sh <out.o> <source> <compiler and flags...>
# Compile one translation unit and run tools/check_exec_prologue.py on the result (the one miscompile pattern of this toolchain that
# has produced wrong code objects from this source: tools/repro_codegen/README.md).
#   1. compile to <out.o>, check the object.  Without labels the check over-approximates basic blocks (a join that no branch targets --
#      the compiler removes the skip branch of a short `if` -- is merged with the body in front of it, whose own reloads then look
#      like spill code in front of the EXEC restore);
#   2. on a hit, emit the device assembly of the same compilation (labels = the compiler's own basic blocks) and check that: clean -> keep;
#   3. still flagged: recompile with the basic SGPR allocator (-mllvm -sgpr-regalloc=basic: SGPR spills at definitions and uses, not at
#      block tops -- the trigger is gone; 1-3 % slower), check again the same way, record the object in build/fallback_objects.txt;
#   4. still flagged: fail.
set -e
out=$1; src=$2; shift 2
here=$(dirname "$0")
check="python3 $here/../../tools/check_exec_prologue.py --quiet"
sed -i "\|^$out:|d" build/fallback_objects.txt 2>/dev/null || true
checked_compile() {   # extra flags in "$extra"
  "$@" $extra -c -o "$out" "$src"
  if $check "$out" > "$out.check" 2>&1; then rm -f "$out.check"; return 0; fi
  "$@" $extra -S --cuda-device-only -o "$out.s" "$src" 2>/dev/null
  if $check "$out.s" > "$out.check" 2>&1; then
    echo "== $out: flagged on the object, clean on the labelled assembly (a join without a branch target)"; rm -f "$out.check" "$out.s"; return 0
  fi
  cat "$out.check"; rm -f "$out.s"
  return 1
}
extra=""
if checked_compile "$@"; then exit 0; fi
echo "== $out: flagged by check_exec_prologue.py, recompiling with -mllvm -sgpr-regalloc=basic"
extra="-mllvm -sgpr-regalloc=basic"
if checked_compile "$@"; then echo "$out: -mllvm -sgpr-regalloc=basic" >> build/fallback_objects.txt; exit 0; fi
echo "== $out: still flagged with the fallback allocator" >&2
rm -f "$out"
exit 1
