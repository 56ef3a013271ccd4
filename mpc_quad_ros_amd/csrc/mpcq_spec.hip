// Shape-specialised instances of the step kernel (compile-time horizon N and RGP basis size nb): every LDS
// offset, trip count and index division is a constant and the stage loops unroll.  One translation unit per
// shape (-DMPCQ_SPEC_N=.. -DMPCQ_SPEC_NB=.., see the Makefile), built at -O3; the any-shape instances live in
// mpcq_api.hip (see the Makefile for why the two are compiled differently).
// tests/test_gpu_parity.py::test_kernel_variants_agree holds all instances of a precision against each other on
// the device, for every specialised shape.
#include <hip/hip_runtime.h>

#include "mpcq_kernels.hpp"

#ifndef MPCQ_SPEC_N
#define MPCQ_SPEC_N 20   // BASELINE configs[1]
#define MPCQ_SPEC_NB 10
#endif
#define MPCQ_CAT3(a, b, c) a##b##_##c
#define MPCQ_SPEC_NAME(prefix, n, nb) MPCQ_CAT3(prefix, n, nb)

// Two kinds of objects per shape (Makefile): spec_N_NB.o = the LOCKSTEP instances (one launch per control period), specrun_N_NB.o
// (-DMPCQ_SPEC_RUN_ONLY) = the FREE-RUNNING instances (mpcq_sim_run), for the shapes listed in SPEC_RUN_SHAPES.
// Round 3 withdrew the specialised free-running instances after wrong results and a device fault that depended on code
// generation; round 4 found the cause (tools/repro_codegen/README.md): ROCm 7.2's register allocator can put VGPR -> AGPR spill
// copies IN FRONT of the EXEC restore of a control-flow join (behind SGPR spills it placed there first), where they execute for
// no lane.  tools/check_exec_prologue.py detects exactly that in the code object; the Makefile runs it on every object it builds
// and recompiles a flagged translation unit with -mllvm -sgpr-regalloc=basic (SGPR spills at definitions and uses instead of
// block tops: the trigger is gone, at 1-3 % of the speed), checks again and fails the build if that is flagged too.
// tests/test_gpu_parity.py::test_free_running_equals_lockstep_every_instance holds the free-running instances against the
// lockstep launches bit for bit on the device.

namespace mpcq {

template <typename T> using StepFn = void (*)(const DevModel<T>, const DevState<T>, const int);

#ifndef MPCQ_SPEC_RUN_ONLY
template <typename T> static StepFn<T> pick(int layout) {   // layout: mpcq::lds_layout (0 LDS | 1 stage records global | 2 compact)
#ifdef MPCQ_RESOURCE_PROBE   // tools/kernel_resources.sh: only the fp64 lockstep instances with the stage records in global memory
  if (layout == 1 && sizeof(T) == 8) return (StepFn<T>)&step_kernel<Cfg<double, true, MPCQ_SPEC_N, MPCQ_SPEC_NB, false>>;
  if (layout == 2 && sizeof(T) == 8) return (StepFn<T>)&step_kernel<Cfg<double, true, MPCQ_SPEC_N, MPCQ_SPEC_NB, false, true>>;
  return nullptr;
#else
  if (layout == 2) return &step_kernel<Cfg<T, true, MPCQ_SPEC_N, MPCQ_SPEC_NB, false, true>>;
  return layout == 1 ? &step_kernel<Cfg<T, true, MPCQ_SPEC_N, MPCQ_SPEC_NB, false>> : &step_kernel<Cfg<T, false, MPCQ_SPEC_N, MPCQ_SPEC_NB, false>>;
#endif
}

StepFn<double> MPCQ_SPEC_NAME(spec_lock_f64_, MPCQ_SPEC_N, MPCQ_SPEC_NB)(int layout) { return pick<double>(layout); }
StepFn<float> MPCQ_SPEC_NAME(spec_lock_f32_, MPCQ_SPEC_N, MPCQ_SPEC_NB)(int layout) { return pick<float>(layout); }

#endif   // !MPCQ_SPEC_RUN_ONLY

#if defined(MPCQ_SPEC_RUN) || defined(MPCQ_SPEC_RUN_ONLY)   // shape-specialised FREE-RUNNING instances (see the note at the top)
template <typename T> static StepFn<T> pick_run(int layout) {
  if (layout == 2) return &step_kernel<Cfg<T, true, MPCQ_SPEC_N, MPCQ_SPEC_NB, true, true>>;
  return layout == 1 ? &step_kernel<Cfg<T, true, MPCQ_SPEC_N, MPCQ_SPEC_NB, true>> : &step_kernel<Cfg<T, false, MPCQ_SPEC_N, MPCQ_SPEC_NB, true>>;
}
StepFn<double> MPCQ_SPEC_NAME(spec_run_f64_, MPCQ_SPEC_N, MPCQ_SPEC_NB)(int layout) { return pick_run<double>(layout); }
StepFn<float> MPCQ_SPEC_NAME(spec_run_f32_, MPCQ_SPEC_N, MPCQ_SPEC_NB)(int layout) { return pick_run<float>(layout); }
#endif

}  // namespace mpcq
