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

// LOCKSTEP instances only (one launch per control period).  The free-running launches (mpcq_sim_run) use the any-shape instance
// of mpcq_api.hip for every shape: the shape-specialised free-running kernels (420-430 registers, ~430 SGPR spills at -O3 with the
// unrolled stage loops) gave code-generation-dependent results and, for shape (20, 20), a device fault in round 3, while the same
// source passes every index / EXEC check of the checked build and the any-shape instance reproduces the lockstep launches bit
// for bit in every configuration tried (tests/test_gpu_parity.py::test_free_running_equals_lockstep_every_instance).

namespace mpcq {

template <typename T> using StepFn = void (*)(const DevModel<T>, const DevState<T>, const int);

template <typename T> static StepFn<T> pick(bool gab) {
#ifdef MPCQ_RESOURCE_PROBE   // tools/kernel_resources.sh: only the lockstep instance with the stage records in global memory
  if (gab && sizeof(T) == 8) return (StepFn<T>)&step_kernel<Cfg<double, true, MPCQ_SPEC_N, MPCQ_SPEC_NB, false>>;
  return nullptr;
#else
  return gab ? &step_kernel<Cfg<T, true, MPCQ_SPEC_N, MPCQ_SPEC_NB, false>> : &step_kernel<Cfg<T, false, MPCQ_SPEC_N, MPCQ_SPEC_NB, false>>;
#endif
}

StepFn<double> MPCQ_SPEC_NAME(spec_lock_f64_, MPCQ_SPEC_N, MPCQ_SPEC_NB)(bool gab) { return pick<double>(gab); }
StepFn<float> MPCQ_SPEC_NAME(spec_lock_f32_, MPCQ_SPEC_N, MPCQ_SPEC_NB)(bool gab) { return pick<float>(gab); }

}  // namespace mpcq
