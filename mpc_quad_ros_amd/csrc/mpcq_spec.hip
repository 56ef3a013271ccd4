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

namespace mpcq {

template <typename T> using StepFn = void (*)(const DevModel<T>, const DevState<T>, const int);

template <typename T, bool RUN> static StepFn<T> pick(bool gab) {
#ifdef MPCQ_RESOURCE_PROBE   // tools/kernel_resources.sh: only the lockstep instance with the stage records in global memory
  if (gab && !RUN && sizeof(T) == 8) return (StepFn<T>)&step_kernel<Cfg<double, true, MPCQ_SPEC_N, MPCQ_SPEC_NB, false>>;
  return nullptr;
#else
  return gab ? &step_kernel<Cfg<T, true, MPCQ_SPEC_N, MPCQ_SPEC_NB, RUN>> : &step_kernel<Cfg<T, false, MPCQ_SPEC_N, MPCQ_SPEC_NB, RUN>>;
#endif
}

// run = the free-running closed-loop variant (mpcq_sim_run)
StepFn<double> MPCQ_SPEC_NAME(spec_step_f64_, MPCQ_SPEC_N, MPCQ_SPEC_NB)(bool gab, bool run) { return run ? pick<double, true>(gab) : pick<double, false>(gab); }
StepFn<float> MPCQ_SPEC_NAME(spec_step_f32_, MPCQ_SPEC_N, MPCQ_SPEC_NB)(bool gab, bool run) { return run ? pick<float, true>(gab) : pick<float, false>(gab); }

}  // namespace mpcq
