// Shape-specialised instances of the step kernel (compile-time horizon N and RGP basis size nb): every LDS
// offset, trip count and index division is a constant and the stage loops unroll.  Built as its own
// translation unit at -O3; the any-shape instances live in mpcq_api.hip (see the Makefile for why the two
// are compiled differently).  tests/test_gpu_parity.py::test_kernel_variants_agree holds all instances of
// a precision against each other on the device.
#include <hip/hip_runtime.h>

#include "mpcq_kernels.hpp"

namespace mpcq {

template <typename T> using StepFn = void (*)(const DevModel<T>, const DevState<T>, const int);

template <typename T, bool RUN> static StepFn<T> pick(int N, int nb, bool gab) {
#ifdef MPCQ_RESOURCE_PROBE   // tools/kernel_resources.sh --probe: only the benchmark instance (fp64, global stage records, lockstep)
  if (N == 20 && nb == 10 && gab && !RUN && sizeof(T) == 8) return (StepFn<T>)&step_kernel<Cfg<double, true, 20, 10, false>>;
  return nullptr;
#else
  if (N == 20 && nb == 10)   // BASELINE configs[1]
    return gab ? &step_kernel<Cfg<T, true, 20, 10, RUN>> : &step_kernel<Cfg<T, false, 20, 10, RUN>>;
  return nullptr;
#endif
}

// run = the free-running closed-loop variant (mpcq_sim_run)
StepFn<double> spec_step_f64(int N, int nb, bool gab, bool run) { return run ? pick<double, true>(N, nb, gab) : pick<double, false>(N, nb, gab); }
StepFn<float> spec_step_f32(int N, int nb, bool gab, bool run) { return run ? pick<float, true>(N, nb, gab) : pick<float, false>(N, nb, gab); }

}  // namespace mpcq
