// ptk_family_f64.hip -- double precision: the ptk_tree64_* / ptk_search64_* entry points of include/ptk.h
// (ptk_backend_f64.hpp) and the kernels of ptk_kernels_f64.hpp.
// One of the translation units of libptk.so (ptk_backend_core.hpp).

#include "ptk_families.hpp"

#include "ptk_backend_f64.hpp"

namespace {
static __global__ void warm_f64_kernel() {}
}  // namespace

namespace ptkf {
// (loads this unit's code object on the calling thread's device: ProcessWarmup of ptk_backend.hip)
void warm_f64() {
  hipLaunchKernelGGL(warm_f64_kernel, dim3(1), dim3(1), 0, nullptr);
}
}  // namespace ptkf
