// placeholder while the sweep is brought up on hardware; replaced by the real planner
#pragma once
#include <hip/hip_runtime.h>
#include <string>
#include "vhp_sweep.cuh"
namespace vhp {
struct PlannerState { int dummy = 0; };
inline void planner_free(PlannerState&) {}
inline int planner_solve(PlannerState&, const DevMap&, const uint8_t*, hipStream_t, hipEvent_t, hipEvent_t, int, int, int, int,
                         double, uint64_t, uint64_t*, double*, double*, int32_t*, uint32_t*, std::string* msg) {
  *msg = "planner not built yet";
  return 100;
}
inline hipError_t launch_queue_sweep(const DevMap&, const uint8_t*, const int32_t*, int, int, void*, int*, hipStream_t) {
  return hipErrorNotSupported;
}
}  // namespace vhp
