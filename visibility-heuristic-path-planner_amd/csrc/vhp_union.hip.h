// vhp_union.hip.h -- the max-union of a batch of visibility fields and the index of the source that attains it.
//
// The planner's union (reference src/visibilityBasedSolver.cpp:417-418: visibility_global_ = max(visibility_, visibility_global_))
// over a BATCH of fields at once, with the label the planner derives from it -- which source lit the cell best (option
// 2: what a sharded batch has to exchange is one union field and one label field per device, not its fields).  One pass over the
// fields, no temporaries: a thread owns two adjacent cells (16-byte loads of fp64 fields), walks the sources in rising order and
// replaces on strict improvement only, so ties go to the lowest source index -- what a sequential max-union does.  HBM-bound: n
// fields read once (8 or 4 bytes per cell and field), 12 bytes per cell written.
//   vhp_union_fields  : fields k = 0 .. n - 1 of one device, labels first_index + k
//   vhp_union_partials: the same reduction over partial results (a union field AND a label field per part: the parts of several
//                       devices after an all-gather); a tie goes to the lowest LABEL
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vhp {

constexpr int kUnionThreads = 256;
constexpr int32_t kUnionNoSource = 0x7fffffff;  // the label of a cell no field covers (n = 0)

template <typename T>
__global__ void __launch_bounds__(kUnionThreads) vhp_union_fields(const T* __restrict__ fields, int n, long long stride, const int32_t* __restrict__ labels_in,
                                                                    int first_index, long long cells, T* __restrict__ best, int32_t* __restrict__ arg) {
  using T2 = T __attribute__((ext_vector_type(2)));
  const long long pairs = cells >> 1;
  const bool aligned = (stride & 1) == 0 && (reinterpret_cast<uintptr_t>(fields) & (2 * sizeof(T) - 1)) == 0 &&
                       (reinterpret_cast<uintptr_t>(best) & (2 * sizeof(T) - 1)) == 0 && (reinterpret_cast<uintptr_t>(arg) & 7) == 0 &&
                       (!labels_in || (reinterpret_cast<uintptr_t>(labels_in) & 7) == 0);
  const long long step = (long long)gridDim.x * blockDim.x;
  if (aligned) {
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < pairs; p += step) {
      T b0 = (T)-1, b1 = (T)-1;
      int32_t a0 = kUnionNoSource, a1 = kUnionNoSource;
      const T* f = fields + 2 * p;
      const int32_t* l = labels_in ? labels_in + 2 * p : nullptr;
      int k = 0;
      // four fields' loads in flight per thread: the kernel is a stream, its only latency is memory's
      for (; k + 4 <= n; k += 4) {
        T2 v[4];
        int2 lab[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          v[u] = __builtin_nontemporal_load(reinterpret_cast<const T2*>(f + (long long)(k + u) * stride));
          lab[u] = l ? *reinterpret_cast<const int2*>(l + (long long)(k + u) * cells) : make_int2(first_index + k + u, first_index + k + u);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (v[u].x > b0 || (v[u].x == b0 && lab[u].x < a0)) { b0 = v[u].x; a0 = lab[u].x; }
          if (v[u].y > b1 || (v[u].y == b1 && lab[u].y < a1)) { b1 = v[u].y; a1 = lab[u].y; }
        }
      }
      for (; k < n; ++k) {
        const T2 v = __builtin_nontemporal_load(reinterpret_cast<const T2*>(f + (long long)k * stride));
        const int2 lab = l ? *reinterpret_cast<const int2*>(l + (long long)k * cells) : make_int2(first_index + k, first_index + k);
        if (v.x > b0 || (v.x == b0 && lab.x < a0)) { b0 = v.x; a0 = lab.x; }
        if (v.y > b1 || (v.y == b1 && lab.y < a1)) { b1 = v.y; a1 = lab.y; }
      }
      *reinterpret_cast<T2*>(best + 2 * p) = T2{b0, b1};
      *reinterpret_cast<int2*>(arg + 2 * p) = make_int2(a0, a1);
    }
  }
  // single cells: the odd last cell, or everything when some pointer or the stride is off the pair grid
  const long long c0 = aligned ? 2 * pairs : 0;
  for (long long c = c0 + (long long)blockIdx.x * blockDim.x + threadIdx.x; c < cells; c += step) {
    T b = (T)-1;
    int32_t a = kUnionNoSource;
    for (int k = 0; k < n; ++k) {
      const T v = fields[(long long)k * stride + c];
      const int32_t lab = labels_in ? labels_in[(long long)k * cells + c] : first_index + k;
      if (v > b || (v == b && lab < a)) { b = v; a = lab; }
    }
    best[c] = b;
    arg[c] = a;
  }
}

template <typename T>
inline hipError_t launch_union(const void* fields, int n, long long stride, const int32_t* labels_in, int first_index, long long cells, void* best,
                               int32_t* arg, int n_cus, hipStream_t stream) {
  const long long pairs = (cells + 1) / 2;
  long long blocks = (pairs + kUnionThreads - 1) / kUnionThreads;
  const long long cap = (long long)(n_cus > 0 ? n_cus : 256) * 16;  // (a grid-stride loop: sixteen workgroups of 256 per CU fill the chip)
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(vhp_union_fields<T>, dim3((unsigned)blocks), dim3(kUnionThreads), 0, stream, static_cast<const T*>(fields), n, stride, labels_in,
                     first_index, cells, static_cast<T*>(best), arg);
  return hipGetLastError();
}

}  // namespace vhp
