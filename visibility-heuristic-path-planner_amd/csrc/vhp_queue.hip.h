// vhp_queue.hip.h -- computeVisibilityUsingQueue() on the device.
//
// Replaces reference src/visibilityBasedSolver.cpp:701-893.  That routine is a FIFO
// flood whose result depends on the exact pop order (a cell is computed at
// its first pop from whatever its upstream neighbours hold at that moment), so it
// cannot be restated as a data-parallel sweep without changing results.  To stay
// bit-exact this kernel emulates the queue literally: one workgroup per source, lane 0
// runs the flood over a queue in HBM (capacity 3*cells + 8: a cell is computed once
// and pushes at most three successors), the other lanes only help with the final
// copy-out.  It is latency bound by construction (it is dead code in the reference,
// its only call site is commented out at solver.cpp:219) and is provided for parity
// of the API surface, not for throughput; sources of a batch still run concurrently.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "vhp.h"
#include "vhp_sweep.hip.h"

namespace vhp {

struct QCell { int x, y; };

template <typename OutT>
__global__ void vhp_queue_flood(int nx, int ny, const uint8_t* __restrict__ occ, const int32_t* __restrict__ src_xy,
                                double* __restrict__ work_all, QCell* __restrict__ queue_all, uint8_t* __restrict__ visited_all,
                                OutT* __restrict__ out_all, long long cells, long long qcap, int* __restrict__ err_flag) {
  const int s = blockIdx.x;
  const int sx = src_xy[2 * s], sy = src_xy[2 * s + 1];
  double* vis = work_all + (size_t)s * cells;
  OutT* out = out_all + (size_t)s * cells;
  bool bad = sx < 0 || sy < 0 || sx >= nx || sy >= ny;
  if (threadIdx.x == 0 && !bad) {
    QCell* q = queue_all + (size_t)s * qcap;
    uint8_t* visited = visited_all + (size_t)s * cells;
    long long head = 0, tail = 0;
    auto at = [&](long long x, long long y) { return (size_t)x + (size_t)y * nx; };
    auto push = [&](long long x, long long y) { q[tail].x = (int)x; q[tail].y = (int)y; ++tail; };
    vis[at(sx, sy)] = 1.0;  // :708
    // seed order of :710-717
    push(sx + 1, sy); push(sx, sy + 1); push(sx - 1, sy); push(sx, sy - 1);
    push(sx + 1, sy + 1); push(sx - 1, sy + 1); push(sx - 1, sy - 1); push(sx + 1, sy - 1);
    while (head < tail) {
      const long long x = q[head].x, y = q[head].y;
      ++head;
      if (x < 0 || y < 0 || x >= nx || y >= ny) continue;  // isValid, :726
      const size_t k = at(x, y);
      if (visited[k]) continue;                            // :729
      if (occ[k] == 0) continue;                           // :732
      const int dx = (int)(x - sx), dy = (int)(y - sy);
      const int qx = dx >= 0 ? 1 : -1, qy = dy >= 0 ? 1 : -1;  // the four arms :739,:777,:815,:853
      const long long xb = x - qx, yb = y - qy;             // one step back toward the source
      const int adx = dx >= 0 ? dx : -dx, ady = dy >= 0 ? dy : -dy;
      double v;
      if (dx == 0) {
        v = vis[at(x, yb)];
        if (v > 0.001) push(x, y + qy);
      } else if (dy == 0) {
        v = vis[at(xb, y)];
        if (v > 0.001) push(x + qx, y);
      } else if (adx == ady) {
        v = vis[at(xb, yb)];
        if (v > 0.001) { push(x, y + qy); push(x + qx, y); push(x + qx, y + qy); }
      } else if (adx > ady) {
        const double c = (double)ady / (double)adx;  // IEEE division, as the reference
        const double a = vis[at(xb, y)];
        v = stencil(a, vis[at(xb, yb)], c);
        if (v > 0.001) { push(x, y + qy); push(x + qx, y); }
      } else {
        const double c = (double)adx / (double)ady;
        const double a = vis[at(x, yb)];
        v = stencil(a, vis[at(xb, yb)], c);
        if (v > 0.001) { push(x + qx, y); push(x, y + qy); }
      }
      // free cell: occupancy factor is 1 (blocked cells never reach this point)
      vis[k] = v;
      visited[k] = 1;
    }
  }
  if (bad && threadIdx.x == 0) atomicOr(err_flag, 1);
  __syncthreads();
  __threadfence_block();
  for (long long k = threadIdx.x; k < cells; k += blockDim.x) out[k] = bad ? OutT(0) : static_cast<OutT>(vis[k]);
}

// raycasting(), reference solver.cpp:267-290: one thread per target cell
__global__ void vhp_raycast(int nx, int ny, const uint8_t* __restrict__ occ, int sx, int sy, double* __restrict__ ray) {
  const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= (long long)nx * ny) return;
  const int x1 = (int)(k % nx), y1 = (int)(k / nx);
  int x0 = sx, y0 = sy;
  const int dx = abs(x1 - x0), dy = abs(y1 - y0);
  const int stepx = x0 < x1 ? 1 : -1, stepy = y0 < y1 ? 1 : -1;
  int err = dx - dy;
  while (x0 != x1 || y0 != y1) {
    if (occ[(size_t)x0 + (size_t)y0 * nx] == 0) {
      ray[(size_t)x0 + (size_t)y0 * nx] = 0.0;
      ray[(size_t)x1 + (size_t)y1 * nx] = 0.0;
      return;
    }
    const int e2 = 2 * err;
    if (e2 > -dy) { err -= dy; x0 += stepx; }
    if (e2 < dx) { err += dx; y0 += stepy; }
  }
}

__global__ void vhp_fill_f64(double* __restrict__ p, double v, size_t n) {
  const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n) p[k] = v;
}

struct QueueScratch {
  double* work = nullptr;
  QCell* queue = nullptr;
  uint8_t* visited = nullptr;
  size_t slice = 0, cells = 0;
};

inline void queue_scratch_free(QueueScratch& q) {
  if (q.work) (void)hipFree(q.work);
  if (q.queue) (void)hipFree(q.queue);
  if (q.visited) (void)hipFree(q.visited);
  q = QueueScratch();
}

inline hipError_t launch_queue_sweep_impl(QueueScratch& qs, const DevMap& m, const uint8_t* d_occ, const int32_t* d_src,
                                          int n_src, int dtype, void* d_out, int* d_err, hipStream_t stream) {
  const size_t cells = (size_t)m.nx * m.ny;
  const size_t qcap = 3 * cells + 8;
  // bound scratch to ~1 GiB: 8 (work) + 8*3 (queue) + 1 (visited) bytes per cell and source
  const size_t slice = std::max<size_t>(1, std::min<size_t>((size_t)n_src, ((size_t)1 << 30) / (cells * 33 + 64)));
  if (qs.slice < slice || qs.cells != cells) {
    queue_scratch_free(qs);
    hipError_t e;
    if ((e = hipMalloc(&qs.work, slice * cells * 8)) != hipSuccess) return e;
    if ((e = hipMalloc(&qs.queue, slice * qcap * sizeof(QCell))) != hipSuccess) return e;
    if ((e = hipMalloc(&qs.visited, slice * cells)) != hipSuccess) return e;
    qs.slice = slice;
    qs.cells = cells;
  }
  const size_t esz = dtype == VHP_F64 ? 8 : 4;
  for (int s0 = 0; s0 < n_src; s0 += (int)slice) {
    const int n = std::min<int>((int)slice, n_src - s0);
    hipError_t e;
    if ((e = hipMemsetAsync(qs.work, 0, (size_t)n * cells * 8, stream)) != hipSuccess) return e;
    if ((e = hipMemsetAsync(qs.visited, 0, (size_t)n * cells, stream)) != hipSuccess) return e;
    void* o = static_cast<char*>(d_out) + (size_t)s0 * cells * esz;
    if (dtype == VHP_F64)
      hipLaunchKernelGGL(vhp_queue_flood<double>, dim3(n), dim3(256), 0, stream, m.nx, m.ny, d_occ, d_src + 2 * (size_t)s0,
                         qs.work, qs.queue, qs.visited, static_cast<double*>(o), (long long)cells, (long long)qcap, d_err);
    else
      hipLaunchKernelGGL(vhp_queue_flood<float>, dim3(n), dim3(256), 0, stream, m.nx, m.ny, d_occ, d_src + 2 * (size_t)s0,
                         qs.work, qs.queue, qs.visited, static_cast<float*>(o), (long long)cells, (long long)qcap, d_err);
    if ((e = hipGetLastError()) != hipSuccess) return e;
  }
  return hipSuccess;
}

}  // namespace vhp
