// vhp_multi.hip -- several devices of one node behind the C ABI (include/vhp.h, vhp_multi_*): the sources of a batch are
// block-partitioned over the devices (independent sources, no exchange step in the sweep), every device sweeps its
// shard on its own stream, and -- where the caller wants every field everywhere -- the shards are all-gathered by direct peer
// copies: xGMI is point to point, every pair of devices has its own link, so the all-gather of N shards is N (N - 1) copies that
// all run at once (no ring, no staging).  Host code only: every device-side call goes through the single-device entry points.
// (The torch.distributed / RCCL form of the same sharding is visibility-heuristic-path-planner_amd/dist.py; bench.py uses that one.)
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "vhp.h"

struct vhp_multi {
  std::vector<vhp_ctx*> ctx;
  std::vector<int> device;
  std::vector<hipStream_t> stream;
  std::vector<int32_t*> d_src;
  std::vector<size_t> d_src_cap;
  std::vector<hipEvent_t> done;
  int nx = 0, ny = 0;
  std::string err;
};

namespace {
int mfail(vhp_multi* m, int code, const std::string& msg) { if (m) m->err = msg; return code; }
struct DevGuard {
  int prev = -1;
  explicit DevGuard(int d) { (void)hipGetDevice(&prev); (void)hipSetDevice(d); }
  ~DevGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};
}  // namespace

extern "C" {

void vhp_multi_shard_bounds(int n_src, int n_devices, int d, int* lo, int* hi) {
  // sizes differ by at most one, the larger shards first (dist.py shard_bounds)
  const int base = n_src / n_devices, rem = n_src % n_devices;
  *lo = d * base + (d < rem ? d : rem);
  *hi = *lo + base + (d < rem ? 1 : 0);
}

int vhp_multi_create(const int* device_ordinals, int n_devices, vhp_multi** out) {
  if (!device_ordinals || !out || n_devices < 1 || n_devices > 64) return VHP_ERR_ARG;
  vhp_multi* m = new vhp_multi;
  for (int d = 0; d < n_devices; ++d) {
    vhp_ctx* c = nullptr;
    const int rc = vhp_create(device_ordinals[d], &c);
    if (rc != VHP_OK) { vhp_multi_destroy(m); return rc; }
    DevGuard g(device_ordinals[d]);
    hipStream_t s = nullptr;
    hipEvent_t e = nullptr;
    if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) {
      vhp_destroy(c);
      vhp_multi_destroy(m);
      return VHP_ERR_HIP;
    }
    vhp_set_stream(c, s);
    m->ctx.push_back(c); m->device.push_back(device_ordinals[d]); m->stream.push_back(s); m->done.push_back(e);
    m->d_src.push_back(nullptr); m->d_src_cap.push_back(0);
  }
  // peer access for the gather (a pair that cannot is served through the host by hipMemcpyPeerAsync itself)
  for (int a = 0; a < n_devices; ++a)
    for (int b = 0; b < n_devices; ++b) {
      if (m->device[a] == m->device[b]) continue;
      int can = 0;
      if (hipDeviceCanAccessPeer(&can, m->device[a], m->device[b]) == hipSuccess && can) {
        DevGuard g(m->device[a]);
        const hipError_t e = hipDeviceEnablePeerAccess(m->device[b], 0);
        if (e != hipSuccess) (void)hipGetLastError();  // (already enabled: fine)
      }
    }
  *out = m;
  return VHP_OK;
}

int vhp_multi_destroy(vhp_multi* m) {
  if (!m) return VHP_ERR_ARG;
  for (size_t d = 0; d < m->ctx.size(); ++d) {
    DevGuard g(m->device[d]);
    if (m->d_src[d]) (void)hipFree(m->d_src[d]);
    if (m->done[d]) (void)hipEventDestroy(m->done[d]);
    vhp_destroy(m->ctx[d]);
    if (m->stream[d]) (void)hipStreamDestroy(m->stream[d]);
  }
  delete m;
  return VHP_OK;
}

const char* vhp_multi_last_error(const vhp_multi* m) { return m ? m->err.c_str() : "null handle"; }
int vhp_multi_devices(const vhp_multi* m) { return m ? (int)m->ctx.size() : 0; }
vhp_ctx* vhp_multi_context(vhp_multi* m, int d) { return (m && d >= 0 && d < (int)m->ctx.size()) ? m->ctx[d] : nullptr; }

int vhp_multi_set_map(vhp_multi* m, const uint8_t* occ_rowmajor, int nx, int ny) {
  if (!m || !occ_rowmajor) return VHP_ERR_ARG;
  for (size_t d = 0; d < m->ctx.size(); ++d) {
    const int rc = vhp_set_map(m->ctx[d], occ_rowmajor, nx, ny);
    if (rc != VHP_OK) return mfail(m, rc, std::string("device ") + std::to_string(m->device[d]) + ": " + vhp_last_error(m->ctx[d]));
  }
  m->nx = nx; m->ny = ny;
  return VHP_OK;
}

int vhp_multi_sweep_batch(vhp_multi* m, const int32_t* src_xy, int n_src, int variant, int dtype, void* const* d_out_per_device) {
  if (!m || !src_xy || !d_out_per_device || n_src < 0) return VHP_ERR_ARG;
  if (m->nx == 0) return mfail(m, VHP_ERR_NO_MAP, "vhp_multi_sweep_batch: no map set");
  const int nd = (int)m->ctx.size();
  // every device's shard is enqueued before any is waited for: the sweeps run side by side
  for (int d = 0; d < nd; ++d) {
    int lo, hi;
    vhp_multi_shard_bounds(n_src, nd, d, &lo, &hi);
    if (hi == lo) continue;
    if (!d_out_per_device[d]) return mfail(m, VHP_ERR_ARG, "vhp_multi_sweep_batch: no output buffer for device " + std::to_string(d));
    DevGuard g(m->device[d]);
    const size_t bytes = (size_t)(hi - lo) * 2 * sizeof(int32_t);
    if (m->d_src_cap[d] < bytes) {
      if (m->d_src[d]) (void)hipFree(m->d_src[d]);
      m->d_src[d] = nullptr; m->d_src_cap[d] = 0;
      if (hipMalloc(&m->d_src[d], bytes) != hipSuccess) return mfail(m, VHP_ERR_HIP, "vhp_multi_sweep_batch: out of device memory for the sources");
      m->d_src_cap[d] = bytes;
    }
    if (hipMemcpyAsync(m->d_src[d], src_xy + 2 * (size_t)lo, bytes, hipMemcpyHostToDevice, m->stream[d]) != hipSuccess) return mfail(m, VHP_ERR_HIP, "source upload failed");
    const int rc = vhp_sweep_batch_device(m->ctx[d], m->d_src[d], hi - lo, variant, dtype, d_out_per_device[d]);
    if (rc != VHP_OK) return mfail(m, rc, std::string("device ") + std::to_string(m->device[d]) + ": " + vhp_last_error(m->ctx[d]));
  }
  int worst = VHP_OK;
  for (int d = 0; d < nd; ++d) {
    const int rc = vhp_sync(m->ctx[d]);
    if (rc != VHP_OK && worst == VHP_OK) { worst = rc; m->err = std::string("device ") + std::to_string(m->device[d]) + ": " + vhp_last_error(m->ctx[d]); }
  }
  return worst;
}

int vhp_multi_allgather_fields(vhp_multi* m, int n_src, int dtype, void* const* d_shard_per_device, void* const* d_all_per_device) {
  if (!m || !d_shard_per_device || !d_all_per_device || n_src < 0) return VHP_ERR_ARG;
  if (m->nx == 0) return mfail(m, VHP_ERR_NO_MAP, "vhp_multi_allgather_fields: no map set");
  const int nd = (int)m->ctx.size();
  const size_t field = (size_t)m->nx * m->ny * (dtype == VHP_F64 ? 8 : 4);
  // the shards are final on their own streams (vhp_multi_sweep_batch waited); device `to` pulls shard `from` on its own stream:
  // N (N - 1) copies in flight at once, one per xGMI link and direction, plus the local ones
  for (int to = 0; to < nd; ++to) {
    DevGuard g(m->device[to]);
    for (int k = 0; k < nd; ++k) {
      const int from = (to + k) % nd;  // (every device starts with its own shard, then its neighbours: no two pull from one source first)
      int lo, hi;
      vhp_multi_shard_bounds(n_src, nd, from, &lo, &hi);
      if (hi == lo) continue;
      if (!d_shard_per_device[from] || !d_all_per_device[to]) return mfail(m, VHP_ERR_ARG, "vhp_multi_allgather_fields: missing buffer");
      char* dst = static_cast<char*>(d_all_per_device[to]) + (size_t)lo * field;
      const hipError_t e = hipMemcpyPeerAsync(dst, m->device[to], d_shard_per_device[from], m->device[from], (size_t)(hi - lo) * field, m->stream[to]);
      if (e != hipSuccess) return mfail(m, VHP_ERR_HIP, std::string("peer copy ") + std::to_string(from) + " -> " + std::to_string(to) + ": " + hipGetErrorString(e));
    }
  }
  for (int d = 0; d < nd; ++d) {
    DevGuard g(m->device[d]);
    if (hipStreamSynchronize(m->stream[d]) != hipSuccess) return mfail(m, VHP_ERR_HIP, "vhp_multi_allgather_fields: a copy failed");
  }
  return VHP_OK;
}

}  // extern "C"
