// vhp_multi.hip -- several devices of one node behind the C ABI (include/vhp.h, vhp_multi_*): the sources of a batch are
// block-partitioned over the devices (independent sources, no exchange step in the sweep), every device sweeps its
// shard on its own stream, and -- where the caller wants every field everywhere -- the shards are all-gathered.
//
// The all-gather, two ways:
//   * direct peer copies (the default).  xGMI is point to point: every pair of devices has a link of its own, so the N (N - 1)
//     copies of an all-gather can all run at once -- IF no two of them wait for each other.  Copies enqueued on one stream run
//     one after the other, so every (destination, source) pair has a stream of its own ("lane" k of device `to` carries the copy
//     from device (to + k) mod N): the N - 1 inbound copies of a device are in flight together, one per link, and in round k no
//     two destinations pull from the same source.  vhp_multi_allgather_plan is that enqueue plan as data (host arithmetic only: the
//     CPU tests check it without a device).
//   * RCCL (vhp_multi_use_rccl): ncclAllGather on a communicator over the listed devices, one group call -- the collective
//     north_star names; equal shards only (a batch that does not divide takes the peer copies), distinct devices only.  librccl
//     is loaded at run time (dlopen), the library does not link against it.
// Host code only: every device-side call goes through the single-device entry points.
// (The torch.distributed / RCCL form of the same sharding is visibility-heuristic-path-planner_amd/dist.py; bench.py uses that one.)
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "vhp.h"

namespace {
// the few RCCL entry points used, by their C signatures (rccl.h: ncclResult_t is an enum, 0 = success; ncclComm_t an opaque pointer)
struct Rccl {
  void* lib = nullptr;
  int (*CommInitAll)(void** comms, int ndev, const int* devlist) = nullptr;
  int (*CommDestroy)(void* comm) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  int (*AllGather)(const void* send, void* recv, size_t count, int datatype, void* comm, hipStream_t stream) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  std::string why;  // what load() failed on (dlerror() hands its message out once: taken right after the failing call)
  bool load() {
    if (lib) return true;
    why.clear();
    for (const char* name : {"librccl.so.1", "librccl.so"}) {
      lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (lib) break;
      const char* de = dlerror();
      if (!why.empty()) why += "; ";
      why += de ? de : "dlopen failed";
    }
    if (!lib) return false;
    why.clear();
    CommInitAll = reinterpret_cast<decltype(CommInitAll)>(dlsym(lib, "ncclCommInitAll"));
    CommDestroy = reinterpret_cast<decltype(CommDestroy)>(dlsym(lib, "ncclCommDestroy"));
    GroupStart = reinterpret_cast<decltype(GroupStart)>(dlsym(lib, "ncclGroupStart"));
    GroupEnd = reinterpret_cast<decltype(GroupEnd)>(dlsym(lib, "ncclGroupEnd"));
    AllGather = reinterpret_cast<decltype(AllGather)>(dlsym(lib, "ncclAllGather"));
    GetErrorString = reinterpret_cast<decltype(GetErrorString)>(dlsym(lib, "ncclGetErrorString"));
    if (CommInitAll && CommDestroy && GroupStart && GroupEnd && AllGather) return true;
    // a library without one of the entry points is no library at all: the next call must not find `lib` set and call through a null pointer
    why = "an entry point is missing (ncclCommInitAll / ncclCommDestroy / ncclGroupStart / ncclGroupEnd / ncclAllGather)";
    CommInitAll = nullptr; CommDestroy = nullptr; GroupStart = nullptr; GroupEnd = nullptr; AllGather = nullptr; GetErrorString = nullptr;
    dlclose(lib);
    lib = nullptr;
    return false;
  }
};
constexpr int kNcclInt32 = 2, kNcclFloat32 = 7, kNcclFloat64 = 8;  // ncclDataType_t (rccl.h)
}  // namespace

struct vhp_multi {
  std::vector<vhp_ctx*> ctx;
  std::vector<int> device;
  std::vector<hipStream_t> stream;
  std::vector<std::vector<hipStream_t>> lane;  // lane[to][k]: the stream of the copy from device (to + k) mod N to device `to`
  std::vector<int32_t*> d_src;
  std::vector<size_t> d_src_cap;
  std::vector<hipEvent_t> done;
  int nx = 0, ny = 0;
  std::string err;
  Rccl rccl;
  std::vector<void*> comms;  // one communicator per device when the RCCL path is on
  // vhp_multi_union_fields: every device's N partial unions and their label fields (its own partial in slot d)
  std::vector<void*> parts_best;
  std::vector<int32_t*> parts_arg;
  std::vector<size_t> parts_cap;   // bytes of parts_best[d] (parts_arg[d] holds the matching N * cells int32)
};

namespace {
int mfail(vhp_multi* m, int code, const std::string& msg) { if (m) m->err = msg; return code; }
struct DevGuard {
  int prev = -1;
  explicit DevGuard(int d) { (void)hipGetDevice(&prev); (void)hipSetDevice(d); }
  ~DevGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};
bool bad_dtype(int dtype) { return dtype != VHP_F64 && dtype != VHP_F32; }
// waits for whatever devices 0 .. upto - 1 still have in flight (an error return must not leave sweeps running into the caller's buffers)
void drain(vhp_multi* m, int upto) {
  for (int d = 0; d < upto; ++d) {
    DevGuard g(m->device[d]);
    (void)hipStreamSynchronize(m->stream[d]);
    for (hipStream_t s : m->lane[d]) (void)hipStreamSynchronize(s);
  }
}
}  // namespace

extern "C" {

void vhp_multi_shard_bounds(int n_src, int n_devices, int d, int* lo, int* hi) {
  if (!lo || !hi) return;
  if (n_devices < 1 || d < 0 || d >= n_devices || n_src < 0) { *lo = *hi = 0; return; }
  // sizes differ by at most one, the larger shards first (dist.py shard_bounds)
  const int base = n_src / n_devices, rem = n_src % n_devices;
  *lo = d * base + (d < rem ? d : rem);
  *hi = *lo + base + (d < rem ? 1 : 0);
}

int vhp_multi_allgather_plan(int n_src, int n_devices, int* to, int* from, int* lane, int* lo, int* hi, int cap) {
  if (n_devices < 1 || n_src < 0) return 0;
  int n = 0;
  // round k: device `to` pulls the shard of device (to + k) mod N on its lane k.  Round 0 is the local copy; in every round the
  // sources are a permutation of the devices, so no two destinations pull from one source at once.
  for (int k = 0; k < n_devices; ++k)
    for (int t = 0; t < n_devices; ++t) {
      const int f = (t + k) % n_devices;
      int a, b;
      vhp_multi_shard_bounds(n_src, n_devices, f, &a, &b);
      if (a == b) continue;
      if (n < cap) {
        if (to) to[n] = t;
        if (from) from[n] = f;
        if (lane) lane[n] = k;
        if (lo) lo[n] = a;
        if (hi) hi[n] = b;
      }
      ++n;
    }
  return n;
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
    std::vector<hipStream_t> lanes((size_t)n_devices, nullptr);
    bool ok = hipStreamCreateWithFlags(&s, hipStreamNonBlocking) == hipSuccess && hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess;
    for (int k = 0; ok && k < n_devices; ++k) ok = hipStreamCreateWithFlags(&lanes[k], hipStreamNonBlocking) == hipSuccess;
    if (!ok) {
      for (hipStream_t l : lanes) if (l) (void)hipStreamDestroy(l);
      if (e) (void)hipEventDestroy(e);
      if (s) (void)hipStreamDestroy(s);
      vhp_destroy(c);
      vhp_multi_destroy(m);
      return VHP_ERR_HIP;
    }
    vhp_set_stream(c, s);
    m->ctx.push_back(c); m->device.push_back(device_ordinals[d]); m->stream.push_back(s); m->done.push_back(e);
    m->lane.push_back(lanes);
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
  if (!m->comms.empty() && m->rccl.CommDestroy)
    for (void* c : m->comms) if (c) (void)m->rccl.CommDestroy(c);
  for (size_t d = 0; d < m->ctx.size(); ++d) {
    DevGuard g(m->device[d]);
    if (m->d_src[d]) (void)hipFree(m->d_src[d]);
    if (d < m->parts_best.size() && m->parts_best[d]) (void)hipFree(m->parts_best[d]);
    if (d < m->parts_arg.size() && m->parts_arg[d]) (void)hipFree(m->parts_arg[d]);
    if (m->done[d]) (void)hipEventDestroy(m->done[d]);
    vhp_destroy(m->ctx[d]);
    for (hipStream_t l : m->lane[d]) if (l) (void)hipStreamDestroy(l);
    if (m->stream[d]) (void)hipStreamDestroy(m->stream[d]);
  }
  delete m;
  return VHP_OK;
}

const char* vhp_multi_last_error(const vhp_multi* m) { return m ? m->err.c_str() : "null handle"; }
int vhp_multi_devices(const vhp_multi* m) { return m ? (int)m->ctx.size() : 0; }
vhp_ctx* vhp_multi_context(vhp_multi* m, int d) { return (m && d >= 0 && d < (int)m->ctx.size()) ? m->ctx[d] : nullptr; }

int vhp_multi_use_rccl(vhp_multi* m, int enable) {
  if (!m) return VHP_ERR_ARG;
  if (!enable) {
    if (m->rccl.CommDestroy) for (void* c : m->comms) if (c) (void)m->rccl.CommDestroy(c);
    m->comms.clear();
    return VHP_OK;
  }
  if (!m->comms.empty()) return VHP_OK;
  const int nd = (int)m->ctx.size();
  for (int a = 0; a < nd; ++a)
    for (int b = a + 1; b < nd; ++b)
      if (m->device[a] == m->device[b]) return mfail(m, VHP_ERR_ARG, "vhp_multi_use_rccl: a communicator needs distinct devices (an ordinal is listed twice)");
  if (!m->rccl.load()) return mfail(m, VHP_ERR_HIP, std::string("vhp_multi_use_rccl: librccl could not be loaded: ") + m->rccl.why);
  m->comms.assign((size_t)nd, nullptr);
  const int rc = m->rccl.CommInitAll(m->comms.data(), nd, m->device.data());
  if (rc != 0) {
    m->comms.clear();
    return mfail(m, VHP_ERR_HIP, std::string("ncclCommInitAll: ") + (m->rccl.GetErrorString ? m->rccl.GetErrorString(rc) : std::to_string(rc)));
  }
  return VHP_OK;
}

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
  if (bad_dtype(dtype)) return mfail(m, VHP_ERR_ARG, "vhp_multi_sweep_batch: bad dtype");
  if (variant != VHP_SWEEP_FULL && variant != VHP_SWEEP_QUEUE) return mfail(m, VHP_ERR_ARG, "vhp_multi_sweep_batch: bad variant");
  const int nd = (int)m->ctx.size();
  for (int d = 0; d < nd; ++d) {  // (every buffer checked before anything is launched)
    int lo, hi;
    vhp_multi_shard_bounds(n_src, nd, d, &lo, &hi);
    if (hi > lo && !d_out_per_device[d]) return mfail(m, VHP_ERR_ARG, "vhp_multi_sweep_batch: no output buffer for device " + std::to_string(d));
  }
  // every device's shard is enqueued before any is waited for: the sweeps run side by side.  An error on device d is returned only
  // after the sweeps already enqueued on devices 0 .. d - 1 have finished: nothing of this call runs on after it has returned.
  for (int d = 0; d < nd; ++d) {
    int lo, hi;
    vhp_multi_shard_bounds(n_src, nd, d, &lo, &hi);
    if (hi == lo) continue;
    DevGuard g(m->device[d]);
    const size_t bytes = (size_t)(hi - lo) * 2 * sizeof(int32_t);
    if (m->d_src_cap[d] < bytes) {
      if (m->d_src[d]) (void)hipFree(m->d_src[d]);
      m->d_src[d] = nullptr; m->d_src_cap[d] = 0;
      if (hipMalloc(&m->d_src[d], bytes) != hipSuccess) { (void)hipGetLastError(); drain(m, d); return mfail(m, VHP_ERR_HIP, "vhp_multi_sweep_batch: out of device memory for the sources"); }
      m->d_src_cap[d] = bytes;
    }
    if (hipMemcpyAsync(m->d_src[d], src_xy + 2 * (size_t)lo, bytes, hipMemcpyHostToDevice, m->stream[d]) != hipSuccess) { drain(m, d + 1); return mfail(m, VHP_ERR_HIP, "source upload failed"); }
    const int rc = vhp_sweep_batch_device(m->ctx[d], m->d_src[d], hi - lo, variant, dtype, d_out_per_device[d]);
    if (rc != VHP_OK) { drain(m, d + 1); return mfail(m, rc, std::string("device ") + std::to_string(m->device[d]) + ": " + vhp_last_error(m->ctx[d])); }
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
  if (bad_dtype(dtype)) return mfail(m, VHP_ERR_ARG, "vhp_multi_allgather_fields: bad dtype");
  const int nd = (int)m->ctx.size();
  const size_t cells = (size_t)m->nx * m->ny;
  const size_t field = cells * (dtype == VHP_F64 ? 8 : 4);
  for (int d = 0; d < nd; ++d) {  // (every buffer checked before a copy is enqueued)
    int lo, hi;
    vhp_multi_shard_bounds(n_src, nd, d, &lo, &hi);
    if ((hi > lo && !d_shard_per_device[d]) || (n_src > 0 && !d_all_per_device[d])) return mfail(m, VHP_ERR_ARG, "vhp_multi_allgather_fields: missing buffer");
  }
  if (n_src == 0) return VHP_OK;
  // the shards are final on their own streams (vhp_multi_sweep_batch waited for them)
  if (!m->comms.empty() && n_src % nd == 0) {
    // RCCL: one ncclAllGather per device inside one group call (a single-process communicator over the listed devices)
    const size_t count = (size_t)(n_src / nd) * cells;
    int rc = m->rccl.GroupStart();
    for (int d = 0; rc == 0 && d < nd; ++d) {
      DevGuard g(m->device[d]);
      rc = m->rccl.AllGather(d_shard_per_device[d], d_all_per_device[d], count, dtype == VHP_F64 ? kNcclFloat64 : kNcclFloat32, m->comms[d], m->stream[d]);
    }
    const int rc_end = m->rccl.GroupEnd();
    if (rc == 0) rc = rc_end;
    drain(m, nd);
    if (rc != 0) return mfail(m, VHP_ERR_HIP, std::string("ncclAllGather: ") + (m->rccl.GetErrorString ? m->rccl.GetErrorString(rc) : std::to_string(rc)));
    return VHP_OK;
  }
  // peer copies, by the plan: the copy (to <- from) on lane (from - to) mod N of device `to`, all N (N - 1) + N of them in flight at once
  const int n_plan = vhp_multi_allgather_plan(n_src, nd, nullptr, nullptr, nullptr, nullptr, nullptr, 0);
  std::vector<int> to((size_t)n_plan), from((size_t)n_plan), lane((size_t)n_plan), lo((size_t)n_plan), hi((size_t)n_plan);
  (void)vhp_multi_allgather_plan(n_src, nd, to.data(), from.data(), lane.data(), lo.data(), hi.data(), n_plan);
  for (int k = 0; k < n_plan; ++k) {
    DevGuard g(m->device[to[k]]);
    char* dst = static_cast<char*>(d_all_per_device[to[k]]) + (size_t)lo[k] * field;
    const hipError_t e = hipMemcpyPeerAsync(dst, m->device[to[k]], d_shard_per_device[from[k]], m->device[from[k]], (size_t)(hi[k] - lo[k]) * field,
                                            m->lane[to[k]][lane[k]]);
    if (e != hipSuccess) {
      drain(m, nd);
      return mfail(m, VHP_ERR_HIP, std::string("peer copy ") + std::to_string(from[k]) + " -> " + std::to_string(to[k]) + ": " + hipGetErrorString(e));
    }
  }
  for (int d = 0; d < nd; ++d) {
    DevGuard g(m->device[d]);
    for (hipStream_t s : m->lane[d])
      if (hipStreamSynchronize(s) != hipSuccess) { drain(m, nd); return mfail(m, VHP_ERR_HIP, "vhp_multi_allgather_fields: a copy failed"); }
  }
  return VHP_OK;
}

int vhp_multi_union_fields(vhp_multi* m, int n_src, int dtype, void* const* d_shard_per_device, void* const* d_best_per_device,
                           int32_t* const* d_arg_per_device) {
  if (!m || !d_shard_per_device || !d_best_per_device || !d_arg_per_device || n_src < 0) return VHP_ERR_ARG;
  if (m->nx == 0) return mfail(m, VHP_ERR_NO_MAP, "vhp_multi_union_fields: no map set");
  if (bad_dtype(dtype)) return mfail(m, VHP_ERR_ARG, "vhp_multi_union_fields: bad dtype");
  const int nd = (int)m->ctx.size();
  const size_t cells = (size_t)m->nx * m->ny, el = dtype == VHP_F64 ? 8 : 4;
  for (int d = 0; d < nd; ++d) {
    int lo, hi;
    vhp_multi_shard_bounds(n_src, nd, d, &lo, &hi);
    if ((hi > lo && !d_shard_per_device[d]) || !d_best_per_device[d] || !d_arg_per_device[d]) return mfail(m, VHP_ERR_ARG, "vhp_multi_union_fields: missing buffer");
  }
  m->parts_best.resize((size_t)nd, nullptr);
  m->parts_arg.resize((size_t)nd, nullptr);
  m->parts_cap.resize((size_t)nd, 0);
  // 1. every device reduces its own shard into slot d of its table of partials (an empty shard: the neutral partial)
  for (int d = 0; d < nd; ++d) {
    DevGuard g(m->device[d]);
    if (m->parts_cap[d] < (size_t)nd * cells * el) {
      if (m->parts_best[d]) (void)hipFree(m->parts_best[d]);
      if (m->parts_arg[d]) (void)hipFree(m->parts_arg[d]);
      m->parts_best[d] = nullptr; m->parts_arg[d] = nullptr; m->parts_cap[d] = 0;
      if (hipMalloc(&m->parts_best[d], (size_t)nd * cells * el) != hipSuccess || hipMalloc(&m->parts_arg[d], (size_t)nd * cells * 4) != hipSuccess) {
        (void)hipGetLastError();
        drain(m, d);
        return mfail(m, VHP_ERR_HIP, "vhp_multi_union_fields: out of device memory for the partial unions");
      }
      m->parts_cap[d] = (size_t)nd * cells * el;
    }
    int lo, hi;
    vhp_multi_shard_bounds(n_src, nd, d, &lo, &hi);
    const int rc = vhp_union_fields_device(m->ctx[d], d_shard_per_device[d], hi - lo, dtype, lo, static_cast<char*>(m->parts_best[d]) + (size_t)d * cells * el,
                                           m->parts_arg[d] + (size_t)d * cells);
    if (rc != VHP_OK) { drain(m, d + 1); return mfail(m, rc, std::string("device ") + std::to_string(m->device[d]) + ": " + vhp_last_error(m->ctx[d])); }
  }
  for (int d = 0; d < nd; ++d) {
    const int rc = vhp_sync(m->ctx[d]);
    if (rc != VHP_OK) { drain(m, nd); return mfail(m, rc, std::string("device ") + std::to_string(m->device[d]) + ": " + vhp_last_error(m->ctx[d])); }
  }
  // 2. the partials travel: one union field and one label field per device, whatever the batch (option 2)
  if (!m->comms.empty()) {
    int rc = m->rccl.GroupStart();
    for (int d = 0; rc == 0 && d < nd; ++d) {
      DevGuard g(m->device[d]);
      rc = m->rccl.AllGather(static_cast<char*>(m->parts_best[d]) + (size_t)d * cells * el, m->parts_best[d], cells, dtype == VHP_F64 ? kNcclFloat64 : kNcclFloat32,
                             m->comms[d], m->stream[d]);
      if (rc == 0) rc = m->rccl.AllGather(m->parts_arg[d] + (size_t)d * cells, m->parts_arg[d], cells, kNcclInt32, m->comms[d], m->stream[d]);
    }
    const int rc_end = m->rccl.GroupEnd();
    if (rc == 0) rc = rc_end;
    drain(m, nd);
    if (rc != 0) return mfail(m, VHP_ERR_HIP, std::string("ncclAllGather (partial unions): ") + (m->rccl.GetErrorString ? m->rccl.GetErrorString(rc) : std::to_string(rc)));
  } else {
    for (int k = 1; k < nd; ++k)       // round k: device `to` pulls slot (to + k) mod N from its owner, on its lane k (see vhp_multi_allgather_plan)
      for (int to = 0; to < nd; ++to) {
        const int from = (to + k) % nd;
        DevGuard g(m->device[to]);
        hipError_t e = hipMemcpyPeerAsync(static_cast<char*>(m->parts_best[to]) + (size_t)from * cells * el, m->device[to],
                                          static_cast<char*>(m->parts_best[from]) + (size_t)from * cells * el, m->device[from], cells * el, m->lane[to][k]);
        if (e == hipSuccess)
          e = hipMemcpyPeerAsync(m->parts_arg[to] + (size_t)from * cells, m->device[to], m->parts_arg[from] + (size_t)from * cells, m->device[from], cells * 4, m->lane[to][k]);
        if (e != hipSuccess) { drain(m, nd); return mfail(m, VHP_ERR_HIP, std::string("peer copy of a partial union: ") + hipGetErrorString(e)); }
      }
    for (int d = 0; d < nd; ++d) {
      DevGuard g(m->device[d]);
      for (hipStream_t s : m->lane[d])
        if (hipStreamSynchronize(s) != hipSuccess) { drain(m, nd); return mfail(m, VHP_ERR_HIP, "vhp_multi_union_fields: a copy failed"); }
    }
  }
  // 3. every device merges the N partials (a tie goes to the lowest label: the lowest source index)
  for (int d = 0; d < nd; ++d) {
    const int rc = vhp_union_partials_device(m->ctx[d], m->parts_best[d], m->parts_arg[d], nd, dtype, d_best_per_device[d], d_arg_per_device[d]);
    if (rc != VHP_OK) { drain(m, d + 1); return mfail(m, rc, std::string("device ") + std::to_string(m->device[d]) + ": " + vhp_last_error(m->ctx[d])); }
  }
  int worst = VHP_OK;
  for (int d = 0; d < nd; ++d) {
    const int rc = vhp_sync(m->ctx[d]);
    if (rc != VHP_OK && worst == VHP_OK) { worst = rc; m->err = std::string("device ") + std::to_string(m->device[d]) + ": " + vhp_last_error(m->ctx[d]); }
  }
  return worst;
}

}  // extern "C"
