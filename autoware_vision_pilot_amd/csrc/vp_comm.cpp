// Multi-camera result exchange of libvp_hip: one RCCL all-gather per frame on the engine's own HIP stream
// (BASELINE configs[3]: "8-camera stream, one camera per GPU, RCCL all-gather for the BEV / PathFinder head").
//
// The reference has no inference-time collective (one camera per backend instance); the gather is the added
// hand-off that gives every rank all cameras' results for a fused ego-path consumer.  It lives BEHIND the C ABI so a C++ /
// ROS2 host needs no Python and no torch: the host distributes the 128-byte unique id (rank 0 creates it) by whatever means
// it already has (ROS parameter, file, MPI, torch.distributed in bench.py), every rank calls vp_comm_create, and per frame
//     vp_enqueue(engine); vp_gather(engine, comm, VP_GATHER_MASK);      // no host sync in between
// enqueues ncclAllGather behind the frame's graph on the SAME stream: it runs as soon as the decode kernel has finished
// and overlaps the next engine's (next frame in flight) encoder.  Records are <= 2.5 MB per rank, so over xGMI the collective
// is latency-bound (tens of microseconds), not per-link bandwidth-bound.
//
// librccl.so (573 MB) is bound with dlopen at the first vp_comm_* call, not at load time: single-camera deployments (the ROS2
// node, the Python twins) never map it, and libvp_hip.so loads on hosts without RCCL.  Types and prototypes come from
// <rccl/rccl.h>; every call below is a direct RCCL call.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstring>
#include <mutex>

#include "vp_handle.hpp"

namespace {

struct Rccl {
  void* handle = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  std::string error;
};

Rccl& rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    // RTLD_NOLOAD first: a host that already mapped RCCL (torch bundles its own copy) must keep using that one
    for (const char* name : {"librccl.so.1", "librccl.so"}) {
      r.handle = dlopen(name, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
      if (r.handle) break;
    }
    if (!r.handle)
      for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        r.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (r.handle) break;
      }
    if (!r.handle) {
      r.error = std::string("cannot load librccl.so: ") + dlerror();
      return;
    }
#define VP_RCCL_SYM(field, sym)                                        \
  r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.handle, sym)); \
  if (!r.field) r.error = std::string("librccl.so lacks ") + sym;
    VP_RCCL_SYM(GetUniqueId, "ncclGetUniqueId")
    VP_RCCL_SYM(CommInitRank, "ncclCommInitRank")
    VP_RCCL_SYM(CommDestroy, "ncclCommDestroy")
    VP_RCCL_SYM(AllGather, "ncclAllGather")
    VP_RCCL_SYM(GetErrorString, "ncclGetErrorString")
#undef VP_RCCL_SYM
  });
  return r;
}

void set_err(char* err, size_t n, const std::string& msg) {
  if (err && n) std::snprintf(err, n, "%s", msg.c_str());
}

}  // namespace

struct vp_comm {
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1, gpu = 0;
  size_t bytes_per_rank = 0;
  void* d_gather = nullptr;   // [world][bytes_per_rank]
  void* h_gather = nullptr;   // pinned host copy, filled by vp_comm_fetch
  size_t last_bytes = 0;      // record size of the last vp_gather
  hipStream_t last_stream = nullptr;  // ... and the stream it was enqueued on: the only stream vp_comm_fetch's copy is ordered on
  hipEvent_t last_done = nullptr;     // recorded behind every vp_gather: "is this communicator still in flight on that stream?"
  std::string err;
};

extern "C" {

static_assert(sizeof(ncclUniqueId) == VP_COMM_ID_BYTES, "VP_COMM_ID_BYTES must match ncclUniqueId");

int vp_comm_unique_id(void* id_out, char* err, size_t err_len) {
  if (!id_out) return VP_ERR_ARG;
  Rccl& r = rccl();
  if (!r.error.empty()) {
    set_err(err, err_len, r.error);
    return VP_ERR_STATE;
  }
  ncclUniqueId id;
  const ncclResult_t rc = r.GetUniqueId(&id);
  if (rc != ncclSuccess) {
    set_err(err, err_len, std::string("ncclGetUniqueId: ") + r.GetErrorString(rc));
    return VP_ERR_HIP;
  }
  std::memcpy(id_out, &id, sizeof(id));
  return VP_OK;
}

int vp_comm_create(vp_comm** out, const void* unique_id, int rank, int world, int gpu_id, size_t bytes_per_rank, char* err,
                   size_t err_len) {
  if (!out) return VP_ERR_ARG;
  *out = nullptr;
  if (!unique_id || world < 1 || rank < 0 || rank >= world || bytes_per_rank == 0) {
    set_err(err, err_len, "vp_comm_create: need a unique id, 0 <= rank < world and a non-zero record size");
    return VP_ERR_ARG;
  }
  Rccl& r = rccl();
  if (!r.error.empty()) {
    set_err(err, err_len, r.error);
    return VP_ERR_STATE;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || gpu_id < 0 || gpu_id >= ndev) {
    set_err(err, err_len, "vp_comm_create: gpu_id out of range");
    return VP_ERR_ARG;
  }
  auto c = std::make_unique<vp_comm>();
  c->rank = rank;
  c->world = world;
  c->gpu = gpu_id;
  c->bytes_per_rank = bytes_per_rank;
  if (hipSetDevice(gpu_id) != hipSuccess || hipMalloc(&c->d_gather, bytes_per_rank * world) != hipSuccess ||
      hipHostMalloc(&c->h_gather, bytes_per_rank * world, hipHostMallocDefault) != hipSuccess) {
    if (c->d_gather) hipFree(c->d_gather);
    set_err(err, err_len, "vp_comm_create: cannot allocate the gather buffers");
    return VP_ERR_HIP;
  }
  ncclUniqueId id;
  std::memcpy(&id, unique_id, sizeof(id));
  const ncclResult_t rc = r.CommInitRank(&c->comm, world, id, rank);
  if (rc != ncclSuccess) {
    hipFree(c->d_gather);
    hipHostFree(c->h_gather);
    set_err(err, err_len, std::string("ncclCommInitRank: ") + r.GetErrorString(rc));
    return VP_ERR_HIP;
  }
  *out = c.release();
  return VP_OK;
}

void vp_comm_destroy(vp_comm* c) {
  if (!c) return;
  hipSetDevice(c->gpu);
  if (c->comm) rccl().CommDestroy(c->comm);
  if (c->last_done) hipEventDestroy(c->last_done);
  if (c->d_gather) hipFree(c->d_gather);
  if (c->h_gather) hipHostFree(c->h_gather);
  delete c;
}

const char* vp_comm_last_error(const vp_comm* c) { return c ? c->err.c_str() : "null communicator"; }
int vp_comm_rank(const vp_comm* c) { return c ? c->rank : VP_ERR_ARG; }
int vp_comm_world(const vp_comm* c) { return c ? c->world : VP_ERR_ARG; }

// All-gather of this engine's last result record into the communicator's device buffer [world][record], enqueued on the
// ENGINE's stream behind whatever the engine has in flight; returns without synchronising the host.
int vp_gather(vp_engine* e, vp_comm* c, int what) {
  if (!e || !e->impl || !c) return VP_ERR_ARG;
  vp::Engine& g = *e->impl;
  const void* src = nullptr;
  size_t bytes = 0;
  if (what == VP_GATHER_MASK) {
    src = g.dev_mask();
    bytes = (size_t)g.out_h() * g.out_w();
  } else if (what == VP_GATHER_LOGITS) {
    src = g.dev_logits();
    bytes = (size_t)g.out_c() * g.out_h() * g.out_w() * sizeof(float);
  }
  if (!src || bytes == 0 || bytes > c->bytes_per_rank || g.gpu() != c->gpu) {
    c->err = e->err = "vp_gather: no such output on this engine, record larger than the communicator's, or different GPUs";
    return VP_ERR_ARG;
  }
  if (!g.have_outputs()) {
    c->err = e->err = "vp_gather: run the engine on a frame first";
    return VP_ERR_STATE;
  }
  if (hipSetDevice(c->gpu) != hipSuccess) return VP_ERR_HIP;
  // ONE COMMUNICATOR PER ENGINE IN FLIGHT, enforced (round 4; a comment until then): RCCL operations of one communicator must not run
  // concurrently on two streams.  The same stream again is ordered by the stream; ANOTHER engine's stream is accepted only once the
  // previous gather has completed there.
  if (c->last_stream && c->last_stream != g.stream() && c->last_done && hipEventQuery(c->last_done) == hipErrorNotReady) {
    c->err = e->err = "vp_gather: this communicator's previous all-gather is still in flight on another engine's stream -- one communicator per "
                      "engine in flight (vp_comm_create one per in-flight engine, or vp_sync the other engine first)";
    return VP_ERR_STATE;
  }
  if (!c->last_done && hipEventCreateWithFlags(&c->last_done, hipEventDisableTiming) != hipSuccess) return VP_ERR_HIP;
  const ncclResult_t rc = rccl().AllGather(src, c->d_gather, bytes, ncclUint8, c->comm, g.stream());
  if (rc != ncclSuccess) {
    c->err = e->err = std::string("ncclAllGather: ") + rccl().GetErrorString(rc);
    return VP_ERR_HIP;
  }
  c->last_bytes = bytes;
  c->last_stream = e->impl->stream();
  if (hipEventRecord(c->last_done, c->last_stream) != hipSuccess) return VP_ERR_HIP;
  return VP_OK;
}

int vp_comm_device_buffer(const vp_comm* c, void** dev, size_t* record_bytes) {
  if (!c || !dev) return VP_ERR_ARG;
  *dev = c->d_gather;
  if (record_bytes) *record_bytes = c->last_bytes;
  return VP_OK;
}

// D2H of the gathered records (world x record_bytes of the last vp_gather) on the engine's stream, then one host sync.
int vp_comm_fetch(vp_comm* c, vp_engine* e, const void** host, size_t* record_bytes) {
  if (!c || !e || !e->impl || !host) return VP_ERR_ARG;
  if (c->last_bytes == 0) {
    c->err = "vp_comm_fetch: nothing gathered yet";
    return VP_ERR_STATE;
  }
  if (e->impl->stream() != c->last_stream) {  // another engine's stream carries no order against the all-gather: stale or partial records
    c->err = "vp_comm_fetch: pass the engine whose stream ran the last vp_gather";
    return VP_ERR_ARG;
  }
  if (hipSetDevice(c->gpu) != hipSuccess) return VP_ERR_HIP;
  if (hipMemcpyAsync(c->h_gather, c->d_gather, c->last_bytes * c->world, hipMemcpyDeviceToHost, e->impl->stream()) != hipSuccess ||
      hipStreamSynchronize(e->impl->stream()) != hipSuccess) {
    c->err = "vp_comm_fetch: copy failed";
    return VP_ERR_HIP;
  }
  *host = c->h_gather;
  if (record_bytes) *record_bytes = c->last_bytes;
  return VP_OK;
}

}  // extern "C"
