// TEST INFRASTRUCTURE: a minimal HIP-on-CPU execution shim, so that the CPU test suite (-m "not gpu") can run the REAL
// source of the non-MFMA kernels (csrc/kernels_misc.hip, kernels_backbone.hip, kernels_autodrive.hip) and check them
// against the oracle without a GPU.  Not a performance model and not part of the product: one workgroup at a time, one
// host thread per work-item, __syncthreads / wave shuffles as real rendezvous, LDS as static storage.
// Built by tests/emul/build.py with the host clang++; `extern __shared__` declarations are rewritten to plain `extern`
// (storage defined in harness.cpp).  Limits: 1-D workgroups, no MFMA / inline-asm kernels (those are GPU-only tests).
#pragma once
#include <math.h>
#include <stdint.h>

#include <algorithm>
#include <cmath>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__ static

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct alignas(16) int4 {
  int x, y, z, w;
};
struct alignas(8) int2 {
  int x, y;
};
using std::max;
using std::min;
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2 };
typedef void* hipStream_t;
inline hipError_t hipGetLastError() { return hipSuccess; }
inline const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "success" : "emulated HIP error"; }

namespace emu {

// Counting barrier whose participant count can shrink (work-items that return early) and be reset between workgroups.
class Barrier {
 public:
  void reset(int n) {
    expected_ = n;
    waiting_ = 0;
  }
  void wait() {
    std::unique_lock<std::mutex> lk(m_);
    const unsigned ph = phase_;
    if (++waiting_ >= expected_) {
      waiting_ = 0;
      ++phase_;
      cv_.notify_all();
    } else {
      cv_.wait(lk, [&] { return phase_ != ph; });
    }
  }
  void drop() {
    std::unique_lock<std::mutex> lk(m_);
    if (--expected_ > 0 && waiting_ >= expected_) {
      waiting_ = 0;
      ++phase_;
      cv_.notify_all();
    }
  }

 private:
  std::mutex m_;
  std::condition_variable cv_;
  int expected_ = 0, waiting_ = 0;
  unsigned phase_ = 0;
};

struct Team {
  int threads = 0;
  Barrier all, block;
  std::vector<std::unique_ptr<Barrier>> wave;
  std::vector<uint64_t> slots;  // [waves][64]
  std::vector<float> mfma_a, mfma_b;  // [waves][64][8]
};

struct Ctx {
  dim3 tid, bid, bdim, gdim;
  Team* team = nullptr;
  int lin = 0;
};
inline thread_local Ctx ctx;

// "Graphs": while a capture is open every launch / copy is executed AND recorded as a closure; hipGraphLaunch replays them.
struct Graph {
  std::vector<std::function<void()>> nodes;
};
inline Graph* capturing = nullptr;

template <class K, class... A>
void run_grid(K kern, dim3 grid, dim3 block, A... args);

template <class K, class... A>
void launch(K kern, dim3 grid, dim3 block, size_t /*dynamic LDS: storage is static in harness.cpp*/, A... args) {
  if (capturing) capturing->nodes.push_back([=] { run_grid(kern, grid, block, args...); });
  run_grid(kern, grid, block, args...);
}

template <class K, class... A>
void run_grid(K kern, dim3 grid, dim3 block, A... args) {
  const int T = (int)(block.x * block.y * block.z);
  const long long B = (long long)grid.x * grid.y * grid.z;
  Team team;
  team.threads = T;
  team.all.reset(T);
  const int waves = (T + 63) / 64;
  for (int w = 0; w < waves; ++w) team.wave.emplace_back(new Barrier);
  team.slots.assign((size_t)waves * 64, 0);
  team.mfma_a.assign((size_t)waves * 64 * 8, 0.f);
  team.mfma_b.assign((size_t)waves * 64 * 8, 0.f);
  auto body = [&](int t) {
    Ctx& c = ctx;
    c.team = &team;
    c.lin = t;
    c.bdim = block;
    c.gdim = grid;
    c.tid = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
    for (long long b = 0; b < B; ++b) {
      c.bid = dim3((unsigned)(b % grid.x), (unsigned)((b / grid.x) % grid.y), (unsigned)(b / ((long long)grid.x * grid.y)));
      team.all.wait();  // everybody has left the previous workgroup (its LDS contents are dead)
      if (t == 0) {
        team.block.reset(T);
        for (int w = 0; w < waves; ++w) team.wave[w]->reset(std::min(64, T - 64 * w));
      }
      team.all.wait();
      kern(args...);
      team.block.drop();
      team.wave[t / 64]->drop();
    }
  };
  std::vector<std::thread> th;
  th.reserve(T);
  for (int t = 0; t < T; ++t) th.emplace_back(body, t);
  for (auto& x : th) x.join();
}

template <class V>
V shfl_xor(V v, int mask) {
  static_assert(sizeof(V) <= 8, "shuffle payload");
  Ctx& c = ctx;
  const int w = c.lin / 64, lane = c.lin % 64;
  uint64_t bits = 0;
  std::memcpy(&bits, &v, sizeof(V));
  c.team->slots[(size_t)w * 64 + lane] = bits;
  c.team->wave[w]->wait();
  const uint64_t got = c.team->slots[(size_t)w * 64 + (lane ^ mask)];
  c.team->wave[w]->wait();
  V r;
  std::memcpy(&r, &got, sizeof(V));
  return r;
}

// v_mfma_f32_32x32x16_f16 (gfx950): D[32x32] = A[32x16] * B[16x32] + C over one wave.  Register layout (CDNA ISA):
//   A: lane l holds row m = l % 32, columns k = 8 * (l / 32) + 0..7;   B: lane l holds column n = l % 32, rows k = 8 * (l / 32) + 0..7;
//   C / D: lane l holds column n = l % 32; element r (0..15) is row m = 8 * (r / 4) + 4 * (l / 32) + (r % 4).
// Products of two fp16 values are exact in fp32; the accumulation order inside the instruction is not architecturally
// defined -- summed here in fp32 in k order (the kernel tests carry a tolerance for that, as the GPU ones do).
template <class H8, class F16>
F16 mfma_32x32x16_f16(H8 a, H8 b, F16 c) {
  Ctx& x = ctx;
  const int w = x.lin / 64, lane = x.lin % 64;
  float* A = x.team->mfma_a.data() + (size_t)w * 512;
  float* B = x.team->mfma_b.data() + (size_t)w * 512;
  for (int i = 0; i < 8; ++i) {
    A[lane * 8 + i] = (float)a[i];
    B[lane * 8 + i] = (float)b[i];
  }
  x.team->wave[w]->wait();
  const int n = lane % 32;
  for (int r = 0; r < 16; ++r) {
    const int m = 8 * (r / 4) + 4 * (lane / 32) + (r % 4);
    float acc = c[r];
    for (int k = 0; k < 16; ++k) acc += A[(m + 32 * (k / 8)) * 8 + (k % 8)] * B[(n + 32 * (k / 8)) * 8 + (k % 8)];
    c[r] = acc;
  }
  x.team->wave[w]->wait();
  return c;
}

}  // namespace emu

#define __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, x, y, z) emu::mfma_32x32x16_f16(a, b, c)
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
#define __builtin_amdgcn_s_sleep(x) ((void)0)
#define __builtin_amdgcn_s_getreg(x) 0u

#define threadIdx (emu::ctx.tid)
#define blockIdx (emu::ctx.bid)
#define blockDim (emu::ctx.bdim)
#define gridDim (emu::ctx.gdim)
#define hipLaunchKernelGGL(kern, grid, block, shmem, st, ...) emu::launch(kern, grid, block, shmem, __VA_ARGS__)

inline void __syncthreads() { emu::ctx.team->block.wait(); }
template <class V>
inline V __shfl_xor(V v, int mask) { return emu::shfl_xor(v, mask); }

// ---- runtime API: "device" memory is host memory, streams are synchronous, graphs are closure lists (see emu::Graph)
enum hipMemcpyKind { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3 };
enum { hipStreamNonBlocking = 1, hipHostMallocDefault = 0, hipStreamCaptureModeThreadLocal = 1, hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
typedef void* hipEvent_t;
typedef emu::Graph* hipGraph_t;
typedef emu::Graph* hipGraphExec_t;
inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
inline hipError_t hipMalloc(void** p, size_t n) { *p = std::aligned_alloc(256, (n + 255) / 256 * 256 + 256); return *p ? hipSuccess : hipErrorOutOfMemory; }
template <class T> inline hipError_t hipMalloc(T** p, size_t n) { return hipMalloc(reinterpret_cast<void**>(p), n); }
inline hipError_t hipFree(void* p) { std::free(p); return hipSuccess; }
inline hipError_t hipHostMalloc(void** p, size_t n, unsigned = 0) { return hipMalloc(p, n); }
template <class T> inline hipError_t hipHostMalloc(T** p, size_t n, unsigned f = 0) { return hipMalloc(reinterpret_cast<void**>(p), n); }
inline hipError_t hipHostFree(void* p) { std::free(p); return hipSuccess; }
inline hipError_t hipMemset(void* p, int v, size_t n) { std::memset(p, v, n); return hipSuccess; }
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { std::memmove(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind k, hipStream_t) {
  if (emu::capturing && k != hipMemcpyHostToDevice) emu::capturing->nodes.push_back([=] { std::memmove(d, s, n); });
  std::memmove(d, s, n);
  return hipSuccess;
}
inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = reinterpret_cast<hipStream_t>(1); return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = reinterpret_cast<hipEvent_t>(1); return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.0f; return hipSuccess; }
inline hipError_t hipFuncSetAttribute(const void*, int, int) { return hipSuccess; }
inline hipError_t hipStreamBeginCapture(hipStream_t, int) { emu::capturing = new emu::Graph; return hipSuccess; }
inline hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t* g) { *g = emu::capturing; emu::capturing = nullptr; return hipSuccess; }
inline hipError_t hipGraphInstantiate(hipGraphExec_t* e, hipGraph_t g, void*, void*, size_t) { *e = new emu::Graph(*g); return hipSuccess; }
inline hipError_t hipGraphLaunch(hipGraphExec_t e, hipStream_t) { for (auto& f : e->nodes) f(); return hipSuccess; }
inline hipError_t hipGraphDestroy(hipGraph_t g) { delete g; return hipSuccess; }
inline hipError_t hipGraphExecDestroy(hipGraphExec_t e) { delete e; return hipSuccess; }

// ---- atomics (device scope == process scope here)
inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
inline unsigned atomicAdd(unsigned* p, unsigned v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
inline int atomicAdd(int* p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
inline unsigned atomicMin(unsigned* p, unsigned v) {
  unsigned old = __atomic_load_n(p, __ATOMIC_RELAXED);
  while (v < old && !__atomic_compare_exchange_n(p, &old, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {
  }
  return old;
}
inline unsigned atomicMax(unsigned* p, unsigned v) {
  unsigned old = __atomic_load_n(p, __ATOMIC_RELAXED);
  while (v > old && !__atomic_compare_exchange_n(p, &old, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {
  }
  return old;
}

// ---- math intrinsics.  The hardware's v_rcp_f32 / v_exp_f32 are ~1 ulp approximations; exact host math stands in
// (only the VP_FP16 activation variants use them, and their tests carry an fp16-sized tolerance).
#define __builtin_amdgcn_rcpf(x) (1.0f / (x))
#define __builtin_amdgcn_exp2f(x) exp2f(x)
inline float __expf(float x) { return expf(x); }
inline float __fdiv_rn(float a, float b) { return a / b; }
inline float __fmul_rn(float a, float b) { return a * b; }
inline float __fadd_rn(float a, float b) { return a + b; }
inline float __fsub_rn(float a, float b) { return a - b; }
inline int __float2int_rn(float x) { return (int)lrintf(x); }           // default rounding mode: half to even
inline long long __float2ll_rn(float x) { return (long long)llrintf(x); }
inline unsigned __float_as_uint(float f) {
  unsigned u;
  std::memcpy(&u, &f, 4);
  return u;
}
inline float __uint_as_float(unsigned u) {
  float f;
  std::memcpy(&f, &u, 4);
  return f;
}
