// TEST INFRASTRUCTURE: a minimal HIP-on-CPU execution shim, so that the CPU test suite (-m "not gpu") can run the REAL
// source of the non-MFMA kernels (csrc/kernels_misc.hip, kernels_backbone.hip, kernels_autodrive.hip) and check them
// against the oracle without a GPU.  Not a performance model and not part of the product: work-items are fibers switched at
// __syncthreads / wave shuffles / MFMA, workgroups run on a few host threads, LDS is thread_local storage.
// Built by tests/emul/build.py with the host clang++; `extern __shared__` declarations are rewritten to `extern thread_local`
// (storage defined in harness.cpp).  Limits: convergent wave operations only (a divergent one aborts with a message).
#pragma once
#include <math.h>
#include <stdint.h>
#include <sys/mman.h>

#include <algorithm>
#include <atomic>
#include <cstdio>
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
#ifdef VP_EMU_TSAN
#define VP_EMU_LDS  // one workgroup at a time: LDS is plain static storage shared by its work-item threads
#else
#define VP_EMU_LDS thread_local  // per worker thread = per running workgroup
#endif
#define __shared__ static VP_EMU_LDS

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

// saves the callee-saved registers and stack pointer of the caller in *save_sp, continues on load_sp (harness.cpp)
extern "C" void emu_switch(void** save_sp, void* load_sp);

namespace emu {

// Execution model: a launch's workgroups are spread over a few OS worker threads; inside a workgroup every work-item is a
// FIBER (own stack, hand-written x86-64 context switch: emu_switch in harness.cpp) of that worker, run in lane order and switched only at synchronisation points (__syncthreads, wave
// shuffles, MFMA).  LDS (`__shared__` -> static thread_local) is therefore private to the workgroup a worker is running.
// Deterministic, no kernel-level blocking; global-memory atomics between concurrently running workgroups are real atomics.
struct Graph {  // "graphs": while a capture is open every launch / copy is recorded as a closure; hipGraphLaunch replays them
  std::vector<std::function<void()>> nodes;
};
inline Graph* capturing = nullptr;

struct Ctx {  // what threadIdx / blockIdx / ... read for the running fiber
  dim3 tid, bid, bdim, gdim;
  int lin = 0;
};

#ifdef VP_EMU_TSAN
// ---- race-check build (-DVP_EMU_TSAN, tests/emul/race_check.cpp): one OS THREAD per work-item, one workgroup at a time,
// barriers on a mutex + condition variable.  ThreadSanitizer then sees exactly the happens-before edges the GPU gives
// (__syncthreads, wave operations, kernel boundaries) and reports any two work-items that touch the same LDS / global bytes
// without one in between.  Slow (a barrier is a futex round trip of 256 threads), so the race check runs small cases.
class Barrier {
 public:
  void reset(int n) {
    std::unique_lock<std::mutex> lk(m_);
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
  int waves = 0;
  Barrier all, block;
  std::vector<std::unique_ptr<Barrier>> wave;
  std::vector<uint64_t> slots;        // [2][waves][64]
  std::vector<float> mfma_a, mfma_b;  // [2][waves][512]
};
inline thread_local Ctx ctx_storage;
inline thread_local Ctx* cur = &ctx_storage;
inline thread_local Team* team = nullptr;
inline thread_local unsigned lane_op = 0;

inline int lin() { return cur->lin; }
inline unsigned next_wave_op() { return lane_op++; }
inline uint64_t* shfl_buf(int wv, unsigned op) { return team->slots.data() + ((size_t)(op & 1) * team->waves + wv) * 64; }
inline float* mfma_buf_a(int wv, unsigned op) { return team->mfma_a.data() + ((size_t)(op & 1) * team->waves + wv) * 512; }
inline float* mfma_buf_b(int wv, unsigned op) { return team->mfma_b.data() + ((size_t)(op & 1) * team->waves + wv) * 512; }
inline void sync_block() { team->block.wait(); }
inline void sync_wave() { team->wave[cur->lin / 64]->wait(); }

template <class K, class... A>
void run_grid(K kern, dim3 grid, dim3 block, A... args) {
  const int T = (int)(block.x * block.y * block.z);
  const long long B = (long long)grid.x * grid.y * grid.z;
  Team tm;
  tm.waves = (T + 63) / 64;
  tm.all.reset(T);
  for (int w = 0; w < tm.waves; ++w) tm.wave.emplace_back(new Barrier);
  tm.slots.assign((size_t)2 * tm.waves * 64, 0);
  tm.mfma_a.assign((size_t)2 * tm.waves * 512, 0.f);
  tm.mfma_b.assign((size_t)2 * tm.waves * 512, 0.f);
  auto body = [&](int t) {
    team = &tm;
    Ctx& c = *cur;
    c.lin = t;
    c.bdim = block;
    c.gdim = grid;
    c.tid = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
    for (long long b = 0; b < B; ++b) {
      c.bid = dim3((unsigned)(b % grid.x), (unsigned)((b / grid.x) % grid.y), (unsigned)(b / ((long long)grid.x * grid.y)));
      lane_op = 0;
      tm.all.wait();  // everybody has left the previous workgroup (its LDS contents are dead)
      if (t == 0) {
        tm.block.reset(T);
        for (int w = 0; w < tm.waves; ++w) tm.wave[w]->reset(std::min(64, T - 64 * w));
      }
      tm.all.wait();
      kern(args...);
      tm.block.drop();  // a work-item that returns stops counting towards every later rendezvous
      tm.wave[t / 64]->drop();
    }
  };
  std::vector<std::thread> th;
  th.reserve(T);
  for (int t = 0; t < T; ++t) th.emplace_back(body, t);
  for (auto& x : th) x.join();
}

#else
enum { RUNNABLE = 0, WAIT_BLOCK = 1, WAIT_WAVE = 2, DONE = 3 };

struct Worker {
  static constexpr size_t kStack = 256 << 10;
  int T = 0, waves = 0, cur = 0, live = 0;
  std::vector<void*> fib;  // saved stack pointer of every suspended work-item
  void* sched = nullptr;   // ... and of the scheduler
  std::vector<Ctx> ctx;
  std::vector<int> state;
  std::vector<unsigned> wait_phase;
  unsigned block_phase = 0;
  int block_arrived = 0;
  std::vector<unsigned> wave_phase;
  std::vector<int> wave_arrived, wave_live;
  std::vector<uint64_t> slots;       // [2][waves][64] shuffle payloads, double-buffered
  std::vector<float> mfma_a, mfma_b; // [2][waves][64][8]
  std::vector<float> mfma_d;         // [2][waves][32][32] product of the wave's A and B, computed once per MFMA
  std::vector<unsigned> lane_ops;    // wave operations executed so far by each work-item (selects the buffer)
  char* stacks = nullptr;
  std::function<void()> body;

  void prepare(int threads) {
    if (threads != T) {
      if (stacks) munmap(stacks, kStack * T);
      T = threads;
      waves = (T + 63) / 64;
      stacks = static_cast<char*>(mmap(nullptr, kStack * T, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE | MAP_STACK, -1, 0));
      fib.resize(T);
      ctx.resize(T);
      state.resize(T);
      wait_phase.resize(T);
      wave_phase.resize(waves);
      wave_arrived.resize(waves);
      wave_live.resize(waves);
      lane_ops.resize(T);
      slots.resize((size_t)2 * waves * 64);
      mfma_a.resize((size_t)2 * waves * 512);
      mfma_b.resize((size_t)2 * waves * 512);
      mfma_d.resize((size_t)2 * waves * 1024);
    }
  }
  ~Worker() {
    if (stacks) munmap(stacks, kStack * T);
  }
};
inline thread_local Worker* worker = nullptr;
inline thread_local Ctx* cur = nullptr;

inline void yield_to_scheduler() {
  Worker* w = worker;
  emu_switch(&w->fib[w->cur], w->sched);
}
inline void release_block(Worker* w) {
  w->block_arrived = 0;
  ++w->block_phase;
}
inline void release_wave(Worker* w, int wv) {
  w->wave_arrived[wv] = 0;
  ++w->wave_phase[wv];
}
inline void sync_block() {
  Worker* w = worker;
  const int i = w->cur;
  w->wait_phase[i] = w->block_phase;
  if (++w->block_arrived >= w->live) {
    release_block(w);
  } else {
    w->state[i] = WAIT_BLOCK;
    yield_to_scheduler();
  }
}
template <class F>
inline void sync_wave_then(F&& last_arriver_work) {  // the work runs once, after every lane has arrived, before any lane continues
  Worker* w = worker;
  const int i = w->cur, wv = i / 64;
  w->wait_phase[i] = w->wave_phase[wv];
  if (++w->wave_arrived[wv] >= w->wave_live[wv]) {
    last_arriver_work();
    release_wave(w, wv);
  } else {
    w->state[i] = WAIT_WAVE;
    yield_to_scheduler();
  }
}
inline void sync_wave() {
  sync_wave_then([] {});
}
inline void fiber_main() {
  Worker* w = worker;
  w->body();
  const int i = w->cur, wv = i / 64;  // a work-item that returns stops counting towards every later rendezvous
  w->state[i] = DONE;
  --w->live;
  --w->wave_live[wv];
  if (w->live > 0 && w->block_arrived >= w->live) release_block(w);
  if (w->wave_live[wv] > 0 && w->wave_arrived[wv] >= w->wave_live[wv]) release_wave(w, wv);
  emu_switch(&w->fib[i], w->sched);  // never resumed
}

inline void run_block(Worker* w, dim3 grid, dim3 block, long long b) {
  const int T = w->T;
  w->live = T;
  w->block_phase = 0;
  w->block_arrived = 0;
  for (int wv = 0; wv < w->waves; ++wv) {
    w->wave_phase[wv] = 0;
    w->wave_arrived[wv] = 0;
    w->wave_live[wv] = std::min(64, T - 64 * wv);
  }
  const dim3 bid((unsigned)(b % grid.x), (unsigned)((b / grid.x) % grid.y), (unsigned)(b / ((long long)grid.x * grid.y)));
  for (int t = 0; t < T; ++t) {
    Ctx& c = w->ctx[t];
    c.lin = t;
    c.bdim = block;
    c.gdim = grid;
    c.bid = bid;
    c.tid = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
    w->state[t] = RUNNABLE;
    w->lane_ops[t] = 0;
    // fresh stack: [six callee-saved registers][entry point][fake return address]; emu_switch pops the registers and
    // `ret`s into fiber_main with the stack pointer at 8 mod 16, as after a call
    void** top = reinterpret_cast<void**>(w->stacks + Worker::kStack * (t + 1));
    top[-1] = nullptr;
    top[-2] = reinterpret_cast<void*>(&fiber_main);
    for (int r = 3; r <= 8; ++r) top[-r] = nullptr;
    w->fib[t] = top - 8;
  }
  while (w->live > 0) {
    bool progressed = false;
    for (int t = 0; t < T; ++t) {
      const int st = w->state[t];
      if (st == DONE) continue;
      if (st == WAIT_BLOCK && w->block_phase == w->wait_phase[t]) continue;
      if (st == WAIT_WAVE && w->wave_phase[t / 64] == w->wait_phase[t]) continue;
      w->state[t] = RUNNABLE;
      w->cur = t;
      cur = &w->ctx[t];
      emu_switch(&w->sched, w->fib[t]);
      progressed = true;
    }
    if (!progressed) {
      std::fprintf(stderr, "tests/emul: deadlock -- divergent __syncthreads / wave operation in workgroup %lld\n", b);
      std::abort();
    }
  }
}

template <class K, class... A>
void run_grid(K kern, dim3 grid, dim3 block, A... args) {
  const int T = (int)(block.x * block.y * block.z);
  const long long B = (long long)grid.x * grid.y * grid.z;
  std::atomic<long long> next{0};
  auto work = [&] {
    Worker me;
    me.prepare(T);
    me.body = [&] { kern(args...); };
    worker = &me;
    for (long long b = next.fetch_add(1); b < B; b = next.fetch_add(1)) run_block(&me, grid, block, b);
    worker = nullptr;
    cur = nullptr;
  };
  const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
  const int nw = (int)std::min<long long>(std::min(8u, hw), B);
  std::vector<std::thread> th;
  for (int i = 0; i < nw; ++i) th.emplace_back(work);
  for (auto& x : th) x.join();
}

inline int lin() { return worker->cur; }
inline unsigned next_wave_op() { return worker->lane_ops[worker->cur]++; }
inline uint64_t* shfl_buf(int wv, unsigned op) { return worker->slots.data() + ((size_t)(op & 1) * worker->waves + wv) * 64; }
inline float* mfma_buf_a(int wv, unsigned op) { return worker->mfma_a.data() + ((size_t)(op & 1) * worker->waves + wv) * 512; }
inline float* mfma_buf_b(int wv, unsigned op) { return worker->mfma_b.data() + ((size_t)(op & 1) * worker->waves + wv) * 512; }
inline float* mfma_buf_d(int wv, unsigned op) { return worker->mfma_d.data() + ((size_t)(op & 1) * worker->waves + wv) * 1024; }

#endif  // VP_EMU_TSAN

template <class K, class... A>
void launch(K kern, dim3 grid, dim3 block, size_t /*dynamic LDS: thread_local storage in harness.cpp*/, A... args) {
  if (capturing) {  // as on the device: work issued into a capturing stream is recorded, not executed
    capturing->nodes.push_back([=] { run_grid(kern, grid, block, args...); });
    return;
  }
  run_grid(kern, grid, block, args...);
}

template <class V>
V shfl_xor(V v, int mask) {
  static_assert(sizeof(V) <= 8, "shuffle payload");
  const int i = lin(), wv = i / 64, lane = i % 64;
  uint64_t* buf = shfl_buf(wv, next_wave_op());  // double-buffered: one rendezvous per operation is enough
  uint64_t bits = 0;
  std::memcpy(&bits, &v, sizeof(V));
  buf[lane] = bits;
  sync_wave();
  const uint64_t got = buf[lane ^ mask];
  V r;
  std::memcpy(&r, &got, sizeof(V));
  return r;
}

// v_mfma_f32_32x32x16_f16 (gfx950): D[32x32] = A[32x16] * B[16x32] + C over one wave.  Register layout (CDNA ISA):
//   A: lane l holds row m = l % 32, columns k = 8 * (l / 32) + 0..7;   B: lane l holds column n = l % 32, rows k = 8 * (l / 32) + 0..7;
//   C / D: lane l holds column n = l % 32; element r (0..15) is row m = 8 * (r / 4) + 4 * (l / 32) + (r % 4).
// Products of two fp16 values are exact in fp32; the accumulation order inside the instruction is not architecturally
// defined -- here D = C + (sum over k, in k order, fp32) (the kernel tests carry a tolerance for that, as the GPU ones do).
template <class H8, class F16>
F16 mfma_32x32x16_f16(H8 a, H8 b, F16 c) {
  const int i = lin(), wv = i / 64, lane = i % 64;
  const unsigned op = next_wave_op();
  float* A = mfma_buf_a(wv, op);
  float* B = mfma_buf_b(wv, op);
  for (int e = 0; e < 8; ++e) {
    A[lane * 8 + e] = (float)a[e];
    B[lane * 8 + e] = (float)b[e];
  }
  const int n = lane % 32;
#ifdef VP_EMU_TSAN
  sync_wave();
  for (int r = 0; r < 16; ++r) {
    const int m = 8 * (r / 4) + 4 * (lane / 32) + (r % 4);
    float acc = 0.0f;
    for (int k = 0; k < 16; ++k) acc += A[(m + 32 * (k / 8)) * 8 + (k % 8)] * B[(n + 32 * (k / 8)) * 8 + (k % 8)];
    c[r] += acc;
  }
#else
  float* D = mfma_buf_d(wv, op);
  sync_wave_then([=] {  // the last lane to arrive multiplies the whole 32x16 by 16x32 tile once (vectorisable row updates)
    float Bt[16][32];
    for (int k = 0; k < 16; ++k)
      for (int nn = 0; nn < 32; ++nn) Bt[k][nn] = B[(nn + 32 * (k / 8)) * 8 + (k % 8)];
    for (int m = 0; m < 32; ++m) {
      float row[32];
      for (int nn = 0; nn < 32; ++nn) row[nn] = 0.0f;
      for (int k = 0; k < 16; ++k) {
        const float av = A[(m + 32 * (k / 8)) * 8 + (k % 8)];
        for (int nn = 0; nn < 32; ++nn) row[nn] += av * Bt[k][nn];
      }
      for (int nn = 0; nn < 32; ++nn) D[m * 32 + nn] = row[nn];
    }
  });
  for (int r = 0; r < 16; ++r) c[r] += D[(8 * (r / 4) + 4 * (lane / 32) + (r % 4)) * 32 + n];
#endif
  return c;
}

// v_mfma_f32_16x16x32_f16 (gfx950): D[16x16] = A[16x32] * B[32x16] + C over one wave.  Register layout:
//   A: lane l holds row m = l % 16, columns k = 8 * (l / 16) + 0..7;   B: lane l holds column n = l % 16, rows k = 8 * (l / 16) + 0..7;
//   C / D: lane l holds column n = l % 16; element r (0..3) is row m = 4 * (l / 16) + r.
template <class H8, class F4>
F4 mfma_16x16x32_f16(H8 a, H8 b, F4 c) {
  const int i = lin(), wv = i / 64, lane = i % 64;
  const unsigned op = next_wave_op();
  float* A = mfma_buf_a(wv, op);
  float* B = mfma_buf_b(wv, op);
  for (int e = 0; e < 8; ++e) {
    A[lane * 8 + e] = (float)a[e];
    B[lane * 8 + e] = (float)b[e];
  }
  sync_wave();
  const int n = lane % 16;
  for (int r = 0; r < 4; ++r) {
    const int m = 4 * (lane / 16) + r;
    float acc = 0.0f;
    for (int k = 0; k < 32; ++k) acc += A[(m + 16 * (k / 8)) * 8 + (k % 8)] * B[(n + 16 * (k / 8)) * 8 + (k % 8)];
    c[r] += acc;
  }
  return c;
}

}  // namespace emu

#define __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, x, y, z) emu::mfma_32x32x16_f16(a, b, c)
#define __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, x, y, z) emu::mfma_16x16x32_f16(a, b, c)
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
// LDS-DMA (global_load_lds_dwordx4): lane i's 16 bytes land at the wave-uniform LDS address + 16 * i.  Performed at issue.
#define VP_GLOBAL_LOAD_LDS16(G, L) std::memcpy(reinterpret_cast<char*>(L) + 16 * (threadIdx.x & 63), reinterpret_cast<const char*>(G), 16)
#define VP_WAIT_VMCNT(N) ((void)0)
#define VP_WAIT_LGKMCNT(N) ((void)0)
#define VP_LDS_BARRIER() __syncthreads()
#define __builtin_amdgcn_s_memrealtime() 0ull
#define __builtin_amdgcn_wave_barrier() emu::sync_wave()  // lanes of a wave run in lockstep on the device; here they must meet
#define __builtin_amdgcn_s_sleep(x) ((void)0)
#define __builtin_amdgcn_s_getreg(x) 0u
#define __builtin_amdgcn_readfirstlane(x) (x)   // used on wave-uniform values only

#define threadIdx (emu::cur->tid)
#define blockIdx (emu::cur->bid)
#define blockDim (emu::cur->bdim)
#define gridDim (emu::cur->gdim)
#define hipLaunchKernelGGL(kern, grid, block, shmem, st, ...) emu::launch(kern, grid, block, shmem, __VA_ARGS__)

inline void __syncthreads() { emu::sync_block(); }
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
inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount = 63 };
inline hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t, int) { *v = 256; return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
inline hipError_t hipMalloc(void** p, size_t n) { *p = std::aligned_alloc(256, (n + 255) / 256 * 256 + 256); return *p ? hipSuccess : hipErrorOutOfMemory; }
template <class T> inline hipError_t hipMalloc(T** p, size_t n) { return hipMalloc(reinterpret_cast<void**>(p), n); }
inline hipError_t hipFree(void* p) { std::free(p); return hipSuccess; }
inline hipError_t hipHostMalloc(void** p, size_t n, unsigned = 0) { return hipMalloc(p, n); }
template <class T> inline hipError_t hipHostMalloc(T** p, size_t n, unsigned f = 0) { return hipMalloc(reinterpret_cast<void**>(p), n); }
inline hipError_t hipHostFree(void* p) { std::free(p); return hipSuccess; }
enum { hipHostRegisterDefault = 0, hipHostRegisterPortable = 1 };
inline hipError_t hipHostRegister(void*, size_t, unsigned) { return hipSuccess; }   // host memory is host memory here
inline hipError_t hipHostUnregister(void*) { return hipSuccess; }
inline hipError_t hipMemset(void* p, int v, size_t n) { std::memset(p, v, n); return hipSuccess; }
inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) {
  if (emu::capturing) {
    emu::capturing->nodes.push_back([=] { std::memset(p, v, n); });
    return hipSuccess;
  }
  std::memset(p, v, n);
  return hipSuccess;
}
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { std::memmove(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind k, hipStream_t) {
  if (emu::capturing) {
    emu::capturing->nodes.push_back([=] { std::memmove(d, s, n); });
    return hipSuccess;
  }
  std::memmove(d, s, n);
  return hipSuccess;
}
inline hipError_t hipMemcpy2DAsync(void* d, size_t dpitch, const void* s, size_t spitch, size_t width, size_t height, hipMemcpyKind, hipStream_t) {
  for (size_t y = 0; y < height; ++y) std::memmove(static_cast<char*>(d) + y * dpitch, static_cast<const char*>(s) + y * spitch, width);
  return hipSuccess;
}
inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = reinterpret_cast<hipStream_t>(1); return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = reinterpret_cast<hipEvent_t>(1); return hipSuccess; }
constexpr unsigned hipEventDisableTiming = 2;
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = reinterpret_cast<hipEvent_t>(1); return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
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
inline unsigned atomicOr(unsigned* p, unsigned v) { return __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }
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
inline void __threadfence_block() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
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
