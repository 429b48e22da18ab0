// Developer tool: how fast can a CU pull L2-resident / HBM data into LDS by LDS-DMA (global_load_lds_dwordx4), against plain
// global_load_dwordx4 into registers?  No compute: the ceiling of the operand path of the LDS-DMA kernels.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/dma_rate.hip -o tools/_dma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../autoware_vision_pilot_amd/csrc/lds_dma.hpp"

// MODE 0: LDS-DMA, linear 1 KB per wave instruction; 1: LDS-DMA, 64-byte row pieces (16 rows per instruction, rows `pitch` bytes apart);
// 2: global_load_dwordx4 into registers (linear).  DEPTH stages of PER instructions per wave in flight.
template <int MODE, int PER, int DEPTH>
__global__ __launch_bounds__(512, 2) void pull_kernel(const char* __restrict__ src, size_t span, int iters, int pitch, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // every workgroup walks its XCD-local window of `span` bytes (so span <= L2 keeps the traffic in L2 after the first pass)
  const size_t base = ((size_t)(blockIdx.x & 7) * span);
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  u32x4 acc = {0, 0, 0, 0};
  size_t off = ((size_t)(blockIdx.x >> 3) * 8 + wave) * (size_t)PER * 1024;
  for (int it = 0; it < iters + DEPTH; ++it) {
    if (it < iters) {
#pragma unroll
      for (int i = 0; i < PER; ++i) {
        size_t o = (off + (size_t)i * 1024) % span;
        const char* g;
        if (MODE == 1) g = src + base + ((o / 1024) * 16 + (lane >> 2)) * (size_t)pitch % span + (lane & 3) * 16;
        else g = src + base + o + lane * 16;
        if (MODE == 2) { const u32x4 v = *reinterpret_cast<const u32x4*>(g); acc ^= v; }
        else VP_GLOBAL_LOAD_LDS16(g, smem + ((it % DEPTH) * 8 + wave) * PER * 1024 + i * 1024);
      }
      off += (size_t)gridDim.x * PER * 1024;
    }
    if (MODE != 2) {
      if (it + 1 < iters + DEPTH && it >= DEPTH - 1) { VP_WAIT_VMCNT((DEPTH - 1) * PER); } else if (it + 1 >= iters + DEPTH) { VP_WAIT_VMCNT(0); }
    }
  }
  if (MODE == 2 && (acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345u) sink[0] = 1;
  if (MODE != 2 && smem[threadIdx.x] == 77 && iters < 0) sink[0] = 2;
}

template <int MODE, int PER, int DEPTH>
static void run(const char* name, const char* src, size_t span, int wgs, int pitch, unsigned* sink) {
  const int iters = 400;
  const int lds = DEPTH * 8 * PER * 1024;
  auto k = pull_kernel<MODE, PER, DEPTH>;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipEvent_t a, b;
  (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  hipLaunchKernelGGL(k, dim3(wgs), dim3(512), MODE == 2 ? 0 : lds, 0, src, span, iters, pitch, sink);
  (void)hipEventRecord(a, 0);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k, dim3(wgs), dim3(512), MODE == 2 ? 0 : lds, 0, src, span, iters, pitch, sink);
  (void)hipEventRecord(b, 0);
  (void)hipEventSynchronize(b);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, a, b);
  const double bytes = 5.0 * wgs * 8.0 * PER * 1024.0 * iters;
  std::printf("%-58s wgs %4d  window/XCD %7.1f MB  %6.2f TB/s aggregate  %6.1f GB/s per workgroup\n", name, wgs, span / 1e6, bytes / (ms * 1e-3) / 1e12,
              bytes / (ms * 1e-3) / 1e9 / wgs);
}

int main() {
  const size_t total = (size_t)8 * 256 * 1024 * 1024;
  char* src; unsigned* sink;
  (void)hipMalloc(&src, total + 4096); (void)hipMemset(src, 1, total + 4096); (void)hipMalloc(&sink, 64);
  for (size_t span : {(size_t)2 << 20, (size_t)256 << 20}) {
    run<0, 6, 2>("LDS-DMA linear, 6 KB/wave/stage, 2 stages", src, span, 256, 0, sink);
    run<0, 6, 3>("LDS-DMA linear, 6 KB/wave/stage, 3 stages", src, span, 256, 0, sink);
    run<0, 3, 6>("LDS-DMA linear, 3 KB/wave/stage, 6 stages", src, span, 256, 0, sink);
    run<1, 6, 3>("LDS-DMA 64-byte row pieces (pitch 1088), 3 stages", src, span, 256, 1088, sink);
    run<2, 6, 3>("global_load_dwordx4 -> registers, linear", src, span, 256, 0, sink);
    run<0, 3, 3>("LDS-DMA linear, 3 KB/wave/stage, 3 stages, 2 WG/CU", src, span, 512, 0, sink);
    run<2, 6, 3>("global_load_dwordx4 -> registers, 2 WG/CU", src, span, 512, 0, sink);
  }
  return 0;
}
