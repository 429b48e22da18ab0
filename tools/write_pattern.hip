// Developer tool: how fast can 52 MB be WRITTEN in the access patterns of the up-sampling layers?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/write_pattern.hip -o tools/_write_pattern
// out = 320 x 640 pixels x 128 fp16 channels (256 B per pixel).  Patterns, one 256-thread workgroup per 128 "input" pixels
// and quadrant (1600 workgroups, 32 KB each), every lane stores 16 B:
//   0 linear       : workgroup b writes bytes [32 KB * b, 32 KB * (b+1))
//   1 pixel shuffle: input pixel (y, x) of a 160x320 map, quadrant (dy, dx) -> output pixel (2y+dy, 2x+dx): 256-B pieces, 512-B stride
//   2 pixel shuffle, quadrant FASTEST in the workgroup order (the 4 quadrants of a pixel tile are written at about the same time)
//   3 row pairs    : a workgroup writes BOTH dx quadrants of its pixels: 512 B contiguous per input pixel, 64 pixels per workgroup
#include <hip/hip_runtime.h>

#include <cstdio>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void wr(u32x4* out, int mode) {
  const int W = 320, tid = threadIdx.x;
  const u32x4 v = {(unsigned)blockIdx.x, (unsigned)tid, 1u, 2u};
  if (mode == 0) {
    for (int i = 0; i < 8; ++i) out[(size_t)blockIdx.x * 2048 + i * 256 + tid] = v;
    return;
  }
  if (mode == 3) {
    const int tile = blockIdx.x >> 1, dy = blockIdx.x & 1;  // 800 tiles of 64 pixels x 2 row parities
    for (int i = 0; i < 8; ++i) {
      const int idx = i * 256 + tid, r = idx >> 5, piece = idx & 31;  // 64 pixels x 32 pieces (512 B)
      const int m = tile * 64 + r, y = m / W, x = m - y * W;
      out[((size_t)(2 * y + dy) * (2 * W) + 2 * x) * 16 + piece] = v;
    }
    return;
  }
  const int tile = mode == 1 ? blockIdx.x % 400 : blockIdx.x >> 2, quad = mode == 1 ? blockIdx.x / 400 : blockIdx.x & 3;
  const int dy = quad >> 1, dx = quad & 1;
  for (int i = 0; i < 8; ++i) {
    const int idx = i * 256 + tid, r = idx >> 4, piece = idx & 15;  // 128 pixels x 16 pieces
    const int m = tile * 128 + r, y = m / W, x = m - y * W;
    out[((size_t)(2 * y + dy) * (2 * W) + (2 * x + dx)) * 16 + piece] = v;
  }
}

int main() {
  u32x4* out;
  hipMalloc(&out, (size_t)320 * 640 * 256);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  const char* names[] = {"linear 32 KB per workgroup", "pixel shuffle, quadrant slowest", "pixel shuffle, quadrant fastest", "both dx quadrants per workgroup (512 B pieces)"};
  for (int mode = 0; mode < 4; ++mode) {
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(wr, dim3(1600), dim3(256), 0, 0, out, mode);
    hipEventRecord(a, 0);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(wr, dim3(1600), dim3(256), 0, 0, out, mode);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    std::printf("mode %d %-48s %6.1f us per 52.4 MB  = %5.2f TB/s\n", mode, names[mode], ms * 50.0f, 52.4288e6 / (ms * 50.0f) * 1e-6);
  }
  return 0;
}
