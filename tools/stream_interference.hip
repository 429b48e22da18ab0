// Developer tool: how much does a SMALL latency-bound kernel on another stream slow a machine-filling MFMA kernel down?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I autoware_vision_pilot_amd/csrc tools/stream_interference.hip -o tools/_stream_interference
// Stream A: 40 launches of the 128->128 320x640 3x3 layer (800 workgroups).  Stream B (optional): a loop of small kernels
// (GRID workgroups x 256 threads, each thread chasing a few dependent global loads then sleeping: ~10 us, like the encoder's
// depthwise / squeeze-excite launches).  Prints the time of A alone and of A with B running.
#include <cstdio>
#include <vector>

#include "../autoware_vision_pilot_amd/csrc/kernels_conv3x3.hip"
#include "../autoware_vision_pilot_amd/csrc/kernels_conv.hip"
#include "tool_ones.hpp"

using namespace vp;

__global__ __launch_bounds__(256) void small_kernel(const int* chain, int* out, int hops, int sleeps) {
  int i = (blockIdx.x * 256 + threadIdx.x) & 0xFFFFF;
  for (int h = 0; h < hops; ++h) i = chain[i];
  for (int s = 0; s < sleeps; ++s) __builtin_amdgcn_s_sleep(64);
  if (i == -1) out[0] = i;
}

int main() {
  const int H = 320, W = 640, C = 128;
  half_t *in, *out, *w;
  float* bias;
  hipMalloc(&in, (size_t)H * W * C * 2);
  hipMalloc(&out, (size_t)H * W * C * 2);
  hipMalloc(&w, (size_t)9 * C * C * 2);
  hipMalloc(&bias, C * 4);
  hipMemset(in, 0, (size_t)H * W * C * 2);
  hipMemset(w, 0, (size_t)9 * C * C * 2);
  hipMemset(bias, 0, C * 4);
  ConvGemmParams p{};
  p.in_hi = in; p.H = H; p.W = W; p.Cin = C; p.w_hi = w; p.bias = bias; p.wscale = tool_dev_ones(C); p.ks = 3; p.Ncols = C; p.CoutW = C;
  p.act = ACT_GELU_F16; p.out_hi = out; p.Cstore = C; p.Creal = C; p.nsplit = 1;
  int *chain, *sink;
  hipMalloc(&chain, (1 << 20) * 4);
  hipMalloc(&sink, 4);
  std::vector<int> hc(1 << 20);
  for (int i = 0; i < (1 << 20); ++i) hc[i] = (int)(((long long)i * 7919 + 104729) & 0xFFFFF);
  hipMemcpy(chain, hc.data(), hc.size() * 4, hipMemcpyHostToDevice);
  hipStream_t sa, sb;
  hipStreamCreateWithFlags(&sa, hipStreamNonBlocking);
  hipStreamCreateWithFlags(&sb, hipStreamNonBlocking);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  launch_conv3x3_halo(p, 0, false, sa);
  hipStreamSynchronize(sa);
  for (int grid : {0, 16, 64, 300, 1600}) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0, sa);
      for (int i = 0; i < 40; ++i) launch_conv3x3_halo(p, 0, false, sa);
      hipEventRecord(e1, sa);
      int nb = 0;
      if (grid > 0) {
        while (hipEventQuery(e1) == hipErrorNotReady) {  // keep stream B busy while A runs
          for (int k = 0; k < 8; ++k) hipLaunchKernelGGL(small_kernel, dim3(grid), dim3(256), 0, sb, chain, sink, 6, 20);
          hipStreamSynchronize(sb);
          nb += 8;
        }
      }
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      if (rep == 1) std::printf("stream B grid %4d (%4d small launches): 3x3 layer %6.1f us per launch\n", grid, nb, ms * 1000.0f / 40);
    }
  }
  return 0;
}
