// ConvTranspose2d(k2,s2) [+ fused 1x1 skip-link conv] for the LARGE maps with SHORT K (head stages: K = 128 and 256+32
// channels, 51 200 / 12 800 input pixels -> 52 / 26 MB of output): a streaming kernel.
//
// The implicit-GEMM kernel (kernels_conv.hip) launches one workgroup per 128x128 output tile; with K this short a
// workgroup is mostly fixed cost (index set-up, pipeline fill, two barriers per K step, epilogue), measured 1.2-1.9 TB/s
// on layers that only move bytes (PMC: ~1000 VALU + 380 SALU instructions per wave for 32 MFMAs).  Here a workgroup is
// PERSISTENT: it keeps the weights of one 64-row slice of the [4*Cout x K] matrix resident in LDS and streams 64-pixel
// tiles through a double-buffered LDS tile (next tile's global loads in flight during the current tile's MFMAs), one
// barrier per tile; every wave converts and stores its own 32x32 result through a wave-private LDS patch (no second
// barrier), pixel-shuffle addresses advance incrementally (no divisions in the loop).
#include <cstdlib>

#include "conv_epilogue.hpp"

namespace vp {

// DEPTH: pixel tiles in flight per workgroup (register ring): one tile of look-ahead is ~0.3 us of work against ~2 us of
// HBM latency, so the loop was latency-bound (2 us per tile) until the ring went 4 deep.
template <int KT, int DEPTH, bool SKIP>  // KT = K / 16 (Cin + Cin2 in 16-channel MFMA steps); SKIP: K extension present
__global__ __launch_bounds__(256) void convt_stream_kernel(const ConvGemmParams p, int tiles_per_slice) {
  constexpr int K = KT * 16, PITCH = K * 2 + 16, CO = 64, PX = 64;
  constexpr int PIECES = PX * (K / 8), XP = (PIECES + 255) / 256;  // 16-byte pieces of one pixel tile per thread
  constexpr int WPIECES = CO * (K / 8), WPP = (WPIECES + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const wlds = smem;                          // [CO][PITCH]
  char* const xlds = smem + CO * PITCH;             // [2][PX][PITCH]
  char* const patch = xlds + 2 * PX * PITCH;        // [4 waves][32 px][80 B]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wco = wave >> 1, wpx = wave & 1;
  const int M = p.H * p.W;
  const int n_tiles = (M + PX - 1) / PX;
  // XCD-aware: the workgroups that walk the SAME pixel tiles (one per weight slice) get consecutive virtual ids, i.e. the
  // same XCD, so the pixel tile is fetched from HBM once and hit in that L2 by the other slices
  int vid;
  {
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = blockIdx.x & 7;
    vid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3);
  }
  const int n_slices = p.Ncols / CO;
  const int slice = vid % n_slices, part = vid / n_slices;
  const int co0 = slice * CO;
  const int quad = co0 / p.Cstore, cq = co0 - quad * p.Cstore;  // the 64-row slice lies inside one quadrant (launcher checks)
  const int dy = quad >> 1, dx = quad & 1;
  const int kc1 = p.Cin / 8;  // pieces coming from the ConvTranspose input; the rest from the skip tensor

  // ---- resident weights: rows co0 .. co0+63 of the [CoutW][K] matrix
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
  for (int i = 0; i < WPP; ++i) {
    const int idx = tid + 256 * i;
    if (idx < WPIECES) {
      const int row = idx / (K / 8), pc = idx - row * (K / 8);
      *reinterpret_cast<u32x4*>(wlds + row * PITCH + pc * 16) = *reinterpret_cast<const u32x4*>(p.w_hi + (size_t)(co0 + row) * K + pc * 8);
    }
  }

  // ---- per-thread staging pattern of a pixel tile (fixed across tiles): piece idx -> (row, piece in row)
  int x_row[XP], x_pc[XP];
#pragma unroll
  for (int i = 0; i < XP; ++i) {
    const int idx = tid + 256 * i;
    x_row[i] = idx < PIECES ? idx / (K / 8) : -1;
    x_pc[i] = idx < PIECES ? idx - (idx / (K / 8)) * (K / 8) : 0;
  }
  u32x4 xr[DEPTH][XP];
  bool px_ok[DEPTH][XP];
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
  // (y, x) of every staged piece's pixel and of this lane's two epilogue rows, advanced INCREMENTALLY from tile to tile
  // (tile index grows by tiles_per_slice -> pixel index by a constant): no division in the loop
  const int adv = tiles_per_slice * PX, adv_y = adv / p.W, adv_x = adv - adv_y * p.W;
  int px_m[XP], px_y[XP], px_x[XP];
#pragma unroll
  for (int i = 0; i < XP; ++i) {
    px_m[i] = part * PX + (x_row[i] >= 0 ? x_row[i] : 0);
    px_y[i] = px_m[i] / p.W;
    px_x[i] = px_m[i] - px_y[i] * p.W;
  }
  int ep_m[2], ep_y[2], ep_x[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    ep_m[i] = part * PX + wpx * 32 + ((lane + 64 * i) >> 2);
    ep_y[i] = ep_m[i] / p.W;
    ep_x[i] = ep_m[i] - ep_y[i] * p.W;
  }
  // macros, not lambdas: closures over register arrays end up in scratch memory
#define VP_XFETCH(SLOT)                                                                                    \
  _Pragma("unroll") for (int i = 0; i < XP; ++i) {                                                         \
    const bool ok_ = x_row[i] >= 0 && px_m[i] < M;                                                         \
    const int o1_ = (ok_ ? px_m[i] : 0) * p.Cin + x_pc[i] * 8; /* tensors are < 2^31 elements */           \
    if constexpr (SKIP) {                                                                                  \
      const long long o2_ = p.in2_delta_hi + ((long long)(2 * (ok_ ? px_y[i] : 0) + dy) * (2 * p.W) + (2 * (ok_ ? px_x[i] : 0) + dx)) * p.Cin2 + \
                            (x_pc[i] - kc1) * 8;                                                           \
      xr[SLOT][i] = *reinterpret_cast<const u32x4*>(p.in_hi + (x_pc[i] < kc1 ? (long long)o1_ : o2_));     \
      px_y[i] += adv_y;                                                                                    \
      px_x[i] += adv_x;                                                                                    \
      if (px_x[i] >= p.W) { px_x[i] -= p.W; ++px_y[i]; }                                                   \
    } else {                                                                                               \
      xr[SLOT][i] = *reinterpret_cast<const u32x4*>(p.in_hi + o1_);                                        \
    }                                                                                                      \
    px_ok[SLOT][i] = ok_;                                                                                  \
    px_m[i] += adv;                                                                                        \
  }
#define VP_XCOMMIT(SLOT, BUF)                                                                                 \
  _Pragma("unroll") for (int i = 0; i < XP; ++i) if (x_row[i] >= 0)                                        \
    *reinterpret_cast<u32x4*>(xlds + ((BUF) * PX + x_row[i]) * PITCH + x_pc[i] * 16) = px_ok[SLOT][i] ? xr[SLOT][i] : zero4;

  // fragment addressing
  const int a_ofs = (wco * 32 + (lane & 31)) * PITCH + (lane >> 5) * 16;
  const int b_ofs = (wpx * 32 + (lane & 31)) * PITCH + (lane >> 5) * 16;
  char* const mypatch = patch + wave * 32 * 80;
  // bias of this lane's channels: register group g holds channels 8g + 4*(lane>>5) + r of the wave's 32
  f32x4_t bias4[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) bias4[g] = *reinterpret_cast<const f32x4_t*>(p.bias + co0 + wco * 32 + 8 * g + 4 * (lane >> 5));

#pragma unroll
  for (int d = 0; d < DEPTH; ++d)
    if (part + d * tiles_per_slice < n_tiles) VP_XFETCH(d)
  int buf = 0;
  for (int base = part; base < n_tiles; base += DEPTH * tiles_per_slice) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const int t = base + d * tiles_per_slice;
      if (t < n_tiles) {
        VP_XCOMMIT(d, buf)
        __syncthreads();  // tile t (and, first time round, the weights) visible; everyone is done with the buffer's previous tile
        if (t + DEPTH * tiles_per_slice < n_tiles) VP_XFETCH(d)
        f32x16_t acc, acc1;  // two chains (even / odd K steps): a single accumulator serialises on the MFMA latency
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = acc1[r] = 0.0f;
        const char* xb = xlds + buf * PX * PITCH;
#pragma unroll
        for (int k = 0; k < KT; ++k) {
          const h8_t a = *reinterpret_cast<const h8_t*>(wlds + a_ofs + k * 32);
          const h8_t b = *reinterpret_cast<const h8_t*>(xb + b_ofs + k * 32);
          if (k & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc1, 0, 0, 0);
          else acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] += acc1[r];
        // ---- wave-private epilogue: bias, fp16, [32 px][32 co] patch (80-byte pitch), then 16-byte pieces to the pixel-shuffled image
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          h4_t h;
#pragma unroll
          for (int r = 0; r < 4; ++r) h[r] = (half_t)(acc[4 * g + r] + bias4[g][r]);
          *reinterpret_cast<h4_t*>(mypatch + (lane & 31) * 80 + (8 * g + 4 * (lane >> 5)) * 2) = h;
        }
        // same wave wrote and reads the patch: LDS operations of a wave complete in order.  The wave barrier emits no
        // instruction (the ISA is unchanged); it states the dependency for the compiler and for the CPU emulation (tests/emul).
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int idx = lane + 64 * i, r = idx >> 2, c8 = idx & 3;  // 32 rows x 4 pieces of 8 channels
          if (ep_m[i] < M) {
            const int o = ((2 * ep_y[i] + dy) * (2 * p.W) + (2 * ep_x[i] + dx)) * p.Cstore + cq + wco * 32 + c8 * 8;
            *reinterpret_cast<h8_t*>(p.out_hi + o) = *reinterpret_cast<const h8_t*>(mypatch + r * 80 + c8 * 16);
          }
          ep_m[i] += adv;
          ep_y[i] += adv_y;
          ep_x[i] += adv_x;
          if (ep_x[i] >= p.W) { ep_x[i] -= p.W; ++ep_y[i]; }
        }
        buf ^= 1;
      }
    }
  }
#undef VP_XFETCH
#undef VP_XCOMMIT
}

template <int KT, int DEPTH, bool SKIP>
static hipError_t launch_stream_cfg(const ConvGemmParams& p, hipStream_t st) {
  constexpr int K = KT * 16, PITCH = K * 2 + 16;
  constexpr int lds = 64 * PITCH + 2 * 64 * PITCH + 4 * 32 * 80;
  static_assert(lds <= 160 * 1024, "LDS budget");
  auto k = convt_stream_kernel<KT, DEPTH, SKIP>;
  static LdsAttrOnce attr_once;
  if (hipError_t e = set_max_dynamic_lds(attr_once, reinterpret_cast<const void*>(k), lds); e != hipSuccess) return e;
  const int M = p.H * p.W, n_tiles = (M + 63) / 64, slices = p.Ncols / 64;
  // persistent: as many workgroups as the LDS plan keeps resident (no second round), at least 4 pixel tiles each
  const int resident = 256 * std::max(1, (160 * 1024) / lds);
  int per_slice = std::max(1, std::min(n_tiles / 4, resident / std::max(1, slices)));
  hipLaunchKernelGGL(k, dim3(slices * per_slice), dim3(256), lds, st, p, per_slice);
  return hipGetLastError();
}

bool convt_stream_supported(const ConvGemmParams& p, bool split) {
  const int K = p.Cin + p.Cin2;
  return !split && p.ks == 1 && p.store_mode == STORE_SHUFFLE2 && p.act == ACT_NONE && p.res_mode == RES_NONE && p.nsplit == 1 &&
         p.out_lo == nullptr && p.post_act == ACT_NONE && p.Cstore % 64 == 0 && p.Ncols % 64 == 0 && p.CoutW == p.Ncols &&
         ((K == 128 && p.Cin2 == 0) || (K == 288 && p.Cin2 > 0 && std::getenv("VP_CONVT_STREAM_K288"))) && p.H * p.W >= 2048;
  // K = 288 (up-sampling stage 3 with its skip link): 123 KiB of LDS -> one workgroup per CU, 35 us vs 29 us for the GEMM kernel: opt-in only
}
hipError_t launch_convt_stream(const ConvGemmParams& p, hipStream_t st) {
  const int K = p.Cin + p.Cin2;
  if (K == 128 && p.Cin2 == 0) return launch_stream_cfg<8, 4, false>(p, st);
  if (K == 288 && p.Cin2 > 0) return launch_stream_cfg<18, 2, true>(p, st);
  return hipErrorInvalidValue;
}

}  // namespace vp
