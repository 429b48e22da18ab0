// ConvTranspose2d(k2,s2) [+ fused 1x1 skip-link conv] on the LARGE maps with SHORT K, weights stationary in REGISTERS
// (head stages: K = 128 channels on 160x320 pixels, K = 256 + 32 on 80x160; 52-105 MB of output per plane pair).
//
// These layers only move bytes (6.7-7.5 GFLOP against 73-131 MB in the parity mode), yet through the implicit-GEMM kernel
// (kernels_conv.hip) they ran at 1.4-2.1 TB/s: one workgroup per 128x128 output tile with a K loop of 4-9 steps is all
// prologue and epilogue, every input pixel is fetched once per pixel-shuffle quadrant, every workgroup re-stages the weights.
// Here the roles are turned round:
//   * a workgroup is PERSISTENT (one per CU, 8 waves) and every wave keeps ITS rows of the [4*Cout x K] weight matrix in
//     VGPRs as ready MFMA A-fragments for the whole launch (K = 128: 64 rows x 128 x (hi, lo) = 128 VGPRs -- the eight waves
//     hold all 512 rows, i.e. all four quadrants; K = 288: 32 rows = 144 VGPRs, the workgroup covers one quadrant);
//   * 32-pixel input tiles stream global -> LDS by LDS-DMA (global_load_lds_dwordx4: coalesced 1 KiB per wave instruction, no
//     VGPRs, no ds_write), three tiles deep: tile k+2 is requested when tile k starts; the tile rows are pitched at an ODD
//     number of 16-byte slots so the B-fragment ds_read_b128 of 32 different pixels are bank-conflict free;
//   * all eight waves read the SAME pixel fragments (2 ds_read_b128 per 16-channel step feed 3 x NT MFMAs); with K = 128 the
//     input is read from HBM exactly once;
//   * the weight rows are PERMUTED when the fragments are loaded so that a lane's 16 accumulators of a tile are 16 CONSECUTIVE
//     output channels of its pixel: the wave-private fp32 patch is written with ds_write_b128 and read back as 8-channel
//     pieces, eight (four) lanes covering 128 (64) contiguous bytes of one output pixel and plane; bias, (hi, lo) split and
//     the pixel-shuffle address reuse epilogue_store8;
//   * one barrier per tile; s_waitcnt vmcnt counts in issue order, so "tile k+1 has landed" = at most {stores of tile k-1,
//     DMA of tile k+2, stores of tile k} still outstanding -- the stores never stall the tile stream.
#include <cstdlib>

#include "conv_epilogue.hpp"
#include "lds_dma.hpp"

namespace vp {

namespace {
// s_waitcnt vmcnt(n) for a wave-uniform runtime n (the instruction takes an immediate); n above the table waits for 24.
__device__ __forceinline__ void wait_vmcnt_le(int n) {
#define VP_W(N) case N: VP_WAIT_VMCNT(N); break;
  switch (n) {
    VP_W(0) VP_W(1) VP_W(2) VP_W(3) VP_W(4) VP_W(5) VP_W(6) VP_W(7) VP_W(8) VP_W(9) VP_W(10) VP_W(11) VP_W(12)
    VP_W(13) VP_W(14) VP_W(15) VP_W(16) VP_W(17) VP_W(18) VP_W(19) VP_W(20) VP_W(21) VP_W(22) VP_W(23)
    default: VP_WAIT_VMCNT(24); break;
  }
#undef VP_W
}
}  // namespace

// KC1 / KC2: 16-channel MFMA steps of the ConvTranspose input / of the skip tensor (K extension); NT: 32-row weight tiles per wave.
template <int KC1, int KC2, int NT, bool SPLIT>
__global__ __launch_bounds__(512, 2) void convt_rs_kernel(const ConvGemmParams p, const int n_groups) {
  constexpr int KC = KC1 + KC2, K = 16 * KC, CH = 2 * KC, CH1 = 2 * KC1;  // CH: 16-byte slots of one pixel row and plane
  constexpr int PITCHC = CH + 1, PITCH = PITCHC * 16;                      // odd slot pitch
  constexpr int PLANES = SPLIT ? 2 : 1;
  constexpr int SLOTS = 32 * PITCHC, NI = (SLOTS + 63) / 64;               // DMA instructions per plane (the last one partly padding)
  constexpr int PLANE_BYTES = NI * 1024, TILE_BYTES = PLANES * PLANE_BYTES;
  constexpr int NINSTR = PLANES * NI, DHI = (NINSTR + 7) / 8, NWHI = NINSTR - 8 * (DHI - 1);  // waves [0, NWHI) issue DHI, the others DHI - 1
  constexpr int COW = 32 * NT, PP = COW * 4 + 16, PATCH_BYTES = 32 * PP;   // wave's channels; fp32 patch [32 px][COW] (+16: odd slot pitch)
  constexpr int CPR = COW / 8, RPP = 64 / CPR, PASSES = 32 / RPP;          // 8-channel pieces per patch row; rows per read-back pass
  constexpr int S = PASSES * PLANES;                                       // global stores per wave and tile
  constexpr int NCH = NT == 1 ? 2 : 1;                                     // accumulator chains per tile (a lone chain serialises on the MFMA latency)
  static_assert(2 * S + DHI <= 24, "wait table");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const tiles = smem;                         // [3][PLANES][PLANE_BYTES]
  char* const patches = smem + 3 * TILE_BYTES;      // [8 waves][PATCH_BYTES]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int vid;  // XCD-aware: the workgroups that walk the SAME pixel tiles (one per weight slice) sit on one XCD's L2
  {
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = blockIdx.x & 7;
    vid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3);
  }
  const int n_slices = p.Ncols / (8 * COW);
  const int slice = vid % n_slices, group = vid / n_slices;
  const int n0 = slice * 8 * COW + wave * COW;                 // first GEMM column (= weight row) of this wave
  const int quad = n0 / p.Cstore, cq = n0 - quad * p.Cstore;   // the wave's channels lie inside one quadrant (launcher checks)
  const int dy = quad >> 1, dx = quad & 1;
  const int qs = (slice * 8 * COW) / p.Cstore, sdy = qs >> 1, sdx = qs & 1;  // skip tensor: the WORKGROUP lies inside one quadrant
  const int M = p.H * p.W, n_tiles = M >> 5;
  const int ntl = group < n_tiles ? (n_tiles - group + n_groups - 1) / n_groups : 0;  // tiles group, group + n_groups, ...

  // ---- DMA plan of this thread: instruction ii = wave + 8 i moves slots [64 j, 64 j + 64) of plane ii / NI
  const half_t* g_src[DHI];
  bool g_skip[DHI];
#pragma unroll
  for (int i = 0; i < DHI; ++i) {
    const int ii = wave + 8 * i;
    const int pl = ii < NINSTR ? ii / NI : 0, j = ii < NINSTR ? ii - pl * NI : 0;
    const int slot = 64 * j + lane;
    int r = slot / PITCHC, col = slot - r * PITCHC;
    if (slot >= SLOTS || col >= CH) { r = 0; col = 0; }  // pad slot: any valid address, never read
    const bool sk = KC2 > 0 && col >= CH1;
    const half_t* base = (SPLIT && pl == 1) ? p.in_lo : p.in_hi;
    if (sk) base += (SPLIT && pl == 1) ? p.in2_delta_lo : p.in2_delta_hi;
    g_src[i] = base + (sk ? (long long)2 * r * p.Cin2 + (col - CH1) * 8 : (long long)r * p.Cin + col * 8);
    g_skip[i] = sk;
  }
#define VP_DMA_TILE(TI, BUF)                                                                                   \
  {                                                                                                            \
    const int m0_ = (TI) << 5;                                                                                 \
    const int y_ = m0_ / p.W, x0_ = m0_ - y_ * p.W;                                                            \
    const long long main_ = (long long)m0_ * p.Cin;                                                            \
    const long long skip_ = KC2 > 0 ? ((long long)(2 * y_ + sdy) * (2 * p.W) + (2 * x0_ + sdx)) * p.Cin2 : 0;  \
    _Pragma("unroll") for (int i = 0; i < DHI; ++i) {                                                          \
      const int ii = wave + 8 * i;                                                                             \
      if (ii < NINSTR) VP_GLOBAL_LOAD_LDS16(g_src[i] + (g_skip[i] ? skip_ : main_), tiles + (BUF) * TILE_BYTES + ii * 1024); \
    }                                                                                                          \
  }
  if (ntl > 0) VP_DMA_TILE(group, 0)
  if (ntl > 1) VP_DMA_TILE(group + n_groups, 1)

  // ---- stationary weights: MFMA row rho = 8 g + 4 h + i of tile T holds weight row n0 + 16 NT h + 16 T + 4 g + i, so that
  // accumulator register 4 g + i of lane (pixel, h) is output channel 16 NT h + 16 T + (4 g + i): 16 consecutive channels
  h8_t ah[NT][KC], al[SPLIT ? NT : 1][SPLIT ? KC : 1];
  {
    const int rho = lane & 31, g = rho >> 3, h = (rho >> 2) & 1, i = rho & 3;
#pragma unroll
    for (int T = 0; T < NT; ++T) {
      const size_t row = (size_t)(n0 + 16 * NT * h + 16 * T + 4 * g + i) * K + (lane >> 5) * 8;
#pragma unroll
      for (int c = 0; c < KC; ++c) {
        ah[T][c] = *reinterpret_cast<const h8_t*>(p.w_hi + row + 16 * c);
        if constexpr (SPLIT) al[T][c] = *reinterpret_cast<const h8_t*>(p.w_lo + row + 16 * c);
      }
    }
  }
  // read-back role of the lane: 8-channel piece pc of patch rows r0, r0 + RPP, ...
  const int pc = lane % CPR, r0 = lane / CPR;
  const int co = n0 + pc * 8;
  const f32x4_t b0 = *reinterpret_cast<const f32x4_t*>(p.bias + co), b1 = *reinterpret_cast<const f32x4_t*>(p.bias + co + 4);
  const f32x4_t ws0 = *reinterpret_cast<const f32x4_t*>(p.wscale + co), ws1 = *reinterpret_cast<const f32x4_t*>(p.wscale + co + 4);
  char* const mypatch = patches + wave * PATCH_BYTES;
  char* const pw = mypatch + (lane & 31) * PP + (16 * NT * (lane >> 5)) * 4;  // + 64 T + 16 g
  const char* const pr = mypatch + r0 * PP + pc * 32;                          // + pass * RPP * PP
  const int b_ofs = (lane & 31) * PITCH + (lane >> 5) * 16;
  const int d_w = wave < NWHI ? DHI : DHI - 1;

  VP_WAIT_VMCNT(0);
  __syncthreads();
  for (int k = 0; k < ntl; ++k) {
    // tile k is resident and visible; tile k+1 is in flight; the buffer of tile k+2 was last read in iteration k-1
    const bool ahead = k + 2 < ntl;
    if (ahead) VP_DMA_TILE(group + (k + 2) * n_groups, (k + 2) % 3)
    const char* tb = tiles + (k % 3) * TILE_BYTES + b_ofs;
    f32x16_t acc[NT][NCH];
#pragma unroll
    for (int T = 0; T < NT; ++T)
#pragma unroll
      for (int n = 0; n < NCH; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[T][n][r] = 0.0f;
#pragma unroll
    for (int c = 0; c < KC; ++c) {
      const h8_t bh = *reinterpret_cast<const h8_t*>(tb + c * 32);
      h8_t bl;
      if constexpr (SPLIT) bl = *reinterpret_cast<const h8_t*>(tb + PLANE_BYTES + c * 32);
#pragma unroll
      for (int T = 0; T < NT; ++T) {
        f32x16_t& a = acc[T][NCH == 2 ? (c & 1) : 0];
        if constexpr (SPLIT) {
          a = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[T][c], bh, a, 0, 0, 0);
          a = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[T][c], bl, a, 0, 0, 0);
        }
        a = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[T][c], bh, a, 0, 0, 0);
      }
    }
    // ---- wave-private epilogue
#pragma unroll
    for (int T = 0; T < NT; ++T)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4_t v;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = NCH == 2 ? acc[T][0][4 * g + r] + acc[T][NCH - 1][4 * g + r] : acc[T][0][4 * g + r];
        *reinterpret_cast<f32x4_t*>(pw + 64 * T + 16 * g) = v;
      }
    // same wave writes and reads the patch: LDS operations of a wave complete in order (the wave barrier emits no instruction;
    // it states the dependency for the compiler and for the CPU emulation, tests/emul)
    __builtin_amdgcn_wave_barrier();
    {
      const int m0 = (group + k * n_groups) << 5;
      const int y = m0 / p.W, x0 = m0 - y * p.W;  // W % 32 == 0: the 32 pixels of a tile share an image row
      const long long o0 = ((long long)(2 * y + dy) * (2 * p.W) + (2 * (x0 + r0) + dx)) * p.Cstore + cq + pc * 8;
#pragma unroll
      for (int pass = 0; pass < PASSES; ++pass) {
        const f32x4_t s0 = *reinterpret_cast<const f32x4_t*>(pr + pass * RPP * PP);
        const f32x4_t s1 = *reinterpret_cast<const f32x4_t*>(pr + pass * RPP * PP + 16);
        float v[8] = {s0[0], s0[1], s0[2], s0[3], s1[0], s1[1], s1[2], s1[3]};
        epilogue_store8<STORE_SHUFFLE2, RES_NONE, ACT_NONE>(p, M, m0 + r0 + pass * RPP, co, v, b0, b1, ws0, ws1, o0 + (long long)2 * pass * RPP * p.Cstore);
      }
    }
    __builtin_amdgcn_wave_barrier();
    if (k + 1 < ntl) {
      wait_vmcnt_le((k > 0 ? S : 0) + (ahead ? d_w : 0) + S);  // everything issued after tile k+1's DMA may stay in flight
      VP_LDS_BARRIER();  // not __syncthreads(): that would drain vmcnt(0) -- the stores and tile k+2 (lds_dma.hpp)
    }
  }
#undef VP_DMA_TILE
}

int convt_rs_shape_case(int H, int W, int cin_pad, int cin2_pad, int ncols, int cstore);

namespace {
template <int KC1, int KC2, int NT, bool SPLIT>
hipError_t launch_rs_cfg(const ConvGemmParams& p, hipStream_t st) {
  constexpr int KC = KC1 + KC2, PITCHC = 2 * KC + 1, NI = (32 * PITCHC + 63) / 64;
  constexpr int lds = 3 * (SPLIT ? 2 : 1) * NI * 1024 + 8 * 32 * (32 * NT * 4 + 16);
  static_assert(lds <= 160 * 1024, "LDS budget");
  auto k = convt_rs_kernel<KC1, KC2, NT, SPLIT>;
  static LdsAttrOnce attr_once;
  if (hipError_t e = set_max_dynamic_lds(attr_once, reinterpret_cast<const void*>(k), lds); e != hipSuccess) return e;
  const int n_tiles = (p.H * p.W) >> 5, slices = p.Ncols / (256 * NT);
  // persistent, one workgroup per CU: the pixel tiles are dealt round-robin to 256 / slices groups
  int groups = std::max(1, std::min(n_tiles, 256 / slices));
  if (p.rs_groups > 0) groups = std::max(1, std::min(groups, p.rs_groups));  // developer / test knob (VP_CONVT_RS_GROUPS, read when the plan is built): more tiles per workgroup
  hipLaunchKernelGGL(k, dim3(slices * groups), dim3(512), lds, st, p, groups);
  return hipGetLastError();
}
// 0 = not covered; else 1: K = 128 (all four quadrants per workgroup), 2: K = 256 + 32 with the skip link (one quadrant per workgroup)
int rs_case(const ConvGemmParams& p) {
  if (p.ks != 1 || p.stride > 1 || p.store_mode != STORE_SHUFFLE2 || p.act != ACT_NONE || p.res_mode != RES_NONE || p.nsplit != 1 ||
      p.post_act != ACT_NONE || p.out_hi == nullptr || p.CoutW != p.Ncols)
    return 0;
  return convt_rs_shape_case(p.H, p.W, p.Cin, p.Cin2, p.Ncols, p.Cstore);
}
}  // namespace

// the shape part of the test (the engine asks before it packs the weights)
int convt_rs_shape_case(int H, int W, int cin_pad, int cin2_pad, int ncols, int cstore) {
  if (W % 32 != 0 || H * W < 2048) return 0;
  if (cin_pad == 128 && cin2_pad == 0 && ncols % 512 == 0 && cstore % 64 == 0) return 1;
  if (cin_pad == 256 && cin2_pad == 32 && ncols % 256 == 0 && cstore % 256 == 0) return 2;
  return 0;
}

bool convt_rs_supported(const ConvGemmParams& p, bool split) { return p.w_hi != nullptr && split == (p.in_lo != nullptr) && split == (p.out_lo != nullptr) && rs_case(p) != 0; }

hipError_t launch_convt_rs(const ConvGemmParams& p, hipStream_t st) {
  const bool split = p.in_lo != nullptr;
  switch (rs_case(p)) {
    case 1: return split ? launch_rs_cfg<8, 0, 2, true>(p, st) : launch_rs_cfg<8, 0, 2, false>(p, st);
    case 2: return split ? launch_rs_cfg<16, 2, 1, true>(p, st) : launch_rs_cfg<16, 2, 1, false>(p, st);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace vp
