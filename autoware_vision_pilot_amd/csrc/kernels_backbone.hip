// EfficientNet-B0 encoder pieces that are NOT GEMMs (torchvision efficientnet_b0().features as used by
// Models/model_components/backbone.py:9-22) and the context MLP (scene_context.py:27-38).  All HBM / latency bound:
// the rules that matter are 16-byte accesses along the channel axis, enough independent loads in flight per lane,
// and as few launches as possible (every launch costs >= ~4 us of the 2 ms frame):
//   stem            conv 3x3/s2 3->32 + BN + SiLU, fp32 planes in -> NHWC out
//   dwconv_pool     depthwise k x k (+BN folded) + SiLU, FUSED with the squeeze-excite average pool: each
//                   workgroup owns a pixel slab x channel group and emits its per-channel partial sums
//                   (deterministic two-level reduction, no atomics, the tensor is not re-read)
//   se_gate_scale   the squeeze-excite tail in one launch: means -> squeeze FC + SiLU -> excite FC + sigmoid -> per-frame
//                   scaling of the projection weights
//   pool_partial    stand-alone channel sums (context block's global average pool)
//   fc              dense layer of the context MLP, one wave per two outputs, input vector staged in LDS
#include "act_io.hpp"
#include "se_phases.hpp"

namespace vp {

// -------------------------------------------------------------------------------------------------- stem
__global__ __launch_bounds__(256) void stem_kernel(const StemParams p) {
  __shared__ float ws[27 * 32 + 32];
  for (int i = threadIdx.x; i < 27 * 32 + 32; i += 256) ws[i] = i < 27 * 32 ? p.w[i] : p.b[i - 27 * 32];
  __syncthreads();
  const int OH = p.H / 2, OW = p.W / 2;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (p.zero) {  // the frame's squeeze-excite accumulators (was a launch of its own, 4.8 us of the replayed graph)
    typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
    const size_t n2 = p.zero_n >> 1, step = (size_t)gridDim.x * 256;
    for (size_t i = t; i < n2; i += step) reinterpret_cast<u64x2*>(p.zero)[i] = u64x2{0ull, 0ull};
  }
  const int g = t & 3, pix = t >> 2;  // 4 threads per pixel, 8 output channels each
  if (pix >= OH * OW) return;
  const int oy = pix / OW, ox = pix - oy * OW;
  float in[27];
#pragma unroll
  for (int ci = 0; ci < 3; ++ci)
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = 2 * oy + ky - 1;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int ix = 2 * ox + kx - 1;
        const bool ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        const float v = p.in[((size_t)ci * p.H + (ok ? iy : 0)) * p.W + (ok ? ix : 0)];
        in[(ci * 3 + ky) * 3 + kx] = ok ? v : 0.0f;
      }
    }
  float acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = ws[27 * 32 + g * 8 + i];
#pragma unroll
  for (int k = 0; k < 27; ++k) {
    const float* wk = ws + k * 32 + g * 8;
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = fmaf(in[k], wk[i], acc[i]);
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = silu_f16(acc[i]);  // x * rcp(1 + exp2(-x log2 e)) in fp32 (v_exp_f32 / v_rcp_f32, |error| <= 3e-7 |silu|): every mode since round 3 -- libm's expf + IEEE division cost 3-9 us per launch (tools/mbf_check.hip)
  store8(p.out, (size_t)pix * 32 + g * 8, acc);
}

// ------------------------------------------------------------------------- depthwise conv + SiLU + pool sums
// One output pixel x 8 channels per thread.  A workgroup owns G channel-octets (G = the largest divisor of C/8 that
// is <= 8, so no lane idles on a ragged channel count) x 256/G consecutive output pixels: a wave's 16-byte accesses cover
// whole 128-byte runs, the K*K x 8G filter slice is staged ONCE per workgroup in LDS (it used to be 2*K*K dependent
// 16-byte global loads per thread -- a serialised latency chain longer than the arithmetic), and the fused squeeze-excite
// pool reduces 256/G pixels per channel in LDS before touching global memory.  All K*K taps are loaded before the first
// FMA (latency-bound layers; branch-free: clamped address + mask) and overlap the filter staging.  Pool: every thread
// adds its 8 activations as 2^24 fixed-point int64 into per-channel LDS accumulators, then one global atomicAdd per
// channel per workgroup into one of `replicas` rows.  Integer addition is associative, so the result is bit-identical
// run to run whatever the order (fp32 atomics would not be).
constexpr float kPoolFix = 16777216.0f;  // 2^24

static int dw_octets_per_group(int CG) {
  for (int g = 8; g > 1; --g)
    if (CG % g == 0) return g;
  return 1;
}

// BATCH (batched encoder): grid.z = camera frame; the same code on that frame's slice of [frames][H][W][C] tensors.
template <int K, bool SPLIT, bool BATCH = false>
__global__ __launch_bounds__(256) void dwconv_pool_kernel(const DwParams pin, const int G) {
  DwParams p = pin;
  if constexpr (BATCH) {
    const size_t f = blockIdx.z, in_fs = (size_t)p.in.H * p.in.W * p.in.C, out_fs = (size_t)p.out.H * p.out.W * p.out.C;
    p.in.hi += f * in_fs;
    if constexpr (SPLIT) p.in.lo += f * in_fs;
    p.out.hi += f * out_fs;
    if constexpr (SPLIT) p.out.lo += f * out_fs;
    p.sums += f * (size_t)p.replicas * p.in.C;
  }
  extern __shared__ unsigned char dw_smem[];
  const int GC = G * 8;                                            // channels of this workgroup
  float* wl = reinterpret_cast<float*>(dw_smem);                   // [K*K][GC]
  unsigned long long* red64 = reinterpret_cast<unsigned long long*>(dw_smem + (size_t)K * K * GC * sizeof(float));  // [GC]
  const int C = p.in.C, c0 = blockIdx.y * GC;
  const int OW = p.out.W, HWo = p.out.H * OW;
  const int PXB = 256 / G;
  const int pl = threadIdx.x / G, og = threadIdx.x - pl * G;
  const int pix = blockIdx.x * PXB + pl;
  const bool live = pl < PXB && pix < HWo;
  const int cc = c0 + og * 8;
  constexpr int pad = (K - 1) / 2;

  h8_t vh[K * K], vl[SPLIT ? K * K : 1];
  bool ok[K * K];
  f32x4_t b0, b1;
  if (live) {
    const int oy = pix / OW, ox = pix - oy * OW;
#pragma unroll
    for (int ky = 0; ky < K; ++ky) {
      const int iy = oy * p.stride + ky - pad;
      const bool yok = (unsigned)iy < (unsigned)p.in.H;
#pragma unroll
      for (int kx = 0; kx < K; ++kx) {
        const int ix = ox * p.stride + kx - pad;
        const bool o = yok && (unsigned)ix < (unsigned)p.in.W;
        const size_t off = ((size_t)(o ? iy : 0) * p.in.W + (o ? ix : 0)) * C + cc;
        ok[ky * K + kx] = o;
        vh[ky * K + kx] = *reinterpret_cast<const h8_t*>(p.in.hi + off);
        if constexpr (SPLIT) vl[ky * K + kx] = *reinterpret_cast<const h8_t*>(p.in.lo + off);
      }
    }
    b0 = *reinterpret_cast<const f32x4_t*>(p.b + cc);
    b1 = *reinterpret_cast<const f32x4_t*>(p.b + cc + 4);
  }
  for (int i = threadIdx.x; i < K * K * G * 2; i += 256) {
    const int tp = i / (G * 2), j = i - tp * (G * 2);
    *reinterpret_cast<f32x4_t*>(wl + tp * GC + j * 4) = *reinterpret_cast<const f32x4_t*>(p.w + (size_t)tp * C + c0 + j * 4);
  }
  if (threadIdx.x < GC) red64[threadIdx.x] = 0ull;
  __syncthreads();
  if (live) {
    float acc[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) { acc[i] = b0[i]; acc[4 + i] = b1[i]; }
#pragma unroll
    for (int tp = 0; tp < K * K; ++tp) {
      const float* wk = wl + tp * GC + og * 8;
      const f32x4_t w0 = *reinterpret_cast<const f32x4_t*>(wk), w1 = *reinterpret_cast<const f32x4_t*>(wk + 4);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        float v = (float)vh[tp][i];
        if constexpr (SPLIT) v += (float)vl[tp][i];
        v = ok[tp] ? v : 0.0f;
        acc[i] = fmaf(v, i < 4 ? w0[i] : w1[i - 4], acc[i]);
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = silu_f16(acc[i]);  // as in stem_kernel
    store8(p.out, (size_t)pix * p.out.C + cc, acc);
#pragma unroll
    for (int i = 0; i < 8; ++i) atomicAdd(&red64[og * 8 + i], (unsigned long long)(long long)__float2ll_rn(acc[i] * kPoolFix));
  }
  __syncthreads();
  if (threadIdx.x < GC) {
    const unsigned long long v = red64[threadIdx.x];
    if (v != 0ull) atomicAdd(p.sums + (size_t)(blockIdx.x & (p.replicas - 1)) * C + c0 + threadIdx.x, v);
  }
}

// ------------------------------------------------------------------------------- stand-alone channel sums
__global__ __launch_bounds__(256) void pool_partial_kernel(const PoolParams p) {
  __shared__ float red[256 * 8];
  const int CG = p.in.C >> 3;
  const int HW = p.in.H * p.in.W;
  const int CGL = CG >= 32 ? 32 : (CG >= 16 ? 16 : (CG >= 8 ? 8 : 4));
  const int PXL = 256 / CGL;
  const int cl = threadIdx.x % CGL, pl = threadIdx.x / CGL;
  const int cg = blockIdx.y * CGL + cl;
  const int per = (HW + p.nslab - 1) / p.nslab;
  const int begin = blockIdx.x * per, end = min(begin + per, HW);
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (cg < CG) {
    int px = begin + pl;
    for (; px + 3 * PXL < end; px += 4 * PXL) {
      float v0[8], v1[8], v2[8], v3[8];
      load8(p.in, (size_t)px * p.in.C + cg * 8, v0);
      load8(p.in, (size_t)(px + PXL) * p.in.C + cg * 8, v1);
      load8(p.in, (size_t)(px + 2 * PXL) * p.in.C + cg * 8, v2);
      load8(p.in, (size_t)(px + 3 * PXL) * p.in.C + cg * 8, v3);
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] += (v0[i] + v1[i]) + (v2[i] + v3[i]);
    }
    for (; px < end; px += PXL) {
      float v[8];
      load8(p.in, (size_t)px * p.in.C + cg * 8, v);
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] += v[i];
    }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) red[threadIdx.x * 8 + i] = acc[i];
  __syncthreads();
  if (pl == 0 && cg < CG) {
    for (int q = 1; q < PXL; ++q)
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] += red[(q * CGL + cl) * 8 + i];
#pragma unroll
    for (int i = 0; i < 8; ++i) p.partial[(size_t)blockIdx.x * p.in.C + cg * 8 + i] = acc[i];
  }
}

// ------------------------------------------------------------------------------------- squeeze-excite tail
// The whole squeeze-excite tail of an MBConv block in ONE launch: channel means from the replica rows of the fused pool ->
// squeeze FC + SiLU -> excite FC + sigmoid -> the gate folded into the K axis of the following 1x1 projection (batch is 1, so
// W'[n][k] = W[n][k] * gate[k] and the activation tensor is never re-written).  A workgroup owns 32 input channels of the
// projection and REBUILDS the means and all squeeze units itself (<= 48 x 1152 multiply-adds, one pass over the L2-resident fc1
// matrix): cheaper than a dependent launch (7-10 us of the single-stream frame; this used to be two launches per block).
// The kernel is a chain of dependent global round trips, so every phase is laid out for as few of them as possible:
//   1 means    the replicas x C 64-bit sums are read as 16-byte pairs by (pair, replica-slice) threads, all loads of a thread
//              independent, the slices summed in LDS (integers: any grouping gives the same bits)
//   2 squeeze  thread = (unit, K segment): all units at once, each thread a 16-byte-wide dot product over its segment, the
//              segments of a unit summed in a fixed order (one wave per unit walking the units in turn: 12 serial round trips)
//   3 excite   8 partial dot products per channel in a fixed order, one lane per channel finishes: sigmoid gate
//   4 scale    that 32-wide column slice of every projection-weight row, (hi, lo) split
template <bool BATCH>
__global__ __launch_bounds__(256) void se_gate_scale_kernel(const SeParams sein, const ScaleWParams swin) {
  SeParams se = sein;
  ScaleWParams p = swin;
  if constexpr (BATCH) {  // grid.y = camera frame: its own sums, its own gate, its own copy of the scaled projection weights
    se.sums += (size_t)blockIdx.y * se.replicas * se.C;
    p.out_hi += (size_t)blockIdx.y * p.rows * p.C;
    if (p.out_lo) p.out_lo += (size_t)blockIdx.y * p.rows * p.C;
  }
  extern __shared__ __attribute__((aligned(16))) unsigned char se_smem[];
  const int acc_rows = (se.C >> 1) >= 256 ? 1 : 256 / (se.C >> 1);
  unsigned long long* const acc = reinterpret_cast<unsigned long long*>(se_smem);   // [acc_rows][C] partial sums of the replica slices
  float* const mean = reinterpret_cast<float*>(se_smem + (size_t)acc_rows * se.C * 8);  // [C]
  __shared__ float red[256];
  __shared__ float s1[64];
  __shared__ float gate[32];
  __shared__ float part[8][32];
  const int tid = threadIdx.x, C = se.C;
  se_means_squeeze(se, acc, mean, red, s1);
  // ---- 3: excite FC of this workgroup's 32 channels
  const int c0 = blockIdx.x * 32;
  {
    const int cl = tid & 31, jp = tid >> 5, c = c0 + cl;
    const float* wr = p.w2 + (size_t)c * p.sq;
    float s = 0.f;
    for (int j = jp; j < p.sq; j += 8) s = fmaf(wr[j], s1[j], s);
    part[jp][cl] = s;
    __syncthreads();
    if (tid < 32) {
      float t = p.b2[c];
#pragma unroll
      for (int q = 0; q < 8; ++q) t += part[q][cl];
      gate[cl] = c < p.Creal ? sigmoid_f(t) : 0.0f;
    }
  }
  __syncthreads();
  // ---- 4: scale the column slice
  const int oct = tid & 3;
  float g[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) g[i] = gate[oct * 8 + i];
#pragma unroll 2
  for (int row = tid >> 2; row < p.rows; row += 64) {
    const size_t off = (size_t)row * p.C + c0 + oct * 8;
    const f32x4_t a = *reinterpret_cast<const f32x4_t*>(p.w + off), b = *reinterpret_cast<const f32x4_t*>(p.w + off + 4);
    h8_t h, l;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float v = (i < 4 ? a[i] : b[i - 4]) * g[i];
      h[i] = (half_t)v;
      l[i] = (half_t)(v - (float)h[i]);
    }
    *reinterpret_cast<h8_t*>(p.out_hi + off) = h;
    if (p.out_lo) *reinterpret_cast<h8_t*>(p.out_lo + off) = l;
  }
}

// ------------------------------------------------------------------------------------------ context MLP
// out[n] = act(b[n] + sum_k W[n][k] x[k]) (scene_context.py:28-38).  x (optionally the average pool rebuilt from
// slab partials, scene_context.py:27) is staged in LDS once per workgroup; each wave produces two outputs.
__global__ __launch_bounds__(256) void fc_kernel(const FcParams p) {
  extern __shared__ __attribute__((aligned(16))) float xs[];
  for (int k = threadIdx.x; k < p.K; k += 256) {
    float xv;
    if (p.partial) {
      xv = 0.f;
#pragma unroll 8
      for (int q = 0; q < p.nslab; ++q) xv += p.partial[(size_t)q * p.Kstride + k];  // independent loads: one round trip, fixed order
      xv *= p.inv_hw;
    } else {
      xv = p.x[k];
    }
    xs[k] = xv;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const f32x4_t* x4 = reinterpret_cast<const f32x4_t*>(xs);
  const int K4 = p.K >> 2;
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int n = blockIdx.x * 8 + wave * 2 + r;
    if (n >= p.N) continue;
    float s = 0.f;
    if (p.w8) {   // fp8 storage: four codes per lane and step (a quarter of the fp32 row's bytes), the row scale applied once at the end
      const unsigned* wr8 = reinterpret_cast<const unsigned*>(p.w8 + (size_t)n * p.K);
#pragma unroll 6
      for (int k = lane; k < K4; k += 64) {
        const unsigned c4 = wr8[k];
        const f32x4_t m = x4[k];
        s += (e4m3_to_float(c4 & 0xffu) * m[0] + e4m3_to_float((c4 >> 8) & 0xffu) * m[1]) + (e4m3_to_float((c4 >> 16) & 0xffu) * m[2] + e4m3_to_float(c4 >> 24) * m[3]);
      }
    } else {
      const f32x4_t* wr = reinterpret_cast<const f32x4_t*>(p.w + (size_t)n * p.K);
#pragma unroll 6
      for (int k = lane; k < K4; k += 64) {
        const f32x4_t a = wr[k], m = x4[k];
        s += (a[0] * m[0] + a[1] * m[1]) + (a[2] * m[2] + a[3] * m[3]);
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) p.out[n] = apply_act((p.w8 ? s * p.wscale8[n] : s) + p.b[n], p.act_rows ? (int)((p.act_rows >> (8 * (n & 3))) & 0xffu) : p.act);
  }
}

// The same matvec for SHORT rows (K <= 64: AutoDrive's CTX expansion on the large maps is a [H*W = 32768][C = 32] matrix): fc_kernel gives a wave two
// rows and a lane one 4-element piece of a row, so with K = 32 eight lanes of 64 work (26 us, measured).  Here a thread owns a row -- K bytes
// (fp8 codes) or 4 K bytes contiguous per thread, neighbouring threads neighbouring rows -- and sums it front to back against x in LDS.
__global__ __launch_bounds__(256) void fc_rows_kernel(const FcParams p) {
  __shared__ __attribute__((aligned(16))) float xs[64];
  if ((int)threadIdx.x < p.K) {
    const int k = threadIdx.x;
    float xv;
    if (p.partial) {
      xv = 0.f;
#pragma unroll 8
      for (int q = 0; q < p.nslab; ++q) xv += p.partial[(size_t)q * p.Kstride + k];
      xv *= p.inv_hw;
    } else {
      xv = p.x[k];
    }
    xs[k] = xv;
  }
  __syncthreads();
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= p.N) return;
  const f32x4_t* x4 = reinterpret_cast<const f32x4_t*>(xs);
  const int K4 = p.K >> 2;
  float s = 0.f;
  if (p.w8) {
    const unsigned* wr8 = reinterpret_cast<const unsigned*>(p.w8 + (size_t)n * p.K);
    for (int k = 0; k < K4; ++k) {
      const unsigned c4 = wr8[k];
      const f32x4_t m = x4[k];
      s += (e4m3_to_float(c4 & 0xffu) * m[0] + e4m3_to_float((c4 >> 8) & 0xffu) * m[1]) + (e4m3_to_float((c4 >> 16) & 0xffu) * m[2] + e4m3_to_float(c4 >> 24) * m[3]);
    }
  } else {
    const f32x4_t* wr = reinterpret_cast<const f32x4_t*>(p.w + (size_t)n * p.K);
    for (int k = 0; k < K4; ++k) {
      const f32x4_t a = wr[k], m = x4[k];
      s += (a[0] * m[0] + a[1] * m[1]) + (a[2] * m[2] + a[3] * m[3]);
    }
  }
  p.out[n] = apply_act((p.w8 ? s * p.wscale8[n] : s) + p.b[n], p.act_rows ? (int)((p.act_rows >> (8 * (n & 3))) & 0xffu) : p.act);
}

// Zeroes the squeeze-excite accumulators once per frame (a kernel, not hipMemsetAsync: the memset was not replayed
// by the captured graph on this ROCm, so the sums kept growing from frame to frame).
__global__ __launch_bounds__(256) void zero_u64_kernel(unsigned long long* p, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = 0ull;
}

// ---------------------------------------------------------------------------------------------- launchers
hipError_t launch_zero_u64(unsigned long long* p, size_t n, hipStream_t st) {
  VP_LAUNCH(zero_u64_kernel, dim3(nblk((long long)n)), dim3(256), 0, st, p, n);
}
hipError_t launch_stem(const StemParams& p, hipStream_t st) {
  VP_LAUNCH(stem_kernel, dim3(nblk((long long)(p.H / 2) * (p.W / 2) * 4)), dim3(256), 0, st, p);
}
hipError_t launch_dwconv(const DwParams& p, hipStream_t st) {
  const int CG = p.in.C >> 3, G = dw_octets_per_group(CG), PXB = 256 / G;
  const dim3 grid((p.out.H * p.out.W + PXB - 1) / PXB, CG / G);
  const size_t lds = (size_t)p.k * p.k * G * 8 * sizeof(float) + (size_t)G * 8 * sizeof(unsigned long long);
  const bool split = p.in.lo != nullptr;
  if (split != (p.out.lo != nullptr)) return hipErrorInvalidValue;
  if (p.frames > 1) {
    const dim3 gridb(grid.x, grid.y, p.frames);
    if (p.k == 3 && !split) VP_LAUNCH((dwconv_pool_kernel<3, false, true>), gridb, dim3(256), lds, st, p, G);
    if (p.k == 5 && !split) VP_LAUNCH((dwconv_pool_kernel<5, false, true>), gridb, dim3(256), lds, st, p, G);
    if (p.k == 3) VP_LAUNCH((dwconv_pool_kernel<3, true, true>), gridb, dim3(256), lds, st, p, G);
    if (p.k == 5) VP_LAUNCH((dwconv_pool_kernel<5, true, true>), gridb, dim3(256), lds, st, p, G);
    return hipErrorInvalidValue;
  }
  if (p.k == 3 && !split) VP_LAUNCH((dwconv_pool_kernel<3, false>), grid, dim3(256), lds, st, p, G);
  if (p.k == 5 && !split) VP_LAUNCH((dwconv_pool_kernel<5, false>), grid, dim3(256), lds, st, p, G);
  if (p.k == 3) VP_LAUNCH((dwconv_pool_kernel<3, true>), grid, dim3(256), lds, st, p, G);
  if (p.k == 5) VP_LAUNCH((dwconv_pool_kernel<5, true>), grid, dim3(256), lds, st, p, G);
  return hipErrorInvalidValue;
}
hipError_t launch_pool_partial(const PoolParams& p, hipStream_t st) {
  const int CG = p.in.C >> 3, CGL = slab_cgl(p.in.C);
  VP_LAUNCH(pool_partial_kernel, dim3(p.nslab, (CG + CGL - 1) / CGL), dim3(256), 0, st, p);
}
hipError_t launch_se_gate_scale(const SeParams& se, const ScaleWParams& sw, hipStream_t st) {
  if (!se.sums || se.C != sw.C || se.sq != sw.sq || se.frames != sw.frames || se.sq < 1 || se.sq > 64 || (se.C & 31)) return hipErrorInvalidValue;
  const int acc_rows = (se.C >> 1) >= 256 ? 1 : 256 / (se.C >> 1);
  const size_t lds = (size_t)se.C * (8 * acc_rows + 4);
  if (se.frames > 1) VP_LAUNCH(se_gate_scale_kernel<true>, dim3(sw.C / 32, se.frames), dim3(256), lds, st, se, sw);
  VP_LAUNCH(se_gate_scale_kernel<false>, dim3(sw.C / 32), dim3(256), lds, st, se, sw);
}
bool fc_rows_ok(const FcParams& p) { return p.K >= 4 && p.K <= 64 && (p.K & 3) == 0 && p.N >= 2048; }
hipError_t launch_fc(const FcParams& p, hipStream_t st) {
  if (p.act_rows && p.N > 4) return hipErrorInvalidValue;
  if (p.rows_kernel) {   // the plan's choice for short rows and many of them (fc_rows_ok)
    if (!fc_rows_ok(p)) return hipErrorInvalidValue;
    VP_LAUNCH(fc_rows_kernel, dim3((p.N + 255) / 256), dim3(256), 0, st, p);
  }
  VP_LAUNCH(fc_kernel, dim3((p.N + 7) / 8), dim3(256), p.K * sizeof(float), st, p);
}

}  // namespace vp
