// Blocks that only AutoDrive needs: channel-slice copies for cat / split, the SPPF
// max-pool, the 2-head 512-token attention of C2PSA and its depthwise positional conv.  The network is ~8 GMAC
// (20x smaller than SceneSeg) on maps of at most 16x32 pixels here, so these are plain latency-sized kernels: one
// 16-byte channel octet per thread, fp32 math, values read as hi (+ lo) like every other HBM-bound kernel.
#include "act_io.hpp"

namespace vp {

__global__ __launch_bounds__(256) void chan_copy_kernel(ActView src, int src_off, ActView dst, int dst_off, int nch) {
  const int CG = nch >> 3;
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= (long long)src.H * src.W * CG) return;
  const int cg = (int)(t % CG);
  const size_t pix = (size_t)(t / CG);
  const size_t so = pix * src.C + src_off + cg * 8, dso = pix * dst.C + dst_off + cg * 8;
  *reinterpret_cast<h8_t*>(dst.hi + dso) = *reinterpret_cast<const h8_t*>(src.hi + so);
  if (dst.lo) *reinterpret_cast<h8_t*>(dst.lo + dso) = *reinterpret_cast<const h8_t*>(src.lo + so);
}

// MaxPool2d(k=5, s=1, p=2): padding never wins a max (PyTorch pads with -inf), so out-of-image taps are skipped.
// max commutes with the (hi, lo) split only through the VALUE, so the comparison is done on hi + lo.
__global__ __launch_bounds__(256) void maxpool5_kernel(ActView src, int src_off, ActView dst, int dst_off, int nch) {
  const int CG = nch >> 3;
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= (long long)src.H * src.W * CG) return;
  const int cg = (int)(t % CG), pix = (int)(t / CG);
  const int y = pix / src.W, x = pix - y * src.W;
  float best[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) best[i] = -3.0e38f;
  for (int dy = -2; dy <= 2; ++dy) {
    const int iy = y + dy;
    if ((unsigned)iy >= (unsigned)src.H) continue;
    for (int dx = -2; dx <= 2; ++dx) {
      const int ix = x + dx;
      if ((unsigned)ix >= (unsigned)src.W) continue;
      float v[8];
      load8(src, ((size_t)iy * src.W + ix) * src.C + src_off + cg * 8, v);
#pragma unroll
      for (int i = 0; i < 8; ++i) best[i] = fmaxf(best[i], v[i]);
    }
  }
  store8(dst, (size_t)pix * dst.C + dst_off + cg * 8, best);
}

// SPPF's pooling pyramid in ONE launch (round 4; it was a channel-slice copy + three chained max-pool launches, 45 us of latency on a 16x32 map):
// y1 = mp5(x), y2 = mp5(y1), y3 = mp5(y2) with -inf padding are the maxima of x over the clipped 5x5, 9x9 and 13x13 windows (a max of maxima
// is the max of the union) and a window maximum is separable: one workgroup holds the WHOLE map of one channel octet (<= 512 pixels, a pixel per
// thread, fp32 values), takes the three row maxima (radius 2 / 4 / 6 along x) into LDS and the column maxima of those.  One global load and one
// LDS round trip per thread (the first form of this kernel walked the 13x13 window per thread through global memory: 64 us -- measured, replaced).
// dst slice 0 = x, slices 1..3 = y1..y3 (common_layers.py:236-243).
__global__ __launch_bounds__(512) void sppf_pool_kernel(ActView src, ActView dst, int nch) {
  __shared__ float a[512 * 8];        // the octet's map
  __shared__ float hrow[3][512 * 8];  // row maxima for radius 2, 4, 6
  const int HW = src.H * src.W, pix = threadIdx.x, cg = blockIdx.x;
  const bool on = pix < HW;
  const int y = on ? pix / src.W : 0, x = on ? pix - y * src.W : 0;
  float ctr[8];
  if (on) {
    load8(src, (size_t)pix * src.C + cg * 8, ctr);
#pragma unroll
    for (int i = 0; i < 8; ++i) a[pix * 8 + i] = ctr[i];
  }
  __syncthreads();
  if (on) {
    float m[3][8];
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
      for (int i = 0; i < 8; ++i) m[k][i] = ctr[i];
#pragma unroll
    for (int d = 1; d <= 6; ++d) {
#pragma unroll
      for (int sgn = -1; sgn <= 1; sgn += 2) {
        const int ix = x + sgn * d;
        if ((unsigned)ix >= (unsigned)src.W) continue;
        const float* v = a + (y * src.W + ix) * 8;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          m[2][i] = fmaxf(m[2][i], v[i]);
          if (d <= 4) m[1][i] = fmaxf(m[1][i], v[i]);
          if (d <= 2) m[0][i] = fmaxf(m[0][i], v[i]);
        }
      }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
      for (int i = 0; i < 8; ++i) hrow[k][pix * 8 + i] = m[k][i];
  }
  __syncthreads();
  if (!on) return;
  const size_t o = (size_t)pix * dst.C + cg * 8;
  store8(dst, o, ctr);
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int r = 2 * (k + 1);
    float best[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) best[i] = hrow[k][pix * 8 + i];
    for (int d = 1; d <= r; ++d) {
#pragma unroll
      for (int sgn = -1; sgn <= 1; sgn += 2) {
        const int iy = y + sgn * d;
        if ((unsigned)iy >= (unsigned)src.H) continue;
        const float* v = hrow[k] + (iy * src.W + x) * 8;
#pragma unroll
        for (int i = 0; i < 8; ++i) best[i] = fmaxf(best[i], v[i]);
      }
    }
    store8(dst, o + (size_t)(k + 1) * nch, best);
  }
}

// Attention.forward (common_layers.py:92-101) for one (head, query token) per workgroup:
//   s_j = scale * sum_d q[d][i] k[d][j]; p = softmax_j(s); out[c][i] = sum_j v[c][j] p_j.
// T = H*W tokens (512), dk = 32, dv = 64: each thread scores T/256 keys, block-wide max / sum through LDS, then dv
// threads accumulate the value rows.  fp32 throughout.
__global__ __launch_bounds__(256) void attention_kernel(const AttnParams p) {
  extern __shared__ float sh[];  // [T] probabilities + [256] reduction scratch + [dk] query
  const int T = p.qkv.H * p.qkv.W;
  float* prob = sh;
  float* red = sh + T;
  float* qv = red + 256;
  const int head = blockIdx.y, qi = blockIdx.x, tid = threadIdx.x;
  const int hc = head * (2 * p.dk + p.dv);
  auto val = [&](int tok, int ch) -> float {
    const size_t o = (size_t)tok * p.qkv.C + ch;
    float v = (float)p.qkv.hi[o];
    if (p.qkv.lo) v += (float)p.qkv.lo[o];
    return v;
  };
  if (tid < p.dk) qv[tid] = val(qi, hc + tid);
  __syncthreads();
  float mx = -3.0e38f;
  for (int j = tid; j < T; j += 256) {
    float s = 0.f;
    for (int d = 0; d < p.dk; ++d) s = fmaf(qv[d], val(j, hc + p.dk + d), s);
    s *= p.scale;
    prob[j] = s;
    mx = fmaxf(mx, s);
  }
  red[tid] = mx;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) red[tid] = fmaxf(red[tid], red[tid + o]);
    __syncthreads();
  }
  mx = red[0];
  __syncthreads();
  float sum = 0.f;
  for (int j = tid; j < T; j += 256) {
    const float e = expf(prob[j] - mx);
    prob[j] = e;
    sum += e;
  }
  red[tid] = sum;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) red[tid] += red[tid + o];
    __syncthreads();
  }
  const float inv = 1.0f / red[0];
  if (tid < p.dv) {
    float acc = 0.f;
    for (int j = 0; j < T; ++j) acc = fmaf(val(j, hc + 2 * p.dk + tid), prob[j], acc);
    acc *= inv;
    const size_t o = (size_t)qi * p.out.C + head * p.dv + tid;
    const half_t h = (half_t)acc;
    p.out.hi[o] = h;
    if (p.out.lo) p.out.lo[o] = (half_t)(acc - (float)h);
    const float vv = val(qi, hc + 2 * p.dk + tid);
    const half_t vh = (half_t)vv;
    p.vout.hi[o] = vh;
    if (p.vout.lo) p.vout.lo[o] = (half_t)(vv - (float)vh);
  }
}

// The same attention for the shape C2PSA has here (dk = 32, dv = 64), QB query tokens per workgroup (round 4: the kernel above spent 168 us of
// AutoDrive's 0.66 ms frame on 50 MFLOP -- one workgroup per query, every key row fetched two bytes at a time, one thread walking the 512 value
// rows).  Here a thread fetches a key row ONCE as 16-byte pieces and scores it against the eight queries (broadcast reads from LDS), the
// probabilities of the block stay in LDS, and the value pass runs one (channel, query pair) per thread over rows that a wave reads as 128
// contiguous bytes.  Arithmetic order per output is the kernel above's: s = fma over d = 0..31 then * scale; softmax against the row maximum;
// out = (sum over j of v * e) * (1 / sum e) with the j sum taken as 32 interleaved slices (fma chains) added in slice order.
template <int QB>
__global__ __launch_bounds__(256) void attention_block_kernel(const AttnParams p) {
  constexpr int DK = 32, DV = 64;
  extern __shared__ float sh[];  // [QB][T] scores -> probabilities | [QB][256] reduction scratch | [QB][DK] queries | [32][QB][DV] value partials
  const int T = p.qkv.H * p.qkv.W;
  float* prob = sh;
  float* red = sh + QB * T;
  float* qv = red + QB * 256;
  const int head = blockIdx.y, q0 = blockIdx.x * QB, tid = threadIdx.x;
  const int hc = head * (2 * DK + DV);
  if (tid < QB * DK / 8) {  // one 8-channel piece of one query per thread
    const int q = tid / (DK / 8), c8 = tid % (DK / 8);
    float v[8];
    const int tok = q0 + q < T ? q0 + q : T - 1;
    load8(p.qkv, (size_t)tok * p.qkv.C + hc + c8 * 8, v);
#pragma unroll
    for (int i = 0; i < 8; ++i) qv[q * DK + c8 * 8 + i] = v[i];
  }
  __syncthreads();
  float mx[QB];
#pragma unroll
  for (int q = 0; q < QB; ++q) mx[q] = -3.0e38f;
  for (int j = tid; j < T; j += 256) {
    float kr[DK];
#pragma unroll
    for (int c8 = 0; c8 < DK / 8; ++c8) load8(p.qkv, (size_t)j * p.qkv.C + hc + DK + c8 * 8, kr + c8 * 8);
#pragma unroll
    for (int q = 0; q < QB; ++q) {
      float s = 0.f;
#pragma unroll
      for (int d = 0; d < DK; ++d) s = fmaf(qv[q * DK + d], kr[d], s);
      s *= p.scale;
      prob[q * T + j] = s;
      mx[q] = fmaxf(mx[q], s);
    }
  }
#pragma unroll
  for (int q = 0; q < QB; ++q) red[q * 256 + tid] = mx[q];
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) {
#pragma unroll
      for (int q = 0; q < QB; ++q) red[q * 256 + tid] = fmaxf(red[q * 256 + tid], red[q * 256 + tid + o]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int q = 0; q < QB; ++q) mx[q] = red[q * 256];
  __syncthreads();
  float sum[QB];
#pragma unroll
  for (int q = 0; q < QB; ++q) sum[q] = 0.f;
  for (int j = tid; j < T; j += 256) {
#pragma unroll
    for (int q = 0; q < QB; ++q) {
      const float e = expf(prob[q * T + j] - mx[q]);
      prob[q * T + j] = e;
      sum[q] += e;
    }
  }
#pragma unroll
  for (int q = 0; q < QB; ++q) red[q * 256 + tid] = sum[q];
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) {
#pragma unroll
      for (int q = 0; q < QB; ++q) red[q * 256 + tid] += red[q * 256 + tid + o];
    }
    __syncthreads();
  }
  // value pass: thread = (channel octet c8, token slice ts): 16-byte value pieces of the tokens j = ts (mod 32) against the block's QB probability
  // rows, every load independent (the first form of this pass -- a thread per channel walking the T rows two bytes at a time -- was a chain of
  // dependent round trips: 140 us, measured); the 32 slices meet in LDS and are summed in slice order
  float* part = qv + QB * DK;  // [32 slices][QB][DV]
  {
    const int c8 = tid & 7, ts = tid >> 3;
    float acc[QB][8];
#pragma unroll
    for (int q = 0; q < QB; ++q)
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[q][i] = 0.f;
#pragma unroll 4
    for (int j = ts; j < T; j += 32) {
      float v[8];
      load8(p.qkv, (size_t)j * p.qkv.C + hc + 2 * DK + c8 * 8, v);
#pragma unroll
      for (int q = 0; q < QB; ++q) {
        const float pj = prob[q * T + j];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[q][i] = fmaf(v[i], pj, acc[q][i]);
      }
    }
#pragma unroll
    for (int q = 0; q < QB; ++q)
#pragma unroll
      for (int i = 0; i < 8; ++i) part[(ts * QB + q) * DV + c8 * 8 + i] = acc[q][i];
  }
  __syncthreads();
  for (int o = tid; o < QB * DV; o += 256) {
    const int q = o / DV, c = o - q * DV, qi = q0 + q;
    if (qi >= T) continue;
    float a = 0.f;
#pragma unroll 8
    for (int ts = 0; ts < 32; ++ts) a += part[(ts * QB + q) * DV + c];
    a *= 1.0f / red[q * 256];
    const size_t oo = (size_t)qi * p.out.C + head * DV + c;
    const half_t h = (half_t)a;
    p.out.hi[oo] = h;
    if (p.out.lo) p.out.lo[oo] = (half_t)(a - (float)h);
    const size_t vo = (size_t)qi * p.qkv.C + hc + 2 * DK + c;
    float vv = (float)p.qkv.hi[vo];
    if (p.qkv.lo) vv += (float)p.qkv.lo[vo];
    const half_t vhh = (half_t)vv;
    p.vout.hi[oo] = vhh;
    if (p.vout.lo) p.vout.lo[oo] = (half_t)(vv - (float)vhh);
  }
}

// out = add + depthwise3x3(in) + b  (Attention: ... + self.conv1(v), common_layers.py:103; BN folded, identity activation)
__global__ __launch_bounds__(256) void dwconv_plain_kernel(const DwPlainParams p) {
  const int CG = p.in.C >> 3;
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= (long long)p.in.H * p.in.W * CG) return;
  const int cg = (int)(t % CG), pix = (int)(t / CG);
  const int y = pix / p.in.W, x = pix - y * p.in.W;
  float acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = p.b[cg * 8 + i];
  for (int ky = 0; ky < 3; ++ky)
    for (int kx = 0; kx < 3; ++kx) {
      const int iy = y + ky - 1, ix = x + kx - 1;
      if ((unsigned)iy >= (unsigned)p.in.H || (unsigned)ix >= (unsigned)p.in.W) continue;
      float v[8];
      load8(p.in, ((size_t)iy * p.in.W + ix) * p.in.C + cg * 8, v);
      const float* wk = p.w + (size_t)(ky * 3 + kx) * p.in.C + cg * 8;
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = fmaf(v[i], wk[i], acc[i]);
    }
  float a[8];
  load8(p.add, (size_t)pix * p.add.C + cg * 8, a);
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] += a[i];
  store8(p.out, (size_t)pix * p.out.C + cg * 8, acc);
}

hipError_t launch_chan_copy(const ActView& src, int src_off, const ActView& dst, int dst_off, int nch, hipStream_t st) {
  VP_LAUNCH(chan_copy_kernel, dim3(nblk((long long)src.H * src.W * (nch >> 3))), dim3(256), 0, st, src, src_off, dst, dst_off, nch);
}
hipError_t launch_maxpool5(const ActView& src, int src_off, const ActView& dst, int dst_off, int nch, hipStream_t st) {
  VP_LAUNCH(maxpool5_kernel, dim3(nblk((long long)src.H * src.W * (nch >> 3))), dim3(256), 0, st, src, src_off, dst, dst_off, nch);
}
// LDS of attention_block_kernel<4>: [4][T] probabilities + [4][256] reduction scratch + [4][32] queries + [32][4][64] value partials, within the
// 64 KB a launch gets without an attribute: T <= 1760 tokens (AutoDrive's P5 at 1024 x 512 input: 512)
static size_t attention_block_lds(int T) { return (size_t)(4 * T + 4 * 256 + 4 * 32 + 32 * 4 * 64) * sizeof(float); }
bool attention_block_ok(const AttnParams& p) { return p.dk == 32 && p.dv == 64 && p.qkv.C % 8 == 0 && attention_block_lds(p.qkv.H * p.qkv.W) <= 64 * 1024; }
bool sppf_pool_ok(const ActView& src, const ActView& dst, int nch) {
  return !(nch & 7) && nch <= src.C && 4 * nch <= dst.C && src.H == dst.H && src.W == dst.W && src.H * src.W <= 512 && (src.lo == nullptr) == (dst.lo == nullptr);
}
hipError_t launch_sppf_pool(const ActView& src, const ActView& dst, int nch, hipStream_t st) {
  if (!sppf_pool_ok(src, dst, nch)) return hipErrorInvalidValue;
  VP_LAUNCH(sppf_pool_kernel, dim3(nch >> 3), dim3(512), 0, st, src, dst, nch);
}
hipError_t launch_attention(const AttnParams& p, hipStream_t st) {
  const int T = p.qkv.H * p.qkv.W;
  if (p.dk > 256 || p.dv > 256) return hipErrorInvalidValue;
  if (p.qblock == 4) {   // the engine selects it at plan time (attention_block_ok)
    if (!attention_block_ok(p)) return hipErrorInvalidValue;
    constexpr int QB = 4;
    const size_t lds = attention_block_lds(T);
    VP_LAUNCH(attention_block_kernel<QB>, dim3((T + QB - 1) / QB, p.heads), dim3(256), lds, st, p);
  }
  if (p.qblock != 0) return hipErrorInvalidValue;
  VP_LAUNCH(attention_kernel, dim3(T, p.heads), dim3(256), (T + 256 + p.dk) * sizeof(float), st, p);
}
hipError_t launch_dwconv_plain(const DwPlainParams& p, hipStream_t st) {
  VP_LAUNCH(dwconv_plain_kernel, dim3(nblk((long long)p.in.H * p.in.W * (p.in.C >> 3))), dim3(256), 0, st, p);
}

}  // namespace vp
