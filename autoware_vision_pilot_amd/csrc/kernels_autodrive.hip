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
hipError_t launch_attention(const AttnParams& p, hipStream_t st) {
  const int T = p.qkv.H * p.qkv.W;
  if (p.dk > 256 || p.dv > 256) return hipErrorInvalidValue;
  VP_LAUNCH(attention_kernel, dim3(T, p.heads), dim3(256), (T + 256 + p.dk) * sizeof(float), st, p);
}
hipError_t launch_dwconv_plain(const DwPlainParams& p, hipStream_t st) {
  VP_LAUNCH(dwconv_plain_kernel, dim3(nblk((long long)p.in.H * p.in.W * (p.in.C >> 3))), dim3(256), 0, st, p);
}

}  // namespace vp
