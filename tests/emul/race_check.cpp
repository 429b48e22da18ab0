// TEST INFRASTRUCTURE: data-race check of the kernels.  Built by build.py (race target) from the same host-compiled sources
// as libvp_emul.so, plus -fsanitize=thread -DVP_EMU_TSAN: every work-item is a ThreadSanitizer fiber and the only
// happens-before edges are the GPU's own (__syncthreads, wave operations, kernel boundaries; shim/hip/hip_runtime.h).
// Runs each kernel family once on small random inputs -- results are not checked here (the emulated parity tests do that);
// ThreadSanitizer reports any pair of work-items touching the same LDS / global bytes without synchronisation.
//   race_check            all production kernels (~2 min); exit code 0 and no report expected
//   race_check --quick    the subset the CPU suite runs (every non-MFMA kernel, two halo tiles, a split-K GEMM, both ConvTranspose kernels)
//   race_check --canary   a deliberately racy kernel; a report is expected (proves the detector sees LDS races)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "act_io.hpp"
#include "vp_hip_path.h"

using namespace vp;

extern "C" {
int emu_stem(const float*, int, int, const float*, const float*, void*, void*);
int emu_dwconv(void*, void*, int, int, int, void*, void*, int, int, const float*, const float*, int, int, unsigned long long*, int);
int emu_se_gate_scale(const unsigned long long*, int, int, int, int, float, const float*, const float*, const float*, void*, void*, int, const float*, const float*, int);
int emu_mbconv_front(void*, void*, int, int, int, const void*, const void*, const float*, const float*, const float*, void*, void*, int, int, int, unsigned long long*, int, const float*, int,
                     unsigned long long*);
int emu_mbconv_back(void*, void*, int, int, int, int, const unsigned long long*, int, int, const float*, const float*, const float*, const float*, int, const float*, const float*,
                    void*, void*, void*, void*, int, const unsigned long long*);
int emu_fc(const float*, const float*, const float*, float*, int, int, int);
int emu_detect(const float*, int, int, float, float, float, int, int, int, int, float*, int*, void*, int, int*);
int emu_pool_partial(void*, void*, int, int, int, float*, int);
int emu_attention(void*, void*, int, int, int, int, int, float, void*, void*, void*, void*, int);
int emu_sppf_pool(void*, void*, int, int, int, void*, void*, int, int);
int emu_depth_viz(const float*, size_t, const uint8_t*, uint8_t*);
int emu_maxpool5(void*, int, int, int, int, void*, int, int, int);
}

static std::mt19937 rng(1);
static std::vector<float> rnd(size_t n, float s = 1.0f) {
  std::normal_distribution<float> d(0.f, s);
  std::vector<float> v(n);
  for (auto& x : v) x = d(rng);
  return v;
}
static std::vector<half_t> rnd16(size_t n) {
  std::vector<float> f = rnd(n);
  std::vector<half_t> v(n);
  for (size_t i = 0; i < n; ++i) v[i] = (half_t)f[i];
  return v;
}

static int conv(int precision, int mode, int cin, int cout, int h, int w, int ks, int act, int tile, int bk, int nsplit) {
  const int k = mode == 1 ? 2 : ks;
  std::vector<float> x = rnd((size_t)cin * h * w), wt = rnd((size_t)cin * cout * k * k, 0.1f), b = rnd(cout, 0.1f);
  std::vector<float> out((size_t)cout * (mode == 1 ? 4 : 1) * h * w);
  char err[256] = {0};
  const int rc = vp_op_conv2d(0, precision, mode, x.data(), cin, h, w, wt.data(), b.data(), cout, ks, act, 0, nullptr, tile, bk, nsplit, out.data(), err, sizeof err);
  if (rc) std::fprintf(stderr, "vp_op_conv2d(tile %d) failed: %s\n", tile, err);
  return rc;
}

static int conv_logits(int precision, int cin, int cout, int h, int w) {  // vp_op_conv2d mode 3: kernels_head.hip
  std::vector<float> x = rnd((size_t)cin * h * w), wt = rnd((size_t)cin * cout * 9, 0.1f), b = rnd(cout, 0.1f), out((size_t)cout * h * w);
  char err[256] = {0};
  const int rc = vp_op_conv2d(0, precision, 3, x.data(), cin, h, w, wt.data(), b.data(), cout, 3, 0, 0, nullptr, -1, -1, -1, out.data(), err, sizeof err);
  if (rc) std::fprintf(stderr, "vp_op_conv2d(mode 3) failed: %s\n", err);
  return rc;
}

// composed up-sampling stage (kernels_upconv.hip) through vp_op_upconv: halo images handed over at a chunk's last step (double-buffered on the 8-wave
// shape, rewritten between two barriers on the 4-wave shape), three weight buffers by LDS-DMA, K slices + finish kernel; precision 0 = the fp16 engines' form
static int upconv(int precision, int cin, int cm, int cout, int cs, int h, int w, int shape, int nsplit) {
  std::vector<float> x = rnd((size_t)cin * h * w), wt = rnd((size_t)cin * cm * 4, 0.1f), bt = rnd(cm, 0.1f), w3 = rnd((size_t)cout * cm * 9, 0.1f), b3 = rnd(cout, 0.1f);
  std::vector<float> sk = rnd((size_t)std::max(cs, 1) * 4 * h * w), ws = rnd((size_t)cm * std::max(cs, 1), 0.1f), bs = rnd(cm, 0.1f), out((size_t)cout * 4 * h * w);
  char err[256] = {0};
  const int rc = vp_op_upconv(0, x.data(), cin, h, w, cs ? sk.data() : nullptr, cs, wt.data(), bt.data(), cs ? ws.data() : nullptr, cs ? bs.data() : nullptr, w3.data(),
                              b3.data(), cm, cout, 1, shape, nsplit, precision, out.data(), err, sizeof err);
  if (rc) std::fprintf(stderr, "vp_op_upconv failed: %s\n", err);
  return rc;
}

__global__ void canary_kernel(float* out) {
  __shared__ float lds[64];
  lds[threadIdx.x] = (float)threadIdx.x;
  out[threadIdx.x] = lds[threadIdx.x ^ 1];  // no __syncthreads between a neighbour's write and this read
}

__global__ void trivial_kernel(float* out) {
  __shared__ float lds[256];
  lds[threadIdx.x] = (float)threadIdx.x;
  __syncthreads();
  out[blockIdx.x * 256 + threadIdx.x] = lds[threadIdx.x ^ 1];
}

int main(int argc, char** argv) {
  if (argc > 3 && !std::strcmp(argv[1], "--trivial")) {
    const int blocks = std::atoi(argv[2]), reps = std::atoi(argv[3]);
    std::vector<float> out((size_t)blocks * 256);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(trivial_kernel, dim3(blocks), dim3(256), 0, nullptr, out.data());
    std::printf("trivial ok\n");
    return 0;
  }
  if (argc > 1 && !std::strcmp(argv[1], "--canary")) {
    std::vector<float> out(64);
    hipLaunchKernelGGL(canary_kernel, dim3(1), dim3(64), 0, nullptr, out.data());
    return 0;
  }
  int bad = 0;
  // flags (any order): --quick = the CPU suite's subset; --no-conv / --conv-only = one half of the launches (the suite runs the two halves as two
  // processes side by side: one OS thread per work-item makes the run system-time-bound); no flag = everything
  bool skip_conv = false, quick = false, conv_only = false;
  for (int a = 1; a < argc; ++a) {
    skip_conv |= !std::strcmp(argv[a], "--no-conv");
    quick |= !std::strcmp(argv[a], "--quick");
    conv_only |= !std::strcmp(argv[a], "--conv-only");
  }
  for (int precision = 0; precision < (quick ? 1 : 2) && !skip_conv; ++precision) {
    for (int tile : {100, 101, 102, 103, 104}) {  // halo 3x3
      if (quick && tile != 101 && tile != 104) continue;
      bad |= conv(precision, 0, 64, 72, 11, 21, 3, 1, tile, -1, tile == 101 || tile == 103 ? 2 : 1);
    }
    for (int tile : {0, 1, 2, 3}) {  // K1 GEMM
      if (quick && tile != 1) continue;
      bad |= conv(precision, 0, 96, 40, 7, 13, 1, 2, tile, tile == 2 ? 64 : 32, tile == 1 ? 2 : 1);
    }
    if (precision == 0) {  // the fp16 engines' form of the pipelined kernels: 64-channel chunks, the chunk's halves as the two LDS planes
      bad |= conv(0, 0, 128, 128, 9, 17, 3, 1, 106, -1, 1);   // two chunks: the double-buffered halo hand-over
      if (!quick) bad |= conv(0, 0, 192, 128, 11, 19, 3, 1, 107, -1, 1);
      bad |= conv(0, 0, 64, 32, 10, 20, 3, 1, 111, -1, 2);   // map kernel, fp16 form (32-channel steps), context geometry, two K slices
    }
    if (precision == 1 || quick) {  // 8-wave fp16x3 kernel: 3 weight buffers, cross-tap fragment prefetch, 2 chunks
      bad |= conv(1, 0, 64, 128, 17, 19, 3, 1, 106, -1, 1);
      bad |= conv(1, 0, 96, 128, 11, 19, 3, 1, 107, -1, 1);
      bad |= conv(1, 0, 96, 64, 9, 17, 3, 1, 108, -1, 1);  // 64-channel shape (three workgroups per CU on the device)
      bad |= conv(1, 0, 64, 32, 20, 40, 3, 1, 111, -1, 2);
      bad |= conv(1, 0, 64, 32, 10, 20, 3, 1, 111, -1, 2);   // its 10x20-region geometry (context block): four waves, ragged seventh pixel tile   // map kernel: double-buffered 16-channel steps by LDS-DMA, one barrier per step
      bad |= conv(1, 0, 160, 128, 10, 20, 3, 1, 107, -1, 2);  // split-K slices  // 4-wave shape: single halo buffer rewritten between two barriers, 3 chunks
    }
    bad |= conv(precision, 0, 32, 40, 5, 8, 3, 1, 1, 32, 1);                                                                                  // generic 3x3
    bad |= conv(precision, 1, 64, 48, 5, 6, 2, 0, -1, -1, -1);                                                                                // ConvTranspose GEMM
    bad |= conv(precision, 1, 128, 64, 4, 8, 2, 1, -1, -1, -1);                                                                               // short-K ConvTranspose GEMM
    if (precision == 0 && !quick) bad |= conv(0, 1, 128, 64, 32, 64, 2, 0, -1, -1, -1);  // >= 2048 px: kernels_convt_rs.hip picked by the engine
    // register-stationary ConvTranspose (kernels_convt_rs.hip): three DMA tile buffers, one barrier per tile, wave-private patches; 7-8 tiles per workgroup
    vp_set_option("VP_CONVT_RS_GROUPS", "9");
    if (!quick) bad |= conv(precision, 1, 128, 128, 32, 64, 2, 0, 5, -1, 1);  // (2048 pixels minimum: ~2 min under the sanitizer, not in the suite's subset)
    vp_set_option("VP_CONVT_RS_GROUPS", nullptr);
    // LDS-DMA GEMM (kernels_gemm_dma.hip): three-stage ring, one barrier per K step, patches over the ring; 256 pixels = two tiles, 8 K steps
    if (precision == 1 || quick) bad |= conv(1, 1, 256, 256, 8, quick ? 16 : 32, 2, 0, 6, -1, quick ? 2 : 1);
    // a head's logits convolution (kernels_head.hip) is reached through mode 3 only: see conv_logits below
  }
  if (!skip_conv) {  // composed up-sampling stages, both precisions: x chunks + the four skip classes (4 / 2 / 2 / 1 taps), both shapes, K slices
    bad |= upconv(1, 64, 16, 128, 24, 5, 17, 7, 2);           // 4-wave shape: single halo image rewritten between two barriers; K slices + finish kernel
    bad |= upconv(0, 128, 16, 128, 24, 5, 17, 6, 1);          // fp16 form on the 8-wave shape: double-buffered halo, 64-channel chunks, half-dead skip chunk
    if (!quick) {
      bad |= upconv(1, 64, 16, 128, 24, 9, 17, 6, 1);
      bad |= upconv(0, 128, 16, 128, 72, 5, 17, 7, 2);        // 96 skip channels: one full + one half-dead chunk per class, K slices
      bad |= upconv(1, 96, 16, 128, 0, 17, 19, 6, 1);
    }
  }
  if (!skip_conv) {  // heads' logits convolution: DMA halo + zero page, slab reduction through LDS (128 channels)
    bad |= conv_logits(1, 128, 3, quick ? 5 : 9, quick ? 17 : 33);
    if (!quick) bad |= conv_logits(0, 64, 1, 17, 20);
  }
  if (conv_only) {
    std::printf("race_check: all launches done%s\n", bad ? " (some FAILED to launch)" : "");
    return bad ? 2 : 0;
  }

  {  // encoder pieces
    const int H = 18, W = 28;
    std::vector<float> x = rnd(3 * H * W), w = rnd(27 * 32, 0.3f), b = rnd(32, 0.1f);
    std::vector<half_t> hi((size_t)(H / 2) * (W / 2) * 32), lo(hi.size());
    bad |= emu_stem(x.data(), H, W, w.data(), b.data(), hi.data(), lo.data());
    for (int k : {3, 5})
      for (int split = 0; split < 2; ++split) {
        const int C = 160, h = 13, ww = 21, stride = k == 5 ? 2 : 1, oh = (h + stride - 1) / stride, ow = (ww + stride - 1) / stride;
        std::vector<half_t> ih = rnd16((size_t)h * ww * C), il = rnd16(ih.size()), oh_((size_t)oh * ow * C), ol_(oh_.size());
        std::vector<float> wk = rnd((size_t)k * k * C, 0.3f), bb = rnd(C, 0.1f);
        std::vector<unsigned long long> sums(8 * C, 0ull);
        bad |= emu_dwconv(ih.data(), split ? il.data() : nullptr, h, ww, C, oh_.data(), split ? ol_.data() : nullptr, oh, ow, wk.data(), bb.data(), k, stride,
                          sums.data(), 8);
        std::vector<float> w1 = rnd(6 * C, 0.2f), b1 = rnd(6, 0.1f), s1(6), w2 = rnd(C * 6, 0.5f), b2 = rnd(C, 0.1f), pw = rnd(64 * C);
        std::vector<half_t> ph((size_t)64 * C), pl(ph.size());
        bad |= emu_se_gate_scale(sums.data(), 8, C, C, 6, 1.0f / (oh * ow), w1.data(), b1.data(), pw.data(), ph.data(), split ? pl.data() : nullptr, 64, w2.data(), b2.data(), 1);
      }
    // fused MBConv front (kernels_mbconv.hip): staging buffers rewritten per K chunk, the fp32 tile laid over them, LDS pool atomics
    for (int cfg = 0; cfg < (quick ? 1 : 2); ++cfg) {
      const int k = cfg ? 5 : 3, stride = cfg ? 2 : 1, cin = 64, cexp = 96, h = 10, ww = 18, oh = h / stride, ow = ww / stride;
      std::vector<half_t> ih = rnd16((size_t)h * ww * cin), il = rnd16(ih.size()), wh = rnd16((size_t)cexp * cin), wl = rnd16(wh.size()), oh_((size_t)oh * ow * cexp), ol_(oh_.size());
      std::vector<float> be = rnd(cexp, 0.1f), wk = rnd((size_t)k * k * cexp, 0.3f), bb = rnd(cexp, 0.1f);
      std::vector<unsigned long long> sums(4 * cexp, 0ull), zs(4 * 64, 0ull);
      std::vector<float> w1 = rnd((size_t)6 * cexp, 0.2f);
      bad |= emu_mbconv_front(ih.data(), il.data(), h, ww, cin, wh.data(), wl.data(), be.data(), wk.data(), bb.data(), oh_.data(), ol_.data(), cexp, k, stride, sums.data(), 4, w1.data(), 6,
                              zs.data());
    }
    // fused MBConv back: the gate phases' LDS hand-offs, the waves' partial tiles meeting in LDS (a wave per K quarter / per pixel tile)
    for (int cfg = 0; cfg < (quick ? 1 : 2); ++cfg) {
      const int C = 96, cout = 64, sq = 6, sqp = 8, h = cfg ? 100 : 5, ww = cfg ? 128 : 9, m = h * ww;   // 12800 pixels: the wide instantiation
      std::vector<half_t> ih = rnd16((size_t)m * C), il = rnd16(ih.size()), rh = rnd16((size_t)m * cout), rl = rnd16(rh.size()), oh_((size_t)m * cout), ol_(oh_.size());
      std::vector<unsigned long long> sums(8 * C, 1ull << 20);
      std::vector<float> w1 = rnd(sq * C, 0.2f), b1 = rnd(sq, 0.1f), w2q = rnd((size_t)sqp * C, 0.5f), b2 = rnd(C, 0.1f), pw = rnd((size_t)cout * C), pb = rnd(cout);
      bad |= emu_mbconv_back(ih.data(), il.data(), h, ww, C, C, sums.data(), 8, sq, w1.data(), b1.data(), w2q.data(), b2.data(), sqp, pw.data(), pb.data(), rh.data(), rl.data(),
                             oh_.data(), ol_.data(), cout, cfg ? sums.data() : nullptr);
    }
    {  // detector decode + NMS (kernels_detect.hip): LDS candidate list, bitonic sort, suppression flags written between barriers
      const int nb = 600, na = 7;
      std::vector<float> raw = rnd((size_t)na * nb, 0.5f);
      for (int i = 0; i < nb; ++i) {
        raw[i] = 40.0f + (float)(i % 23) * 9.0f;
        raw[nb + i] = 50.0f + (float)(i % 17) * 11.0f;
        raw[2 * nb + i] = 30.0f + (float)(i % 5);
        raw[3 * nb + i] = 25.0f + (float)(i % 7);
      }
      std::vector<float> boxes((size_t)nb * 4);
      std::vector<int> cls(nb), count(2);
      std::vector<float> out((size_t)nb * 6);
      bad |= emu_detect(raw.data(), na, nb, 0.1f, 0.4f, 0.5f, 0, 140, 1280, 720, boxes.data(), cls.data(), out.data(), nb, count.data());
    }
    std::vector<float> fx = rnd(200), fw = rnd(37 * 200, 0.1f), fb = rnd(37), fo(37);
    bad |= emu_fc(fx.data(), fw.data(), fb.data(), fo.data(), 37, 200, 1);
    std::vector<half_t> ph = rnd16(10 * 20 * 96), pl = rnd16(ph.size());
    std::vector<float> partial(4 * 96);
    bad |= emu_pool_partial(ph.data(), pl.data(), 10, 20, 96, partial.data(), 4);
  }
  {  // AutoDrive attention, SPPF pool; depth visualisation
    const int heads = 2, dk = 8, dv = 16, Hq = 6, Wq = 7, T = Hq * Wq, per = 2 * dk + dv;
    std::vector<half_t> qh = rnd16((size_t)T * heads * per), ql = rnd16(qh.size()), oh((size_t)T * heads * dv), ol(oh.size()), vh(oh.size()), vl(oh.size());
    bad |= emu_attention(qh.data(), ql.data(), Hq, Wq, heads, dk, dv, 0.35f, oh.data(), ol.data(), vh.data(), vl.data(), 0);
    {  // four query tokens per workgroup (dk = 32, dv = 64): LDS scores, two block reductions, value partials of 32 token slices meeting in LDS
      const int dk2 = 32, dv2 = 64, per2 = 2 * dk2 + dv2;
      std::vector<half_t> q2 = rnd16((size_t)T * heads * per2), q2l = rnd16(q2.size()), o2((size_t)T * heads * dv2), o2l(o2.size()), v2(o2.size()), v2l(o2.size());
      bad |= emu_attention(q2.data(), q2l.data(), Hq, Wq, heads, dk2, dv2, 0.18f, o2.data(), o2l.data(), v2.data(), v2l.data(), 4);
    }
    {  // SPPF pyramid: the octet's map and its three row-maximum images in LDS, two barriers
      std::vector<half_t> s2 = rnd16(11 * 14 * 32), s2l = rnd16(s2.size()), d2(11 * 14 * 64), d2l(d2.size());
      bad |= emu_sppf_pool(s2.data(), s2l.data(), 11, 14, 32, d2.data(), d2l.data(), 64, 16);
    }
    std::vector<half_t> src = rnd16(9 * 12 * 32), dst(9 * 12 * 64);
    bad |= emu_maxpool5(src.data(), 9, 12, 32, 8, dst.data(), 64, 32, 16);
    std::vector<float> depth = rnd(37 * 53);
    std::vector<uint8_t> lut(768, 7), img(37 * 53 * 3);
    bad |= emu_depth_viz(depth.data(), depth.size(), lut.data(), img.data());
  }
  std::printf("race_check: all launches done%s\n", bad ? " (some FAILED to launch)" : "");
  return bad ? 2 : 0;
}
