// TEST INFRASTRUCTURE (see shim/hip/hip_runtime.h): storage for the kernels' dynamic LDS + plain-C entry points onto the
// kernel launchers for the CPU tests (tests/test_kernels_emulated.py).  Linked with every csrc/ source compiled for the host
// (build.py), so the emulated library also exports the whole C ABI of include/vp_hip.h.
#include <vector>
#include "act_io.hpp"

namespace vp {  // dynamic-LDS arrays of the kernels (per worker thread = per running workgroup): 160 KiB, the size of a CU's LDS
alignas(16) VP_EMU_LDS char smem[160 << 10];
alignas(16) VP_EMU_LDS unsigned char dw_smem[160 << 10];
alignas(16) VP_EMU_LDS unsigned char se_smem[64 << 10];
alignas(16) VP_EMU_LDS unsigned char bk_smem[64 << 10];
alignas(16) VP_EMU_LDS unsigned char det_smem[160 << 10];
alignas(16) VP_EMU_LDS float xs[40 << 10];
alignas(16) VP_EMU_LDS float sh[40 << 10];
alignas(16) VP_EMU_LDS float smem_f[12 << 10];   // ctx_exp_conv1_kernel (round 5)
}  // namespace vp

using namespace vp;

#ifndef VP_EMU_TSAN
// Work-item context switch of the fiber core (System V x86-64): callee-saved registers + stack pointer.  ucontext's
// swapcontext costs two sigprocmask system calls per switch; an emulated MFMA is 64 switches.
asm(R"(
.text
.globl emu_switch
.type emu_switch,@function
emu_switch:
  pushq %rbp
  pushq %rbx
  pushq %r12
  pushq %r13
  pushq %r14
  pushq %r15
  movq %rsp, (%rdi)
  movq %rsi, %rsp
  popq %r15
  popq %r14
  popq %r13
  popq %r12
  popq %rbx
  popq %rbp
  ret
.size emu_switch, .-emu_switch
)");
#endif

static ActView view(void* hi, void* lo, int H, int W, int C) { return ActView{static_cast<half_t*>(hi), static_cast<half_t*>(lo), H, W, C}; }

extern "C" {

int emu_preprocess(const uint8_t* frame, int stride, const int* xtab, const int* ytab, int out_h, int out_w, const int* src_c,
                   const float* mean3, const float* std3, float* out) {
  PreprocessParams p{};
  p.frame = frame; p.stride = stride; p.xtab = xtab; p.ytab = ytab; p.out_h = out_h; p.out_w = out_w; p.out = out;
  for (int c = 0; c < 3; ++c) { p.src_c[c] = src_c[c]; p.mean[c] = mean3[c]; p.stdv[c] = std3[c]; }
  return launch_preprocess(p, nullptr);
}
int emu_preprocess_form(const uint8_t* frame, int stride, const int* xtab, const int* ytab, int out_h, int out_w, const int* src_c,
                        const float* mean3, const float* std3, float* out, int norm_form) {
  PreprocessParams p{};
  p.frame = frame; p.stride = stride; p.xtab = xtab; p.ytab = ytab; p.out_h = out_h; p.out_w = out_w; p.out = out; p.norm_form = norm_form;
  for (int c = 0; c < 3; ++c) { p.src_c[c] = src_c[c]; p.mean[c] = mean3[c]; p.stdv[c] = std3[c]; }
  return launch_preprocess(p, nullptr);
}
int emu_pil_resample(const uint8_t* frame, int stride, int in_h, int in_w, int out_h, int out_w, const int* hb, const int* hk, int hks, const int* vb,
                     const int* vk, int vks, uint8_t* tmp, const int* src_c, const float* mean3, const float* std3, float* out) {
  PilResampleParams p{};
  p.frame = frame; p.stride = stride; p.in_h = in_h; p.in_w = in_w; p.out_h = out_h; p.out_w = out_w; p.hb = hb; p.hk = hk; p.hks = hks;
  p.vb = vb; p.vk = vk; p.vks = vks; p.tmp = tmp; p.out = out;
  for (int c = 0; c < 3; ++c) { p.src_c[c] = src_c[c]; p.mean[c] = mean3[c]; p.stdv[c] = std3[c]; }
  return launch_pil_resample(p, nullptr);
}
int emu_decode_mask(const float* logits, int C, int HW, int mode, uint8_t* out) { return launch_decode_mask(logits, C, HW, mode, out, nullptr); }
int emu_resize_nearest(const uint8_t* src, int sw, const int* ytab, const int* xtab, int oh, int ow, uint8_t* dst) {
  return launch_resize_nearest(src, sw, ytab, xtab, oh, ow, dst, nullptr);
}
int emu_resize_bilinear_f32(const float* src, int sw, const int* yi, const float* yf, const int* xi, const float* xf, int oh, int ow, float* dst) {
  return launch_resize_bilinear_f32(src, sw, yi, yf, xi, xf, oh, ow, dst, nullptr);
}
int emu_viz_blend(const uint8_t* mask, int mw, const int* ytab, const int* xtab, const uint8_t* frame, int stride, int oh, int ow,
                  const uint8_t* lut, int frame_is_rgb, uint8_t* dst) {
  return launch_viz_blend(mask, mw, ytab, xtab, frame, stride, oh, ow, lut, frame_is_rgb, dst, nullptr);
}
int emu_depth_viz(const float* src, size_t n, const uint8_t* lut, uint8_t* dst) {
  unsigned mm[2] = {0xFFFFFFFFu, 0u};
  if (launch_minmax_f32(src, n, mm, nullptr)) return 1;
  return launch_depth_colorize(src, n, mm, lut, dst, nullptr);
}
int emu_stem(const float* in, int H, int W, const float* w, const float* b, void* out_hi, void* out_lo) {
  StemParams p{in, H, W, w, b, view(out_hi, out_lo, H / 2, W / 2, 32)};
  return launch_stem(p, nullptr);
}
int emu_stem_zero(const float* in, int H, int W, const float* w, const float* b, void* out_hi, void* out_lo, unsigned long long* zero, size_t zero_n) {
  StemParams p{in, H, W, w, b, view(out_hi, out_lo, H / 2, W / 2, 32)};
  p.zero = zero;   // the frame's squeeze-excite accumulators, zeroed by this launch (engine.cpp build_backbone)
  p.zero_n = zero_n;
  return launch_stem(p, nullptr);
}
int emu_dwconv(void* in_hi, void* in_lo, int H, int W, int C, void* out_hi, void* out_lo, int OH, int OW, const float* w, const float* b, int k,
               int stride, unsigned long long* sums, int replicas) {
  DwParams p{view(in_hi, in_lo, H, W, C), view(out_hi, out_lo, OH, OW, C), w, b, k, stride, sums, replicas};
  return launch_dwconv(p, nullptr);
}
int emu_dwconv_batched(void* in_hi, void* in_lo, int H, int W, int C, void* out_hi, void* out_lo, int OH, int OW, const float* w, const float* b, int k,
                       int stride, unsigned long long* sums, int replicas, int frames) {
  DwParams p{view(in_hi, in_lo, H, W, C), view(out_hi, out_lo, OH, OW, C), w, b, k, stride, sums, replicas, frames};
  return launch_dwconv(p, nullptr);
}
int emu_mbconv_front(void* in_hi, void* in_lo, int H, int W, int Cin, const void* w_hi, const void* w_lo, const float* b_exp, const float* w_dw,
                     const float* b_dw, void* out_hi, void* out_lo, int Cexp, int k, int stride, unsigned long long* sums, int replicas, const float* w1, int sq,
                     unsigned long long* zsums) {
  MbFrontParams p{};
  static const std::vector<float> ones(4096, 1.0f);   // un-prescaled test weights (MbFrontParams::s_exp)
  p.s_exp = ones.data();
  p.w1 = w1; p.sq = sq; p.zsums = zsums;
  p.in = view(in_hi, in_lo, H, W, Cin);
  p.w_hi = static_cast<const half_t*>(w_hi); p.w_lo = static_cast<const half_t*>(w_lo); p.b_exp = b_exp; p.w_dw = w_dw; p.b_dw = b_dw;
  p.out = view(out_hi, out_lo, H / stride, W / stride, Cexp);
  p.k = k; p.stride = stride; p.sums = sums; p.replicas = replicas;
  return launch_mbconv_front(p, nullptr);
}
int emu_mbconv_back(void* in_hi, void* in_lo, int H, int W, int C, int Creal, const unsigned long long* sums, int replicas, int sq, const float* w1,
                    const float* b1, const float* w2q, const float* b2, int sqp, const float* w, const float* bias, void* res_hi, void* res_lo,
                    void* out_hi, void* out_lo, int Cout, const unsigned long long* zsums) {
  MbBackParams p{};
  static const std::vector<float> ones(4096, 1.0f);   // un-prescaled test weights (MbBackParams::wscale)
  p.wscale = ones.data();
  p.zsums = zsums;
  p.in = view(in_hi, in_lo, H, W, C);
  p.se.sums = sums; p.se.replicas = replicas; p.se.C = C; p.se.Creal = Creal; p.se.sq = sq; p.se.inv_hw = 1.0f / (float)(H * W); p.se.w1 = w1; p.se.b1 = b1;
  p.se.frames = 1;
  p.w2q = w2q; p.b2 = b2; p.sqp = sqp; p.w = w; p.bias = bias;
  if (res_hi) p.res = view(res_hi, res_lo, H, W, Cout);
  p.out = view(out_hi, out_lo, H, W, Cout);
  return launch_mbconv_back(p, nullptr);
}
int emu_se_gate_scale(const unsigned long long* sums, int replicas, int C, int Creal, int sq, float inv_hw, const float* w1, const float* b1,
                      const float* w, void* out_hi, void* out_lo, int rows, const float* w2, const float* b2, int frames) {
  SeParams p{};
  p.sums = sums; p.replicas = replicas; p.C = C; p.Creal = Creal; p.sq = sq; p.inv_hw = inv_hw; p.w1 = w1; p.b1 = b1; p.frames = frames;
  ScaleWParams q{};
  q.w = w; q.out_hi = static_cast<half_t*>(out_hi); q.out_lo = static_cast<half_t*>(out_lo); q.rows = rows; q.C = C; q.w2 = w2; q.b2 = b2;
  q.sq = sq; q.Creal = Creal; q.frames = frames;
  return launch_se_gate_scale(p, q, nullptr);
}
int emu_detect(const float* raw, int num_attrs, int num_boxes, float conf, float iou, float scale, int pad_x, int pad_y, int orig_w, int orig_h, float* boxes, int* cls,
               void* out, int out_cap, int* count) {
  DetectParams p{};
  p.raw = raw; p.num_attrs = num_attrs; p.num_boxes = num_boxes; p.conf_thresh = conf; p.iou_thresh = iou; p.scale = scale; p.pad_x = pad_x; p.pad_y = pad_y;
  p.orig_w = orig_w; p.orig_h = orig_h; p.boxes = boxes; p.cls = cls; p.out = static_cast<Detection*>(out); p.out_cap = out_cap; p.count = count;
  return launch_detect_decode_nms(p, nullptr);
}
int emu_fc(const float* x, const float* w, const float* b, float* out, int N, int K, int act) {
  FcParams p{};
  p.x = x; p.w = w; p.b = b; p.out = out; p.N = N; p.K = K; p.act = act; p.Kstride = K;
  return launch_fc(p, nullptr);
}
int emu_maxpool5(void* src_hi, int H, int W, int C, int src_off, void* dst_hi, int dstC, int dst_off, int nch) {
  return launch_maxpool5(view(src_hi, nullptr, H, W, C), src_off, view(dst_hi, nullptr, H, W, dstC), dst_off, nch, nullptr);
}
int emu_sppf_pool(void* src_hi, void* src_lo, int H, int W, int C, void* dst_hi, void* dst_lo, int dstC, int nch) {
  return launch_sppf_pool(view(src_hi, src_lo, H, W, C), view(dst_hi, dst_lo, H, W, dstC), nch, nullptr);
}
int emu_attention(void* qkv_hi, void* qkv_lo, int H, int W, int heads, int dk, int dv, float scale, void* out_hi, void* out_lo, void* v_hi, void* v_lo, int qblock) {
  AttnParams p{view(qkv_hi, qkv_lo, H, W, heads * (2 * dk + dv)), view(out_hi, out_lo, H, W, heads * dv), view(v_hi, v_lo, H, W, heads * dv), heads, dk, dv, scale, qblock};
  return launch_attention(p, nullptr);
}
int emu_pool_partial(void* hi, void* lo, int H, int W, int C, float* partial, int nslab) {
  PoolParams p{view(hi, lo, H, W, C), partial, nslab};
  return launch_pool_partial(p, nullptr);
}
int emu_fc_pooled(const float* partial, int nslab, int Kstride, float inv_hw, const float* w, const float* b, float* out, int N, int K, int act) {
  FcParams p{};
  p.w = w; p.b = b; p.out = out; p.N = N; p.K = K; p.act = act; p.partial = partial; p.nslab = nslab; p.Kstride = Kstride; p.inv_hw = inv_hw;
  return launch_fc(p, nullptr);
}
// a thread per row (short rows, many of them: AutoDrive's CTX expansion): fp32 rows or e4m3 codes + row scales
int emu_fc_rows(const float* partial, int nslab, int Kstride, float inv_hw, const float* w, const unsigned char* w8, const float* wscale8, const float* b,
                float* out, int N, int K, int act) {
  FcParams p{};
  p.w = w; p.w8 = w8; p.wscale8 = wscale8; p.b = b; p.out = out; p.N = N; p.K = K; p.act = act; p.partial = partial; p.nslab = nslab; p.Kstride = Kstride;
  p.inv_hw = inv_hw; p.rows_kernel = 1;
  return launch_fc(p, nullptr);
}
int emu_ctx_conv1(const float* map, int H, int W, const float* w, const float* b, void* hi, void* lo, int C, int act) {
  CtxConv1Params p{map, H, W, w, b, view(hi, lo, H, W, C), act};
  return launch_ctx_conv1(p, nullptr);
}
// f[i]: hi/lo planes, H, W, C of the five backbone taps; out: 10x20-style map with Cout padded channels
int emu_fusion(void** hi, void** lo, const int* H, const int* W, const int* C, const int* creal, const int* shift, void* out_hi, void* out_lo, int OH,
               int OW, int Cout, int Creal_out, int octets) {
  FusionParams p{};
  for (int i = 0; i < 5; ++i) {
    p.f[i] = view(hi[i], lo ? lo[i] : nullptr, H[i], W[i], C[i]);
    p.creal[i] = creal[i];
    p.shift[i] = shift[i];
  }
  p.out = view(out_hi, out_lo, OH, OW, Cout);
  p.Creal_out = Creal_out;
  p.octets = octets;
  return launch_fusion(p, nullptr);
}
int emu_chan_copy(void* shi, void* slo, int H, int W, int C, int src_off, void* dhi, void* dlo, int dC, int dst_off, int nch) {
  return launch_chan_copy(view(shi, slo, H, W, C), src_off, view(dhi, dlo, H, W, dC), dst_off, nch, nullptr);
}
int emu_dwconv_plain(void* ihi, void* ilo, void* ahi, void* alo, void* ohi, void* olo, int H, int W, int C, const float* w, const float* b) {
  DwPlainParams p{view(ihi, ilo, H, W, C), view(ohi, olo, H, W, C), view(ahi, alo, H, W, C), w, b};
  return launch_dwconv_plain(p, nullptr);
}
int emu_nchw_to_act(const float* src, int Creal, void* hi, void* lo, int H, int W, int C) { return launch_nchw_to_act(src, Creal, view(hi, lo, H, W, C), nullptr); }
int emu_act_to_nchw(void* hi, void* lo, int H, int W, int C, int Creal, float* dst) { return launch_act_to_nchw(view(hi, lo, H, W, C), Creal, dst, nullptr); }

}  // extern "C"
