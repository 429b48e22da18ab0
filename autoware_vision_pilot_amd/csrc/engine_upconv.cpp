// libvp_hip engine: the COMPOSED up-sampling stages (round 6).  The reference's decoders apply ConvTranspose2d(k2, s2) (+ a 1x1 skip link) and then a 3x3
// convolution with no nonlinearity in between (scene_neck.py:29-35,41-46,52-57; scene_seg_head.py:24-29,35-38; scene_3d_head.py:26-31,38-41): one linear
// map.  At load the three weight sets are multiplied out (fp64 accumulation of exact fp32 products, on the device) into
//     Wx[phase][a][b][co][ci]  -- a 4x4 / stride-2 / pad-1 transposed convolution of the LOW-resolution tensor = per output phase a 2x2 convolution,
//     Ws[ty][tx][co][cs]       -- a 3x3 convolution of the skip tensor,
//     bias[row class][column class][co] -- the three biases seen through the high-resolution taps that lie inside the map,
// prescaled per (phase, output channel) row, split into (hi, lo) fp16 planes and packed in the step order of kernels_upconv.hip.  Exact like BatchNorm
// folding; 0.40-0.51x the multiply-adds of the two launches it replaces.  The oracle keeps the reference's three-op form (oracle/nets.py).
#include "engine_internal.hpp"

namespace vp {

namespace {

// (tap index a of the 2x2 low-resolution window, sub-pixel d of the ConvTranspose kernel) that high-resolution tap tt in {0, 1, 2} of the 3x3 window
// around an output pixel of phase bit q touches: high-resolution offset hr = q + tt - 1 -> low-resolution offset o = floor(hr / 2), d = hr - 2 o,
// a = o + 1 - q (the window starts at low-resolution offset q - 1)
inline void tap_of(int q, int tt, int* a, int* d) {
  const int hr = q + tt - 1;
  const int o = hr >= 0 ? hr / 2 : -1;
  *d = hr - 2 * o;
  *a = o + 1 - q;
}

// (hi, lo) fp16 planes of v * pre from a double (the composed weights are carried in fp64 until here)
inline void split_half_d(double v, double pre, half_t* hi, half_t* lo) {
  const double x = v * pre;
  if (!(std::fabs(x) <= 65504.0)) throw RangeError("composed up-sampling weight " + std::to_string(v) + " is outside the fp16 range the matrix pipe carries: re-scale the checkpoint");
  const half_t h = (half_t)x;
  *hi = h;
  *lo = truncate_lo((half_t)(x - (double)h));
}

}  // namespace

// wt [cin][cm][2][2], bt [cm]: ConvTranspose2d(k2, s2);  ws [cm][cs] (1x1), bs [cm]: skip link (cs = 0: none);  w3 [cout][cm][3][3], b3 [cout]
void Engine::compose_upconv(const float* wt, const float* bt, const float* ws, const float* bs, const float* w3, const float* b3, int cin, int cm, int cout,
                            int cs, UpconvComposed* out) {
  if (cin < 1 || cm < 1 || cout < 1 || cs < 0 || !wt || !bt || !w3 || !b3 || (cs > 0 && (!ws || !bs))) throw std::invalid_argument("compose_upconv: bad argument");
  out->cin = cin;
  out->cm = cm;
  out->cout = cout;
  out->cs = cs;
  // K-contiguous operands: W3r[t][co][cm], WTr[d][ci][cm], Br[cs + 1][cm] = the skip weights transposed, then the ConvTranspose bias as one more row;
  // the skip link's bias is a one-column group of its own and the two are added in fp64 (bt + bs in fp32 would round)
  std::vector<float> w3r((size_t)9 * cout * cm), wtr((size_t)4 * cin * cm), br((size_t)(cs + 1) * cm);
  for (int co = 0; co < cout; ++co)
    for (int m = 0; m < cm; ++m)
      for (int t = 0; t < 9; ++t) w3r[((size_t)t * cout + co) * cm + m] = w3[((size_t)co * cm + m) * 9 + t];
  for (int ci = 0; ci < cin; ++ci)
    for (int m = 0; m < cm; ++m)
      for (int d = 0; d < 4; ++d) wtr[((size_t)d * cin + ci) * cm + m] = wt[((size_t)ci * cm + m) * 4 + d];
  for (int m = 0; m < cm; ++m) {
    for (int c = 0; c < cs; ++c) br[(size_t)c * cm + m] = ws[(size_t)m * cs + c];
    br[(size_t)cs * cm + m] = bt[m];
  }
  std::vector<float> bsr;
  if (cs > 0) bsr.assign(bs, bs + cm);
  float* d_w3r = dupload(w3r);
  float* d_wtr = dupload(wtr);
  float* d_br = dupload(br);
  float* d_bsr = cs > 0 ? dupload(bsr) : nullptr;
  const size_t n_wx = (size_t)16 * cout * cin, n_ws = (size_t)9 * cout * (cs + 1);
  double* d_wx = static_cast<double*>(dalloc(n_wx * sizeof(double), false));
  double* d_ws = static_cast<double*>(dalloc(n_ws * sizeof(double), false));
  std::vector<ComposeGemmParams> gx(16), gs(cs > 0 ? 18 : 9);
  for (int phase = 0; phase < 4; ++phase) {
    const int py = phase >> 1, px = phase & 1;
    for (int k = 0; k < 4; ++k) {
      ComposeGemmParams& g = gx[phase * 4 + k];
      g = ComposeGemmParams{};
      g.M = cout;
      g.N = cin;
      g.K = cm;
      g.c = d_wx + (size_t)(phase * 4 + k) * cout * cin;
    }
    for (int ty = 0; ty < 3; ++ty)
      for (int tx = 0; tx < 3; ++tx) {
        int a, dy, b, dx;
        tap_of(py, ty, &a, &dy);
        tap_of(px, tx, &b, &dx);
        ComposeGemmParams& g = gx[phase * 4 + a * 2 + b];
        g.a[g.pairs] = d_w3r + (size_t)(ty * 3 + tx) * cout * cm;
        g.b[g.pairs] = d_wtr + (size_t)(dy * 2 + dx) * cin * cm;
        ++g.pairs;
      }
  }
  for (int t = 0; t < 9; ++t) {
    ComposeGemmParams& g = gs[t];
    g = ComposeGemmParams{};
    g.M = cout;
    g.N = cs + 1;
    g.K = cm;
    g.c = d_ws + (size_t)t * cout * (cs + 1);
    g.a[0] = d_w3r + (size_t)t * cout * cm;
    g.b[0] = d_br;
    g.pairs = 1;
    if (cs > 0) {  // V_skip[t][co] = sum_cm W3[co][cm][t] * bs[cm]: its own 1-column group, added on the host (keeps bt and bs exact)
      ComposeGemmParams& h = gs[9 + t];
      h = ComposeGemmParams{};
      h.M = cout;
      h.N = 1;
      h.K = cm;
      h.a[0] = g.a[0];
      h.b[0] = d_bsr;
      h.pairs = 1;
    }
  }
  double* d_vs = cs > 0 ? static_cast<double*>(dalloc((size_t)9 * cout * sizeof(double), false)) : nullptr;
  for (int t = 0; t < 9 && cs > 0; ++t) gs[9 + t].c = d_vs + (size_t)t * cout;
  ComposeGemmParams* d_gx = dupload(gx);
  std::vector<ComposeGemmParams> gs_main(gs.begin(), gs.begin() + 9);
  ComposeGemmParams* d_gs = dupload(gs_main);
  VP_HIP_CHECK(launch_compose_gemm(d_gx, 16, cout, cin, stream_));
  VP_HIP_CHECK(launch_compose_gemm(d_gs, 9, cout, cs + 1, stream_));
  ComposeGemmParams* d_gv = nullptr;
  if (cs > 0) {
    std::vector<ComposeGemmParams> gv(gs.begin() + 9, gs.end());
    d_gv = dupload(gv);
    VP_HIP_CHECK(launch_compose_gemm(d_gv, 9, cout, 1, stream_));
  }
  out->wx.resize(n_wx);
  std::vector<double> wsb(n_ws), vs(cs > 0 ? (size_t)9 * cout : 0);
  copy_d2h(out->wx.data(), d_wx, n_wx * sizeof(double));
  copy_d2h(wsb.data(), d_ws, n_ws * sizeof(double));
  if (cs > 0) copy_d2h(vs.data(), d_vs, vs.size() * sizeof(double));
  for (void* q : {(void*)d_w3r, (void*)d_wtr, (void*)d_br, (void*)d_bsr, (void*)d_wx, (void*)d_ws, (void*)d_vs, (void*)d_gx, (void*)d_gs, (void*)d_gv})
    if (q) dfree(q);
  out->ws.assign((size_t)9 * cout * cs, 0.0);
  std::vector<double> V((size_t)9 * cout);
  for (int t = 0; t < 9; ++t)
    for (int co = 0; co < cout; ++co) {
      const double* row = wsb.data() + ((size_t)t * cout + co) * (cs + 1);
      for (int c = 0; c < cs; ++c) out->ws[((size_t)t * cout + co) * cs + c] = row[c];
      V[(size_t)t * cout + co] = row[cs] + (cs > 0 ? vs[(size_t)t * cout + co] : 0.0);
    }
  // bias seen by an output pixel: b3 + the intermediate biases through the taps that lie inside the map (class 0: first row / column of the
  // high-resolution map, tap 0 outside; class 2: last, tap 2 outside; class 1: all three)
  out->bias.assign((size_t)9 * cout, 0.0);
  for (int rc = 0; rc < 3; ++rc)
    for (int cc = 0; cc < 3; ++cc)
      for (int co = 0; co < cout; ++co) {
        double b = (double)b3[co];
        for (int ty = (rc == 0 ? 1 : 0); ty <= (rc == 2 ? 1 : 2); ++ty)
          for (int tx = (cc == 0 ? 1 : 0); tx <= (cc == 2 ? 1 : 2); ++tx) b += V[(size_t)(ty * 3 + tx) * cout + co];
        out->bias[(size_t)(rc * 3 + cc) * cout + co] = b;
      }
}

// both precision modes since the fp16 form of the kernel exists (X1, kernels_upconv.hip); VP_UPCONV=0 (developer knob): the three-operator plan
bool Engine::upconv_wanted() const { return !fp8_storage() && !dev_option_is("VP_UPCONV", '0') && (split() || !dev_option_is("VP_UPCONV_F16", '0')); }

// One composed stage as ONE op of the plan.  in: low-resolution tensor (H x W x cin), skip_in: 2H x 2W x cs or null.
Act* Engine::add_upconv(const std::string& name, const Act* in, const Act* skip_in, const std::vector<float>& wt, const std::vector<float>& bt,
                        const std::vector<float>& ws, const std::vector<float>& bs, const std::vector<float>& w3, const std::vector<float>& b3, int cm,
                        int cout, int act, int shape, int nsplit, const std::string& out_name) {
  const bool x1 = !split();   // VP_FP16 engines: 64-channel chunks, the halves in the two weight planes (kernels_upconv.hip X1)
  if (x1 && in->C % 64 != 0) throw std::invalid_argument("composed up-sampling stage in an fp16 engine: input channels a multiple of 64: " + name);
  const int cin = in->Creal, cin_pad = in->C, cs = skip_in ? skip_in->Creal : 0, cs_pad = skip_in ? skip_in->C : 0;
  if (wt.size() != (size_t)cin * cm * 4 || bt.size() != (size_t)cm) throw std::runtime_error("convT weight size mismatch: " + name);
  if (w3.size() != (size_t)cout * cm * 9 || b3.size() != (size_t)cout) throw std::runtime_error("conv weight size mismatch: " + name);
  if (skip_in && (ws.size() != (size_t)cm * cs || bs.size() != (size_t)cm)) throw std::runtime_error("skip conv weight size mismatch: " + name);
  if (skip_in && (skip_in->H != in->H * 2 || skip_in->W != in->W * 2)) throw std::runtime_error("skip tensor size mismatch: " + name);
  // the load-time range guard of every other layer (engine.cpp split_half): a weight of the checkpoint beyond the fp16 range (or non-finite) is refused --
  // the composed, prescaled product could carry it, the activations it produces would not survive
  for (const std::vector<float>* v : {&wt, &w3, skip_in ? &ws : nullptr})
    if (v)
      for (float x : *v)
        if (!(std::fabs(x) <= 65504.0f))
          throw RangeError("weight " + std::to_string(x) + " is outside the fp16 range the matrix pipe carries (|w| <= 65504): re-scale the checkpoint");
  UpconvComposed cw;
  compose_upconv(wt.data(), bt.data(), skip_in ? ws.data() : nullptr, skip_in ? bs.data() : nullptr, w3.data(), b3.data(), cin, cm, cout, cs, &cw);

  const int ncols = round_up(cout, 32), coutw = round_up(ncols, 128);
  // the chunk list's channel counts: fp16 engines walk 64-channel chunks = descriptors over halved counts (the skip tensor rounded up to 64 first)
  const int cin_v = x1 ? cin_pad / 2 : cin_pad, cs_v = x1 ? round_up(cs_pad, 64) / 2 : cs_pad, chm = x1 ? 2 : 1, chunk_w = x1 ? 64 : 32;
  const int S = upconv_steps(cin_v, cs_v), n_chunks = upconv_chunks(cin_v, cs_v);
  std::vector<half_t> hi((size_t)4 * S * coutw * 32, (half_t)0.0f), lo(hi.size(), (half_t)0.0f);
  std::vector<float> post((size_t)4 * coutw, 1.0f), bias((size_t)9 * coutw, 0.0f);
  for (int k = 0; k < 9; ++k)
    for (int co = 0; co < cout; ++co) bias[(size_t)k * coutw + co] = (float)cw.bias[(size_t)k * cout + co];
  parallel_rows(4 * cout, [&](int r_begin, int r_end) {
    for (int r = r_begin; r < r_end; ++r) {
      const int phase = r / cout, co = r - phase * cout, py = phase >> 1, px = phase & 1;
      // prescale per (phase, output channel) row over everything its K axis carries: the four 2x2 taps of x and the nine skip taps
      double amax = 0.0;
      for (int k = 0; k < 4; ++k) {
        const double* row = cw.wx.data() + ((size_t)(phase * 4 + k) * cout + co) * cin;
        for (int ci = 0; ci < cin; ++ci) amax = std::max(amax, std::fabs(row[ci]));
      }
      for (int t = 0; t < 9 && cs > 0; ++t) {
        const double* row = cw.ws.data() + ((size_t)t * cout + co) * cs;
        for (int c = 0; c < cs; ++c) amax = std::max(amax, std::fabs(row[c]));
      }
      const int sexp = prescale_exp((float)amax);
      const double pre = std::ldexp(1.0, sexp);
      post[(size_t)phase * coutw + co] = std::ldexp(1.0f, -sexp);
      for (int c = 0; c < n_chunks; ++c) {
        const UpconvChunk d = upconv_chunk(c, py, px, cin_v, cs_v);
        const int creal = d.skip ? cs : cin;
        for (int k = 0; k < d.nt; ++k) {
          const int a = d.a0 + (d.nb == 2 ? (k >> 1) : k), b = d.b0 + (d.nb == 2 ? (k & 1) : 0);
          const double* row;
          if (d.skip) {
            const int ty = d.qy != py ? 2 * a : 1, tx = d.qx != px ? 2 * b : 1;
            row = cw.ws.data() + ((size_t)(ty * 3 + tx) * cout + co) * cs;
          } else {
            row = cw.wx.data() + ((size_t)(phase * 4 + a * 2 + b) * cout + co) * cin;
          }
          for (int i = 0; i < chunk_w && chm * d.ch0 + i < creal; ++i) {
            const size_t dst = upconv_pack_index(phase, d.step0 + k, co, i & 31, S, coutw);
            if (x1) {   // one fp16 value per weight; plane = which half of the 64-channel chunk
              half_t h, l;
              split_half_d(row[chm * d.ch0 + i], pre, &h, &l);
              (i >> 5 ? lo : hi)[dst] = h;
            } else {
              split_half_d(row[d.ch0 + i], pre, &hi[dst], &lo[dst]);
            }
          }
        }
      }
    }
  });
  Act* out = new_act(out_name.empty() ? name : out_name, cout, in->H * 2, in->W * 2);
  UpconvParams p{};
  p.in_hi = in->hi;
  p.in_lo = x1 ? nullptr : in->lo;
  p.sk_hi = skip_in ? skip_in->hi : nullptr;
  p.sk_lo = (skip_in && !x1) ? skip_in->lo : nullptr;
  p.H = in->H;
  p.W = in->W;
  p.Cin = cin_pad;
  p.Cs = cs_pad;
  p.w_hi = dupload(hi);
  p.w_lo = dupload(lo);
  wbytes_[1] += 2 * (hi.size() + lo.size());
  p.bias = dupload(bias);
  p.wscale = dupload(post);
  p.CoutW = coutw;
  p.Ncols = ncols;
  p.Cstore = out->C;
  p.out_hi = out->hi;
  p.out_lo = x1 ? nullptr : out->lo;
  p.act = (x1 && act == ACT_GELU) ? ACT_GELU_F16 : act;   // VP_FP16: reduced-instruction activation (common.hpp), as push_conv_op
  // shape and K slices: the dispatch rule (measured, profiles/r06_*), or the caller's choice
  const long long t16 = (long long)((in->H + 15) / 16) * ((in->W + 15) / 16), t8 = (long long)((in->H + 7) / 8) * ((in->W + 15) / 16);
  const int n_co = coutw / 128;
  int sh = shape;
  {
    int ot = -1, on = 0;   // this stage's shape / K slices forced by name (VP_PLAN_OVERRIDE: the in-frame tuner)
    if (sh < 0 && plan_override(name, &ot, &on)) {
      sh = ot;
      if (on > 0 && nsplit <= 0) nsplit = on;
    }
  }
  if (sh < 0) {
    // Measured per stage on the MI355X (profiles/r06_upconv_shapes.txt, SceneSeg, us, shape 6 / shape 7): 10x20 -> 20x40 67.2 / 73.1, 20x40 -> 40x80
    // 87.8 / 77.5, 40x80 -> 80x160 106.4 / 106.2, 80x160 -> 160x320 119.8 / 125.5, 160x320 -> 320x640 (128 channels) 118.4 / 103.9: the 8-row patches
    // where the map's rows leave a 16-row patch mostly empty (20 and 40 rows), and where the K loop is short (Cin <= 128: two independent workgroups
    // per CU overlap prologue and epilogue, as for kernels_conv3x3_x3.hip); the 8-wave shape elsewhere
    const double eff6 = (double)in->H * in->W / (double)(t16 * 256), eff7 = (double)in->H * in->W / (double)(t8 * 128);
    const char* e = dev_option("VP_UPCONV_SHAPE");   // developer knob: 6 / 7 on every stage
    sh = e ? std::atoi(e) : ((eff7 > 1.1 * eff6 || cin_pad <= 128) ? 7 : 6);
  }
  if (sh != 6 && sh != 7) throw std::invalid_argument("composed up-sampling stage: shape 6 (16x16 patches, 8 waves) or 7 (8x16 patches, 4 waves): " + name);
  const long long blocks = (sh == 6 ? t16 : t8) * 4 * n_co;
  int ns = nsplit;
  if (ns <= 0) {
    // one round of workgroups (256 at one per CU for shape 6, 512 for shape 7), at least ~12 steps per slice, fp32 partials under ~26 MB
    const long long slots = sh == 6 ? 256 : 512;
    ns = (int)std::max<long long>(1, slots / blocks);
    const double slice_mb = 4.0 * in->H * in->W * coutw * 4.0 / 1e6;
    while (ns > 1 && ns * slice_mb > 26.0) --ns;
    while (ns > 1 && S / ns < 12) --ns;
    if (const char* e = dev_option("VP_UPCONV_NSPLIT")) ns = std::max(1, std::atoi(e));   // developer knob
  }
  ns = std::max(1, std::min(ns, n_chunks));
  p.nsplit = ns;
  p.partial = ns > 1 ? static_cast<float*>(dalloc((size_t)ns * 4 * in->H * in->W * coutw * sizeof(float), false)) : nullptr;
  if (!upconv_supported(p, sh)) throw std::invalid_argument("composed up-sampling stage: shape not covered: " + name);
  Op op;
  op.name = name;
  op.kernel = std::string(x1 ? "upconv_x1" : "upconv_x3") + (sh == 6 ? "w8<co128,px256" : "w4<co128,px128") + (x1 ? ",k64>" : ">") + (ns > 1 ? "+splitk" : "");
  if (ns > 1) op.launch = "nsplit=" + std::to_string(ns);
  const double M2 = 4.0 * in->H * in->W;
  // algorithmic work of the REFERENCE formulation (ConvTranspose on input pixels x 4 taps, 1x1 skip and 3x3 on output pixels) ...
  op.flops = 2.0 * (M2 * cin * cm + M2 * cs * cm + M2 * 9.0 * cm * cout);
  // ... and what this launch executes (real channels): 4 taps of x and 9 of the skip tensor per output pixel
  op.flops_executed = 2.0 * M2 * cout * (4.0 * cin + 9.0 * cs);
  op.bytes = (x1 ? 2.0 : 4.0) * ((M2 / 4 * cin + M2 * cs + M2 * cout) + cout * (16.0 * cin + 9.0 * cs));
  op.run = [p, sh](hipStream_t st) { return launch_upconv(p, sh, st); };
  ops_.push_back(std::move(op));
  return out;
}

}  // namespace vp
