// libvp_hip engine, part 2 of 3: the DISPATCH RULES -- which kernel, tile and split factor every convolution of a plan gets, and the
// weight packing that kernel expects.  Every rule here is a measured choice (the comment at the rule names the profile); the outcome for
// every layer of every network is pinned by tests/golden/kernel_plan.json (tests/test_engine_emulated.py).
#include "engine_internal.hpp"

namespace vp {

// Round 5: the dispatch rules trade a layer's OWN latency for CU-time where the rest of the frame can use the freed CUs (decode_layer_5 on 100
// workgroups, the neck's map layers on half the workgroups, the K = 288 ConvTranspose on 128) -- right for a camera with forked heads and for several
// cameras per GPU, the configurations bench.py measures; a host that runs ONE network on ONE camera, one frame at a time, has nothing to put on the
// freed CUs and pays 3-7 % of its frame (SceneSeg alone 1.79 -> 1.91 ms).  The creation flag VP_PLAN_LATENCY (round 6: per ENGINE, OR-ed into vp_create's
// precision argument) selects the round-4 choices for that engine; the developer option VP_PLAN_TARGET=latency does it for every engine created while it
// is set (A/B runs of bench.py).  Either way the choice changes vp_plan_hash().  Default / "throughput": the rules above.
bool Engine::plan_latency() const { return (precision_ & 32) != 0 || dev_option_is("VP_PLAN_TARGET", 'l'); }

// Developer option VP_PLAN_OVERRIDE = "<layer>=<tile>[:<nsplit>];<layer>=..." (round 6, tools/plan_search.py): ONE layer's kernel choice, by name --
// <layer> matches a launch whose name ENDS with it ("decode_layer_5" = that layer of every network created while the option is set), <tile> is the
// halo-tile number of a 3x3 convolution (1 / 3: halo kernel 128 / 64 channels, 6 / 7 / 8: pipelined shapes, 11 / 12: map kernels) or the shape (6 / 7)
// of a composed up-sampling stage, <nsplit> its K slices (absent / 0: the rule's).  The in-frame tuner flips single layers with it; the rules below
// are what its table (profiles/r06_plan_search.tsv) says.  Part of the plan hash like every option.
bool plan_override(const std::string& name, int* tile, int* nsplit) {
  const char* e = dev_option("VP_PLAN_OVERRIDE");
  if (!e) return false;
  const std::string s(e);
  size_t pos = 0;
  while (pos < s.size()) {
    size_t end = s.find(';', pos);
    if (end == std::string::npos) end = s.size();
    const std::string item = s.substr(pos, end - pos);
    pos = end + 1;
    const size_t eq = item.rfind('=');
    if (eq == std::string::npos || eq == 0) continue;
    const std::string key = item.substr(0, eq), val = item.substr(eq + 1);
    if (name.size() < key.size() || name.compare(name.size() - key.size(), key.size(), key) != 0) continue;
    const size_t colon = val.find(':');
    *tile = std::atoi(val.substr(0, colon).c_str());
    *nsplit = colon == std::string::npos ? 0 : std::atoi(val.substr(colon + 1).c_str());
    return true;
  }
  return false;
}

// ------------------------------------------------------------------------------------------- conv planning
void Engine::choose_conv_cfg(int M, int ncols, int cin_pad, int ks, const ConvOpts& o, PackedConv* pc) {
  auto cdiv = [](long long a, long long b) { return (a + b - 1) / b; };
  int tile;
  if (o.tile >= 0) {
    tile = o.tile;
  } else if (ncols <= 32) {
    tile = 3;
  } else if (ncols % 128 == 0 && (cdiv(M, 128) * (ncols / 128) >= 192 || (M <= 256 && ncols >= 2048))) {
    tile = 0;  // second case: weight-dominated GEMMs on tiny maps (first up-sampling stage: 200 pixels x 5120 rows): 29 vs 36 us
  } else if (cdiv(M, 128) * cdiv(ncols, 64) >= 128 || M >= 2048) {
    tile = 1;
  } else {
    tile = 2;
  }
  pc->tile = tile;
  pc->bk = o.bk > 0 ? o.bk : 64;  // 128-byte rows per load, half the K steps of BK=32
  if (cin_pad % pc->bk != 0) pc->bk = 32;
  pc->CoutW = round_up(ncols, conv_tile_co(tile));
  const long long blocks = cdiv(M, conv_tile_px(tile)) * (pc->CoutW / conv_tile_co(tile));
  const int S = ks * ks * (cin_pad / pc->bk);
  int ns = 1;
  if (o.nsplit > 0) {
    ns = o.nsplit;
  } else if (blocks < 128 && S >= 32) {  // split-K pays only for long K loops: it costs a second (finish) launch
    ns = (int)std::min<long long>(cdiv(384, blocks), std::max(1, S / 8));
    ns = std::max(1, std::min(ns, 32));
  } else if (ks == 1 && blocks < 64) {
    // long-K 1x1 GEMMs on small maps (the MBConv projections of stages 4-7: K = 480..1152 on 200-800 pixels, 12-42 workgroups
    // walking 8-18 dependent staging steps): slices of >= 3 steps up to ~128 workgroups.  Measured (parity mode, per launch incl.
    // the finish kernel): 25 -> 15 us on stages 6 / 7, 20 -> 15 on stage 5, SceneSeg single stream 2.076 -> 2.015 ms.
    // VP_PROJ_SPLIT = minimum steps per slice (0 = never split).
    const char* e = dev_option("VP_PROJ_SPLIT");
    const int min_steps = e ? std::atoi(e) : 3;
    if (min_steps > 0 && S >= 2 * min_steps) ns = std::max(1, std::min<int>(S / min_steps, (int)cdiv(128, blocks)));
  }
  pc->nsplit = std::min(ns, std::max(1, S));
}

void Engine::push_conv_op(const std::string& name, const Act* in, const PackedConv& pc, int ks, int ncols, const ConvOpts& o, Act* out,
                          int store_mode, int cout_real) {
  ConvGemmParams p{};
  const int cstride = std::max(1, o.stride);
  p.in_hi = in->hi;
  p.in_lo = in->lo;
  p.H = in->H / cstride;  // output size (== input size for stride 1)
  p.W = in->W / cstride;
  p.stride = cstride;
  p.Hin = in->H;
  p.Win = in->W;
  p.post_act = o.post_act;
  p.Cin = in->C;
  p.w_hi = pc.w_hi;
  p.w_lo = pc.w_lo;
  p.bias = pc.bias;
  p.wscale = pc.wscale;
  p.w8 = pc.w8;
  p.ks = ks;
  p.Ncols = ncols;
  p.CoutW = pc.CoutW;
  p.act = (o.act >= ACT_GELU && o.act <= ACT_SIGMOID && !split()) ? (o.act | ACT_F16) : o.act;  // VP_FP16: reduced-instruction activations (common.hpp)
  p.res_mode = o.res_mode;
  p.res_hi = o.res ? o.res->hi : nullptr;
  p.res_lo = o.res ? o.res->lo : nullptr;
  p.store_mode = store_mode;
  p.out_hi = out ? out->hi : nullptr;
  p.out_lo = out ? out->lo : nullptr;
  p.Cstore = out ? out->C : 0;
  p.out_f32 = o.logits_out;
  p.Creal = cout_real;
  p.zeros = static_cast<const half_t*>(zero_page());
  // the head's logits convolution decodes the mask in its epilogue (one launch and a 2.4 MB re-read less per network)
  const bool fuse_decode = store_mode == STORE_NCHW_F32 && o.logits_out == d_logits_ && d_mask_ && cout_real <= 8 && ncols <= 32 && kind_ >= 0 &&
                           kind_ != 4 && !(dev_option_is("VP_FUSE_DECODE", '0'));
  if (fuse_decode) {
    p.mask_out = d_mask_;
    decode_fused_ = true;
  }
  p.nsplit = pc.nsplit;
  if (o.in2) {
    p.Cin2 = o.in2->C;
    p.in2_delta_hi = o.in2->hi - in->hi;
    p.in2_delta_lo = (in->lo && o.in2->lo) ? o.in2->lo - in->lo : 0;
  }
  const int M = p.H * p.W;
  p.partial = (pc.nsplit > 1 || pc.tile == 111 || pc.tile == 112) ? static_cast<float*>(dalloc((size_t)pc.nsplit * M * pc.CoutW * sizeof(float), false)) : nullptr;
  const int tile = pc.tile, bk = pc.bk;
  const bool sp = split();
  Op op;
  op.name = name;
  if (pc.nsplit > 1) op.launch = "nsplit=" + std::to_string(pc.nsplit);   // the tag only says "+splitk"
  op.flops = 2.0 * M * (double)cout_real * (store_mode == STORE_SHUFFLE2 ? 4 : 1) * in->Creal * ks * ks;
  const double esz = sp ? 4.0 : 2.0;
  op.bytes = esz * ((double)M * in->Creal + (double)M * (store_mode == STORE_SHUFFLE2 ? 4 : 1) * cout_real) +
             (pc.w8 ? 1.0 : esz) * ((double)cout_real * (store_mode == STORE_SHUFFLE2 ? 4 : 1) * in->Creal * ks * ks);
  if (o.in2) {  // fused skip-link: extra K columns read at every OUTPUT pixel
    op.flops += 2.0 * M * 4.0 * cout_real * o.in2->Creal;
    op.bytes += esz * ((double)M * 4.0 * o.in2->Creal + (double)cout_real * o.in2->Creal);
  }
  if (tile >= 100) {
    int ht = tile - 100;
    // ---- optional per-layer tile autotune (VP_AUTOTUNE=1 enables; measured +-1 % on the frames-in-flight bench, so
    // the static heuristic is the default): time the tile shapes that share this weight packing and keep the fastest.  The K order per output is identical for every tile, so the choice never changes a result bit.
    const char* at_env = dev_option("VP_AUTOTUNE");
    const bool explicit_tile = o.tile >= 100;
    if (!explicit_tile && at_env && at_env[0] == '1' && kind_ >= 0) {
      std::vector<int> cand;
      for (int c : {0, 1, 2, 3, 4}) {
        if (sp && (c == 0 || c == 2)) continue;
        if (pc.CoutW % halo_tile_co(c) != 0) continue;
        if (halo_tile_co(c) > pc.CoutW) continue;
        if (ncols <= 32 && c != 4) continue;
        if (ncols > 32 && c == 4) continue;
        cand.push_back(c);
      }
      // Cost = wall time of 6 launches spread over 3 streams: the engine is meant to run with several frames in
      // flight, so what counts is the CU-time a tile shape consumes under contention, not its latency alone.
      hipStream_t ts[3] = {stream_, nullptr, nullptr};
      VP_HIP_CHECK(hipStreamCreateWithFlags(&ts[1], hipStreamNonBlocking));
      VP_HIP_CHECK(hipStreamCreateWithFlags(&ts[2], hipStreamNonBlocking));
      double best = 1e30;
      int best_c = ht;
      for (int c : cand) {
        hipError_t e = launch_conv3x3_halo(p, c, sp, stream_);  // warm-up (also sets the LDS attribute)
        if (e != hipSuccess) continue;
        VP_HIP_CHECK(hipStreamSynchronize(stream_));
        const auto t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < 6 && e == hipSuccess; ++r) e = launch_conv3x3_halo(p, c, sp, ts[r % 3]);
        for (hipStream_t s : ts) VP_HIP_CHECK(hipStreamSynchronize(s));
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        if (e == hipSuccess && us < best) {
          best = us;
          best_c = c;
        }
      }
      hipStreamDestroy(ts[1]);
      hipStreamDestroy(ts[2]);
      ht = best_c;
    }
    if (ht == 11) {
      if (!conv3x3_map_supported(p)) throw std::invalid_argument("halo tile 11 (map kernel): 3x3 stride 1, 20x40 or 10x20 regions: " + name);
      op.kernel = std::string("conv3x3_map<co32,") + (conv3x3_map_geometry(p.H, p.W) == 1 ? "px800," : "px200,") + (sp ? "x3>" : "x1,k32>") + "+splitk";
      op.run = [p](hipStream_t st) { return launch_conv3x3_map(p, st); };
      ops_.push_back(std::move(op));
      return;
    }
    if (ht == 12) {
      if (!conv3x3_map2_supported(p)) throw std::invalid_argument("halo tile 12 (map kernel, 64-channel slabs): parity mode, 3x3 stride 1, 20x40 regions, output channels padded to 64: " + name);
      op.kernel = "conv3x3_map2<co64,px800,x3>+splitk";
      op.launch += " wgs=" + std::to_string((p.H / 20) * (p.W / 40) * (p.CoutW / 64) * p.nsplit);
      op.run = [p](hipStream_t st) { return launch_conv3x3_map2(p, st); };
      ops_.push_back(std::move(op));
      return;
    }
    if (ht >= 6 && ht <= 9) {
      if (!conv3x3_x3_supported(p, ht)) throw std::invalid_argument("halo tiles 6 - 9 (pipelined kernels): conv + bias + {GELU, none}, NHWC, 128- (64-) channel tiles, fp16: input channels a multiple of 64; any epilogue with split-K (7 / 8): " + name);
      // fp16 engines ("x1"): the same schedule on 64-channel chunks, the chunk's two halves in the two LDS planes (kernels_conv3x3_x3.hip X1)
      op.kernel = std::string(sp ? "conv3x3_x3" : "conv3x3_x1") + (ht == 6 ? "w8<co128,px256" : (ht == 7 ? "w4<co128,px128" : (ht == 8 ? "w4<co64,px128" : "w4<co64,px256"))) + (sp ? ">" : ",k64>") + (pc.nsplit > 1 ? "+splitk" : "");
      op.run = [p, ht](hipStream_t st) { return launch_conv3x3_x3(p, ht, st); };
      ops_.push_back(std::move(op));
      return;
    }
    // the heads' logits convolution: weights stationary in registers, 16x16x32 MFMA, LDS-DMA halo (kernels_head.hip); same weight
    // packing as halo tile 4.  VP_HEAD_CONV=0 keeps the halo kernel.
    if (ht == 4 && o.tile < 0 && head_conv_supported(p) && !(dev_option_is("VP_HEAD_CONV", '0'))) {
      const void* zeros = zero_page();
      op.kernel = std::string("head_conv3x3<c") + std::to_string(p.Cin) + (sp ? ",x3>" : ",x1>") + (fuse_decode ? "+decode" : "");
      op.run = [this, p, zeros, fuse_decode](hipStream_t st) {
        ConvGemmParams q = p;
        if (fuse_decode) q.decode_mode = decode_mode_;  // vp_set_decode_mode may change it between frames (it invalidates the graph)
        return launch_head_conv(q, zeros, st);
      };
      ops_.push_back(std::move(op));
      return;
    }
    // ",regepi": the register-GELU single-pass epilogue instantiation (same condition as launch_halo_cfg)
    const bool regepi = !sp && p.act == ACT_GELU_F16 && p.res_mode == RES_NONE && p.store_mode == STORE_NHWC && p.nsplit == 1 && !p.w8;
    op.kernel = "conv3x3_halo<co" + std::to_string(halo_tile_co(ht)) + ",px" + std::to_string(halo_tile_px(ht)) + (sp ? ",x3" : ",x1") +
                (p.w8 ? ",w8" : "") + (regepi ? ",regepi>" : ">") + (pc.nsplit > 1 ? "+splitk" : "");
    if (fuse_decode) {
      op.kernel += "+decode";
      op.run = [this, p, ht, sp](hipStream_t st) {
        ConvGemmParams q = p;
        q.decode_mode = decode_mode_;  // vp_set_decode_mode may change it between frames (it invalidates the graph)
        return launch_conv3x3_halo(q, ht, sp, st);
      };
    } else {
      op.run = [p, ht, sp](hipStream_t st) { return launch_conv3x3_halo(p, ht, sp, st); };
    }
  } else if (tile == 6) {  // its weights are packed in its own layout (add_convT*): no other kernel may take this launch
    if (!gemm_dma_supported(p, sp))
      throw std::invalid_argument("LDS-DMA GEMM kernel (tile 6): ConvTranspose k2 s2 (+ skip link) + bias, 4 * Cout and Cout multiples of 256, K >= 256, >= 128 pixels: " + name);
    op.kernel = std::string("gemm_dma<co256,px128,") + (sp ? "x3>" : "x1>") + (pc.nsplit > 1 ? "+splitk" : "");
    op.run = [p](hipStream_t st) { return launch_gemm_dma(p, st); };
  } else if (tile == 5 || (!(dev_option_is("VP_CONVT_RS", '0')) && convt_rs_supported(p, sp))) {
    if (!convt_rs_supported(p, sp))
      throw std::invalid_argument("register-stationary ConvTranspose kernel (tile 5): k2 s2 + bias, K = 128 or 256 + 32 (skip link), map width a multiple of 32, >= 2048 pixels: " + name);
    op.kernel = "convt_rs<k" + std::to_string(p.Cin + p.Cin2) + (sp ? ",x3>" : ",x1>");
    // Round 5: the K = 256 + 32 case (ConvTranspose + skip link 80x160 -> 160x320: four quadrant slices per pixel range) runs on 32 pixel-tile groups per
    // slice = 128 persistent workgroups instead of 256.  An HBM-side byte mover does not need every CU to pull its bytes: ALONE the launch is slower
    // (37.8 -> 52.7 us), with the other head or other cameras on the freed CUs the frame gains 1.2-1.3 % on two boxes and one camera's p50 is level or
    // better (profiles/r05_convt_groups_ab.txt: 403.6 / 402.9 -> 408.4, 414.4 / 413.5 -> 418.7 / 419.9 frames/s; 24 groups +0.7 %, the K = 128 case at
    // 192 / 128 groups +-0).  VP_CONVT_RS_GROUPS_K288 / VP_CONVT_RS_GROUPS (developer knobs, one per shape case) override; both are read HERE, at plan
    // time, and are part of the plan hash.
    if (p.Cin2 > 0 && !plan_latency()) p.rs_groups = 32;
    if (const char* e = dev_option(p.Cin2 > 0 ? "VP_CONVT_RS_GROUPS_K288" : "VP_CONVT_RS_GROUPS")) p.rs_groups = std::max(1, std::atoi(e));
    if (p.rs_groups > 0) op.launch += "groups=" + std::to_string(p.rs_groups);
    op.run = [p](hipStream_t st) { return launch_convt_rs(p, st); };
  } else {
    const int epi = (ks == 1 && !sp && !p.w8) ? regepi_case(p, conv_tile_co(tile), sp) : 0;  // same rule as launch_cfg (kernels_conv.hip)
    op.kernel = "conv_gemm<bk" + std::to_string(bk) + ",co" + std::to_string(conv_tile_co(tile)) + ",px" +
                std::to_string(conv_tile_px(tile)) + (sp ? ",x3" : ",x1") + (p.w8 ? ",w8" : "") + (epi ? ",regepi" + std::to_string(epi) + ">" : ">") +
                (pc.nsplit > 1 ? "+splitk" : "");
    if (fuse_decode) {
      op.kernel += "+decode";
      op.run = [this, p, tile, bk, sp](hipStream_t st) {
        ConvGemmParams q = p;
        q.decode_mode = decode_mode_;
        return launch_conv_gemm(q, tile, bk, sp, st);
      };
    } else {
      op.run = [p, tile, bk, sp](hipStream_t st) { return launch_conv_gemm(p, tile, bk, sp, st); };
    }
  }
  ops_.push_back(std::move(op));
}

// w: [cout][cin][ks][ks] fp32 (already BN-folded where applicable), b: [cout]
Act* Engine::add_conv(const std::string& name, const Act* in, const std::vector<float>& w, const std::vector<float>& b, int cout, int ks,
                      const ConvOpts& o_in, Act* out_override) {
  ConvOpts o = o_in;
  {
    int ot = -1, on = 0;   // one layer's choice forced by name (VP_PLAN_OVERRIDE: the in-frame tuner)
    if (ks == 3 && o.tile < 0 && plan_override(name, &ot, &on)) {
      o.tile = 100 + ot;
      if (on > 0) o.nsplit = on;
    }
  }
  const int cin = in->Creal, cin_pad = in->C;
  if (w.size() != (size_t)cout * cin * ks * ks) throw std::runtime_error("conv weight size mismatch: " + name);
  const int cstride = std::max(1, o.stride);
  if (cstride > 1 && (ks != 3 || in->H % cstride || in->W % cstride)) throw std::invalid_argument("strided conv: 3x3 on even maps only: " + name);
  const int M = (in->H / cstride) * (in->W / cstride);
  const int ncols = round_up(cout, 32);
  PackedConv pc;
  const int taps = ks * ks;
  // ---- 3x3: LDS-resident halo kernel (kernels_conv3x3.hip) unless overridden (tile >= 100 selects a halo tile)
  int halo = -1;
  if (ks == 3 && cstride == 1 && in->H >= 8 && in->W >= 16) {
    const char* env = dev_option("VP_CONV3X3");
    const bool force_v1 = (env && std::strcmp(env, "v1") == 0) || (o.tile >= 0 && o.tile < 100);
    if (o.tile >= 100) {
      halo = o.tile - 100;
    } else if (!force_v1) {
      auto cdiv = [](long long a, long long b) { return (a + b - 1) / b; };
      const long long t256 = cdiv(in->H, 16) * cdiv(in->W, 16), t128 = cdiv(in->H, 8) * cdiv(in->W, 16);
      if (ncols <= 32) {
        halo = (dev_option("VP_HEAD_TILE5") && t256 >= 400) ? 5 : 4;
      } else if (ncols % 128 != 0) {
        halo = (!split() && t256 * cdiv(ncols, 64) >= 400) ? 2 : 3;
      } else {
        halo = (!split() && t256 * (ncols / 128) >= 400) ? 0 : 1;
        // fewer 128-channel tiles than CUs: 64-channel tiles double the workgroup count, so the layer needs no split-K
        // (decode_layer_5: 46 us vs 57 us with a 3-way split) or half the split factor and half the fp32 partial
        // traffic (neck layers at 20x40 / 40x80: -1..-5 us each; profiles/r01_splitk_ablation.txt)
        if (halo == 1 && t128 * (ncols / 128) < 256) halo = 3;
      }
    }
    if (halo >= 0 && split() && (halo == 0 || halo == 2)) halo += 1;
    // parity mode, 128-channel tiles: the pipelined kernels of kernels_conv3x3_x3.hip -- halo tile 7 (8x16 patches, two
    // independent workgroups per CU; also the split-K shape of the small-map layers) or 6 (16x16 patches, one 8-wave workgroup)
    {
      const char* envx = dev_option("VP_X3_TILE");  // developer knob: 0 = halo kernel, 6 / 7 = force that shape
      auto cdiv = [](long long a, long long b) { return (a + b - 1) / b; };
      const long long wgs16 = cdiv(in->H, 16) * cdiv(in->W, 16) * (ncols / 128), wgs8 = cdiv(in->H, 8) * cdiv(in->W, 16) * (ncols / 128);
      // measured per layer (profiles/r02_layers_*): the 4-wave shape wins where the K loop is short (Cin <= 128: prologue and
      // epilogue weigh most and two independent workgroups per CU overlap them), the 8-wave shape elsewhere (half the weight
      // staging per MFMA)
      const int want = envx ? std::atoi(envx) : (cin_pad <= 128 ? 7 : 6);
      const bool plain = (o.act == ACT_GELU || o.act == ACT_NONE) && o.res_mode == RES_NONE && o.post_act == ACT_NONE;
      // round 4: the VP_FP16 engines take the same shapes on 64-channel chunks (X1: the planes are the chunk's halves) for their big layers --
      // the halo kernel's lone-wave schedule left them at 0.29-0.31 of peak.  VP_F16_BIG=0 (developer knob, A/B timing): the halo kernel.
      const char* envf = dev_option("VP_F16_BIG");
      if (!split() && !fp8_storage() && o.tile < 0 && !(envf && envf[0] == '0') && halo >= 0 && halo <= 3 && ncols % 128 == 0 && cin_pad % 64 == 0 && !o.logits_out &&
          !o.in2 && plain && wgs16 >= [this] { const char* e = dev_option("VP_F16_MIN_WGS"); return e ? std::atoi(e) : (plan_latency() ? 160 : 36); }()) {
        // measured per layer (profiles/r04_layers_sceneseg_fp16_big_ab.tsv): the 8-wave shape wins where ONE round of its workgroups covers the
        // map and the K loop is long (decode_layer_4: 79.6 -> 68.9 us, decode_layer_7: 45.6 -> 40.9); with two rounds (decode_layer_6: 75.6 ->
        // 77.2) or two chunks per workgroup (decode_layer_8: 85.8 -> 95.6 / 83.7 on the 4-wave shape) the halo kernel's two workgroups per CU
        // do as well or better.  VP_F16_BIG=6 / 7 forces a shape on every eligible layer.
        // Round 6, throughput plan: the in-frame search (tools/plan_search.py --precision fp16, profiles/r06_plan_search_fp16.tsv) moves the smaller layers
        // onto the pipelined shapes as well, as round 5 did for the parity mode -- fewer, fatter workgroups leave CUs to the other cameras: decode_layer_5
        // (100 tiles of 16x16 x 128 channels) and decode_layer_3 (60) on the 8-wave shape (+0.5 % / +1.1 % frames/s on the metric pair), decode_layer_1
        // (20x40 map, 36 tiles) on the 4-wave shape in four K slices (+0.6 %); together 1040 -> 1067 frames/s.  The latency plan keeps the 160-tile floor.
        if (envf) halo = std::atoi(envf) == 6 ? 6 : 7;
        else if (wgs16 <= 256 && cin_pad >= 256 && wgs16 >= 60) halo = 6;
        else if (wgs16 < 60 && cin_pad >= 256 && in->H * in->W <= 800 && o.nsplit <= 0) {
          halo = 7;
          o.nsplit = 4;
        }
      }
      if (split() && o.tile < 0 && want != 0 && (halo == 1 || halo == 3) && ncols % 128 == 0 && !o.logits_out && !o.in2) {
        // Smaller layers stay on the halo kernel's 64-channel tiles (two workgroups per CU, twice the workgroup count): measured
        // on MI355X, the 4-wave shape without split-K took 139 vs 100 us on decode_layer_5 (200 patches) and its split-K form
        // (kernel support kept, tile 107 + nsplit) 63 vs 47 / 79 vs 70 us on decode_layer_1 / 3 (profiles/r02_splitk_x3w4.txt)
        // Round 5: the floor is 100 tiles of 16x16 x 128 channels (it was 160): decode_layer_5 (512 -> 256 on 80x160: exactly 100) moves from the halo kernel's
        // 400 64-channel workgroups to 100 8-wave workgroups of the pipelined shape.  ALONE the layer is slower (101 -> 138 us: 100 CUs), per occupied CU
        // it runs at 2.2 TFLOP/s against 1.16 -- 100 CUs do not pull the package to its power limit -- and with the other head or the other cameras on
        // the remaining CUs the frame rate gains 1.5-3 % at an unchanged one-camera p50 (profiles/r05_dec5_x3w8_ab.txt).  Below 100 (the 40x80 / 20x40
        // maps: 60 / 36 workgroups) the same move LOSES 2 % / 11 %: their launches get 2-4x longer than the rest of the frame can cover.
        const char* envw = dev_option("VP_X3_MIN_WGS");  // developer knob: fewest 16x16 workgroups for the pipelined shapes
        if (plain && wgs16 >= (envw ? std::atoi(envw) : (plan_latency() ? 160 : 100))) halo = want == 6 ? 6 : 7;
        (void)wgs8;
      }
    }
    // parity mode, small maps of the neck (20x40 / 40x80 pixels, long K): one workgroup = a 32-channel weight slab x a K slice x ALL pixels
    // of a 20x40 region (kernels_conv3x3_map.hip, halo tile 11): the 8x16 tiles re-stream the weights once per pixel tile and are bound by
    // that traffic.  Any epilogue (it always ends in the finish kernel).  VP_MAP3X3=0 (developer knob, A/B timing): the tiled kernels.
    {
      const char* envm = dev_option("VP_MAP3X3");
      const bool on = !(envm && envm[0] == '0');
      // round 4: the context block's 10x20 maps take the kernel's 10x20-region geometry (four waves, two workgroups per CU; VP_CTX3=0: the
      // halo kernel's 8x16 tiles + split-K as before) -- context_layer_4 has 128 input channels, so the channel floor is per geometry
      const char* envc = dev_option("VP_CTX3");
      const int geom = conv3x3_map_geometry(in->H, in->W);
      const bool geom_on = geom == 1 ? cin_pad >= 256 : (geom == 2 && cin_pad >= 128 && !(envc && envc[0] == '0'));
      // round 4: the VP_FP16 engines take the kernel too (X1: steps of 32 channels, the halves in the two planes; VP_F16_MAP=0: the halo kernel's
      // 64-channel tiles + split-K as before)
      const char* envf = dev_option("VP_F16_MAP");
      // measured per layer (profiles/r04_layers_sceneseg_fp16_map_ab.tsv, halo kernel + split-K -> map kernel, us): context_layer_4..6 20.5 / 16.5 / 22.7 ->
      // 18.5 / 14.3 / 19.9, decode_layer_0 (1280 channels on 20x40) 40.4 -> 35.5; decode_layer_1..3 28.2 / 43.7 / 37.1 -> 29.9 / 45.5 / 37.8: the kernel
      // where it wins (the 10x20 geometry, >= 1024 input channels on a single region); VP_F16_MAP=1 everywhere it fits
      const bool f16_rule = (envf && envf[0] == '1') || geom == 2 || (cin_pad >= 1024 && M <= 800);
      const bool prec_on = split() || (!fp8_storage() && cin_pad % 32 == 0 && !(envf && envf[0] == '0') && f16_rule);
      int map_max_m = 3200;
      if (const char* e = dev_option("VP_MAP_MAX_M")) map_max_m = std::atoi(e);   // developer knob: largest map (pixels) the map kernels take
      if (o.tile == 111 || (on && prec_on && o.tile < 0 && halo >= 0 && ncols > 32 && !o.logits_out && !o.in2 && M <= map_max_m && geom_on &&
                            conv3x3_map_shape_ok(in->H, in->W, cin_pad, round_up(ncols, 32))))
        halo = 11;
      // round 5: 64-channel slabs (two M tiles per wave, kernels_conv3x3_map.hip "map2") where the geometry is the neck's and the output channels pad to 64
      // without waste: parity mode only.  VP_MAP2=0 (developer knob, A/B timing): tile 11 everywhere.
      int min_regions = 1;
      if (const char* e = dev_option("VP_MAP2_MIN_REGIONS")) min_regions = std::atoi(e);   // developer knob: 4 = the 40x80 maps only
      if (halo == 11 && o.tile < 0 && split() && geom == 1 && !dev_option_is("VP_MAP2", '0') && !(plan_latency() && !dev_option_is("VP_MAP2", '1')) &&
          round_up(ncols, 64) == round_up(ncols, 32) &&
          (in->H / 20) * (in->W / 40) >= min_regions && conv3x3_map2_shape_ok(in->H, in->W, cin_pad, round_up(ncols, 64)))
        halo = 12;
    }
    if (o.tile == 112) halo = 12;
    if (halo == 12 && (!split() || fp8_storage() || !conv3x3_map2_shape_ok(in->H, in->W, cin_pad, round_up(ncols, 64))))
      throw std::invalid_argument("halo tile 12 (map kernel, 64-channel slabs): parity mode, maps that tile into 20x40 regions, >= 32 input channels: " + name);
    // VP_WEIGHTS_FP8 as storage (AutoDrive engines and the operator entry): the halo kernel's 8x16-pixel tiles of 64 / 32 channels are the ones
    // instantiated with the byte-weight staging path (kernels_conv3x3.hip W8)
    if (fp8_storage() && o.tile < 0 && halo >= 0) halo = ncols <= 32 ? 4 : 3;
    if (halo == 11 && (fp8_storage() || (!split() && cin_pad % 32 != 0) || !conv3x3_map_shape_ok(in->H, in->W, cin_pad, round_up(ncols, 32))))
      throw std::invalid_argument("halo tile 11 (map kernel): maps that tile into 20x40 or 10x20 regions, >= 32 input channels (a multiple of 32 in the fp16 engines): " + name);
    if (halo >= 6 && halo <= 9 && !split() && cin_pad % 64 != 0) throw std::invalid_argument("halo tiles 6 - 9 in the fp16 engines: input channels a multiple of 64: " + name);
  }
  if (halo >= 0) {
    pc.tile = 100 + halo;
    pc.bk = 32;
    pc.CoutW = round_up(ncols, halo_tile_co(halo));
    if (halo == 12) {  // 64-channel slabs: a workgroup is twice the work of tile 11's, so the SAME K split gives half the workgroups (see the kernel's header)
      const int regions = (in->H / 20) * (in->W / 40), n_co = pc.CoutW / 64, KS = cin_pad / 16;
      int slots = 128;
      if (const char* e = dev_option("VP_MAP2_SLOTS")) slots = std::max(1, std::atoi(e));   // developer knob: workgroups a layer aims at
      int ns = o.nsplit > 0 ? o.nsplit : std::max(1, slots / (regions * n_co));
      const double slice_mb = (double)M * pc.CoutW * 4.0 / 1e6;
      double cap_mb = 27.0;   // 40x80 maps: four K slices (26.2 MB of slabs) -- measured 124 -> 103 / 91 -> 76 us per layer and +1.1 % frames/s against three (gpurun r5c04)
      if (const char* e = dev_option("VP_MAP2_CAP_MB")) cap_mb = std::atof(e);   // developer knob: ceiling of the fp32 slabs a layer may write
      while (o.nsplit <= 0 && ns > 2 && ns * slice_mb > cap_mb) --ns;
      pc.bk = 16;
      pc.nsplit = std::max(1, std::min(ns, std::max(1, KS / 2)));
    } else if (halo == 11) {  // map kernel: K slices until ~one round of workgroups (one per CU at 20x40 regions, two at 10x20), fp32 slabs <= 24 MB
      const int geom = conv3x3_map_geometry(in->H, in->W);
      const int regions = geom == 1 ? (in->H / 20) * (in->W / 40) : (in->H / 10) * (in->W / 20), n_co = pc.CoutW / 32, KS = cin_pad / (split() ? 16 : 32);
      const int slots = geom == 1 ? 256 : 512;   // 155 KB of LDS = one workgroup per CU (a 257th waits a whole round); 74 KB = two
      int ns = o.nsplit > 0 ? o.nsplit : std::max(1, slots / (regions * n_co));
      const double slice_mb = (double)M * pc.CoutW * 4.0 / 1e6;
      while (o.nsplit <= 0 && ns > 2 && ns * slice_mb > 26.0) --ns;
      if (const char* e = dev_option("VP_MAP_NSPLIT_PCT")) {   // developer knob (K-slice sweeps of the map kernel): percentage of the rule's factor
        if (o.nsplit <= 0 && ns > 1) ns = std::max(1, (int)(ns * std::atoi(e) / 100.0 + 0.5));
      }
      pc.bk = split() ? 16 : 32;
      pc.nsplit = std::max(1, std::min(ns, std::max(1, KS / 2)));
    } else {
    const int KC = cin_pad / 32;
    const long long blocks = (long long)((in->H + halo_tile_th(halo) - 1) / halo_tile_th(halo)) * ((in->W + 15) / 16) *
                             (pc.CoutW / halo_tile_co(halo));
    // split-K: aim at ONE machine-wide wave of workgroups (256); go towards two only while the K loop per slice stays
    // long (> 6 chunks = 54 tap steps), and keep the fp32 partials (written + re-read by the finish kernel at
    // ~4.5 TB/s) under ~24 MB.  Measured per layer in profiles/r01_splitk_ablation.txt.
    int ns = 1;
    if (o.nsplit > 0) {
      ns = o.nsplit;
    } else if (halo == 4 && o.logits_out && cout <= 4 && (cin_pad == 64 || cin_pad == 128) && o.res_mode == RES_NONE && o.act == ACT_NONE &&
               cstride == 1 && !(dev_option_is("VP_HEAD_CONV", '0'))) {
      ns = 1;  // a head's logits convolution goes to kernels_head.hip (persistent workgroups: needs no split on any map size)
    } else if (blocks < 256 && split()) {
      // parity mode (measured per layer with VP_NSPLIT_FORCE = 1..16, profiles/r02_splitk_sweep_fp16x3.txt): ONE full round of
      // workgroups at two per CU -- ns = floor(512 / blocks), at most one slice per input chunk.  A little past 512 (the fp16
      // rule gave 540 / 600 workgroups on the 20x40 / 40x80 layers) a second, nearly empty round costs 10-20 % of the layer.
      ns = (int)std::max<long long>(1, 512 / blocks);
      const double slice_mb = (double)M * pc.CoutW * 4.0 / 1e6;
      while (ns > 2 && ns * slice_mb > 24.0) --ns;
    } else if (blocks < 256) {
      ns = (int)((256 + blocks - 1) / blocks);
      while (KC / ns > 6 && blocks * ns < 512) ++ns;
      const double slice_mb = (double)M * pc.CoutW * 4.0 / 1e6;
      while (ns > 2 && ns * slice_mb > 24.0) --ns;
      ns = std::min(ns, std::max(1, KC / 2));
    }
    if (const char* e = dev_option("VP_NSPLIT_PCT")) {  // developer knob (split-K sweeps): percentage applied to the heuristic's factor
      if (o.nsplit <= 0 && ns > 1) ns = std::max(1, (int)(ns * std::atoi(e) / 100.0 + 0.5));
    }
    if (const char* e = dev_option("VP_NSPLIT_FORCE")) {  // developer knob: one factor for every layer the heuristic splits
      if (o.nsplit <= 0 && ns > 1) ns = std::max(1, std::atoi(e));
    }
    pc.nsplit = std::max(1, std::min(ns, KC));
    if (halo == 6) pc.nsplit = 1;  // one 8-wave workgroup per CU, >= 160 tiles: no split-K shape
    // 64-channel tiles of the parity mode: the pipelined kernel's 64-channel shape (halo tile 8: same tiles, same split factor as
    // halo tile 3, three workgroups per CU).  Measured on MI355X (SceneSeg, us, halo tile 3 -> tile 8): decode_layer_0..3 70.5 /
    // 48.1 / 82.6 / 60.2 -> 77.9 / 54.6 / 90.2 / 65.6, decode_layer_5 98.6 -> 106.5, decode_layer_9 (128 -> 64 channels on
    // 320x640) 108.0 -> 102.2; 389 -> 381 frames/s with it everywhere.  So: only the short-K big-map case (as for tile 7);
    // VP_X3_C64=1 wherever the epilogue fits (plain, or anything behind split-K), =0 nowhere.
    if ((halo == 3 || halo == 2) && !split() && !fp8_storage() && o.tile < 0 && cin_pad % 64 == 0 && ncols == 64 && !o.logits_out && !o.in2 && cstride == 1) {
      // fp16 engines, 64-channel outputs on the big maps (decode_layer_9: 128 -> 64 channels on 320x640): the pipelined 64-channel shape
      const char* e8 = dev_option("VP_F16_BIG");
      const bool plain8 = (o.act == ACT_GELU || o.act == ACT_NONE) && o.res_mode == RES_NONE && o.post_act == ACT_NONE;
      if (plain8 && e8 && e8[0] != '0' && M >= 65536) {   // measured level with the halo kernel (decode_layer_9: 52.2 -> 52.0 us): only when forced
        halo = 8;
        pc.tile = 108;
        pc.nsplit = 1;
      }
    }
    if (halo == 3 && split() && o.tile < 0 && cin_pad % 32 == 0 && !o.logits_out && !o.in2 && cstride == 1) {
      const char* e8 = dev_option("VP_X3_C64");
      const bool plain8 = (o.act == ACT_GELU || o.act == ACT_NONE) && o.res_mode == RES_NONE && o.post_act == ACT_NONE;
      const bool fits = plain8 || pc.nsplit > 1;
      const bool want8 = e8 ? e8[0] == '1' : (pc.nsplit == 1 && cin_pad <= 128 && M >= 65536);
      if (fits && want8) {
        halo = 8;
        pc.tile = 108;
      }
    }
    // round 5: 64-channel tiles of the parity mode on 16x16 pixels with the four waves side by side (halo tile 9: every wave 64 channels x 64 pixels, 8
    // fragment reads per 12 MFMAs instead of 6 per 6).  Built for the 64-channel-output layers without split-K on VERDICT round 4's reading that they are
    // fragment-read-bound; MEASURED SLOWER (profiles/r05_x3_t16_ab.txt): decode_layer_5 99.2 -> 135.1 us (200 workgroups of four waves = one wave per
    // SIMD on 200 CUs), decode_layer_9 103.9 -> 110.6 us (800 tiles on 512 slots: two rounds), frames/s level (412.8 against 412.7).  Kept as a shape
    // (tile 109, tests), selected only by VP_X3_T16=1.
    if ((halo == 3 || halo == 8) && split() && o.tile < 0 && pc.nsplit == 1 && cin_pad % 32 == 0 && !o.logits_out && !o.in2 && cstride == 1 &&
        (o.act == ACT_GELU || o.act == ACT_NONE) && o.res_mode == RES_NONE && o.post_act == ACT_NONE && dev_option_is("VP_X3_T16", '1') && M >= 12800) {
      halo = 9;
      pc.tile = 109;
    }
    }
  } else {
    ConvOpts og = o;
    if (fp8_storage() && og.tile < 0) og.tile = 2;   // the generic kernel's 64 x 64 tile carries the byte-weight staging path (kernels_conv.hip W8)
    choose_conv_cfg(M, ncols, cin_pad, ks, og, &pc);
  }
  // VP_WEIGHTS_FP8: real e4m3 storage where the layer's kernel stages its weights through registers (the generic GEMM kernel and the halo
  // kernel's tiles 0-5: every matrix layer of AutoDrive); the LDS-DMA / register-stationary kernels copy weight images verbatim and keep
  // de-quantised fp16 planes
  const bool w8 = fp8_storage() && ((halo < 0 && pc.tile == 2) || halo == 3 || halo == 4);
  // fp16 engines on the pipelined kernels (halo tiles 6 - 8 without the lo plane): 64-channel chunks, the chunk's halves in two plane arrays
  const bool k64 = halo >= 6 && halo <= 9 && !split();
  const bool k32map = halo == 11 && !split();   // map kernel in the fp16 engines: 32-channel steps, the step's halves in the two plane arrays
  std::vector<half_t> hi(w8 ? 0 : (size_t)taps * pc.CoutW * cin_pad / ((k64 || k32map) ? 2 : 1), (half_t)0.0f), lo(((split() || k64 || k32map) && !w8) ? hi.size() : 0, (half_t)0.0f);
  std::vector<uint8_t> codes(w8 ? (size_t)taps * pc.CoutW * cin_pad : 0, (uint8_t)0);
  RowScale rs = row_prescale(w.data(), cout, (size_t)cin * taps, pc.CoutW);
  if (w8)   // the rows' quantisation scales instead of the power-of-two prescale (e4m3 values need none: |q| in [2^-9, 448] are normal fp16 numbers)
    for (int co = 0; co < cout; ++co) {
      float amax = 0.0f;
      for (size_t i = 0; i < (size_t)cin * taps; ++i) amax = std::max(amax, std::fabs(w[(size_t)co * cin * taps + i]));
      rs.post[co] = fp8_row_scale(amax);
      rs.pre[co] = 1.0f / rs.post[co];
    }
  parallel_rows(cout, [&](int co_begin, int co_end) {
  for (int co = co_begin; co < co_end; ++co)
    for (int ci = 0; ci < cin; ++ci)
      for (int t = 0; t < taps; ++t) {
        const float v = w[((size_t)co * cin + ci) * taps + t];
        // generic kernel: [tap][CoutW][Cin] ; halo kernel: [cin/32][tap][CoutW][32] (contiguous per-tap tiles)
        // halo tiles 6 / 7 (kernels_conv3x3_x3.hip) copy weight tiles to LDS by LDS-DMA, a LINEAR copy: the tile is stored
        // in its LDS image order, i.e. with the 16-byte chunks of a row XOR-swizzled by (row >> 2) & 3
        const int ci_sw = (halo >= 6 && halo <= 9) ? ((((ci & 31) >> 3) ^ ((co >> 2) & 3)) << 3 | (ci & 7)) : (ci & 31);
        const size_t d = k32map ? conv3x3_map_pack_index_k32(co, ci, t, cin_pad)
                         : halo == 12 ? conv3x3_map2_pack_index(co, ci, t, cin_pad)
                         : halo == 11 ? conv3x3_map_pack_index(co, ci, t, cin_pad)
                         : k64     ? ((((size_t)(ci >> 6) * 9 + t) * pc.CoutW + co) * 32 + ci_sw)
                         : halo >= 0 ? ((((size_t)(ci >> 5) * 9 + t) * pc.CoutW + co) * 32 + ci_sw)
                                   : (((size_t)t * pc.CoutW + co) * cin_pad + ci);
        if (w8) {
          if (!(std::fabs(v) <= 65504.0f)) throw RangeError("weight " + std::to_string(v) + " is outside the fp16 range the matrix pipe carries (|w| <= 65504): re-scale the checkpoint");
          codes[d] = e4m3_encode(v / rs.post[co]);   // v / S is on the e4m3 grid up to the fold's float rounding: the nearest code IS the quantiser's
          continue;
        }
        half_t h, l;
        split_half(v, rs.pre[co], &h, &l);
        if (k64) {
          ((ci >> 5) & 1 ? lo : hi)[d] = h;
          continue;
        }
        if (k32map) {
          ((ci >> 4) & 1 ? lo : hi)[d] = h;
          continue;
        }
        hi[d] = h;
        if (split()) lo[d] = l;
      }
  });
  std::vector<float> bias(pc.CoutW, 0.0f);
  for (int co = 0; co < cout; ++co) bias[co] = b[co];
  if (w8) {
    pc.w8 = dupload(codes);
    wbytes_[0] += codes.size();
  } else {
    pc.w_hi = dupload(hi);
    pc.w_lo = (split() || k64 || k32map) ? dupload(lo) : nullptr;
    wbytes_[1] += 2 * (hi.size() + lo.size());
  }
  pc.bias = dupload(bias);
  pc.wscale = dupload(rs.post);
  Act* out = nullptr;
  int store = STORE_NHWC;
  if (o.logits_out) {
    store = STORE_NCHW_F32;
  } else {
    out = out_override ? out_override : new_act(name, cout, in->H / cstride, in->W / cstride);
  }
  push_conv_op(name, in, pc, ks, ncols, o, out, store, cout);
  return out;
}

// The small-map up-sampling GEMMs go to the LDS-DMA pipelined kernel (kernels_gemm_dma.hip; tile 6).  Measured per layer on
// MI355X (SceneSeg neck, us, old -> new): parity mode 42.1 / 41.0 / 42.2 -> 37.9 / 37.7 / 37.7 and +2.8 % frames/s with three
// frames in flight (fewer workgroups at a higher rate leave CUs to the other frames); fp16 30.4 / 23.4 / 25.4 -> 26.8 / 26.8 /
// 21.7: the fp16 engines take it from 2048 pixels up only.  VP_GEMM_DMA=1: wherever the shape fits, =0: never.
bool Engine::gemm_dma_wanted(int H, int W, int ncols, int cin_pad, int cin2_pad, int cstore) const {
  const int M = H * W;
  const char* e = dev_option("VP_GEMM_DMA");
  if (e && e[0] == '0') return false;
  if (!split() && M < 2048 && !(e && e[0] == '1')) return false;
  const char* rs = dev_option("VP_CONVT_RS");  // the register-stationary kernel's shapes are its own (and keep the plain weight layout)
  if (!(rs && rs[0] == '0') && convt_rs_shape_case(H, W, cin_pad, cin2_pad, ncols, cstore) != 0) return false;
  return gemm_dma_shape_ok(M, ncols, cin_pad, cin2_pad, cstore);
}

// split-K factor of that kernel: towards ~160 workgroups while a slice keeps >= 8 K steps (VP_GEMM_DMA_NSPLIT: developer knob)
int Engine::gemm_dma_nsplit(int M, int ncols, int kw) const {
  if (const char* e = dev_option("VP_GEMM_DMA_NSPLIT")) return std::max(1, std::atoi(e));
  const int tiles = ((M + 127) / 128) * (ncols / 256), steps = kw / 32;
  int ns = 1;
  while (tiles * ns < 128 && steps / (ns + 1) >= 8) ++ns;
  return ns;
}

// ConvTranspose2d(k2,s2): w [cin][cout][2][2] -> GEMM rows n = (dy*2+dx)*Cout_pad + co over input pixels.
Act* Engine::add_convT(const std::string& name, const Act* in, const std::vector<float>& w, const std::vector<float>& b, int cout,
                       const ConvOpts& o) {
  const int cin = in->Creal, cin_pad = in->C;
  if (w.size() != (size_t)cin * cout * 4) throw std::runtime_error("convT weight size mismatch: " + name);
  Act* out = new_act(name, cout, in->H * 2, in->W * 2);
  const int cpad = out->C;
  const int ncols = 4 * cpad;
  PackedConv pc;
  ConvOpts oo = o;
  oo.pixel_shuffle = true;
  // parity mode: 128-channel tiles with 32-channel K blocks (measured with VP_CONVT_TILE / VP_CONVT_BK over all tiles:
  // upsample_layer_4 80.9 -> 62.5 us, upsample_layer_1 + skip 54.7 -> 41.6 us, the others unchanged)
  if (split() && oo.tile < 0 && ncols % 128 == 0) {
    oo.tile = 0;
    if (oo.bk < 0) oo.bk = 32;
  }
  if (gemm_dma_wanted(in->H, in->W, ncols, cin_pad, 0, cpad) && o.tile < 0) {
    oo.tile = 6;
    oo.nsplit = gemm_dma_nsplit(in->H * in->W, ncols, cin_pad);
  }
  if (const char* e = dev_option("VP_CONVT_TILE")) oo.tile = std::atoi(e);  // developer knobs (tile / BK sweeps)
  if (const char* e = dev_option("VP_CONVT_BK")) oo.bk = std::atoi(e);
  choose_conv_cfg(in->H * in->W, ncols, cin_pad, 1, oo, &pc);
  if (pc.tile == 6 && !(gemm_dma_shape_ok(in->H * in->W, ncols, cin_pad, 0, cpad) && pc.CoutW == ncols))  // before the weights are packed in its layout
    throw std::invalid_argument("LDS-DMA GEMM kernel (tile 6): ConvTranspose k2 s2 (+ skip link) + bias, 4 * Cout and Cout multiples of 256, K >= 256, >= 128 pixels: " + name);
  std::vector<half_t> hi((size_t)pc.CoutW * cin_pad, (half_t)0.0f), lo(split() ? hi.size() : 0, (half_t)0.0f);
  std::vector<float> bias(pc.CoutW, 0.0f), post(pc.CoutW, 1.0f);
  parallel_rows(4 * cout, [&](int r_begin, int r_end) {
    for (int r = r_begin; r < r_end; ++r) {
      const int q = r / cout, co = r - q * cout;
      const int n = q * cpad + co;
      bias[n] = b[co];
      float amax = 0.0f;   // prescale per GEMM row n = (quadrant, output channel)
      for (int ci = 0; ci < cin; ++ci) amax = std::max(amax, std::fabs(w[((size_t)ci * cout + co) * 4 + q]));
      const int sexp = prescale_exp(amax);
      const float pre = std::ldexp(1.0f, sexp);
      post[n] = std::ldexp(1.0f, -sexp);
      for (int ci = 0; ci < cin; ++ci) {
        const float v = w[((size_t)ci * cout + co) * 4 + q];  // [ci][co][dy][dx], q = dy*2+dx
        half_t h, l;
        split_half(v, pre, &h, &l);
        const size_t d = pc.tile == 6 ? gemm_dma_pack_index(n, ci, cin_pad) : (size_t)n * cin_pad + ci;
        hi[d] = h;
        if (split()) lo[d] = l;
      }
    }
  });
  pc.w_hi = dupload(hi);
  pc.w_lo = split() ? dupload(lo) : nullptr;
  wbytes_[1] += 2 * (hi.size() + lo.size());
  pc.bias = dupload(bias);
  pc.wscale = dupload(post);
  push_conv_op(name, in, pc, 1, ncols, o, out, STORE_SHUFFLE2, cout);
  return out;
}

// d = ConvTranspose2d(k2,s2)(x) + Conv1x1(skip) in ONE GEMM (scene_neck.py:29-31 and the other up/skip pairs): the
// skip conv's input channels are appended to the K axis (kernels_conv.hip, K extension), biases are summed.  The
// intermediate up-sampled tensor is never written or re-read (it was the largest HBM stream of these layers) and the
// skip-link launch disappears.  Falls back to the two-op form when a workgroup's channel tile would straddle
// pixel-shuffle quadrants.
Act* Engine::add_convT_skip(const std::string& up_name, const std::string& skip_name, const Act* in, const Act* skip_in,
                            const std::vector<float>& wt, const std::vector<float>& bt, const std::vector<float>& ws,
                            const std::vector<float>& bs, int cout) {
  const int cin = in->Creal, cin_pad = in->C, cs = skip_in->Creal, cs_pad = skip_in->C;
  if (wt.size() != (size_t)cin * cout * 4) throw std::runtime_error("convT weight size mismatch: " + up_name);
  if (ws.size() != (size_t)cout * cs) throw std::runtime_error("skip conv weight size mismatch: " + skip_name);
  if (skip_in->H != in->H * 2 || skip_in->W != in->W * 2) throw std::runtime_error("skip tensor size mismatch: " + skip_name);
  const char* env = dev_option("VP_FUSE_SKIP");
  const int cpad = round_up(cout, 32);
  const int ncols = 4 * cpad;
  ConvOpts o;
  o.in2 = skip_in;
  if ((cin_pad | cs_pad) % 64 != 0) o.bk = 32;  // K steps must not straddle the two tensors
  if (split() && ncols % 128 == 0) {  // parity mode: see add_convT
    o.tile = 0;
    o.bk = 32;
  }
  if (gemm_dma_wanted(in->H, in->W, ncols, cin_pad, cs_pad, cpad)) {
    o.tile = 6;
    o.nsplit = gemm_dma_nsplit(in->H * in->W, ncols, cin_pad + cs_pad);
  }
  if (const char* e = dev_option("VP_CONVT_TILE")) o.tile = std::atoi(e);
  PackedConv pc;
  choose_conv_cfg(in->H * in->W, ncols, cin_pad + cs_pad, 1, o, &pc);
  if (pc.tile == 6 && !(gemm_dma_shape_ok(in->H * in->W, ncols, cin_pad, cs_pad, cpad) && pc.CoutW == ncols))
    throw std::invalid_argument("LDS-DMA GEMM kernel (tile 6): shape not covered: " + up_name);
  const bool fusable = (cpad % conv_tile_co(pc.tile) == 0 && !(env && env[0] == '0')) || pc.tile == 6;
  if (!fusable) {
    Act* u = add_convT(up_name, in, wt, bt, cout, ConvOpts{});
    ConvOpts so;
    so.res_mode = RES_ADD;
    so.res = u;
    add_conv(skip_name, skip_in, ws, bs, cout, 1, so, u);
    return u;
  }
  Act* out = new_act(up_name, cout, in->H * 2, in->W * 2);
  const int kw = cin_pad + cs_pad;
  std::vector<half_t> hi((size_t)pc.CoutW * kw, (half_t)0.0f), lo(split() ? hi.size() : 0, (half_t)0.0f);
  std::vector<float> bias(pc.CoutW, 0.0f), post(pc.CoutW, 1.0f);
  parallel_rows(4 * cout, [&](int r_begin, int r_end) {
    for (int r = r_begin; r < r_end; ++r) {
      const int q = r / cout, co = r - q * cout;
      const int n = q * cpad + co;
      bias[n] = bt[co] + bs[co];
      float amax = 0.0f;   // prescale per GEMM row over BOTH weight sets of its K axis
      for (int k = 0; k < cin + cs; ++k) amax = std::max(amax, std::fabs(k < cin ? wt[((size_t)k * cout + co) * 4 + q] : ws[(size_t)co * cs + (k - cin)]));
      const int sexp = prescale_exp(amax);
      const float pre = std::ldexp(1.0f, sexp);
      post[n] = std::ldexp(1.0f, -sexp);
      for (int k = 0; k < cin + cs; ++k) {
        const float v = k < cin ? wt[((size_t)k * cout + co) * 4 + q] : ws[(size_t)co * cs + (k - cin)];
        const size_t col = k < cin ? k : cin_pad + (k - cin);
        half_t h, l;
        split_half(v, pre, &h, &l);
        const size_t d = pc.tile == 6 ? gemm_dma_pack_index(n, (int)col, kw) : (size_t)n * kw + col;
        hi[d] = h;
        if (split()) lo[d] = l;
      }
    }
  });
  pc.w_hi = dupload(hi);
  pc.w_lo = split() ? dupload(lo) : nullptr;
  wbytes_[1] += 2 * (hi.size() + lo.size());
  pc.bias = dupload(bias);
  pc.wscale = dupload(post);
  push_conv_op(up_name + "+" + skip_name.substr(skip_name.rfind('.') == std::string::npos ? 0 : skip_name.rfind('.') + 1), in, pc, 1, ncols, o,
               out, STORE_SHUFFLE2, cout);
  return out;
}

}  // namespace vp
