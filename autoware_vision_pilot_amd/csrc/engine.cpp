#include "engine.hpp"

#include "conv_epilogue.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>

namespace vp {

static inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

// ================================================================================================ WeightBlob
void WeightBlob::parse(const void* blob, size_t bytes) {
  const uint8_t* p = static_cast<const uint8_t*>(blob);
  const uint8_t* end = p + bytes;
  auto need = [&](size_t n) {
    if ((size_t)(end - p) < n) throw std::runtime_error("weight blob truncated");
  };
  need(8);
  if (std::memcmp(p, "VPW1", 4) != 0) throw std::runtime_error("weight blob: bad magic (expected VPW1)");
  uint32_t count;
  std::memcpy(&count, p + 4, 4);
  p += 8;
  for (uint32_t i = 0; i < count; ++i) {
    need(2);
    uint16_t nl;
    std::memcpy(&nl, p, 2);
    p += 2;
    need(nl + 1);
    std::string name(reinterpret_cast<const char*>(p), nl);
    p += nl;
    const int nd = *p++;
    need(4 * (size_t)nd);
    HostTensor t;
    size_t n = 1;
    for (int d = 0; d < nd; ++d) {
      uint32_t v;
      std::memcpy(&v, p, 4);
      p += 4;
      t.shape.push_back((int)v);
      n *= v;
    }
    need(4 * n);
    t.data.resize(n);
    std::memcpy(t.data.data(), p, 4 * n);
    p += 4 * n;
    t_.emplace(std::move(name), std::move(t));
  }
}
const HostTensor& WeightBlob::get(const std::string& key) const {
  auto it = t_.find(key);
  if (it == t_.end()) throw std::runtime_error("weight blob: missing tensor '" + key + "'");
  return it->second;
}

// ======================================================================================== folding + packing
namespace {

constexpr float kBnEps = 1e-5f;  // torchvision efficientnet_b0 BatchNorm2d default

struct Folded {
  std::vector<float> w, b;
  int cout = 0, cin = 0, k = 0;  // cin = per-group input channels
};

// conv(no bias) + BatchNorm(eval) -> conv with bias:  w' = w * g/sqrt(v+eps),  b' = beta - mean * g/sqrt(v+eps).
// A blob converted from an ONNX file exported with do_constant_folding=True (Models/exports/convert_pytorch_to_onnx.py:
// 144-154) carries that product already: `<conv>.weight` + `<conv>.bias` and NO norm tensors (weights.py export_onnx) --
// taken as is.  Anything in between (norm present but incomplete, or neither) fails loudly in blob.get().
Folded fold_conv(const WeightBlob& blob, const std::string& conv, const std::string& norm, float eps) {
  const HostTensor& w = blob.get(conv + ".weight");
  if (w.shape.size() != 4) throw std::runtime_error("conv weight rank != 4: " + conv);
  Folded f;
  f.cout = w.shape[0];
  f.cin = w.shape[1];
  f.k = w.shape[2];
  const size_t per = (size_t)f.cin * f.k * f.k;
  if (!blob.has(norm + ".weight") && blob.has(conv + ".bias")) {  // exporter-folded
    const HostTensor& b = blob.get(conv + ".bias");
    if ((int)b.data.size() != f.cout) throw std::runtime_error("folded conv bias length != Cout: " + conv);
    f.w = w.data;
    f.b = b.data;
    return f;
  }
  const HostTensor& g = blob.get(norm + ".weight");
  const HostTensor& beta = blob.get(norm + ".bias");
  const HostTensor& mean = blob.get(norm + ".running_mean");
  const HostTensor& var = blob.get(norm + ".running_var");
  f.w.resize(w.data.size());
  f.b.resize(f.cout);
  for (int co = 0; co < f.cout; ++co) {
    const float s = g.data[co] / std::sqrt(var.data[co] + eps);
    for (size_t i = 0; i < per; ++i) f.w[co * per + i] = w.data[co * per + i] * s;
    f.b[co] = beta.data[co] - mean.data[co] * s;
  }
  return f;
}

// torchvision Conv2dNormActivation: Sequential(0: conv, 1: BatchNorm2d eps 1e-5)
Folded fold_conv_bn(const WeightBlob& blob, const std::string& p) { return fold_conv(blob, p + ".0", p + ".1", kBnEps); }

// common_layers.py:5-14 Conv: `conv` (no bias) + `norm` (BatchNorm2d eps 1e-3)
Folded fold_conv_norm(const WeightBlob& blob, const std::string& p) { return fold_conv(blob, p + ".conv", p + ".norm", 1e-3f); }

void split_half(float v, half_t* hi, half_t* lo) {
  // Both precision modes carry weights on fp16 planes: a folded weight beyond the fp16 range (or non-finite) would load as inf and
  // every result would silently be garbage.  (Small weights are safe: below 2^-14 the hi plane is subnormal and the lo plane picks up less,
  // a loss of relative precision on values that contribute nothing at the 1e-3 bar.)
  if (!(std::fabs(v) <= 65504.0f)) throw RangeError("weight " + std::to_string(v) + " is outside the fp16 range the matrix pipe carries (|w| <= 65504): re-scale the checkpoint");
  const half_t h = (half_t)v;
  *hi = h;
  *lo = (half_t)(v - (float)h);
}

}  // namespace

// ==================================================================================================== Engine
Engine::Engine(int kind, const WeightBlob* blob, int precision, int gpu_id, Engine* base, int frames, int frame_index)
    : kind_(kind), precision_(precision), gpu_(gpu_id), frames_(frames), frame_index_(frame_index), base_(base) {
  if (frames < 1 || frames > 16) throw std::invalid_argument("frames must be 1..16");
  if (frames > 1 && (base || kind < 0 || kind > 3)) throw std::invalid_argument("a batched encoder is a base engine of a scene network kind");
  if (base && (frame_index < 0 || frame_index >= base->frames_)) throw std::invalid_argument("frame_index out of the base engine's range");
  if (!base && frame_index != 0) throw std::invalid_argument("frame_index needs a batched base engine");
  if ((precision & 15) > 1 || (precision & ~17) != 0) throw std::invalid_argument("precision must be VP_FP16 or VP_FP16X3 (optionally | VP_WEIGHTS_FP8)");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
    throw std::runtime_error("libvp_hip: no HIP device visible (this library has no CPU fallback)");
  if (gpu_id < 0 || gpu_id >= ndev) throw std::invalid_argument("gpu_id out of range");
  VP_HIP_CHECK(hipSetDevice(gpu_id));
  try {
    construct(kind, blob, precision, gpu_id, base);
  } catch (...) {  // ~Engine never runs for a throwing constructor: free the stream, events and every dalloc() made so far
    release();
    throw;
  }
}

void Engine::construct(int kind, const WeightBlob* blob, int precision, int gpu_id, Engine* base) {
  {  // process-unique plan epochs: a combined graph keyed on (engine address, epoch) can never match a later engine at the same address
    static std::atomic<unsigned long long> next_epoch{1};
    plan_epoch_ = next_epoch.fetch_add(1ull << 32);
  }
  if (base) {
    if (kind < 0 || base->kind_ < 0) throw std::invalid_argument("shared engines need model kinds on both sides");
    if (base->base_) throw std::invalid_argument("the base of a shared engine must own its whole network");
    if (base->precision_ != precision || base->gpu_ != gpu_id)
      throw std::invalid_argument("shared engine: precision and gpu_id must equal the base engine's");
    stream_ = base->stream_;  // same stream: this engine's launches are ordered after the base engine's
  } else {
    VP_HIP_CHECK(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
  }
  VP_HIP_CHECK(hipEventCreate(&ev0_));
  VP_HIP_CHECK(hipEventCreate(&ev1_));
  if (kind >= 0) {
    if (!blob) throw std::invalid_argument("weights required");
    if (fp8_weights()) {
      WeightBlob q = *blob;
      q.quantize_fp8_e4m3();
      build_model(q);
    } else {
      build_model(*blob);
    }
    finish_plan();
  }
}

Engine::~Engine() { release(); }

void Engine::release() {
  hipSetDevice(gpu_);
  if (stream_) hipStreamSynchronize(stream_);
  if (graph_exec_) hipGraphExecDestroy(graph_exec_);
  if (graph_) hipGraphDestroy(graph_);
  graph_exec_ = nullptr;
  graph_ = nullptr;
  if (multi_exec_) hipGraphExecDestroy(multi_exec_);
  if (multi_graph_) hipGraphDestroy(multi_graph_);
  multi_exec_ = nullptr;
  multi_graph_ = nullptr;
  for (hipStream_t sd : side_streams_) hipStreamDestroy(sd);
  side_streams_.clear();
  for (hipEvent_t ev : side_events_) hipEventDestroy(ev);
  side_events_.clear();
  for (void* p : allocs_) hipFree(p);
  allocs_.clear();
  d_zero_ = nullptr;
  if (h_logits_) hipHostFree(h_logits_);
  if (h_mask_) hipHostFree(h_mask_);
  if (h_frame_) hipHostFree(h_frame_);
  if (h_status_) hipHostFree(h_status_);
  h_status_ = nullptr;
  d_status_ = nullptr;
  h_logits_ = nullptr;
  h_mask_ = nullptr;
  h_frame_ = nullptr;
  if (ev0_) hipEventDestroy(ev0_);
  if (ev1_) hipEventDestroy(ev1_);
  ev0_ = ev1_ = nullptr;
  for (hipEvent_t& ev : h_frame_ev_) {
    if (ev) hipEventDestroy(ev);
    ev = nullptr;
  }
  if (stream_ && !base_) hipStreamDestroy(stream_);
  stream_ = nullptr;
}

unsigned long long WeightBlob::group_hash(const std::string& prefix) const {
  unsigned long long h = 1469598103934665603ull;
  auto mix = [&h](const void* d, size_t n) {
    const unsigned char* b = static_cast<const unsigned char*>(d);
    for (size_t i = 0; i < n; ++i) {
      h ^= b[i];
      h *= 1099511628211ull;
    }
  };
  size_t n = 0;
  for (auto it = t_.lower_bound(prefix); it != t_.end() && it->first.compare(0, prefix.size(), prefix) == 0; ++it, ++n) {
    const std::string suffix = it->first.substr(prefix.size());
    mix(suffix.data(), suffix.size());
    mix(it->second.shape.data(), it->second.shape.size() * sizeof(int));
    mix(it->second.data.data(), it->second.data.size() * sizeof(float));
  }
  mix(&n, sizeof(n));
  return h;
}

void WeightBlob::quantize_fp8_e4m3() {
  for (auto& kv : t_) {
    HostTensor& t = kv.second;
    const std::string& k = kv.first;
    if (t.shape.size() < 2 || k.size() < 7 || k.compare(k.size() - 7, 7, ".weight") != 0) continue;
    const size_t rows = (size_t)t.shape[0], per = t.data.size() / rows;
    for (size_t r = 0; r < rows; ++r) {
      float* v = t.data.data() + r * per;
      double amax = 0.0;
      for (size_t i = 0; i < per; ++i) amax = std::max(amax, std::fabs((double)v[i]));
      const double scale = std::max(amax, 1e-30) / 448.0;
      for (size_t i = 0; i < per; ++i) {
        const double x = (double)v[i] / scale, mag = std::fabs(x);
        double e = std::floor(std::log2(std::max(mag, std::ldexp(1.0, -9))));
        e = std::min(std::max(e, -6.0), 8.0);
        const double step = std::ldexp(1.0, (int)e - 3);
        const double q = std::min(std::nearbyint(mag / step) * step, 448.0) * (x > 0 ? 1.0 : (x < 0 ? -1.0 : 0.0));
        v[i] = (float)(q * scale);
      }
    }
  }
}

void* Engine::dalloc(size_t bytes, bool zero) {
  void* p = nullptr;
  VP_HIP_CHECK(hipMalloc(&p, std::max<size_t>(bytes, 256)));
  allocs_.push_back(p);
  if (zero) VP_HIP_CHECK(hipMemset(p, 0, std::max<size_t>(bytes, 256)));
  return p;
}
const void* Engine::zero_page() {
  if (!d_zero_) d_zero_ = dalloc(256, true);
  return d_zero_;
}

template <class T>
T* Engine::dupload(const std::vector<T>& v) {
  T* d = static_cast<T*>(dalloc(v.size() * sizeof(T), false));
  VP_HIP_CHECK(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  return d;
}

Act* Engine::frame_view(const Act* a, int f) {
  if (a->frames <= 1) return const_cast<Act*>(a);
  auto v = std::make_unique<Act>(*a);
  v->frames = 1;
  v->H = a->H / a->frames;
  v->name = a->name + "#" + std::to_string(f);
  const size_t off = (size_t)f * v->H * v->W * v->C;
  v->hi = a->hi + off;
  if (a->lo) v->lo = a->lo + off;
  acts_.push_back(std::move(v));
  return acts_.back().get();
}

Act* Engine::new_act(const std::string& name, int creal, int h, int w) {
  auto a = std::make_unique<Act>();
  a->name = name;
  a->Creal = creal;
  a->C = round_up(creal, 32);
  a->H = h;
  a->W = w;
  a->hi = static_cast<half_t*>(dalloc(a->elems() * sizeof(half_t)));
  if (split()) a->lo = static_cast<half_t*>(dalloc(a->elems() * sizeof(half_t)));
  acts_.push_back(std::move(a));
  return acts_.back().get();
}

void Engine::upload_act(Act* a, const float* chw) {
  float* d = static_cast<float*>(dalloc((size_t)a->Creal * a->H * a->W * sizeof(float), false));
  VP_HIP_CHECK(hipMemcpy(d, chw, (size_t)a->Creal * a->H * a->W * sizeof(float), hipMemcpyHostToDevice));
  VP_HIP_CHECK(launch_nchw_to_act(d, a->Creal, a->view(), stream_));
  VP_HIP_CHECK(hipStreamSynchronize(stream_));
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
    const char* e = std::getenv("VP_PROJ_SPLIT");
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
                           kind_ != 4 && !(std::getenv("VP_FUSE_DECODE") && std::getenv("VP_FUSE_DECODE")[0] == '0');
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
  p.partial = pc.nsplit > 1 ? static_cast<float*>(dalloc((size_t)pc.nsplit * M * pc.CoutW * sizeof(float), false)) : nullptr;
  // split-K layers finish in the workgroup that arrives last on a tile (conv_epilogue.hpp splitk_arrive_and_finish): one zeroed arrival
  // counter per output tile (no tile is smaller than 64 pixels x 32 channels).  VP_SPLITK_FOLD=0 (developer knob, A/B timing): the
  // separate splitk_finish_kernel launch.
  static const char* env_fold = std::getenv("VP_SPLITK_FOLD");
  if (pc.nsplit > 1 && !(env_fold && env_fold[0] == '0'))
    p.tile_count = static_cast<unsigned*>(dalloc((size_t)((M + 63) / 64) * ((pc.CoutW + 31) / 32) * sizeof(unsigned), true));
  const int tile = pc.tile, bk = pc.bk;
  const bool sp = split();
  Op op;
  op.name = name;
  op.flops = 2.0 * M * (double)cout_real * (store_mode == STORE_SHUFFLE2 ? 4 : 1) * in->Creal * ks * ks;
  const double esz = sp ? 4.0 : 2.0;
  op.bytes = esz * ((double)M * in->Creal + (double)M * (store_mode == STORE_SHUFFLE2 ? 4 : 1) * cout_real +
                    (double)cout_real * (store_mode == STORE_SHUFFLE2 ? 4 : 1) * in->Creal * ks * ks);
  if (o.in2) {  // fused skip-link: extra K columns read at every OUTPUT pixel
    op.flops += 2.0 * M * 4.0 * cout_real * o.in2->Creal;
    op.bytes += esz * ((double)M * 4.0 * o.in2->Creal + (double)cout_real * o.in2->Creal);
  }
  if (tile >= 100) {
    int ht = tile - 100;
    // ---- optional per-layer tile autotune (VP_AUTOTUNE=1 enables; measured +-1 % on the frames-in-flight bench, so
    // the static heuristic is the default): time the tile shapes that share this weight packing and keep the fastest.  The K order per output is identical for every tile, so the choice never changes a result bit.
    static const char* at_env = std::getenv("VP_AUTOTUNE");
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
    if (ht >= 6 && ht <= 8) {
      if (!conv3x3_x3_supported(p, ht)) throw std::invalid_argument("halo tiles 6 - 8 (pipelined fp16x3 kernels): conv + bias + {GELU, none}, NHWC, 128- (64-) channel tiles; any epilogue with split-K (7 / 8): " + name);
      op.kernel = std::string(ht == 6 ? "conv3x3_x3w8<co128,px256>" : (ht == 7 ? "conv3x3_x3w4<co128,px128>" : "conv3x3_x3w4<co64,px128>")) + (pc.nsplit > 1 ? "+splitk" : "");
      op.run = [p, ht](hipStream_t st) { return launch_conv3x3_x3(p, ht, st); };
      ops_.push_back(std::move(op));
      return;
    }
    // the heads' logits convolution: weights stationary in registers, 16x16x32 MFMA, LDS-DMA halo (kernels_head.hip); same weight
    // packing as halo tile 4.  VP_HEAD_CONV=0 keeps the halo kernel.
    if (ht == 4 && o.tile < 0 && head_conv_supported(p) && !(std::getenv("VP_HEAD_CONV") && std::getenv("VP_HEAD_CONV")[0] == '0')) {
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
    const bool regepi = !sp && p.act == ACT_GELU_F16 && p.res_mode == RES_NONE && p.store_mode == STORE_NHWC && p.nsplit == 1;
    op.kernel = "conv3x3_halo<co" + std::to_string(halo_tile_co(ht)) + ",px" + std::to_string(halo_tile_px(ht)) + (sp ? ",x3" : ",x1") +
                (regepi ? ",regepi>" : ">") + (pc.nsplit > 1 ? "+splitk" : "");
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
  } else if (tile == 5 || (!(std::getenv("VP_CONVT_RS") && std::getenv("VP_CONVT_RS")[0] == '0') && convt_rs_supported(p, sp))) {
    if (!convt_rs_supported(p, sp))
      throw std::invalid_argument("register-stationary ConvTranspose kernel (tile 5): k2 s2 + bias, K = 128 or 256 + 32 (skip link), map width a multiple of 32, >= 2048 pixels: " + name);
    op.kernel = "convt_rs<k" + std::to_string(p.Cin + p.Cin2) + (sp ? ",x3>" : ",x1>");
    op.run = [p](hipStream_t st) { return launch_convt_rs(p, st); };
  } else {
    const int epi = (ks == 1 && !sp) ? regepi_case(p, conv_tile_co(tile), sp) : 0;  // same rule as launch_cfg (kernels_conv.hip)
    op.kernel = "conv_gemm<bk" + std::to_string(bk) + ",co" + std::to_string(conv_tile_co(tile)) + ",px" +
                std::to_string(conv_tile_px(tile)) + (sp ? ",x3" : ",x1") + (epi ? ",regepi" + std::to_string(epi) + ">" : ">") +
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
                      const ConvOpts& o, Act* out_override) {
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
    static const char* env = std::getenv("VP_CONV3X3");
    const bool force_v1 = (env && std::strcmp(env, "v1") == 0) || (o.tile >= 0 && o.tile < 100);
    if (o.tile >= 100) {
      halo = o.tile - 100;
    } else if (!force_v1) {
      auto cdiv = [](long long a, long long b) { return (a + b - 1) / b; };
      const long long t256 = cdiv(in->H, 16) * cdiv(in->W, 16), t128 = cdiv(in->H, 8) * cdiv(in->W, 16);
      if (ncols <= 32) {
        halo = (std::getenv("VP_HEAD_TILE5") && t256 >= 400) ? 5 : 4;
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
      static const char* envx = std::getenv("VP_X3_TILE");  // developer knob: 0 = halo kernel, 6 / 7 = force that shape
      auto cdiv = [](long long a, long long b) { return (a + b - 1) / b; };
      const long long wgs16 = cdiv(in->H, 16) * cdiv(in->W, 16) * (ncols / 128), wgs8 = cdiv(in->H, 8) * cdiv(in->W, 16) * (ncols / 128);
      // measured per layer (profiles/r02_layers_*): the 4-wave shape wins where the K loop is short (Cin <= 128: prologue and
      // epilogue weigh most and two independent workgroups per CU overlap them), the 8-wave shape elsewhere (half the weight
      // staging per MFMA)
      const int want = envx ? std::atoi(envx) : (cin_pad <= 128 ? 7 : 6);
      const bool plain = (o.act == ACT_GELU || o.act == ACT_NONE) && o.res_mode == RES_NONE && o.post_act == ACT_NONE;
      if (split() && o.tile < 0 && want != 0 && (halo == 1 || halo == 3) && ncols % 128 == 0 && !o.logits_out && !o.in2) {
        // Smaller layers stay on the halo kernel's 64-channel tiles (two workgroups per CU, twice the workgroup count): measured
        // on MI355X, the 4-wave shape without split-K took 139 vs 100 us on decode_layer_5 (200 patches) and its split-K form
        // (kernel support kept, tile 107 + nsplit) 63 vs 47 / 79 vs 70 us on decode_layer_1 / 3 (profiles/r02_splitk_x3w4.txt)
        static const char* envw = std::getenv("VP_X3_MIN_WGS");  // developer knob: fewest 16x16 workgroups for the pipelined shapes
        if (plain && wgs16 >= (envw ? std::atoi(envw) : 160)) halo = want == 6 ? 6 : 7;
        (void)wgs8;
      }
    }
    if (halo >= 6 && halo <= 8 && !split()) throw std::invalid_argument("halo tiles 6 - 8 are fp16x3 kernels: " + name);
  }
  if (halo >= 0) {
    pc.tile = 100 + halo;
    pc.bk = 32;
    pc.CoutW = round_up(ncols, halo_tile_co(halo));
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
               cstride == 1 && !(std::getenv("VP_HEAD_CONV") && std::getenv("VP_HEAD_CONV")[0] == '0')) {
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
    if (const char* e = std::getenv("VP_NSPLIT_PCT")) {  // developer knob (split-K sweeps): percentage applied to the heuristic's factor
      if (o.nsplit <= 0 && ns > 1) ns = std::max(1, (int)(ns * std::atoi(e) / 100.0 + 0.5));
    }
    if (const char* e = std::getenv("VP_NSPLIT_FORCE")) {  // developer knob: one factor for every layer the heuristic splits
      if (o.nsplit <= 0 && ns > 1) ns = std::max(1, std::atoi(e));
    }
    pc.nsplit = std::max(1, std::min(ns, KC));
    if (halo == 6) pc.nsplit = 1;  // one 8-wave workgroup per CU, >= 160 tiles: no split-K shape
    // 64-channel tiles of the parity mode: the pipelined kernel's 64-channel shape (halo tile 8: same tiles, same split factor as
    // halo tile 3, three workgroups per CU).  Measured on MI355X (SceneSeg, us, halo tile 3 -> tile 8): decode_layer_0..3 70.5 /
    // 48.1 / 82.6 / 60.2 -> 77.9 / 54.6 / 90.2 / 65.6, decode_layer_5 98.6 -> 106.5, decode_layer_9 (128 -> 64 channels on
    // 320x640) 108.0 -> 102.2; 389 -> 381 frames/s with it everywhere.  So: only the short-K big-map case (as for tile 7);
    // VP_X3_C64=1 wherever the epilogue fits (plain, or anything behind split-K), =0 nowhere.
    if (halo == 3 && split() && o.tile < 0 && cin_pad % 32 == 0 && !o.logits_out && !o.in2 && cstride == 1) {
      const char* e8 = std::getenv("VP_X3_C64");
      const bool plain8 = (o.act == ACT_GELU || o.act == ACT_NONE) && o.res_mode == RES_NONE && o.post_act == ACT_NONE;
      const bool fits = plain8 || pc.nsplit > 1;
      const bool want8 = e8 ? e8[0] == '1' : (pc.nsplit == 1 && cin_pad <= 128 && M >= 65536);
      if (fits && want8) {
        halo = 8;
        pc.tile = 108;
      }
    }
  } else {
    choose_conv_cfg(M, ncols, cin_pad, ks, o, &pc);
  }
  std::vector<half_t> hi((size_t)taps * pc.CoutW * cin_pad, (half_t)0.0f), lo(split() ? hi.size() : 0, (half_t)0.0f);
  for (int co = 0; co < cout; ++co)
    for (int ci = 0; ci < cin; ++ci)
      for (int t = 0; t < taps; ++t) {
        const float v = w[((size_t)co * cin + ci) * taps + t];
        // generic kernel: [tap][CoutW][Cin] ; halo kernel: [cin/32][tap][CoutW][32] (contiguous per-tap tiles)
        // halo tiles 6 / 7 (kernels_conv3x3_x3.hip) copy weight tiles to LDS by LDS-DMA, a LINEAR copy: the tile is stored
        // in its LDS image order, i.e. with the 16-byte chunks of a row XOR-swizzled by (row >> 2) & 3
        const int ci_sw = (halo >= 6 && halo <= 8) ? ((((ci & 31) >> 3) ^ ((co >> 2) & 3)) << 3 | (ci & 7)) : (ci & 31);
        const size_t d = halo >= 0 ? ((((size_t)(ci >> 5) * 9 + t) * pc.CoutW + co) * 32 + ci_sw)
                                   : (((size_t)t * pc.CoutW + co) * cin_pad + ci);
        half_t h, l;
        split_half(v, &h, &l);
        hi[d] = h;
        if (split()) lo[d] = l;
      }
  std::vector<float> bias(pc.CoutW, 0.0f);
  for (int co = 0; co < cout; ++co) bias[co] = b[co];
  pc.w_hi = dupload(hi);
  pc.w_lo = split() ? dupload(lo) : nullptr;
  pc.bias = dupload(bias);
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
  const char* e = std::getenv("VP_GEMM_DMA");
  if (e && e[0] == '0') return false;
  if (!split() && M < 2048 && !(e && e[0] == '1')) return false;
  const char* rs = std::getenv("VP_CONVT_RS");  // the register-stationary kernel's shapes are its own (and keep the plain weight layout)
  if (!(rs && rs[0] == '0') && convt_rs_shape_case(H, W, cin_pad, cin2_pad, ncols, cstore) != 0) return false;
  return gemm_dma_shape_ok(M, ncols, cin_pad, cin2_pad, cstore);
}

// split-K factor of that kernel: towards ~160 workgroups while a slice keeps >= 8 K steps (VP_GEMM_DMA_NSPLIT: developer knob)
int Engine::gemm_dma_nsplit(int M, int ncols, int kw) const {
  if (const char* e = std::getenv("VP_GEMM_DMA_NSPLIT")) return std::max(1, std::atoi(e));
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
  if (const char* e = std::getenv("VP_CONVT_TILE")) oo.tile = std::atoi(e);  // developer knobs (tile / BK sweeps)
  if (const char* e = std::getenv("VP_CONVT_BK")) oo.bk = std::atoi(e);
  choose_conv_cfg(in->H * in->W, ncols, cin_pad, 1, oo, &pc);
  if (pc.tile == 6 && !(gemm_dma_shape_ok(in->H * in->W, ncols, cin_pad, 0, cpad) && pc.CoutW == ncols))  // before the weights are packed in its layout
    throw std::invalid_argument("LDS-DMA GEMM kernel (tile 6): ConvTranspose k2 s2 (+ skip link) + bias, 4 * Cout and Cout multiples of 256, K >= 256, >= 128 pixels: " + name);
  std::vector<half_t> hi((size_t)pc.CoutW * cin_pad, (half_t)0.0f), lo(split() ? hi.size() : 0, (half_t)0.0f);
  std::vector<float> bias(pc.CoutW, 0.0f);
  for (int q = 0; q < 4; ++q)
    for (int co = 0; co < cout; ++co) {
      const int n = q * cpad + co;
      bias[n] = b[co];
      for (int ci = 0; ci < cin; ++ci) {
        const float v = w[((size_t)ci * cout + co) * 4 + q];  // [ci][co][dy][dx], q = dy*2+dx
        half_t h, l;
        split_half(v, &h, &l);
        const size_t d = pc.tile == 6 ? gemm_dma_pack_index(n, ci, cin_pad) : (size_t)n * cin_pad + ci;
        hi[d] = h;
        if (split()) lo[d] = l;
      }
    }
  pc.w_hi = dupload(hi);
  pc.w_lo = split() ? dupload(lo) : nullptr;
  pc.bias = dupload(bias);
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
  static const char* env = std::getenv("VP_FUSE_SKIP");
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
  if (const char* e = std::getenv("VP_CONVT_TILE")) o.tile = std::atoi(e);
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
  std::vector<float> bias(pc.CoutW, 0.0f);
  for (int q = 0; q < 4; ++q)
    for (int co = 0; co < cout; ++co) {
      const int n = q * cpad + co;
      bias[n] = bt[co] + bs[co];
      for (int k = 0; k < cin + cs; ++k) {
        const float v = k < cin ? wt[((size_t)k * cout + co) * 4 + q] : ws[(size_t)co * cs + (k - cin)];
        const size_t col = k < cin ? k : cin_pad + (k - cin);
        half_t h, l;
        split_half(v, &h, &l);
        const size_t d = pc.tile == 6 ? gemm_dma_pack_index(n, (int)col, kw) : (size_t)n * kw + col;
        hi[d] = h;
        if (split()) lo[d] = l;
      }
    }
  pc.w_hi = dupload(hi);
  pc.w_lo = split() ? dupload(lo) : nullptr;
  pc.bias = dupload(bias);
  push_conv_op(up_name + "+" + skip_name.substr(skip_name.rfind('.') == std::string::npos ? 0 : skip_name.rfind('.') + 1), in, pc, 1, ncols, o,
               out, STORE_SHUFFLE2, cout);
  return out;
}

// ------------------------------------------------------------------------------------------------ backbone
// torchvision efficientnet_b0().features as used by Models/model_components/backbone.py:9-22.
std::vector<Act*> Engine::build_backbone(const WeightBlob& blob, const std::string& P) {
  struct Stage { int e, k, s, cin, cout, n; };
  static const Stage stages[7] = {{1, 3, 1, 32, 16, 1}, {6, 3, 2, 16, 24, 2}, {6, 5, 2, 24, 40, 2}, {6, 3, 2, 40, 80, 3},
                                  {6, 5, 1, 80, 112, 3}, {6, 5, 2, 112, 192, 4}, {6, 3, 1, 192, 320, 1}};
  std::vector<Act*> stage_out;
  // ---- squeeze-excite pool accumulators of all 16 MBConv blocks: one arena, one memset node per frame
  constexpr int kSeBlocks = 16;
  const int N = frames_;  // batched encoder: activations are N frames stacked along H, one launch covers all of them where it can
  const size_t se_words = (size_t)N * ((size_t)kSeBlocks * 1536 * 8 + (size_t)6 * 256 * kSeMaxReplicas);  // >= sum of N*replicas*C below (checked)
  unsigned long long* se_arena = static_cast<unsigned long long*>(dalloc(se_words * sizeof(unsigned long long)));
  int se_block = 0;
  size_t se_used = 0;
  {
    Op op;
    op.name = "se_pool_zero";
    op.kernel = "zero_u64";
    op.bytes = 8.0 * se_words;
    op.run = [se_arena, se_words](hipStream_t st) { return launch_zero_u64(se_arena, se_words, st); };
    ops_.push_back(std::move(op));
  }
  // ---- features[0]: stem
  Act* x;
  {
    Folded f = fold_conv_bn(blob, P + "0");
    std::vector<float> wk(27 * 32);
    for (int co = 0; co < 32; ++co)
      for (int k = 0; k < 27; ++k) wk[k * 32 + co] = f.w[co * 27 + k];
    StemParams sp{};
    sp.H = net_h();
    sp.W = net_w();
    sp.w = dupload(wk);
    sp.b = dupload(f.b);
    x = new_act(P + "0", 32, N * (net_h() / 2), net_w() / 2);
    x->frames = N;
    for (int fi = 0; fi < N; ++fi) {
      sp.in = d_input_ + (size_t)fi * 3 * net_h() * net_w();
      sp.out = frame_view(x, fi)->view();
      Op op;
      op.name = P + "0";
      op.flops = 2.0 * 27 * 32 * (x->H / N) * x->W;
      op.bytes = 4.0 * 3 * net_h() * net_w() + 2.0 * x->elems() / N;
      op.run = [sp](hipStream_t st) { return launch_stem(sp, st); };
      ops_.push_back(std::move(op));
    }
  }
  stage_out.push_back(x);
  for (int si = 0; si < 7; ++si) {
    const Stage& S = stages[si];
    for (int bi = 0; bi < S.n; ++bi) {
      const std::string bp = P + std::to_string(si + 1) + "." + std::to_string(bi) + ".block.";
      const int cin = bi == 0 ? S.cin : S.cout, stride = bi == 0 ? S.s : 1, cexp = cin * S.e;
      int j = 0;
      const Act* y = x;
      if (S.e != 1) {
        Folded f = fold_conv_bn(blob, bp + std::to_string(j));
        ConvOpts o;
        o.act = ACT_SILU;
        Act* ye = add_conv(bp + std::to_string(j), x, f.w, f.b, cexp, 1, o);  // 1x1: M = N*H*W pixels in one launch
        ye->frames = N;
        y = ye;
        ++j;
      }
      // depthwise (+ fused squeeze-excite average pool: int64 fixed-point channel sums, zeroed once per frame)
      Act* z = new_act(bp + std::to_string(j), cexp, y->H / stride, y->W / stride);  // y->H = N * per-frame height, all even
      z->frames = N;
      const int sq = std::max(1, cin / 4);
      const int HWz = z->H / N * z->W;  // per frame
      // replica rows for the pool atomics (a workgroup covers >= 32 pixels of one channel group): ~4 per row, 8..64 rows
      int se_rep = 8;
      while (se_rep < kSeMaxReplicas && se_rep * 4 * 32 <= HWz) se_rep *= 2;
      if (se_used + (size_t)N * se_rep * z->C > se_words) throw std::runtime_error("SE arena too small");
      unsigned long long* sums = se_arena + se_used;  // [N][se_rep][C]
      se_used += (size_t)N * se_rep * z->C;
      ++se_block;
      {
        Folded f = fold_conv_bn(blob, bp + std::to_string(j));
        const int kk = S.k * S.k;
        std::vector<float> wk((size_t)kk * z->C, 0.0f), bk(z->C, 0.0f);
        for (int c = 0; c < cexp; ++c) {
          for (int t = 0; t < kk; ++t) wk[(size_t)t * z->C + c] = f.w[(size_t)c * kk + t];
          bk[c] = f.b[c];
        }
        DwParams dp{};
        dp.in = ActView{y->hi, y->lo, y->H / N, y->W, y->C};   // per-frame geometry; the kernel's grid.z walks the frames
        dp.out = ActView{z->hi, z->lo, z->H / N, z->W, z->C};
        dp.frames = N;
        dp.w = dupload(wk);
        dp.b = dupload(bk);
        dp.k = S.k;
        dp.stride = stride;
        dp.sums = sums;
        dp.replicas = se_rep;
        Op op;
        op.name = bp + std::to_string(j);
        op.flops = 2.0 * kk * cexp * z->H * z->W;
        op.bytes = (split() ? 4.0 : 2.0) * (y->elems() + z->elems());
        op.kernel = S.k == 3 ? "dwconv_pool<3>" : "dwconv_pool<5>";
        if (N > 1) op.kernel += ",batch";
        op.run = [dp](hipStream_t st) { return launch_dwconv(dp, st); };
        ops_.push_back(std::move(op));
        ++j;
      }
      // squeeze-excite -> per-frame scaled projection weights
      SeParams se{};
      const float *se_w2 = nullptr, *se_b2 = nullptr;
      std::string se_name;
      {
        const std::string sp = bp + std::to_string(j);
        const HostTensor& w1 = blob.get(sp + ".fc1.weight");
        const HostTensor& b1 = blob.get(sp + ".fc1.bias");
        const HostTensor& w2 = blob.get(sp + ".fc2.weight");
        const HostTensor& b2 = blob.get(sp + ".fc2.bias");
        if (w1.shape[0] != sq || w1.shape[1] != cexp) throw std::runtime_error("SE fc1 shape mismatch: " + sp);
        std::vector<float> w1p((size_t)sq * z->C, 0.0f), w2p((size_t)z->C * sq, 0.0f), b2p(z->C, 0.0f);
        for (int q = 0; q < sq; ++q)
          for (int c = 0; c < cexp; ++c) w1p[(size_t)q * z->C + c] = w1.data[(size_t)q * cexp + c];
        for (int c = 0; c < cexp; ++c) {
          for (int q = 0; q < sq; ++q) w2p[(size_t)c * sq + q] = w2.data[(size_t)c * sq + q];
          b2p[c] = b2.data[c];
        }
        se.sums = sums;
        se.replicas = se_rep;
        se.C = z->C;
        se.Creal = cexp;
        se.sq = sq;
        se.inv_hw = 1.0f / (float)HWz;
        se.w1 = dupload(w1p);
        se.b1 = dupload(b1.data);
        se_w2 = dupload(w2p);
        se_b2 = dupload(b2p);
        se.frames = N;
        se_name = sp;
        ++j;
      }
      // project 1x1 (+BN folded) with SE scale folded into K, optional residual
      {
        Folded f = fold_conv_bn(blob, bp + std::to_string(j));
        const int ncols = round_up(S.cout, 32);
        ConvOpts o;
        const bool residual = (stride == 1 && cin == S.cout);
        if (residual) {
          o.res_mode = RES_ADD;
          o.res = x;
        }
        PackedConv pc;
        choose_conv_cfg(HWz, ncols, z->C, 1, o, &pc);  // per frame: every frame has its own gate, hence its own scaled weights
        std::vector<float> wf((size_t)pc.CoutW * z->C, 0.0f), bias(pc.CoutW, 0.0f);
        for (int co = 0; co < S.cout; ++co) {
          for (int c = 0; c < cexp; ++c) wf[(size_t)co * z->C + c] = f.w[(size_t)co * cexp + c];
          bias[co] = f.b[co];
        }
        ScaleWParams sw{};
        sw.w = dupload(wf);
        sw.rows = pc.CoutW;
        sw.C = z->C;
        sw.w2 = se_w2;
        sw.b2 = se_b2;
        sw.sq = sq;
        sw.Creal = cexp;
        sw.out_hi = static_cast<half_t*>(dalloc((size_t)N * wf.size() * sizeof(half_t)));  // [N][CoutW][C]
        sw.out_lo = split() ? static_cast<half_t*>(dalloc((size_t)N * wf.size() * sizeof(half_t))) : nullptr;
        sw.frames = N;
        {
          Op op;
          op.name = se_name + ".se";
          op.flops = 4.0 * sq * cexp;
          op.bytes = 6.0 * wf.size() * N;
          op.kernel = "se_gate_scale";
          op.run = [se, sw](hipStream_t st) { return launch_se_gate_scale(se, sw, st); };
          ops_.push_back(std::move(op));
        }
        pc.bias = dupload(bias);
        Act* out = new_act(bp + std::to_string(j), S.cout, z->H, z->W);
        out->frames = N;
        for (int fi = 0; fi < N; ++fi) {
          pc.w_hi = sw.out_hi + (size_t)fi * wf.size();
          pc.w_lo = sw.out_lo ? sw.out_lo + (size_t)fi * wf.size() : nullptr;
          ConvOpts of = o;
          if (residual) of.res = frame_view(x, fi);
          push_conv_op(bp + std::to_string(j), frame_view(z, fi), pc, 1, ncols, of, frame_view(out, fi), STORE_NHWC, S.cout);
        }
        x = out;
      }
    }
    stage_out.push_back(x);
  }
  {
    Folded f = fold_conv_bn(blob, P + "8");
    ConvOpts o;
    o.act = ACT_SILU;
    Act* last = add_conv(P + "8", x, f.w, f.b, 1280, 1, o);
    last->frames = N;
    stage_out.push_back(last);
  }
  // taps l0, l2, l3, l4, l8 (backbone.py:22)
  return {stage_out[0], stage_out[2], stage_out[3], stage_out[4], stage_out[8]};
}

// ------------------------------------------------------------------------------------------------- context
// scene_context.py:25-57 (== depth_context.py, auto_steer_context.py with 1456 channels)
Act* Engine::build_context(const WeightBlob& blob, const std::string& p, const Act* deep, int cctx) {
  const int HW = deep->H * deep->W;
  const int nslab = 8;  // 200 pixels in 8 slabs of 25: 40 workgroups with ONE round of loads each (one slab: 5 workgroups walking 25 pixels
                        // per thread in 7 dependent rounds, 21 us); the first FC sums the slab partials in a fixed order
  float* partial = static_cast<float*>(dalloc((size_t)nslab * deep->C * sizeof(float)));
  {
    PoolParams pp{deep->view(), partial, nslab};
    Op op;
    op.name = p + "avgpool";
    op.run = [pp](hipStream_t st) { return launch_pool_partial(pp, st); };
    ops_.push_back(std::move(op));
  }
  const float* x = nullptr;
  int K = cctx;
  const int widths[3] = {800, 800, 200};
  for (int i = 0; i < 3; ++i) {
    const std::string lp = p + "context_layer_" + std::to_string(i);
    const HostTensor& w = blob.get(lp + ".weight");
    const HostTensor& b = blob.get(lp + ".bias");
    if (w.shape[0] != widths[i] || w.shape[1] != K) throw std::runtime_error("context MLP shape mismatch: " + lp);
    FcParams fp{};
    fp.x = x;
    fp.w = dupload(w.data);
    fp.b = dupload(b.data);
    fp.N = widths[i];
    fp.K = K;
    fp.act = i < 2 ? ACT_GELU : ACT_SIGMOID;
    fp.out = static_cast<float*>(dalloc(widths[i] * sizeof(float)));
    if (i == 0) {
      fp.partial = partial;
      fp.nslab = nslab;
      fp.Kstride = deep->C;
      fp.inv_hw = 1.0f / (float)HW;
    }
    Op op;
    op.name = lp;
    op.flops = 2.0 * widths[i] * K;
    op.bytes = 4.0 * widths[i] * K;
    op.run = [fp](hipStream_t st) { return launch_fc(fp, st); };
    ops_.push_back(std::move(op));
    x = fp.out;
    K = widths[i];
  }
  Act* c = new_act(p + "context_layer_3", 128, deep->H, deep->W);
  {
    const HostTensor& w = blob.get(p + "context_layer_3.weight");  // [128][1][3][3]
    const HostTensor& b = blob.get(p + "context_layer_3.bias");
    std::vector<float> wk(9 * c->C, 0.0f), bk(c->C, 0.0f);
    for (int co = 0; co < 128; ++co) {
      for (int t = 0; t < 9; ++t) wk[t * c->C + co] = w.data[co * 9 + t];
      bk[co] = b.data[co];
    }
    CtxConv1Params cp{};
    cp.act = ACT_GELU;
    cp.map = x;
    cp.H = deep->H;
    cp.W = deep->W;
    cp.w = dupload(wk);
    cp.b = dupload(bk);
    cp.out = c->view();
    Op op;
    op.name = p + "context_layer_3";
    op.flops = 2.0 * 9 * 128 * HW;
    op.run = [cp](hipStream_t st) { return launch_ctx_conv1(cp, st); };
    ops_.push_back(std::move(op));
  }
  const int couts[3] = {256, 512, cctx};
  for (int i = 0; i < 3; ++i) {
    const std::string lp = p + "context_layer_" + std::to_string(4 + i);
    ConvOpts o;
    o.act = ACT_GELU;
    if (i == 2) {  // context = gelu(c7) * features + features  (scene_context.py:53-56)
      o.res_mode = RES_MULADD;
      o.res = deep;
    }
    c = add_conv(lp, c, blob.get(lp + ".weight").data, blob.get(lp + ".bias").data, couts[i], 3, o);
  }
  return c;
}

// ---------------------------------------------------------------------------------------------------- neck
// scene_neck.py:26-60 (== scene_3d_neck.py, ego_path_neck.py)
Act* Engine::build_neck(const WeightBlob& blob, const std::string& p, const Act* ctx, const std::vector<Act*>& feats, int cctx) {
  const int up_c[3] = {cctx, 768, 512};
  const int d_c[6] = {768, 768, 512, 512, 512, 256};
  const Act* x = ctx;
  for (int blk = 0; blk < 3; ++blk) {
    const std::string up = p + "upsample_layer_" + std::to_string(blk), sk = p + "skip_link_layer_" + std::to_string(blk);
    // d = upsample(x) + skip(feature): one GEMM over K = [x channels | feature channels]
    x = add_convT_skip(up, sk, x, feats[3 - blk], blob.get(up + ".weight").data, blob.get(up + ".bias").data,
                       blob.get(sk + ".weight").data, blob.get(sk + ".bias").data, up_c[blk]);
    for (int k = 0; k < 2; ++k) {
      const std::string dl = p + "decode_layer_" + std::to_string(2 * blk + k);
      ConvOpts o;
      o.act = ACT_GELU;
      x = add_conv(dl, x, blob.get(dl + ".weight").data, blob.get(dl + ".bias").data, d_c[2 * blk + k], 3, o);
    }
  }
  return const_cast<Act*>(x);
}

// ---------------------------------------------------------------------------------------------------- heads
void Engine::build_head(const WeightBlob& blob, const std::string& p, const Act* neck, const std::vector<Act*>& feats) {
  auto W = [&](const std::string& k) -> const std::vector<float>& { return blob.get(p + k + ".weight").data; };
  auto B = [&](const std::string& k) -> const std::vector<float>& { return blob.get(p + k + ".bias").data; };
  ConvOpts gelu;
  gelu.act = ACT_GELU;
  const Act* x = neck;
  std::string last;
  int c_last = 0;
  if (kind_ == 3) {  // ego_lanes_head.py:18-26 (80x160)
    x = add_conv(p + "decode_layer_6", x, W("decode_layer_6"), B("decode_layer_6"), 256, 3, gelu);
    x = add_conv(p + "decode_layer_7", x, W("decode_layer_7"), B("decode_layer_7"), 128, 3, gelu);
    last = "decode_layer_8";
    c_last = 3;
  } else {  // scene_seg_head.py:21-44, scene_3d_head.py:23-47, domain_seg_head.py:21-44
    const int c9 = kind_ == 1 ? 128 : 64;
    const Act* u = add_convT_skip(p + "upsample_layer_3", p + "skip_link_layer_3", x, feats[0], W("upsample_layer_3"),
                                  B("upsample_layer_3"), W("skip_link_layer_3"), B("skip_link_layer_3"), 256);
    x = add_conv(p + "decode_layer_6", u, W("decode_layer_6"), B("decode_layer_6"), 256, 3, gelu);
    x = add_conv(p + "decode_layer_7", x, W("decode_layer_7"), B("decode_layer_7"), 128, 3, gelu);
    x = add_convT(p + "upsample_layer_4", x, W("upsample_layer_4"), B("upsample_layer_4"), 128, ConvOpts{});
    x = add_conv(p + "decode_layer_8", x, W("decode_layer_8"), B("decode_layer_8"), 128, 3, gelu);
    x = add_conv(p + "decode_layer_9", x, W("decode_layer_9"), B("decode_layer_9"), c9, 3, gelu);
    last = "decode_layer_10";
    c_last = kind_ == 0 ? 3 : 1;
  }
  out_c_ = c_last;
  out_h_ = x->H;
  out_w_ = x->W;
  d_logits_ = static_cast<float*>(dalloc((size_t)out_c_ * out_h_ * out_w_ * sizeof(float)));
  d_mask_ = static_cast<uint8_t*>(dalloc((size_t)out_h_ * out_w_));
  ConvOpts fo;
  fo.logits_out = d_logits_;
  add_conv(p + last, x, W(last), B(last), c_last, 3, fo);
}

void Engine::build_model(const WeightBlob& blob) {
  struct Prefix { const char *bb, *ctx, *neck, *head; };
  static const Prefix P[4] = {
      {"Backbone.encoder.", "SceneContext.", "SceneNeck.", "SceneSegHead."},
      {"PreTrainedBackbone.pretrainedBackBone.encoder.", "DepthContext.", "DepthNeck.", "SuperDepthHead."},
      {"DomainSegUpstream.pretrainedBackBone.encoder.", "DomainSegUpstream.pretrainedContext.", "DomainSegUpstream.pretrainedNeck.",
       "DomainSegHead."},
      {"BEVBackbone.encoder.", "AutoSteerContext.", "EgopathNeck.", "EgoLanesHead."}};
  if (kind_ == 4) {
    if (base_) throw std::invalid_argument("AutoDrive engines cannot be shared-prefix engines");
    build_autodrive(blob);
    return;
  }
  if (kind_ < 0 || kind_ > 3) throw std::invalid_argument("unknown model kind");
  const Prefix& pf = P[kind_];
  hash_bb_ = blob.group_hash(pf.bb);
  hash_ctx_ = blob.group_hash(pf.ctx);
  hash_neck_ = blob.group_hash(pf.neck);
  if (base_) {
    // Which prefix of the network is the base engine's?  Scene3D / DomainSeg are built on a pre-trained SceneSeg
    // (scene_3d_network.py:13, domain_seg_network.py:11): same backbone parameters, DomainSeg also the same context
    // and neck.  EgoLanes fuses all five taps before its context (ego_lanes_network.py:30-36): backbone only.
    if (hash_bb_ != base_->hash_bb_) throw std::invalid_argument("shared engine: backbone parameters differ from the base engine's");
    shared_level_ = 1;
    if (base_->frames_ == 1 && kind_ != 3 && base_->kind_ != 3 && hash_ctx_ == base_->hash_ctx_ && hash_neck_ == base_->hash_neck_) shared_level_ = 2;
  }
  if (!base_) {
  d_input_ = static_cast<float*>(dalloc((size_t)frames_ * 3 * net_h() * net_w() * sizeof(float)));
  // op 0 (.. frames-1): preprocess (parameters are patched per frame geometry in ensure_tables)
  for (int fi = 0; fi < frames_; ++fi) {
    Op op;
    op.name = "preprocess";
    op.run = [this, fi](hipStream_t st) {
      PreprocessParams pp{};
      pp.frame = d_frame_ + (size_t)fi * frame_h_ * frame_stride_;
      pp.stride = frame_stride_;
      pp.xtab = d_xtab_;
      pp.ytab = d_ytab_;
      pp.out_h = net_h();
      pp.out_w = net_w();
      // plane colour order: RGB planes or BGR planes; source byte index depends on the frame's pixel format
      static const float mean_rgb[3] = {0.485f, 0.456f, 0.406f}, std_rgb[3] = {0.229f, 0.224f, 0.225f};
      for (int c = 0; c < 3; ++c) {
        const int colour = plane_order_ == 1 ? c : 2 - c;       // 0=R 1=G 2=B
        pp.src_c[c] = pixel_format_ == 1 ? colour : 2 - colour;  // RGB8: R at byte 0 ; BGR8: R at byte 2
        pp.mean[c] = mean_rgb[colour];
        pp.stdv[c] = std_rgb[colour];
      }
      pp.out = d_input_ + (size_t)fi * 3 * net_h() * net_w();
      if (resize_mode_ != 0) return launch_pil_resample(pil_params(pp), st);
      return launch_preprocess(pp, st);
    };
    ops_.push_back(std::move(op));
    first_net_op_ = (size_t)frames_;
  }
  }  // !base_
  std::vector<Act*> feats = base_ ? base_->feats_ : build_backbone(blob, pf.bb);
  if (!base_) n_fork_ops_ = ops_.size();
  if (base_ && base_->frames_ > 1)
    for (Act*& t : feats) t = frame_view(t, frame_index_);
  feats_ = feats;
  if (frames_ > 1) return;  // batched encoder: taps only
  const int cctx = kind_ == 3 ? 1456 : 1280;
  const Act* deep = feats[4];
  if (kind_ == 3) {  // backbone_feature_fusion.py:13-38
    Act* fused = new_act("BackboneFeatureFusion", 1456, feats[4]->H, feats[4]->W);
    FusionParams fp{};
    const int shifts[5] = {4, 3, 2, 1, 0};
    for (int i = 0; i < 5; ++i) {
      fp.f[i] = feats[i]->view();
      fp.creal[i] = feats[i]->Creal;
      fp.shift[i] = shifts[i];
    }
    fp.out = fused->view();
    fp.Creal_out = 1456;
    Op op;
    op.name = "BackboneFeatureFusion";
    op.run = [fp](hipStream_t st) { return launch_fusion(fp, st); };
    ops_.push_back(std::move(op));
    deep = fused;
  }
  Act* neck;
  if (shared_level_ == 2) {
    neck = base_->neck_out_;
  } else {
    Act* ctx = build_context(blob, pf.ctx, deep, cctx);
    neck = build_neck(blob, pf.neck, ctx, feats, cctx);
  }
  neck_out_ = neck;
  build_head(blob, pf.head, neck, feats);
  decode_mode_ = kind_ == 3 ? 1 : 0;
}

// ------------------------------------------------------------------------------------------------ AutoDrive
// autodrive_network.py:32-36, autodrive_backbone.py:8-48, autodrive_head.py:70-87, common_layers.py (Conv, CTX, SPPF,
// C2PSA, PSABlock, Attention).  One plan: preprocess -> backbone (P5 256x16x32) -> [shift: previous frame's P5 to the
// first half of the 512-channel head input] -> [place: this frame's P5 to the second half] -> head -> 3 scalars.
void Engine::build_autodrive(const WeightBlob& blob) {
  auto T = [&](const std::string& k) -> const HostTensor& { return blob.get(k); };
  auto push = [&](const std::string& name, const char* kernel, std::function<hipError_t(hipStream_t)> fn, double flops = 0, double bytes = 0) {
    Op op;
    op.name = name;
    op.kernel = kernel;
    op.flops = flops;
    op.bytes = bytes;
    op.run = std::move(fn);
    ops_.push_back(std::move(op));
  };
  d_input_ = static_cast<float*>(dalloc((size_t)3 * net_h() * net_w() * sizeof(float)));
  plane_order_ = 1;  // RGB planes, ImageNet constants (visualizations/AutoDrive/video_visualization.py:29-33)
  resize_mode_ = 1;  // ... behind PIL's antialiased Image.resize(..., Image.BILINEAR) (same lines)
  push("preprocess", "preprocess", [this](hipStream_t st) {
    PreprocessParams pp{};
    pp.frame = d_frame_;
    pp.stride = frame_stride_;
    pp.xtab = d_xtab_;
    pp.ytab = d_ytab_;
    pp.out_h = net_h();
    pp.out_w = net_w();
    static const float mean_rgb[3] = {0.485f, 0.456f, 0.406f}, std_rgb[3] = {0.229f, 0.224f, 0.225f};
    for (int c = 0; c < 3; ++c) {
      const int colour = plane_order_ == 1 ? c : 2 - c;
      pp.src_c[c] = pixel_format_ == 1 ? colour : 2 - colour;
      pp.mean[c] = mean_rgb[colour];
      pp.stdv[c] = std_rgb[colour];
    }
    pp.out = d_input_;
    if (resize_mode_ != 0) return launch_pil_resample(pil_params(pp), st);
    return launch_preprocess(pp, st);
  });
  first_net_op_ = 1;

  // ---- p1: Conv 3->16 k3 s2 + BN + SiLU on the fp32 planes (the stem kernel computes 32 output channels: 16 are padding)
  Act* x;
  {
    Folded f = fold_conv_norm(blob, "backbone.p1");
    if (f.cout > 32 || f.cin != 3 || f.k != 3) throw std::runtime_error("backbone.p1 shape mismatch");
    std::vector<float> wk(27 * 32, 0.0f), bk(32, 0.0f);
    for (int co = 0; co < f.cout; ++co) {
      for (int k = 0; k < 27; ++k) wk[k * 32 + co] = f.w[co * 27 + k];
      bk[co] = f.b[co];
    }
    StemParams sp{};
    sp.in = d_input_;
    sp.H = net_h();
    sp.W = net_w();
    sp.w = dupload(wk);
    sp.b = dupload(bk);
    x = new_act("backbone.p1", f.cout, net_h() / 2, net_w() / 2);
    sp.out = x->view();
    push("backbone.p1", "stem", [sp](hipStream_t st) { return launch_stem(sp, st); }, 2.0 * 27 * f.cout * x->H * x->W,
         4.0 * 3 * net_h() * net_w() + 2.0 * x->elems());
  }
  auto conv_bn = [&](const std::string& name, const Act* in, int ks, int stride, int act) -> Act* {
    Folded f = fold_conv_norm(blob, name);
    ConvOpts o;
    o.act = act;
    o.stride = stride;
    return add_conv(name, in, f.w, f.b, f.cout, ks, o);
  };
  // ---- p2..p5: strided conv + CTX (common_layers.py:183-227)
  const char* stage_names[4] = {"backbone.p2", "backbone.p3", "backbone.p4", "backbone.p5"};
  for (int si = 0; si < 4; ++si) {
    const std::string sn = stage_names[si];
    Act* a = conv_bn(sn + ".0", x, 3, 2, ACT_SILU);
    const std::string cp = sn + ".1";
    const int HW = a->H * a->W, C = a->Creal;
    // mean over H, W (:206) as slab partial sums
    const int nslab = std::max(1, std::min(64, HW / 512));
    float* partial = static_cast<float*>(dalloc((size_t)nslab * a->C * sizeof(float)));
    {
      PoolParams pp{a->view(), partial, nslab};
      push(cp + ".mean", "pool_partial", [pp](hipStream_t st) { return launch_pool_partial(pp, st); });
    }
    // exp0: Conv1d(C -> H*W, k3, pad 1) on a length-1 sequence == the centre tap as a [H*W][C] matrix (:210), SiLU twice (:211-213)
    const HostTensor& ew = T(cp + ".exp0.weight");
    const HostTensor& eb = T(cp + ".exp0.bias");
    if (ew.shape.size() != 3 || ew.shape[0] != HW || ew.shape[1] != C || ew.shape[2] != 3) throw std::runtime_error("exp0 shape mismatch: " + cp);
    std::vector<float> wm((size_t)HW * a->C, 0.0f);
    for (int n = 0; n < HW; ++n)
      for (int c = 0; c < C; ++c) wm[(size_t)n * a->C + c] = ew.data[((size_t)n * C + c) * 3 + 1];
    FcParams fp{};
    fp.w = dupload(wm);
    fp.b = dupload(eb.data);
    fp.N = HW;
    fp.K = a->C;
    fp.act = ACT_SILU2;
    fp.out = static_cast<float*>(dalloc((size_t)HW * sizeof(float)));
    fp.partial = partial;
    fp.nslab = nslab;
    fp.Kstride = a->C;
    fp.inv_hw = 1.0f / (float)HW;
    push(cp + ".exp0", "fc", [fp](hipStream_t st) { return launch_fc(fp, st); }, 2.0 * HW * C, 4.0 * HW * C);
    // ctx0: conv3x3 1 -> C/2 + SiLU (:216-217)
    const HostTensor& w0 = T(cp + ".ctx0.weight");
    const HostTensor& b0 = T(cp + ".ctx0.bias");
    const int c0n = w0.shape[0];
    Act* c2 = new_act(cp + ".ctx0", c0n, a->H, a->W);
    {
      std::vector<float> wk((size_t)9 * c2->C, 0.0f), bk(c2->C, 0.0f);
      for (int co = 0; co < c0n; ++co) {
        for (int t = 0; t < 9; ++t) wk[(size_t)t * c2->C + co] = w0.data[(size_t)co * 9 + t];
        bk[co] = b0.data[co];
      }
      CtxConv1Params cpp{};
      cpp.map = fp.out;
      cpp.H = a->H;
      cpp.W = a->W;
      cpp.w = dupload(wk);
      cpp.b = dupload(bk);
      cpp.out = c2->view();
      cpp.act = ACT_SILU;
      push(cp + ".ctx0", "ctx_conv1", [cpp](hipStream_t st) { return launch_ctx_conv1(cpp, st); }, 2.0 * 9 * c0n * HW);
    }
    // ctx1: conv3x3 C/2 -> C + SiLU, gate: c4*x + x, SiLU (:218-224)
    ConvOpts o1;
    o1.act = ACT_SILU;
    o1.res_mode = RES_MULADD;
    o1.res = a;
    o1.post_act = ACT_SILU;
    Act* g = add_conv(cp + ".ctx1", c2, T(cp + ".ctx1.weight").data, T(cp + ".ctx1.bias").data, C, 3, o1);
    // ctx2: conv3x3 C -> Cout, no activation (:225)
    x = add_conv(cp + ".ctx2", g, T(cp + ".ctx2.weight").data, T(cp + ".ctx2.bias").data, T(cp + ".ctx2.weight").shape[0], 3, ConvOpts{});
  }
  // ---- SPPF (common_layers.py:230-243): cv1, three chained 5x5 max-pools, concat, cv2
  {
    const std::string sp = "backbone.p5.2";
    Act* c1 = conv_bn(sp + ".cv1", x, 1, 1, ACT_SILU);
    const int c_ = c1->Creal;
    Act* cat = new_act(sp + ".cat", 4 * c_, x->H, x->W);
    const ActView cv = cat->view(), c1v = c1->view();
    push(sp + ".cat0", "chan_copy", [=](hipStream_t st) { return launch_chan_copy(c1v, 0, cv, 0, c_, st); });
    for (int i = 0; i < 3; ++i)
      push(sp + ".maxpool" + std::to_string(i), "maxpool5", [=](hipStream_t st) { return launch_maxpool5(cv, i * c_, cv, (i + 1) * c_, c_, st); });
    x = conv_bn(sp + ".cv2", cat, 1, 1, ACT_SILU);
  }
  // ---- C2PSA (common_layers.py:246-257) with one PSABlock (:107-118) and its Attention (:78-104)
  {
    const std::string cp = "backbone.p5.3", mb = cp + ".middle_block";
    Act* t = conv_bn(cp + ".cv1", x, 1, 1, ACT_SILU);       // 2*c_ channels: [a | y]
    const int c_ = t->Creal / 2;
    Act* y = new_act(cp + ".y", c_, t->H, t->W);
    const ActView tv = t->view(), yv = y->view();
    push(cp + ".split", "chan_copy", [=](hipStream_t st) { return launch_chan_copy(tv, c_, yv, 0, c_, st); });
    const int heads = c_ / 64, dv = c_ / heads, dk = dv / 2;
    Act* qkv = conv_bn(mb + ".conv1.qkv", y, 1, 1, ACT_NONE);
    if (qkv->Creal != heads * (2 * dk + dv)) throw std::runtime_error("attention qkv width mismatch");
    Act* att = new_act(mb + ".conv1.attn", c_, t->H, t->W);
    Act* vv = new_act(mb + ".conv1.v", c_, t->H, t->W);
    AttnParams ap{};
    ap.qkv = qkv->view();
    ap.out = att->view();
    ap.vout = vv->view();
    ap.heads = heads;
    ap.dk = dk;
    ap.dv = dv;
    ap.scale = 1.0f / std::sqrt((float)dk);
    const int Tn = t->H * t->W;
    push(mb + ".conv1.attention", "attention", [ap](hipStream_t st) { return launch_attention(ap, st); }, 2.0 * heads * Tn * (double)Tn * (dk + dv));
    // + depthwise 3x3 positional conv of v (BN folded, identity activation)
    Act* pe = new_act(mb + ".conv1.pe", c_, t->H, t->W);
    {
      Folded f = fold_conv_norm(blob, mb + ".conv1.conv1");
      std::vector<float> wk((size_t)9 * vv->C, 0.0f), bk(vv->C, 0.0f);
      for (int c = 0; c < c_; ++c) {
        for (int k = 0; k < 9; ++k) wk[(size_t)k * vv->C + c] = f.w[(size_t)c * 9 + k];
        bk[c] = f.b[c];
      }
      DwPlainParams dp{};
      dp.in = vv->view();
      dp.add = att->view();
      dp.out = pe->view();
      dp.w = dupload(wk);
      dp.b = dupload(bk);
      push(mb + ".conv1.conv1", "dwconv_plain", [dp](hipStream_t st) { return launch_dwconv_plain(dp, st); }, 2.0 * 9 * c_ * Tn);
    }
    // x = y + conv2(attn + pe)
    Act* y1;
    {
      Folded f = fold_conv_norm(blob, mb + ".conv1.conv2");
      ConvOpts o;
      o.res_mode = RES_ADD;
      o.res = y;
      y1 = add_conv(mb + ".conv1.conv2", pe, f.w, f.b, f.cout, 1, o);
    }
    // x = x + ffn(x)
    Act* h = conv_bn(mb + ".conv2.0", y1, 1, 1, ACT_SILU);
    Act* y2;
    {
      Folded f = fold_conv_norm(blob, mb + ".conv2.1");
      ConvOpts o;
      o.res_mode = RES_ADD;
      o.res = y1;
      y2 = add_conv(mb + ".conv2.1", h, f.w, f.b, f.cout, 1, o);
    }
    const ActView y2v = y2->view();
    push(cp + ".cat", "chan_copy", [=](hipStream_t st) { return launch_chan_copy(y2v, 0, tv, c_, c_, st); });  // cat((a, y), 1) in place
    x = conv_bn(cp + ".cv2", t, 1, 1, ACT_SILU);
  }
  // ---- head (autodrive_head.py:70-87): cat([prev, curr]) -> 3 x (conv3x3 + SiLU) -> flatten (C-major) -> MLP
  const int c5 = x->Creal;
  Act* cat = new_act("head.cat", 2 * c5, x->H, x->W);
  {
    const ActView cv = cat->view(), xv = x->view();
    ad_shift_op_ = ops_.size();
    push("head.shift_prev", "chan_copy", [=](hipStream_t st) { return launch_chan_copy(cv, c5, cv, 0, c5, st); });
    ad_place_op_ = ops_.size();
    push("head.place_curr", "chan_copy", [=](hipStream_t st) { return launch_chan_copy(xv, 0, cv, c5, c5, st); });
  }
  ConvOpts so;
  so.act = ACT_SILU;
  Act* h1 = add_conv("head.conv_1", cat, T("head.conv_1.weight").data, T("head.conv_1.bias").data, T("head.conv_1.weight").shape[0], 3, so);
  Act* h2 = add_conv("head.conv_2", h1, T("head.conv_2.weight").data, T("head.conv_2.bias").data, T("head.conv_2.weight").shape[0], 3, so);
  Act* h3 = add_conv("head.conv_3", h2, T("head.conv_3.weight").data, T("head.conv_3.bias").data, T("head.conv_3.weight").shape[0], 3, so);
  const int flat = h3->Creal * h3->H * h3->W;
  float* v0 = static_cast<float*>(dalloc((size_t)flat * sizeof(float)));
  {
    const ActView hv = h3->view();
    const int cr = h3->Creal;
    push("head.flatten", "act_to_nchw", [=](hipStream_t st) { return launch_act_to_nchw(hv, cr, v0, st); });
  }
  out_c_ = 3;
  out_h_ = 1;
  out_w_ = 1;
  d_logits_ = static_cast<float*>(dalloc(3 * sizeof(float)));
  d_mask_ = static_cast<uint8_t*>(dalloc(1));
  auto fc = [&](const std::string& name, const float* in, int K, int act, float* out) -> float* {
    const HostTensor& w = T(name + ".weight");
    const HostTensor& b = T(name + ".bias");
    if (w.shape[1] != K) throw std::runtime_error("linear shape mismatch: " + name);
    FcParams fp{};
    fp.x = in;
    fp.w = dupload(w.data);
    fp.b = dupload(b.data);
    fp.N = w.shape[0];
    fp.K = K;
    fp.act = act;
    fp.out = out ? out : static_cast<float*>(dalloc((size_t)w.shape[0] * sizeof(float)));
    push(name, "fc", [fp](hipStream_t st) { return launch_fc(fp, st); }, 2.0 * fp.N * K, 4.0 * fp.N * K);
    return fp.out;
  };
  const float* f1 = fc("head.fc1.0", v0, flat, ACT_SILU, nullptr);
  const float* f2 = fc("head.fc2.0", f1, 768, ACT_SILU, nullptr);
  fc("head.distance_head.0", f2, 512, ACT_RELU, d_logits_ + 0);
  fc("head.curvature_head.0", f2, 512, ACT_TANH, d_logits_ + 1);
  fc("head.flag_head", f2, 512, ACT_NONE, d_logits_ + 2);
  decode_mode_ = 0;
}

// Runs preprocess + backbone on the resident frame and parks its P5 features in the "current" slot, WITHOUT the head:
// the next enqueue() shifts them to "previous".  Eager launches (twice per stream at most: vp_infer_pair / first frame).
void Engine::prime_previous() {
  if (kind_ != 4) throw std::invalid_argument("prime_previous: AutoDrive engines only");
  VP_HIP_CHECK(hipSetDevice(gpu_));
  if (!input_is_tensor_ && !d_frame_) throw std::runtime_error("no frame resident");
  for (size_t i = input_is_tensor_ ? first_net_op_ : 0; i <= ad_place_op_; ++i) {
    if (i == ad_shift_op_) continue;
    hipError_t e = ops_[i].run(stream_);
    if (e != hipSuccess) throw std::runtime_error("launch failed in layer '" + ops_[i].name + "': " + hipGetErrorString(e));
  }
  ad_primed_ = true;
}

void Engine::finish_plan() {
  if (d_logits_ && !h_logits_) {
    VP_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&h_logits_), (size_t)out_c_ * out_h_ * out_w_ * sizeof(float), hipHostMallocDefault));
    VP_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&h_mask_), (size_t)out_h_ * out_w_, hipHostMallocDefault));
    if (kind_ != 4 && !decode_fused_) {  // AutoDrive returns three scalars: no mask to decode; else the logits conv already decoded
      Op op;
      op.name = "decode";
      op.bytes = 4.0 * out_c_ * out_h_ * out_w_ + out_h_ * out_w_;
      op.run = [this](hipStream_t st) { return launch_decode_mask(d_logits_, out_c_, out_h_ * out_w_, decode_mode_, d_mask_, st); };
      ops_.push_back(std::move(op));
    }
    // range probe: an activation that left the fp16 range surfaces in the logits as inf / NaN (kernels_misc.hip finite_probe_kernel)
    d_status_ = static_cast<unsigned*>(dalloc(sizeof(unsigned), true));
    VP_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&h_status_), sizeof(unsigned), hipHostMallocDefault));
    *h_status_ = 0;
    Op probe;
    probe.name = "finite_probe";
    probe.kernel = "finite_probe";
    probe.bytes = 4.0 * out_c_ * out_h_ * out_w_;
    probe.run = [this](hipStream_t st) {
      return finite_check_ ? launch_finite_probe(d_logits_, (size_t)out_c_ * out_h_ * out_w_, d_status_, st) : hipSuccess;
    };
    ops_.push_back(std::move(probe));
  }
  // kernel tags of the non-GEMM launches (the conv ops set theirs in push_conv_op)
  auto ends_with = [](const std::string& s, const char* suf) {
    const size_t n = std::strlen(suf);
    return s.size() >= n && s.compare(s.size() - n, n, suf) == 0;
  };
  for (Op& op : ops_) {
    if (!op.kernel.empty()) continue;
    const std::string& n = op.name;
    if (n == "preprocess") op.kernel = "preprocess";
    else if (n == "decode") op.kernel = "decode_mask";
    else if (n == "BackboneFeatureFusion") op.kernel = "fusion";
    else if (ends_with(n, ".avgpool") || ends_with(n, "avgpool")) op.kernel = "pool_partial";
    else if (ends_with(n, "context_layer_3")) op.kernel = "ctx_conv1";
    else if (n.find("context_layer_") != std::string::npos) op.kernel = "fc";
    else if (ends_with(n, "encoder.0")) op.kernel = "stem";
    else op.kernel = "dwconv";
  }
  VP_HIP_CHECK(hipStreamSynchronize(stream_));
  VP_HIP_CHECK(hipDeviceSynchronize());
}

// ------------------------------------------------------------------------------------------ frame handling
void Engine::set_input_format(int pixel_format, int plane_order) {
  if (pixel_format < 0 || pixel_format > 1 || plane_order < 0 || plane_order > 1) throw std::invalid_argument("bad input format");
  if (pixel_format != pixel_format_ || plane_order != plane_order_) { graph_valid_ = false; ++plan_epoch_; }
  pixel_format_ = pixel_format;
  plane_order_ = plane_order;
}
void Engine::set_decode_mode(int mode) {
  if (mode < 0 || mode > 2) throw std::invalid_argument("bad decode mode");
  if (mode != decode_mode_) { graph_valid_ = false; ++plan_epoch_; }
  decode_mode_ = mode;
}

// 11-bit fixed-point bilinear taps -- must stay bit-identical to oracle/pre_post.py linear_taps_u8.
static void linear_taps_u8(int src, int dst, std::vector<int>* tab) {
  tab->resize((size_t)dst * 4);
  const double scale = (double)src / (double)dst;
  for (int d = 0; d < dst; ++d) {
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    int s = (int)std::floor(f);
    f -= (float)s;
    if (s < 0) {
      f = 0.0f;
      s = 0;
    }
    if (s >= src - 1) {
      f = 0.0f;
      s = src - 1;
    }
    (*tab)[4 * d + 0] = s;
    (*tab)[4 * d + 1] = std::min(s + 1, src - 1);
    (*tab)[4 * d + 2] = (int)std::nearbyint((1.0f - f) * 2048.0f);
    (*tab)[4 * d + 3] = (int)std::nearbyint(f * 2048.0f);
  }
}

// Pillow's precompute_coeffs + normalize_coeffs_8bpc (src/libImaging/Resample.c) in the same double arithmetic, operation for
// operation (restated and pinned against PIL in oracle/pre_post.py pil_resample_coeffs): filter support scaled by the
// down-scaling factor, taps normalised to sum 1, quantised to 22 fractional bits.  filter: 1 = BILINEAR, 2 = BICUBIC (a = -0.5).
#pragma clang fp contract(off)
int pil_coeffs(int in_size, int out_size, int filter, std::vector<int>* bounds, std::vector<int>* kk) {
  auto weight = [filter](double x) -> double {
    if (x < 0.0) x = -x;
    if (filter == 1) return x < 1.0 ? 1.0 - x : 0.0;
    const double a = -0.5;
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
    if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
    return 0.0;
  };
  double filterscale = (double)in_size / out_size;
  const double scale = filterscale;
  if (filterscale < 1.0) filterscale = 1.0;
  const double support = (filter == 1 ? 1.0 : 2.0) * filterscale;
  const int ksize = (int)std::ceil(support) * 2 + 1;
  bounds->assign((size_t)out_size * 2, 0);
  kk->assign((size_t)out_size * ksize, 0);
  std::vector<double> k(ksize);
  const double ss = 1.0 / filterscale;
  for (int xx = 0; xx < out_size; ++xx) {
    const double center = (xx + 0.5) * scale;
    double ww = 0.0;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    for (int x = 0; x < xmax; ++x) {
      const double w = weight((x + xmin - center + 0.5) * ss);
      k[x] = w;
      ww += w;
    }
    for (int x = 0; x < xmax; ++x) {
      if (ww != 0.0) k[x] /= ww;
      (*kk)[(size_t)xx * ksize + x] = k[x] < 0 ? (int)(-0.5 + k[x] * (1 << 22)) : (int)(0.5 + k[x] * (1 << 22));
    }
    (*bounds)[2 * xx] = xmin;
    (*bounds)[2 * xx + 1] = xmax;
  }
  return ksize;
}

PilResampleParams Engine::pil_params(const PreprocessParams& pp) const {
  PilResampleParams q{};
  q.frame = pp.frame;
  q.stride = pp.stride;
  q.in_h = frame_h_;
  q.in_w = frame_w_;
  q.out_h = pp.out_h;
  q.out_w = pp.out_w;
  q.hb = d_pil_hb_;
  q.hk = d_pil_hk_;
  q.hks = pil_hks_;
  q.vb = d_pil_vb_;
  q.vk = d_pil_vk_;
  q.vks = pil_vks_;
  q.tmp = d_pil_tmp_;
  for (int c = 0; c < 3; ++c) {
    q.src_c[c] = pp.src_c[c];
    q.mean[c] = pp.mean[c];
    q.stdv[c] = pp.stdv[c];
  }
  q.out = pp.out;
  return q;
}

void Engine::set_resize_mode(int mode) {
  if (mode < 0 || mode > 2) throw std::invalid_argument("resize mode: 0 = cv::resize INTER_LINEAR model, 1 = PIL BILINEAR, 2 = PIL BICUBIC");
  if (base_) throw std::invalid_argument("shared engine: the base engine owns the frame path");
  if (mode != resize_mode_) {
    resize_mode_ = mode;
    tab_h_ = tab_w_ = 0;
    graph_valid_ = false;
    ++plan_epoch_;
  }
}

void Engine::ensure_tables(int h, int w) {
  if (h == tab_h_ && w == tab_w_) return;
  if (resize_mode_ != 0) {  // Pillow's resample: per-output tap tables for both passes + the u8 image between them
    std::vector<int> hb, hk, vb, vk;
    pil_hks_ = pil_coeffs(w, net_w(), resize_mode_, &hb, &hk);
    pil_vks_ = pil_coeffs(h, net_h(), resize_mode_, &vb, &vk);
    VP_HIP_CHECK(hipStreamSynchronize(stream_));
    d_pil_hb_ = dupload(hb);
    d_pil_hk_ = dupload(hk);
    d_pil_vb_ = dupload(vb);
    d_pil_vk_ = dupload(vk);
    d_pil_tmp_ = static_cast<uint8_t*>(dalloc((size_t)h * net_w() * 3, false));
    tab_h_ = h;
    tab_w_ = w;
    return;
  }
  std::vector<int> xt, yt;
  linear_taps_u8(w, net_w(), &xt);
  linear_taps_u8(h, net_h(), &yt);
  if (!d_xtab_) {
    d_xtab_ = static_cast<int*>(dalloc(net_w() * 4 * sizeof(int)));
    d_ytab_ = static_cast<int*>(dalloc(net_h() * 4 * sizeof(int)));
  }
  VP_HIP_CHECK(hipStreamSynchronize(stream_));
  VP_HIP_CHECK(hipMemcpy(d_xtab_, xt.data(), xt.size() * sizeof(int), hipMemcpyHostToDevice));
  VP_HIP_CHECK(hipMemcpy(d_ytab_, yt.data(), yt.size() * sizeof(int), hipMemcpyHostToDevice));
  tab_h_ = h;
  tab_w_ = w;
}

void Engine::upload_frame(const uint8_t* frame, int h, int w, int stride, int index) {
  if (base_) throw std::invalid_argument("shared engine: frames go to the base engine (vp_infer on the base, then vp_infer_shared)");
  if (!frame || h < 2 || w < 2 || stride < 3 * w) throw std::invalid_argument("bad frame geometry");
  if (index < 0 || index >= frames_) throw std::invalid_argument("frame index out of range");
  VP_HIP_CHECK(hipSetDevice(gpu_));
  const size_t need = (size_t)h * stride;
  if ((h != frame_h_ || w != frame_w_ || stride != frame_stride_) && frames_ > 1 && index != 0 && frame_h_ != 0)
    throw std::invalid_argument("batched encoder: all frames of a pass share one geometry (upload slot 0 first to change it)");
  if (need * frames_ > frame_cap_) {
    VP_HIP_CHECK(hipStreamSynchronize(stream_));
    d_frame_ = static_cast<uint8_t*>(dalloc(need * frames_, true));
    frame_cap_ = need * frames_;
    { graph_valid_ = false; ++plan_epoch_; }
  }
  if (h != frame_h_ || w != frame_w_ || stride != frame_stride_) { graph_valid_ = false; ++plan_epoch_; }
  ensure_tables(h, w);
  frame_h_ = h;
  frame_w_ = w;
  frame_stride_ = stride;
  // A strided view (cv::Mat ROI, numpy slice) guarantees only (h-1)*stride + 3*w readable bytes: the tail of the last row
  // belongs to the parent image or to nobody.  Packed frames go as one copy, views row by row (hipMemcpy2D).
  // The caller's buffer is pageable (cv::Mat); the copy is staged through this engine's pinned buffer so the transfer
  // itself is one DMA that overlaps other engines' kernels (the reference does the same H2D: tensorrt_backend.cpp:184-186).
  uint8_t* dst = d_frame_ + (size_t)index * need;
  const size_t packed = (size_t)(h - 1) * stride + (size_t)3 * w;
  if (pinned_staging_) {
    if (need > h_frame_cap_) {
      VP_HIP_CHECK(hipStreamSynchronize(stream_));
      if (h_frame_) hipHostFree(h_frame_);
      h_frame_ = nullptr;
      VP_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&h_frame_), need * 2, hipHostMallocDefault));  // two slots: frame n+1 is staged while n flies
      h_frame_cap_ = need;
    }
    h_frame_slot_ ^= 1;
    uint8_t* slot = h_frame_ + (size_t)h_frame_slot_ * h_frame_cap_;
    // A pinned source makes the H2D below truly asynchronous: the copy that last read this slot (two uploads ago) may still be
    // queued behind earlier frames' graphs.  Its event orders this host write behind it.
    if (!h_frame_ev_[h_frame_slot_]) VP_HIP_CHECK(hipEventCreateWithFlags(&h_frame_ev_[h_frame_slot_], hipEventDisableTiming));
    else VP_HIP_CHECK(hipEventSynchronize(h_frame_ev_[h_frame_slot_]));
    if (stride == 3 * w) {
      std::memcpy(slot, frame, packed);
    } else {
      for (int y = 0; y < h; ++y) std::memcpy(slot + (size_t)y * stride, frame + (size_t)y * stride, (size_t)3 * w);
    }
    VP_HIP_CHECK(hipMemcpyAsync(dst, slot, packed, hipMemcpyHostToDevice, stream_));
    VP_HIP_CHECK(hipEventRecord(h_frame_ev_[h_frame_slot_], stream_));
  } else if (stride == 3 * w) {
    VP_HIP_CHECK(hipMemcpyAsync(dst, frame, packed, hipMemcpyHostToDevice, stream_));
  } else {
    VP_HIP_CHECK(hipMemcpy2DAsync(dst, stride, frame, stride, (size_t)3 * w, h, hipMemcpyHostToDevice, stream_));
  }
  if (input_is_tensor_) { graph_valid_ = false; ++plan_epoch_; }
  input_is_tensor_ = false;
}

void Engine::upload_tensor(const float* nchw) {
  if (base_) throw std::invalid_argument("shared engine: tensors go to the base engine");
  if (frames_ > 1) throw std::invalid_argument("batched encoder: frames only (vp_upload_frame_n)");
  if (!nchw) throw std::invalid_argument("null tensor");
  VP_HIP_CHECK(hipSetDevice(gpu_));
  VP_HIP_CHECK(hipMemcpyAsync(d_input_, nchw, (size_t)3 * net_h() * net_w() * sizeof(float), hipMemcpyHostToDevice, stream_));
  if (!input_is_tensor_) { graph_valid_ = false; ++plan_epoch_; }
  input_is_tensor_ = true;
}

void Engine::run_ops(hipStream_t st, size_t begin, size_t end) {
  for (size_t i = begin; i < end; ++i) {
    hipError_t e = ops_[i].run(st);
    if (e != hipSuccess) throw std::runtime_error("launch failed in layer '" + ops_[i].name + "': " + hipGetErrorString(e));
  }
}
void Engine::run_eager() { run_ops(stream_, input_is_tensor_ ? first_net_op_ : 0, ops_.size()); }

// One frame through this engine AND its shared-prefix heads as ONE graph launch on this engine's stream (so every stream-order
// guarantee of the separate vp_enqueue calls holds).  Inside the graph the heads that consume only the backbone (shared level 1:
// Scene3D, EgoLanes on a SceneSeg base) are forked onto side streams right behind the backbone and joined at the end: a single
// frame's two or three decoders overlap (the small-map neck layers and the 200-tile big layers leave CUs idle on their own).
// Same kernels, same arguments, same results as base.enqueue() followed by head.enqueue().
void Engine::enqueue_multi(const std::vector<Engine*>& heads) {
  VP_HIP_CHECK(hipSetDevice(gpu_));
  if (base_) throw std::invalid_argument("enqueue_multi: call it on the engine that owns the encoder");
  for (Engine* h : heads)
    if (!h || h->base_ != this || h->stream_ != stream_) throw std::invalid_argument("enqueue_multi: every head must be a shared-prefix engine of this engine");
  bool plain = !multi_fork_ || !use_graph_ || !warmed_ || kind_ == 4 || frames_ > 1 || n_fork_ops_ == 0 || heads.empty();
  for (Engine* h : heads) plain = plain || !h->warmed_ || !h->use_graph_;
  if (plain) {  // first frames (eager warm-up), graph replay switched off, or nothing to fork
    enqueue();
    for (Engine* h : heads) h->enqueue();
    return;
  }
  if (!input_is_tensor_ && !d_frame_) throw std::runtime_error("no frame resident: call vp_upload_frame / vp_infer first");
  std::vector<std::pair<const Engine*, unsigned long long>> key{{this, plan_epoch_}};
  for (Engine* h : heads) key.emplace_back(h, h->plan_epoch_);
  if (!multi_exec_ || key != multi_key_) {
    if (multi_exec_) hipGraphExecDestroy(multi_exec_);
    if (multi_graph_) hipGraphDestroy(multi_graph_);
    multi_exec_ = nullptr;
    multi_graph_ = nullptr;
    size_t n_side = 0;
    for (Engine* h : heads) n_side += h->shared_level_ == 1 ? 1 : 0;
    while (side_streams_.size() < n_side) {
      hipStream_t s = nullptr;
      VP_HIP_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
      side_streams_.push_back(s);
    }
    while (side_events_.size() < n_side + 1) {
      hipEvent_t e = nullptr;
      VP_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
      side_events_.push_back(e);
    }
    VP_HIP_CHECK(hipStreamBeginCapture(stream_, hipStreamCaptureModeThreadLocal));
    try {
      const size_t first = input_is_tensor_ ? first_net_op_ : 0;
      run_ops(stream_, first, n_fork_ops_);
      VP_HIP_CHECK(hipEventRecord(side_events_[0], stream_));
      size_t si = 0;
      for (Engine* h : heads)
        if (h->shared_level_ == 1) {
          VP_HIP_CHECK(hipStreamWaitEvent(side_streams_[si], side_events_[0], 0));
          h->run_ops(side_streams_[si], 0, h->ops_.size());
          VP_HIP_CHECK(hipEventRecord(side_events_[1 + si], side_streams_[si]));
          ++si;
        }
      run_ops(stream_, n_fork_ops_, ops_.size());
      for (Engine* h : heads)
        if (h->shared_level_ != 1) h->run_ops(stream_, 0, h->ops_.size());  // needs this engine's context + neck: behind them, in order
      for (size_t i = 0; i < si; ++i) VP_HIP_CHECK(hipStreamWaitEvent(stream_, side_events_[1 + i], 0));
    } catch (...) {
      hipGraph_t g = nullptr;
      hipStreamEndCapture(stream_, &g);
      if (g) hipGraphDestroy(g);
      throw;
    }
    VP_HIP_CHECK(hipStreamEndCapture(stream_, &multi_graph_));
    VP_HIP_CHECK(hipGraphInstantiate(&multi_exec_, multi_graph_, nullptr, nullptr, 0));
    multi_key_ = key;
  }
  VP_HIP_CHECK(hipGraphLaunch(multi_exec_, stream_));
  have_outputs_ = true;
  host_logits_valid_ = host_mask_valid_ = false;
  for (Engine* h : heads) {
    h->have_outputs_ = true;
    h->host_logits_valid_ = h->host_mask_valid_ = false;
  }
}

void Engine::capture_graph() {
  if (graph_exec_) {
    hipGraphExecDestroy(graph_exec_);
    graph_exec_ = nullptr;
  }
  if (graph_) {
    hipGraphDestroy(graph_);
    graph_ = nullptr;
  }
  VP_HIP_CHECK(hipStreamBeginCapture(stream_, hipStreamCaptureModeThreadLocal));
  try {
    run_eager();
  } catch (...) {
    hipGraph_t g = nullptr;
    hipStreamEndCapture(stream_, &g);
    if (g) hipGraphDestroy(g);
    throw;
  }
  VP_HIP_CHECK(hipStreamEndCapture(stream_, &graph_));
  VP_HIP_CHECK(hipGraphInstantiate(&graph_exec_, graph_, nullptr, nullptr, 0));
  graph_valid_ = true;
}

void Engine::enqueue() {
  VP_HIP_CHECK(hipSetDevice(gpu_));
  if (base_) {
    if (!base_->have_outputs_) throw std::runtime_error("shared engine: run the base engine on a frame first");
  } else if (!input_is_tensor_ && !d_frame_) {
    throw std::runtime_error("no frame resident: call vp_upload_frame / vp_infer first");
  }
  if (kind_ == 4 && !ad_primed_) prime_previous();  // first frame of a stream: previous := current
  if (!warmed_) {  // first pass is eager: sets kernel attributes and surfaces launch errors with layer names
    run_eager();
    VP_HIP_CHECK(hipStreamSynchronize(stream_));
    warmed_ = true;
    have_outputs_ = true;
    host_logits_valid_ = host_mask_valid_ = false;
    if (!use_graph_ || kind_ == 4) return;  // AutoDrive carries state (feature shift): a frame must run exactly once
  }
  if (use_graph_) {
    if (!graph_valid_) capture_graph();
    VP_HIP_CHECK(hipGraphLaunch(graph_exec_, stream_));
  } else {
    run_eager();
  }
  have_outputs_ = true;
  host_logits_valid_ = host_mask_valid_ = false;
}

void Engine::sync() {
  VP_HIP_CHECK(hipStreamSynchronize(stream_));
  check_status();
}

// The probe's verdict on the pass whose outputs were last fetched (enqueue_fetch copies the flag behind them).  Loud, once: the flag is
// cleared so that the next frame is judged on its own.
void Engine::check_status() {
  if (!status_pending_ || !h_status_) return;
  status_pending_ = false;
  if (*h_status_ == 0) return;
  *h_status_ = 0;
  VP_HIP_CHECK(hipMemsetAsync(d_status_, 0, sizeof(unsigned), stream_));
  throw RangeError("non-finite value (inf / NaN) in the network output: an activation left the fp16 range of the matrix pipe (|x| > 65504) "
                   "or the input / weights were not finite; outputs of this frame are invalid");
}

void Engine::fetch_outputs() {
  if (!d_logits_) throw std::runtime_error("this engine has no outputs (batched encoder: fetch from its shared-prefix engines)");
  enqueue_fetch();
  VP_HIP_CHECK(hipStreamSynchronize(stream_));
  check_status();
}

// D2H of the outputs the caller selected (vp_set_outputs), asynchronous on the engine stream, into pinned host memory.
void Engine::enqueue_fetch() {
  if (!d_logits_) throw std::runtime_error("this engine has no outputs (batched encoder: fetch from its shared-prefix engines)");
  if (outputs_ & 1)
    VP_HIP_CHECK(hipMemcpyAsync(h_logits_, d_logits_, (size_t)out_c_ * out_h_ * out_w_ * sizeof(float), hipMemcpyDeviceToHost, stream_));
  if ((outputs_ & 2) && d_mask_) VP_HIP_CHECK(hipMemcpyAsync(h_mask_, d_mask_, (size_t)out_h_ * out_w_, hipMemcpyDeviceToHost, stream_));
  host_logits_valid_ = (outputs_ & 1) != 0;
  host_mask_valid_ = (outputs_ & 2) != 0;
  if (finite_check_ && d_status_) {
    VP_HIP_CHECK(hipMemcpyAsync(h_status_, d_status_, sizeof(unsigned), hipMemcpyDeviceToHost, stream_));
    status_pending_ = true;
  }
}

// Lazy variants behind vp_logits / vp_mask_u8: an output de-selected with vp_set_outputs is fetched on first use.
const float* Engine::host_logits() {
  if (!host_logits_valid_ && d_logits_ && have_outputs_) {
    VP_HIP_CHECK(hipSetDevice(gpu_));
    VP_HIP_CHECK(hipMemcpyAsync(h_logits_, d_logits_, (size_t)out_c_ * out_h_ * out_w_ * sizeof(float), hipMemcpyDeviceToHost, stream_));
    VP_HIP_CHECK(hipStreamSynchronize(stream_));
    host_logits_valid_ = true;
  }
  return h_logits_;
}
const uint8_t* Engine::host_mask() {
  if (!host_mask_valid_ && d_mask_ && have_outputs_) {
    VP_HIP_CHECK(hipSetDevice(gpu_));
    VP_HIP_CHECK(hipMemcpyAsync(h_mask_, d_mask_, (size_t)out_h_ * out_w_, hipMemcpyDeviceToHost, stream_));
    VP_HIP_CHECK(hipStreamSynchronize(stream_));
    host_mask_valid_ = true;
  }
  return h_mask_;
}

void Engine::copy_outputs_device(void* logits_dst, void* mask_dst) {
  if (logits_dst)
    VP_HIP_CHECK(hipMemcpyAsync(logits_dst, d_logits_, (size_t)out_c_ * out_h_ * out_w_ * sizeof(float), hipMemcpyDeviceToDevice, stream_));
  if (mask_dst) VP_HIP_CHECK(hipMemcpyAsync(mask_dst, d_mask_, (size_t)out_h_ * out_w_, hipMemcpyDeviceToDevice, stream_));
}

void Engine::read_input_tensor(float* dst) {
  VP_HIP_CHECK(hipStreamSynchronize(stream_));
  VP_HIP_CHECK(hipMemcpy(dst, d_input_, (size_t)3 * net_h() * net_w() * sizeof(float), hipMemcpyDeviceToHost));
}

// OpenCV resizeNN index table (oracle/pre_post.py nearest_index)
static void nearest_tab(int src, int dst, int* tab) {
  const double inv = (double)dst / (double)src;
  const double ifx = 1.0 / inv;
  for (int d = 0; d < dst; ++d) tab[d] = std::min((int)std::floor(d * ifx), src - 1);
}
static void linear_taps_f32(int src, int dst, int* idx, float* wgt) {
  const double scale = (double)src / (double)dst;
  for (int d = 0; d < dst; ++d) {
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    int s = (int)std::floor(f);
    f -= (float)s;
    if (s < 0) {
      f = 0.0f;
      s = 0;
    }
    if (s >= src - 1) {
      f = 0.0f;
      s = src - 1;
    }
    idx[2 * d] = s;
    idx[2 * d + 1] = std::min(s + 1, src - 1);
    wgt[2 * d] = 1.0f - f;
    wgt[2 * d + 1] = f;
  }
}

void Engine::mask_resized(uint8_t* dst, int h, int w) {
  if (!have_outputs_) throw std::runtime_error("Inference has not been run yet");
  if (!dst || h < 1 || w < 1) throw std::invalid_argument("bad resize target");
  const size_t need = (size_t)h * w, tabn = (size_t)(h + w);
  if (need > resize_cap_) {
    d_resize_out_ = dalloc(std::max(need, (size_t)4 * h * w), false);
    resize_cap_ = std::max(need, (size_t)4 * h * w);
  }
  if (tabn * 4 > rs_tab_cap_) {
    d_rs_tab_ = static_cast<int*>(dalloc(tabn * 4 * sizeof(int), false));
    rs_tab_cap_ = tabn * 4;
  }
  std::vector<int> tab(h + w);
  nearest_tab(out_h_, h, tab.data());
  nearest_tab(out_w_, w, tab.data() + h);
  VP_HIP_CHECK(hipMemcpyAsync(d_rs_tab_, tab.data(), tab.size() * sizeof(int), hipMemcpyHostToDevice, stream_));
  VP_HIP_CHECK(launch_resize_nearest(d_mask_, out_w_, d_rs_tab_, d_rs_tab_ + h, h, w, static_cast<uint8_t*>(d_resize_out_), stream_));
  VP_HIP_CHECK(hipMemcpyAsync(dst, d_resize_out_, need, hipMemcpyDeviceToHost, stream_));
  VP_HIP_CHECK(hipStreamSynchronize(stream_));
}

// MasksVisualizationEngine::visualize on the device: the mask of the LAST inference, coloured, nearest-resized to the frame
// that produced it and blended 50/50 with that (still resident) frame; BGR8 out, frame size.
void Engine::visualize_mask(int viz_type, uint8_t* dst, int dst_h, int dst_w) {
  if (!have_outputs_) throw std::runtime_error("Inference has not been run yet");
  if (!dst || viz_type < 0 || viz_type > 2) throw std::invalid_argument("bad visualisation request");
  if (!d_frame_ || input_is_tensor_ || base_) throw std::runtime_error("visualize_mask needs the frame path (vp_infer) on a base engine");
  const int h = frame_h_, w = frame_w_;
  // The blend writes h*w*3 bytes: the caller's buffer must have the geometry of the frame that was inferred last (the
  // reference takes the size from original_image itself, masks_visualization_engine.cpp:19-27, so it cannot mismatch).
  if (dst_h != h || dst_w != w)
    throw std::invalid_argument("visualize_mask: destination is " + std::to_string(dst_w) + "x" + std::to_string(dst_h) +
                                " but the last inferred frame was " + std::to_string(w) + "x" + std::to_string(h));
  if (!d_viz_lut_) {
    // createColorMask (masks_visualization_engine.cpp:41-58), BGR
    std::vector<uint8_t> lut(3 * 256 * 3, 0);
    for (int v = 1; v < 256; ++v) { lut[(0 * 256 + v) * 3 + 2] = 255; }                          // "scene": 1..255 -> (0,0,255)
    const uint8_t dom0[3] = {255, 93, 61}, dom255[3] = {145, 28, 255};                            // "domain"
    for (int c = 0; c < 3; ++c) { lut[(1 * 256 + 0) * 3 + c] = dom0[c]; lut[(1 * 256 + 255) * 3 + c] = dom255[c]; }
    const uint8_t ego[3][3] = {{255, 0, 0}, {255, 0, 200}, {0, 153, 0}};                          // "egolanes": labels 0,1,2
    for (int v = 0; v < 3; ++v)
      for (int c = 0; c < 3; ++c) lut[(2 * 256 + v) * 3 + c] = ego[v][c];
    d_viz_lut_ = dupload(lut);
  }
  const size_t need = (size_t)3 * h * w, tabn = (size_t)(h + w);
  if (need > resize_cap_) {
    d_resize_out_ = dalloc(std::max(need, (size_t)4 * h * w), false);
    resize_cap_ = std::max(need, (size_t)4 * h * w);
  }
  if (tabn * 4 > rs_tab_cap_) {
    d_rs_tab_ = static_cast<int*>(dalloc(tabn * 4 * sizeof(int), false));
    rs_tab_cap_ = tabn * 4;
  }
  std::vector<int> tab(h + w);
  nearest_tab(out_h_, h, tab.data());
  nearest_tab(out_w_, w, tab.data() + h);
  VP_HIP_CHECK(hipMemcpyAsync(d_rs_tab_, tab.data(), tab.size() * sizeof(int), hipMemcpyHostToDevice, stream_));
  VP_HIP_CHECK(launch_viz_blend(d_mask_, out_w_, d_rs_tab_, d_rs_tab_ + h, d_frame_, frame_stride_, h, w, d_viz_lut_ + (size_t)viz_type * 768,
                                pixel_format_ == 1, static_cast<uint8_t*>(d_resize_out_), stream_));
  VP_HIP_CHECK(hipMemcpyAsync(dst, d_resize_out_, need, hipMemcpyDeviceToHost, stream_));
  VP_HIP_CHECK(hipStreamSynchronize(stream_));
}

void Engine::resize_depth_on_device(int h, int w) {
  const size_t need = (size_t)4 * h * w, tabn = (size_t)4 * (h + w);
  if (need > resize_cap_) {
    d_resize_out_ = dalloc(need, false);
    resize_cap_ = need;
  }
  if (tabn > rs_tab_cap_) {
    d_rs_tab_ = static_cast<int*>(dalloc(tabn * sizeof(int), false));
    rs_tab_cap_ = tabn;
  }
  // host tap tables are members: they must outlive the asynchronous copies (every caller syncs the stream before returning)
  std::vector<int>& idx = rs_idx_host_;
  std::vector<float>& wgt = rs_wgt_host_;
  idx.assign(2 * (h + w), 0);
  wgt.assign(2 * (h + w), 0.f);
  linear_taps_f32(out_h_, h, idx.data(), wgt.data());
  linear_taps_f32(out_w_, w, idx.data() + 2 * h, wgt.data() + 2 * h);
  int* d_idx = d_rs_tab_;
  float* d_wgt = reinterpret_cast<float*>(d_rs_tab_ + 2 * (h + w));
  VP_HIP_CHECK(hipMemcpyAsync(d_idx, idx.data(), idx.size() * sizeof(int), hipMemcpyHostToDevice, stream_));
  VP_HIP_CHECK(hipMemcpyAsync(d_wgt, wgt.data(), wgt.size() * sizeof(float), hipMemcpyHostToDevice, stream_));
  VP_HIP_CHECK(launch_resize_bilinear_f32(d_logits_, out_w_, d_idx, d_wgt, d_idx + 2 * h, d_wgt + 2 * h, h, w,
                                          static_cast<float*>(d_resize_out_), stream_));
}

void Engine::depth_resized(float* dst, int h, int w) {
  if (!have_outputs_) throw std::runtime_error("Inference has not been run yet");
  if (!dst || h < 1 || w < 1) throw std::invalid_argument("bad resize target");
  resize_depth_on_device(h, w);
  VP_HIP_CHECK(hipMemcpyAsync(dst, d_resize_out_, (size_t)4 * h * w, hipMemcpyDeviceToHost, stream_));
  VP_HIP_CHECK(hipStreamSynchronize(stream_));
}

// DepthVisualizationEngine::visualize (depth_visualization_engine.cpp:9-26) on the device: plane 0 of the logits,
// bilinear-resized to h x w (what the depth topic carries, run_model_node.cpp:100-104), min-max normalised to u8 and
// mapped through COLORMAP_VIRIDIS; BGR8 out.
void Engine::visualize_depth(uint8_t* dst, int h, int w) {
  if (!have_outputs_) throw std::runtime_error("Inference has not been run yet");
  if (!dst || h < 1 || w < 1) throw std::invalid_argument("bad visualisation target");
  static const uint8_t kViridisBgr[256 * 3] = {
#include "viridis_lut.inc"
  };
  if (!d_viridis_) {
    d_viridis_ = dupload(std::vector<uint8_t>(kViridisBgr, kViridisBgr + sizeof(kViridisBgr)));
    d_minmax_ = static_cast<unsigned*>(dalloc(2 * sizeof(unsigned), false));
  }
  const size_t n = (size_t)h * w;
  if (3 * n > depth_viz_cap_) {
    d_depth_viz_ = dalloc(3 * n, false);
    depth_viz_cap_ = 3 * n;
  }
  resize_depth_on_device(h, w);
  static const unsigned kInit[2] = {0xFFFFFFFFu, 0u};
  VP_HIP_CHECK(hipMemcpyAsync(d_minmax_, kInit, sizeof(kInit), hipMemcpyHostToDevice, stream_));
  VP_HIP_CHECK(launch_minmax_f32(static_cast<const float*>(d_resize_out_), n, d_minmax_, stream_));
  VP_HIP_CHECK(launch_depth_colorize(static_cast<const float*>(d_resize_out_), n, d_minmax_, d_viridis_, static_cast<uint8_t*>(d_depth_viz_), stream_));
  VP_HIP_CHECK(hipMemcpyAsync(dst, d_depth_viz_, 3 * n, hipMemcpyDeviceToHost, stream_));
  VP_HIP_CHECK(hipStreamSynchronize(stream_));
}

// -------------------------------------------------------------------------------------------------- timing
void Engine::timer_begin() { VP_HIP_CHECK(hipEventRecord(ev0_, stream_)); }
float Engine::timer_end() {
  VP_HIP_CHECK(hipEventRecord(ev1_, stream_));
  VP_HIP_CHECK(hipEventSynchronize(ev1_));
  float ms = 0.f;
  VP_HIP_CHECK(hipEventElapsedTime(&ms, ev0_, ev1_));
  return ms;
}

int Engine::profile_layers(int iters, float* ms, int cap) {
  const size_t first = input_is_tensor_ ? first_net_op_ : 0;
  const int n = (int)ops_.size();
  if (cap < n) throw std::invalid_argument("profile buffer too small");
  if (base_) {
    if (!base_->have_outputs_) throw std::runtime_error("shared engine: run the base engine on a frame first");
  } else if (!input_is_tensor_ && !d_frame_) {
    throw std::runtime_error("no frame resident");
  }
  std::vector<hipEvent_t> ev(n + 1);
  for (auto& e : ev) VP_HIP_CHECK(hipEventCreate(&e));
  std::vector<double> acc(n, 0.0);
  for (int it = 0; it < iters + 1; ++it) {  // iteration 0 is a warm-up
    for (int i = (int)first; i < n; ++i) {
      VP_HIP_CHECK(hipEventRecord(ev[i], stream_));
      hipError_t e = ops_[i].run(stream_);
      if (e != hipSuccess) throw std::runtime_error("launch failed in layer '" + ops_[i].name + "'");
    }
    VP_HIP_CHECK(hipEventRecord(ev[n], stream_));
    VP_HIP_CHECK(hipStreamSynchronize(stream_));
    if (it == 0) continue;
    for (int i = (int)first; i < n; ++i) {
      float t = 0.f;
      VP_HIP_CHECK(hipEventElapsedTime(&t, ev[i], ev[i + 1]));
      acc[i] += t;
    }
  }
  for (int i = 0; i < n; ++i) ms[i] = (float)(acc[i] / std::max(1, iters));
  for (auto& e : ev) hipEventDestroy(e);
  warmed_ = true;
  have_outputs_ = true;
  return n;
}

void Engine::read_act(int i, float* dst) {
  if (i < 0 || i >= (int)acts_.size()) throw std::invalid_argument("tensor index out of range");
  const Act& a = *acts_[i];
  const size_t n = (size_t)a.Creal * a.H * a.W;
  float* d = nullptr;
  VP_HIP_CHECK(hipStreamSynchronize(stream_));
  VP_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&d), n * sizeof(float)));
  hipError_t e = launch_act_to_nchw(a.view(), a.Creal, d, stream_);
  if (e == hipSuccess) e = hipStreamSynchronize(stream_);
  if (e == hipSuccess) e = hipMemcpy(dst, d, n * sizeof(float), hipMemcpyDeviceToHost);
  hipFree(d);
  VP_HIP_CHECK(e);
}

}  // namespace vp
