// libvp_hip engine, part 1 of 3: weight container, BatchNorm folding, engine construction / release and the per-network PLAN
// (backbone, context, neck, heads, AutoDrive).  Part 2 = engine_dispatch.cpp (which kernel a layer gets, weight packing), part 3 =
// engine_io.cpp (frames in, graph replay, outputs out).
#include "engine_internal.hpp"

namespace vp {

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

// Experiment knob VP_WLO_KEEP_BITS (round 6; read once per engine in Engine::construct): explicit mantissa bits the LOW planes of
// the weights keep (10 = all; fewer toggling bits through a power-limited matrix pipe: profiles/r06_mfma_power_lo_trunc.txt).  Default: untouched.
std::atomic<int> g_wlo_keep{10};
half_t truncate_lo(half_t lo) {
  const int keep = g_wlo_keep.load(std::memory_order_relaxed);
  if (keep >= 10) return lo;
  unsigned short bits;
  std::memcpy(&bits, &lo, 2);
  bits &= (unsigned short)~((1u << (10 - std::max(0, keep))) - 1u);
  std::memcpy(&lo, &bits, 2);
  return lo;
}

void split_half(float v, float pre, half_t* hi, half_t* lo) {
  // Both precision modes carry weights on fp16 planes: a folded weight beyond the fp16 range (or non-finite) is refused at load (the
  // prescale could carry it, the activations it produces would not survive).  Small weights: see prescale_exp (engine_internal.hpp).
  if (!(std::fabs(v) <= 65504.0f)) throw RangeError("weight " + std::to_string(v) + " is outside the fp16 range the matrix pipe carries (|w| <= 65504): re-scale the checkpoint");
  const float x = v * pre;
  const half_t h = (half_t)x;
  *hi = h;
  *lo = truncate_lo((half_t)(x - (float)h));
}

uint8_t e4m3_encode(float q) {
  const uint8_t sign = std::signbit(q) ? 0x80 : 0x00;
  const float mag = std::fabs(q);
  if (!(mag > 0.0f)) return sign;
  if (mag >= 448.0f) return sign | 0x7E;                       // largest finite code (0x7F is NaN)
  int e = 0;
  std::frexp(mag, &e);                                         // mag = f * 2^e, f in [0.5, 1)  ->  exponent of the leading bit: e - 1
  int E = e - 1;
  if (E < -6) {                                                // subnormal range: multiples of 2^-9
    const int m = (int)std::nearbyint(std::ldexp(mag, 9));     // 0 .. 8
    return sign | (uint8_t)(m >= 8 ? 0x08 : m);                // 8 * 2^-9 = 2^-6: the smallest normal
  }
  int m = (int)std::nearbyint(std::ldexp(mag, 3 - E)) - 8;     // mantissa 0 .. 8
  if (m == 8) {
    m = 0;
    ++E;
  }
  const int code = ((E + 7) << 3) | m;
  return sign | (uint8_t)(code > 0x7E ? 0x7E : code);
}

int prescale_exp(float amax) {
  if (!(amax > 0.0f) || !std::isfinite(amax)) return 0;
  int e = 0;
  std::frexp(amax, &e);  // amax = m * 2^e, m in [0.5, 1)  =>  amax * 2^(14 - e) in [2^13, 2^14)
  return std::max(-8, std::min(14 - e, 60));
}

RowScale row_prescale(const float* w, size_t rows, size_t per, size_t rows_alloc) {
  RowScale rs;
  rs.pre.assign(rows, 1.0f);
  rs.post.assign(std::max(rows, rows_alloc), 1.0f);
  for (size_t r = 0; r < rows; ++r) {
    float amax = 0.0f;
    for (size_t i = 0; i < per; ++i) {
      const float a = std::fabs(w[r * per + i]);
      if (a > amax) amax = a;   // NaN compares false: split_half reports it
    }
    const int s = prescale_exp(amax);
    rs.pre[r] = std::ldexp(1.0f, s);
    rs.post[r] = std::ldexp(1.0f, -s);
  }
  return rs;
}

// ==================================================================================================== Engine
Engine::Engine(int kind, const WeightBlob* blob, int precision, int gpu_id, Engine* base, int frames, int frame_index)
    : kind_(kind), precision_(precision), gpu_(gpu_id), frames_(frames), frame_index_(frame_index), base_(base) {
  if (frames < 1 || frames > 16) throw std::invalid_argument("frames must be 1..16");
  if (frames > 1 && (base || kind < 0 || kind > 3)) throw std::invalid_argument("a batched encoder is a base engine of a scene network kind");
  if (base && (frame_index < 0 || frame_index >= base->frames_)) throw std::invalid_argument("frame_index out of the base engine's range");
  if (!base && frame_index != 0) throw std::invalid_argument("frame_index needs a batched base engine");
  if ((precision & 15) > 1 || (precision & ~(1 | 16 | 32)) != 0) throw std::invalid_argument("precision must be VP_FP16 or VP_FP16X3 (optionally | VP_WEIGHTS_FP8 | VP_PLAN_LATENCY)");
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
  {
    const char* e = dev_option("VP_WLO_KEEP_BITS");
    g_wlo_keep.store(e ? std::atoi(e) : 10, std::memory_order_relaxed);
  }
  if (base) {
    if (kind < 0 || base->kind_ < 0) throw std::invalid_argument("shared engines need model kinds on both sides");
    if (base->base_) throw std::invalid_argument("the base of a shared engine must own its whole network");
    if ((base->precision_ & ~32) != (precision & ~32) || base->gpu_ != gpu_id)   // (the plan target is per engine: a head may differ from its base)
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
    // the values go in as 64-bit words (two floats per multiply): byte-wise FNV over the ~150 MB of a scene network was 0.2 s of every vp_create.
    // An in-process identity only (compared between engines of one process, never stored)
    const float* v = it->second.data.data();
    const size_t nv = it->second.data.size();
    size_t i = 0;
    for (; i + 2 <= nv; i += 2) {
      unsigned long long w;
      std::memcpy(&w, v + i, 8);
      h ^= w;
      h *= 1099511628211ull;
      h ^= h >> 29;
    }
    if (i < nv) mix(v + i, sizeof(float));
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
  if (zero) fill_zero(p, std::max<size_t>(bytes, 256));
  return p;
}
// The engine's OWN (non-blocking) stream, never a private one: an extra stream per engine shifted the runtime's stream -> hardware-queue
// assignment and cost the several-cameras rate 13 % (409 -> 354 frames/s, measured the same afternoon) -- compute streams of different cameras
// ended up sharing a queue.  These calls come from construction, from the frame-geometry tables (before the pass is enqueued) and from
// read-backs behind a synchronise: the stream is never capturing then.
void Engine::copy_h2d(void* d, const void* h, size_t bytes) {
  if (!bytes) return;
  VP_HIP_CHECK(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, stream_));
  VP_HIP_CHECK(hipStreamSynchronize(stream_));
}
void Engine::copy_d2h(void* h, const void* d, size_t bytes) {
  if (!bytes) return;
  VP_HIP_CHECK(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, stream_));
  VP_HIP_CHECK(hipStreamSynchronize(stream_));
}
void Engine::fill_zero(void* d, size_t bytes) {
  if (!bytes) return;
  VP_HIP_CHECK(hipMemsetAsync(d, 0, bytes, stream_));
  VP_HIP_CHECK(hipStreamSynchronize(stream_));
}
void Engine::dfree(void* p) {
  if (!p) return;
  auto it = std::find(allocs_.begin(), allocs_.end(), p);
  if (it != allocs_.end()) allocs_.erase(it);
  hipFree(p);
}
void Engine::upload_fc_weights(FcParams* fp, const std::vector<float>& w, int N, int K, const std::vector<float>* row_amax) {
  if (w.size() != (size_t)N * K) throw std::runtime_error("FC weight size mismatch");
  if (!fp8_storage() || (K & 3) != 0) {
    fp->w = dupload(w);
    wbytes_[2] += 4 * w.size();
    return;
  }
  std::vector<uint8_t> codes((size_t)N * K);
  std::vector<float> scale(N);
  parallel_rows(N, [&](int n_begin, int n_end) {
    for (int n = n_begin; n < n_end; ++n) {
      float amax = row_amax ? (*row_amax)[n] : 0.0f;
      if (!row_amax)
        for (int k = 0; k < K; ++k) amax = std::max(amax, std::fabs(w[(size_t)n * K + k]));
      scale[n] = fp8_row_scale(amax);
      for (int k = 0; k < K; ++k) codes[(size_t)n * K + k] = e4m3_encode(w[(size_t)n * K + k] / scale[n]);
    }
  });
  fp->w8 = dupload(codes);
  fp->wscale8 = dupload(scale);
  wbytes_[0] += codes.size();
}
const void* Engine::zero_page() {
  if (!d_zero_) d_zero_ = dalloc(256, true);
  return d_zero_;
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
  copy_h2d(d, chw, (size_t)a->Creal * a->H * a->W * sizeof(float));
  VP_HIP_CHECK(launch_nchw_to_act(d, a->Creal, a->view(), stream_));
  VP_HIP_CHECK(hipStreamSynchronize(stream_));
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
  const size_t se_words = (size_t)N * ((size_t)kSeBlocks * 1536 * 8 + (size_t)6 * 256 * kSeMaxReplicas) + (size_t)kSeBlocks * 64 * kSeMaxReplicas;  // >= sum of N*replicas*C (+ replicas*64 squeeze sums) below (checked)
  unsigned long long* se_arena = static_cast<unsigned long long*>(dalloc(se_words * sizeof(unsigned long long)));
  int se_block = 0;
  size_t se_used = 0;
  // the accumulators are zeroed once per frame: by the stem kernel when it is ONE launch (one frame per pass), else by a launch of its own
  const bool zero_in_stem = N == 1 && se_words % 2 == 0;
  if (!zero_in_stem) {
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
    if (zero_in_stem) {
      sp.zero = se_arena;
      sp.zero_n = se_words;
    }
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
      // one frame per pass (both precisions since round 4): expand 1x1 -> depthwise (+ pool) as ONE launch, the expanded tensor never leaves the CU
      // (kernels_mbconv.hip).  VP_MBCONV_FUSE=0 (developer knob, A/B timing): the two launches.
      const char* env_mb = dev_option("VP_MBCONV_FUSE");
      const bool fuse_front = S.e != 1 && N == 1 && !(env_mb && env_mb[0] == '0');   // both precisions since round 4 (fp16: one plane, one MFMA per product)
      Folded f_exp;
      if (fuse_front) {
        f_exp = fold_conv_bn(blob, bp + std::to_string(j));
        ++j;
      } else if (S.e != 1) {
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
      // fused front + fused back: the squeeze FC travels with the pool sums ([se_rep][64] fixed-point numbers, same arena: zeroed per frame)
      const char* env_mbb0 = dev_option("VP_MBCONV_BACK");
      unsigned long long* zsums = nullptr;
      if (fuse_front && !(env_mbb0 && env_mbb0[0] == '0')) {
        if (se_used + (size_t)se_rep * 64 > se_words) throw std::runtime_error("SE arena too small");
        zsums = se_arena + se_used;
        se_used += (size_t)se_rep * 64;
      }
      const float* d_se_w1 = nullptr;   // squeeze FC matrix, uploaded once (shared by the front and back halves)
      if (zsums) {
        const std::string sp = bp + std::to_string(j + 1);
        const HostTensor& w1 = blob.get(sp + ".fc1.weight");
        if (w1.shape[0] != sq || w1.shape[1] != cexp) throw std::runtime_error("SE fc1 shape mismatch: " + sp);
        std::vector<float> w1p((size_t)sq * z->C, 0.0f);
        for (int q = 0; q < sq; ++q)
          for (int c = 0; c < cexp; ++c) w1p[(size_t)q * z->C + c] = w1.data[(size_t)q * cexp + c];
        d_se_w1 = dupload(w1p);
      }
      {
        Folded f = fold_conv_bn(blob, bp + std::to_string(j));
        const int kk = S.k * S.k;
        std::vector<float> wk((size_t)kk * z->C, 0.0f), bk(z->C, 0.0f);
        for (int c = 0; c < cexp; ++c) {
          for (int t = 0; t < kk; ++t) wk[(size_t)t * z->C + c] = f.w[(size_t)c * kk + t];
          bk[c] = f.b[c];
        }
        if (fuse_front) {
          if (f_exp.cout != cexp || f_exp.cin != cin || f_exp.k != 1) throw std::runtime_error("expand conv shape mismatch: " + bp);
          std::vector<half_t> wh((size_t)z->C * x->C, (half_t)0.0f), wlo(wh.size(), (half_t)0.0f);
          std::vector<float> be(z->C, 0.0f);
          const RowScale rs = row_prescale(f_exp.w.data(), cexp, cin, z->C);
          for (int co = 0; co < cexp; ++co) {
            for (int ci = 0; ci < cin; ++ci) split_half(f_exp.w[(size_t)co * cin + ci], rs.pre[co], &wh[(size_t)co * x->C + ci], &wlo[(size_t)co * x->C + ci]);
            be[co] = f_exp.b[co];
          }
          MbFrontParams mp{};
          mp.in = x->view();
          mp.w_hi = dupload(wh);
          mp.w_lo = split() ? dupload(wlo) : nullptr;
          wbytes_[1] += 2 * wh.size() * (split() ? 2 : 1);
          wbytes_[2] += 4 * wk.size();
          mp.b_exp = dupload(be);
          mp.s_exp = dupload(rs.post);
          mp.w_dw = dupload(wk);
          mp.b_dw = dupload(bk);
          mp.out = z->view();
          mp.k = S.k;
          mp.stride = stride;
          mp.sums = zsums ? nullptr : sums;   // the back half starts from the squeeze sums: the per-channel sums are not needed
          mp.replicas = se_rep;
          mp.w1 = d_se_w1;
          mp.sq = sq;
          mp.zsums = zsums;
          if (!mbconv_front_supported(mp)) throw std::runtime_error("fused MBConv front: unsupported shape: " + bp);
          Op op;
          op.name = bp + "0+" + std::to_string(j);   // expand + depthwise
          op.flops = 2.0 * cexp * cin * x->H * x->W + 2.0 * kk * cexp * z->H * z->W;
          op.bytes = (split() ? 4.0 : 2.0) * (x->elems() + z->elems());
          op.kernel = std::string("mbconv_front<k") + std::to_string(S.k) + ",s" + std::to_string(stride) + (split() ? ">" : ",x1>");
          op.run = [mp](hipStream_t st) { return launch_mbconv_front(mp, st); };
          ops_.push_back(std::move(op));
          ++j;
        } else {
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
      }
      // squeeze-excite -> per-frame scaled projection weights
      // one frame per pass (both precisions since round 4): squeeze-excite tail + projection (+ residual) as ONE launch (kernels_mbconv.hip, back half).
      // VP_MBCONV_BACK=0 (developer knob, A/B timing): se_gate_scale + the projection GEMM (+ its split-K finish).
      const char* env_mbb = dev_option("VP_MBCONV_BACK");
      const bool fuse_back = N == 1 && !(env_mbb && env_mbb[0] == '0');   // both precisions since round 4
      SeParams se{};
      const float *se_w2 = nullptr, *se_b2 = nullptr;
      std::vector<float> se_w2_host;  // [C][sq]
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
        se.w1 = d_se_w1 ? d_se_w1 : dupload(w1p);
        se.b1 = dupload(b1.data);
        if (fuse_back) se_w2_host = w2p;
        else se_w2 = dupload(w2p);
        se_b2 = dupload(b2p);
        se.frames = N;
        se_name = sp;
        ++j;
      }
      // project 1x1 (+BN folded) with SE scale folded into K, optional residual
      if (fuse_back) {
        Folded f = fold_conv_bn(blob, bp + std::to_string(j));
        if (f.cout != S.cout || f.cin != cexp || f.k != 1) throw std::runtime_error("projection conv shape mismatch: " + bp);
        const bool residual = (stride == 1 && cin == S.cout);
        Act* out = new_act(bp + std::to_string(j), S.cout, z->H, z->W);
        out->frames = N;
        std::vector<float> wf((size_t)out->C * z->C, 0.0f), bias(out->C, 0.0f);
        const RowScale rs = row_prescale(f.w.data(), S.cout, cexp, out->C);   // the gate (0, 1) multiplies on the device: the row maximum only shrinks
        for (int co = 0; co < S.cout; ++co) {
          for (int c = 0; c < cexp; ++c) wf[(size_t)co * z->C + c] = f.w[(size_t)co * cexp + c] * rs.pre[co];
          bias[co] = f.b[co];
        }
        for (float v : f.w)
          if (!(std::fabs(v) <= 65504.0f)) throw RangeError("projection weight " + std::to_string(v) + " of " + bp + " is outside the fp16 range the matrix pipe carries (|w| <= 65504): re-scale the checkpoint");
        const int sqp = round_up(sq, 4);
        std::vector<float> w2q((size_t)sqp * z->C, 0.0f);
        for (int c = 0; c < z->C; ++c)
          for (int q = 0; q < sq; ++q) w2q[((size_t)(q >> 2) * z->C + c) * 4 + (q & 3)] = se_w2_host[(size_t)c * sq + q];
        MbBackParams mb{};
        mb.in = z->view();
        mb.se = se;
        mb.se.frames = 1;
        mb.zsums = zsums;
        mb.w2q = dupload(w2q);
        mb.b2 = se_b2;
        mb.sqp = sqp;
        mb.w = dupload(wf);
        wbytes_[2] += 4 * (wf.size() + w2q.size());
        mb.bias = dupload(bias);
        mb.wscale = dupload(rs.post);
        if (residual) mb.res = x->view();
        mb.out = out->view();
        if (!mbconv_back_supported(mb)) throw std::runtime_error("fused MBConv back: unsupported shape: " + bp);
        Op op;
        op.name = bp + std::to_string(j - 1) + "+" + std::to_string(j);   // squeeze-excite + projection
        op.flops = 2.0 * cexp * S.cout * z->H * z->W + 4.0 * sq * cexp;
        op.bytes = (split() ? 4.0 : 2.0) * (z->elems() + out->elems() + (residual ? x->elems() : 0)) + 4.0 * wf.size();
        op.kernel = std::string("mbconv_back<wm") + (z->H * z->W >= 12800 ? "4" : "1") + (split() ? ">" : ",x1>");
        op.run = [mb](hipStream_t st) { return launch_mbconv_back(mb, st); };
        ops_.push_back(std::move(op));
        x = out;
      } else {
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
        const RowScale rs = row_prescale(f.w.data(), S.cout, cexp, pc.CoutW);   // se_gate_scale multiplies by the gate (0, 1) and splits on the device
        for (int co = 0; co < S.cout; ++co) {
          for (int c = 0; c < cexp; ++c) wf[(size_t)co * z->C + c] = f.w[(size_t)co * cexp + c] * rs.pre[co];
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
        pc.wscale = dupload(rs.post);
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
  // round 5: context_layer_2 (the matvec that builds the 10x20 map) and context_layer_3 (the 3x3 convolution that reads it) in ONE launch
  // (kernels_misc.hip ctx_exp_conv1_kernel); VP_CTX_FUSE=0 (developer knob, A/B timing): two launches as before
  const bool fuse23 = !dev_option_is("VP_CTX_FUSE", '0') && widths[2] == HW;
  FcParams fp2{};
  for (int i = 0; i < 3; ++i) {
    const std::string lp = p + "context_layer_" + std::to_string(i);
    const HostTensor& w = blob.get(lp + ".weight");
    const HostTensor& b = blob.get(lp + ".bias");
    if (w.shape[0] != widths[i] || w.shape[1] != K) throw std::runtime_error("context MLP shape mismatch: " + lp);
    FcParams fp{};
    fp.x = x;
    upload_fc_weights(&fp, w.data, widths[i], K);
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
    if (i == 2 && fuse23) {
      fp2 = fp;
      break;
    }
    Op op;
    op.name = lp;
    op.flops = 2.0 * widths[i] * K;
    op.bytes = (fp.w8 ? 1.0 : 4.0) * widths[i] * K;
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
    op.flops = 2.0 * 9 * 128 * HW;
    // 2x2 patches: 50 workgroups on the 10x20 map, the 16 map rows under a patch in ONE pass of 16 lanes per 800-element row (13 independent loads per
    // lane).  (First cut: 8x8 patches, a wave per row = 25 dependent passes per workgroup: 69 us against 21 for the two launches it replaced.)
    CtxExpConv1Params q{fp2, cp, 2, 16};
    if (fuse23 && ctx_exp_conv1_ok(q)) {
      op.name = p + "context_layer_2+3";
      op.kernel = "ctx_exp_conv1<t2>";
      op.flops += 2.0 * fp2.N * fp2.K;
      op.bytes = (fp2.w8 ? 1.0 : 4.0) * fp2.N * fp2.K;
      op.run = [q](hipStream_t st) { return launch_ctx_exp_conv1(q, st); };
    } else {
      if (fuse23) {   // the shape does not fit the fused kernel: the matvec on its own after all
        Op o2;
        o2.name = p + "context_layer_2";
        o2.flops = 2.0 * fp2.N * fp2.K;
        o2.bytes = (fp2.w8 ? 1.0 : 4.0) * fp2.N * fp2.K;
        const FcParams fpc = fp2;
        o2.run = [fpc](hipStream_t st) { return launch_fc(fpc, st); };
        ops_.push_back(std::move(o2));
        cp.map = fp2.out;
      }
      op.name = p + "context_layer_3";
      op.run = [cp](hipStream_t st) { return launch_ctx_conv1(cp, st); };
    }
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
    // round 6: upsample + skip link + the first 3x3 of the block are one linear map -- composed at load into ONE launch (engine_upconv.cpp)
    const bool composed = upconv_wanted();
    if (composed) {
      const std::string dl = p + "decode_layer_" + std::to_string(2 * blk);
      x = add_upconv(up + "+skip_link_layer_" + std::to_string(blk) + "+decode_layer_" + std::to_string(2 * blk), x, feats[3 - blk],
                     blob.get(up + ".weight").data, blob.get(up + ".bias").data, blob.get(sk + ".weight").data, blob.get(sk + ".bias").data,
                     blob.get(dl + ".weight").data, blob.get(dl + ".bias").data, up_c[blk], d_c[2 * blk], ACT_GELU, -1, 0, dl);
    } else {
    x = add_convT_skip(up, sk, x, feats[3 - blk], blob.get(up + ".weight").data, blob.get(up + ".bias").data,
                       blob.get(sk + ".weight").data, blob.get(sk + ".bias").data, up_c[blk]);
    }
    for (int k = composed ? 1 : 0; k < 2; ++k) {
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
    const bool composed = upconv_wanted();   // round 6: (upsample + skip link | upsample) + the 3x3 behind it as ONE launch (engine_upconv.cpp)
    if (composed) {
      x = add_upconv(p + "upsample_layer_3+skip_link_layer_3+decode_layer_6", x, feats[0], W("upsample_layer_3"), B("upsample_layer_3"),
                     W("skip_link_layer_3"), B("skip_link_layer_3"), W("decode_layer_6"), B("decode_layer_6"), 256, 256, ACT_GELU, -1, 0, p + "decode_layer_6");
    } else {
    const Act* u = add_convT_skip(p + "upsample_layer_3", p + "skip_link_layer_3", x, feats[0], W("upsample_layer_3"),
                                  B("upsample_layer_3"), W("skip_link_layer_3"), B("skip_link_layer_3"), 256);
    x = add_conv(p + "decode_layer_6", u, W("decode_layer_6"), B("decode_layer_6"), 256, 3, gelu);
    }
    x = add_conv(p + "decode_layer_7", x, W("decode_layer_7"), B("decode_layer_7"), 128, 3, gelu);
    if (composed) {
      const std::vector<float> none;
      x = add_upconv(p + "upsample_layer_4+decode_layer_8", x, nullptr, W("upsample_layer_4"), B("upsample_layer_4"), none, none, W("decode_layer_8"),
                     B("decode_layer_8"), 128, 128, ACT_GELU, -1, 0, p + "decode_layer_8");
    } else {
    x = add_convT(p + "upsample_layer_4", x, W("upsample_layer_4"), B("upsample_layer_4"), 128, ConvOpts{});
    x = add_conv(p + "decode_layer_8", x, W("decode_layer_8"), B("decode_layer_8"), 128, 3, gelu);
    }
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
      pp.norm_form = norm_form_;
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
    fp.octets = fusion_octets_ok(fp) ? 1 : 0;
    Op op;
    op.name = "BackboneFeatureFusion";
    op.kernel = fp.octets ? "fusion<octets>" : "fusion";
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
  Act* head_cat = nullptr;
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
    pp.norm_form = norm_form_;
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
    // 64 pixels per slab (round 4; it was 512): on the 16x32 map of p5 ONE workgroup walked the 512 pixels, 16 dependent round trips = 23 us of a
    // 0.66 ms frame; the FC behind it sums the slabs in their fixed order either way
    const int nslab = std::max(1, std::min(64, HW / 64));
    float* partial = static_cast<float*>(dalloc((size_t)nslab * a->C * sizeof(float)));
    {
      PoolParams pp{a->view(), partial, nslab};
      push(cp + ".mean", "pool_partial", [pp](hipStream_t st) { return launch_pool_partial(pp, st); });
    }
    // exp0: Conv1d(C -> H*W, k3, pad 1) on a length-1 sequence == the centre tap as a [H*W][C] matrix (:210), SiLU twice (:211-213)
    const HostTensor& ew = T(cp + ".exp0.weight");
    const HostTensor& eb = T(cp + ".exp0.bias");
    if (ew.shape.size() != 3 || ew.shape[0] != HW || ew.shape[1] != C || ew.shape[2] != 3) throw std::runtime_error("exp0 shape mismatch: " + cp);
    std::vector<float> wm((size_t)HW * a->C, 0.0f), row_amax(HW, 0.0f);
    for (int n = 0; n < HW; ++n)
      for (int c = 0; c < C; ++c) {
        wm[(size_t)n * a->C + c] = ew.data[((size_t)n * C + c) * 3 + 1];
        for (int t = 0; t < 3; ++t) row_amax[n] = std::max(row_amax[n], std::fabs(ew.data[((size_t)n * C + c) * 3 + t]));   // the quantiser's row: all three taps
      }
    FcParams fp{};
    upload_fc_weights(&fp, wm, HW, a->C, &row_amax);
    fp.b = dupload(eb.data);
    fp.N = HW;
    fp.K = a->C;
    fp.act = ACT_SILU2;
    fp.out = static_cast<float*>(dalloc((size_t)HW * sizeof(float)));
    fp.partial = partial;
    fp.nslab = nslab;
    fp.Kstride = a->C;
    fp.inv_hw = 1.0f / (float)HW;
    fp.rows_kernel = fc_rows_ok(fp) ? 1 : 0;   // p2 / p3: 32768 x 32 and 8192 x 64 -- a thread per row
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
      // round 5: exp0 (the matvec that builds the H x W map) and ctx0 (the convolution that reads it) CAN run as one launch (kernels_misc.hip
      // ctx_exp_conv1_kernel: 16x16 patches on the large maps, 8x8 from 2048 pixels down) -- built for VERDICT round 4 item 7 and MEASURED SLOWER here
      // (gpurun r5c06, fp16, us per launch pair -> fused): p2 19.0 -> 23.2, p3 17.6 -> 22.2, p4 16.7 -> 15.6, p5 14.9 -> 19.3; frame p50 0.452 -> 0.479 ms
      // with four launches fewer: the two launches spread the matvec over 32-128 workgroups, a patch's workgroup walks its 100-324 rows alone.
      // Selected only by VP_CTX_FUSE=1; the scene networks' context_layer_2 + 3 (200 rows, 2x2 patches, 50 workgroups) keep the fused form (21 -> 17 us).
      // lanes per row: as few as keep every map row under a patch in one pass (16x16: 324 rows, one lane each in two passes; 8x8: 100 rows, two lanes each)
      CtxExpConv1Params q{fp, cpp, HW > 2048 ? 16 : 8, HW > 2048 ? 1 : 2};
      if (dev_option_is("VP_CTX_FUSE", '1') && ctx_exp_conv1_ok(q)) {
        push(cp + ".exp0+ctx0", q.tile == 16 ? "ctx_exp_conv1<t16>" : "ctx_exp_conv1<t8>", [q](hipStream_t st) { return launch_ctx_exp_conv1(q, st); },
             2.0 * HW * C + 2.0 * 9 * c0n * HW, (fp.w8 ? 1.0 : 4.0) * HW * C);
      } else {
        push(cp + ".exp0", fp.rows_kernel ? "fc<rows>" : "fc", [fp](hipStream_t st) { return launch_fc(fp, st); }, 2.0 * HW * C, (fp.w8 ? 1.0 : 4.0) * HW * C);
        push(cp + ".ctx0", "ctx_conv1", [cpp](hipStream_t st) { return launch_ctx_conv1(cpp, st); }, 2.0 * 9 * c0n * HW);
      }
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
    // cat((x, mp5(x), mp5(mp5(x)), mp5^3(x))) in one launch (round 4: a slice copy + three chained max-pool launches before; kernels_autodrive.hip)
    if (sppf_pool_ok(c1v, cv, c_)) {
      push(sp + ".pyramid", "sppf_pool", [=](hipStream_t st) { return launch_sppf_pool(c1v, cv, c_, st); });
    } else {   // maps above 512 pixels (other input sizes): the slice copy and the three chained 5x5 max-pools
      push(sp + ".cat0", "chan_copy", [=](hipStream_t st) { return launch_chan_copy(c1v, 0, cv, 0, c_, st); });
      for (int i = 0; i < 3; ++i)
        push(sp + ".maxpool" + std::to_string(i), "maxpool5", [=](hipStream_t st) { return launch_maxpool5(cv, i * c_, cv, (i + 1) * c_, c_, st); });
    }
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
    {  // round 4: four query tokens per workgroup (kernels_autodrive.hip attention_block_kernel); VP_ATTN_BLOCK=0 (developer knob, A/B timing): one per workgroup
      const char* ab = dev_option("VP_ATTN_BLOCK");
      ap.qblock = (attention_block_ok(ap) && !(ab && ab[0] == '0')) ? 4 : 0;
    }
    push(mb + ".conv1.attention", ap.qblock ? "attention<q4>" : "attention", [ap](hipStream_t st) { return launch_attention(ap, st); },
         2.0 * heads * Tn * (double)Tn * (dk + dv));
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
    // ---- head input (autodrive_head.py:70-87): cat([prev, curr]).  Round 5: cv2 -- the backbone's last convolution -- stores this frame's P5 STRAIGHT into
    // the second half of the 512-channel tensor (an alias view with the tensor's row pitch: the epilogues store at pixel * Cstore + channel), so the
    // "place" copy launch is gone; the previous frame's features move to the first half right BEFORE it ("shift": the only copy a streaming pair needs).
    Folded f2 = fold_conv_norm(blob, cp + ".cv2");
    const int c5 = f2.cout;
    if (c5 % 32 != 0) throw std::runtime_error("backbone.p5.3.cv2: output channels must be a multiple of 32");
    head_cat = new_act("head.cat", 2 * c5, t->H, t->W);
    {
      const ActView cv = head_cat->view();
      ad_shift_op_ = ops_.size();
      push("head.shift_prev", "chan_copy", [=](hipStream_t st) { return launch_chan_copy(cv, c5, cv, 0, c5, st); });
    }
    auto slice = std::make_unique<Act>();
    slice->name = cp + ".cv2";
    slice->Creal = c5;
    slice->C = head_cat->C;            // row pitch of the tensor it lives in
    slice->H = head_cat->H;
    slice->W = head_cat->W;
    slice->hi = head_cat->hi + c5;
    slice->lo = head_cat->lo ? head_cat->lo + c5 : nullptr;
    acts_.push_back(std::move(slice));
    ConvOpts o2;
    o2.act = ACT_SILU;
    ad_place_op_ = ops_.size();        // prime_previous() runs the plan up to and including this launch (minus the shift)
    x = add_conv(cp + ".cv2", t, f2.w, f2.b, f2.cout, 1, o2, acts_.back().get());
    if (ops_.size() != ad_place_op_ + 1) throw std::runtime_error("backbone.p5.3.cv2: expected ONE launch (no split-K finish) in front of the head");
    // the alias store relies on the kernels' store guard `channel < Ncols` with Ncols = round_up(c5, 32) = c5 (checked above): a tile wider than c5 computes
    // padding rows but never stores them into the neighbouring "previous frame" half (tests/test_gpu_autodrive.py::test_autodrive_stream_of_four_frames)
  }
  // ---- head: 3 x (conv3x3 + SiLU) -> flatten (C-major) -> MLP
  Act* cat = head_cat;
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
    upload_fc_weights(&fp, w.data, w.shape[0], K);
    fp.b = dupload(b.data);
    fp.N = w.shape[0];
    fp.K = K;
    fp.act = act;
    fp.out = out ? out : static_cast<float*>(dalloc((size_t)w.shape[0] * sizeof(float)));
    push(name, "fc", [fp](hipStream_t st) { return launch_fc(fp, st); }, 2.0 * fp.N * K, (fp.w8 ? 1.0 : 4.0) * fp.N * K);
    return fp.out;
  };
  const float* f1 = fc("head.fc1.0", v0, flat, ACT_SILU, nullptr);
  const float* f2 = fc("head.fc2.0", f1, 768, ACT_SILU, nullptr);
  {  // the three scalar heads (autodrive_head.py:60-66: ReLU / tanh / none on one row each) as ONE stacked [3][512] matrix with per-row activations
     // (round 4: three launches before); row arithmetic is the single-row kernel's
    const char* names[3] = {"head.distance_head.0", "head.curvature_head.0", "head.flag_head"};
    const int acts[3] = {ACT_RELU, ACT_TANH, ACT_NONE};
    std::vector<float> w3, b3;
    FcParams fp{};
    for (int r = 0; r < 3; ++r) {
      const HostTensor& w = T(std::string(names[r]) + ".weight");
      const HostTensor& b = T(std::string(names[r]) + ".bias");
      if (w.shape.size() != 2 || w.shape[0] != 1 || w.shape[1] != 512 || b.data.size() != 1) throw std::runtime_error(std::string("linear shape mismatch: ") + names[r]);
      w3.insert(w3.end(), w.data.begin(), w.data.end());
      b3.push_back(b.data[0]);
      fp.act_rows |= (unsigned)acts[r] << (8 * r);
    }
    fp.act_rows |= 0x80000000u;   // non-zero even if every row's code were 0 (byte 3 is never a row: N = 3)
    fp.x = f2;
    upload_fc_weights(&fp, w3, 3, 512);
    fp.b = dupload(b3);
    fp.N = 3;
    fp.K = 512;
    fp.act = ACT_NONE;
    fp.out = d_logits_;
    push("head.distance+curvature+flag", "fc", [fp](hipStream_t st) { return launch_fc(fp, st); }, 2.0 * 3 * 512, (fp.w8 ? 1.0 : 4.0) * 3 * 512);
  }
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
    d_status_ = static_cast<unsigned*>(dalloc(VP_PROBE_BLOCKS * sizeof(unsigned), true));
    VP_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&h_status_), VP_PROBE_BLOCKS * sizeof(unsigned), hipHostMallocDefault));
    std::memset(h_status_, 0, VP_PROBE_BLOCKS * sizeof(unsigned));
    Op probe;
    probe.name = "finite_probe";
    probe.kernel = "finite_probe";
    probe.bytes = 4.0 * out_c_ * out_h_ * out_w_;
    probe.run = [this](hipStream_t st) -> hipError_t {
      if (!finite_check_) return hipSuccess;
      // per-pass verdict: the scan OVERWRITES every word (no clear node in the graph, nothing sticky: ADVICE round 4)
      return launch_finite_probe(d_logits_, (size_t)out_c_ * out_h_ * out_w_, d_status_, st);
    };
    ops_.push_back(std::move(probe));
    if (kind_ == 3) {   // EgoLanes: the AutoSteer hand-over ring, when enabled (two device-to-device copy nodes behind the pass)
      Op ring;
      ring.name = "lane_ring";
      ring.kernel = "lane_ring";
      ring.bytes = 3.0 * 4.0 * out_c_ * out_h_ * out_w_;
      ring.run = [this](hipStream_t st) -> hipError_t {
        if (!lane_ring_) return hipSuccess;
        const size_t n = (size_t)out_c_ * out_h_ * out_w_;
        if (hipError_t e = hipMemcpyAsync(d_lane_ring_, d_lane_ring_ + n, n * sizeof(float), hipMemcpyDeviceToDevice, st); e != hipSuccess) return e;   // t-1 := t
        return hipMemcpyAsync(d_lane_ring_ + n, d_logits_, n * sizeof(float), hipMemcpyDeviceToDevice, st);                                                 // t := this pass
      };
      ops_.push_back(std::move(ring));
    }
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
  VP_HIP_CHECK(hipStreamSynchronize(stream_));   // (no device-wide synchronise: every upload above waited for its own copy)
}

}  // namespace vp
