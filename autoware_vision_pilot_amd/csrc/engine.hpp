// Host-side engine of libvp_hip: weight blob -> folded/packed device weights -> static per-model launch plan
// (one HIP stream, replayed as a hipGraph) behind the C ABI of include/vp_hip.h.
#pragma once
#include <functional>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "kernels.hpp"

namespace vp {

struct HostTensor {
  std::vector<int> shape;
  std::vector<float> data;
  size_t numel() const { return data.size(); }
};

// "VPW1" container: u32 count, then per tensor {u16 name_len, name, u8 ndim, u32 dims[ndim], fp32 data}.
// Written by autoware_vision_pilot_amd/weights.py from a reference state_dict (keys kept verbatim).
class WeightBlob {
 public:
  void parse(const void* blob, size_t bytes);
  const HostTensor& get(const std::string& key) const;
  bool has(const std::string& key) const { return t_.count(key) != 0; }
  size_t size() const { return t_.size(); }
  // FNV-1a over (key suffix, shape, fp32 bytes) of every tensor whose key starts with `prefix`, in key order:
  // equal hashes <=> the sub-network under that prefix carries the same parameters (shared-prefix engines).
  unsigned long long group_hash(const std::string& prefix) const;
  // BASELINE configs[4] "fp8 weights": every conv / linear weight -> per-output-channel symmetric OCP e4m3 (max 448,
  // 3 mantissa bits, subnormals), stored back DE-quantised (oracle/autodrive.py quantize_fp8_e4m3, same arithmetic).
  void quantize_fp8_e4m3();

 private:
  std::map<std::string, HostTensor> t_;
};

// onnx_reader.cpp: ONNX file (reference exporter settings) -> tensors in state_dict naming / the same as a VPW1 blob.
std::map<std::string, HostTensor> load_onnx_state_dict(const std::string& path);
std::vector<char> onnx_to_blob(const std::string& path);

// Pillow's resample coefficient tables (engine.cpp): bounds [out][2], coefficients [out][ksize] with 22 fractional bits; returns ksize
int pil_coeffs(int in_size, int out_size, int filter, std::vector<int>* bounds, std::vector<int>* kk);
// 11-bit fixed-point bilinear taps of the cv::resize INTER_LINEAR restatement ([dst][4] = i0, i1, w0, w1; engine_io.cpp, oracle/pre_post.py linear_taps_u8)
void linear_taps_u8(int src, int dst, std::vector<int>* tab);

// a value left the representable range (vp_status VP_ERR_RANGE): a weight beyond fp16 at load, inf / NaN in a network's output
struct RangeError : std::runtime_error {
  using std::runtime_error::runtime_error;
};

struct Act {
  std::string name;
  int Creal = 0, C = 0, H = 0, W = 0;
  int frames = 1;  // batched encoder: `frames` camera frames stacked along H (H = frames * per-frame height)
  half_t* hi = nullptr;
  half_t* lo = nullptr;
  ActView view() const { return ActView{hi, lo, H, W, C}; }
  size_t elems() const { return (size_t)H * W * C; }
};

struct Op {
  std::string name;
  std::string kernel;  // kernel instantiation tag (groups launches for the roofline report)
  std::string launch;  // launch geometry fixed at PLAN time that the tag does not spell (split-K factor, persistent-group count): hashed by vp_plan_hash
  double flops = 0, bytes = 0;
  // matrix work the launch EXECUTES when that differs from the reference formulation's count in `flops` (composed up-sampling stages: flops keeps
  // SURVEY.md 8d's three-op count so whole-frame figures stay comparable across rounds); 0 = same as flops
  double flops_executed = 0;
  std::function<hipError_t(hipStream_t)> run;
};

// ConvTranspose2d(k2, s2) [+ 1x1 skip link] -> Conv3x3 multiplied out at load (engine_upconv.cpp): fp64, real (un-padded) channel counts
struct UpconvComposed {
  int cin = 0, cm = 0, cout = 0, cs = 0;
  std::vector<double> wx;    // [4 phases (py * 2 + px)][4 taps (a * 2 + b)][cout][cin]: out(2y + py, 2x + px) += wx . x(y + py - 1 + a, x + px - 1 + b)
  std::vector<double> ws;    // [9 taps (ty * 3 + tx)][cout][cs]: 3x3 convolution of the skip tensor
  std::vector<double> bias;  // [9 classes (row class * 3 + column class)][cout]; class 0 / 1 / 2 = first / inner / last row (column) of the output map
};

struct ConvOpts {
  int act = ACT_NONE;
  int res_mode = RES_NONE;
  const Act* res = nullptr;
  int tile = -1, bk = -1, nsplit = -1;
  float* logits_out = nullptr;  // STORE_NCHW_F32 target
  const Act* in2 = nullptr;     // K-extension tensor of the fused ConvTranspose + skip-link GEMM (add_convT_skip)
  int stride = 1;               // 3x3 stride-2 convs of AutoDrive (generic GEMM kernel only)
  int post_act = ACT_NONE;      // activation after the residual (CTX: SiLU(c4*x + x))
  bool pixel_shuffle = false;   // set by add_convT*: the GEMM stores with STORE_SHUFFLE2 (never the pointwise kernel)
};

// Frame pools registered for DMA (vp_register_frames): process-wide, engine_io.cpp
int register_frame_range(const void* pool, size_t bytes);
int unregister_frame_range(const void* pool);
bool frame_range_registered(const void* p, size_t bytes);

class Engine {
 public:
  // base != nullptr builds a SHARED-PREFIX engine (vp_create_shared): sub-networks whose parameters equal the base
  // engine's (backbone; backbone + context + neck) are not rebuilt -- this engine's plan starts from the base
  // engine's feature tensors and runs on the base engine's stream, after it, on the frame the base last processed.
  // frames > 1 (base == nullptr): BATCHED ENCODER -- preprocess + EfficientNet backbone of `frames` camera frames per pass,
  // no context / neck / head; its taps feed shared-prefix engines built with (base = this, frame_index = f).
  Engine(int kind, const WeightBlob* blob, int precision, int gpu_id, Engine* base = nullptr, int frames = 1, int frame_index = 0);
  ~Engine();
  Engine(const Engine&) = delete;
  Engine& operator=(const Engine&) = delete;

  // configuration
  void set_input_format(int pixel_format, int plane_order);
  void set_decode_mode(int mode);
  // u8 -> [0, 1] ahead of (x - mean) / std: 0 = q / 255 (torchvision to_tensor: the Python operator API), 1 = q * fl(1/255)
  // (cv::Mat::convertTo(CV_32FC3, 1.0 / 255.0): the C++ front-ends, onnx_runtime_backend.cpp:45, onnxruntime_engine.cpp:85)
  void set_norm_form(int form);
  int norm_form() const { return norm_form_; }
  // AutoSteer hand-over: the raw EgoLanes logits of the last TWO passes, [t-1 | t] = fp32 {1, 6, 80, 160}, kept on the device and
  // updated behind every pass (production_release main.cpp:472-534 builds the same buffer on the host with a circular buffer and two memcpys)
  void set_lane_ring(bool on);
  bool lane_ring() const { return lane_ring_; }
  float* dev_lane_ring() const { return d_lane_ring_; }
  int lane_ring_frames() const { return ring_frames_; }   // 0, 1, 2: the reference runs AutoSteer only once two frames are in
  void fetch_lane_ring(float* dst);                        // D2H of the 6 x 80 x 160 floats + sync
  // frame resize ahead of the network: 0 = the integer bilinear modelled on cv::resize INTER_LINEAR (the C++ nodes; default of the scene
  // networks), 1 / 2 = Pillow's antialiased BILINEAR / BICUBIC (the Python scripts' Image.resize; 1 is AutoDrive's default)
  void set_resize_mode(int mode);
  int resize_mode() const { return resize_mode_; }
  int net_h() const { return kind_ == 4 ? 512 : 320; }   // AutoDrive: autodrive_network.py:8-9
  int net_w() const { return kind_ == 4 ? 1024 : 640; }
  // AutoDrive (kind 4) pairs every frame with the previous one: infer_pair() runs the backbone on `prev` first
  // (features only), enqueue() then shifts the feature slots and runs backbone + head on the resident frame.
  void prime_previous();

  // frame path
  void upload_frame(const uint8_t* frame, int h, int w, int stride, int index = 0);  // index: slot of a batched encoder
  int frames() const { return frames_; }
  void upload_tensor(const float* nchw);
  void enqueue();
  void sync();
  void fetch_outputs();      // enqueue_fetch() + sync
  void enqueue_fetch();      // async D2H of the selected outputs (set_outputs) into the pinned host buffers
  // bit 0: logits, bit 1: mask.  De-selected outputs stay in HBM and are copied on first use (vp_logits / vp_mask_u8).
  void set_outputs(int mask) {
    if (mask < 0 || mask > 3) throw std::invalid_argument("outputs: bit 0 = logits, bit 1 = mask");
    outputs_ = mask;
  }
  // stage vp_infer's H2D through a pinned double buffer owned by the engine (default on)
  void set_pinned_staging(bool on) { pinned_staging_ = on; }
  // inf / NaN probe on the logits at the end of every pass (default on; a 3 us kernel): reported by the next synchronising call
  void set_finite_check(bool on) {
    if (on != finite_check_) {
      finite_check_ = on;
      graph_valid_ = false;
      ++plan_epoch_;
    }
  }
  void check_status();  // throws RangeError if the last fetched pass flagged a non-finite output
  bool poll_status();   // the same verdict without the throw (vp_infer_multi polls every member, then reports once)
  void copy_outputs_device(void* logits_dst, void* mask_dst);
  void mask_resized(uint8_t* dst, int h, int w);
  void depth_resized(float* dst, int h, int w);
  void visualize_mask(int viz_type, uint8_t* dst_bgr, int dst_h, int dst_w);
  void visualize_depth(uint8_t* dst_bgr, int h, int w);
  void read_input_tensor(float* dst);

  const float* host_logits();   // pinned host copy; fetched now if the last pass did not copy it
  const uint8_t* host_mask();
  int frame_h() const { return frame_h_; }
  int frame_w() const { return frame_w_; }
  void* dev_logits() const { return d_logits_; }
  void* dev_mask() const { return d_mask_; }
  float* alloc_f32(size_t n) { return static_cast<float*>(dalloc(n * sizeof(float))); }  // test operator entry: a logits buffer
  int out_c() const { return out_c_; }
  int out_h() const { return out_h_; }
  int out_w() const { return out_w_; }
  bool have_outputs() const { return have_outputs_; }
  void use_graph(bool on) { use_graph_ = on; }

  void timer_begin();
  float timer_end();
  int profile_layers(int iters, float* ms, int cap);

  const std::vector<Op>& ops() const { return ops_; }
  const std::vector<std::unique_ptr<Act>>& acts() const { return acts_; }
  void read_act(int i, float* dst);

  // plan-building blocks (public for the single-op test entry)
  Act* new_act(const std::string& name, int creal, int h, int w);
  Act* add_conv(const std::string& name, const Act* in, const std::vector<float>& w, const std::vector<float>& b, int cout, int ks,
                const ConvOpts& o, Act* out_override = nullptr);
  Act* add_convT_skip(const std::string& up_name, const std::string& skip_name, const Act* in, const Act* skip_in,
                      const std::vector<float>& wt, const std::vector<float>& bt, const std::vector<float>& ws,
                      const std::vector<float>& bs, int cout);
  Act* add_convT(const std::string& name, const Act* in, const std::vector<float>& w, const std::vector<float>& b, int cout,
                 const ConvOpts& o);
  void compose_upconv(const float* wt, const float* bt, const float* ws, const float* bs, const float* w3, const float* b3, int cin, int cm, int cout,
                      int cs, UpconvComposed* out);
  // shape: 6 / 7 (kernels.hpp), -1 = dispatch rule; nsplit: K slices, <= 0 = dispatch rule
  Act* add_upconv(const std::string& name, const Act* in, const Act* skip_in, const std::vector<float>& wt, const std::vector<float>& bt,
                  const std::vector<float>& ws, const std::vector<float>& bs, const std::vector<float>& w3, const std::vector<float>& b3, int cm,
                  int cout, int act, int shape = -1, int nsplit = 0, const std::string& out_name = std::string());
  bool upconv_wanted() const;  // composed up-sampling stages in this engine's plan (parity mode; VP_UPCONV=0: the three-op form of rounds 1-5)
  void run_eager();
  void run_ops(hipStream_t st, size_t begin, size_t end);
  void enqueue_multi(const std::vector<Engine*>& heads);
  void set_multi_fork(bool on) { multi_fork_ = on; }
  void upload_act(Act* a, const float* chw);
  hipStream_t stream() const { return stream_; }
  int gpu() const { return gpu_; }
  int decode_mode() const { return decode_mode_; }
  bool host_logits_current() const { return host_logits_valid_; }  // the pinned host logits belong to the LAST pass
  bool split() const { return (precision_ & 15) == 1; }
  bool fp8_weights() const { return (precision_ & 16) != 0; }
  // kernel plan target of THIS engine (round 6: a creation flag, VP_PLAN_LATENCY; the process-wide developer option VP_PLAN_TARGET still selects it for
  // every engine created while it is set): the round-4 choices for the CU-time rules of engine_dispatch.cpp
  bool plan_latency() const;
  // ... kept as e4m3 BYTES in HBM (round 4): AutoDrive engines (BASELINE configs[4]) and the operator entry; a scene network with the flag keeps
  // de-quantised fp16 planes on its LDS-DMA / register-stationary kernels
  bool fp8_storage() const { return fp8_weights() && (kind_ == 4 || kind_ < 0); }
  int shared_level() const { return shared_level_; }  // 0 own network, 1 backbone shared, 2 backbone + context + neck shared
  // device bytes of the network's WEIGHT tensors by storage class (bias / scale vectors and tables not counted): [0] e4m3 codes, [1] fp16 planes, [2] fp32
  const unsigned long long* weight_bytes() const { return wbytes_; }

  std::string last_error;

 private:
  struct PackedConv {
    half_t* w_hi = nullptr;
    half_t* w_lo = nullptr;
    float* bias = nullptr;
    float* wscale = nullptr;  // [CoutW] 2^-prescale of the weight rows (engine_internal.hpp prescale_exp), or the fp8 row scales
    uint8_t* w8 = nullptr;    // VP_WEIGHTS_FP8 on a kernel that stages weights through registers: e4m3 codes in that kernel's layout (w_hi / w_lo null)
    int CoutW = 0, tile = 0, bk = 32, nsplit = 1;
  };
  void construct(int kind, const WeightBlob* blob, int precision, int gpu_id, Engine* base);
  void release();  // frees every device / host resource; idempotent (destructor and failed construction)
  void* dalloc(size_t bytes, bool zero = true);
  void dfree(void* p);  // hipFree + forget (buffers that are re-grown at run time; the caller has synchronised the stream)
  // Synchronous copies / fills WITHOUT the legacy stream: hipMemcpy / hipMemset run on the NULL stream, which implicitly joins every blocking
  // stream of the device, and the runtime refuses that while ANY thread captures a graph ("would make the legacy stream depend on a
  // capturing ... stream") -- an engine created or read on one host thread invalidated the capture another thread's engine was recording
  // (round 4: seen once vp_create got fast enough to overlap it).  These go through the engine's own non-blocking stream and wait for it.
 public:
  void copy_h2d(void* d, const void* h, size_t bytes);
  void copy_d2h(void* h, const void* d, size_t bytes);
  void fill_zero(void* d, size_t bytes);
 private:
  template <class T>
  void upload_grow(T*& d, size_t& cap_elems, const std::vector<T>& v);  // re-uses d while v fits, else frees it and allocates anew
  const void* zero_page();  // 256 bytes of zeros in device memory (LDS-DMA source for out-of-image pixels, kernels_head.hip)
  template <class T>
  T* dupload(const std::vector<T>& v);
  // FC matrix [N][K] -> fp32 rows, or (VP_WEIGHTS_FP8) e4m3 codes + row scales.  row_amax: per-row maximum the quantiser saw when it is not the
  // maximum of the K values handed in (AutoDrive's exp0: a row's three Conv1d taps share one scale, only the centre tap is used), else null
  void upload_fc_weights(FcParams* fp, const std::vector<float>& w, int N, int K, const std::vector<float>* row_amax = nullptr);
  int gemm_dma_nsplit(int M, int ncols, int kw) const;
  bool gemm_dma_wanted(int H, int W, int ncols, int cin_pad, int cin2_pad, int cstore) const;
  void choose_conv_cfg(int M, int ncols, int cin_pad, int ks, const ConvOpts& o, PackedConv* pc);
  void push_conv_op(const std::string& name, const Act* in, const PackedConv& pc, int ks, int ncols, const ConvOpts& o, Act* out,
                    int store_mode, int cout_real);

  void build_model(const WeightBlob& blob);
  void build_autodrive(const WeightBlob& blob);
  std::vector<Act*> build_backbone(const WeightBlob& blob, const std::string& prefix);
  Act* build_context(const WeightBlob& blob, const std::string& p, const Act* deep, int cctx);
  Act* build_neck(const WeightBlob& blob, const std::string& p, const Act* ctx, const std::vector<Act*>& feats, int cctx);
  void build_head(const WeightBlob& blob, const std::string& p, const Act* neck, const std::vector<Act*>& feats);
  void finish_plan();
  void ensure_tables(int h, int w);
  void capture_graph();

  Act* frame_view(const Act* a, int f);  // per-frame window of a stacked activation (no allocation)
  int kind_, precision_, gpu_;
  int frames_ = 1, frame_index_ = 0;
  hipStream_t stream_ = nullptr;
  Engine* base_ = nullptr;
  int shared_level_ = 0;
  unsigned long long hash_bb_ = 0, hash_ctx_ = 0, hash_neck_ = 0;
  std::vector<Act*> feats_;   // backbone taps f0..f4 (kept for shared-prefix engines)
  Act* neck_out_ = nullptr;
  size_t ad_shift_op_ = 0, ad_place_op_ = 0;  // AutoDrive: ops_[shift] moves curr->prev features, ops_[place] stores the new ones
  bool ad_primed_ = false;
  std::vector<void*> allocs_;
  unsigned long long wbytes_[3] = {0, 0, 0};
  std::vector<std::unique_ptr<Act>> acts_;
  std::vector<Op> ops_;
  size_t first_net_op_ = 0;  // ops_[0] is the preprocess op (skipped for vp_infer_tensor)

  // input
  int pixel_format_ = 0, plane_order_ = 0, decode_mode_ = 0, norm_form_ = 0;
  uint8_t* d_frame_ = nullptr;
  size_t frame_cap_ = 0;
  int frame_h_ = 0, frame_w_ = 0, frame_stride_ = 0;
  int* d_xtab_ = nullptr;
  int* d_ytab_ = nullptr;
  int tab_h_ = 0, tab_w_ = 0;
  float* d_input_ = nullptr;  // [3][320][640]
  int resize_mode_ = 0;
  int* d_pil_hb_ = nullptr;
  int* d_pil_hk_ = nullptr;
  int* d_pil_vb_ = nullptr;
  int* d_pil_vk_ = nullptr;
  int pil_hks_ = 0, pil_vks_ = 0;
  size_t pil_cap_[5] = {0, 0, 0, 0, 0};  // capacities (elements) of hb, hk, vb, vk, tmp
  uint8_t* d_pil_tmp_ = nullptr;
  PilResampleParams pil_params(const PreprocessParams& pp) const;
  bool input_is_tensor_ = false;

  // outputs
  int out_c_ = 0, out_h_ = 0, out_w_ = 0;
  float* d_logits_ = nullptr;
  uint8_t* d_mask_ = nullptr;
  void* d_zero_ = nullptr;
  float* h_logits_ = nullptr;
  uint8_t* h_mask_ = nullptr;
  bool have_outputs_ = false;
  bool decode_fused_ = false;  // the logits convolution writes d_mask_ in its epilogue: no separate decode launch
  int outputs_ = 3;  // vp_set_outputs: bit 0 logits, bit 1 mask copied to the host by vp_infer*
  bool host_logits_valid_ = false, host_mask_valid_ = false;
  float* d_lane_ring_ = nullptr;  // [6][out_h][out_w] fp32: previous | current EgoLanes logits (vp_set_lane_ring)
  bool lane_ring_ = false;
  int ring_frames_ = 0;
  void note_pass() { if (lane_ring_ && ring_frames_ < 2) ++ring_frames_; }
  unsigned* d_status_ = nullptr;  // sticky non-finite flag written by the probe kernel
  unsigned* h_status_ = nullptr;  // pinned copy, fetched with the outputs
  bool finite_check_ = true, status_pending_ = false;
  // pinned staging of the caller's (pageable) frame: two slots
  uint8_t* h_frame_ = nullptr;
  size_t h_frame_cap_ = 0;
  int h_frame_slot_ = 0;
  hipEvent_t h_frame_ev_[2] = {nullptr, nullptr};  // recorded behind the H2D that reads a slot; awaited before the slot is rewritten
  bool pinned_staging_ = true;

  // split-K scratch
  std::vector<float**> partial_slots_;

  // resize scratch
  void* d_resize_out_ = nullptr;
  size_t resize_cap_ = 0;
  int* d_rs_tab_ = nullptr;
  uint8_t* d_viz_lut_ = nullptr;  // [3 viz types][256][3]
  uint8_t* d_viridis_ = nullptr;  // [256][3] BGR (viridis_lut.inc)
  unsigned* d_minmax_ = nullptr;  // order keys of the depth plane's min / max
  void* d_depth_viz_ = nullptr;
  size_t depth_viz_cap_ = 0;
  void resize_depth_on_device(int h, int w);  // fp32 h x w plane -> d_resize_out_
  std::vector<int> rs_idx_host_;
  std::vector<float> rs_wgt_host_;
  size_t rs_tab_cap_ = 0;

  // graph
  bool use_graph_ = true;
  hipGraph_t graph_ = nullptr;
  hipGraphExec_t graph_exec_ = nullptr;
  // enqueue_multi: ONE graph for this engine and its shared-prefix heads, the heads that need only the backbone forked onto side
  // streams behind it (they overlap this engine's own context / neck / head); keyed by the head list and every member's plan epoch
  hipGraph_t multi_graph_ = nullptr;
  hipGraphExec_t multi_exec_ = nullptr;
  std::vector<std::pair<const Engine*, unsigned long long>> multi_key_;
  std::vector<hipStream_t> side_streams_;
  std::vector<hipEvent_t> side_events_;   // [0] fork, [1 + i] join of side stream i
  bool multi_fork_ = true;                // vp_set_multi_fork
  size_t n_fork_ops_ = 0;                 // ops_[0, n_fork_ops_): preprocess + backbone (what a level-1 head consumes)
  unsigned long long plan_epoch_ = 0;     // bumped whenever graph_valid_ is cleared
  bool graph_valid_ = false;
  bool warmed_ = false;
  hipEvent_t ev0_ = nullptr, ev1_ = nullptr;
};

#define VP_HIP_CHECK(expr)                                                                                   \
  do {                                                                                                       \
    hipError_t _e = (expr);                                                                                  \
    if (_e != hipSuccess)                                                                                    \
      throw std::runtime_error(std::string("HIP error: ") + hipGetErrorString(_e) + " at " #expr);           \
  } while (0)

}  // namespace vp
