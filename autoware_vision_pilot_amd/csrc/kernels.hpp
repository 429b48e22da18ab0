// Kernel parameter blocks + host launchers of libvp_hip (definitions in kernels_conv.hip / kernels_misc.hip).
#pragma once
#include "common.hpp"

namespace vp {

// developer option set through vp_set_option (options.cpp), or nullptr -- the library never reads the environment
const char* dev_option(const char* key);
bool dev_option_is(const char* key, char first);  // value set and starting with `first`: one lookup

struct PreprocessParams {
  const uint8_t* frame;  // device, HxWx3
  int stride;            // bytes per row
  const int* xtab;       // [out_w][4] = x0, x1, a0, a1
  const int* ytab;       // [out_h][4] = y0, y1, b0, b1
  int out_h, out_w;
  int src_c[3];          // source byte index feeding output plane 0,1,2
  float mean[3], stdv[3];
  float* out;            // [3][out_h][out_w]
  int norm_form;         // 0: q / 255 (torchvision to_tensor), 1: q * fl(1/255) (cv::Mat::convertTo) -- kernels_misc.hip unit_from_u8
};

// Pillow's antialiased resample (PIL.Image.resize with BILINEAR / BICUBIC; src/libImaging/Resample.c 8bpc path) + /255 + (x-mean)/std +
// HWC->CHW: the AutoDrive frame path (Models/visualizations/AutoDrive/video_visualization.py:29-33) and the scene networks' Python
// visualisation scripts.  Definition pinned against PIL itself: oracle/pre_post.py resize_pil_u8.  Two passes, as Pillow's: horizontal
// into a u8 image [in_h][out_w][3], vertical from it; coefficient tables (22 fractional bits) are built on the host in double.
struct PilResampleParams {
  const uint8_t* frame;  // device, in_h x in_w x 3
  int stride;            // bytes per row
  int in_h, in_w, out_h, out_w;
  const int* hb;         // [out_w][2] = first source column, tap count
  const int* hk;         // [out_w][hks]
  int hks;
  const int* vb;         // [out_h][2]
  const int* vk;         // [out_h][vks]
  int vks;
  uint8_t* tmp;          // [in_h][out_w][3]
  int src_c[3];
  float mean[3], stdv[3];
  float* out;            // [3][out_h][out_w]
  int norm_form;         // as PreprocessParams
};

struct StemParams {
  const float* in;  // [3][H][W]
  int H, W;         // input size (output is H/2 x W/2)
  const float* w;   // [27][32]  (k = (ci*3+ky)*3+kx, co fastest), BN scale folded
  const float* b;   // [32]
  ActView out;      // C = 32
  unsigned long long* zero;  // optional: the squeeze-excite accumulators of the frame, zeroed here (every block's pool comes later)
  size_t zero_n;
};

struct DwParams {
  ActView in, out;
  const float* w;  // [k*k][C]
  const float* b;  // [C]
  int k, stride;
  // squeeze-excite average pool, fused: per-channel sums of the OUTPUT accumulated as 2^24 fixed-point int64
  // (integer atomics are associative -> bit-deterministic regardless of workgroup order); zeroed once per frame
  unsigned long long* sums;  // [replicas][C]
  int replicas;              // power of two <= kSeMaxReplicas: workgroup x-index & (replicas-1) picks the row
  // batched encoder (frames > 1): grid.z = frame; tensors are [frames][H][W][C], sums [frames][replicas][C]
  int frames;
};

// MBConv front half in one launch (kernels_mbconv.hip): expand 1x1 + SiLU -> depthwise k x k / stride + SiLU -> SE pool sums
struct MbFrontParams {
  ActView in;           // block input, H x W x Cin_pad, (hi, lo)
  const half_t* w_hi;   // expand weights [Cexp_pad][Cin_pad], BN folded, (hi, lo); zero rows / columns in the padding
  const half_t* w_lo;
  const float* b_exp;   // [Cexp_pad]
  const float* s_exp;   // [Cexp_pad] 2^-prescale of the expand weight rows (see ConvGemmParams::wscale)
  const float* w_dw;    // [k*k][Cexp_pad] fp32, BN folded
  const float* b_dw;    // [Cexp_pad]
  ActView out;          // H/stride x W/stride x Cexp_pad, (hi, lo)
  int k, stride;
  unsigned long long* sums;  // [replicas][Cexp_pad] fixed-point channel sums (as DwParams); may be null when zsums is given
  int replicas;
  // optional: the squeeze FC of the block's squeeze-excite, taken here from the workgroup's own channel sums (it is linear in them):
  // zsums[replica][j] += fixed(sum_c w1[j][c] * patch_sum[c]) -- the back half (mbconv_back) then starts from sq numbers, not from the means
  const float* w1;            // [sq][Cexp_pad] or null
  int sq;
  unsigned long long* zsums;  // [replicas][64] 2^24 fixed point
};

struct PoolParams {
  ActView in;
  float* partial;  // [nslab][C]
  int nslab;
};

struct SeParams {
  const unsigned long long* sums;  // [replicas][C] fixed-point channel sums of the fused pool (see DwParams)
  int replicas;
  int C, Creal, sq;
  float inv_hw;
  const float* w1;  // [sq][C]   (zero in pad columns)
  const float* b1;  // [sq]
  int frames;       // batched encoder: grid.y = frame; sums [frames][replicas][C]
};

struct ScaleWParams {
  const float* w;   // [rows][C] fp32 projection weights (BN folded)
  half_t* out_hi;   // [rows][C] = w * gate[c], (hi, lo) split
  half_t* out_lo;   // may be null
  int rows, C;
  const float* w2;  // [C][sq]  excite FC: gate[c] = sigmoid(b2[c] + w2[c][:] . s1)
  const float* b2;  // [C]
  int sq, Creal;
  int frames;       // batched encoder: grid.y = frame; out_hi / out_lo [frames][rows][C]
};

// MBConv back half in one launch (kernels_mbconv.hip): squeeze-excite gate -> projection 1x1 (+BN) with the gate folded into its K axis
// (+ residual).  One frame per pass, (hi, lo) activations.
struct MbBackParams {
  ActView in;          // depthwise output, H x W x C (hi, lo)
  SeParams se;         // pool sums of that tensor, squeeze FC (frames == 1, se.C == in.C)
  const unsigned long long* zsums;  // [se.replicas][64] pre-activation squeeze sums from mbconv_front (MbFrontParams::zsums), or null: means + FC here
  const float* w2q;    // excite FC as [sqp / 4][C][4]: w2q[(q * C + c) * 4 + i] = fc2.weight[c][4 q + i], zero beyond sq
  const float* b2;     // [C]
  int sqp;             // sq rounded up to a multiple of 4
  const float* w;      // [out.C][C] fp32 projection weights (BN folded) TIMES the row's power-of-two prescale, zero rows / columns in the padding
  const float* bias;   // [out.C]
  const float* wscale; // [out.C] 2^-prescale (see ConvGemmParams::wscale)
  ActView res;         // residual (the block's input, same geometry as out) or hi == nullptr
  ActView out;         // H x W x Cout_pad (hi, lo)
};

// AutoSpeed detector pre / post-processing (kernels_detect.hip; autospeed/onnxruntime_engine.cpp:71-113, :170-290).
struct LetterboxParams {
  const uint8_t* frame;  // device, h x w x 3 BGR
  int stride;            // bytes per row
  const int* xtab;       // [new_w][4] bilinear taps of the resize to new_w x new_h (as PreprocessParams)
  const int* ytab;       // [new_h][4]
  int new_w, new_h, pad_x, pad_y;
  int out_h, out_w;
  float* out;            // [3][out_h][out_w] planes R, G, B in [0, 1]; 114 / 255 outside the pasted image
};

struct Detection {       // == autoware_pov::vision::autospeed::Detection (detection.hpp:8-12) == vp_detection
  float x1, y1, x2, y2, confidence;
  int class_id;
};

struct DetectParams {
  const float* raw;      // [num_attrs][num_boxes]: cx, cy, w, h (letterbox pixels), then the class scores
  int num_attrs, num_boxes;
  float conf_thresh, iou_thresh;
  float scale;           // letterbox geometry of the frame the tensor came from
  int pad_x, pad_y, orig_w, orig_h;
  float* boxes;          // scratch [num_boxes][4]: candidate boxes in sorted order
  int* cls;              // scratch [num_boxes]
  Detection* out;        // [out_cap]
  int out_cap;
  int* count;            // [2]: detections kept (may exceed out_cap: only out_cap are written), candidates above the threshold
};
constexpr int kDetectMaxBoxes = 16384;

struct FcParams {
  const float* x;
  const float* w;  // [N][K]
  // VP_WEIGHTS_FP8, real storage (round 4): when non-null the matrix is OCP e4m3 CODES [N][K] (one byte per weight; w is null) and
  // wscale8[n] the row's quantisation scale: out[n] = act(b[n] + wscale8[n] * sum_k q[n][k] x[k])
  const uint8_t* w8;
  const float* wscale8;
  const float* b;
  float* out;
  int N, K, act;
  // per-row activations of a small stacked matrix (N <= 4): byte n = ActFn of row n; 0 = `act` for every row (AutoDrive's three scalar heads in one launch)
  unsigned act_rows;
  int rows_kernel;       // 1: a thread per row (fc_rows_kernel: K <= 64, N >= 2048 -- fc_rows_ok); 0: two rows per wave
  const float* partial;  // optional: x = mean over nslab partial sums (avg-pool input, scene_context.py:27)
  int nslab, Kstride;
  float inv_hw;
};

bool fc_rows_ok(const FcParams& p);

struct CtxConv1Params {
  const float* map;  // [H][W] fp32
  int H, W;
  const float* w;    // [9][C]
  const float* b;    // [C]
  ActView out;
  int act;           // ACT_GELU (scene_context.py:46-47) or ACT_SILU (CTX.ctx0, common_layers.py:216-217)
};

// Round 5: the matvec that BUILDS the one-channel map and the 3x3 convolution that reads it, in one launch (ctx_exp_conv1_kernel):
//   scene networks: context_layer_2 (Linear 800 -> 200 + sigmoid, reshaped 10x20: scene_context.py:36-43) + context_layer_3 (conv 1 -> 128 + GELU, :46-47);
//   AutoDrive CTX : exp0 (C -> H*W, SiLU twice: common_layers.py:210-213) + ctx0 (conv 1 -> C/2 + SiLU, :216-217).
// fc.out is unused (the map never leaves LDS); fc.partial / nslab as for launch_fc; cv.map is unused.
struct CtxExpConv1Params {
  FcParams fc;
  CtxConv1Params cv;
  int tile;     // square pixel patch per workgroup: 16, 8 or 2 -- the (tile + 2)^2 map rows under it should fit ONE pass of 256 / glanes rows
  int glanes;   // lanes that share one row of the matvec (a power of two <= 64): few enough that a lane's pieces of a row are <= ~16 loads
};
bool ctx_exp_conv1_ok(const CtxExpConv1Params& p);
hipError_t launch_ctx_exp_conv1(const CtxExpConv1Params& p, hipStream_t st);

struct FusionParams {
  ActView f[5];
  int creal[5];   // 32,24,40,80,1280
  int shift[5];   // 4,3,2,1,0
  ActView out;    // 10x20 x 1472 (1456 real)
  int Creal_out;
  int octets;     // 1: one workgroup per output pixel, 16-byte pieces, window slices meeting in LDS (fusion_octets_ok); 0: a thread per output element
};
bool fusion_octets_ok(const FusionParams& p);

// tile ids: 0 = 128co x 128px, 1 = 64co x 128px, 2 = 64co x 64px, 3 = 32co x 128px ; bk = 32 | 64
hipError_t launch_conv_gemm(const ConvGemmParams& p, int tile, int bk, bool split, hipStream_t st);
int conv_tile_co(int tile);
int conv_tile_px(int tile);
// 3x3 halo kernel (kernels_conv3x3.hip); weights packed [cin/32][9][CoutW][32].
// halo tile ids: 0 = 128co x 16x16 px, 1 = 128co x 8x16, 2 = 64co x 16x16, 3 = 64co x 8x16, 4 = 32co x 8x16
hipError_t launch_conv3x3_halo(const ConvGemmParams& p, int tile, bool split, hipStream_t st);
// halo tiles 6 - 8: the pipelined fp16x3 kernels (kernels_conv3x3_x3.hip); conv + bias + {GELU, none}, NHWC
bool conv3x3_x3_supported(const ConvGemmParams& p, int shape);
hipError_t launch_conv3x3_x3(const ConvGemmParams& p, int shape, hipStream_t st);
int halo_tile_co(int tile);
int halo_tile_px(int tile);
int halo_tile_th(int tile);

// small-map 3x3 of the parity mode (kernels_conv3x3_map.hip; halo tile id 11): a workgroup = 32 output channels x a K slice x all 800 pixels of a
// 20x40 region; weights packed by conv3x3_map_pack_index; always finishes through splitk_finish_kernel (p.partial required, nsplit >= 1)
int conv3x3_map_geometry(int H, int W);  // 1 = 20x40 regions (neck), 2 = 10x20 regions (context block), 0 = neither
bool conv3x3_map_shape_ok(int H, int W, int cin_pad, int coutw);
size_t conv3x3_map_pack_index(int co, int ci, int t, int cin_pad);
size_t conv3x3_map_pack_index_k32(int co, int ci, int t, int cin_pad);   // fp16 engines: steps of 32 channels, plane = (ci >> 4) & 1
bool conv3x3_map_supported(const ConvGemmParams& p);
hipError_t launch_conv3x3_map(const ConvGemmParams& p, hipStream_t st);
// halo tile 12 ("map2", round 5): the map kernel on 64-channel weight slabs (two M tiles per wave), parity mode, 20x40 regions; weights packed by
// conv3x3_map2_pack_index; always finishes through splitk_finish_kernel
bool conv3x3_map2_shape_ok(int H, int W, int cin_pad, int coutw);
bool conv3x3_map2_supported(const ConvGemmParams& p);
size_t conv3x3_map2_pack_index(int co, int ci, int t, int cin_pad);
hipError_t launch_conv3x3_map2(const ConvGemmParams& p, hipStream_t st);
// last convolution of a head: 3x3, 64 / 128 channels -> <= 4 logit channels, fp32 NCHW + fused decode (kernels_head.hip); weights packed
// as for halo tile 4; zeros = the engine's zero page (>= 16 bytes of zeros in device memory)
bool head_conv_supported(const ConvGemmParams& p);
hipError_t launch_head_conv(const ConvGemmParams& p, const void* zeros, hipStream_t st);
// ConvTranspose (+ skip link) GEMM on the small maps: 8 waves, both operands by LDS-DMA three K steps deep (kernels_gemm_dma.hip)
bool gemm_dma_shape_ok(int M, int ncols, int cin_pad, int cin2_pad, int cstore);
size_t gemm_dma_pack_index(int n, int k, int kw);  // host packing of its weights (LDS image order per 256 x 32 tile)
bool gemm_dma_supported(const ConvGemmParams& p, bool split);
hipError_t launch_gemm_dma(const ConvGemmParams& p, hipStream_t st);
// register-stationary weights, pixel tiles by LDS-DMA (kernels_convt_rs.hip): K = 128, or 256 + 32 with the skip link; both precisions
bool convt_rs_supported(const ConvGemmParams& p, bool split);
int convt_rs_shape_case(int H, int W, int cin_pad, int cin2_pad, int ncols, int cstore);  // 0 = not covered
hipError_t launch_convt_rs(const ConvGemmParams& p, hipStream_t st);
// ---- composed up-sampling stage (kernels_upconv.hip, round 6) ------------------------------------------------------------------------
// ConvTranspose2d(k2, s2) [+ Conv1x1(skip)] followed by Conv3x3 with NO nonlinearity in between (scene_neck.py:29-35,41-46,52-57;
// scene_seg_head.py:24-29,35-38; scene_3d_head.py:26-31,38-41) is ONE linear map: a transposed convolution with a 4x4 kernel, stride 2, pad 1
// from the LOW-resolution tensor -- per output phase (Y & 1, X & 1) a 2x2 convolution of it -- plus a 3x3 convolution of the skip tensor with
// pre-multiplied weights, plus a bias that depends only on which of the 9 high-resolution taps are inside the map.  The weights are composed
// at load (upconv_compose: fp64 accumulation on the device) and the stage is ONE launch: 0.40-0.51x the matrix work of the two it replaces.
struct UpconvParams {
  const half_t* in_hi;   // low-resolution input  H x W x Cin (NHWC, Cin a multiple of 32), (hi, lo)
  const half_t* in_lo;
  const half_t* sk_hi;   // skip tensor 2H x 2W x Cs (Cs a multiple of 32) or null (Cs = 0)
  const half_t* sk_lo;
  int H, W, Cin, Cs;
  const half_t* w_hi;    // [4 phases][S steps][CoutW][32] in LDS image order (upconv_pack_index), S = upconv_steps(Cin, Cs)
  const half_t* w_lo;
  const float* bias;     // [9][CoutW]: (row class * 3 + column class), class 0 = first row / column of the OUTPUT map, 2 = last, 1 = inside
  const float* wscale;   // [4][CoutW] 2^-prescale of the weight rows of each phase (ConvGemmParams::wscale)
  int CoutW, Ncols, Cstore;
  half_t* out_hi;        // 2H x 2W x Cstore
  half_t* out_lo;
  int act;               // ACT_GELU | ACT_NONE
  int nsplit;            // K slices over the chunk list (> 1: fp32 partials + upconv_finish_kernel)
  float* partial;        // [nsplit][4 H W][CoutW]
};
// shapes: 6 = 8 waves, 16x16 low-resolution pixels x 128 channels per phase (one workgroup per CU); 7 = 4 waves, 8x16 pixels x 128 channels (two per CU)
bool upconv_supported(const UpconvParams& p, int shape);
// The K axis of one output phase (py, px) is a list of 32-channel CHUNKS, each with its own tap list (a K "step" = one chunk x one tap = one weight
// tile of CoutW x 32 and one shifted read of the chunk's halo image):
//   chunks [0, Cin / 32): the low-resolution input, 4 taps (a, b) in {0, 1}^2 = low-resolution pixel (y + py - 1 + a, x + px - 1 + b);
//   then per CLASS (qy, qx) of skip pixels (the skip tensor seen as four half-resolution images S[qy][qx](y, x) = skip(2y + qy, 2x + qx)), Cs / 32
//   chunks each: the taps of the 3x3 window around output pixel (2y + py, 2x + px) that fall on that class -- 2 per axis where the class bit differs
//   from the phase bit (high-resolution taps 0 and 2: a = 0, 1), 1 where it is equal (tap 1: a = 1 - py).  Class order (1-py, 1-px) [4 taps],
//   (1-py, px) [2], (py, 1-px) [2], (py, px) [1]: 9 * Cs / 32 steps, neighbours in the order share their cache lines.
// tap k of a chunk: a = a0 + k / nb, b = b0 + k % nb; the halo image (origin = low-resolution pixel (y0 - 1, x0 - 1) of the workgroup's patch) is read
// at rows + py + a, columns + px + b.
struct UpconvChunk {
  int skip, ch0, qy, qx, nt, nb, a0, b0, step0;
};
__host__ __device__ inline UpconvChunk upconv_chunk(int c, int py, int px, int cin_pad, int cs_pad) {
  const int kcx = cin_pad >> 5, kcs = cs_pad >> 5;
  UpconvChunk d;
  if (c < kcx || kcs == 0) {
    d.skip = 0; d.ch0 = c << 5; d.qy = 0; d.qx = 0; d.nt = 4; d.nb = 2; d.a0 = 0; d.b0 = 0; d.step0 = 4 * c;
    return d;
  }
  const int r = c - kcx, cls = r / kcs, cc = r - cls * kcs;
  d.skip = 1;
  d.ch0 = cc << 5;
  d.qy = cls < 2 ? 1 - py : py;
  d.qx = (cls & 1) ? px : 1 - px;
  const int na = d.qy != py ? 2 : 1;
  d.nb = d.qx != px ? 2 : 1;
  d.a0 = na == 2 ? 0 : 1 - py;
  d.b0 = d.nb == 2 ? 0 : 1 - px;
  d.nt = na * d.nb;
  const int cls_step0 = cls == 0 ? 0 : (cls == 1 ? 4 : (cls == 2 ? 6 : 8));
  d.step0 = 4 * kcx + cls_step0 * kcs + cc * d.nt;
  return d;
}
__host__ __device__ inline int upconv_chunks(int cin_pad, int cs_pad) { return (cin_pad >> 5) + 4 * (cs_pad >> 5); }
__host__ __device__ inline int upconv_steps(int cin_pad, int cs_pad) { return 4 * (cin_pad >> 5) + 9 * (cs_pad >> 5); }   // K steps of one phase
// packed position of weight element (phase, step, output channel co, channel i of the step's 32): [phase][step][CoutW][32] with the 16-byte pieces of
// a row XOR-swizzled by (co >> 2) & 3 -- the LDS image of a (CoutW x 32) tile, so the LDS-DMA copy is linear (as halo tiles 6 - 9, engine_dispatch.cpp)
inline size_t upconv_pack_index(int phase, int step, int co, int i, int steps, int coutw) {
  const int i_sw = ((((i & 31) >> 3) ^ ((co >> 2) & 3)) << 3) | (i & 7);
  return (((size_t)phase * steps + step) * coutw + co) * 32 + i_sw;
}
hipError_t launch_upconv(const UpconvParams& p, int shape, hipStream_t st);
// fp64-accumulating GEMM of the weight composition: C[g][m][n] = sum over the group's (A_i, B_i) pairs of sum_k A_i[m][k] * B_i[n][k] (fp32 in, fp64 out)
struct ComposeGemmParams {
  const float* a[4];   // [M][K]
  const float* b[4];   // [N][K]
  int pairs, M, N, K;
  double* c;           // [M][N]
};
hipError_t launch_compose_gemm(const ComposeGemmParams* groups_dev, int n_groups, int M, int N, hipStream_t st);
hipError_t launch_preprocess(const PreprocessParams& p, hipStream_t st);
hipError_t launch_pil_resample(const PilResampleParams& p, hipStream_t st);  // both passes
hipError_t launch_stem(const StemParams& p, hipStream_t st);
hipError_t launch_dwconv(const DwParams& p, hipStream_t st);
bool mbconv_front_supported(const MbFrontParams& p);
hipError_t launch_mbconv_front(const MbFrontParams& p, hipStream_t st);
hipError_t launch_letterbox(const LetterboxParams& p, hipStream_t st);
hipError_t launch_detect_decode_nms(const DetectParams& p, hipStream_t st);
bool mbconv_back_supported(const MbBackParams& p);
hipError_t launch_mbconv_back(const MbBackParams& p, hipStream_t st);
hipError_t launch_pool_partial(const PoolParams& p, hipStream_t st);
hipError_t launch_zero_u64(unsigned long long* p, size_t n, hipStream_t st);
// The fused average pool spreads its atomics over `replicas` rows: same-address atomics serialise in L2 (measured:
// 100+ per address cost 30 us on the 80x160 layers), so layers with many workgroups get up to 64 rows.
constexpr int kSeMaxReplicas = 64;
// squeeze-excite tail of an MBConv block in one launch: means, squeeze FC, excite FC, gate folded into the projection weights
hipError_t launch_se_gate_scale(const SeParams& se, const ScaleWParams& sw, hipStream_t st);
hipError_t launch_fc(const FcParams& p, hipStream_t st);
hipError_t launch_ctx_conv1(const CtxConv1Params& p, hipStream_t st);
hipError_t launch_fusion(const FusionParams& p, hipStream_t st);
hipError_t launch_decode_mask(const float* logits, int C, int HW, int mode, uint8_t* out, hipStream_t st);
hipError_t launch_resize_nearest(const uint8_t* src, int sw, const int* ytab, const int* xtab, int oh, int ow, uint8_t* dst,
                                 hipStream_t st);
hipError_t launch_resize_bilinear_f32(const float* src, int sw, const int* yi, const float* yf, const int* xi, const float* xf,
                                      int oh, int ow, float* dst, hipStream_t st);
hipError_t launch_minmax_f32(const float* src, size_t n, unsigned* mm, hipStream_t st);
constexpr int VP_PROBE_BLOCKS = 64;  // words of the range probe's verdict
hipError_t launch_finite_probe(const float* src, size_t n, unsigned* flags, hipStream_t st);  // flags[0..VP_PROBE_BLOCKS) = per-workgroup 0 / 1, overwritten every pass
hipError_t launch_depth_colorize(const float* src, size_t n, const unsigned* mm, const uint8_t* lut, uint8_t* dst, hipStream_t st);
hipError_t launch_viz_blend(const uint8_t* mask, int mw, const int* ytab, const int* xtab, const uint8_t* frame, int stride, int oh, int ow,
                            const uint8_t* lut, int frame_is_rgb, uint8_t* dst, hipStream_t st);
hipError_t launch_nchw_to_act(const float* src, int Creal, const ActView& a, hipStream_t st);
hipError_t launch_act_to_nchw(const ActView& a, int Creal, float* dst, hipStream_t st);

// ---- AutoDrive blocks (kernels_autodrive.hip)
// channel-slice copy: dst[pix][dst_off + c] = src[pix][src_off + c], c < nch (torch.cat / split / chunk along C)
hipError_t launch_chan_copy(const ActView& src, int src_off, const ActView& dst, int dst_off, int nch, hipStream_t st);
// MaxPool2d(5, stride 1, pad 2) on a channel slice (SPPF, common_layers.py:236-243)
hipError_t launch_maxpool5(const ActView& src, int src_off, const ActView& dst, int dst_off, int nch, hipStream_t st);
// SPPF's pyramid in one launch: dst slice 0 = src[0, nch), slices 1..3 = the 5x5 / 9x9 / 13x13 clipped window maxima (= three chained MaxPool2d(5, 1, 2))
bool sppf_pool_ok(const ActView& src, const ActView& dst, int nch);   // maps of at most 512 pixels (one workgroup holds an octet's map)
hipError_t launch_sppf_pool(const ActView& src, const ActView& dst, int nch, hipStream_t st);
struct AttnParams {
  ActView qkv;   // [HW][heads * (2*dk + dv)]: per head q(dk) | k(dk) | v(dv)   (Attention.forward, common_layers.py:95-99)
  ActView out;   // [HW][heads * dv] = v @ softmax(q^T k * scale)^T
  ActView vout;  // [HW][heads * dv] copy of v (input of the depthwise positional conv)
  int heads, dk, dv;
  float scale;
  int qblock;    // 4: four query tokens per workgroup (attention_block_kernel: dk = 32, dv = 64, <= 1760 tokens); 0: one workgroup per query token
};
bool attention_block_ok(const AttnParams& p);
hipError_t launch_attention(const AttnParams& p, hipStream_t st);
struct DwPlainParams {
  ActView in, out;   // out = add + dwconv3x3(in) (BN folded, no activation), common_layers.py:103
  ActView add;
  const float* w;    // [9][C]
  const float* b;    // [C]
};
hipError_t launch_dwconv_plain(const DwPlainParams& p, hipStream_t st);

}  // namespace vp
