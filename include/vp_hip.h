/*
 * vp_hip.h -- C ABI of libvp_hip.so: MI355X-native (gfx950) engine for the VisionPilot per-frame hot path
 *             preprocess -> SceneSeg / Scene3D / DomainSeg / EgoLanes forward -> argmax / threshold decode.
 *
 * This is the drop-in boundary.  Plain pointers and sizes only; no exceptions cross it.
 * Reference interfaces each entry point replaces (paths relative to the reference repo):
 *
 *   vp_create / vp_create_from_memory
 *       OnnxRuntimeBackend::OnnxRuntimeBackend(model_path, precision, gpu_id)
 *         VisionPilot/middleware_recipes/common/backends/onnx_runtime_backend.cpp:9-39
 *       TensorRTBackend ctor            .../common/backends/tensorrt_backend.cpp:35-60
 *       EgoLanesOnnxEngine ctor         VisionPilot/production_release/src/inference/onnxruntime_engine.cpp:13-66
 *       SceneSegNetworkInfer.__init__   Models/inference/scene_seg_infer.py:12-36
 *   vp_infer
 *       InferenceBackend::doInference(const cv::Mat&)  .../common/include/inference_backend_base.hpp:19,
 *         onnx_runtime_backend.cpp:41-82 (preprocess + Run)
 *       EgoLanesOnnxEngine::doInference onnxruntime_engine.cpp:104-135 (+ preprocessEgoLanes :72-102)
 *   vp_infer_tensor
 *       XInfer.inference() after ToTensor/Normalize   Models/inference/scene_seg_infer.py:44-49
 *   vp_logits
 *       InferenceBackend::getRawTensorData / getTensorShape   inference_backend_base.hpp:22-23
 *   vp_input_hw
 *       getModelInputHeight / getModelInputWidth              inference_backend_base.hpp:25-26
 *   vp_mask_u8 (VP_DECODE_*)
 *       RunModelNode::onImage argmax / threshold loops  ROS2/models/src/run_model_node.cpp:144-171
 *       MasksVisualizationKernels::createMaskFromTensor{CUDA,HIP}, createEgoLanesMaskFromTensorCUDA
 *         common/visualizers/cuda_visualization_kernels.cu:13-75, masks_viz.hip.cpp:11-97
 *       torch.max(dim=2)                                 Models/inference/scene_seg_infer.py:52-55
 *   vp_mask_resized_u8 / vp_depth_resized_f32
 *       cv::resize(mask, INTER_NEAREST) / cv::resize(depth, INTER_LINEAR)  run_model_node.cpp:104,177
 *
 * Threading: one engine per caller thread; an engine owns its HIP stream, every buffer it uses and its kernel plan, so several
 * engines may live in one process and be created / driven from different threads concurrently.  Process-wide
 * state, all of it optional and each behind its own mutex: (1) the developer options of vp_set_option (read when an engine's plan
 * is built; no production path sets one); (2) the frame pools of vp_register_frames (a table of page-locked ranges, consulted per
 * upload); (3) RCCL's dlopen handle (vp_comm_*).  Nothing an engine computes depends on another engine's existence.
 * All functions return 0 on success and a negative vp_status on failure; vp_last_error() gives the text.
 */
#ifndef VP_HIP_H_
#define VP_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct vp_engine vp_engine;

enum vp_status { VP_OK = 0, VP_ERR_ARG = -1, VP_ERR_WEIGHTS = -2, VP_ERR_HIP = -3, VP_ERR_STATE = -4,
                 VP_ERR_RANGE = -5 /* a value left the fp16 range the matrix pipe carries: see vp_set_finite_check */ };

/* VP_AUTODRIVE (BASELINE configs[4]): Models/model_components/autodrive/autodrive_network.py:32-36 --
 * shared backbone on the previous and the current frame (network input 1x3x512x1024, RGB planes), head -> three scalars
 * returned through vp_logits as shape {1,3,1,1} = (d_norm, curvature, flag_logit); no mask. */
enum vp_model_kind { VP_SCENESEG = 0, VP_SCENE3D = 1, VP_DOMAINSEG = 2, VP_EGOLANES = 3, VP_AUTODRIVE = 4 };

/* VP_FP16  : fp16 tensors, fp16 MFMA, fp32 accumulate (the reference's "fp16" configuration).
 * VP_FP16X3: fp32-class accuracy on the fp16 matrix pipe: every tensor is a (hi, lo) fp16 pair and every
 *            product is three MFMAs (hi*hi + hi*lo + lo*hi), fp32 accumulate -- the parity mode (1e-3). */
enum vp_precision { VP_FP16 = 0, VP_FP16X3 = 1,
                    /* OR-able flag: every conv / linear weight is quantised at load to per-output-row OCP e4m3 (fp8) (BASELINE
                     * configs[4] "fp8 weights"; the reference's PTQ flow: Models/exports/quantization/PTQ/AutoDrive).  Round 4: it is
                     * STORAGE, not only numerics -- on the kernels that stage weights through registers (the generic GEMM kernel, the
                     * 3x3 halo kernel, the FC kernel: every matrix / FC layer of VP_AUTODRIVE) HBM holds one e4m3 byte per weight plus
                     * a per-row fp32 scale; the staging path converts 8 codes -> 8 fp16 values on their way to LDS (exact: every e4m3
                     * value is an fp16 value) and the epilogue applies the scale to the fp32 accumulator.  The arithmetic stays the
                     * fp16 matrix pipe with fp16 / (hi, lo) activations (fp8 MFMA would quantise the ACTIVATIONS to 3 mantissa bits:
                     * outside the 1e-3 bar).  Layers on the LDS-DMA / register-stationary kernels of the scene networks keep
                     * de-quantised fp16 planes.  vp_weight_bytes reports the split. */
                    VP_WEIGHTS_FP8 = 16,
                    /* OR-able flag (round 6): the LATENCY kernel plan for THIS engine.  The default plan spends a layer's own latency where the rest
                     * of the frame can use the CUs it frees (forked heads on one camera, several cameras per GPU); a host that runs ONE network on
                     * ONE camera, one frame at a time, has nothing to put there and gets 3-7 % of its frame back with this flag.  Per engine: a
                     * process that holds several engines (VisionPilot/production_release/main.cpp:505-535: EgoLanes beside AutoSpeed / AutoSteer)
                     * chooses for each; results differ from the default plan's only in fp32 summation order; vp_plan_hash() tells the plans apart.
                     * A shared-prefix engine may choose differently from its base. */
                    VP_PLAN_LATENCY = 32 };

enum vp_pixel_format { VP_BGR8 = 0, VP_RGB8 = 1 };
/* Plane order of the network input: BGR planes = middleware "common" backends (onnx_runtime_backend.cpp:47-57),
 * RGB planes = EgoLanes engines and the Python wrappers.  Normalisation constants follow the plane's colour. */
enum vp_plane_order { VP_PLANES_BGR = 0, VP_PLANES_RGB = 1 };

enum vp_decode_mode {
  VP_DECODE_SEG_MASK = 0,   /* C>1: 255 where argmax==1 else 0 ; C==1: 255 where logit>0 */
  VP_DECODE_LANE_LABEL = 1, /* {2,1,0,255} priority label */
  VP_DECODE_CLASS_INDEX = 2 /* raw argmax index (first maximum wins) */
};

/* ---- lifetime ------------------------------------------------------------------------------------------
 * weights_path: a VPW1 blob (autoware_vision_pilot_amd/weights.py) or -- as the reference's backends take
 * (`model_path: *.onnx`, ROS2/models/config/autoseg.yaml:3; OnnxRuntimeBackend ctor onnx_runtime_backend.cpp:14-38) --
 * an `.onnx` file made by Models/exports/convert_pytorch_to_onnx.py:144-154.  ONNX files are read by the library's own
 * wire-format parser (csrc/onnx_reader.cpp): weights only, the graph is the engine's. */
int vp_create(vp_engine** out, int model_kind, const char* weights_path, int precision, int gpu_id, char* err, size_t err_len);
int vp_create_from_memory(vp_engine** out, int model_kind, const void* blob, size_t blob_bytes, int precision, int gpu_id,
                          char* err, size_t err_len);
void vp_destroy(vp_engine* e);
const char* vp_last_error(const vp_engine* e);
/* Host only (no HIP device needed): ONNX file -> VPW1 blob file, the conversion vp_create does in memory. */
int vp_convert_onnx(const char* onnx_path, const char* vpw_path, char* err, size_t err_len);

/* ---- shared-prefix engines (BASELINE configs[2]: SceneSeg + Scene3D + EgoLanes on one camera) -----------
 * The reference builds Scene3D and DomainSeg ON a pre-trained SceneSeg: Scene3DNetwork wraps its backbone
 * (Models/model_components/scene_3d_network.py:9-13, pre_trained_backbone.py), DomainSegNetwork its backbone +
 * context + neck (domain_seg_network.py:9-12, domain_seg_upstream.py), but every ONNX/TensorRT backend instance
 * re-runs that prefix per model.  vp_create_shared builds an engine whose plan STARTS from the base engine's feature
 * tensors: sub-networks whose parameters are byte-identical to the base's are not rebuilt or re-run
 * (vp_shared_level: 1 = backbone, 2 = backbone + context + neck).  It runs on the base engine's stream:
 *     vp_infer(base, frame, ...);  vp_infer_shared(head2);  vp_infer_shared(head3);
 * Errors: VP_ERR_ARG if the backbone parameters differ, if precision / gpu differ from the base, or if the base is
 * itself a shared engine.  The base must outlive its shared engines; a shared engine accepts no frames of its own. */
int vp_create_shared(vp_engine** out, vp_engine* base, int model_kind, const char* weights_path, int precision, int gpu_id,
                     char* err, size_t err_len);
int vp_create_shared_from_memory(vp_engine** out, vp_engine* base, int model_kind, const void* blob, size_t blob_bytes,
                                 int precision, int gpu_id, char* err, size_t err_len);
int vp_shared_level(const vp_engine* e);
int vp_infer_shared(vp_engine* e);

/* ---- batched encoder (multi-camera rigs, BASELINE configs[2]/[3]: several cameras per GPU) -----------------
 * Every backend instance of the reference runs batch 1 (onnx_runtime_backend.cpp:59, tensorrt_backend.cpp:124), one
 * instance per camera.  The EfficientNet encoder is ~100 latency-sized launches per frame; a batched encoder engine runs
 * preprocess + encoder for `frames` cameras per pass (activations [frames][H][W][C]: 1x1 convolutions as one GEMM over all
 * pixels, depthwise + squeeze-excite kernels with the frame on a grid axis, the SE-gated projections per frame) and owns no
 * context / neck / head.  Each camera's head is a shared-prefix engine on its slot:
 *     vp_create_batched(&enc, VP_SCENESEG, "SceneSeg.onnx", VP_FP16, gpu, 3, ...);
 *     for f in 0..2: vp_create_shared_frame(&head[f], enc, f, VP_SCENESEG, "SceneSeg.onnx", VP_FP16, gpu, ...);
 *     per pass: vp_upload_frame_n(enc, f, frame_f, h, w, stride) x3;  vp_enqueue(enc);  vp_infer_shared(head[f]) x3.
 * All frames of a pass share one geometry.  Errors as for vp_create_shared; frames in 1..16.  Per camera the results equal
 * the single-frame engine's up to fp32 summation order (the split-K factor of a 1x1 GEMM follows its pixel count, which the
 * batch multiplies; measured bit-identical on the tested networks, the tests hold them to 1e-3).  Timed in DESIGN.md.
 * vp_upload_frame* stages pageable frames through two pinned slots: a slot is rewritten only after the H2D copy that last read it
 * has completed (per-slot event), so any number of uploads may be queued without a synchronisation in between. */
int vp_create_batched(vp_engine** out, int model_kind, const char* weights_path, int precision, int gpu_id, int frames, char* err,
                      size_t err_len);
int vp_create_batched_from_memory(vp_engine** out, int model_kind, const void* blob, size_t blob_bytes, int precision, int gpu_id,
                                  int frames, char* err, size_t err_len);
int vp_create_shared_frame(vp_engine** out, vp_engine* base, int frame_index, int model_kind, const char* weights_path, int precision,
                           int gpu_id, char* err, size_t err_len);
int vp_create_shared_frame_from_memory(vp_engine** out, vp_engine* base, int frame_index, int model_kind, const void* blob,
                                       size_t blob_bytes, int precision, int gpu_id, char* err, size_t err_len);
int vp_frames(const vp_engine* e);
int vp_upload_frame_n(vp_engine* e, int index, const uint8_t* frame, int h, int w, int stride_bytes);

/* ---- configuration ------------------------------------------------------------------------------------- */
int vp_set_input_format(vp_engine* e, int pixel_format, int plane_order);
int vp_set_decode_mode(vp_engine* e, int decode_mode);
/* How a frame of another size reaches the network's input size.  VP_RESIZE_CV_LINEAR (default of the scene networks): the integer
 * bilinear modelled on cv::resize INTER_LINEAR, what the C++ nodes do (onnx_runtime_backend.cpp:44).  VP_RESIZE_PIL_BILINEAR
 * (default of VP_AUTODRIVE) / VP_RESIZE_PIL_BICUBIC: Pillow's antialiased Image.resize, bit-exact against Pillow 12.2 -- the
 * Python scripts' frame path (Models/visualizations/AutoDrive/video_visualization.py:29-33 BILINEAR; the scene networks'
 * visualisation scripts call Image.resize((640, 320)) with Pillow's default filter, BICUBIC).  VP_ERR_ARG on a shared engine. */
enum vp_resize_mode { VP_RESIZE_CV_LINEAR = 0, VP_RESIZE_PIL_BILINEAR = 1, VP_RESIZE_PIL_BICUBIC = 2 };
int vp_set_resize_mode(vp_engine* e, int resize_mode);
int vp_get_resize_mode(const vp_engine* e);
/* host only: one axis' tap tables of the VP_RESIZE_PIL_* modes -- bounds[out][2] = {first source index, taps}, coeffs[out][ksize] with
 * 22 fractional bits; returns ksize (> 0) or VP_ERR_ARG (coeffs_cap < out_size * ksize ints).  For tests / external checks. */
int vp_resample_coeffs(int in_size, int out_size, int resize_mode, int* bounds, int* coeffs, int coeffs_cap);
/* The u8 -> [0, 1] step ahead of (x - mean) / std exists in two spellings that differ in the last bit for 322 of the 768 (byte, channel)
 * pairs (<= 7.2e-7): VP_NORM_TORCHVISION q / 255 -- torchvision's to_tensor, the Python operator API (Models/inference/scene_seg_infer.py:
 * 15-20; default) -- and VP_NORM_OPENCV q * fl(1/255) -- cv::Mat::convertTo(CV_32FC3, 1.0 / 255.0) of the C++ front-ends
 * (onnx_runtime_backend.cpp:45-49, tensorrt_backend.cpp:164-168, production_release onnxruntime_engine.cpp:85,94-100), which
 * adapters/hip_backend.hpp and adapters/egolanes_hip_engine.hpp select.  VP_ERR_ARG on a shared engine. */
enum vp_norm_form { VP_NORM_TORCHVISION = 0, VP_NORM_OPENCV = 1 };
int vp_set_norm_form(vp_engine* e, int norm_form);
int vp_get_norm_form(const vp_engine* e);
int vp_get_decode_mode(const vp_engine* e);      /* the vp_decode_mode in force (>= 0), or VP_ERR_ARG */
int vp_gpu_id(const vp_engine* e);               /* the device the engine lives on */
int vp_device_count(void);                       /* GPUs visible to this process (0 = none): a thread-per-GPU host sizes its world with it */
int vp_host_logits_current(const vp_engine* e);  /* 1 = the host pointer vp_logits() hands out holds the LAST pass's logits */
int vp_input_hw(const vp_engine* e, int* h, int* w);

/* Which outputs vp_infer / vp_infer_shared / vp_infer_multi copy to the host before they return (default both, what
 * TensorRTBackend::doInference does: tensorrt_backend.cpp:184-199).  A de-selected output stays in HBM and is copied on the
 * first vp_logits / vp_mask_u8 call instead: a host that only consumes the decoded mask (RunModelNode publishes the mask,
 * run_model_node.cpp:173-190) saves the 2.4 MB fp32 logits transfer per frame. */
enum vp_output_bits { VP_OUT_LOGITS = 1, VP_OUT_MASK = 2 };
int vp_set_outputs(vp_engine* e, int output_bits);
/* vp_infer stages the caller's pageable frame through a pinned double buffer owned by the engine (default 1). */
int vp_set_pinned_staging(vp_engine* e, int enable);
/* FRAME POOLS (round 5).  The staging above is a single-threaded host memcpy of the whole frame (2.76 MB at 1280x720: ~130 us of a 3.5 ms
 * node frame).  A host that owns its frame buffers (a cv::Mat pool, a camera ring, a ROS message arena) registers them ONCE:
 *   vp_register_frames(pool, bytes)  page-locks [pool, pool + bytes) for DMA (hipHostRegister, portable across devices); process-wide;
 * and every vp_infer* / vp_upload_frame* whose frame lies wholly inside a registered range is then copied by ONE DMA straight from the
 * caller's memory -- no staging copy.  Contract for the asynchronous calls (vp_upload_frame*): such a frame must stay unmodified until the
 * pass that consumes it has been synchronised (vp_sync / vp_fetch_outputs / vp_infer*); the staged path copied it before returning, this one
 * does not.  The synchronous vp_infer* calls return after the copy, as before.  Frames outside every registered range take the staged path.
 * What TensorRTBackend does with its own pinned input buffer (tensorrt_backend.cpp:184-186), offered to the caller's buffers instead.
 * vp_unregister_frames(pool) releases a range registered with exactly that pointer (after the last pass that read it was synchronised).
 * Returns VP_ERR_ARG (null / zero / overlapping an existing range / unknown pointer) or VP_ERR_HIP (the runtime refused to lock the pages).
 * Device context: the range is registered PORTABLE (usable from every device); the call itself runs on the calling thread's current HIP device
 * (device 0 for a thread that never called hipSetDevice) -- a host may call it before or after vp_create, from any thread. */
int vp_register_frames(const void* pool, size_t bytes);
int vp_unregister_frames(const void* pool);
/* {1, C, H, W} of the output tensor WITHOUT touching the device: vp_logits also returns the shape but fetches a de-selected tensor
 * (vp_set_outputs) on the way.  Same state rule as vp_logits: VP_ERR_STATE before the first inference (onnx_runtime_backend.cpp:86-91). */
int vp_output_shape(const vp_engine* e, int64_t shape[4]);
/* RANGE GUARD.  Both precision modes compute on fp16 planes: VP_FP16X3 has fp32-class SIGNIFICAND (hi + lo) but fp16 EXPONENT range.
 *   - at load: a BN-folded weight with |w| > 65504 (or non-finite) fails vp_create* with VP_ERR_RANGE instead of loading as inf;
 *   - per frame: an activation beyond 65504 becomes inf and reaches the logits as inf / NaN; a probe kernel over the logits at the end
 *     of every pass (3 us, default ON) raises a flag that travels with the outputs, and the call that synchronises on that pass --
 *     vp_infer*, vp_fetch_outputs, vp_sync -- returns VP_ERR_RANGE (once per offending frame; the outputs of that frame are invalid).
 * The reference computes in fp32 and has no such limit; networks whose activations stay below 65504 (trained, normalised ones do by
 * orders of magnitude: |x| < 600 over the seeded sweeps of tests/test_gpu_parity_sweep.py) are unaffected.  enable = 0 drops the probe. */
int vp_set_finite_check(vp_engine* e, int enable);

/* ---- synchronous per-frame path (host buffers in, host buffers out) -------------------------------------- */
int vp_infer(vp_engine* e, const uint8_t* frame, int h, int w, int stride_bytes);
/* One frame through a base engine AND its shared-prefix heads (BASELINE metric: SceneSeg + Scene3D on one camera): one H2D,
 * all networks enqueued back to back on the base engine's stream, the selected outputs of every engine copied D2H behind
 * them, ONE host synchronisation -- instead of vp_infer + n x vp_infer_shared (n + 1 synchronisations). */
int vp_infer_multi(vp_engine* base, vp_engine* const* shared, int n_shared, const uint8_t* frame, int h, int w, int stride_bytes);
/* Asynchronous half of it for a frame that is already resident (vp_upload_frame): the base engine and its shared-prefix heads as
 * ONE graph launch on the base engine's stream.  Inside the graph the heads that consume only the backbone (Scene3D, EgoLanes on a
 * SceneSeg base) are forked behind it and joined at the end, so one frame's decoders overlap; results are bit-identical to
 * vp_enqueue(base) followed by vp_enqueue(head) for every head, and so is the stream order seen from outside. */
int vp_enqueue_multi(vp_engine* base, vp_engine* const* shared, int n_shared);
/* The fork is a LATENCY lever (measured on MI355X, SceneSeg + Scene3D: one camera 3.49 -> 3.19 ms per frame; three heads 4.47 ->
 * 3.99 ms) and a throughput loss when several cameras are already in flight on one GPU (three cameras: 390 -> 333 frames/s: the
 * chip is full, more concurrent kernels only contend).  Default 1 (the reference node is synchronous, one frame at a time);
 * a host that pipelines several cameras per GPU sets 0: vp_enqueue_multi / vp_infer_multi then enqueue one engine after the other. */
int vp_set_multi_fork(vp_engine* base, int enable);
int vp_infer_tensor(vp_engine* e, const float* nchw_1x3x320x640);
/* AutoDrive.forward(image_prev, image_curr) (autodrive_network.py:32-36): backbone on `prev`, then backbone + head on `curr`.
 * Plain vp_infer on a VP_AUTODRIVE engine is the streaming form: the previous call's frame is `prev` (the first frame of
 * a stream pairs with itself).  Both frames share h, w, stride. */
int vp_infer_pair(vp_engine* e, const uint8_t* prev, const uint8_t* curr, int h, int w, int stride_bytes);
int vp_logits(const vp_engine* e, const float** data, int64_t shape[4]);       /* host pointer, valid until next infer */
int vp_mask_u8(const vp_engine* e, const uint8_t** data, int* h, int* w);      /* host pointer, network resolution */
/* MasksVisualizationKernels::createMaskFromTensor{CUDA,HIP}(const float* host_tensor, shape, cv::Mat&) and
 * createEgoLanesMaskFromTensorCUDA (common/include/masks_visualization_kernels.hpp:14-45; Zenoh models/run_model.cpp): stateless,
 * HOST logits in, u8 mask out at tensor resolution, decode_mode as vp_decode_mode.  adapters/masks_visualization_kernels_hip.hpp
 * gives it the reference's exact signature (and skips the upload when the tensor is an engine's own logits). */
int vp_decode_logits_host(int gpu_id, const float* logits_nchw, int channels, int h, int w, int decode_mode, uint8_t* mask_out);
int vp_mask_resized_u8(vp_engine* e, uint8_t* dst, int h, int w);              /* nearest, to frame size */
int vp_depth_resized_f32(vp_engine* e, float* dst, int h, int w);              /* bilinear, plane 0 of the logits */
/* MasksVisualizationEngine::visualize (middleware_recipes/common/visualizers/masks_visualization_engine.cpp:11-58, SURVEY.md
 * 8f N4): colour LUT on the last mask (viz_type: 0 "scene", 1 "domain", 2 "egolanes" -- use VP_DECODE_LANE_LABEL for it),
 * nearest resize to the frame of the last vp_infer, 50/50 blend with that frame.  dst: BGR8 [frame_h][frame_w][3], packed. */
enum vp_viz_type { VP_VIZ_SCENE = 0, VP_VIZ_DOMAIN = 1, VP_VIZ_EGOLANES = 2 };
/* h, w: geometry of dst; VP_ERR_ARG unless it equals the last inferred frame's (the reference reads the size off original_image). */
int vp_visualize_mask_bgr8(vp_engine* e, int viz_type, uint8_t* dst_bgr8, int h, int w);
int vp_frame_hw(const vp_engine* e, int* h, int* w);                           /* geometry of the last uploaded frame */
/* DepthVisualizationEngine::visualize (middleware_recipes/common/visualizers/depth_visualization_engine.cpp:9-26): plane 0 of
 * the logits bilinear-resized to h x w (the map vp_depth_resized_f32 returns, run_model_node.cpp:100-104), min-max
 * normalised to u8 (convertTo with alpha = 255/(max-min), beta = -min*alpha; all zero if max == min) and mapped through
 * COLORMAP_VIRIDIS.  dst: BGR8 [h][w][3], packed.  Table and convertTo rounding restated from published sources, not
 * pinned against OpenCV (absent here) -- see DESIGN.md. */
int vp_visualize_depth_bgr8(vp_engine* e, uint8_t* dst_bgr8, int h, int w);
int vp_input_tensor(vp_engine* e, float* dst_1x3x320x640);                     /* the post-resize network input */

/* ---- AutoSteer hand-over (VisionPilot/production_release/main.cpp:472-534) ---------------------------
 * The production app feeds its AutoSteer head the raw EgoLanes logits of the last TWO frames, concatenated [t-1 | t] = fp32 {1, 6, 80, 160}:
 * a boost::circular_buffer of host copies plus two memcpys per frame (main.cpp:508-530), handed to AutoSteerOnnxEngine::inference
 * (src/inference/autosteer_engine.cpp:104-195).  With vp_set_lane_ring(e, 1) a VP_EGOLANES engine keeps that buffer ON THE DEVICE: two
 * device-to-device copies behind every pass (inside the replayed graph), no host round trip -- a device-resident AutoSteer runtime reads
 * vp_lane_ring_device, the reference's host-side engine reads vp_lane_ring_fetch.  *frames_valid = passes since enabling, capped at 2: the
 * reference runs AutoSteer only once two frames are in (main.cpp:521).  The AutoSteer GRAPH itself is not in the reference tree (ONNX only,
 * production_release/README.md:112): the head is not built, its input is.  VP_ERR_ARG on other model kinds, VP_ERR_STATE while disabled. */
int vp_set_lane_ring(vp_engine* e, int enable);
int vp_lane_ring_device(const vp_engine* e, void** dev_f32_6x80x160, int* frames_valid);
int vp_lane_ring_fetch(vp_engine* e, float* dst_host_6x80x160, int* frames_valid);
/* The other end of that head, AutoSteerOnnxEngine::postProcess (autosteer_engine.cpp:160-185): arg-max over the `classes` (61) logits of the head's
 * SECOND output, strict '>' from class 0 (the first maximum wins, a NaN never wins), steering angle = class - 30 degrees.  Host arithmetic on 61 floats
 * (no device work): offered so that a host replacing the engine class keeps the decode bit for bit.  Returns 0 (the reference's failure value) for
 * logits == NULL or classes < 1. */
float vp_autosteer_angle(const float* logits, int classes);

/* ---- asynchronous / device-resident path (bench, multi-GPU) ---------------------------------------------- */
int vp_upload_frame(vp_engine* e, const uint8_t* frame, int h, int w, int stride_bytes); /* H2D, frame stays in HBM */
int vp_enqueue(vp_engine* e);                                                  /* one pass over the resident frame */
int vp_sync(vp_engine* e);
int vp_fetch_outputs(vp_engine* e);                                            /* D2H logits + mask, then sync */
int vp_device_outputs(const vp_engine* e, void** logits_f32, void** mask_u8);  /* device pointers (D2D gathers) */
int vp_copy_outputs_device(vp_engine* e, void* logits_dst, void* mask_dst);   /* async D2D on the engine stream */
int vp_use_graph(vp_engine* e, int enable);                                    /* hipGraph replay (default on) */
int vp_timer_begin(vp_engine* e);                                              /* hipEvent on the engine stream */
int vp_timer_end(vp_engine* e, float* elapsed_ms);                             /* records, syncs, returns elapsed */

/* ---- multi-camera result exchange (BASELINE configs[3]) ---------------------------------------
 * One process (or thread) per GPU, camera r on rank r, weights replicated, no data-path collective in the networks.  For the
 * fused ego-path / BEV consumer every rank needs all cameras' results: ONE ncclAllGather (RCCL over xGMI) per frame of a
 * fixed-size record -- the u8 mask (320x640 = 204.8 KB; EgoLanes label map 80x160 = 12.8 KB) or the fp32 logits (EgoLanes
 * 3x80x160 = 153.6 KB) -- enqueued on the ENGINE's stream behind the frame's graph, no host synchronisation:
 *     rank 0: vp_comm_unique_id(id);  host distributes the 128 bytes (ROS parameter / file / MPI / ...)
 *     all   : vp_comm_create(&c, id, rank, world, gpu, record_bytes_max, ...)
 *     frame : vp_enqueue(e);  vp_gather(e, c, VP_GATHER_MASK);            // overlaps the next in-flight frame's encoder
 *     read  : vp_comm_device_buffer (device consumer) or vp_comm_fetch (host copy, synchronises).
 * RCCL is bound with dlopen at the first vp_comm_* call (librccl.so is not a load-time dependency of libvp_hip.so).
 * One communicator per engine in flight: RCCL operations on one communicator must not run concurrently on two streams -- ENFORCED:
 * vp_gather with an engine on another stream while the communicator's previous all-gather has not completed returns VP_ERR_STATE.
 * The reference has no counterpart (single camera per backend instance). */
typedef struct vp_comm vp_comm;
#define VP_COMM_ID_BYTES 128
enum vp_gather_what { VP_GATHER_MASK = 0, VP_GATHER_LOGITS = 1 };
int vp_comm_unique_id(void* id_out_128, char* err, size_t err_len);
int vp_comm_create(vp_comm** out, const void* unique_id_128, int rank, int world, int gpu_id, size_t record_bytes_max, char* err,
                   size_t err_len);
void vp_comm_destroy(vp_comm* c);
const char* vp_comm_last_error(const vp_comm* c);
int vp_comm_rank(const vp_comm* c);
int vp_comm_world(const vp_comm* c);
int vp_gather(vp_engine* e, vp_comm* c, int what);                              /* async, on the engine's stream */
int vp_comm_device_buffer(const vp_comm* c, void** dev, size_t* record_bytes);  /* [world][record_bytes] */
int vp_comm_fetch(vp_comm* c, vp_engine* e, const void** host, size_t* record_bytes); /* D2H + sync; pinned, valid until next fetch */

/* ---- introspection (profiling, per-layer parity tests) --------------------------------------------------- */
int vp_layer_count(const vp_engine* e);
int vp_layer_info(const vp_engine* e, int i, const char** name, double* flops, double* bytes);
int vp_layer_kernel(const vp_engine* e, int i, const char** kernel_tag);        /* kernel instantiation of launch i */
int vp_layer_launch(const vp_engine* e, int i, const char** launch);            /* its geometry beyond the tag: "nsplit=4", "groups=32", "" */
/* vp_layer_info's flops count the REFERENCE formulation of the layer(s) a launch computes (one multiply-add per product of the
 * reference's operators).  A composed up-sampling stage (round 6: ConvTranspose [+ skip link] -> 3x3 multiplied out at load) executes 0.40-0.51x
 * of that; this returns what the launch executes (= vp_layer_info's figure for every other launch). */
int vp_layer_flops_executed(const vp_engine* e, int i, double* flops);
int vp_profile_layers(vp_engine* e, int iters, float* ms_per_layer, int capacity); /* eager, event per launch */
int vp_tensor_count(const vp_engine* e);
int vp_tensor_info(const vp_engine* e, int i, const char** name, int* c, int* h, int* w);
int vp_tensor_read(vp_engine* e, int i, float* dst_chw);                       /* fp32 CHW copy of an activation */

/* ---- AutoSpeed detector: letterbox preprocess and decode + NMS on the device ------------------------------------
 * Replaces AutoSpeedOnnxEngine::preprocessAutoSpeed (VisionPilot/middleware_recipes/common/backends/autospeed/
 * onnxruntime_engine.cpp:71-113) and ::postProcess / ::computeIoU / ::applyNMS (:170-290); the TensorRT engine (tensorrt_engine.cpp)
 * holds the same two stages.  The detector network is NOT part of this library: the host runs it between the two calls and hands its
 * output tensor [num_attrs][num_boxes] (cx, cy, w, h in letterbox pixels, then the class scores) to vp_detect_postprocess.
 * vp_detection is the reference's `Detection` (autospeed/detection.hpp:8-12), field for field.  A handle owns its stream and buffers and is
 * not thread-safe (one handle per calling thread, as the reference has one engine object per thread); every call returns synchronised.
 * With raw_on_device the caller's producer must have finished writing the tensor (the call does not wait on foreign streams). */
typedef struct vp_detect vp_detect;
typedef struct vp_detection {
  float x1, y1, x2, y2; /* image coordinates, clamped to the frame */
  float confidence;
  int class_id;         /* -1: no class score was positive (the reference's argmax starts from 0) */
} vp_detection;
/* net_h x net_w: the detector's input (640 x 640); max_boxes <= 16384 and max_attrs >= 5 bound the output tensors accepted later */
int vp_detect_create(vp_detect** out, int gpu_id, int net_h, int net_w, int max_boxes, int max_attrs, char* err, size_t err_len);
void vp_detect_destroy(vp_detect* d);
const char* vp_detect_last_error(const vp_detect* d);
/* BGR8 frame -> [3][net_h][net_w] fp32 planes R, G, B in [0, 1]: resize keeping the aspect (scale = min(net_w / w, net_h / h), new size
 * truncated), centred on a canvas of 114s.  The tensor stays on the device (vp_detect_input_device); dst_host_chw may be NULL.  Remembers
 * the letterbox geometry for the next vp_detect_postprocess (the reference's scale_, pad_x_, pad_y_, orig_width_, orig_height_). */
int vp_detect_preprocess(vp_detect* d, const uint8_t* bgr, int h, int w, int stride_bytes, float* dst_host_chw);
int vp_detect_input_device(const vp_detect* d, void** dev_f32_chw);
int vp_detect_letterbox(const vp_detect* d, float* scale, int* pad_x, int* pad_y);
int vp_detect_set_letterbox(vp_detect* d, float scale, int pad_x, int pad_y, int orig_w, int orig_h); /* geometry from elsewhere */
/* raw: host pointer, or device pointer when raw_on_device != 0.  Per box: strict-'>' argmax of the class scores from 0, `score <
 * conf_thresh` dropped, box back to image coordinates and clamped; then sorted by confidence (descending; equal confidences keep the box
 * order, which the reference's std::sort leaves unspecified) and greedily suppressed per class at IoU > iou_thresh.  *count = detections
 * kept; min(*count, out_cap) of them are written, in the reference's output order. */
int vp_detect_postprocess(vp_detect* d, const float* raw, int raw_on_device, int num_attrs, int num_boxes, float conf_thresh, float iou_thresh,
                          vp_detection* out, int out_cap, int* count);

/* ---- single-operator entry (unit parity tests of the MFMA conv kernel; not used by callers) -------------- */
/* mode 0: Conv2d k=ks stride 1 pad ks/2, weight [Cout][Cin][ks][ks]; mode 1: ConvTranspose2d k2 s2, weight
 * [Cin][Cout][2][2].  act: 0 none 1 GELU 2 SiLU.  res_mode: 0 none 1 add 2 mul-add; res has the output shape.
 * mode 2: ConvTranspose2d k2 s2 of `in` + Conv2d 1x1 of a skip tensor in ONE launch (the decoders' up-sample + skip-link
 * pairs): `res` is the skip INPUT [res_mode channels][2h][2w], weight = ConvTranspose weight followed by the 1x1 weight
 * [Cout][res_mode], bias = the two bias vectors one after the other; act / tile / bk / nsplit are ignored.
 * mode 3: a head's last convolution (3x3, no activation): the kernel writes fp32 NCHW logits itself.
 * tile/bk/nsplit: -1 = engine heuristic.  in/res/out are fp32 CHW host buffers. */
int vp_op_conv2d(int gpu_id, int precision, int mode, const float* in, int cin, int h, int w, const float* weight, const float* bias,
                 int cout, int ks, int act, int res_mode, const float* res, int tile, int bk, int nsplit, float* out,
                 char* err, size_t err_len);

/* ---- composed up-sampling stage (round 6; unit parity tests, not used by callers) --------------------------
 * The decoders apply ConvTranspose2d(k2, s2) [+ Conv2d 1x1 of a skip tensor] and then Conv2d 3x3 with NO nonlinearity in between
 * (Models/model_components/scene_neck.py:29-35,41-46,52-57; scene_seg_head.py:24-29,35-38; scene_3d_head.py:26-31,38-41): one linear map.
 * The engine multiplies the three weight sets out at load (fp64 accumulation) and runs the stage as ONE launch on the LOW-resolution
 * tensor: per output phase (Y & 1, X & 1) a 2x2 convolution of it, a 3x3 convolution of the skip tensor with pre-multiplied weights,
 * and a bias that depends only on which high-resolution taps lie inside the map -- 0.40-0.51x the matrix work, same result.
 * vp_compose_upconv: the composition alone.  wt [cin][cm][2][2], bt [cm] (ConvTranspose2d); ws [cm][cs], bs [cm] (1x1 skip link; cs = 0:
 *   none, ws / bs / wsk may be NULL); w3 [cout][cm][3][3], b3 [cout].  Results (fp64):
 *   wx [4 phases py*2+px][4 taps a*2+b][cout][cin]: out(2y+py, 2x+px) += wx . x(y+py-1+a, x+px-1+b) (zero outside the low-resolution map);
 *   wsk [9 taps ty*3+tx][cout][cs]: plain 3x3 / pad 1 convolution of the skip tensor;
 *   bias [9 classes rc*3+cc][cout]: rc / cc = 0 first, 1 inner, 2 last row / column of the OUTPUT map.
 * vp_op_upconv: the whole stage through the engine's kernel: in [cin][h][w], skip [cs][2h][2w] -> out [cout][2h][2w], fp32 CHW host
 *   buffers; act 0 none, 1 GELU; shape 6 / 7 or -1 (dispatch rule); nsplit K slices or 0 (dispatch rule); precision VP_FP16X3 (parity mode) or
 *   VP_FP16 (the fp16 engines' form: 64-channel chunks, one MFMA per product; input channels a multiple of 64 after padding). */
int vp_compose_upconv(int gpu_id, const float* wt, const float* bt, const float* ws, const float* bs, const float* w3, const float* b3, int cin, int cm,
                      int cout, int cs, double* wx, double* wsk, double* bias, char* err, size_t err_len);
int vp_op_upconv(int gpu_id, const float* in, int cin, int h, int w, const float* skip, int cs, const float* wt, const float* bt, const float* ws,
                 const float* bs, const float* w3, const float* b3, int cm, int cout, int act, int shape, int nsplit, int precision, float* out,
                 char* err, size_t err_len);

/* ---- developer options ------------------------------------------------------------------------------------
 * The library NEVER reads the environment.  The dispatch rules' developer knobs (A/B timing of a kernel against the one it replaced,
 * tile / split-K sweeps: DESIGN.md section 3, csrc/options.cpp for the key list) are set here, process-wide, and affect engines
 * created AFTERWARDS; value NULL removes a key; an unknown key is VP_ERR_ARG.  No option is needed in production and none changes a
 * result beyond fp32 summation order.  vp_version() lists every option in force; vp_plan_hash() = FNV-1a over (launch name, kernel
 * tag, launch geometry) of an engine's plan, so a host (bench.py does) can record exactly which kernels ran.
 * "VP_PLAN_TARGET" = "latency" | "throughput" (anything else: VP_ERR_ARG) selects the kernel plan target for EVERY engine created while it is set --
 * a developer knob for A/B runs since round 6; hosts choose per engine with the creation flag VP_PLAN_LATENCY (vp_precision). */
int vp_set_option(const char* key, const char* value);
const char* vp_get_option(const char* key);       /* NULL when unset */
void vp_clear_options(void);
unsigned long long vp_plan_hash(const vp_engine* e);
/* Device bytes of an engine's matrix / FC WEIGHT tensors by storage class: OCP e4m3 codes (VP_WEIGHTS_FP8 on the kernels that convert on
 * the fly), fp16 planes ((hi, lo) pairs in the parity mode), fp32 (FC rows, the fused MBConv projection / depthwise filters).  Bias and
 * scale vectors, tables and activations are not counted.  Any pointer may be NULL. */
int vp_weight_bytes(const vp_engine* e, unsigned long long* fp8_bytes, unsigned long long* fp16_bytes, unsigned long long* fp32_bytes);
/* host only (tests / external checks): the (hi, lo) fp16 planes the engine makes of a weight matrix [rows][per_row] and the per-row
 * 2^-s of its power-of-two prescale: fp32(hi) + fp32(lo) = w * 2^s with the row maximum in [2^13, 2^14), post_scale[r] = 2^-s.
 * VP_ERR_RANGE for a weight beyond the fp16 range, as vp_create*. */
int vp_split_weight_rows(const float* w, int rows, int per_row, uint16_t* hi, uint16_t* lo, float* post_scale);
/* host only (tests): what VP_WEIGHTS_FP8 storage keeps of a weight matrix -- one OCP e4m3 code per element and the per-row scale
 * (row maximum / 448); for rows that came out of the per-row quantiser (possibly re-scaled since, as BatchNorm folding does)
 * code x scale reproduces the value to fp32 rounding. */
int vp_fp8_encode_rows(const float* w, int rows, int per_row, uint8_t* codes, float* row_scale);

const char* vp_version(void);

#ifdef __cplusplus
}
#endif
#endif /* VP_HIP_H_ */
