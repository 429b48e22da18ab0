// Multi-camera host in C++ -- no Python, no torch -- over the C ABI only (include/vp_hip.h): what a ROS2 / Zenoh host of an 8-camera rig
// links (north_star: "host code stays C++ ... one camera per GPU ... RCCL/xGMI all-gather only for the fused BEV / ego-path head";
// BASELINE configs[3]).  One THREAD per GPU, camera r on GPU r:
//     vp_create_from_memory (SceneSeg)  +  vp_create_shared_from_memory (Scene3D on the SAME encoder: BASELINE's metric configuration)
//     rank 0: vp_comm_unique_id; every rank: vp_comm_create(rank, world = cameras)
//     per frame:  vp_upload_frame -> vp_enqueue_multi(base, {scene3d}) -> vp_gather(base, VP_GATHER_MASK)      (no host sync in the loop)
//     end:        vp_comm_fetch: every rank holds every camera's class map; checked against the rank's own vp_mask_u8 and rank 0's copy
// world = min(visible GPUs, requested cameras): 1 on a single-GPU box (the collective still runs through RCCL), 8 on the node.
// Prints ONE JSON line: frames/s over all cameras (steady clock around the loop between two all-rank rendezvous), rccl_world, ok.
//   usage: multicam_host SCENESEG.vpw SCENE3D.vpw [cameras=8] [frames=50] [dump.bin]
//   dump.bin (tests/test_adapters.py): u32 world, u32 h, u32 w, then per camera the LAST frame (h*w*3 BGR) and its gathered 320x640 class map.
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "vp_hip.h"

namespace {

constexpr int kFrameH = 720, kFrameW = 1280, kNetH = 320, kNetW = 640;

std::vector<char> read_file(const char* path) {
  std::ifstream f(path, std::ios::binary);
  if (!f) return {};
  return std::vector<char>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

// a smooth scene + noise, different per camera and frame (the class maps must differ between cameras for the check to mean anything)
void make_frame(std::vector<uint8_t>* buf, int cam, int frame) {
  buf->resize((size_t)kFrameH * kFrameW * 3);
  uint32_t s = 2654435761u * (uint32_t)(cam * 131 + frame * 7 + 1);
  for (int y = 0; y < kFrameH; ++y)
    for (int x = 0; x < kFrameW; ++x)
      for (int c = 0; c < 3; ++c) {
        s = s * 1664525u + 1013904223u;
        const int base = (x * (3 + cam) / 16 + y * (5 + c) / 8 + 40 * c + 29 * cam + 11 * frame) & 255;
        (*buf)[((size_t)y * kFrameW + x) * 3 + c] = (uint8_t)((base * 3 + (int)(s >> 26)) / 4 + (c == cam % 3 ? 20 : 0));
      }
}

struct Rendezvous {  // all ranks meet (start / stop of the timed loop)
  std::mutex m;
  std::condition_variable cv;
  int waiting = 0, phase = 0, n = 1;
  void wait() {
    std::unique_lock<std::mutex> lk(m);
    const int ph = phase;
    if (++waiting == n) {
      waiting = 0;
      ++phase;
      cv.notify_all();
    } else {
      cv.wait(lk, [&] { return phase != ph; });
    }
  }
};

struct Shared {
  std::vector<char> seg, s3d;
  uint8_t uid[VP_COMM_ID_BYTES];
  int world = 1, frames = 50;
  Rendezvous rv;
  std::atomic<int> failures{0};
  std::vector<std::vector<uint8_t>> last_frame, gathered;  // [rank]
  double loop_s = 0.0;
};

#define CK(expr, what)                                                              \
  do {                                                                              \
    const int rc_ = (expr);                                                         \
    if (rc_ != VP_OK) {                                                             \
      std::fprintf(stderr, "rank %d: %s failed (%d): %s\n", rank, what, rc_, err); \
      std::_Exit(1); /* the other ranks wait for this one in RCCL / at the rendezvous */ \
    }                                                                               \
  } while (0)

void camera_thread(Shared* sh, int rank) {
  char err[512] = "";
  vp_engine *seg = nullptr, *s3d = nullptr;
  vp_comm* comm = nullptr;
  CK(vp_create_from_memory(&seg, VP_SCENESEG, sh->seg.data(), sh->seg.size(), VP_FP16X3, rank, err, sizeof err), "vp_create (SceneSeg)");
  CK(vp_create_shared_from_memory(&s3d, seg, VP_SCENE3D, sh->s3d.data(), sh->s3d.size(), VP_FP16X3, rank, err, sizeof err), "vp_create_shared (Scene3D)");
  CK(vp_set_decode_mode(seg, VP_DECODE_CLASS_INDEX), "vp_set_decode_mode");
  CK(vp_comm_create(&comm, sh->uid, rank, sh->world, rank, (size_t)kNetH * kNetW, err, sizeof err), "vp_comm_create");
  std::vector<uint8_t> frame;
  vp_engine* heads[1] = {s3d};
  // warm-up: plan, graph capture, RCCL's first collective
  make_frame(&frame, rank, -1);
  CK(vp_upload_frame(seg, frame.data(), kFrameH, kFrameW, kFrameW * 3), "vp_upload_frame");
  for (int i = 0; i < 3; ++i) {
    if (vp_enqueue_multi(seg, heads, 1) != VP_OK || vp_gather(seg, comm, VP_GATHER_MASK) != VP_OK) {
      std::fprintf(stderr, "rank %d: warm-up failed: %s / %s\n", rank, vp_last_error(seg), vp_comm_last_error(comm));
      std::_Exit(1);
    }
  }
  CK(vp_sync(seg), "vp_sync");
  sh->rv.wait();
  const auto t0 = std::chrono::steady_clock::now();
  for (int f = 0; f < sh->frames; ++f) {
    make_frame(&frame, rank, f);  // the camera driver's work, on this host thread
    if (vp_upload_frame(seg, frame.data(), kFrameH, kFrameW, kFrameW * 3) != VP_OK || vp_enqueue_multi(seg, heads, 1) != VP_OK ||
        vp_gather(seg, comm, VP_GATHER_MASK) != VP_OK) {
      std::fprintf(stderr, "rank %d frame %d: %s / %s\n", rank, f, vp_last_error(seg), vp_comm_last_error(comm));
      std::_Exit(1);
    }
  }
  if (vp_sync(seg) != VP_OK) sh->failures++;
  sh->rv.wait();
  if (rank == 0) sh->loop_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  // every rank holds every camera's class map of the last frame
  const void* host = nullptr;
  size_t rec = 0;
  if (vp_comm_fetch(comm, seg, &host, &rec) != VP_OK || rec != (size_t)kNetH * kNetW) {
    std::fprintf(stderr, "rank %d: vp_comm_fetch: %s\n", rank, vp_comm_last_error(comm));
    sh->failures++;
  } else {
    const uint8_t* all = static_cast<const uint8_t*>(host);
    const uint8_t* mine = nullptr;
    int mh = 0, mw = 0;
    if (vp_mask_u8(seg, &mine, &mh, &mw) != VP_OK || mh != kNetH || mw != kNetW || std::memcmp(all + (size_t)rank * rec, mine, rec) != 0) {
      std::fprintf(stderr, "rank %d: its record in the gathered buffer is not its own class map\n", rank);
      sh->failures++;
    }
    sh->gathered[rank].assign(all, all + rec * sh->world);
    sh->last_frame[rank] = frame;
    // Scene3D ran on the same encoder pass: its depth map must exist and be finite
    const float* depth = nullptr;
    int64_t shape[4];
    if (vp_logits(s3d, &depth, shape) != VP_OK || shape[1] != 1 || shape[2] != kNetH || shape[3] != kNetW || !(depth[0] == depth[0])) {
      std::fprintf(stderr, "rank %d: Scene3D output missing\n", rank);
      sh->failures++;
    }
  }
  vp_comm_destroy(comm);
  vp_destroy(s3d);
  vp_destroy(seg);
}

}  // namespace

int main(int argc, char** argv) {
  if (argc < 3) {
    std::fprintf(stderr, "usage: %s SCENESEG.vpw SCENE3D.vpw [cameras=8] [frames=50] [dump.bin]\n", argv[0]);
    return 2;
  }
  Shared sh;
  sh.seg = read_file(argv[1]);
  sh.s3d = read_file(argv[2]);
  if (sh.seg.empty() || sh.s3d.empty()) {
    std::fprintf(stderr, "cannot read the weight blobs\n");
    return 2;
  }
  const int want = argc > 3 ? std::atoi(argv[3]) : 8;
  sh.frames = argc > 4 ? std::atoi(argv[4]) : 50;
  int gpus = vp_device_count();
  if (gpus < 1) {
    std::fprintf(stderr, "no GPU visible\n");
    return 3;
  }
  sh.world = std::max(1, std::min(gpus, want));
  sh.rv.n = sh.world;
  sh.last_frame.resize(sh.world);
  sh.gathered.resize(sh.world);
  char err[512] = "";
  if (vp_comm_unique_id(sh.uid, err, sizeof err) != VP_OK) {
    std::fprintf(stderr, "vp_comm_unique_id: %s\n", err);
    return 3;
  }
  std::vector<std::thread> th;
  for (int r = 0; r < sh.world; ++r) th.emplace_back(camera_thread, &sh, r);
  for (auto& t : th) t.join();
  int bad = sh.failures.load();
  for (int r = 1; r < sh.world && bad == 0; ++r)
    if (sh.gathered[r] != sh.gathered[0]) {
      std::fprintf(stderr, "rank %d gathered different records than rank 0\n", r);
      ++bad;
    }
  if (argc > 5 && bad == 0) {
    std::ofstream o(argv[5], std::ios::binary);
    const uint32_t hdr[3] = {(uint32_t)sh.world, (uint32_t)kFrameH, (uint32_t)kFrameW};
    o.write(reinterpret_cast<const char*>(hdr), sizeof hdr);
    for (int r = 0; r < sh.world; ++r) {
      o.write(reinterpret_cast<const char*>(sh.last_frame[r].data()), (std::streamsize)sh.last_frame[r].size());
      o.write(reinterpret_cast<const char*>(sh.gathered[0].data() + (size_t)r * kNetH * kNetW), (std::streamsize)kNetH * kNetW);
    }
  }
  std::printf("{\"host\": \"c++ thread-per-gpu\", \"rccl_world\": %d, \"gpus_visible\": %d, \"cameras\": %d, \"frames_per_camera\": %d, "
              "\"networks\": \"SceneSeg+Scene3D shared encoder, fp16x3\", \"gather\": \"class map 320x640 u8 per frame, RCCL all-gather\", "
              "\"frames_per_s\": %.2f, \"ok\": %s}\n",
              sh.world, gpus, sh.world, sh.frames, sh.loop_s > 0 ? sh.world * sh.frames / sh.loop_s : 0.0, bad == 0 ? "true" : "false");
  return bad == 0 ? 0 : 1;
}
