// Developer options of libvp_hip (vp_set_option, include/vp_hip.h).  Up to round 3 the dispatch rules read ~25 VP_* ENVIRONMENT variables: a
// ROS node's environment could silently change which kernels run.  The library no longer reads the environment at all; the same keys are
// set through the C ABI, process-wide, by whoever wants them (the tools under tools/, the A/B tests), affect engines created AFTERWARDS, and
// every non-default option shows in vp_version() and in an engine's plan hash (vp_plan_hash) -- bench.py prints both.
#include <atomic>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <string>

#include "../../include/vp_hip.h"
#include "kernels.hpp"

namespace vp {
namespace {
std::mutex g_mu;
std::atomic<int> g_count{0};
std::map<std::string, const char*> g_opts;
std::deque<std::string> g_arena;   // values are never moved or freed: pointers handed out stay valid for the life of the process
bool g_version_stale = true;        // vp_version()'s text is rebuilt only after the options changed (it used to append a copy to the arena per call)
std::deque<std::string> g_versions; // retired texts stay alive: a pointer vp_version() handed out earlier remains valid

// the keys the dispatch rules know (a typo fails loudly instead of silently doing nothing)
const char* const kKeys[] = {"VP_MBCONV_FUSE", "VP_MBCONV_BACK", "VP_PROJ_SPLIT", "VP_FUSE_DECODE", "VP_AUTOTUNE", "VP_HEAD_CONV", "VP_CONVT_RS",
                             "VP_CONV3X3", "VP_HEAD_TILE5", "VP_X3_TILE", "VP_X3_MIN_WGS", "VP_MAP3X3", "VP_NSPLIT_PCT", "VP_NSPLIT_FORCE", "VP_X3_C64",
                             "VP_GEMM_DMA", "VP_GEMM_DMA_NSPLIT", "VP_CONVT_TILE", "VP_CONVT_BK", "VP_FUSE_SKIP", "VP_CONVT_RS_GROUPS", "VP_F16_BIG",
                             "VP_CTX3", "VP_MAP_TAPSPLIT", "VP_ATTN_BLOCK", "VP_F16_MAP", "VP_MAP_NSPLIT_PCT", "VP_MAP2", "VP_MAP2_SLOTS", "VP_X3_T16", "VP_MAP2_CAP_MB", "VP_MAP2_MIN_REGIONS", "VP_CTX_FUSE", "VP_MAP_MAX_M", "VP_F16_MIN_WGS", "VP_CONVT_RS_GROUPS_K288", "VP_PLAN_TARGET", "VP_UPCONV", "VP_UPCONV_SHAPE", "VP_UPCONV_NSPLIT", "VP_PLAN_OVERRIDE", "VP_WLO_KEEP_BITS", "VP_UPCONV_F16"};
}  // namespace

const char* dev_option(const char* key) {
  if (g_count.load(std::memory_order_acquire) == 0) return nullptr;   // the production case: one atomic load
  std::lock_guard<std::mutex> lk(g_mu);
  auto it = g_opts.find(key);
  return it == g_opts.end() ? nullptr : it->second;
}
// ONE lookup per question (a second dev_option(key) could see a concurrent vp_set_option(key, NULL) and return nullptr: ADVICE round 4)
bool dev_option_is(const char* key, char first) {
  const char* v = dev_option(key);
  return v != nullptr && v[0] == first;
}

}  // namespace vp

extern "C" {

int vp_set_option(const char* key, const char* value) {
  if (!key) return VP_ERR_ARG;
  bool known = false;
  for (const char* k : vp::kKeys) known = known || std::strcmp(k, key) == 0;
  if (!known) return VP_ERR_ARG;
  // the one key with a closed value set: a typo ("Latency", "lat") must not silently mean "throughput" 
  if (value && std::strcmp(key, "VP_PLAN_TARGET") == 0 && std::strcmp(value, "latency") != 0 && std::strcmp(value, "throughput") != 0) return VP_ERR_ARG;
  std::lock_guard<std::mutex> lk(vp::g_mu);
  if (!value) {
    vp::g_opts.erase(key);
  } else {
    vp::g_arena.emplace_back(value);
    vp::g_opts[key] = vp::g_arena.back().c_str();
  }
  vp::g_version_stale = true;
  vp::g_count.store((int)vp::g_opts.size(), std::memory_order_release);
  return VP_OK;
}

const char* vp_get_option(const char* key) { return key ? vp::dev_option(key) : nullptr; }

void vp_clear_options(void) {
  std::lock_guard<std::mutex> lk(vp::g_mu);
  vp::g_opts.clear();
  vp::g_version_stale = true;
  vp::g_count.store(0, std::memory_order_release);
}

const char* vp_version(void) {
  std::lock_guard<std::mutex> lk(vp::g_mu);
  if (vp::g_version_stale) {
    std::string s = "libvp_hip 0.5 (gfx950; options:";
    if (vp::g_opts.empty()) s += " none";
    for (const auto& kv : vp::g_opts) s += std::string(" ") + kv.first + "=" + kv.second;
    s += ")";
    if (vp::g_versions.empty() || vp::g_versions.back() != s) vp::g_versions.push_back(s);   // grows with option CHANGES, not with calls
    vp::g_version_stale = false;
  }
  return vp::g_versions.back().c_str();
}

}  // extern "C"
