// The opaque handle behind include/vp_hip.h (shared by vp_api.cpp and vp_comm.cpp).
#pragma once
#include <memory>
#include <string>

#include "../../include/vp_hip.h"
#include "engine.hpp"

struct vp_engine {
  std::unique_ptr<vp::Engine> impl;
  std::string err;
};
