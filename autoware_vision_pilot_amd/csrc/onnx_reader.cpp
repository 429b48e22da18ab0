// ONNX file -> "VPW1" weight blob, host only.  The reference's C++ backends take `model_path: *.onnx`
// (ROS2/models/config/autoseg.yaml:3; files made by Models/exports/convert_pytorch_to_onnx.py:144-154: opset 18,
// export_params, do_constant_folding, external_data=False); vp_create accepts the same path through this reader.
//
// Self-contained protobuf WIRE-FORMAT parser -- no protobuf / onnx library.  Field numbers used (onnx.proto3):
//   ModelProto.graph = 7;  GraphProto.node = 1, .initializer = 5;
//   NodeProto.input = 1, .output = 2, .name = 3, .op_type = 4;
//   TensorProto.dims = 1, .data_type = 2, .float_data = 4, .name = 8, .raw_data = 9, .double_data = 10.
// Naming rules (identical to autoware_vision_pilot_amd/weights.py load_onnx_state_dict, which the CPU tests compare
// this against tensor by tensor):
//   * an initializer that kept its state_dict name ("a.b.weight") is copied verbatim;
//   * a Conv / ConvTranspose whose weight is an exporter-made anonymous tensor ("onnx::Conv_626": Conv+BatchNorm fused
//     by constant folding) is named from the node's module scope, "/backbone/p3/p3.0/conv/Conv" -> "backbone.p3.0.conv",
//     and emitted as <conv>.weight + <conv>.bias (engine.cpp fold_conv takes that form in place of conv + norm);
//   * a second invocation of one module ("..._1", identical tensors) is dropped;
//   * a bias-free Linear exported as MatMul with a transposed anonymous operand becomes <module>.weight.
#include <cstdint>
#include <cstring>
#include <fstream>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

#include "engine.hpp"

namespace vp {
namespace {

struct View {
  const uint8_t* p = nullptr;
  size_t n = 0;
};

uint64_t varint(View& v) {
  uint64_t r = 0;
  for (int s = 0; s < 64; s += 7) {
    if (!v.n) throw std::runtime_error("onnx: truncated varint");
    const uint8_t b = *v.p++;
    --v.n;
    r |= (uint64_t)(b & 0x7F) << s;
    if (!(b & 0x80)) return r;
  }
  throw std::runtime_error("onnx: varint too long");
}

// Calls f(field, wire_type, varint_value, payload) for every field of one message.
template <class F>
void for_each_field(View m, F&& f) {
  while (m.n) {
    const uint64_t key = varint(m);
    const int field = (int)(key >> 3), wt = (int)(key & 7);
    uint64_t val = 0;
    View ld;
    if (wt == 0) {
      val = varint(m);
    } else if (wt == 1 || wt == 5) {
      const size_t w = wt == 1 ? 8 : 4;
      if (m.n < w) throw std::runtime_error("onnx: truncated fixed field");
      ld = View{m.p, w};
      m.p += w;
      m.n -= w;
    } else if (wt == 2) {
      const uint64_t len = varint(m);
      if (len > m.n) throw std::runtime_error("onnx: truncated length-delimited field");
      ld = View{m.p, (size_t)len};
      m.p += len;
      m.n -= len;
    } else {
      throw std::runtime_error("onnx: unsupported protobuf wire type " + std::to_string(wt));
    }
    f(field, wt, val, ld);
  }
}

std::string str(View v) { return std::string(reinterpret_cast<const char*>(v.p), v.n); }

float half_bits_to_float(uint16_t h) {
  const uint32_t sign = (uint32_t)(h & 0x8000) << 16;
  uint32_t exp = (h >> 10) & 0x1F, man = h & 0x3FF, bits;
  if (exp == 0) {
    if (man == 0) {
      bits = sign;
    } else {  // subnormal: renormalise
      int e = -1;
      do {
        man <<= 1;
        ++e;
      } while (!(man & 0x400));
      bits = sign | (uint32_t)(127 - 15 - e) << 23 | (man & 0x3FF) << 13;
    }
  } else if (exp == 31) {
    bits = sign | 0x7F800000u | man << 13;
  } else {
    bits = sign | (exp + 112) << 23 | man << 13;
  }
  float f;
  std::memcpy(&f, &bits, 4);
  return f;
}

struct Init {
  HostTensor t;
  bool is_float = false;
};

// onnx.TensorProto -> (name, tensor).  Non-floating tensors are kept as placeholders (is_float = false).
std::pair<std::string, Init> tensor_proto(View m) {
  std::vector<int64_t> dims;
  int dtype = 1;
  std::string name;
  View raw;
  bool has_raw = false;
  std::vector<float> floats;
  std::vector<double> doubles;
  for_each_field(m, [&](int f, int wt, uint64_t v, View ld) {
    if (f == 1) {
      if (wt == 0) {
        dims.push_back((int64_t)v);
      } else if (wt == 2) {
        while (ld.n) dims.push_back((int64_t)varint(ld));
      }
    } else if (f == 2 && wt == 0) {
      dtype = (int)v;
    } else if (f == 8 && wt == 2) {
      name = str(ld);
    } else if (f == 9 && wt == 2) {
      raw = ld;
      has_raw = true;
    } else if (f == 4 && (wt == 2 || wt == 5)) {
      for (size_t i = 0; i + 4 <= ld.n; i += 4) {
        float x;
        std::memcpy(&x, ld.p + i, 4);
        floats.push_back(x);
      }
    } else if (f == 10 && (wt == 2 || wt == 1)) {
      for (size_t i = 0; i + 8 <= ld.n; i += 8) {
        double x;
        std::memcpy(&x, ld.p + i, 8);
        doubles.push_back(x);
      }
    }
  });
  Init out;
  size_t numel = 1;
  for (int64_t d : dims) {
    if (d < 0 || d > INT32_MAX) throw std::runtime_error("onnx: bad dimension in tensor '" + name + "'");
    out.t.shape.push_back((int)d);
    numel *= (size_t)d;
  }
  if (dtype != 1 && dtype != 10 && dtype != 11) return {name, out};  // FLOAT, FLOAT16, DOUBLE only
  out.is_float = true;
  std::vector<float>& d = out.t.data;
  if (has_raw) {
    const size_t w = dtype == 1 ? 4 : (dtype == 10 ? 2 : 8);
    if (raw.n != numel * w) throw std::runtime_error("onnx: raw_data size mismatch in tensor '" + name + "'");
    d.resize(numel);
    for (size_t i = 0; i < numel; ++i) {
      if (dtype == 1) {
        std::memcpy(&d[i], raw.p + 4 * i, 4);
      } else if (dtype == 10) {
        uint16_t h;
        std::memcpy(&h, raw.p + 2 * i, 2);
        d[i] = half_bits_to_float(h);
      } else {
        double x;
        std::memcpy(&x, raw.p + 8 * i, 8);
        d[i] = (float)x;
      }
    }
  } else if (!floats.empty()) {
    d = std::move(floats);
  } else if (!doubles.empty()) {
    d.assign(doubles.begin(), doubles.end());
  }
  if (d.size() != numel) throw std::runtime_error("onnx: element count mismatch in tensor '" + name + "'");
  return {name, out};
}

struct Node {
  std::string op, name;
  std::vector<std::string> in;
};

bool anonymous(const std::string& n) { return n.rfind("onnx::", 0) == 0 || n.find('.') == std::string::npos; }

// "/backbone/p5/p5.3/middle_block/conv2/conv2.0/conv/Conv" -> "backbone.p5.3.middle_block.conv2.0.conv": one path
// segment per module level; a container's child is spelled "<container>.<idx>".
std::string scope_prefix(const std::string& node_name) {
  std::vector<std::string> seg;
  size_t i = 0;
  while (i < node_name.size()) {
    const size_t j = node_name.find('/', i);
    const size_t e = j == std::string::npos ? node_name.size() : j;
    if (e > i) seg.push_back(node_name.substr(i, e - i));
    i = e + 1;
  }
  if (!seg.empty()) seg.pop_back();  // the op's own name
  std::vector<std::string> comps;
  for (const std::string& s : seg) {
    if (!comps.empty() && s.rfind(comps.back() + ".", 0) == 0)
      comps.back() = s;
    else
      comps.push_back(s);
  }
  std::string out;
  for (const std::string& c : comps) out += (out.empty() ? "" : ".") + c;
  return out;
}

bool same(const HostTensor& a, const HostTensor& b) {
  return a.shape == b.shape && a.data.size() == b.data.size() &&
         std::memcmp(a.data.data(), b.data.data(), a.data.size() * sizeof(float)) == 0;
}

}  // namespace

std::map<std::string, HostTensor> load_onnx_state_dict(const std::string& path) {
  std::ifstream f(path, std::ios::binary | std::ios::ate);
  if (!f) throw std::runtime_error("cannot open weight file: " + path);
  const std::streamsize n = f.tellg();
  f.seekg(0);
  std::vector<uint8_t> buf((size_t)n);
  if (n && !f.read(reinterpret_cast<char*>(buf.data()), n)) throw std::runtime_error("short read on weight file: " + path);

  std::map<std::string, Init> inits;
  std::vector<Node> nodes;
  for_each_field(View{buf.data(), buf.size()}, [&](int mf, int mwt, uint64_t, View graph) {
    if (mf != 7 || mwt != 2) return;
    for_each_field(graph, [&](int gf, int gwt, uint64_t, View v) {
      if (gwt != 2) return;
      if (gf == 5) {
        auto kv = tensor_proto(v);
        if (!kv.first.empty()) inits[kv.first] = std::move(kv.second);
      } else if (gf == 1) {
        Node nd;
        for_each_field(v, [&](int nf, int nwt, uint64_t, View s) {
          if (nwt != 2) return;
          if (nf == 1) nd.in.push_back(str(s));
          if (nf == 3) nd.name = str(s);
          if (nf == 4) nd.op = str(s);
        });
        nodes.push_back(std::move(nd));
      }
    });
  });

  std::map<std::string, HostTensor> out;
  for (const auto& kv : inits)
    if (!anonymous(kv.first) && kv.second.is_float) out[kv.first] = kv.second.t;

  auto float_init = [&](const std::string& name) -> const HostTensor* {
    auto it = inits.find(name);
    return (it != inits.end() && it->second.is_float) ? &it->second.t : nullptr;
  };
  std::string unnamed;
  for (const Node& nd : nodes) {
    std::vector<std::pair<std::string, HostTensor>> ts;
    if ((nd.op == "Conv" || nd.op == "ConvTranspose") && nd.in.size() >= 2 && anonymous(nd.in[1]) && float_init(nd.in[1])) {
      ts.emplace_back("weight", *float_init(nd.in[1]));
      if (nd.in.size() >= 3 && float_init(nd.in[2])) ts.emplace_back("bias", *float_init(nd.in[2]));
    } else if (nd.op == "MatMul" && nd.in.size() == 2 && anonymous(nd.in[1]) && float_init(nd.in[1]) &&
               float_init(nd.in[1])->shape.size() == 2) {
      const HostTensor& w = *float_init(nd.in[1]);
      HostTensor t;
      const int r = w.shape[0], c = w.shape[1];
      t.shape = {c, r};
      t.data.resize(w.data.size());
      for (int i = 0; i < r; ++i)
        for (int j = 0; j < c; ++j) t.data[(size_t)j * r + i] = w.data[(size_t)i * c + j];
      ts.emplace_back("weight", std::move(t));
    } else {
      continue;
    }
    const std::string prefix = scope_prefix(nd.name);
    if (prefix.empty()) {
      if (unnamed.size() < 200) unnamed += (unnamed.empty() ? "" : "; ") + nd.op + " node '" + nd.name + "' (" + nd.in[1] + ")";
      continue;
    }
    // 2nd, 3rd ... invocation of one module: same tensors under the un-suffixed name
    const size_t dot = prefix.rfind('.');
    const std::string head = dot == std::string::npos ? "" : prefix.substr(0, dot + 1);
    const std::string leaf = dot == std::string::npos ? prefix : prefix.substr(dot + 1);
    const size_t us = leaf.rfind('_');
    if (us != std::string::npos && us + 1 < leaf.size() && leaf.find_first_not_of("0123456789", us + 1) == std::string::npos) {
      const std::string first = head + leaf.substr(0, us);
      bool dup = out.count(first + ".weight") != 0;
      for (const auto& t : ts) {
        auto it = out.find(first + "." + t.first);
        dup = dup && it != out.end() && same(it->second, t.second);
      }
      if (dup) continue;
    }
    for (auto& t : ts) {
      const std::string key = prefix + "." + t.first;
      auto it = out.find(key);
      if (it != out.end() && !same(it->second, t.second))
        throw std::runtime_error("onnx: two different tensors map to '" + key + "' (node '" + nd.name + "')");
      out[key] = std::move(t.second);
    }
  }
  if (!unnamed.empty()) throw std::runtime_error("onnx: cannot name exporter-folded weights without a module scope: " + unnamed);
  if (out.empty()) throw std::runtime_error("onnx: no floating-point weights found in " + path);
  return out;
}

// Serialises to the "VPW1" container (engine.hpp WeightBlob::parse).
std::vector<char> onnx_to_blob(const std::string& path) {
  const std::map<std::string, HostTensor> sd = load_onnx_state_dict(path);
  std::vector<char> b;
  auto put = [&](const void* p, size_t n) { b.insert(b.end(), reinterpret_cast<const char*>(p), reinterpret_cast<const char*>(p) + n); };
  put("VPW1", 4);
  const uint32_t count = (uint32_t)sd.size();
  put(&count, 4);
  for (const auto& kv : sd) {
    if (kv.first.size() > 0xFFFF || kv.second.shape.size() > 255) throw std::runtime_error("onnx: tensor name / rank too large: " + kv.first);
    const uint16_t nl = (uint16_t)kv.first.size();
    put(&nl, 2);
    put(kv.first.data(), nl);
    const uint8_t nd = (uint8_t)kv.second.shape.size();
    put(&nd, 1);
    for (int d : kv.second.shape) {
      const uint32_t u = (uint32_t)d;
      put(&u, 4);
    }
    put(kv.second.data.data(), kv.second.data.size() * sizeof(float));
  }
  return b;
}

}  // namespace vp
