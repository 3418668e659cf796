#include "Hip.h"

#include <map>
#include <mutex>
#include <stdexcept>
#include <string>

vc2hip_ctx *hipContext(int device) {
  static std::mutex m;
  static std::map<int, vc2hip_ctx *> ctxs;
  std::lock_guard<std::mutex> lock(m);
  auto it = ctxs.find(device);
  if (it != ctxs.end()) return it->second;
  vc2hip_ctx *c = nullptr;
  const int rc = vc2hip_create(device, &c);
  if (rc != VC2HIP_OK)
    throw std::runtime_error(std::string("vc2hip: cannot create a context on HIP device ") + std::to_string(device) +
                             " (" + vc2hip_error_string(rc) + "); there is no CPU fallback");
  ctxs[device] = c;
  return c;
}

void hipCheck(vc2hip_ctx *ctx, int rc) {
  if (rc == VC2HIP_OK) return;
  const std::string msg = vc2hip_last_error(ctx);
  switch (rc) {
    case VC2HIP_EINVAL: throw std::invalid_argument(msg);
    case VC2HIP_EBOUNDED: throw std::length_error(msg);
    case VC2HIP_EHIP: throw std::runtime_error(msg);
    default: throw std::logic_error(msg);
  }
}
