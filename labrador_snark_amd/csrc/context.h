// context.h — host-side state behind the opaque lab_ctx / lab_vec handles.
// Internal to the shared library.
#pragma once
#include <hip/hip_runtime.h>

#include <string>

#include "../../include/labrador_hip.h"
#include "kernels.h"

struct lab_ctx {
  uint64_t q = 0;
  int device = 0;
  int nlimbs = 0;
  bool direct = false;             // q itself is the single NTT prime
  uint32_t primes[lab::MAX_LIMBS] = {0, 0, 0, 0};
  uint64_t max_len = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  lab::LimbTable* d_tabs = nullptr;   // [nlimbs] on device
  lab::CrtTable* d_crt = nullptr;
  lab::LimbTable h_tabs[lab::MAX_LIMBS];
  lab::CrtTable h_crt;
  // growable device scratch, carved per call
  void* scratch = nullptr;
  size_t scratch_bytes = 0;
  mutable std::string err;
};

struct lab_vec {
  lab_ctx* ctx = nullptr;
  size_t count = 0;
  uint32_t* d = nullptr;  // [nlimbs][count][64]
  lab::HatView view(size_t off) const { return lab::HatView{d + off * lab::D, count * lab::D}; }
};

namespace lab {

int fail(const lab_ctx* ctx, int code, const char* fmt, ...);
// ensures ctx->scratch holds at least `bytes`; contents are not preserved
int ensure_scratch(lab_ctx* ctx, size_t bytes);
bool is_device_pointer(const void* p);

#define LAB_HIP(ctx, expr)                                                                         \
  do {                                                                                             \
    hipError_t _e = (expr);                                                                        \
    if (_e != hipSuccess) return lab::fail(ctx, LAB_ERR_HIP, "%s: %s", #expr, hipGetErrorString(_e)); \
  } while (0)

// carve helper: sequential 256-byte aligned sub-allocations out of ctx->scratch
struct Carver {
  char* base;
  size_t off = 0;
  explicit Carver(void* b) : base(static_cast<char*>(b)) {}
  template <typename T>
  T* take(size_t n) {
    T* r = reinterpret_cast<T*>(base + off);
    off += (n * sizeof(T) + 255) & ~size_t(255);
    return r;
  }
  static size_t padded(size_t bytes) { return (bytes + 255) & ~size_t(255); }
};

}  // namespace lab
