// context.h — host-side state behind the opaque lab_ctx / lab_vec handles.
// Internal to the shared library.
#pragma once
#include <hip/hip_runtime.h>

#include <string>
#include <utility>
#include <vector>

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
  // measurement hook (lab_ctx_profile_begin/end): HIP events around the mat-vec kernel
  bool prof_on = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_events;
  size_t prof_used = 0;
  double prof_bytes = 0.0;
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

namespace lab {

// Host/device staging for boundary pointers.  Usage: sum need() over all
// pointers (+ temporaries), ensure_scratch, then in()/out() carve and copy, and
// finish() copies results back (synchronising only if something went to the host).
struct Staging {
  lab_ctx* c;
  Carver* cv;
  struct Pending { void* host; const void* dev; size_t bytes; };
  Pending pend[16];
  int npend = 0;
  Staging(lab_ctx* ctx, Carver* carver) : c(ctx), cv(carver) {}
  static size_t need(const void* p, size_t bytes) { return (p && !is_device_pointer(p)) ? Carver::padded(bytes) : 0; }
  template <typename T>
  int in(const T* p, size_t n, const T** dev) {
    if (!p || is_device_pointer(p)) { *dev = p; return LAB_OK; }
    T* slot = cv->take<T>(n);
    LAB_HIP(c, hipMemcpyAsync(slot, p, n * sizeof(T), hipMemcpyHostToDevice, c->stream));
    *dev = slot;
    return LAB_OK;
  }
  template <typename T>
  T* out(T* p, size_t n) {
    if (!p || is_device_pointer(p)) return p;
    T* slot = cv->take<T>(n);
    pend[npend++] = Pending{p, slot, n * sizeof(T)};
    return slot;
  }
  int finish() {
    if (!npend) return LAB_OK;
    for (int i = 0; i < npend; i++)
      LAB_HIP(c, hipMemcpyAsync(pend[i].host, pend[i].dev, pend[i].bytes, hipMemcpyDeviceToHost, c->stream));
    LAB_HIP(c, hipStreamSynchronize(c->stream));
    return LAB_OK;
  }
};

int check_range(lab_ctx* c, const lab_vec* v, size_t off, size_t n, const char* what);
// launch_matvec_partial wrapped in the profiling events when the hook is active
int matvec_partial_timed(lab_ctx* c, HatView partial, ConstHatView M, ConstHatView V, size_t nmat, size_t rows, size_t cols);
// hat -> coefficient conversion of `count` elements of `src` into device memory `dcoeffs`, using `tmp` (>= count elements of hat scratch)
int from_hat(lab_ctx* c, ConstHatView src, size_t count, uint32_t* tmp, uint64_t* dcoeffs);
int to_hat(lab_ctx* c, const uint64_t* dcoeffs, HatView dst, size_t count);

}  // namespace lab
