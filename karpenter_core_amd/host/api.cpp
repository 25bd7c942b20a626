// api.cpp -- C entry points of libkshost.so: the host-side mirror of the reference call sites
//   scheduler := provisioner.NewScheduler(ctx, pods, stateNodes, opts)   (provisioner.go:301, helpers.go:85)
//   nodes, existing, err := scheduler.Solve(ctx, pods)                   (provisioner.go:307, helpers.go:93)
// ksh_open == NewScheduler (+NewTopology) + flattening + upload; ksh_solve == Solve through the
// libksolve C ABI (HIP kernels).  There is no CPU scheduling path in this library.
#include <chrono>
#include <cstring>
#include <string>

#include "encode.hpp"

namespace {
struct Handle {
  std::unique_ptr<ksh::Encoded> enc; ks_dev_problem* dev = nullptr; std::unique_ptr<ksh::Encoded::ResultBuf> rb;
  ~Handle() { if (dev) ks_problem_free(dev); }
};
thread_local std::string g_err;
int set_err(int code, const std::string& m) { g_err = m; return code; }
}  // namespace

extern "C" {

const char* ksh_last_error(void) { return g_err.c_str(); }
void ksh_free(char* p) { free(p); }

// Parse KSP1, run the host half of NewScheduler/NewTopology, flatten.  No GPU needed.
int ksh_open(const char* ksp_text, size_t len, uint32_t flags, void** out) {
  *out = nullptr;
  try {
    auto h = std::make_unique<Handle>();
    h->enc = ksh::encode(ksp::Parser(ksp_text, len).parse(), flags);
    h->rb = h->enc->make_result();
    *out = h.release(); return KS_OK;
  } catch (const ksh::Unsupported& e) { return set_err(KS_ERR_UNSUPPORTED, e.what());
  } catch (const std::exception& e) { return set_err(KS_ERR_INVALID, e.what()); }
}
void ksh_close(void* h) { delete (Handle*)h; }
const ks_problem* ksh_problem(void* h) { return &((Handle*)h)->enc->prob; }

// Upload the flat problem to HBM (idempotent).
int ksh_upload(void* hv, int device) {
  Handle* h = (Handle*)hv; if (h->dev) return KS_OK;
  int rc = ks_problem_upload(&h->enc->prob, device, &h->dev);
  if (rc != KS_OK) return set_err(rc, ks_last_error());
  return KS_OK;
}

// Solve (device-resident inputs).  out_text may be NULL (skip decode).
int ksh_solve(void* hv, char** out_text, float* kernel_ms, double* wall_ms) {
  Handle* h = (Handle*)hv;
  int rc = ksh_upload(hv, 0); if (rc != KS_OK) return rc;
  auto t0 = std::chrono::steady_clock::now();
  rc = ks_solve_dev(h->dev, &h->rb->r, kernel_ms);
  double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  if (wall_ms) *wall_ms = dt * 1e3;
  if (rc != KS_OK) return set_err(rc, ks_last_error());
  if (out_text) { std::string s = h->enc->decode(h->rb->r, dt); *out_text = strdup(s.c_str()); }
  return KS_OK;
}

// N independent problems in one launch (consolidation what-ifs, deprovisioning/helpers.go:42-115).
int ksh_solve_batch(void** hv, uint32_t n, char** out_texts, float* kernel_ms, double* wall_ms) {
  std::vector<ks_dev_problem*> ds(n); std::vector<ks_result*> rs(n);
  for (uint32_t i = 0; i < n; ++i) { int rc = ksh_upload(hv[i], 0); if (rc != KS_OK) return rc; ds[i] = ((Handle*)hv[i])->dev; rs[i] = &((Handle*)hv[i])->rb->r; }
  auto t0 = std::chrono::steady_clock::now();
  int rc = ks_solve_batch_dev(ds.data(), n, rs.data(), kernel_ms);
  double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  if (wall_ms) *wall_ms = dt * 1e3;
  if (rc != KS_OK) return set_err(rc, ks_last_error());
  if (out_texts) for (uint32_t i = 0; i < n; ++i) { std::string s = ((Handle*)hv[i])->enc->decode(*rs[i], dt); out_texts[i] = strdup(s.c_str()); }
  return KS_OK;
}

// Consolidation price stage on the results of the last ksh_solve / ksh_solve_batch, which are still on the device:
// for handle i, of new node node[i]'s InstanceTypeOptions keep the types whose worst launch price is < max_price[i]
// (filterByPrice, deprovisioning/helpers.go:148-157).  out_masks: n * ceil(T_max/64) words with row stride `stride_words`.
int ksh_price_filter(void** hv, uint32_t n, const uint32_t* node, const double* max_price, const uint32_t* spot_only, uint64_t* out_masks, uint32_t stride_words, uint32_t* out_counts) {
  std::vector<ks_dev_problem*> ds(n); std::vector<uint64_t*> outs(n);
  for (uint32_t i = 0; i < n; ++i) {
    Handle* h = (Handle*)hv[i]; if (!h->dev) return set_err(KS_ERR_INVALID, "price filter before solve");
    if ((h->enc->prob.T + 63) / 64 > stride_words) return set_err(KS_ERR_INVALID, "mask row too short");
    ds[i] = h->dev; outs[i] = out_masks + (size_t)i * stride_words;
  }
  int rc = ks_price_filter_dev(ds.data(), n, node, max_price, spot_only, outs.data(), out_counts);
  if (rc != KS_OK) return set_err(rc, ks_last_error());
  return KS_OK;
}

// Static feasibility grid [M][C][TW]; `out` may be NULL (timing only).
int ksh_grid(void* hv, uint64_t* out, float* kernel_ms) {
  Handle* h = (Handle*)hv; int rc = ksh_upload(hv, 0); if (rc != KS_OK) return rc;
  rc = ks_feasibility_grid(h->dev, out, kernel_ms);
  if (rc != KS_OK) return set_err(rc, ks_last_error());
  return KS_OK;
}

// One-shot convenience: KSP1 text in, KSR1 text out.
int ksh_solve_ksp(const char* ksp_text, size_t len, uint32_t flags, char** out_text) {
  void* h = nullptr; int rc = ksh_open(ksp_text, len, flags, &h); if (rc != KS_OK) return rc;
  rc = ksh_solve(h, out_text, nullptr, nullptr); ksh_close(h); return rc;
}

// dims for tests / bench: [P,C,T,M,E,K,R,G,GH,S]
void ksh_dims(void* hv, uint32_t* d) { const ks_problem& p = ((Handle*)hv)->enc->prob; uint32_t v[10] = {p.P, p.C, p.T, p.M, p.E, p.K, p.R, p.G, p.GH, p.S}; memcpy(d, v, sizeof v); }

}  // extern "C"
