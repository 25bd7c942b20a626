// api.cpp -- C entry points of libkshost.so: the host-side mirror of the reference call sites
//   scheduler := provisioner.NewScheduler(ctx, pods, stateNodes, opts)   (provisioner.go:301, helpers.go:85)
//   nodes, existing, err := scheduler.Solve(ctx, pods)                   (provisioner.go:307, helpers.go:93)
// ksh_open == NewScheduler (+NewTopology) + flattening + upload; ksh_solve == Solve through the
// libksolve C ABI (HIP kernels).  There is no CPU scheduling path in this library.
#include <atomic>
#include <chrono>
#include <cstring>
#include <string>
#include <thread>

#include "encode.hpp"

namespace {
struct Handle {
  std::unique_ptr<ksh::Encoded> enc; ks_dev_problem* dev = nullptr; std::unique_ptr<ksh::Encoded::ResultBuf> rb;
  ~Handle() { if (dev) ks_problem_free(dev); }
};
thread_local std::string g_err;
// Host threads for the what-if flattening: the cores this process may actually use -- a container usually sees every core of
// the machine but runs under a cgroup CPU quota (cpu.max = "quota period"); oversubscribing it is slower than one thread.
uint32_t default_threads() {
  uint32_t hw = std::max(1u, std::thread::hardware_concurrency());
  if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
    long long quota = -1, period = 0; char buf[64] = {0};
    if (fscanf(f, "%63s %lld", buf, &period) == 2 && strcmp(buf, "max") != 0) { quota = atoll(buf); if (quota > 0 && period > 0) hw = std::min<uint32_t>(hw, (uint32_t)std::max<long long>(1, quota / period)); }
    fclose(f);
  }
  return std::min(hw, 16u);
}
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

// Consolidation what-ifs over ONE cluster snapshot (deprovisioning/helpers.go:42-115 simulateScheduling): the snapshot is
// parsed once -- `base` lists every state node and, as its pod batch, every bound pod with its full spec; pod_node[i] is the
// node (index into base nodes) pod i runs on -- and what-if w is derived natively: its candidate nodes cand[cand_off[w] ..
// cand_off[w+1]) leave the state-node list (helpers.go:48-61), their pods, in candidate order, become the pending batch, the
// flattening (NewScheduler / NewTopology host half) runs on `nthreads` host threads.  out_handles[w] is a ksh_open handle.
int ksh_open_whatifs(const char* base_text, size_t len, uint32_t flags, uint32_t n, const uint32_t* cand_off, const uint32_t* cand,
                     const int32_t* pod_node, uint32_t nthreads, void** out_handles) {
  for (uint32_t w = 0; w < n; ++w) out_handles[w] = nullptr;
  try {
    const ksp::Problem base = ksp::Parser(base_text, len).parse();
    std::vector<std::vector<uint32_t>> by_node(base.nodes.size());
    for (size_t i = 0; i < base.pods.size(); ++i) { if (pod_node[i] < 0 || (size_t)pod_node[i] >= base.nodes.size()) return set_err(KS_ERR_INVALID, "pod_node out of range"); by_node[pod_node[i]].push_back((uint32_t)i); }
    for (uint32_t i = 0; i < cand_off[n]; ++i) if (cand[i] >= base.nodes.size()) return set_err(KS_ERR_INVALID, "candidate node out of range");
    std::atomic<uint32_t> next{0}; std::atomic<int> rc{KS_OK}; std::vector<std::string> errs(n);
    auto work = [&]() {
      for (;;) {
        const uint32_t w = next.fetch_add(1); if (w >= n) return;
        try {
          ksp::Problem pr; pr.extra_well_known = base.extra_well_known; pr.instance_types = base.instance_types; pr.provisioners = base.provisioners;
          pr.nodes = base.nodes; pr.cluster_pods = base.cluster_pods; pr.daemons = base.daemons; pr.simulation_mode = true;
          for (uint32_t i = cand_off[w]; i < cand_off[w + 1]; ++i) { pr.nodes[cand[i]].in_state = false; for (uint32_t p : by_node[cand[i]]) pr.pods.push_back(base.pods[p]); }
          auto h = std::make_unique<Handle>();
          h->enc = ksh::encode(std::move(pr), flags);
          h->rb = h->enc->make_result();
          out_handles[w] = h.release();
        } catch (const ksh::Unsupported& e) { errs[w] = e.what(); rc = KS_ERR_UNSUPPORTED;
        } catch (const std::exception& e) { errs[w] = e.what(); rc = KS_ERR_INVALID; }
      }
    };
    const uint32_t nt = std::max(1u, std::min(nthreads ? nthreads : default_threads(), n));
    std::vector<std::thread> pool; for (uint32_t t = 1; t < nt; ++t) pool.emplace_back(work);
    work(); for (auto& t : pool) t.join();
    if (rc != KS_OK) { std::string m; for (auto& e : errs) if (!e.empty()) { m = e; break; } for (uint32_t w = 0; w < n; ++w) { delete (Handle*)out_handles[w]; out_handles[w] = nullptr; } return set_err(rc, m); }
    return KS_OK;
  } catch (const std::exception& e) { return set_err(KS_ERR_INVALID, e.what()); }
}

// FNV-1a over every array behind the handle's ks_problem: two construction routes produced the same flat problem iff equal.
uint64_t ksh_fingerprint(void* hv) {
  const ksh::Encoded& E = *((Handle*)hv)->enc; uint64_t h = 1469598103934665603ull;
  auto mix = [&](const void* p, size_t bytes) { const unsigned char* c = (const unsigned char*)p; for (size_t i = 0; i < bytes; ++i) { h ^= c[i]; h *= 1099511628211ull; } };
  auto vec = [&](const auto& v) { uint64_t n = v.size(); mix(&n, 8); if (n) mix(v.data(), n * sizeof(v[0])); };
  auto rs = [&](const ksh::ReqSetsStore& r) { vec(r.present); vec(r.complement); vec(r.mask); vec(r.gt); vec(r.lt); vec(r.it_state); };
  const ks_problem& p = E.prob; const uint32_t dims[16] = {p.P, p.C, p.T, p.M, p.E, p.K, p.R, p.G, p.GH, p.S, p.SC, p.max_new_nodes, p.flags, p.wellknown_mask, p.n_ct, p.n_topologies}; mix(dims, sizeof dims);
  vec(E.key_nvalues); vec(E.value_int); vec(E.it_present); vec(E.it_complement); vec(E.it_mask); vec(E.it_offer); vec(E.it_price); vec(E.it_alloc); vec(E.it_cap);
  vec(E.its_inter); vec(E.its_fail); vec(E.its_nidne); vec(E.its_types); rs(E.tmpl); rs(E.en); rs(E.cls); rs(E.flt);
  vec(E.tmpl_taints); vec(E.tmpl_types); vec(E.tmpl_daemon); vec(E.tmpl_remaining); vec(E.tmpl_daemon_present); vec(E.tmpl_limit_present);
  vec(E.en_taints); vec(E.en_avail); vec(E.en_requests); vec(E.en_requests_present); vec(E.en_port_off);
  vec(E.cls_hn_mode); vec(E.cls_hn_off); vec(E.hn_list); vec(E.cls_requests); vec(E.cls_requests_present); vec(E.cls_tolerated); vec(E.cls_port_off); vec(E.ports);
  vec(E.cls_own_off); vec(E.own_list); vec(E.cls_sel_off); vec(E.sel_list); vec(E.cls_isel_off); vec(E.isel_list); vec(E.cls_iown_off); vec(E.iown_list);
  vec(E.pod_stage_off); vec(E.stage_cls); vec(E.queue); vec(E.grp_type); vec(E.grp_active); vec(E.grp_key); vec(E.grp_max_skew); vec(E.grp_count); vec(E.grp_hslot);
  vec(E.grph_count); vec(E.grph_extra_pos); vec(E.grp_filter_off);
  return h;
}

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
