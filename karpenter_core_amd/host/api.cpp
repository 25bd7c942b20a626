// api.cpp -- C entry points of libkshost.so: the host-side mirror of the reference call sites
//   scheduler := provisioner.NewScheduler(ctx, pods, stateNodes, opts)   (provisioner.go:301, helpers.go:85)
//   nodes, existing, err := scheduler.Solve(ctx, pods)                   (provisioner.go:307, helpers.go:93)
// ksh_open == NewScheduler (+NewTopology) + flattening + upload; ksh_solve == Solve through the
// libksolve C ABI (HIP kernels).  There is no CPU scheduling path in this library.
#include <atomic>
#include <mutex>
#include <chrono>
#include <cstring>
#include <string>
#include <thread>
#include <unordered_map>

#include "encode.hpp"

namespace {
// A batch of what-ifs derived on the device (ks_whatifs_open): the arena, the resident snapshot it points into and what the candidate sets were
struct DeltaBatch {
  ks_whatif_batch* b = nullptr; std::shared_ptr<void> base_dev; std::shared_ptr<const ksh::SnapshotBase> sb; std::vector<uint32_t> cand_off, cand;
  ~DeltaBatch() { if (b) ks_whatifs_free(b); }
};
struct Handle {
  std::unique_ptr<ksh::Encoded> enc; ks_dev_problem* dev = nullptr; std::unique_ptr<ksh::Encoded::ResultBuf> rb;
  std::shared_ptr<DeltaBatch> delta; uint32_t delta_index = 0;      // a derived what-if: `dev` is a view the batch owns; result pod ids follow the snapshot's queue order
  std::shared_ptr<void> base_dev;      // what-ifs over a snapshot: the snapshot's own flattening, resident on the device (shared catalogue + derived tables)
  bool solved = false;                 // rb holds the result of a successful solve (the result buffers are raw memory until then)
  bool dev_result = false;             // the device holds the result of a successful solve (price filter / launch pick / records read it there)
  std::vector<uint32_t> csr_off; std::vector<int32_t> csr_pods;      // ksh_result_arrays: the pods of every node in commit order (built on demand from pod_seq)
  ~Handle() { if (dev && !delta) ks_problem_free(dev); }
};
// KSR1 text of the result in h->rb.  A derived what-if numbers its pods in the snapshot's queue order; callers number a what-if's pods in
// candidate order, then pod order (ksh_open_whatifs): translate before decoding.
static std::string decode_handle(Handle* h, double dt) {
  if (!h->delta) return h->enc->decode(h->rb->r, dt);
  const DeltaBatch& D = *h->delta; const ksh::DeltaInputs in = ksh::delta_inputs(*D.sb); const uint32_t w = h->delta_index;
  std::vector<std::pair<uint32_t, uint32_t>> byrank;      // (rank in the snapshot's queue, candidate-order index)
  for (uint32_t i = D.cand_off[w]; i < D.cand_off[w + 1]; ++i) for (uint32_t p : (*in.by_node)[D.cand[i]]) byrank.push_back({in.pod_rank[p], (uint32_t)byrank.size()});
  std::sort(byrank.begin(), byrank.end());
  const uint32_t P = (uint32_t)byrank.size(); const ks_result& r = h->rb->r;
  std::vector<int32_t> pn(P + 1), ps(P + 1), pq(P + 1), un(P + 1); std::vector<uint32_t> pr(P + 1);
  for (uint32_t k = 0; k < P; ++k) { const uint32_t c = byrank[k].second; pn[c] = r.pod_node[k]; ps[c] = r.pod_stage[k]; pq[c] = r.pod_seq[k]; pr[c] = r.pod_reason[k]; }
  for (uint32_t i = 0; i < r.n_unscheduled; ++i) un[i] = (int32_t)byrank[(uint32_t)r.unscheduled[i]].second;
  ks_result t = r; t.pod_node = pn.data(); t.pod_stage = ps.data(); t.pod_seq = pq.data(); t.pod_reason = pr.data(); t.unscheduled = un.data();
  return h->enc->decode(t, dt);
}
thread_local std::string g_err;
uint32_t default_threads() { return ksh::host_threads(); }
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
    auto t0 = std::chrono::steady_clock::now();
    ksp::Problem pr = ksp::Parser(ksp_text, len).parse();
    auto t1 = std::chrono::steady_clock::now();
    h->enc = ksh::encode(std::move(pr), flags);
    auto t2 = std::chrono::steady_clock::now();
    h->rb = h->enc->make_result();
    if (getenv("KSH_TIMING")) fprintf(stderr, "ksh_open: parse %.2f ms, encode %.2f ms, result buffers %.2f ms\n", std::chrono::duration<double, std::milli>(t1 - t0).count(),
                                      std::chrono::duration<double, std::milli>(t2 - t1).count(), std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t2).count());
    *out = h.release(); return KS_OK;
  } catch (const ksh::Unsupported& e) { return set_err(KS_ERR_UNSUPPORTED, e.what());
  } catch (const std::exception& e) { return set_err(KS_ERR_INVALID, e.what()); }
}
void ksh_close(void* h) { delete (Handle*)h; }
const ks_problem* ksh_problem(void* h) { return &((Handle*)h)->enc->prob; }

// ---- the pod list in memory ----
// ksh_parse turns KSP1 text into the C++ objects (the analogue of the []*v1.Pod, []*cloudprovider.InstanceType, []*state.Node a Go
// caller holds); ksh_solve_from_pods then does what the reference does from that point: NewScheduler's flattening incl. NewQueue's
// sort and every per-pod computation, upload, the HIP kernels, read-back -- the window bench.py times as "solve_from_pods".
// The caller's objects in memory.  A cluster snapshot additionally keeps its flattening (ksh::SnapshotBase) once a what-if batch asked for
// it: a consolidation pass probes many candidate sets against the same snapshot (multinodeconsolidation.go:86-114 binary search,
// singlenodeconsolidation.go:54 scan), and everything that does not depend on the candidate set is flattened once.
struct Parsed {
  std::shared_ptr<const ksp::Problem> pr;
  std::mutex mu; std::shared_ptr<const ksh::SnapshotBase> sb; std::vector<int32_t> sb_pod_node; uint32_t sb_flags = 0;
  ksh::EnvCache env;      // the flattening of everything but the pods, reused by the next batch with the same universe signature
  // ksh_env_apply (round 6): once events were applied the library holds the bindings itself -- bind[i] = the node pod i is bound to, -1 for a pod that was unbound
  // (it stays in place: nothing that points into the problem may move) -- and the names of what is alive
  bool bind_set = false, had_cluster_pods = false; std::vector<int32_t> bind; std::unordered_map<std::string, uint32_t> live_node, live_pod; uint64_t tombstones = 0; uint32_t applied = 0;
};
// the bindings a what-if call means: the caller's array, or -- after ksh_env_apply -- the library's own
static const int32_t* bindings_of(Parsed* P, const int32_t* pod_node) { return pod_node ? pod_node : (P->bind_set ? P->bind.data() : nullptr); }
int ksh_parse(const char* ksp_text, size_t len, void** out) {
  *out = nullptr;
  try { auto p = std::make_unique<Parsed>(); p->pr = std::make_shared<const ksp::Problem>(ksp::Parser(ksp_text, len).parse()); *out = p.release(); return KS_OK; }
  catch (const std::exception& e) { return set_err(KS_ERR_INVALID, e.what()); }
}
int ksh_env_ingest(const ksh_env_block* env, void** out, double* ms) {
  if (out) *out = nullptr;
  if (!out || !env || !env->str_off || !env->words || (!env->str_bytes && env->n_strings)) return set_err(KS_ERR_INVALID, "null argument");
  try {
    auto t0 = std::chrono::steady_clock::now();
    for (uint32_t i = 0; i < env->n_strings; ++i) if (env->str_off[i + 1] < env->str_off[i]) return set_err(KS_ERR_INVALID, "env block: string offsets not ascending");
    if (env->n_strings && env->str_off[env->n_strings] > env->str_bytes_len) return set_err(KS_ERR_INVALID, "env block: string offsets reach beyond str_bytes_len");
    ksh_pod_block strings{}; strings.n_strings = env->n_strings; strings.str_off = env->str_off; strings.str_bytes = env->str_bytes; strings.str_bytes_len = env->str_bytes_len;
    auto p = std::make_unique<Parsed>();
    p->pr = std::make_shared<const ksp::Problem>(ksp::EnvReader(strings, env->words, env->words + env->n_words).read_env());
    if (ms) *ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    *out = p.release(); return KS_OK;
  } catch (const std::exception& e) { return set_err(KS_ERR_INVALID, e.what()); }
}
void ksh_parsed_free(void* p) { delete (Parsed*)p; }
// ms[0..5]: flatten (host) | upload | static tables + feasibility grid | pack kernel (HIP events) | whole ks_solve_dev incl. read-back | total wall
}  // extern "C"
template <class ENC> static int solve_from(ENC&& make_encoded, int device, void** out_handle, double* ms) {
  if (out_handle) *out_handle = nullptr;
  try {
    using clk = std::chrono::steady_clock; auto now = [] { return clk::now(); }; auto since = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    auto t0 = now();
    auto h = std::make_unique<Handle>();
    h->enc = make_encoded();
    auto t0b = now();
    h->rb = h->enc->make_result();
    auto t1 = now();
    if (getenv("KSH_TIMING")) fprintf(stderr, "  solve_from: encode %.2f ms, make_result %.2f ms\n", since(t0, t0b), since(t0b, t1));
    int rc = ks_problem_upload(&h->enc->prob, device, &h->dev); if (rc != KS_OK) return set_err(rc, ks_last_error());
    auto t2 = now();
    float grid_ms = 0; rc = ks_feasibility_grid(h->dev, nullptr, &grid_ms); if (rc != KS_OK) return set_err(rc, ks_last_error());
    auto t3 = now();
    float kms = 0; rc = ks_solve_dev(h->dev, &h->rb->r, &kms); if (rc != KS_OK) return set_err(rc, ks_last_error());
    h->solved = true; h->dev_result = true;
    auto t4 = now();
    if (ms) { ms[0] = since(t0, t1); ms[1] = since(t1, t2); ms[2] = since(t2, t3); ms[3] = kms; ms[4] = since(t3, t4); ms[5] = since(t0, t4); }
    if (out_handle) *out_handle = h.release();
    return KS_OK;
  } catch (const ksh::Unsupported& e) { return set_err(KS_ERR_UNSUPPORTED, e.what());
  } catch (const std::exception& e) { return set_err(KS_ERR_INVALID, e.what()); }
}
extern "C" {
int ksh_solve_from_pods(void* parsed, int device, uint32_t flags, void** out_handle, double* ms) {
  return solve_from([&] { return ksh::encode(((Parsed*)parsed)->pr, flags, &((Parsed*)parsed)->env); }, device, out_handle, ms);
}

// ---- binary pod ingress (kshost.h): flat pod records -> the batch in compact form; then Solve for that batch against an environment ----
struct Batch { std::shared_ptr<const ksp::PodBatch> b; };
int ksh_pods_ingest(const ksh_pod_block* blocks, uint32_t n_blocks, void** out_batch, double* ms) {
  if (out_batch) *out_batch = nullptr;
  if (!out_batch || (n_blocks && !blocks)) return set_err(KS_ERR_INVALID, "null argument");
  try {
    auto t0 = std::chrono::steady_clock::now();
    auto b = std::make_unique<Batch>(); b->b = ksh::ingest_pod_blocks(blocks, n_blocks);
    if (ms) *ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    *out_batch = b.release(); return KS_OK;
  } catch (const std::exception& e) { return set_err(KS_ERR_INVALID, e.what()); }
}
void ksh_pods_free(void* batch) { delete (Batch*)batch; }
int ksh_pods_count(void* batch, uint32_t* n_pods, uint32_t* n_specs) {
  if (!batch) return set_err(KS_ERR_INVALID, "null argument");
  if (n_pods) *n_pods = (uint32_t)((Batch*)batch)->b->size();
  if (n_specs) *n_specs = (uint32_t)((Batch*)batch)->b->specs.size();
  return KS_OK;
}
int ksh_solve_from_batch(void* parsed_env, void* batch, int device, uint32_t flags, void** out_handle, double* ms) {
  if (!parsed_env || !batch) return set_err(KS_ERR_INVALID, "null argument");
  return solve_from([&] { return ksh::encode(((Parsed*)parsed_env)->pr, ((Batch*)batch)->b, flags, &((Parsed*)parsed_env)->env); }, device, out_handle, ms);
}
// flatten only (no GPU needed) the problem a ksh_parse holds: ksh_open without the text
int ksh_open_parsed(void* parsed, uint32_t flags, void** out) {
  if (out) *out = nullptr;
  if (!parsed || !out) return set_err(KS_ERR_INVALID, "null argument");
  try {
    auto h = std::make_unique<Handle>();
    h->enc = ksh::encode(((Parsed*)parsed)->pr, flags, &((Parsed*)parsed)->env);
    h->rb = h->enc->make_result();
    *out = h.release(); return KS_OK;
  } catch (const ksh::Unsupported& e) { return set_err(KS_ERR_UNSUPPORTED, e.what());
  } catch (const std::exception& e) { return set_err(KS_ERR_INVALID, e.what()); }
}
int ksh_open_batch(void* parsed_env, void* batch, uint32_t flags, void** out) {
  if (out) *out = nullptr;
  if (!parsed_env || !batch || !out) return set_err(KS_ERR_INVALID, "null argument");
  try {
    auto h = std::make_unique<Handle>();
    h->enc = ksh::encode(((Parsed*)parsed_env)->pr, ((Batch*)batch)->b, flags, &((Parsed*)parsed_env)->env);
    h->rb = h->enc->make_result();
    *out = h.release(); return KS_OK;
  } catch (const ksh::Unsupported& e) { return set_err(KS_ERR_UNSUPPORTED, e.what());
  } catch (const std::exception& e) { return set_err(KS_ERR_INVALID, e.what()); }
}
// KSR1 text of the result a handle holds (after ksh_solve / ksh_solve_from_pods)
int ksh_result_text(void* hv, char** out_text) {
  Handle* h = (Handle*)hv; if (out_text) *out_text = nullptr;
  if (!h || !out_text) return set_err(KS_ERR_INVALID, "null argument");
  if (!h->solved) return set_err(KS_ERR_INVALID, "the handle holds no result: solve it first (or the last solve failed)");
  try { std::string s = decode_handle(h, 0.0); *out_text = strdup(s.c_str()); return KS_OK; }
  catch (const std::exception& e) { return set_err(KS_ERR_INVALID, e.what()); }
}

// Fixed-size record of the result a handle holds -- what consolidation reads of a simulation (consolidation.go:190-260):
// out[0] = number of new nodes, out[1] = number of unscheduled pods, out[2..2+words) = InstanceTypeOptions of new node 0 as a bitmask
// (zero if there is none).  No text, no per-pod data: this is what the ranks all-gather.
int ksh_result_summary(void* hv, uint64_t* out, uint32_t words) {
  Handle* h = (Handle*)hv; if (!h || !out) return set_err(KS_ERR_INVALID, "null argument");
  if (!h->solved) return set_err(KS_ERR_INVALID, "the handle holds no result: solve it first (or the last solve failed)");
  const ks_result& r = h->rb->r; const uint32_t TW = (h->enc->prob.T + 63) / 64;
  if (words < TW) return set_err(KS_ERR_INVALID, "summary row too short");
  out[0] = r.n_new; out[1] = r.n_unscheduled;
  for (uint32_t w = 0; w < words; ++w) out[2 + w] = (r.n_new && w < TW) ? r.node_types[w] : 0;
  return KS_OK;
}

// Consolidation what-ifs over ONE cluster snapshot (deprovisioning/helpers.go:42-115 simulateScheduling): the snapshot is
// parsed once -- `base` lists every state node and, as its pod batch, every bound pod with its full spec; pod_node[i] is the
// node (index into base nodes) pod i runs on -- and what-if w is derived natively: its candidate nodes cand[cand_off[w] ..
// cand_off[w+1]) leave the state-node list (helpers.go:48-61), their pods, in candidate order, become the pending batch, the
// flattening (NewScheduler / NewTopology host half) runs on `nthreads` host threads.  out_handles[w] is a ksh_open handle.
static int open_whatifs_over(std::shared_ptr<const ksp::Problem> snapshot, uint32_t flags, uint32_t n, const uint32_t* cand_off, const uint32_t* cand,
                             const int32_t* pod_node, uint32_t nthreads, void** out_handles, Parsed* cache = nullptr) {
  for (uint32_t w = 0; w < n; ++w) out_handles[w] = nullptr;
  try {
    for (uint32_t i = 0; i < cand_off[n]; ++i) if (cand[i] >= snapshot->nodes.size()) return set_err(KS_ERR_INVALID, "candidate node out of range");
    // the snapshot is flattened ONCE (catalogue, universes, templates, every state node's row); a what-if adds only what its candidate set decides
    const bool timing = getenv("KSH_TIMING") != nullptr; auto t0 = std::chrono::steady_clock::now();
    std::shared_ptr<const ksh::SnapshotBase> sb;
    if (cache) {
      std::lock_guard<std::mutex> g(cache->mu);
      pod_node = bindings_of(cache, pod_node); if (!pod_node && !snapshot->pods.empty()) return set_err(KS_ERR_INVALID, "no bindings (pod_node)");
      const size_t np = snapshot->pods.size();
      if (cache->sb && cache->sb_flags == flags && cache->sb_pod_node.size() == np && std::equal(pod_node, pod_node + np, cache->sb_pod_node.begin())) sb = cache->sb;
      else { sb = ksh::make_snapshot_base(snapshot, pod_node, flags, cache->sb_flags == flags ? cache->sb.get() : nullptr); cache->sb = sb; cache->sb_flags = flags; cache->sb_pod_node.assign(pod_node, pod_node + np); }
    } else { if (!pod_node && !snapshot->pods.empty()) return set_err(KS_ERR_INVALID, "no bindings (pod_node)"); sb = ksh::make_snapshot_base(snapshot, pod_node, flags); }
    if (timing) { auto t1 = std::chrono::steady_clock::now(); fprintf(stderr, "  what-ifs: snapshot base %8.2f ms\n", std::chrono::duration<double, std::milli>(t1 - t0).count()); t0 = t1; }
    std::atomic<uint32_t> next{0}; std::atomic<int> rc{KS_OK}; std::vector<std::string> errs(n);
    auto work = [&]() {
      for (;;) {
        const uint32_t w = next.fetch_add(1); if (w >= n) return;
        try {
          auto h = std::make_unique<Handle>();
          h->enc = ksh::encode_whatif(*sb, cand + cand_off[w], cand_off[w + 1] - cand_off[w], flags);
          h->rb = h->enc->make_result();
          out_handles[w] = h.release();
        } catch (const ksh::Unsupported& e) { errs[w] = e.what(); rc = KS_ERR_UNSUPPORTED;
        } catch (const std::exception& e) { errs[w] = e.what(); rc = KS_ERR_INVALID; }
      }
    };
    const uint32_t nt = std::max(1u, std::min(nthreads ? nthreads : default_threads(), n));
    std::vector<std::thread> pool; for (uint32_t t = 1; t < nt; ++t) pool.emplace_back(work);
    work(); for (auto& t : pool) t.join();
    if (timing) { auto t1 = std::chrono::steady_clock::now(); fprintf(stderr, "  what-ifs: %u flattened on %u threads %8.2f ms\n", n, nt, std::chrono::duration<double, std::milli>(t1 - t0).count()); }
    if (rc != KS_OK) { std::string m; for (auto& e : errs) if (!e.empty()) { m = e; break; } for (uint32_t w = 0; w < n; ++w) { delete (Handle*)out_handles[w]; out_handles[w] = nullptr; } return set_err(rc, m); }
    return KS_OK;
  } catch (const ksh::Unsupported& e) { return set_err(KS_ERR_UNSUPPORTED, e.what());
  } catch (const std::exception& e) { return set_err(KS_ERR_INVALID, e.what()); }
}
int ksh_open_whatifs(const char* base_text, size_t len, uint32_t flags, uint32_t n, const uint32_t* cand_off, const uint32_t* cand,
                     const int32_t* pod_node, uint32_t nthreads, void** out_handles) {
  try {
    auto snapshot = std::make_shared<ksp::Problem>(ksp::Parser(base_text, len).parse());
    for (auto& nd : snapshot->nodes) nd.in_state = true;
    snapshot->simulation_mode = true;
    return open_whatifs_over(std::shared_ptr<const ksp::Problem>(snapshot), flags, n, cand_off, cand, pod_node, nthreads, out_handles);
  } catch (const std::exception& e) { for (uint32_t w = 0; w < n; ++w) out_handles[w] = nullptr; return set_err(KS_ERR_INVALID, e.what()); }
}
// The same over a snapshot the caller already holds as objects (ksh_parse): every node of it is a state node, every pod a bound pod.
int ksh_open_whatifs_parsed(void* parsed, uint32_t flags, uint32_t n, const uint32_t* cand_off, const uint32_t* cand, const int32_t* pod_node, uint32_t nthreads, void** out_handles) {
  return open_whatifs_over(((Parsed*)parsed)->pr, flags, n, cand_off, cand, pod_node, nthreads, out_handles, (Parsed*)parsed);
}

// FNV-1a over every array behind the handle's ks_problem: two construction routes produced the same flat problem iff equal.
uint64_t ksh_fingerprint(void* hv) {
  const ksh::Encoded& E = *((Handle*)hv)->enc; uint64_t h = 1469598103934665603ull;
  auto mix = [&](const void* p, size_t bytes) { const unsigned char* c = (const unsigned char*)p; for (size_t i = 0; i < bytes; ++i) { h ^= c[i]; h *= 1099511628211ull; } };
  auto vec = [&](const auto& v) { uint64_t n = v.size(); mix(&n, 8); if (n) mix(v.data(), n * sizeof(v[0])); };
  auto rs = [&](const ksh::ReqSetsStore& r) { vec(r.present); vec(r.complement); vec(r.mask); vec(r.gt); vec(r.lt); vec(r.it_state); };
  const ks_problem& p = E.prob; const uint32_t dims[16] = {p.P, p.C, p.T, p.M, p.E, p.K, p.R, p.G, p.GH, p.S, p.SC, p.max_new_nodes, p.flags, p.wellknown_mask, p.n_ct, p.n_topologies}; mix(dims, sizeof dims);
  const ksh::Encoded& C = E.catalogue(); const ksh::Encoded& L = E.lattice();
  vec(E.key_nvalues); vec(E.value_int); vec(C.it_present); vec(C.it_complement); vec(C.it_mask); vec(C.it_offer); vec(C.it_price); vec(C.it_alloc); vec(C.it_cap);
  vec(L.its_inter); vec(L.its_fail); vec(L.its_nidne); vec(L.its_types); rs(E.tmpl); rs(E.en); rs(E.cls); rs(E.flt);
  vec(E.tmpl_taints); vec(E.tmpl_types); vec(E.tmpl_daemon); vec(E.tmpl_remaining); vec(E.tmpl_daemon_present); vec(E.tmpl_limit_present);
  vec(E.en_taints); vec(E.en_avail); vec(E.en_requests); vec(E.en_requests_present); vec(E.en_port_off);
  vec(E.cls_hn_mode); vec(E.cls_hn_off); vec(E.hn_list); vec(E.cls_requests); vec(E.cls_requests_present); vec(E.cls_tolerated); vec(E.cls_port_off); vec(E.ports);
  vec(E.en_vol_limit); vec(E.en_vol_count); vec(E.en_vol_set); vec(E.cls_vol_off); vec(E.vol_list);
  vec(E.cls_own_off); vec(E.own_list); vec(E.cls_sel_off); vec(E.sel_list); vec(E.cls_isel_off); vec(E.isel_list); vec(E.cls_iown_off); vec(E.iown_list);
  vec(E.pod_stage_off); vec(E.stage_cls); vec(E.queue); vec(E.grp_type); vec(E.grp_active); vec(E.grp_key); vec(E.grp_max_skew); vec(E.grp_count); vec(E.grp_hslot);
  vec(E.grph_count); vec(E.grph_extra_pos); vec(E.grp_filter_off);
  return h;
}

}  // extern "C"
// The snapshot's own flattening resident on `device` with its tables built (once per snapshot and device; shared by every what-if over it).
static int resident_base(const ksh::Encoded* base, int device, std::shared_ptr<void>* out) {
  std::lock_guard<std::mutex> g(base->dev_mu);
  auto it = base->dev_resident.find(device);
  if (it != base->dev_resident.end()) { *out = it->second; return KS_OK; }
  ks_dev_problem* raw = nullptr;
  int rc = ks_problem_upload(&base->prob, device, &raw);
  if (rc == KS_OK) rc = ks_problem_prepare(raw);
  if (rc != KS_OK) { if (raw) ks_problem_free(raw); return set_err(rc, ks_last_error()); }
  *out = std::shared_ptr<void>(raw, [](void* p) { ks_problem_free((ks_dev_problem*)p); });
  base->dev_resident[device] = *out;
  return KS_OK;
}
extern "C" {
// ---- the snapshot kept current by events instead of re-ingested (state.Cluster's UpdateNode / DeleteNode / UpdatePod / DeletePod, cluster.go) ----
// The problem object is patched in place -- new nodes and pods are appended (the vectors were parsed with room: nothing moves), what leaves stays as a tombstone
// (a node out of state, a pod bound nowhere) -- and the snapshot's flattening, if there is one, is continued from the one before (ksh::make_snapshot_base `before`).
// Not to be called while another thread uses handles opened over this snapshot; handles opened BEFORE the call keep solving what they were opened for.
int ksh_env_apply(void* parsed, const int32_t* pod_node, const char* ksd_text, size_t len, uint32_t info[4]) {
  if (info) info[0] = info[1] = info[2] = info[3] = 0;
  if (!parsed || !ksd_text) return set_err(KS_ERR_INVALID, "null argument");
  Parsed* P = (Parsed*)parsed;
  try {
    std::vector<ksp::DeltaEvent> ev = ksp::Parser(ksd_text, len).parse_delta();
    std::lock_guard<std::mutex> g(P->mu);
    ksp::Problem& pr = const_cast<ksp::Problem&>(*P->pr);      // (the only writer; see above)
    if (!P->bind_set) {
      if (!pod_node && !pr.pods.empty()) return set_err(KS_ERR_INVALID, "the first ksh_env_apply needs the bindings (pod_node) of the snapshot's pods");
      P->bind.assign(pod_node, pod_node + pr.pods.size());
      for (size_t i = 0; i < pr.pods.size(); ++i) if (P->bind[i] >= (int32_t)pr.nodes.size()) return set_err(KS_ERR_INVALID, "pod_node out of range");
      for (size_t i = 0; i < pr.nodes.size(); ++i) if (pr.nodes[i].in_state && !P->live_node.emplace(pr.nodes[i].name, (uint32_t)i).second) { P->live_node.clear(); return set_err(KS_ERR_INVALID, "two state nodes share a name"); }
      for (size_t i = 0; i < pr.pods.size(); ++i) if (P->bind[i] >= 0 && !P->live_pod.emplace(pr.pods[i].uid, (uint32_t)i).second) { P->live_node.clear(); P->live_pod.clear(); return set_err(KS_ERR_INVALID, "two bound pods share a uid"); }
      P->had_cluster_pods = !pr.cluster_pods.empty(); P->bind_set = true;
    } else if (pod_node && !std::equal(pod_node, pod_node + pr.pods.size(), P->bind.begin())) return set_err(KS_ERR_INVALID, "the bindings passed differ from the ones the library holds since the last ksh_env_apply (pass NULL)");
    auto unbind = [&](uint32_t i) {
      ksp::Pod& p = pr.pods[i]; ksp::StateNode& n = pr.nodes[P->bind[i]];
      const ksp::ResList req = ksh::RequestsForPod(p);      // state.Node.cleanupForPod (node.go:175-182): Available() = Allocatable - the requests of the pods still there
      for (auto& kv : n.available) { auto r = req.find(kv.first); if (r != req.end()) kv.second += r->second; }
      for (auto& c : p.containers) for (auto& hp : c.ports) if (hp.port != 0) for (size_t k = 0; k < n.host_ports.size(); ++k) if (n.host_ports[k].ip == hp.ip && n.host_ports[k].port == hp.port && n.host_ports[k].proto == hp.proto) { n.host_ports.erase(n.host_ports.begin() + k); break; }
      for (auto& v : p.volumes) for (size_t k = 0; k < n.volumes.size(); ++k) if (n.volumes[k].driver == v.driver && n.volumes[k].pvc == v.pvc) { n.volumes.erase(n.volumes.begin() + k); break; }
      if (P->had_cluster_pods) for (size_t k = 0; k < pr.cluster_pods.size(); ++k) if (pr.cluster_pods[k].uid == p.uid) { pr.cluster_pods.erase(pr.cluster_pods.begin() + k); break; }
      P->live_pod.erase(p.uid); P->bind[i] = -1;
      p.uid = std::string("\1unbound-") + std::to_string(++P->tombstones);      // (uids stay unique: the same pod may be bound again)
    };
    uint32_t done = 0; std::string why;
    for (auto& e : ev) {
      if (e.kind == ksp::DeltaEvent::NodeAdd) {
        if (P->live_node.count(e.node.name)) { why = "NODE+: a state node named " + e.node.name + " exists"; break; }
        if (pr.nodes.size() == pr.nodes.capacity()) { why = "NODE+: the snapshot's spare room for nodes is used up (ingest it again)"; break; }
        e.node.in_state = true; P->live_node.emplace(e.node.name, (uint32_t)pr.nodes.size()); pr.nodes.push_back(std::move(e.node));
      } else if (e.kind == ksp::DeltaEvent::NodeRemove) {
        auto it = P->live_node.find(e.name); if (it == P->live_node.end()) { why = "NODE-: no state node named " + e.name; break; }
        const uint32_t nd = it->second;
        for (uint32_t i = 0; i < P->bind.size(); ++i) if (P->bind[i] == (int32_t)nd) unbind(i);      // (its pods go with it)
        pr.nodes[nd].in_state = false; P->live_node.erase(it);
        pr.nodes[nd].name = std::string("\1gone-") + std::to_string(++P->tombstones) + ":" + pr.nodes[nd].name;      // (the name may come back)
      } else if (e.kind == ksp::DeltaEvent::PodBind) {
        auto it = P->live_node.find(e.name); if (it == P->live_node.end()) { why = "BIND: no state node named " + e.name; break; }
        if (P->live_pod.count(e.pod.uid)) { why = "BIND: pod " + e.pod.uid + " is bound already (UNBIND it first)"; break; }
        if (pr.pods.size() == pr.pods.capacity()) { why = "BIND: the snapshot's spare room for pods is used up (ingest it again)"; break; }
        ksp::StateNode& n = pr.nodes[it->second];
        const ksp::ResList req = ksh::RequestsForPod(e.pod);      // state.Node.updateForPod (node.go:161-173)
        for (auto& kv : n.available) { auto r = req.find(kv.first); if (r != req.end()) kv.second -= r->second; }
        for (auto& c : e.pod.containers) for (auto& hp : c.ports) if (hp.port != 0) n.host_ports.push_back(hp);
        for (auto& v : e.pod.volumes) n.volumes.push_back(v);
        if (P->had_cluster_pods) { ksp::ClusterPod cp; cp.uid = e.pod.uid; cp.ns = e.pod.ns; cp.node_name = n.name; cp.labels = e.pod.labels; cp.anti_required = e.pod.anti_required; pr.cluster_pods.push_back(std::move(cp)); }
        P->live_pod.emplace(e.pod.uid, (uint32_t)pr.pods.size()); pr.pods.push_back(std::move(e.pod)); P->bind.push_back((int32_t)it->second);
      } else {
        auto it = P->live_pod.find(e.name); if (it == P->live_pod.end()) { why = "UNBIND: no bound pod with uid " + e.name; break; }
        unbind(it->second);
      }
      ++done;
    }
    P->applied += done; { std::lock_guard<std::mutex> ge(P->env.mu); P->env.base.reset(); }      // (a Solve over these objects flattens its environment again)
    // the snapshot's flattening follows, continued from the one before when there is one
    bool continued = false;
    if (P->sb) {
      std::shared_ptr<const ksh::SnapshotBase> before = P->sb;
      try { P->sb = ksh::make_snapshot_base(P->pr, P->bind.data(), P->sb_flags, before.get()); P->sb_pod_node = P->bind; continued = ksh::snapshot_continued(*P->sb); }
      catch (...) { P->sb.reset(); P->sb_pod_node.clear(); throw; }
      ksh::dispose_later(std::move(before));      // (the flattening before: torn down off this thread, once the handles that still use it are closed)
    }
    if (info) { info[0] = done; info[1] = (uint32_t)pr.nodes.size(); info[2] = (uint32_t)pr.pods.size(); info[3] = continued ? 1u : 0u; }
    if (done != ev.size()) return set_err(KS_ERR_INVALID, "event " + std::to_string(done) + ": " + why + " (the events before it were applied)");
    return KS_OK;
  } catch (const ksh::Unsupported& e) { return set_err(KS_ERR_UNSUPPORTED, e.what());
  } catch (const std::exception& e) { return set_err(KS_ERR_INVALID, e.what()); }
}
// the bindings the library holds after ksh_env_apply: out[0 .. n_pods) (cap entries at most); the sizes either way
int ksh_snapshot_bindings(void* parsed, int32_t* out, uint32_t cap, uint32_t* n_pods, uint32_t* n_nodes) {
  if (!parsed) return set_err(KS_ERR_INVALID, "null argument");
  Parsed* P = (Parsed*)parsed; std::lock_guard<std::mutex> g(P->mu);
  if (n_pods) *n_pods = (uint32_t)P->pr->pods.size();
  if (n_nodes) *n_nodes = (uint32_t)P->pr->nodes.size();
  if (out) { if (!P->bind_set) return set_err(KS_ERR_INVALID, "no ksh_env_apply yet: the caller holds the bindings"); std::copy_n(P->bind.begin(), std::min<size_t>(cap, P->bind.size()), out); }
  return KS_OK;
}
// FNV-1a over the snapshot's flattening (the flat problem + the tables the device derivation reads); `cold` != 0: of a flattening made from scratch for the comparison
// (tests: a continued flattening must equal it).  The snapshot must have been flattened (a what-if batch opened) or is flattened now.
int ksh_snapshot_fingerprint(void* parsed, const int32_t* pod_node, uint32_t flags, int cold, uint64_t* out) {
  if (!parsed || !out) return set_err(KS_ERR_INVALID, "null argument");
  try {
    Parsed* P = (Parsed*)parsed; std::lock_guard<std::mutex> g(P->mu);
    const int32_t* pn = bindings_of(P, pod_node); if (!pn && !P->pr->pods.empty()) return set_err(KS_ERR_INVALID, "no bindings");
    std::shared_ptr<const ksh::SnapshotBase> sb;
    if (cold) sb = ksh::make_snapshot_base(P->pr, pn, flags);
    else {
      const size_t np = P->pr->pods.size();
      if (!(P->sb && P->sb_flags == flags && P->sb_pod_node.size() == np && std::equal(pn, pn + np, P->sb_pod_node.begin()))) { P->sb = ksh::make_snapshot_base(P->pr, pn, flags); P->sb_flags = flags; P->sb_pod_node.assign(pn, pn + np); }
      sb = P->sb;
    }
    const ksh::DeltaInputs in = ksh::delta_inputs(*sb);
    uint64_t a; { Handle tmp; tmp.enc.reset(const_cast<ksh::Encoded*>(in.base.get())); a = ksh_fingerprint(&tmp); tmp.enc.release(); }
    *out = a ^ (ksh::snapshot_fingerprint(*sb) * 0x9E3779B97F4A7C15ull);
    return KS_OK;
  } catch (const ksh::Unsupported& e) { return set_err(KS_ERR_UNSUPPORTED, e.what());
  } catch (const std::exception& e) { return set_err(KS_ERR_INVALID, e.what()); }
}

// What-ifs DERIVED on the device from the resident snapshot (include/ksolve.h ks_whatifs_open): no per-what-if flattening, an upload of KBs.
// Same contract as ksh_open_whatifs_parsed, with the problems already resident on `device`; KS_ERR_UNSUPPORTED (nothing opened) when the
// snapshot's what-ifs do not differ by their candidate sets alone -- the caller then uses ksh_open_whatifs_parsed.
int ksh_open_whatifs_derived(void* parsed, uint32_t flags, uint32_t n, const uint32_t* cand_off, const uint32_t* cand, const int32_t* pod_node, int device, void** out_handles) {
  for (uint32_t w = 0; w < n; ++w) out_handles[w] = nullptr;
  try {
    Parsed* P = (Parsed*)parsed; std::shared_ptr<const ksp::Problem> snapshot = P->pr;
    for (uint32_t i = 0; i < cand_off[n]; ++i) if (cand[i] >= snapshot->nodes.size()) return set_err(KS_ERR_INVALID, "candidate node out of range");
    if (flags & KS_FLAG_STATS) return set_err(KS_ERR_UNSUPPORTED, "derived what-ifs carry no reference-algorithm statistics");
    std::shared_ptr<const ksh::SnapshotBase> sb;
    { std::lock_guard<std::mutex> g(P->mu);
      pod_node = bindings_of(P, pod_node); if (!pod_node && !snapshot->pods.empty()) return set_err(KS_ERR_INVALID, "no bindings (pod_node)");
      const size_t np = snapshot->pods.size();
      if (P->sb && P->sb_flags == flags && P->sb_pod_node.size() == np && std::equal(pod_node, pod_node + np, P->sb_pod_node.begin())) sb = P->sb;
      else { auto ts = std::chrono::steady_clock::now(); sb = ksh::make_snapshot_base(snapshot, pod_node, flags, P->sb_flags == flags ? P->sb.get() : nullptr); P->sb = sb; P->sb_flags = flags; P->sb_pod_node.assign(pod_node, pod_node + np);
             if (getenv("KSH_TIMING")) fprintf(stderr, "  derived what-ifs: %-28s %8.2f ms\n", "snapshot flattened (once)", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - ts).count()); } }
    const bool timing = getenv("KSH_TIMING") != nullptr; auto t0 = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) { if (!timing) return; auto t1 = std::chrono::steady_clock::now(); fprintf(stderr, "  derived what-ifs: %-28s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count()); t0 = t1; };
    const ksh::DeltaInputs in = ksh::delta_inputs(*sb);
    if (!in.eligible) return set_err(KS_ERR_UNSUPPORTED, "what-ifs of this snapshot cannot be derived on the device: " + in.why);
    auto D = std::make_shared<DeltaBatch>(); D->sb = sb; D->cand_off.assign(cand_off, cand_off + n + 1); D->cand.assign(cand, cand + cand_off[n]);
    int rc = resident_base(in.base.get(), device, &D->base_dev); if (rc != KS_OK) return rc;
    lap("snapshot resident");
    const ks_problem& bp = in.base->prob; const uint32_t M = bp.M, R = bp.R;
    std::vector<uint32_t> npods(n, 0); std::vector<int64_t> rem((size_t)n * M * R);
    for (uint32_t w = 0; w < n; ++w) {
      int64_t* rw = &rem[(size_t)w * M * R]; std::copy(bp.tmpl_remaining, bp.tmpl_remaining + (size_t)M * R, rw);
      for (uint32_t i = cand_off[w]; i < cand_off[w + 1]; ++i) {
        const uint32_t nd = cand[i]; npods[w] += (uint32_t)(*in.by_node)[nd].size();
        const int32_t m = in.node_tmpl[nd]; if (m >= 0) for (uint32_t r = 0; r < R; ++r) rw[(size_t)m * R + r] += in.node_cap[(size_t)nd * R + r];      // remainingResources: the node's capacity comes back (scheduler.go:244-246)
      }
    }
    lap("masks / remaining (host)");
    rc = ks_whatifs_open((const ks_dev_problem*)D->base_dev.get(), in.n_nodes, P->sb_pod_node.data(), in.node_row, n, cand_off, cand, npods.data(), rem.data(), in.topo, &D->b);
    if (rc != KS_OK) return set_err(rc, ks_last_error());
    lap("ks_whatifs_open (device)");
    ks_dev_problem* const* views = ks_whatifs_problems(D->b);
    for (uint32_t w = 0; w < n; ++w) {
      auto h = std::make_unique<Handle>();
      h->enc = std::make_unique<ksh::Encoded>(); h->enc->src = snapshot; h->enc->shared = in.base; h->enc->shared_lattice = true; h->enc->view = true;
      h->enc->prob = bp; h->enc->prob.P = npods[w]; h->enc->prob.max_new_nodes = npods[w] ? npods[w] : 1;
      h->dev = views[w]; h->delta = D; h->delta_index = w; h->base_dev = D->base_dev;
      out_handles[w] = h.release();
    }
    lap("handles");
    return KS_OK;
  } catch (const ksh::Unsupported& e) { return set_err(KS_ERR_UNSUPPORTED, e.what());
  } catch (const std::exception& e) { return set_err(KS_ERR_INVALID, e.what()); }
}

// CPU self-check of what ksh_open_whatifs_derived would derive on the device for ONE candidate set (kshost.h).
int ksh_check_whatif_derivation(void* parsed, uint32_t flags, const uint32_t* cand, uint32_t ncand, const int32_t* pod_node) {
  try {
    Parsed* P = (Parsed*)parsed; std::shared_ptr<const ksp::Problem> snapshot = P->pr;
    std::shared_ptr<const ksh::SnapshotBase> sb;
    { std::lock_guard<std::mutex> g(P->mu);
      pod_node = bindings_of(P, pod_node); if (!pod_node && !snapshot->pods.empty()) return set_err(KS_ERR_INVALID, "no bindings (pod_node)");
      const size_t np = snapshot->pods.size();
      if (P->sb && P->sb_flags == flags && P->sb_pod_node.size() == np && std::equal(pod_node, pod_node + np, P->sb_pod_node.begin())) sb = P->sb;
      else { sb = ksh::make_snapshot_base(snapshot, pod_node, flags, P->sb_flags == flags ? P->sb.get() : nullptr); P->sb = sb; P->sb_flags = flags; P->sb_pod_node.assign(pod_node, pod_node + np); } }
    const std::string why = ksh::check_derived_topology(*sb, cand, ncand, flags);
    return why.empty() ? KS_OK : set_err(why.rfind("not derivable", 0) == 0 ? KS_ERR_UNSUPPORTED : KS_ERR_INVALID, why);
  } catch (const ksh::Unsupported& e) { return set_err(KS_ERR_UNSUPPORTED, e.what());
  } catch (const std::exception& e) { return set_err(KS_ERR_INVALID, e.what()); }
}

// Upload the flat problem to HBM (idempotent).
int ksh_upload(void* hv, int device) {
  Handle* h = (Handle*)hv;
  if (h->dev) return ks_problem_device(h->dev) == device ? KS_OK : set_err(KS_ERR_INVALID, "problem already resident on another device");
  int rc;
  if (const ksh::Encoded* base = h->enc->shared.get()) {
    // A what-if flattened over a shared snapshot: the snapshot's flattening goes to the device once (catalogue, prices, lattice, the tables
    // derived from them); the what-if then uploads only what its candidate set decides.
    std::shared_ptr<void> bd;
    {
      std::lock_guard<std::mutex> g(base->dev_mu);
      auto it = base->dev_resident.find(device);
      if (it != base->dev_resident.end()) bd = it->second;
      else {
        ks_dev_problem* raw = nullptr;
        rc = ks_problem_upload(&base->prob, device, &raw);
        if (rc == KS_OK) rc = ks_problem_prepare(raw);
        if (rc != KS_OK) { if (raw) ks_problem_free(raw); return set_err(rc, ks_last_error()); }
        bd = std::shared_ptr<void>(raw, [](void* p) { ks_problem_free((ks_dev_problem*)p); });
        base->dev_resident[device] = bd;
      }
    }
    rc = ks_problem_upload_shared(&h->enc->prob, (const ks_dev_problem*)bd.get(), &h->dev);
    if (rc == KS_OK) h->base_dev = bd;
  } else rc = ks_problem_upload(&h->enc->prob, device, &h->dev);
  if (rc != KS_OK) return set_err(rc, ks_last_error());
  return KS_OK;
}

// Upload a batch (what-ifs of one snapshot) on host threads: the per-problem cost is packing its arrays into the pinned staging buffer.
int ksh_upload_batch(void** handles, uint32_t n, int device, uint32_t nthreads) {
  if (!n) return KS_OK;
  int rc0 = ksh_upload(handles[0], device); if (rc0 != KS_OK) return rc0;      // the first one also makes the snapshot resident (once)
  std::atomic<uint32_t> next{1}; std::atomic<int> rc{KS_OK}; std::mutex emu; std::string emsg;
  auto work = [&]() {
    for (;;) {
      const uint32_t i = next.fetch_add(1); if (i >= n) return;
      const int r = ksh_upload(handles[i], device);
      if (r != KS_OK) { rc = r; std::lock_guard<std::mutex> g(emu); if (emsg.empty()) emsg = g_err; }
    }
  };
  const uint32_t nt = std::max(1u, std::min(nthreads ? nthreads : default_threads(), n - 1));
  std::vector<std::thread> pool; for (uint32_t t = 1; t < nt; ++t) pool.emplace_back(work);
  work(); for (auto& t : pool) t.join();
  if (rc != KS_OK) return set_err(rc, emsg);
  return KS_OK;
}
// The fixed-size records of a batch in one call: out[i*(2+words) ..] as ksh_result_summary.
int ksh_result_summaries(void** handles, uint32_t n, uint64_t* out, uint32_t words) {
  for (uint32_t i = 0; i < n; ++i) { const int rc = ksh_result_summary(handles[i], out + (size_t)i * (2 + words), words); if (rc != KS_OK) return rc; }
  return KS_OK;
}

// Solve (device-resident inputs).  out_text may be NULL (skip decode).
int ksh_solve(void* hv, char** out_text, float* kernel_ms, double* wall_ms) {
  Handle* h = (Handle*)hv;
  int rc = h->dev ? KS_OK : ksh_upload(hv, ks_current_device()); if (rc != KS_OK) return rc;      // not uploaded yet: the calling thread's current HIP device
  auto t0 = std::chrono::steady_clock::now();
  h->solved = false; h->dev_result = false;
  if (!h->rb) h->rb = h->enc->make_result();
  rc = ks_solve_dev(h->dev, &h->rb->r, kernel_ms);
  double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  if (wall_ms) *wall_ms = dt * 1e3;
  if (rc != KS_OK) return set_err(rc, ks_last_error());
  h->solved = true; h->dev_result = true;
  try { if (out_text) { std::string s = decode_handle(h, dt); *out_text = strdup(s.c_str()); } }
  catch (const std::exception& e) { return set_err(KS_ERR_INVALID, e.what()); }
  return KS_OK;
}

// N independent problems in one launch (consolidation what-ifs, deprovisioning/helpers.go:42-115).
int ksh_solve_batch(void** hv, uint32_t n, char** out_texts, float* kernel_ms, double* wall_ms) {
  std::vector<ks_dev_problem*> ds(n); std::vector<ks_result*> rs(n);
  const int dev = ks_current_device();
  for (uint32_t i = 0; i < n; ++i) { Handle* h = (Handle*)hv[i]; if (!h->rb) h->rb = h->enc->make_result(); }      // (derived what-ifs allocate their result buffers on first use)
  for (uint32_t i = 0; i < n; ++i) { int rc = ((Handle*)hv[i])->dev ? KS_OK : ksh_upload(hv[i], dev); if (rc != KS_OK) return rc; ds[i] = ((Handle*)hv[i])->dev; rs[i] = &((Handle*)hv[i])->rb->r; }
  auto t0 = std::chrono::steady_clock::now();
  for (uint32_t i = 0; i < n; ++i) { ((Handle*)hv[i])->solved = false; ((Handle*)hv[i])->dev_result = false; }
  int rc = ks_solve_batch_dev(ds.data(), n, rs.data(), kernel_ms);
  double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  if (wall_ms) *wall_ms = dt * 1e3;
  if (rc != KS_OK) return set_err(rc, ks_last_error());
  for (uint32_t i = 0; i < n; ++i) { ((Handle*)hv[i])->solved = true; ((Handle*)hv[i])->dev_result = true; }
  try { if (out_texts) for (uint32_t i = 0; i < n; ++i) { std::string s = decode_handle((Handle*)hv[i], dt); out_texts[i] = strdup(s.c_str()); } }
  catch (const std::exception& e) { return set_err(KS_ERR_INVALID, e.what()); }
  return KS_OK;
}

// The same launch with the results left on the device (no read-back but the error words), and the fixed-size records of the batch built there
// into a caller-owned DEVICE buffer [n][3 + words] of uint64 -- [ids[i], n_new, n_unscheduled, new node 0's InstanceTypeOptions] -- which a
// fan-out hands to its one all-gather as is (multinodeconsolidation.go:74-114: many candidate sets, one decision record each).
int ksh_solve_batch_resident(void** hv, uint32_t n, float* kernel_ms, double* wall_ms) {
  std::vector<ks_dev_problem*> ds(n);
  const int dev = ks_current_device();
  for (uint32_t i = 0; i < n; ++i) { int rc = ((Handle*)hv[i])->dev ? KS_OK : ksh_upload(hv[i], dev); if (rc != KS_OK) return rc; ds[i] = ((Handle*)hv[i])->dev; }
  auto t0 = std::chrono::steady_clock::now();
  for (uint32_t i = 0; i < n; ++i) { ((Handle*)hv[i])->solved = false; ((Handle*)hv[i])->dev_result = false; }
  int rc = ks_solve_batch_dev(ds.data(), n, nullptr, kernel_ms);
  if (wall_ms) *wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  if (rc != KS_OK) return set_err(rc, ks_last_error());
  for (uint32_t i = 0; i < n; ++i) ((Handle*)hv[i])->dev_result = true;
  return KS_OK;
}
int ksh_result_records_dev(void** hv, uint32_t n, const uint64_t* ids, uint32_t words, void* d_out) {
  std::vector<ks_dev_problem*> ds(n);
  for (uint32_t i = 0; i < n; ++i) { Handle* h = (Handle*)hv[i]; if (!h->dev || !h->dev_result) return set_err(KS_ERR_INVALID, "records before solve"); ds[i] = h->dev; }
  int rc = ks_batch_records_dev(ds.data(), n, ids, words, d_out);
  if (rc != KS_OK) return set_err(rc, ks_last_error());
  return KS_OK;
}

// Consolidation price stage on the results of the last ksh_solve / ksh_solve_batch, which are still on the device:
// for handle i, of new node node[i]'s InstanceTypeOptions keep the types whose worst launch price is < max_price[i]
// (filterByPrice, deprovisioning/helpers.go:148-157).  out_masks: n * ceil(T_max/64) words with row stride `stride_words`.
int ksh_price_filter(void** hv, uint32_t n, const uint32_t* node, const double* max_price, const uint32_t* spot_only, uint64_t* out_masks, uint32_t stride_words, uint32_t* out_counts) {
  std::vector<ks_dev_problem*> ds(n); std::vector<uint64_t*> outs(n);
  for (uint32_t i = 0; i < n; ++i) {
    Handle* h = (Handle*)hv[i]; if (!h->dev || !h->dev_result) return set_err(KS_ERR_INVALID, "price filter before solve");
    if ((h->enc->prob.T + 63) / 64 > stride_words) return set_err(KS_ERR_INVALID, "mask row too short");
    ds[i] = h->dev; outs[i] = out_masks + (size_t)i * stride_words;
  }
  int rc = ks_price_filter_dev(ds.data(), n, node, max_price, spot_only, outs.data(), out_counts);
  if (rc != KS_OK) return set_err(rc, ks_last_error());
  return KS_OK;
}

// Launch-time pick of the in-memory provider (fake/cloudprovider.go:79-84) and instanceTypesAreSubset (helpers.go:118-122) on results that are
// still on the device.  launch pick: out_type = instance-type index (into the problem's catalogue) or -1, out_zone / out_ct = value ids in the zone /
// capacity-type universes (ksh_key_value resolves them), out_price.
int ksh_launch_pick(void** hv, uint32_t n, const uint32_t* node, int32_t* out_type, int32_t* out_zone, int32_t* out_ct, double* out_price) {
  std::vector<ks_dev_problem*> ds(n); std::vector<int32_t> pair(n);
  for (uint32_t i = 0; i < n; ++i) { Handle* h = (Handle*)hv[i]; if (!h->dev || !h->dev_result) return set_err(KS_ERR_INVALID, "launch pick before solve"); ds[i] = h->dev; }
  int rc = ks_launch_pick_dev(ds.data(), n, node, out_type, pair.data(), out_price);
  if (rc != KS_OK) return set_err(rc, ks_last_error());
  for (uint32_t i = 0; i < n; ++i) { const uint32_t nct = ((Handle*)hv[i])->enc->prob.n_ct; out_zone[i] = pair[i] < 0 ? -1 : pair[i] / (int32_t)nct; out_ct[i] = pair[i] < 0 ? -1 : pair[i] % (int32_t)nct; }
  return KS_OK;
}
// value `v` of the zone (which = 0) or capacity-type (which = 1) universe of the handle's problem; NULL when out of range (owned by the handle)
const char* ksh_key_value(void* hv, int which, int32_t v) {
  const ksh::Encoded& E = ((Handle*)hv)->enc->names(); const int32_t k = which == 0 ? E.prob.key_zone : E.prob.key_ct;
  if (k < 0 || v < 0 || (size_t)v >= E.key_values[k].size()) return nullptr;
  return E.key_values[k][v].c_str();
}
int ksh_types_subset(void** hv, uint32_t n, const uint32_t* node, const uint64_t* lhs, uint32_t stride_words, uint32_t* out) {
  std::vector<ks_dev_problem*> ds(n);
  for (uint32_t i = 0; i < n; ++i) { Handle* h = (Handle*)hv[i]; if (!h->dev || !h->dev_result) return set_err(KS_ERR_INVALID, "subset test before solve"); ds[i] = h->dev; }
  int rc = ks_types_subset_dev(ds.data(), n, node, lhs, stride_words, out);
  if (rc != KS_OK) return set_err(rc, ks_last_error());
  return KS_OK;
}

// Static feasibility grid [M][C][TW]; `out` may be NULL (timing only).
int ksh_grid(void* hv, uint64_t* out, float* kernel_ms) {
  Handle* h = (Handle*)hv; int rc = h->dev ? KS_OK : ksh_upload(hv, ks_current_device()); if (rc != KS_OK) return rc;
  rc = ks_feasibility_grid(h->dev, out, kernel_ms);
  if (rc != KS_OK) return set_err(rc, ks_last_error());
  return KS_OK;
}

// The grid's rows split over GPUs: a range of rows computed on the handle's device / rows computed elsewhere installed.
int ksh_grid_rows(void* hv, uint32_t row_lo, uint32_t row_hi, uint64_t* out_rows, void* out_rows_dev, float* kernel_ms) {
  Handle* h = (Handle*)hv; int rc = h->dev ? KS_OK : ksh_upload(hv, ks_current_device()); if (rc != KS_OK) return rc;
  rc = ks_feasibility_grid_rows(h->dev, row_lo, row_hi, out_rows, out_rows_dev, kernel_ms);
  if (rc != KS_OK) return set_err(rc, ks_last_error());
  return KS_OK;
}
int ksh_grid_install(void* hv, uint32_t row_lo, uint32_t row_hi, const uint64_t* rows, const void* rows_dev, int complete) {
  Handle* h = (Handle*)hv; int rc = h->dev ? KS_OK : ksh_upload(hv, ks_current_device()); if (rc != KS_OK) return rc;
  rc = ks_feasibility_grid_install(h->dev, row_lo, row_hi, rows, rows_dev, complete);
  if (rc != KS_OK) return set_err(rc, ks_last_error());
  return KS_OK;
}

// One-shot convenience: KSP1 text in, KSR1 text out.
int ksh_solve_ksp(const char* ksp_text, size_t len, uint32_t flags, char** out_text) {
  void* h = nullptr; int rc = ksh_open(ksp_text, len, flags, &h); if (rc != KS_OK) return rc;
  rc = ksh_solve(h, out_text, nullptr, nullptr); ksh_close(h); return rc;
}

extern "C" int ks_debug_classes(ks_dev_problem*, void*, void*);
int ksh_debug_classes(void* hv, void* briefs, void* plans) { Handle* h = (Handle*)hv; if (!h->dev) return KS_ERR_INVALID; return ks_debug_classes(h->dev, briefs, plans); }

// dims for tests / bench: [P,C,T,M,E,K,R,G,GH,S]
// The result as ARRAYS (round 5): what scheduler.Solve returns -- Node.Pods in commit order, InstanceTypeOptions, Requests, Requirements, the relaxation stage every pod
// ended at, the per-pod failure reasons -- without a text round trip.  Pointers into the handle's own result buffers: valid until the handle is solved again or closed.
int ksh_result_arrays_get(void* hv, ksh_result_arrays* out) {
  Handle* h = (Handle*)hv;
  if (!h || !out) return set_err(KS_ERR_INVALID, "null argument");
  if (!h->solved || !h->rb) return set_err(KS_ERR_INVALID, "result arrays before a solve");
  if (h->delta) return set_err(KS_ERR_UNSUPPORTED, "a what-if derived on the device numbers its pods in the snapshot's queue order: read it through ksh_result_text / ksh_result_summaries");
  const ks_problem& p = h->enc->prob; const ks_result& r = h->rb->r;
  const uint32_t nn = p.E + r.n_new;
  h->csr_off.assign((size_t)nn + 1, 0);
  uint32_t placed = 0;
  for (uint32_t i = 0; i < p.P; ++i) if (r.pod_node[i] >= 0) { h->csr_off[(size_t)r.pod_node[i] + 1]++; ++placed; }
  for (uint32_t n = 0; n < nn; ++n) h->csr_off[n + 1] += h->csr_off[n];
  // commit order within a node = ascending commit number; the numbers are unique over the Solve, so one pass in that order fills every node's list in place
  std::vector<int32_t> by_seq(placed, -1);
  for (uint32_t i = 0; i < p.P; ++i) if (r.pod_node[i] >= 0) { const int32_t sq = r.pod_seq[i]; if (sq < 0 || (uint32_t)sq >= placed || by_seq[sq] >= 0) return set_err(KS_ERR_INTERNAL, "commit numbers are not a permutation"); by_seq[sq] = (int32_t)i; }
  h->csr_pods.assign(placed, -1);
  { std::vector<uint32_t> fill(h->csr_off.begin(), h->csr_off.end() - 1); for (uint32_t sq = 0; sq < placed; ++sq) { const int32_t i = by_seq[sq]; h->csr_pods[fill[r.pod_node[i]]++] = i; } }
  memset(out, 0, sizeof *out);
  out->n_pods = p.P; out->n_existing = p.E; out->n_new = r.n_new; out->n_unscheduled = r.n_unscheduled; out->types_words = (p.T + 63) / 64; out->n_resources = p.R; out->n_keys = p.K;
  out->pod_node = r.pod_node; out->pod_stage = r.pod_stage; out->pod_reason = r.pod_reason; out->unscheduled = r.unscheduled;
  out->node_pods_off = h->csr_off.data(); out->node_pods = h->csr_pods.data();
  out->node_tmpl = r.node_tmpl; out->node_types = r.node_types; out->node_requests = r.node_requests; out->node_requests_present = r.node_requests_present;
  out->node_present = r.node_present; out->node_complement = r.node_complement; out->node_mask = r.node_mask; out->node_gt = r.node_gt; out->node_lt = r.node_lt; out->node_it_state = r.node_it_state;
  return KS_OK;
}
// the names behind the arrays: requirement key k, its interned value v (a value class names its first member; ksh_result_text lists every member), resource r;
// NULL when out of range (owned by the handle)
const char* ksh_name(void* hv, int what /* 0 key, 1 value of key a, 2 resource */, uint32_t a, uint32_t b) {
  const ksh::Encoded& E = ((Handle*)hv)->enc->names();
  if (what == 0) return a < E.key_names.size() ? E.key_names[a].c_str() : nullptr;
  if (what == 1) return (a < E.key_values.size() && b < E.key_values[a].size()) ? E.key_values[a][b].c_str() : nullptr;
  if (what == 2) return a < E.res_names.size() ? E.res_names[a].c_str() : nullptr;
  return nullptr;
}
// The what-if fan-out in one call (ks_solve_batch_sharded): shard s = handles[shard_off[s] .. shard_off[s + 1]) (every handle of a shard uploaded to the same device),
// ids[] names each handle's what-if; out_rows[n][3 + words] comes back ordered by id.
int ksh_solve_whatifs_sharded(void** hv, const uint32_t* shard_off, uint32_t nshards, const uint64_t* ids, uint32_t words, uint64_t* out_rows, float* kernel_ms_max) {
  if (!hv || !shard_off || !ids || !out_rows) return set_err(KS_ERR_INVALID, "null argument");
  const uint32_t n = shard_off[nshards];
  std::vector<ks_dev_problem*> ds(n);
  for (uint32_t i = 0; i < n; ++i) { Handle* h = (Handle*)hv[i]; if (!h->dev) return set_err(KS_ERR_INVALID, "a what-if that is not resident (ksh_upload / ksh_upload_batch / ksh_open_whatifs_derived first)"); ds[i] = h->dev; h->solved = false; h->dev_result = false; }
  std::vector<ks_dev_problem* const*> sp(nshards); std::vector<uint32_t> sn(nshards); std::vector<const uint64_t*> si(nshards);
  for (uint32_t s = 0; s < nshards; ++s) { sp[s] = ds.data() + shard_off[s]; sn[s] = shard_off[s + 1] - shard_off[s]; si[s] = ids + shard_off[s]; }
  int rc = ks_solve_batch_sharded(sp.data(), sn.data(), si.data(), nshards, words, out_rows, kernel_ms_max);
  if (rc != KS_OK) return set_err(rc, ks_last_error());
  for (uint32_t i = 0; i < n; ++i) ((Handle*)hv[i])->dev_result = true;
  return KS_OK;
}
int ksh_rr_status(void* hv, int* out2) { Handle* h = (Handle*)hv; if (!h || !h->dev) return KS_ERR_INVALID; return ks_problem_rr_status(h->dev, out2, out2 + 1); }
void ksh_dims(void* hv, uint32_t* d) { const ks_problem& p = ((Handle*)hv)->enc->prob; uint32_t v[10] = {p.P, p.C, p.T, p.M, p.E, p.K, p.R, p.G, p.GH, p.S}; memcpy(d, v, sizeof v); }

}  // extern "C"
