// encode.cpp -- see encode.hpp.  Reference citations are relative to aws/karpenter-core pkg/.
#include "encode.hpp"

#include <atomic>
#include <future>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <unistd.h>
#include <exception>
#include <cstdio>
#include <cstring>
#include <sstream>
#include <string_view>
#include <thread>
#include <unordered_map>
#include <unordered_set>

namespace ksh {

namespace {

using ksp::Expr; using ksp::Pod; using ksp::StrMap;

const char* kBuiltinWellKnown[] = {ksp::kProvisionerName, ksp::kZone, ksp::kRegion, ksp::kInstanceType, ksp::kArch, ksp::kOS, ksp::kCapacityType};

struct TermRef { int type; std::string key; std::set<std::string> namespaces; ksp::Selector selector; int32_t max_skew; };   // one topology constraint of a pod spec

std::string selector_identity(const ksp::Selector& s) {
  if (s.nil) return "nil";
  std::string r = "sel{";
  for (auto& kv : s.match_labels) r += kv.first + "=" + kv.second + ",";
  std::vector<std::string> es;
  for (auto& e : s.match_exprs) { std::vector<std::string> vs = e.values; std::sort(vs.begin(), vs.end()); std::string x = e.key + ":" + std::to_string((int)e.op) + "["; for (auto& v : vs) x += v + ","; es.push_back(x + "]"); }
  std::sort(es.begin(), es.end()); for (auto& e : es) r += e + ";";
  return r + "}";
}

struct Filter { bool always = true; std::vector<Requirements> terms; };   // TopologyNodeFilter, topologynodefilter.go:28
Filter MakeTopologyNodeFilter(const Pod& p) {                             // :30-50
  Filter f; f.always = false;
  Requirements sel = Requirements::FromLabels(p.node_selector);
  if (p.required_affinity.empty()) { f.terms.push_back(sel); return f; }
  for (auto& term : p.required_affinity) { Requirements r; r.Add(sel); r.Add(Requirements::FromExprs(term)); f.terms.push_back(r); }
  return f;
}
bool FilterMatches(const Filter& f, const Requirements& reqs, const std::set<std::string>& wk) {   // MatchesRequirements :57-70
  if (f.always || f.terms.empty()) return true;
  for (auto& t : f.terms) if (reqs.Compatible(t, wk)) return true;
  return false;
}
// hashstructure v2.0.2 hashes exported fields only: of a filter term (map[string]*Requirement) only the keys.
std::string filter_identity(const Filter& f) {
  if (f.always) return "nofilter";
  std::vector<std::string> terms;
  for (auto& t : f.terms) { std::string x = "("; for (auto& kv : t.m) x += kv.first + ","; terms.push_back(x + ")"); }
  std::sort(terms.begin(), terms.end()); std::string r; for (auto& x : terms) r += x; return r;
}
std::string filter_content(const Filter& f) {
  if (f.always) return "nofilter";
  std::string r;
  for (auto& t : f.terms) { r += "("; for (auto& kv : t.m) r += kv.first + ":" + kv.second.identity() + ";"; r += ")"; }
  return r;
}

struct Group {
  int type; std::string key; std::set<std::string> namespaces; ksp::Selector selector; int32_t max_skew; Filter filter; bool inverse; bool active;
  std::string filter_sig;
  std::map<std::string, int32_t> counts;   // domain -> count (registered domains only)
};

// Preferences.Relax (preferences.go:36-145) applied to a spec copy; returns false when nothing is left to relax.
bool Relax(Pod& pod, bool toleratePreferNoSchedule) {
  if (pod.required_affinity.size() > 1) { pod.required_affinity.erase(pod.required_affinity.begin()); return true; }
  auto byw = [](const ksp::WeightedTerm& a, const ksp::WeightedTerm& b) { return a.weight > b.weight; };
  if (!pod.affinity_preferred.empty()) { std::stable_sort(pod.affinity_preferred.begin(), pod.affinity_preferred.end(), byw); pod.affinity_preferred.erase(pod.affinity_preferred.begin()); return true; }
  if (!pod.anti_preferred.empty()) { std::stable_sort(pod.anti_preferred.begin(), pod.anti_preferred.end(), byw); pod.anti_preferred.erase(pod.anti_preferred.begin()); return true; }
  if (!pod.preferred_affinity.empty()) {
    std::stable_sort(pod.preferred_affinity.begin(), pod.preferred_affinity.end(), [](const ksp::PreferredTerm& a, const ksp::PreferredTerm& b) { return a.weight > b.weight; });
    pod.preferred_affinity.erase(pod.preferred_affinity.begin()); return true;
  }
  for (size_t i = 0; i < pod.spread.size(); ++i) if (pod.spread[i].schedule_anyway) { pod.spread[i] = pod.spread.back(); pod.spread.pop_back(); return true; }
  if (toleratePreferNoSchedule) {
    for (auto& t : pod.tolerations) if (t.key.empty() && t.effect == "PreferNoSchedule" && t.op == "Exists" && t.value.empty()) return false;
    pod.tolerations.push_back({"", "Exists", "", "PreferNoSchedule"}); return true;
  }
  return false;
}

std::string canon_ip(const std::string& s) {
  int a, b, c, d; char tail;
  if (sscanf(s.c_str(), "%d.%d.%d.%d%c", &a, &b, &c, &d, &tail) == 4 && a >= 0 && a < 256 && b >= 0 && b < 256 && c >= 0 && c < 256 && d >= 0 && d < 256) { char buf[32]; snprintf(buf, sizeof buf, "%d.%d.%d.%d", a, b, c, d); return buf; }
  std::string r = s; for (auto& ch : r) ch = (char)tolower(ch);
  if (r == "0:0:0:0:0:0:0:0") r = "::";
  return r;
}

void sig_map(std::string& s, const StrMap& m) { for (auto& kv : m) { s += kv.first; s += '\1'; s += kv.second; s += '\2'; } s += '\3'; }
void sig_res(std::string& s, const ksp::ResList& r) { for (auto& kv : r) { s += kv.first; s += '\1'; s += std::to_string(kv.second); s += '\2'; } s += '\3'; }

// ---- host threads: the cores this process may use (a container usually sees every core of the machine but runs under a cgroup
// CPU quota, cpu.max = "quota period"; oversubscribing the quota is slower than one thread) ----
}  // namespace
uint32_t host_threads() {
  static const uint32_t n = [] {
    if (const char* e = getenv("KSH_THREADS")) { int v = atoi(e); if (v > 0) return (uint32_t)v; }
    uint32_t hw = std::max(1u, std::thread::hardware_concurrency());
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
      long long period = 0; char buf[64] = {0};
      if (fscanf(f, "%63s %lld", buf, &period) == 2 && strcmp(buf, "max") != 0) { long long quota = atoll(buf); if (quota > 0 && period > 0) hw = std::min<uint32_t>(hw, (uint32_t)std::max<long long>(1, quota / period)); }
      fclose(f);
    }
    return std::min(hw, 64u);      // (measured on a 256-thread host, round 6: 32 -> 11.9 ms, 64 -> 10.5 ms, 128 -> 12.2 ms for 100k pods; round 3: 16 -> 20.6, 32 -> 18.1, 64 -> 17.7)
  }();
  return n;
}
namespace {
// ---- persistent worker threads.  A flattening has a dozen short parallel phases (hashing, classing, sort runs, merges ...); spawning and
// joining std::threads for each costs more than several of the phases themselves.  The workers are created on first use and sleep on a
// condition variable between phases; one parallel region runs at a time (a second caller -- two Solves flattening concurrently -- simply
// spawns threads of its own, as does a process that inherited the pool object through fork() without its threads). ----
class WorkerPool {
 public:
  static WorkerPool& get(int which) { static WorkerPool p[2]; return p[which & 1]; }      // (a second pool: the queue sort runs beside the classing, encode_pods)
  // body(t) for t in [0, n): t = 0 runs on the caller; returns false (nothing run) when the pool cannot be used right now
  template <class B> bool run(uint32_t n, B&& body) {
    if (n <= 1 || getpid() != pid_ || busy_.exchange(true)) return false;
    ensure(n - 1);
    std::function<void(uint32_t)> fn = [&](uint32_t t) { body(t); };
    { std::lock_guard<std::mutex> g(m_); job_ = &fn; njobs_ = n; next_ = 1; pending_ = n - 1; ++gen_; }
    cv_.notify_all();
    body(0u);
    for (;;) {      // the caller helps with whatever is left, then waits for the stragglers
      uint32_t t; { std::lock_guard<std::mutex> g(m_); if (next_ >= njobs_) break; t = next_++; }
      body(t); { std::lock_guard<std::mutex> g(m_); --pending_; }
    }
    { std::unique_lock<std::mutex> g(m_); done_.wait(g, [&] { return pending_ == 0; }); job_ = nullptr; }
    busy_ = false; return true;
  }
 private:
  WorkerPool() : pid_(getpid()) {}
  ~WorkerPool() {
    if (getpid() != pid_) { for (auto& t : th_) t.detach(); return; }      // a forked child holds the object but not the threads (nor a usable mutex)
    { std::lock_guard<std::mutex> g(m_); stop_ = true; ++gen_; } cv_.notify_all();
    for (auto& t : th_) t.join();
  }
  void ensure(uint32_t want) { while (th_.size() < want) th_.emplace_back([this] { loop(); }); }
  void loop() {
    uint64_t seen = 0;
    for (;;) {
      std::unique_lock<std::mutex> g(m_);
      cv_.wait(g, [&] { return gen_ != seen; }); seen = gen_;
      if (stop_) return;
      while (job_ && next_ < njobs_) {
        const uint32_t t = next_++; auto* fn = job_;
        g.unlock(); (*fn)(t); g.lock();
        if (--pending_ == 0) done_.notify_all();
      }
    }
  }
  const pid_t pid_; std::atomic<bool> busy_{false};
  std::mutex m_; std::condition_variable cv_, done_; std::vector<std::thread> th_;
  std::function<void(uint32_t)>* job_ = nullptr; uint32_t njobs_ = 0, next_ = 0, pending_ = 0; uint64_t gen_ = 0; bool stop_ = false;
};
// body(t) for t in [0, n) on n threads (bodies must not throw)
thread_local int tl_pool = 0;      // which pool the parallel regions of this thread use
thread_local bool tl_sync_confirm = false;      // this thread's flattening confirms its spec merges in line (the retry after ConfirmFailed)
struct ConfirmFailed {};
template <class B> void run_threads(uint32_t n, B&& body) {
  if (n <= 1) { body(0u); return; }
  if (WorkerPool::get(tl_pool).run(n, body)) return;
  std::vector<std::thread> pool; for (uint32_t t = 1; t < n; ++t) pool.emplace_back([&, t] { body(t); });
  body(0u); for (auto& th : pool) th.join();
}
// fn(begin, end, thread) over [0, n) in contiguous chunks of at least `grain` items; an exception in a chunk is rethrown after the join
template <class F> void parallel_chunks(size_t n, F&& fn, size_t grain = 2048) {
  const uint32_t nt = (uint32_t)std::max<size_t>(1, std::min<size_t>(host_threads(), n / grain));
  if (nt == 1) { fn((size_t)0, n, 0u); return; }
  const size_t per = (n + nt - 1) / nt; std::vector<std::exception_ptr> errs(nt);
  run_threads(nt, [&](uint32_t t) { try { fn(std::min(n, t * per), std::min(n, (t + 1) * per), t); } catch (...) { errs[t] = std::current_exception(); } });
  for (auto& e : errs) if (e) std::rethrow_exception(e);
}

// ---- 128-bit streaming hash of everything of a pod spec that Solve can read (everything but uid and creationTimestamp).
// It only FINDS candidates for deduplication; equality is then confirmed field by field (same_spec), so a collision costs time,
// never correctness. ----
struct Hash128 {
  uint64_t a = 0x9E3779B97F4A7C15ull, b = 0xC2B2AE3D27D4EB4Full;
  static uint64_t rotl(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
  void u(uint64_t v) { a = (rotl(a, 27) ^ v) * 0x9FB21C651E98DF25ull; b = (rotl(b, 31) + v) * 0xD6E8FEB86659FD93ull; }
  void bytes(const char* p, size_t n) {
    u(n);
    while (n >= 8) { uint64_t v; memcpy(&v, p, 8); u(v); p += 8; n -= 8; }
    if (n) { uint64_t v = 0; memcpy(&v, p, n); u(v); }
  }
  void str(const std::string& s) { bytes(s.data(), s.size()); }
  void map(const StrMap& m) { u(m.size()); for (auto& kv : m) { str(kv.first); str(kv.second); } }
  void res(const ksp::ResList& r) { u(r.size()); for (auto& kv : r) { str(kv.first); u((uint64_t)kv.second); } }
  void exprs(const std::vector<Expr>& es) { u(es.size()); for (auto& e : es) { str(e.key); u((uint64_t)e.op); u(e.values.size()); for (auto& v : e.values) str(v); } }
  void selector(const ksp::Selector& x) { u(x.nil ? 1 : 2); if (!x.nil) { map(x.match_labels); exprs(x.match_exprs); } }
  void term(const ksp::AffinityTerm& t) { str(t.topology_key); u(t.namespaces.size()); for (auto& n : t.namespaces) str(n); selector(t.selector); }
  void finish() { a ^= a >> 32; a *= 0xFF51AFD7ED558CCDull; a ^= a >> 29; b ^= b >> 31; b *= 0xC4CEB9FE1A85EC53ull; b ^= b >> 33; }
};
Hash128 spec_hash(const Pod& p, const std::vector<uint32_t>* vol = nullptr) {
  Hash128 h; if (vol) { h.u(vol->size()); for (uint32_t e : *vol) h.u(e); }
  h.str(p.ns); h.map(p.labels); h.map(p.node_selector);
  h.u(p.required_affinity.size()); for (auto& t : p.required_affinity) h.exprs(t);
  h.u(p.preferred_affinity.size()); for (auto& t : p.preferred_affinity) { h.u((uint64_t)t.weight); h.exprs(t.exprs); }
  h.u(p.tolerations.size()); for (auto& t : p.tolerations) { h.str(t.key); h.str(t.op); h.str(t.value); h.str(t.effect); }
  h.u(p.containers.size()); for (auto& c : p.containers) { h.res(c.requests); h.res(c.limits); h.u(c.ports.size()); for (auto& hp : c.ports) { h.str(hp.ip); h.u((uint64_t)hp.port); h.str(hp.proto); } }
  h.u(p.init_containers.size()); for (auto& c : p.init_containers) { h.res(c.requests); h.res(c.limits); }
  h.u(p.spread.size()); for (auto& t : p.spread) { h.u((uint64_t)t.max_skew); h.str(t.key); h.u(t.schedule_anyway); h.selector(t.selector); }
  h.u(p.affinity_required.size()); for (auto& t : p.affinity_required) h.term(t);
  h.u(p.affinity_preferred.size()); for (auto& t : p.affinity_preferred) { h.u((uint64_t)t.weight); h.term(t.term); }
  h.u(p.anti_required.size()); for (auto& t : p.anti_required) h.term(t);
  h.u(p.anti_preferred.size()); for (auto& t : p.anti_preferred) { h.u((uint64_t)t.weight); h.term(t.term); }
  h.finish(); return h;
}
bool same_exprs(const std::vector<Expr>& a, const std::vector<Expr>& b) {
  if (a.size() != b.size()) return false;
  for (size_t i = 0; i < a.size(); ++i) if (a[i].key != b[i].key || a[i].op != b[i].op || a[i].values != b[i].values) return false;
  return true;
}
bool same_selector(const ksp::Selector& a, const ksp::Selector& b) { return a.nil == b.nil && (a.nil || (a.match_labels == b.match_labels && same_exprs(a.match_exprs, b.match_exprs))); }
bool same_term(const ksp::AffinityTerm& a, const ksp::AffinityTerm& b) { return a.topology_key == b.topology_key && a.namespaces == b.namespaces && same_selector(a.selector, b.selector); }
template <class V, class F> bool same_vec(const V& a, const V& b, F&& eq) { if (a.size() != b.size()) return false; for (size_t i = 0; i < a.size(); ++i) if (!eq(a[i], b[i])) return false; return true; }
// Field-by-field equality of two pod specs (uid and creationTimestamp excluded): the pods are interchangeable for Solve.
bool same_spec(const Pod& a, const Pod& b) {
  if (a.ns != b.ns || a.labels != b.labels || a.node_selector != b.node_selector) return false;
  if (!same_vec(a.required_affinity, b.required_affinity, same_exprs)) return false;
  if (!same_vec(a.preferred_affinity, b.preferred_affinity, [](const ksp::PreferredTerm& x, const ksp::PreferredTerm& y) { return x.weight == y.weight && same_exprs(x.exprs, y.exprs); })) return false;
  if (!same_vec(a.tolerations, b.tolerations, [](const ksp::Toleration& x, const ksp::Toleration& y) { return x.key == y.key && x.op == y.op && x.value == y.value && x.effect == y.effect; })) return false;
  auto same_ports = [](const ksp::HostPort& x, const ksp::HostPort& y) { return x.ip == y.ip && x.port == y.port && x.proto == y.proto; };
  if (!same_vec(a.containers, b.containers, [&](const ksp::Container& x, const ksp::Container& y) { return x.requests == y.requests && x.limits == y.limits && same_vec(x.ports, y.ports, same_ports); })) return false;
  if (!same_vec(a.init_containers, b.init_containers, [](const ksp::Container& x, const ksp::Container& y) { return x.requests == y.requests && x.limits == y.limits; })) return false;
  if (!same_vec(a.spread, b.spread, [](const ksp::Spread& x, const ksp::Spread& y) { return x.max_skew == y.max_skew && x.key == y.key && x.schedule_anyway == y.schedule_anyway && same_selector(x.selector, y.selector); })) return false;
  auto same_w = [](const ksp::WeightedTerm& x, const ksp::WeightedTerm& y) { return x.weight == y.weight && same_term(x.term, y.term); };
  return same_vec(a.affinity_required, b.affinity_required, same_term) && same_vec(a.affinity_preferred, b.affinity_preferred, same_w) &&
         same_vec(a.anti_required, b.anti_required, same_term) && same_vec(a.anti_preferred, b.anti_preferred, same_w);
}
uint64_t str_hash(std::string_view s) { Hash128 h; h.bytes(s.data(), s.size()); h.finish(); return h.a ^ h.b; }

struct Builder {
  Encoded& E; const ksp::Problem& pr; uint32_t flags;
  std::set<std::string> wellKnown;
  std::map<std::string, int> key_id; std::vector<std::set<std::string>> key_vals;   // pre-pass universes
  std::vector<std::set<std::string>> key_named; std::vector<std::set<long long>> key_bounds; std::set<std::string> topo_keys; bool passive_values = false;   // (value classes, see Encoded::key_members)
  std::map<std::string, int> res_id;
  std::map<std::string, int> taint_id;
  std::map<std::string, uint32_t> ip_id, proto_id;
  std::map<std::string, std::set<std::string>> domains;   // topology domain universe, provisioner.go:267-276
  std::map<std::string, int> hostname_to_existing;                   // hostname -> row of the existing-node tables (single problems and the snapshot base)
  std::map<std::string, int> hostname_to_node;                       // base only: hostname -> node index (what-ifs translate through existing_row)
  std::vector<int> existing_row;                                     // what-if mode: node index -> row of THIS what-if's existing-node tables (-1: left / not owned)
  std::vector<uint8_t> node_owned;                                   // base only: state.Node.Owned() per node
  bool any_volume_limits = false;                                    // base only
  int existing_of_hostname(const std::string& h) const {
    if (base) { auto it = base->hostname_to_node.find(h); return it == base->hostname_to_node.end() ? -1 : existing_row[it->second]; }
    auto it = hostname_to_existing.find(h); return it == hostname_to_existing.end() ? -1 : it->second;
  }
  const ksp::StateNode* node_named(const std::string& n) const {
    const auto& m = base ? base->node_by_name : node_by_name; auto it = m.find(n); return it == m.end() ? nullptr : it->second;
  }
  std::vector<std::unique_ptr<Group>> groups; std::map<std::string, int> topo_by_id, inverse_by_id;   // creation order; inverse flagged
  std::vector<int> group_order, group_remap;          // encoded group index -> creation index, and back (encode_groups: topologies first, then inverse groups)
  bool shared_filter_differs = false;                 // two pods share a spread group while their node filters differ in content (the first one's counts)
  std::vector<Requirement> it_reqs; std::map<std::string, int> it_state_id;   // node-side instance-type states (index 0 = absent)
  std::vector<Requirement> it_cols; std::map<std::string, int> it_col_id;     // pod-side instance-type requirements (classes, topology filters; 0 = none)
  // UIDs of the batch: open-addressing table of pod indices (topology.go:66-70 excludes the batch from countDomains)
  struct UidSet {
    const std::vector<std::string_view>* uids = nullptr; std::vector<uint32_t> tab; uint64_t mask = 0;
    bool count(const std::string& uid) const {
      if (tab.empty()) return false;
      for (uint64_t i = str_hash(uid) & mask;; i = (i + 1) & mask) { const uint32_t e = tab[i]; if (!e) return false; if ((*uids)[e - 1] == uid) return true; }
    }
    int64_t find(const std::string& uid) const {      // the batch pod with that uid, -1 if none
      if (tab.empty()) return -1;
      for (uint64_t i = str_hash(uid) & mask;; i = (i + 1) & mask) { const uint32_t e = tab[i]; if (!e) return -1; if ((*uids)[e - 1] == uid) return (int64_t)e - 1; }
    }
  } batch_uids;
  std::map<std::string, const ksp::StateNode*> node_by_name;
  bool toleratePreferNoSchedule = false;
  uint32_t K = 0, R = 0, T = 0, TW = 0;

  // The pending batch: every pod of the problem, or -- for a what-if flattened over a shared snapshot -- the pods of its candidate nodes.
  std::vector<const Pod*> podp;                       // (binary ingress: pod i points at its SPEC, ksp::PodBatch::specs)
  const ksp::PodBatch* lite = nullptr;                // binary ingress: the batch in compact form (uids / timestamps live there)
  std::vector<std::string_view> uidv;                 // uid of pod i (a view into the pod object or the batch's uid bytes)
  int64_t ts_of(size_t i) const { return lite ? lite->ts[i] : podp[i]->creation_ts; }
  const Builder* base = nullptr;                      // what-if mode: the finished flattening of the whole snapshot
  bool env_mode = false;                              // `base` is a cached flattening of the SAME environment for another batch (EnvCache): nothing leaves, the pods are new
  std::vector<uint8_t> env_removed;                   // env mode: the nodes that are not in state, in the role of `removed'
  const std::vector<uint8_t>* removed = nullptr;      // what-if mode: nodes that leave the state-node list (helpers.go:48-61)
  bool node_in_state(size_t i) const { return removed ? !(*removed)[i] : pr.nodes[i].in_state; }
  std::vector<uint32_t> pod_rank;                     // base only: a pod's position in the snapshot-wide queue order
  std::vector<int> base_existing_of;                  // base only: node index -> row of the base's existing-node tables (-1: not owned)
  std::vector<ksp::ResList> base_remaining;           // base only: remainingResources with every node in state
  // ---- a snapshot's flattening after ksh_env_apply (round 6): `prev` is the flattening of the SAME ksp::Problem object before the events -- pods and nodes were
  // appended to it or left in place as tombstones, nothing moved.  Whatever of this run is a function of things that did not change is taken from it, provided the
  // universes come out the same (`warm`); every shortcut reproduces what the full run would have written, byte for byte (tests/test_env_apply.py compares fingerprints).
  const Builder* prev = nullptr; bool warm = false, keep_warm_state = false;
  std::vector<Hash128> spec_hs; std::vector<int32_t> spec_tab; std::vector<uint32_t> spec_first;      // kept by dedupe_specs: the hash of every spec's first pod, the table over them, the first pods
  std::map<std::string, int> pre_it_state_id, pre_it_col_id;      // kept by encode_it_states: the lattice's states / columns before its closure
  std::string act_sig; std::map<std::string, int> act_key_id, act_res_id; size_t n_nodes_built = 0, n_pods_built = 0;      // kept by run(): what collect_active left, how large the problem was

  Builder(Encoded& e, uint32_t f) : E(e), pr(*e.src), flags(f), lite(e.batch.get()) {}
  std::chrono::steady_clock::time_point tl_ = std::chrono::steady_clock::now();
  void sublap(const char* what) { if (!getenv("KSH_TIMING")) return; auto t1 = std::chrono::steady_clock::now(); fprintf(stderr, "      . %-26s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(t1 - tl_).count()); tl_ = t1; }

  // ---------- universes ----------
  int key_of(const std::string& k, bool create) {
    auto it = key_id.find(k); if (it != key_id.end()) return it->second;
    if (!create) return -1;
    int id = (int)key_vals.size(); key_id[k] = id; key_vals.emplace_back(); key_named.emplace_back(); key_bounds.emplace_back(); return id;
  }
  void note_value(int id, const std::string& v) { key_vals[id].insert(v); if (!passive_values) key_named[id].insert(v); }
  static bool special_key(const std::string& k) { return k == ksp::kHostname || k == ksp::kInstanceType; }
  void note_expr(const Expr& e) {
    std::string k = ksp::normalize_key(e.key); if (special_key(k)) return;
    int id = key_of(k, true);
    if (e.op == Op::In || e.op == Op::NotIn) for (auto& v : e.values) note_value(id, v);
    if ((e.op == Op::Gt || e.op == Op::Lt) && !e.values.empty()) { long long x; if (Atoi(e.values[0], &x)) key_bounds[id].insert(x); }
  }
  void note_label(const std::string& key, const std::string& v) { std::string k = ksp::normalize_key(key); if (special_key(k)) return; note_value(key_of(k, true), v); }
  int res_of(const std::string& r) { auto it = res_id.find(r); if (it != res_id.end()) return it->second; int id = (int)res_id.size(); res_id[r] = id; return id; }
  void note_res(const ksp::ResList& l) { for (auto& kv : l) res_of(kv.first); }
  void note_pod(const Pod& p) {
    for (auto& kv : p.node_selector) note_label(kv.first, kv.second);
    for (auto& t : p.required_affinity) for (auto& e : t) note_expr(e);
    for (auto& t : p.preferred_affinity) for (auto& e : t.exprs) note_expr(e);
    for (auto& c : p.containers) { note_res(c.requests); note_res(c.limits); }
    for (auto& c : p.init_containers) { note_res(c.requests); note_res(c.limits); }
    auto topo_key = [&](const std::string& k) { if (k == ksp::kInstanceType) throw Unsupported("topology key node.kubernetes.io/instance-type"); if (k != ksp::kHostname) { key_of(k, true); topo_keys.insert(k); } };
    for (auto& s : p.spread) topo_key(s.key);
    for (auto& t : p.affinity_required) topo_key(t.topology_key);
    for (auto& t : p.affinity_preferred) topo_key(t.term.topology_key);
    for (auto& t : p.anti_required) topo_key(t.topology_key);
    for (auto& t : p.anti_preferred) topo_key(t.term.topology_key);
  }

  void collect_active() {
    for (auto k : kBuiltinWellKnown) wellKnown.insert(k);
    for (auto& k : pr.extra_well_known) wellKnown.insert(k);
    res_of("cpu"); res_of("memory"); res_of("pods");
    key_of(ksp::kZone, true); key_of(ksp::kCapacityType, true);
    for (auto& p : pr.provisioners) {
      for (auto& e : p.requirements) { if (ksp::normalize_key(e.key) == ksp::kHostname) throw Unsupported("provisioner requirement on kubernetes.io/hostname"); note_expr(e); }
      for (auto& kv : p.labels) { if (ksp::normalize_key(kv.first) == ksp::kHostname) throw Unsupported("provisioner label kubernetes.io/hostname"); note_label(kv.first, kv.second); }
      note_label(ksp::kProvisionerName, p.name);
      if (p.has_limits) note_res(p.limits);
      for (auto& t : p.taints) if (t.effect == "PreferNoSchedule") toleratePreferNoSchedule = true;
    }
    for (auto& si : specs) note_pod(si.stages[0].spec);      // every pod is one of the distinct specs (dedupe_specs), first occurrences in pod order
    for (auto& p : pr.daemons) note_pod(p);
    for (auto& cp : pr.cluster_pods) for (auto& t : cp.anti_required) { if (t.topology_key != ksp::kHostname) { key_of(t.topology_key, true); topo_keys.insert(t.topology_key); } }
    passive_values = true;      // from here on values join a universe without being named by anything that could tell them apart
  }
  // What the batch, the provisioners and the daemonsets contribute to the universes, in canonical form: two batches with equal signatures flatten
  // against the SAME catalogue encoding (EnvCache, encode()).
  std::string active_signature() const {
    std::string s; std::vector<std::pair<std::string, int>> ks(key_id.begin(), key_id.end());      // (std::map: ascending key names)
    for (auto& kv : ks) { s += kv.first; s += '\1'; for (auto& v : key_named[kv.second]) { s += v; s += '\2'; } s += '\3'; for (long long b : key_bounds[kv.second]) { s += std::to_string(b); s += ','; } s += '\4'; }
    s += '\5'; for (auto& k : topo_keys) { s += k; s += '\1'; } s += '\5'; for (auto& kv : res_id) { s += kv.first; s += '\1'; } s += toleratePreferNoSchedule ? "T" : "F";
    return s;
  }
  void collect_universes() { collect_active(); collect_passive(); }
  // May this run continue `prev`?  After collect_active: the same problem object, what the batch / provisioners / daemonsets name unchanged (keys with their ids,
  // named values, bounds, topology keys, resources), and nothing a new node carries is new to a universe.
  bool can_continue() const {
    if (!prev || &prev->pr != &pr || prev->flags != flags || prev->act_sig.empty() || prev->base || prev->lite) return false;
    if (key_id != prev->act_key_id || res_id != prev->act_res_id || active_signature() != prev->act_sig) return false;
    if (prev->n_nodes_built > pr.nodes.size() || prev->T != pr.instance_types.size()) return false;
    for (size_t i = prev->n_nodes_built; i < pr.nodes.size(); ++i) {
      const auto& n = pr.nodes[i];
      for (auto& kv : n.labels) { const std::string k = ksp::normalize_key(kv.first);
        auto it = prev->key_id.find(k); if (it != prev->key_id.end() && !prev->key_vals[it->second].count(kv.second)) return false;
        auto raw = prev->key_id.find(kv.first); if (raw != prev->key_id.end() && raw->first != k && !prev->key_vals[raw->second].count(kv.second)) return false; }
      for (const ksp::ResList* l : {&n.available, &n.capacity, &n.daemonset_requests}) for (auto& kv : *l) if (!prev->res_id.count(kv.first)) return false;
    }
    return true;
  }
  void collect_passive() {
    if (warm) {      // the catalogue's and the old nodes' contributions are in the previous universes, the new nodes add nothing (can_continue): adopt them and the tables made from them
      const Encoded& B = prev->E;
      key_vals = prev->key_vals; res_id = prev->res_id; K = prev->K; R = prev->R; T = prev->T; TW = prev->TW;
      E.key_names = B.key_names; E.key_values = B.key_values; E.key_nvalues = B.key_nvalues; E.value_int = B.value_int; E.key_members = B.key_members; E.key_class = B.key_class; E.key_ints = B.key_ints; E.res_names = B.res_names;
      return;
    }
    // Instance types last: a label key that ONLY instance types carry (real catalogues have many, often with hundreds of
    // values -- the fake provider's `integer` has one per type) can never meet a node requirement: node requirements come from
    // provisioners, pods, topology keys and existing-node labels, and Intersects / Compatible only look at keys both sides
    // have (requirements.go:123-133,189-206).  Such keys are left out of the encoding altogether.
    for (auto& it : pr.instance_types) {
      for (auto& e : it.requirements) {
        if (e.op == Op::Gt || e.op == Op::Lt) throw Unsupported("instance type requirement with Gt/Lt bounds");
        const std::string k = ksp::normalize_key(e.key);
        if (special_key(k) || key_id.count(k)) note_expr(e);
      }
      for (auto& o : it.offerings) { note_value(key_of(ksp::kZone, true), o.zone); note_value(key_of(ksp::kCapacityType, true), o.capacity_type); }
      note_res(it.capacity); note_res(it.overhead);
    }
    // node labels: only keys something else references matter (existing-node requirements are never
    // returned); values of referenced keys join the universe (they become topology domains / In sets)
    for (auto& n : pr.nodes) {
      for (auto& kv : n.labels) { std::string k = ksp::normalize_key(kv.first); auto it = key_id.find(k); if (it != key_id.end()) note_value(it->second, kv.second);
        auto raw = key_id.find(kv.first); if (raw != key_id.end() && raw->first != k) note_value(raw->second, kv.second); }
      note_res(n.available); note_res(n.capacity); note_res(n.daemonset_requests);
    }
    K = (uint32_t)key_vals.size(); R = (uint32_t)res_id.size(); T = (uint32_t)pr.instance_types.size(); TW = (T + 63) / 64;
    if (K > KS_MAX_KEYS) throw Unsupported("more than 32 distinct label keys on the path");
    if (R > KS_MAX_RES) throw Unsupported("more than 8 distinct resource names");
    E.key_names.assign(K, ""); for (auto& kv : key_id) E.key_names[kv.second] = kv.first;
    E.key_values.resize(K); E.key_nvalues.assign(K, 0); E.value_int.assign((size_t)K * 64, INT32_MIN); E.key_members.assign(K, {}); E.key_class.assign(K, {}); E.key_ints.assign(K, {});
    for (uint32_t k = 0; k < K; ++k) {
      if (key_vals[k].size() > 64) {
        // Value classes.  Only for keys whose values are never told apart one by one: not a topology key (domains are counted per value, and a
        // requirement with exactly one value is a domain choice), not the zone / capacity type (offerings are looked up per value).
        const std::string& kn = E.key_names[k];
        if (topo_keys.count(kn) || kn == ksp::kZone || kn == ksp::kCapacityType) throw Unsupported("label key " + kn + " has more than 64 distinct values and is a topology / offering key");
        const std::vector<long long> bounds(key_bounds[k].begin(), key_bounds[k].end());
        std::map<std::string, std::vector<std::string>> cls;      // class id (sortable) -> members
        for (auto& v : key_vals[k]) {
          std::string id;
          if (key_named[k].count(v)) id = "n:" + v;
          else { long long x; if (!Atoi(v, &x)) id = "p:~"; else { const size_t lo = std::lower_bound(bounds.begin(), bounds.end(), x) - bounds.begin(); char buf[48]; snprintf(buf, sizeof buf, "p:%08zu%c", lo, (lo < bounds.size() && bounds[lo] == x) ? '=' : '<'); id = buf; } }
          cls[id].push_back(v);
        }
        if (cls.size() > 64) throw Unsupported("label key " + kn + " has more than 64 values that pods / provisioners name or bound apart");
        // class order = ascending representative (the smallest member), like any other universe
        std::vector<std::pair<std::string, std::vector<std::string>>> ordered;
        for (auto& kv : cls) { auto mem = kv.second; std::sort(mem.begin(), mem.end()); ordered.emplace_back(mem.front(), std::move(mem)); }
        std::sort(ordered.begin(), ordered.end());
        E.key_members[k].resize(ordered.size());
        for (size_t c = 0; c < ordered.size(); ++c) { E.key_values[k].push_back(ordered[c].first); E.key_members[k][c] = ordered[c].second; for (auto& m : ordered[c].second) E.key_class[k][m] = (int)c; }
        E.key_nvalues[k] = (uint32_t)E.key_values[k].size();
      } else { E.key_values[k].assign(key_vals[k].begin(), key_vals[k].end()); E.key_nvalues[k] = (uint32_t)E.key_values[k].size(); }
      // integers of the key: label values that parse (a value class: its representative stands for every member) and every Gt / Lt bound
      std::set<long long> ints(key_bounds[k].begin(), key_bounds[k].end()); bool wide = false;
      for (auto& v : E.key_values[k]) { long long x; if (Atoi(v, &x)) ints.insert(x); }
      for (long long x : ints) if (x <= INT32_MIN + 1 || x >= INT32_MAX - 1) wide = true;
      if (wide) E.key_ints[k].assign(ints.begin(), ints.end());
      for (size_t v = 0; v < E.key_values[k].size(); ++v) { long long x; if (Atoi(E.key_values[k][v], &x)) E.value_int[k * 64 + v] = int_code((int)k, x); }
    }
    E.res_names.assign(R, ""); for (auto& kv : res_id) E.res_names[kv.second] = kv.first;
  }

  // the kernel-side code of integer x on key k: x itself, or its rank among the key's integers (Encoded::key_ints)
  int32_t int_code(int k, long long x) const {
    if ((size_t)k >= E.key_ints.size() || E.key_ints[k].empty()) { if (x <= INT32_MIN + 1 || x >= INT32_MAX - 1) throw Unsupported("integer outside int32 on a key whose universe was closed without it"); return (int32_t)x; }
    auto& v = E.key_ints[k]; auto it = std::lower_bound(v.begin(), v.end(), x);
    if (it == v.end() || *it != x) throw std::logic_error("integer missing from the key's universe");
    return (int32_t)(it - v.begin());
  }
  int value_id(int k, const std::string& v) const {
    if ((size_t)k < E.key_class.size() && !E.key_class[k].empty()) { auto c = E.key_class[k].find(v); return c == E.key_class[k].end() ? -1 : c->second; }
    auto& vs = E.key_values[k]; auto it = std::lower_bound(vs.begin(), vs.end(), v);
    if (it == vs.end() || *it != v) return -1; return (int)(it - vs.begin());
  }

  // ---------- requirement encoding ----------
  // What-if mode: the snapshot's lattice is consulted in place; it is copied only if the what-if needs a state / column the snapshot lacks.
  bool lattice_adopted = false;
  void adopt_lattice() { it_reqs = base->it_reqs; it_state_id = base->it_state_id; it_cols = base->it_cols; it_col_id = base->it_col_id; lattice_adopted = true; }
  int it_state_of(const Requirement& r) {
    std::string id = r.identity();
    if (base && !lattice_adopted) { auto bt = base->it_state_id.find(id); if (bt != base->it_state_id.end()) return bt->second; adopt_lattice(); }
    auto it = it_state_id.find(id); if (it != it_state_id.end()) return it->second;
    if (it_reqs.size() >= KS_MAX_ITSTATES) throw Unsupported("more than 65534 distinct instance-type requirements (closure)");
    int s = (int)it_reqs.size(); it_reqs.push_back(r); it_state_id[id] = s; return s;
  }
  int it_col_of(const Requirement& r) {
    std::string id = r.identity();
    if (base && !lattice_adopted) { auto bt = base->it_col_id.find(id); if (bt != base->it_col_id.end()) return bt->second; adopt_lattice(); }
    auto it = it_col_id.find(id); if (it != it_col_id.end()) return it->second;
    if (it_cols.size() >= KS_MAX_ITSTATES) throw Unsupported("more than 65534 distinct pod-side instance-type requirements");
    int s = (int)it_cols.size(); it_cols.push_back(r); it_col_id[id] = s; return s;
  }
  // Appends one requirement set; returns its index.  `hn` receives the hostname requirement if any.
  uint32_t push_reqs(ReqSetsStore& st, const Requirements& rs, const Requirement** hn, bool allow_hostname, bool pod_side) {
    uint32_t idx = st.n++; st.present.push_back(0); st.complement.push_back(0); st.it_state.push_back(0);
    st.mask.resize((size_t)st.n * K, 0); st.gt.resize((size_t)st.n * K, KS_NO_BOUND_GT); st.lt.resize((size_t)st.n * K, KS_NO_BOUND_LT);
    if (hn) *hn = nullptr;
    for (auto& kv : rs.m) {
      const Requirement& r = kv.second;
      if (kv.first == ksp::kHostname) { if (!allow_hostname) throw Unsupported("hostname requirement outside a pod spec"); if (hn) *hn = &kv.second; continue; }
      if (kv.first == ksp::kInstanceType) { st.it_state[idx] = pod_side ? it_col_of(r) : it_state_of(r); continue; }
      int k = key_of(kv.first, false); if (k < 0) throw std::logic_error("key missing from universe: " + kv.first);
      st.present[idx] |= 1u << k; if (r.complement) st.complement[idx] |= 1u << k;
      uint64_t m = 0; for (auto& v : r.values) { int vid = value_id(k, v); if (vid < 0) throw std::logic_error("value missing from universe: " + v); m |= 1ull << vid; }
      st.mask[(size_t)idx * K + k] = m;
      if (r.greaterThan) st.gt[(size_t)idx * K + k] = int_code(k, *r.greaterThan);
      if (r.lessThan) st.lt[(size_t)idx * K + k] = int_code(k, *r.lessThan);
    }
    return idx;
  }
  Requirements restrict_to_known_keys(const Requirements& rs) const {   // drop keys nothing references (existing-node labels)
    Requirements out; for (auto& kv : rs.m) if (kv.first == ksp::kInstanceType || key_id.count(kv.first)) out.m.emplace(kv.first, kv.second); return out;
  }
  void res_vec(const ksp::ResList& l, std::vector<int64_t>& out, uint32_t* present) {
    size_t base = out.size(); out.resize(base + R, 0); uint32_t p = 0;
    for (auto& kv : l) { int r = res_id.at(kv.first); out[base + r] = kv.second; p |= 1u << r; }
    if (present) *present = p;
  }
  uint64_t taint_mask(const std::vector<ksp::Taint>& ts) {
    uint64_t m = 0;
    for (auto& t : ts) { std::string id = t.key + "\1" + t.value + "\1" + t.effect; auto it = taint_id.find(id); int i;
      if (it == taint_id.end()) { if (taint_id.size() >= 64) throw Unsupported("more than 64 distinct taints"); i = (int)taint_id.size(); taint_id[id] = i; taints.push_back(t); } else i = it->second;
      m |= 1ull << i; }
    return m;
  }
  std::vector<ksp::Taint> taints;
  uint64_t port_entry(const std::string& ip_in, int32_t port, const std::string& proto) {
    std::string ip = canon_ip(ip_in.empty() ? "0.0.0.0" : ip_in);
    uint32_t iid = 0; if (!(ip == "0.0.0.0" || ip == "::")) { auto it = ip_id.find(ip); if (it == ip_id.end()) { iid = (uint32_t)ip_id.size() + 1; ip_id[ip] = iid; } else iid = it->second; }
    uint32_t pid; auto pt = proto_id.find(proto); if (pt == proto_id.end()) { pid = (uint32_t)proto_id.size() + 1; if (pid > 255) throw Unsupported("too many protocols"); proto_id[proto] = pid; } else pid = pt->second;
    return ((uint64_t)pid << 56) | ((uint64_t)(uint32_t)port << 32) | iid;
  }

  // ---------- instance types ----------
  std::shared_ptr<std::vector<Requirements>> it_requirements_p = std::make_shared<std::vector<Requirements>>();      // (shared with the flattening that continues this one, `prev`)
  void encode_instance_types() {
    if (warm) {      // same catalogue object, same universes: the arrays are the previous flattening's
      const Encoded& B = prev->E;
      E.it_present = B.it_present; E.it_complement = B.it_complement; E.it_mask = B.it_mask; E.it_offer = B.it_offer; E.it_alloc = B.it_alloc; E.it_cap = B.it_cap; E.it_price = B.it_price; E.it_price_lo = B.it_price_lo;
      it_requirements_p = prev->it_requirements_p;
      return;
    }
    std::vector<Requirements>& it_requirements = *it_requirements_p;
    E.it_present.assign(T, 0); E.it_complement.assign(T, 0); E.it_mask.assign((size_t)K * T, 0); E.it_offer.assign(T, 0);
    E.it_alloc.assign((size_t)R * T, 0); E.it_cap.assign((size_t)R * T, 0);
    const int kz = key_id.at(ksp::kZone), kc = key_id.at(ksp::kCapacityType);
    const uint32_t nct = E.key_nvalues[kc];
    if ((uint64_t)E.key_nvalues[kz] * nct > 64 || nct > 32) throw Unsupported("more than 64 zone x capacity-type pairs");
    const uint32_t NP = (uint32_t)E.key_nvalues[kz] * nct; E.it_price.assign((size_t)T * NP, -1.0); E.it_price_lo.assign((size_t)T * NP, 1.7976931348623157e308);
    it_requirements.resize(T);
    parallel_chunks(T, [&](size_t tb_, size_t te_, uint32_t) { for (uint32_t t = (uint32_t)tb_; t < (uint32_t)te_; ++t) {      // (a type only writes its own column / row)
      const auto& it = pr.instance_types[t];
      Requirements rs = Requirements::FromExprs(it.requirements); it_requirements[t] = rs;
      for (auto& kv : rs.m) {
        if (kv.first == ksp::kInstanceType) continue;
        if (kv.first == ksp::kHostname) throw Unsupported("instance type requirement on hostname");
        auto kf = key_id.find(kv.first); if (kf == key_id.end()) continue;      // a key nothing else references (collect_universes)
        int k = kf->second;
        E.it_present[t] |= 1u << k; if (kv.second.complement) E.it_complement[t] |= 1u << k;
        uint64_t m = 0; for (auto& v : kv.second.values) m |= 1ull << value_id(k, v);
        E.it_mask[(size_t)k * T + t] = m;
      }
      for (auto& o : it.offerings) if (o.available) {
        const uint32_t pair = value_id(kz, o.zone) * nct + value_id(kc, o.capacity_type);
        E.it_offer[t] |= 1ull << pair;
        double& pp = E.it_price[(size_t)t * NP + pair]; if (o.price > pp) pp = o.price;     // worstLaunchPrice takes the maximum (helpers.go:303,312)
        double& pl = E.it_price_lo[(size_t)t * NP + pair]; if (o.price < pl) pl = o.price;  // Offerings.Cheapest the minimum (types.go:141)
      }
      ksp::ResList alloc = Subtract(it.capacity, it.overhead);   // Allocatable(), types.go:87-89
      for (auto& kv : alloc) E.it_alloc[(size_t)res_id.at(kv.first) * T + t] = kv.second;
      for (auto& kv : it.capacity) E.it_cap[(size_t)res_id.at(kv.first) * T + t] = kv.second;
    } }, 128);
  }

  // ---------- templates ----------
  std::vector<Requirements> tmpl_reqs;
  void encode_templates() {
    for (auto& p : pr.provisioners) E.templates.push_back(&p);
    std::stable_sort(E.templates.begin(), E.templates.end(), [](const ksp::Provisioner* a, const ksp::Provisioner* b) { return a->weight > b->weight; });   // OrderByWeight
    if (E.templates.empty()) throw ksp::Error("no provisioners found");
    const uint32_t M = (uint32_t)E.templates.size(); const std::vector<Requirements>& it_requirements = *it_requirements_p;
    if (warm) domains = prev->domains;      // (only this function writes it, from the provisioners and the catalogue)
    E.tmpl_types.assign((size_t)M * TW, 0);
    for (uint32_t m = 0; m < M; ++m) {
      const auto& p = *E.templates[m];
      Requirements rs; rs.Add(Requirements::FromExprs(p.requirements));   // NewMachineTemplate, machinetemplate.go:46-62
      StrMap labels = p.labels; labels[ksp::kProvisionerName] = p.name; rs.Add(Requirements::FromLabels(labels));
      tmpl_reqs.push_back(rs);
      push_reqs(E.tmpl, rs, nullptr, false, false);
      E.tmpl_taints.push_back(taint_mask(p.taints));
      for (int idx : p.instance_types) { if (idx < 0 || (uint32_t)idx >= T) throw ksp::Error("instance type index out of range"); E.tmpl_types[(size_t)m * TW + idx / 64] |= 1ull << (idx % 64); }
      // topology domain universe, provisioner.go:267-276
      // (a catalogue repeats a handful of zones / architectures / ... thousands of times: values already seen for a key are recognised by a
      // hashed view of the set's own strings before the ordered set is asked)
      if (!warm) { struct Seen { const std::string* key; std::set<std::string>* dom; std::unordered_set<std::string_view> vals; };
        std::vector<Seen> seen;
        for (int idx : p.instance_types) for (auto& kv : it_requirements[idx].m) {
          Seen* sn = nullptr; for (auto& c : seen) if (*c.key == kv.first) { sn = &c; break; }
          if (!sn) { seen.push_back(Seen{&kv.first, &domains[kv.first], {}}); sn = &seen.back(); for (auto& have : *sn->dom) sn->vals.insert(std::string_view(have)); }
          for (auto& v : kv.second.values) if (!sn->vals.count(std::string_view(v))) { auto ins = sn->dom->insert(v); sn->vals.insert(std::string_view(*ins.first)); }
        } }
      Requirements preq = Requirements::FromExprs(p.requirements);
      if (!warm) for (auto& kv : preq.m) if (kv.second.Operator() == Op::In) for (auto& v : kv.second.values) domains[kv.first].insert(v);
    }
  }

  // ---------- volumes: ExistingNode.Add's volumeUsage.Validate + VolumeCount.Exceeds (existingnode.go:87-94, volumeusage.go:102-143) ----------
  // Only existing nodes track volumes.  A claim id counts once per node however many pods mount it, so the flat form distinguishes
  //   shared claims   -- mounted by two or more batch pods, or already on a state node: one bit each in a per-node set;
  //   unique claims   -- mounted by exactly one batch pod and on no node (every generic ephemeral volume, a StatefulSet's claims): a count.
  // Pods that differ only in the NAMES of their unique claims then still share a class.  Drivers no in-state node limits are dropped.
  std::map<std::string, int> vol_driver_id;
  std::unordered_map<std::string, uint32_t> vol_ref, vol_shared;
  bool pods_have_volumes = false; int blocked_taint = -1;
  static std::string vol_pair(const ksp::Volume& v) { return v.driver + '\1' + v.pvc; }
  void collect_volumes() {
    if (base && !base->any_volume_limits) { for (auto* p : podp) if (p->volume_error) { pods_have_volumes = true; break; } return; }      // no node limits a volume: only a failed lookup matters
    for (size_t i = 0; i < pr.nodes.size(); ++i) if (node_in_state(i) && pr.nodes[i].owned()) for (auto& kv : pr.nodes[i].volume_limits) if (!vol_driver_id.count(kv.first)) { const int id = (int)vol_driver_id.size(); vol_driver_id[kv.first] = id; }
    if (vol_driver_id.size() > 64) throw Unsupported("more than 64 CSI drivers with volume limits");
    { std::atomic<bool> any{false};      // (a scan over every pod object: on the worker threads)
      parallel_chunks(podp.size(), [&](size_t b, size_t e, uint32_t) { for (size_t i = b; i < e && !any.load(std::memory_order_relaxed); ++i) if (podp[i]->volume_error || !podp[i]->volumes.empty()) any = true; }, 16384);
      pods_have_volumes = any; }
    if (!pods_have_volumes || vol_driver_id.empty()) return;
    std::unordered_set<std::string> on_node;
    for (size_t i = 0; i < pr.nodes.size(); ++i) if (node_in_state(i) && pr.nodes[i].owned()) for (auto& v : pr.nodes[i].volumes) if (vol_driver_id.count(v.driver)) on_node.insert(vol_pair(v));
    std::vector<std::string> mine;
    for (auto* p : podp) {
      mine.clear(); for (auto& v : p->volumes) if (vol_driver_id.count(v.driver)) mine.push_back(vol_pair(v));
      std::sort(mine.begin(), mine.end()); mine.erase(std::unique(mine.begin(), mine.end()), mine.end());
      for (auto& s2 : mine) ++vol_ref[s2];
    }
    for (auto* p : podp) for (auto& v : p->volumes) if (vol_driver_id.count(v.driver)) {
      std::string s2 = vol_pair(v);
      if ((vol_ref[s2] >= 2 || on_node.count(s2)) && !vol_shared.count(s2)) { const uint32_t id = (uint32_t)vol_shared.size(); vol_shared.emplace(std::move(s2), id); }
    }
    if (vol_shared.size() >= (1u << 24)) throw Unsupported("more than 16M shared volume claims");
  }
  // What ExistingNode.Add needs of a pod's volumes: tagged words ordered by driver (ks_problem.vol_list).
  std::vector<uint32_t> vol_entries(const Pod& p) const {
    std::vector<uint32_t> out;
    if (p.volume_error) { out.push_back(0xFFFFFFFFu); return out; }
    if (p.volumes.empty() || vol_driver_id.empty()) return out;
    std::set<std::string> seen; std::map<int, uint32_t> uniq;
    for (auto& v : p.volumes) {
      auto d = vol_driver_id.find(v.driver); if (d == vol_driver_id.end()) continue;
      std::string s2 = vol_pair(v); if (!seen.insert(s2).second) continue;
      auto sh = vol_shared.find(s2);
      if (sh != vol_shared.end()) out.push_back(((uint32_t)d->second << 24) | sh->second); else uniq[d->second]++;
    }
    for (auto& kv : uniq) { if (kv.second >= (1u << 24)) throw Unsupported("a pod with more than 16M volumes"); out.push_back(0x80000000u | ((uint32_t)kv.first << 24) | kv.second); }
    std::sort(out.begin(), out.end(), [](uint32_t a, uint32_t b) { const uint32_t da = (a >> 24) & 63u, db = (b >> 24) & 63u; return da != db ? da < db : a < b; });
    return out;
  }
  bool same_pod(const Pod& a, const Pod& b) const { return same_spec(a, b) && (!pods_have_volumes || vol_entries(a) == vol_entries(b)); }
  // A node whose mounted volumes already exceed one of its limits refuses every pod (Exceeds ranges over the union's drivers, whatever the
  // pod adds): such nodes carry a taint of their own that no toleration matches.
  static bool over_volume_limit(const ksp::StateNode& n) {
    if (n.volume_limits.empty() || n.volumes.empty()) return false;
    std::map<std::string, std::set<std::string>> u; for (auto& v : n.volumes) u[v.driver].insert(v.pvc);
    for (auto& kv : n.volume_limits) { auto it = u.find(kv.first); if (it != u.end() && (int64_t)it->second.size() > (int64_t)kv.second) return true; }
    return false;
  }
  uint64_t blocked_mask() {
    if (blocked_taint < 0) { ksp::Taint t{std::string("\1volume-limits-exceeded"), "", "NoSchedule"}; taint_mask({t}); blocked_taint = taint_id.at(t.key + "\1" + t.value + "\1" + t.effect); }
    return 1ull << blocked_taint;
  }
  void encode_volumes() {
    const uint32_t NE = (uint32_t)E.existing.size(), ND = pods_have_volumes ? (uint32_t)vol_driver_id.size() : 0u, SW = (uint32_t)((vol_shared.size() + 63) / 64);
    E.prob.ND = ND; E.prob.SW = ND ? SW : 0;
    if (E.cls_vol_off.size() != (size_t)E.cls.n + 1) E.cls_vol_off.resize((size_t)E.cls.n + 1, (uint32_t)E.vol_list.size());
    if (!ND) return;
    E.en_vol_limit.assign((size_t)NE * ND, INT32_MAX); E.en_vol_count.assign((size_t)NE * ND, 0); E.en_vol_set.assign((size_t)NE * SW, 0);
    for (uint32_t e = 0; e < NE; ++e) {
      const auto& n = pr.nodes[E.existing[e]];
      for (auto& kv : n.volume_limits) E.en_vol_limit[(size_t)e * ND + vol_driver_id.at(kv.first)] = kv.second;
      std::set<std::string> seen;
      for (auto& v : n.volumes) {
        auto d = vol_driver_id.find(v.driver); if (d == vol_driver_id.end()) continue;
        std::string s2 = vol_pair(v); if (!seen.insert(s2).second) continue;
        E.en_vol_count[(size_t)e * ND + d->second]++;
        auto sh = vol_shared.find(s2); if (sh != vol_shared.end()) E.en_vol_set[(size_t)e * SW + sh->second / 64] |= 1ull << (sh->second % 64);
      }
    }
  }

  // ---------- existing nodes ----------
  void encode_existing() {
    if (base) {      // the name / hostname indices are the snapshot's; only the row numbering depends on which nodes left
      existing_row.assign(pr.nodes.size(), -1);
      for (size_t i = 0; i < pr.nodes.size(); ++i) if (!(*removed)[i] && base->base_existing_of[i] >= 0) { existing_row[i] = (int)E.existing.size(); E.existing.push_back((int)i); }      // (a row of the base: owned AND in state -- a node ksh_env_apply took out of state stays in the list)
      E.en_port_off.assign(1, 0);
      for (int i : E.existing) { for (auto& hp : pr.nodes[i].host_ports) E.ports.push_back(port_entry(hp.ip, hp.port, hp.proto)); E.en_port_off.push_back((uint32_t)E.ports.size()); }
      return;
    }
    node_owned.assign(pr.nodes.size(), 0);
    for (size_t i = 0; i < pr.nodes.size(); ++i) { node_by_name[pr.nodes[i].name] = &pr.nodes[i]; node_owned[i] = pr.nodes[i].owned() ? 1 : 0; if (node_in_state(i) && node_owned[i]) E.existing.push_back((int)i); if (!pr.nodes[i].volume_limits.empty()) any_volume_limits = true; }
    const uint32_t NE = (uint32_t)E.existing.size();
    E.en_port_off.assign(1, 0);
    for (uint32_t e = 0; e < NE; ++e) {
      const auto& n = pr.nodes[E.existing[e]];
      auto hl = n.labels.find(ksp::kHostname); std::string hostname = (hl == n.labels.end() || hl->second.empty()) ? n.name : hl->second;
      hostname_to_existing[hostname] = (int)e; hostname_to_node[hostname] = E.existing[e];
      // existing host-port reservations come first in ports[] (the kernel seeds its pool from them)
      for (auto& hp : n.host_ports) E.ports.push_back(port_entry(hp.ip, hp.port, hp.proto));
      E.en_port_off.push_back((uint32_t)E.ports.size());
    }
  }
  // What-if over a shared snapshot: a state node's row does not depend on which OTHER nodes leave, so the rows come from the flattening of
  // the whole snapshot; only remainingResources (limits minus the capacity of the nodes that stay, scheduler.go:244-246) is redone.
  void encode_existing_rest_from_base() {
    const Encoded& B = base->E; const uint32_t NE = (uint32_t)E.existing.size(), M = (uint32_t)E.templates.size();
    E.en.n = NE; E.en.present.resize(NE); E.en.complement.resize(NE); E.en.it_state.resize(NE);
    E.en.mask.resize((size_t)NE * K); E.en.gt.resize((size_t)NE * K); E.en.lt.resize((size_t)NE * K);
    E.en_taints.resize(NE); E.en_avail.resize((size_t)NE * R); E.en_requests.resize((size_t)NE * R); E.en_requests_present.resize(NE);
    for (uint32_t e = 0; e < NE; ++e) {
      const int b = base->base_existing_of[E.existing[e]];
      E.en.present[e] = B.en.present[b]; E.en.complement[e] = B.en.complement[b]; E.en.it_state[e] = B.en.it_state[b];
      std::copy_n(&B.en.mask[(size_t)b * K], K, &E.en.mask[(size_t)e * K]); std::copy_n(&B.en.gt[(size_t)b * K], K, &E.en.gt[(size_t)e * K]); std::copy_n(&B.en.lt[(size_t)b * K], K, &E.en.lt[(size_t)e * K]);
      E.en_taints[e] = B.en_taints[b]; E.en_requests_present[e] = B.en_requests_present[b];
      std::copy_n(&B.en_avail[(size_t)b * R], R, &E.en_avail[(size_t)e * R]); std::copy_n(&B.en_requests[(size_t)b * R], R, &E.en_requests[(size_t)e * R]);
    }
    std::vector<ksp::ResList> remaining = base->base_remaining;
    if (!env_mode) for (size_t i = 0; i < pr.nodes.size(); ++i) if ((*removed)[i] && base->base_existing_of[i] >= 0) {      // (env mode: the same nodes are in state as in the cached flattening)
      auto pl = pr.nodes[i].labels.find(ksp::kProvisionerName);
      for (uint32_t m = 0; m < M; ++m) if (E.templates[m]->has_limits && E.templates[m]->name == pl->second) for (auto& kv : remaining[m]) { auto c = pr.nodes[i].capacity.find(kv.first); if (c != pr.nodes[i].capacity.end()) kv.second += c->second; }
    }
    for (uint32_t m = 0; m < M; ++m) res_vec(remaining[m], E.tmpl_remaining, nullptr);
  }
  void adopt_base() {
    const Builder& b = *base; const Encoded& B = b.E;
    wellKnown = b.wellKnown; key_id = b.key_id; res_id = b.res_id; taint_id = b.taint_id; taints = b.taints; ip_id = b.ip_id; proto_id = b.proto_id;      // (domains: read in place)
    blocked_taint = b.blocked_taint; toleratePreferNoSchedule = b.toleratePreferNoSchedule; K = b.K; R = b.R; T = b.T; TW = b.TW;
    E.key_names = B.key_names; E.key_values = B.key_values; E.key_members = B.key_members; E.key_class = B.key_class; E.key_ints = B.key_ints; E.res_names = B.res_names; E.key_nvalues = B.key_nvalues; E.value_int = B.value_int;
    // the catalogue arrays (it_*) stay the snapshot's: Encoded::shared keeps them alive, finish() points ks_problem at them
    E.templates = B.templates; E.tmpl = B.tmpl; E.tmpl_taints = B.tmpl_taints; E.tmpl_types = B.tmpl_types; E.tmpl_daemon = B.tmpl_daemon; E.tmpl_daemon_present = B.tmpl_daemon_present;
    E.tmpl_limit_present = B.tmpl_limit_present;
  }

  void encode_existing_rest() {
    const uint32_t NE = (uint32_t)E.existing.size();
    const uint32_t M = (uint32_t)E.templates.size();
    base_existing_of.assign(pr.nodes.size(), -1); for (uint32_t e = 0; e < NE; ++e) base_existing_of[E.existing[e]] = (int)e;
    std::unordered_map<std::string, int32_t> warm_it_state;
    // remainingResources, scheduler.go:71-75,244-246
    std::vector<ksp::ResList> remaining(M);
    for (uint32_t m = 0; m < M; ++m) if (E.templates[m]->has_limits) remaining[m] = E.templates[m]->limits;
    for (uint32_t e = 0; e < NE; ++e) {
      const auto& n = pr.nodes[E.existing[e]];
      const int pb = (warm && pr.daemons.empty() && (size_t)E.existing[e] < prev->base_existing_of.size()) ? prev->base_existing_of[E.existing[e]] : -1;
      if (pb >= 0) {
        // The node was a row of the previous flattening and its labels are what they were: the row's requirement words are copied; what numbers things in order of
        // first use (the instance-type state, taints) or changes with the pods bound to the node (available) is worked out as always.
        const ReqSetsStore& B = prev->E.en; ReqSetsStore& st = E.en; const uint32_t idx = st.n++;
        st.present.push_back(B.present[pb]); st.complement.push_back(B.complement[pb]);
        st.mask.insert(st.mask.end(), B.mask.begin() + (size_t)pb * K, B.mask.begin() + (size_t)(pb + 1) * K); st.gt.insert(st.gt.end(), B.gt.begin() + (size_t)pb * K, B.gt.begin() + (size_t)(pb + 1) * K);
        st.lt.insert(st.lt.end(), B.lt.begin() + (size_t)pb * K, B.lt.begin() + (size_t)(pb + 1) * K); st.it_state.push_back(0);
        { auto il = n.labels.find(ksp::kInstanceType); const bool beta = n.labels.count("beta.kubernetes.io/instance-type") != 0;      // (labels.go:103-109: the one alias of the key)
          auto seen = (il != n.labels.end() && !beta) ? warm_it_state.find(il->second) : warm_it_state.end();
          if (seen != warm_it_state.end()) st.it_state[idx] = seen->second;      // (a value met before in this run: the state it was given then -- it_state_of numbers by first use)
          else if (il != n.labels.end() && !beta) { st.it_state[idx] = it_state_of(Requirement::New(ksp::kInstanceType, Op::In, {il->second})); warm_it_state.emplace(il->second, st.it_state[idx]); }      // (FromLabels of the one label)
          else if (il != n.labels.end() || beta) {
            StrMap one; for (auto& kv : n.labels) if (ksp::normalize_key(kv.first) == ksp::kInstanceType) one.emplace(kv.first, kv.second);
            const Requirements r1 = Requirements::FromLabels(one); auto f = r1.m.find(ksp::kInstanceType);
            if (f != r1.m.end()) { st.it_state[idx] = it_state_of(f->second); if (!beta) warm_it_state.emplace(il->second, st.it_state[idx]); }
          } }
        E.en_taints.push_back(taint_mask(n.taints) | (over_volume_limit(n) ? blocked_mask() : 0ull));
        res_vec(n.available, E.en_avail, nullptr);
        ksp::ResList dr; dr["pods"] = 0;
        ksp::ResList rem = Subtract(dr, n.daemonset_requests); for (auto& kv : rem) if (kv.second < 0) kv.second = 0;
        uint32_t pm; res_vec(rem, E.en_requests, &pm); E.en_requests_present.push_back(pm);
        auto pl = n.labels.find(ksp::kProvisionerName);
        for (uint32_t m = 0; m < M; ++m) if (E.templates[m]->has_limits && E.templates[m]->name == pl->second) remaining[m] = Subtract(remaining[m], n.capacity);
        continue;
      }
      Requirements full = Requirements::FromLabels(n.labels);
      Requirements hostless; for (auto& kv : full.m) if (kv.first != ksp::kHostname) hostless.m.emplace(kv.first, kv.second);
      push_reqs(E.en, restrict_to_known_keys(hostless), nullptr, false, false);
      E.en_taints.push_back(taint_mask(n.taints) | (over_volume_limit(n) ? blocked_mask() : 0ull));
      res_vec(n.available, E.en_avail, nullptr);
      // daemons that should still land on this node, scheduler.go:229-240 + existingnode.go:41-53
      ksp::ResList dr; int count = 0;
      for (auto& d : pr.daemons) { Pod dp = d; if (!Tolerates(n.taints, dp)) continue; if (!full.Compatible(NewPodRequirements(dp), wellKnown)) continue; dr = Merge(dr, RequestsForPod(d)); ++count; }
      dr["pods"] = (int64_t)count * 1000;
      ksp::ResList rem = Subtract(dr, n.daemonset_requests); for (auto& kv : rem) if (kv.second < 0) kv.second = 0;
      uint32_t pm; res_vec(rem, E.en_requests, &pm); E.en_requests_present.push_back(pm);
      auto pl = n.labels.find(ksp::kProvisionerName);
      for (uint32_t m = 0; m < M; ++m) if (E.templates[m]->has_limits && E.templates[m]->name == pl->second) remaining[m] = Subtract(remaining[m], n.capacity);
    }
    base_remaining = remaining;
    for (uint32_t m = 0; m < M; ++m) {
      const auto& p = *E.templates[m];
      // getDaemonOverhead, scheduler.go:250-267
      ksp::ResList dr; int count = 0;
      for (auto& d : pr.daemons) { Pod dp = d; if (!Tolerates(p.taints, dp)) continue; if (!tmpl_reqs[m].Compatible(NewPodRequirements(dp), wellKnown)) continue; dr = Merge(dr, RequestsForPod(d)); ++count; }
      dr["pods"] = (int64_t)count * 1000;
      uint32_t pm; res_vec(dr, E.tmpl_daemon, &pm); E.tmpl_daemon_present.push_back(pm);
      uint32_t lm = 0xFFFFFFFFu; if (p.has_limits) { lm = 0; for (auto& kv : p.limits) lm |= 1u << res_id.at(kv.first); }
      E.tmpl_limit_present.push_back(lm);
      res_vec(remaining[m], E.tmpl_remaining, nullptr);
    }
  }

  // ---------- topology groups ----------
  // countDomains, topology.go:231-276 (the API-server lookups are answered from the cluster snapshot)
  void count_domains(Group& g) {
    for (auto& cp : pr.cluster_pods) {
      if (!g.namespaces.count(cp.ns)) continue;
      if (!(g.selector.nil || SelectorMatches(g.selector, cp.labels))) continue;   // TopologyListOptions: nil selector lists everything
      if (batch_uids.count(cp.uid)) continue;
      const ksp::StateNode* nptr = node_named(cp.node_name); if (!nptr) continue;
      const auto& node = *nptr;
      auto lt = node.labels.find(g.key); bool ok = lt != node.labels.end(); std::string domain = ok ? lt->second : "";
      if (!ok && g.key == ksp::kHostname) { domain = node.name; ok = true; }
      if (!ok) continue;
      if (!FilterMatches(g.filter, Requirements::FromLabels(node.labels), wellKnown)) continue;
      g.counts[domain]++;
    }
  }
  static std::string group_id(int type, const std::string& key, const std::set<std::string>& nss, const ksp::Selector& sel, int32_t max_skew, const Filter& f) {
    std::string id = key + "|" + std::to_string(type) + "|"; for (auto& n : nss) id += n + ","; id += "|" + selector_identity(sel) + "|" + std::to_string(max_skew) + "|" + filter_identity(f);
    return id;
  }
  int get_group(bool inverse, int type, const std::string& key, const std::set<std::string>& nss, const ksp::Selector& sel, int32_t max_skew, const Pod& owner, bool active_now) {
    Filter f; if (type == 0) f = MakeTopologyNodeFilter(owner);
    const std::string id = group_id(type, key, nss, sel, max_skew, f);
    auto& index = inverse ? inverse_by_id : topo_by_id;
    auto it = index.find(id);
    if (it != index.end()) {
      Group& g = *groups[it->second];
      if (!g.active && !active_now && g.filter_sig != filter_content(f)) throw Unsupported("late-created topology group whose node filter depends on which pod relaxes first");
      if (g.filter_sig != filter_content(f)) shared_filter_differs = true;      // the group keeps its creator's filter (the identity hashes the filter's KEYS only): which pod came first matters
      return it->second;
    }
    auto g = std::make_unique<Group>(); g->type = type; g->key = key; g->namespaces = nss; g->selector = sel; g->max_skew = max_skew; g->filter = f; g->inverse = inverse; g->active = active_now;
    g->filter_sig = filter_content(f);
    const auto& doms = base ? base->domains : domains;
    auto d = doms.find(key); if (d != doms.end()) for (auto& v : d->second) g->counts[v] = 0;   // NewTopologyGroup, topologygroup.go:64-68
    if (!inverse) count_domains(*g);
    int idx = (int)groups.size(); groups.push_back(std::move(g)); index[id] = idx; return idx;
  }
  static std::set<std::string> ns_list(const std::string& ns, const std::vector<std::string>& l) { if (l.empty()) return {ns}; return std::set<std::string>(l.begin(), l.end()); }

  struct SpecGroups { std::vector<int> own, iown; };
  // Topology.Update for one spec (topology.go:86-117); active_now == called from NewTopology
  SpecGroups groups_of(const Pod& p, bool active_now) {
    SpecGroups sg;
    if (!p.anti_required.empty() || !p.anti_preferred.empty())
      for (auto& t : p.anti_required) { int g = get_group(true, 2, t.topology_key, ns_list(p.ns, t.namespaces), t.selector, INT32_MAX, p, true); if (std::find(sg.iown.begin(), sg.iown.end(), g) == sg.iown.end()) sg.iown.push_back(g); }
    auto add = [&](int g) { if (std::find(sg.own.begin(), sg.own.end(), g) == sg.own.end()) sg.own.push_back(g); };
    for (auto& cs : p.spread) add(get_group(false, 0, cs.key, {p.ns}, cs.selector, cs.max_skew, p, active_now));
    for (auto& t : p.affinity_required) add(get_group(false, 1, t.topology_key, ns_list(p.ns, t.namespaces), t.selector, INT32_MAX, p, active_now));
    for (auto& w : p.affinity_preferred) add(get_group(false, 1, w.term.topology_key, ns_list(p.ns, w.term.namespaces), w.term.selector, INT32_MAX, p, active_now));
    for (auto& t : p.anti_required) add(get_group(false, 2, t.topology_key, ns_list(p.ns, t.namespaces), t.selector, INT32_MAX, p, active_now));
    for (auto& w : p.anti_preferred) add(get_group(false, 2, w.term.topology_key, ns_list(p.ns, w.term.namespaces), w.term.selector, INT32_MAX, p, active_now));
    return sg;
  }

  // ---------- pods / classes ----------
  struct StageInfo { Pod spec; Requirements reqs; SpecGroups sg; };
  struct SpecInfo { std::vector<StageInfo> stages; std::vector<uint32_t> cls; };
  std::vector<SpecInfo> specs; std::unordered_map<std::string, int> spec_by_sig; std::vector<int> pod_spec;
  struct ClassRec { std::string sig; };
  std::unordered_map<std::string, uint32_t> class_by_sig;
  std::vector<std::pair<std::string, StrMap>> labelsets; std::map<std::string, int> labelset_id;   // (ns, labels)
  std::vector<int> cls_labelset; std::vector<SpecGroups> cls_groups;

  // Pods -> distinct specs (everything Solve can read of a pod except uid / creationTimestamp), in order of first occurrence.
  // Hashing and the field-by-field confirmation run on all host threads; the table is filled in pod order so that spec ids, and
  // with them the creation order of topology groups (NewTopology's Update per pod, topology.go:72-78), do not depend on threading.
  void build_uid_table(uint32_t P, bool check_unique) {
    uint64_t cap = 64; while (cap < 4ull * P) cap <<= 1;
    batch_uids.uids = &uidv; batch_uids.mask = cap - 1; batch_uids.tab.assign(cap, 0);
    for (uint32_t i = 0; i < P; ++i) {
      uint64_t j = str_hash(uidv[i]) & batch_uids.mask;
      for (;; j = (j + 1) & batch_uids.mask) { const uint32_t e = batch_uids.tab[j]; if (!e) break; if (check_unique && uidv[e - 1] == uidv[i]) throw ksp::Error("pod UIDs must be unique (queue.go:102-108 needs a total order)"); }
      batch_uids.tab[j] = i + 1;
    }
  }
  void dedupe_specs() {
    if (lite) {
      if (base && !env_mode) throw ksp::Error("a binary pod batch cannot be a what-if over a snapshot");
      podp.resize(lite->size()); uidv.resize(lite->size());
      parallel_chunks(lite->size(), [&](size_t b, size_t e, uint32_t) { for (size_t i = b; i < e; ++i) { podp[i] = &lite->specs[lite->pod_spec[i]]; uidv[i] = lite->uid(i); } });
    } else {
      if (podp.empty() && !base) { podp.resize(pr.pods.size()); const Pod* p0 = pr.pods.data(); parallel_chunks(podp.size(), [&](size_t b, size_t e, uint32_t) { for (size_t i = b; i < e; ++i) podp[i] = p0 + i; }, 16384); }
      uidv.resize(podp.size()); parallel_chunks(podp.size(), [&](size_t b, size_t e, uint32_t) { for (size_t i = b; i < e; ++i) uidv[i] = podp[i]->uid; }, 16384);
    }
    const uint32_t P = (uint32_t)podp.size(); collect_volumes(); sublap("(start)");
    if (lite) {
      // The ingest already partitioned the batch into distinct specs (records equal word for word, then field by field across blocks), in order
      // of first occurrence; two of them may still be the same spec written differently -- that costs a spec, not correctness.
      pod_spec.assign(lite->pod_spec.begin(), lite->pod_spec.end());
      if (!pr.cluster_pods.empty()) build_uid_table(P, true);
      specs.resize(lite->specs.size());
      parallel_chunks(specs.size(), [&](size_t b, size_t e, uint32_t) { for (size_t s2 = b; s2 < e; ++s2) { StageInfo st; st.spec = lite->specs[s2]; specs[s2].stages.push_back(std::move(st)); } }, 128);
      sublap("specs from the batch");
      return;
    }
    if (base && !env_mode) {
      // What-if over a snapshot: its pods ARE snapshot pods, and the snapshot's flattening already knows which of them share a spec (a partition
      // at least as fine as this what-if needs).  Local spec ids in order of first occurrence, as always.
      const Pod* p0 = pr.pods.data();
      std::vector<int32_t> local(base->specs.size(), -1); std::vector<uint32_t> first; pod_spec.assign(P, -1);
      for (uint32_t i = 0; i < P; ++i) { const int bs = base->pod_spec[podp[i] - p0]; if (local[bs] < 0) { local[bs] = (int32_t)first.size(); first.push_back(i); } pod_spec[i] = local[bs]; }
      if (!pr.cluster_pods.empty()) build_uid_table(P, false);      // countDomains / inverse anti-affinity ask which cluster pods are in the batch
      specs.resize(first.size());
      for (size_t s2 = 0; s2 < first.size(); ++s2) { StageInfo st; st.spec = *podp[first[s2]]; st.spec.uid.clear(); specs[s2].stages.push_back(std::move(st)); }
      sublap("specs from the snapshot");
      return;
    }
    if (prev && !pods_have_volumes && !prev->pods_have_volumes && !prev->spec_tab.empty() && prev->n_pods_built <= P && pr.cluster_pods.empty() == prev->batch_uids.tab.empty()) {
      // The pods the previous flattening saw keep their specs (first occurrences in pod order: appending pods cannot renumber them); only the new ones are hashed,
      // looked up among the specs' first pods and confirmed field by field.
      const uint32_t P0 = (uint32_t)prev->n_pods_built;
      pod_spec.assign(prev->pod_spec.begin(), prev->pod_spec.begin() + P0); pod_spec.resize(P, -1);
      spec_first = prev->spec_first; spec_hs = prev->spec_hs; spec_tab = prev->spec_tab;
      uint64_t cap = spec_tab.size();
      if (cap < 4ull * P) {      // (the table's size is a function of P: a fresh run would have sized it so; its content is rebuilt in spec order)
        while (cap < 4ull * P) cap <<= 1;
        spec_tab.assign(cap, -1);
        for (size_t s2 = 0; s2 < spec_first.size(); ++s2) { uint64_t j = spec_hs[s2].a & (cap - 1); while (spec_tab[j] >= 0) j = (j + 1) & (cap - 1); spec_tab[j] = (int32_t)s2; }
      }
      for (uint32_t i = P0; i < P; ++i) {
        const Hash128 h = spec_hash(*podp[i]);
        uint64_t j = h.a & (cap - 1); int found = -1;
        for (;; j = (j + 1) & (cap - 1)) { const int32_t sidx = spec_tab[j]; if (sidx < 0) break; const Hash128& o = spec_hs[sidx]; if (o.a == h.a && o.b == h.b) { found = sidx; break; } }
        if (found >= 0 && !same_pod(*podp[spec_first[found]], *podp[i])) { found = -1; for (size_t s2 = 0; s2 < spec_first.size() && found < 0; ++s2) if (same_pod(*podp[spec_first[s2]], *podp[i])) found = (int)s2; if (found < 0) { found = (int)spec_first.size(); spec_first.push_back(i); spec_hs.push_back(h); } }      // (a collision: the full run's own way out)
        else if (found < 0) { found = (int)spec_first.size(); spec_tab[j] = found; spec_first.push_back(i); spec_hs.push_back(h); }
        pod_spec[i] = found;
      }
      if (!pr.cluster_pods.empty()) build_uid_table(P, true);
      const std::vector<uint32_t>& first = spec_first;
      sublap("specs from the flattening before"); specs.resize(first.size());
      parallel_chunks(first.size(), [&](size_t b, size_t e, uint32_t) { for (size_t s2 = b; s2 < e; ++s2) { StageInfo st; st.spec = *podp[first[s2]]; st.spec.uid.clear(); specs[s2].stages.push_back(std::move(st)); } }, 128);
      return;
    }
    std::vector<Hash128> hs(P);
    parallel_chunks(P, [&](size_t b, size_t e, uint32_t) { for (size_t i = b; i < e; ++i) {
      if (pods_have_volumes) { const std::vector<uint32_t> ve = vol_entries(*podp[i]); hs[i] = spec_hash(*podp[i], &ve); } else hs[i] = spec_hash(*podp[i]);
      } });
    sublap("hash"); uint64_t cap = 64; while (cap < 4ull * P) cap <<= 1;
    if (!pr.cluster_pods.empty()) build_uid_table(P, true);      // countDomains / inverse anti-affinity ask which cluster pods are in the batch; without cluster pods nobody asks, and the
                                                                 // uniqueness of the UIDs is checked on the sorted queue instead (encode_pods)
    sublap("uid table"); pod_spec.assign(P, -1);
    std::vector<int32_t> tab(cap, -1); std::vector<uint32_t> first;      // table of spec ids; first[s] = first pod with spec s
    for (uint32_t i = 0; i < P; ++i) {
      uint64_t j = hs[i].a & (cap - 1);
      for (;; j = (j + 1) & (cap - 1)) { const int32_t sidx = tab[j]; if (sidx < 0) break; const Hash128& o = hs[first[sidx]]; if (o.a == hs[i].a && o.b == hs[i].b) { pod_spec[i] = sidx; break; } }
      if (pod_spec[i] < 0) { tab[j] = (int32_t)first.size(); pod_spec[i] = (int32_t)first.size(); first.push_back(i); }
    }
    sublap("spec table");
    // Two pods share a spec when their 128-bit spec hashes agree (two independent 64-bit lanes over every field: a false merge
    // needs a 2^-128 event for random inputs, but pod specs are tenant-supplied).  Every merge is therefore confirmed field by field (on the
    // worker pool; KSH_NO_CONFIRM_SPECS=1 skips it for A/B timing only); a pod
    // that merely collided would get a spec of its own.
    // The Solve of a large batch does not wait for the confirmation: it runs on a thread of its own (second worker pool) beside what follows -- the specs' copies,
    // the universes, the signature -- and is joined before the classing (encode_pods); should it ever find a merge that does not hold, the flattening starts over
    // with the confirmation in line (ConfirmFailed, encode()).
    if (!keep_warm_state && !tl_sync_confirm && P >= 8192 && host_threads() >= 4 && !getenv("KSH_NO_CONFIRM_SPECS") && !getenv("KSH_SYNC_CONFIRM")) {
      confirm_first = first;
      try {
        confirmer = std::async(std::launch::async, [this, P] {
          tl_pool = 1; std::atomic<bool> any{false};
          parallel_chunks(P, [&](size_t b, size_t e, uint32_t) { for (size_t i = b; i < e; ++i) { const uint32_t f = confirm_first[pod_spec[i]]; if (f != i && !same_pod(*podp[f], *podp[i])) any = true; } });
          return any.load() || getenv("KSH_TEST_CONFIRM_FAILS") != nullptr; });
      } catch (const std::system_error&) {}
    }
    std::vector<uint8_t> bad(P, 0);
    if (!confirmer.valid() && !getenv("KSH_NO_CONFIRM_SPECS")) parallel_chunks(P, [&](size_t b, size_t e, uint32_t) { for (size_t i = b; i < e; ++i) { const uint32_t f = first[pod_spec[i]]; if (f != i && !same_pod(*podp[f], *podp[i])) bad[i] = 1; } });
    for (uint32_t i = 0; i < P; ++i) if (bad[i]) {
      int found = -1; for (size_t s2 = 0; s2 < first.size() && found < 0; ++s2) if (same_pod(*podp[first[s2]], *podp[i])) found = (int)s2;
      if (found < 0) { found = (int)first.size(); first.push_back(i); }
      pod_spec[i] = found;
    }
    sublap("confirm"); specs.resize(first.size());
    if (keep_warm_state) {      // (what the next flattening of this snapshot starts from; a spec found by the collision path has no slot in the table -- such a run is not continued)
      bool clean = true; for (uint32_t i = 0; i < P && clean; ++i) if (bad[i]) clean = false;
      if (clean) { spec_first = first; spec_tab = tab; spec_hs.resize(first.size()); for (size_t s2 = 0; s2 < first.size(); ++s2) spec_hs[s2] = hs[first[s2]]; }
    }
    parallel_chunks(first.size(), [&](size_t b, size_t e, uint32_t) { for (size_t s2 = b; s2 < e; ++s2) { StageInfo st; st.spec = *podp[first[s2]]; st.spec.uid.clear(); specs[s2].stages.push_back(std::move(st)); } }, 128);      // (a spec is a deep copy of a pod: strings, vectors)
  }

  void encode_pods() {
    const uint32_t P = (uint32_t)podp.size();
    // NewQueue's sort starts first and runs beside everything below (queue_sort): it reads the pods' uids / timestamps and, per spec, cpu and memory -- taken here,
    // before the chains below grow the specs' stage vectors
    if (confirmer.valid() && confirmer.get()) throw ConfirmFailed();      // (the batch's partition into specs, confirmed beside the work since dedupe_specs)
    std::future<void> sorter;
    if (!(base && !env_mode)) {
      spec_cm.resize(specs.size());
      parallel_chunks(specs.size(), [&](size_t b, size_t e, uint32_t) { for (size_t s2 = b; s2 < e; ++s2) { const ksp::ResList rq = RequestsForPod(specs[s2].stages[0].spec);
        auto c = rq.find("cpu"), m = rq.find("memory"); spec_cm[s2] = {c == rq.end() ? 0 : c->second, m == rq.end() ? 0 : m->second}; } }, 256);
      if (P >= 8192 && host_threads() >= 4 && !getenv("KSH_SYNC_QUEUE_SORT")) { try { sorter = std::async(std::launch::async, [this] { tl_pool = 1; queue_sort(); }); } catch (const std::system_error&) {} }
    }
    // updateInverseAffinities, topology.go:181-199 (cluster pods with required anti-affinity, not in the batch)
    for (auto& cp : pr.cluster_pods) {
      if (cp.anti_required.empty() || batch_uids.count(cp.uid)) continue;
      const ksp::StateNode* nptr = node_named(cp.node_name); if (!nptr) continue;
      Pod dummy; dummy.ns = cp.ns;
      for (auto& t : cp.anti_required) {
        int gi = get_group(true, 2, t.topology_key, ns_list(cp.ns, t.namespaces), t.selector, INT32_MAX, dummy, true);
        auto lt = nptr->labels.find(groups[gi]->key); if (lt != nptr->labels.end()) groups[gi]->counts[lt->second]++;
      }
    }
    sublap("inverse affinities");
    // The pure halves of both passes -- NewPodRequirements of every stage, Preferences.Relax down each spec's chain -- run on the worker pool;
    // what registers topology groups (groups_of) then runs serially in the reference's order.
    parallel_chunks(specs.size(), [&](size_t b, size_t e, uint32_t) { for (size_t s2 = b; s2 < e; ++s2) {
      SpecInfo& si = specs[s2]; si.stages[0].reqs = NewPodRequirements(si.stages[0].spec);
      for (;;) {
        Pod next = si.stages.back().spec;
        if (!Relax(next, toleratePreferNoSchedule)) break;
        StageInfo st; st.spec = std::move(next); st.reqs = NewPodRequirements(st.spec);
        si.stages.push_back(std::move(st));
        if (si.stages.size() > 64) throw Unsupported("more than 64 relaxation stages");
      } } }, 64);
    sublap("requirements + chains (pool)");
    // pass A: NewTopology's Update(pod) over the distinct stage-0 specs in order of first occurrence
    for (auto& si : specs) si.stages[0].sg = groups_of(si.stages[0].spec, true);
    sublap("pass A");
    // pass B: relaxation chains (Topology.Update after each relaxation)
    for (auto& si : specs) for (size_t k = 1; k < si.stages.size(); ++k) si.stages[k].sg = groups_of(si.stages[k].spec, false);
    sublap("pass B");
    // classes
    E.cls_hn_off.assign(1, 0); E.cls_port_off.assign(1, (uint32_t)E.ports.size()); E.cls_vol_off.assign(1, 0);
    { std::vector<size_t> soff(specs.size() + 1, 0); for (size_t s2 = 0; s2 < specs.size(); ++s2) soff[s2 + 1] = soff[s2] + specs[s2].stages.size();
      std::vector<ClassPre> pre(soff.back());
      parallel_chunks(specs.size(), [&](size_t b, size_t e, uint32_t) { for (size_t s2 = b; s2 < e; ++s2) for (size_t k = 0; k < specs[s2].stages.size(); ++k) class_pre(specs[s2].stages[k], pre[soff[s2] + k]); }, 64);
      sublap("class signatures (pool)");
      for (size_t s2 = 0; s2 < specs.size(); ++s2) for (size_t k = 0; k < specs[s2].stages.size(); ++k) specs[s2].cls.push_back(class_of(specs[s2].stages[k], pre[soff[s2] + k])); }
    // pods -> stage chains, queue order
    sublap("classes"); E.pod_stage_off.resize((size_t)P + 1); E.pod_stage_off[0] = 0;
    for (uint32_t i = 0; i < P; ++i) E.pod_stage_off[i + 1] = E.pod_stage_off[i] + (uint32_t)specs[pod_spec[i]].cls.size();
    E.stage_cls.resize(E.pod_stage_off[P]);
    parallel_chunks(P, [&](size_t b, size_t e, uint32_t) { for (size_t i = b; i < e; ++i) { const auto& cl = specs[pod_spec[i]].cls; std::copy(cl.begin(), cl.end(), E.stage_cls.begin() + E.pod_stage_off[i]); } });
    // NewQueue: byCPUAndMemoryDescending, queue.go:74-110
    if (base && !env_mode) {      // the order is a total order on pods: a what-if's queue is its pods in the snapshot's order
      const Pod* p0 = pr.pods.data(); std::vector<std::pair<uint32_t, uint32_t>> k(P);
      for (uint32_t i = 0; i < P; ++i) k[i] = {base->pod_rank[podp[i] - p0], i};
      std::sort(k.begin(), k.end());
      E.queue.resize(P); for (uint32_t i = 0; i < P; ++i) E.queue[i] = k[i].second; sublap("queue from the snapshot's order");
      return;
    }
    // NewQueue's sort ran beside the classing (queue_sort, started above): its result, or what it threw
    sublap("chains");
    if (sorter.valid()) sorter.get(); else queue_sort();
    sublap("queue sort (joined)");
  }

  // NewQueue: byCPUAndMemoryDescending, queue.go:74-110.  cpu and memory of a pod are its spec's RequestsForPods (spec_cm, made before anything else touches the specs);
  // timestamps and uids are the pods' own: nothing here reads what the classing writes, so it runs on a thread of its own (second worker pool) while the classes are made.
  std::vector<std::pair<int64_t, int64_t>> spec_cm; std::vector<uint32_t> confirm_first; std::future<bool> confirmer;      // (the future last: destroyed -- waited for -- first)
  void queue_sort() {
    const uint32_t P = (uint32_t)podp.size();
    // The order is total (UIDs are unique), so any correct sort gives the reference's queue: chunks are sorted on the host threads
    // and merged pairwise.  Keys are gathered first so a comparison touches one 32-byte record per side and the uid only on ties.
    struct QKey { int64_t cpu, mem, ts; uint64_t u0, u1; uint32_t pod, ulen; };     // u0,u1: the uid's first 16 bytes, big-endian (byte-wise string order)
    std::vector<QKey> keys(P);
    auto be64 = [](std::string_view s2, size_t off) { uint64_t v = 0; for (size_t j = 0; j < 8; ++j) v = (v << 8) | (off + j < s2.size() ? (unsigned char)s2[off + j] : 0u); return v; };
    parallel_chunks(P, [&](size_t b, size_t e, uint32_t) { for (size_t i = b; i < e; ++i) { const auto& cm0 = spec_cm[pod_spec[i]]; const std::string_view u = uidv[i];
      keys[i] = QKey{cm0.first, cm0.second, ts_of(i), be64(u, 0), be64(u, 8), (uint32_t)i, (uint32_t)u.size()}; } });
    auto less = [&](const QKey& a, const QKey& b) {
      if (a.cpu != b.cpu) return a.cpu > b.cpu;
      if (a.mem != b.mem) return a.mem > b.mem;
      if (a.ts != b.ts) return a.ts < b.ts;
      if (a.u0 != b.u0) return a.u0 < b.u0;
      if (a.u1 != b.u1) return a.u1 < b.u1;
      if (a.ulen <= 16 && b.ulen <= 16) return a.ulen < b.ulen;          // equal 16-byte prefixes incl. zero padding: the shorter one is a prefix (NUL bytes inside a uid fall through to the full compare)
      return uidv[a.pod] < uidv[b.pod];
    };
    // cpu and memory are a function of the spec, so the queue falls into a few (cpu, memory) buckets -- 30 for BASELINE configs[2] -- whose order is
    // known at once; inside a bucket only (timestamp, uid) decide.  Buckets are counted, filled and sorted independently on the worker threads; a
    // batch that is mostly ONE bucket takes the chunked merge sort below instead.
    bool bucketed = false;
    if (P >= 8192) {
      const std::vector<std::pair<int64_t, int64_t>>& cm = spec_cm;
      std::vector<std::pair<int64_t, int64_t>> dist(cm); std::sort(dist.begin(), dist.end(), [](auto& a, auto& b) { return a.first != b.first ? a.first > b.first : a.second > b.second; });
      dist.erase(std::unique(dist.begin(), dist.end()), dist.end());
      const size_t NB = dist.size();
      if (NB >= 2 && NB <= 4096) {
        std::vector<uint32_t> bucket_of(specs.size());
        for (size_t s2 = 0; s2 < specs.size(); ++s2) bucket_of[s2] = (uint32_t)(std::lower_bound(dist.begin(), dist.end(), cm[s2], [](auto& a, auto& b) { return a.first != b.first ? a.first > b.first : a.second > b.second; }) - dist.begin());
        std::vector<size_t> off(NB + 1, 0); for (uint32_t i = 0; i < P; ++i) ++off[bucket_of[pod_spec[i]] + 1];
        size_t biggest = 0; for (size_t b = 0; b < NB; ++b) { biggest = std::max(biggest, off[b + 1]); off[b + 1] += off[b]; }
        if (biggest <= P / 3) {
          std::vector<QKey> sorted(P); { std::vector<size_t> at(off.begin(), off.end() - 1); for (uint32_t i = 0; i < P; ++i) sorted[at[bucket_of[pod_spec[i]]]++] = keys[i]; }
          std::atomic<size_t> next{0};
          auto lessb = [&](const QKey& a, const QKey& b) {
            if (a.ts != b.ts) return a.ts < b.ts;
            if (a.u0 != b.u0) return a.u0 < b.u0;
            if (a.u1 != b.u1) return a.u1 < b.u1;
            if (a.ulen <= 16 && b.ulen <= 16) return a.ulen < b.ulen;
            return uidv[a.pod] < uidv[b.pod];
          };
          run_threads((uint32_t)std::min<size_t>(host_threads(), NB), [&](uint32_t) { for (;;) { const size_t b = next.fetch_add(1); if (b >= NB) return; std::sort(sorted.begin() + off[b], sorted.begin() + off[b + 1], lessb); } });
          keys.swap(sorted); bucketed = true;
        }
      }
    }
    if (!bucketed) {
      uint32_t nt = 1; while (nt * 2 <= host_threads() && (size_t)nt * 2 * 4096 <= P) nt *= 2;       // power of two: pairwise merge rounds
      std::vector<size_t> cut(nt + 1); for (uint32_t t = 0; t <= nt; ++t) cut[t] = (size_t)P * t / nt;
      auto run = [&](uint32_t n, auto&& body) { run_threads(n, body); };
      run(nt, [&](uint32_t t) { std::sort(keys.begin() + cut[t], keys.begin() + cut[t + 1], less); });
      std::vector<QKey> tmp(nt > 1 ? P : 0);
      for (uint32_t w = 1; w < nt; w *= 2) {
        run(nt / (2 * w), [&](uint32_t t) { const size_t a = cut[2 * w * t], m = cut[2 * w * t + w], z = cut[2 * w * t + 2 * w]; std::merge(keys.begin() + a, keys.begin() + m, keys.begin() + m, keys.begin() + z, tmp.begin() + a, less); });
        keys.swap(tmp);
      }
    }
    // equal UIDs end up next to each other (everything before the UID in the order is a function of the spec and the timestamp)
    { std::atomic<bool> dup{false};
      parallel_chunks(P, [&](size_t b, size_t e, uint32_t) { for (size_t i = std::max<size_t>(b, 1); i < e; ++i) if (keys[i].u0 == keys[i - 1].u0 && keys[i].u1 == keys[i - 1].u1 && keys[i].ulen == keys[i - 1].ulen && keys[i].ts == keys[i - 1].ts &&
                                                                                                                 keys[i].cpu == keys[i - 1].cpu && keys[i].mem == keys[i - 1].mem && uidv[keys[i].pod] == uidv[keys[i - 1].pod]) dup = true; });
      if (dup) throw ksp::Error("pod UIDs must be unique (queue.go:102-108 needs a total order)"); }
    E.queue.resize(P); for (uint32_t i = 0; i < P; ++i) E.queue[i] = keys[i].pod;
    if (keep_warm_state || !env_mode) { pod_rank.resize(P); for (uint32_t i = 0; i < P; ++i) pod_rank[E.queue[i]] = i; }      // (read by the what-ifs over a snapshot: not by a Solve over a cached environment)
  }

  // class_of in two halves: the PURE one -- the signature of everything Node.Add reads of the stage but the ids the serial half hands out (label set, topology groups) --
  // runs on the worker pool for every stage at once (strings, quantities, tolerations: what class_of spent its time on); the serial half interns.
  struct ClassPre { std::string ls, sig; ksp::ResList req; uint64_t tol = 0; std::vector<uint32_t> ve; std::vector<uint64_t> pe; };
  void class_pre(const StageInfo& st, ClassPre& o) const {
    const Pod& p = st.spec;
    // label set (namespace + labels decide which selectors select the pod)
    o.ls = p.ns; o.ls += '\3'; sig_map(o.ls, p.labels);
    // signature of everything Node.Add reads
    std::string& sig = o.sig; sig.reserve(256);
    for (auto& kv : st.reqs.m) { sig += kv.first; sig += '\1'; sig += kv.second.identity(); sig += '\2'; } sig += '\4';
    o.req = RequestsForPod(p); sig_res(sig, o.req);
    uint64_t tol = 0; for (size_t i = 0; i < taints.size(); ++i) { if ((int)i == blocked_taint) continue; bool ok = false; for (auto& t : p.tolerations) ok = ok || ToleratesTaint(t, taints[i]); if (ok) tol |= 1ull << i; }
    o.tol = tol;
    sig += std::to_string(tol); sig += '\4';
    if (pods_have_volumes) o.ve = vol_entries(p);
    for (auto e : o.ve) { sig += std::to_string(e); sig += ','; } sig += '\4';
  }      // (host ports: port_entry hands out ids in order of first use -- the serial half's)
  uint32_t class_of(const StageInfo& st, ClassPre& pre) {
    const Pod& p = st.spec;
    int lsid; auto li = labelset_id.find(pre.ls); if (li == labelset_id.end()) { lsid = (int)labelsets.size(); labelsets.emplace_back(p.ns, p.labels); labelset_id[pre.ls] = lsid; } else lsid = li->second;
    std::string& sig = pre.sig; const ksp::ResList& req = pre.req; const uint64_t tol = pre.tol; const std::vector<uint32_t>& ve = pre.ve; std::vector<uint64_t>& pe = pre.pe;
    for (auto& c : p.containers) for (auto& hp : c.ports) if (hp.port != 0) pe.push_back(port_entry(hp.ip, hp.port, hp.proto));
    for (auto e : pe) { sig += std::to_string(e); sig += ','; } sig += '\4';
    sig += std::to_string(lsid); sig += '\4';
    for (int g : st.sg.own) { sig += std::to_string(g); sig += ','; } sig += '\4';
    for (int g : st.sg.iown) { sig += std::to_string(g); sig += ','; } sig += '\4';
    auto it = class_by_sig.find(sig); if (it != class_by_sig.end()) return it->second;
    const Requirement* hn = nullptr;
    uint32_t c = push_reqs(E.cls, st.reqs, &hn, true, true);
    uint8_t mode = 0;
    if (hn) {
      if (hn->greaterThan || hn->lessThan) throw Unsupported("Gt/Lt on kubernetes.io/hostname");
      mode = hn->complement ? 2 : 1;
      for (auto& v : hn->values) { const int e = existing_of_hostname(v); if (e >= 0) E.hn_list.push_back((uint32_t)e); }
    }
    E.cls_hn_mode.push_back(mode); E.cls_hn_off.push_back((uint32_t)E.hn_list.size());
    uint32_t pm; res_vec(req, E.cls_requests, &pm); E.cls_requests_present.push_back(pm);
    E.cls_tolerated.push_back(tol);
    for (auto e : pe) E.ports.push_back(e); E.cls_port_off.push_back((uint32_t)E.ports.size());
    for (auto e : ve) E.vol_list.push_back(e); E.cls_vol_off.push_back((uint32_t)E.vol_list.size());
    cls_labelset.push_back(lsid); cls_groups.push_back(st.sg);
    if (mode != 0) for (int g : st.sg.own) if (groups[g]->key == ksp::kHostname && groups[g]->type == 1) throw Unsupported("hostname pod-affinity combined with a hostname node selector");
    class_by_sig.emplace(std::move(sig), c);
    return c;
  }

  // the taint universe must be complete before classes compute `tolerated`
  void collect_taints() {
    for (auto& p : pr.provisioners) taint_mask(p.taints);
    for (size_t i = 0; i < pr.nodes.size(); ++i) if (node_in_state(i) && pr.nodes[i].owned()) { taint_mask(pr.nodes[i].taints); if (over_volume_limit(pr.nodes[i])) blocked_mask(); }
  }

  // ---------- group tables + per-class membership lists ----------
  void encode_groups() {
    // order: topologies first, then inverse (ks_problem.n_topologies)
    std::vector<int> order, remap(groups.size(), -1);
    for (size_t g = 0; g < groups.size(); ++g) if (!groups[g]->inverse) order.push_back((int)g);
    const uint32_t ntopo = (uint32_t)order.size();
    for (size_t g = 0; g < groups.size(); ++g) if (groups[g]->inverse) order.push_back((int)g);
    for (size_t i = 0; i < order.size(); ++i) remap[order[i]] = (int)i;
    group_order = order; group_remap = remap;
    const uint32_t G = (uint32_t)order.size(); const uint32_t NE = (uint32_t)E.existing.size();
    E.grp_count.assign((size_t)G * 64, -1); E.grp_filter_off.assign(1, 0);
    uint32_t GH = 0;
    for (uint32_t gi = 0; gi < G; ++gi) {
      Group& g = *groups[order[gi]];
      E.grp_type.push_back((uint8_t)g.type); E.grp_max_skew.push_back(g.max_skew); E.grp_active.push_back(g.active ? 1 : 0);
      if (!g.filter.always) for (auto& t : g.filter.terms) {
        for (auto& kv : t.m) if (kv.first == ksp::kHostname) throw Unsupported("topology node filter on hostname");
        push_reqs(E.flt, t, nullptr, false, true);
      }
      E.grp_filter_off.push_back(E.flt.n);
      if (g.key == ksp::kHostname) {
        E.grp_key.push_back(KS_KEY_HOSTNAME); E.grp_hslot.push_back((int32_t)GH); ++GH;
        int32_t extra = 0; std::vector<int32_t> row(NE, g.active ? 0 : -1);   // NewExistingNode registers its hostname in every group that exists (existingnode.go:73)
        for (auto& kv : g.counts) { const int e = existing_of_hostname(kv.first); if (e >= 0) row[e] = kv.second; else if (kv.second > 0) ++extra; }
        E.grph_count.insert(E.grph_count.end(), row.begin(), row.end()); E.grph_extra_pos.push_back(extra);
      } else {
        int k = key_id.at(g.key); E.grp_key.push_back(k); E.grp_hslot.push_back(-1);
        for (auto& kv : g.counts) { int v = value_id(k, kv.first); if (v < 0) { key_missing(g.key, kv.first); } E.grp_count[(size_t)gi * 64 + v] = kv.second; }
      }
    }
    E.prob.G = G; E.prob.GH = GH; E.prob.n_topologies = ntopo;
    // which groups select each label set (TopologyGroup.selects, topologygroup.go:246-252)
    std::vector<std::vector<uint32_t>> sel(labelsets.size()), isel(labelsets.size());
    for (size_t l = 0; l < labelsets.size(); ++l) for (uint32_t gi = 0; gi < G; ++gi) {
      const Group& g = *groups[order[gi]];
      if (g.namespaces.count(labelsets[l].first) && SelectorMatches(g.selector, labelsets[l].second)) (g.inverse ? isel[l] : sel[l]).push_back(gi);
    }
    const uint32_t C = E.cls.n;
    E.cls_own_off.assign(1, 0); E.cls_sel_off.assign(1, 0); E.cls_isel_off.assign(1, 0); E.cls_iown_off.assign(1, 0);
    for (uint32_t c = 0; c < C; ++c) {
      const int l = cls_labelset[c];
      for (int g : cls_groups[c].own) { uint32_t gi = (uint32_t)remap[g]; bool self = std::find(sel[l].begin(), sel[l].end(), gi) != sel[l].end(); E.own_list.push_back(gi | (self ? 0x80000000u : 0)); }
      if (cls_groups[c].own.size() + isel[l].size() > 24) throw Unsupported("a pod is constrained by more than 24 topology groups");
      E.cls_own_off.push_back((uint32_t)E.own_list.size());
      for (uint32_t gi : sel[l]) E.sel_list.push_back(gi); E.cls_sel_off.push_back((uint32_t)E.sel_list.size());
      for (uint32_t gi : isel[l]) E.isel_list.push_back(gi); E.cls_isel_off.push_back((uint32_t)E.isel_list.size());
      for (int g : cls_groups[c].iown) E.iown_list.push_back((uint32_t)remap[g]); E.cls_iown_off.push_back((uint32_t)E.iown_list.size());
      // touched-key budget of the kernel (KS_MAX_TOUCH = 12, KS_MAX_HOST = 3)
      std::set<int> touched; for (uint32_t k = 0; k < K; ++k) if ((E.cls.present[c] >> k) & 1u) touched.insert((int)k);
      for (int g : cls_groups[c].own) if (groups[g]->key != ksp::kHostname) touched.insert(key_id.at(groups[g]->key));
      for (uint32_t gi : isel[l]) if (E.grp_key[gi] >= 0) touched.insert(E.grp_key[gi]);
      if (touched.size() > 12) throw Unsupported("a pod touches more than 12 label keys (own requirements + topology keys)");
      size_t nh = 0; for (int g : cls_groups[c].own) if (groups[g]->key == ksp::kHostname) ++nh;
      for (uint32_t gi : isel[l]) if (E.grp_key[gi] == KS_KEY_HOSTNAME) ++nh;
      if (nh > 3) throw Unsupported("a pod is constrained by more than 3 hostname-keyed topology groups");
    }
  }
  [[noreturn]] void key_missing(const std::string& k, const std::string& v) { throw std::logic_error("topology domain " + v + " of key " + k + " missing from the universe"); }

  // ---------- instance-type-key lattice ----------
  void encode_it_states() {
    if (base && !lattice_adopted) {      // every class / filter / node state of the what-if is one the snapshot already has: its tables are used in place
      E.shared_lattice = true; E.prob.S = base->E.prob.S; E.prob.SC = base->E.prob.SC;
      return;
    }
    const std::vector<Requirements>& it_requirements = base ? *base->it_requirements_p : *this->it_requirements_p;
    // Node states are closed under intersection with every pod-side requirement (a node only ever narrows its
    // instance-type requirement by a class's, node.go:79 / existingnode.go:102); it_state_of appends while we iterate.
    if (it_reqs.empty()) it_reqs.push_back(Requirement());   // state 0 placeholder ("absent")
    if (it_cols.empty()) it_cols.push_back(Requirement());
    if (keep_warm_state) { pre_it_state_id = it_state_id; pre_it_col_id = it_col_id; }
    if (warm && !prev->pre_it_state_id.empty() && it_state_id == prev->pre_it_state_id && it_col_id == prev->pre_it_col_id && it_reqs.size() == prev->pre_it_state_id.size() + 1) {
      // the same node-side states and pod-side columns in the same order, over the same catalogue: the closure and its tables are the previous flattening's
      const Encoded& B = prev->E;
      it_reqs = prev->it_reqs; it_state_id = prev->it_state_id;
      E.its_inter = B.its_inter; E.its_fail = B.its_fail; E.its_nidne = B.its_nidne; E.its_types = B.its_types; E.it_states = B.it_states; E.prob.S = B.prob.S; E.prob.SC = B.prob.SC;
      return;
    }
    const uint32_t SC = (uint32_t)it_cols.size();
    for (size_t a = 1; a < it_reqs.size(); ++a) for (uint32_t b = 1; b < SC; ++b) it_state_of(it_cols[b].Intersection(it_reqs[a]));
    for (uint32_t b = 1; b < SC; ++b) it_state_of(it_cols[b]);      // absent ∩ b = b
    for (size_t a = 1; a < it_reqs.size(); ++a) for (uint32_t b = 1; b < SC; ++b) it_state_of(it_cols[b].Intersection(it_reqs[a]));
    const uint32_t S = (uint32_t)it_reqs.size();
    E.its_inter.assign((size_t)S * SC, 0); E.its_fail.assign((size_t)S * SC, 0); E.its_nidne.assign(S, 0); E.its_types.assign((size_t)S * TW, 0);
    for (uint32_t a = 0; a < S; ++a) for (uint32_t b = 0; b < SC; ++b) {
      uint32_t r; bool fail = false;
      if (b == 0) r = a; else if (a == 0) r = (uint32_t)it_state_id.at(it_cols[b].identity());
      else { Requirement x = it_cols[b].Intersection(it_reqs[a]); r = (uint32_t)it_state_id.at(x.identity()); fail = x.Len() == 0 && !(it_cols[b].IsNotInOrDoesNotExist() && it_reqs[a].IsNotInOrDoesNotExist()); }
      E.its_inter[(size_t)a * SC + b] = (uint16_t)r; E.its_fail[(size_t)a * SC + b] = fail ? 1 : 0;
    }
    for (uint32_t s = 1; s < S; ++s) E.its_nidne[s] = it_reqs[s].IsNotInOrDoesNotExist() ? 1 : 0;
    // its_types[s]: types whose own instance-type requirement intersects state s.  Types with the usual
    // `instance-type In [name]` are resolved through a name index (|values| work per state, not T).
    std::map<std::string, std::vector<uint32_t>> by_name; std::vector<uint32_t> general; std::vector<uint64_t> simple(TW, 0), unconstrained(TW, 0);
    for (uint32_t t = 0; t < T; ++t) {
      auto it = it_requirements[t].m.find(ksp::kInstanceType);
      if (it == it_requirements[t].m.end()) { unconstrained[t / 64] |= 1ull << (t % 64); continue; }
      const Requirement& a = it->second;
      if (!a.complement && a.values.size() == 1 && !a.greaterThan && !a.lessThan) { by_name[*a.values.begin()].push_back(t); simple[t / 64] |= 1ull << (t % 64); }
      else general.push_back(t);
    }
    for (uint32_t s = 0; s < S; ++s) {
      uint64_t* row = &E.its_types[(size_t)s * TW];
      for (uint32_t w = 0; w < TW; ++w) row[w] = unconstrained[w];
      const Requirement& q = it_reqs[s];
      if (s == 0) { for (uint32_t w = 0; w < TW; ++w) row[w] |= simple[w]; for (uint32_t t : general) row[t / 64] |= 1ull << (t % 64); continue; }
      if (!q.greaterThan && !q.lessThan) {
        if (q.complement) { for (uint32_t w = 0; w < TW; ++w) row[w] |= simple[w]; for (auto& v : q.values) { auto f = by_name.find(v); if (f != by_name.end()) for (uint32_t t : f->second) row[t / 64] &= ~(1ull << (t % 64)); } }
        else for (auto& v : q.values) { auto f = by_name.find(v); if (f != by_name.end()) for (uint32_t t : f->second) row[t / 64] |= 1ull << (t % 64); }
      } else for (auto& kv : by_name) if (q.Has(kv.first)) for (uint32_t t : kv.second) row[t / 64] |= 1ull << (t % 64);
      for (uint32_t t : general) {
        const Requirement& a = it_requirements[t].m.find(ksp::kInstanceType)->second; Requirement x = a.Intersection(q);
        if (!(x.Len() == 0 && !(q.IsNotInOrDoesNotExist() && a.IsNotInOrDoesNotExist()))) row[t / 64] |= 1ull << (t % 64);
      }
    }
    E.it_states = it_reqs; E.prob.S = S; E.prob.SC = SC;
  }

  void finish() {
    ks_problem& p = E.prob;
    p.P = (uint32_t)podp.size(); p.C = E.cls.n; p.T = T; p.M = (uint32_t)E.templates.size(); p.E = (uint32_t)E.existing.size(); p.K = K; p.R = R;
    p.max_new_nodes = p.P ? p.P : 1; p.flags = flags | ((pr.simulation_mode || (base && !env_mode)) ? KS_FLAG_SIMULATION : 0);
    p.wellknown_mask = 0; for (uint32_t k = 0; k < K; ++k) if (wellKnown.count(E.key_names[k])) p.wellknown_mask |= 1u << k;
    p.key_nvalues = E.key_nvalues.data(); p.value_int = E.value_int.data(); p.key_zone = key_id.at(ksp::kZone); p.key_ct = key_id.at(ksp::kCapacityType); p.n_ct = E.key_nvalues[p.key_ct];
    const Encoded& CAT = E.catalogue(); const Encoded& LAT = E.lattice();
    p.it_present = CAT.it_present.data(); p.it_complement = CAT.it_complement.data(); p.it_mask = CAT.it_mask.data(); p.it_alloc = CAT.it_alloc.data(); p.it_cap = CAT.it_cap.data(); p.it_offer = CAT.it_offer.data(); p.it_price = CAT.it_price.data(); p.it_price_lo = CAT.it_price_lo.data(); p.ct_spot = value_id(p.key_ct, "spot"); p.ct_ondemand = value_id(p.key_ct, "on-demand");
    p.its_inter = LAT.its_inter.data(); p.its_fail = LAT.its_fail.data(); p.its_nidne = LAT.its_nidne.data(); p.its_types = LAT.its_types.data();
    p.tmpl = E.tmpl.view(); p.tmpl_taints = E.tmpl_taints.data(); p.tmpl_daemon = E.tmpl_daemon.data(); p.tmpl_daemon_present = E.tmpl_daemon_present.data(); p.tmpl_types = E.tmpl_types.data();
    p.tmpl_limit_present = E.tmpl_limit_present.data(); p.tmpl_remaining = E.tmpl_remaining.data();
    p.en = E.en.view(); p.en_taints = E.en_taints.data(); p.en_avail = E.en_avail.data(); p.en_requests = E.en_requests.data(); p.en_requests_present = E.en_requests_present.data(); p.en_port_off = E.en_port_off.data();
    p.cls = E.cls.view(); p.cls_hn_mode = E.cls_hn_mode.data(); p.cls_hn_off = E.cls_hn_off.data(); p.hn_list = E.hn_list.data(); p.cls_requests = E.cls_requests.data(); p.cls_requests_present = E.cls_requests_present.data();
    p.cls_tolerated = E.cls_tolerated.data(); p.cls_port_off = E.cls_port_off.data(); p.ports = E.ports.data();
    p.en_vol_limit = E.en_vol_limit.data(); p.en_vol_count = E.en_vol_count.data(); p.en_vol_set = E.en_vol_set.data(); p.cls_vol_off = E.cls_vol_off.data(); p.vol_list = E.vol_list.data();
    p.cls_own_off = E.cls_own_off.data(); p.own_list = E.own_list.data(); p.cls_sel_off = E.cls_sel_off.data(); p.sel_list = E.sel_list.data();
    p.cls_isel_off = E.cls_isel_off.data(); p.isel_list = E.isel_list.data(); p.cls_iown_off = E.cls_iown_off.data(); p.iown_list = E.iown_list.data();
    p.pod_stage_off = E.pod_stage_off.data(); p.stage_cls = E.stage_cls.data(); p.queue = E.queue.data();
    p.grp_type = E.grp_type.data(); p.grp_key = E.grp_key.data(); p.grp_max_skew = E.grp_max_skew.data(); p.grp_active = E.grp_active.data(); p.grp_filter_off = E.grp_filter_off.data(); p.flt = E.flt.view();
    p.grp_count = E.grp_count.data(); p.grp_hslot = E.grp_hslot.data(); p.grph_count = E.grph_count.data(); p.grph_extra_pos = E.grph_extra_pos.data();
  }

  bool specs_done = false, active_done = false;      // encode() with an EnvCache runs these two first (it needs the signature)
  // env mode: the batch against the cached flattening of its environment -- everything that does not depend on the pods is adopted
  void run_env() {
    const bool timing = getenv("KSH_TIMING") != nullptr; auto t0 = std::chrono::steady_clock::now();
    adopt_base(); encode_existing(); encode_pods(); encode_existing_rest_from_base(); encode_groups(); encode_it_states(); encode_volumes(); finish();
    if (timing) fprintf(stderr, "  encode %-24s %8.2f ms\n", "batch over cached env", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
  }
  void run() {
    const bool timing = getenv("KSH_TIMING") != nullptr;
    auto t0 = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) { if (!timing) return; auto t1 = std::chrono::steady_clock::now(); fprintf(stderr, "  encode %-24s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count()); t0 = t1; };
    if (base) {      // a what-if over a shared snapshot: catalogue, universes, templates and state-node rows come from the snapshot's flattening
      adopt_base(); dedupe_specs(); encode_existing(); encode_pods(); encode_existing_rest_from_base(); encode_groups(); encode_it_states(); encode_volumes(); finish(); lap("what-if over shared snapshot");
      return;
    }
    if (!specs_done) { dedupe_specs(); lap("dedupe_specs"); }
    if (!active_done) collect_active();
    if (keep_warm_state) { act_sig = active_signature(); act_key_id = key_id; act_res_id = res_id; n_nodes_built = pr.nodes.size(); n_pods_built = podp.size(); }
    warm = can_continue();
    collect_passive(); lap(warm ? "universes (continued)" : "collect_universes");
    encode_instance_types(); lap("encode_instance_types");
    it_reqs.push_back(Requirement()); it_cols.push_back(Requirement());   // state / column 0 == key absent
    encode_templates(); lap("encode_templates");
    encode_existing();
    collect_taints(); lap("encode_existing+taints");
    encode_pods(); lap("encode_pods");
    encode_existing_rest(); lap("encode_existing_rest");
    encode_groups(); lap("encode_groups");
    encode_it_states(); lap("encode_it_states");
    encode_volumes();
    finish(); lap("finish");
  }
};

}  // namespace

// The flattening of an environment (catalogue, universes, templates, state-node rows, instance-type lattice) for ONE universe signature, kept
// with the caller's objects: the next batch with the same signature adopts it instead of encoding 2 000 instance types again.
// Teardown off the caller's thread: what a flattening no longer needs is handed over and destroyed here (one thread per process, started on first use; a process
// that inherited the object through fork() without its thread starts its own).
class Reaper {
 public:
  static Reaper& get() { static Reaper* r = new Reaper(); return *r; }      // (never destroyed: the thread may outlive static destructors)
  void take(std::shared_ptr<void> p) {
    if (!p) return;
    std::lock_guard<std::mutex> g(m_);
    if (pid_ != getpid()) { pid_ = getpid(); q_.clear(); started_ = false; }
    q_.push_back(std::move(p));
    if (!started_) { try { std::thread([this] { loop(); }).detach(); started_ = true; } catch (...) { q_.clear(); return; } }      // (no thread: destroyed here, by clear())
    cv_.notify_one();
  }
 private:
  void loop() {
    for (;;) {
      std::vector<std::shared_ptr<void>> mine;
      { std::unique_lock<std::mutex> g(m_); cv_.wait(g, [&] { return !q_.empty(); }); mine.swap(q_); }
      mine.clear();
    }
  }
  std::mutex m_; std::condition_variable cv_; std::vector<std::shared_ptr<void>> q_; bool started_ = false; pid_t pid_ = getpid();
};
void dispose_later(std::shared_ptr<const void> p) { Reaper::get().take(std::const_pointer_cast<void>(std::move(p))); }
struct EnvBase { std::string sig; uint32_t flags = 0; std::shared_ptr<Encoded> enc; std::unique_ptr<Builder> builder; };
EnvCache::EnvCache() {}
EnvCache::~EnvCache() {}
static std::unique_ptr<Encoded> encode_cached(std::unique_ptr<Encoded> e, uint32_t flags, EnvCache* cache) {
  static const bool off = getenv("KSH_NO_ENV_CACHE") != nullptr;
  // (the builder's working set -- a deep copy of every distinct spec, the per-pod tables -- is torn down on a thread of its own, while the GPU solves: a millisecond
  // of free() that Solve's caller need not wait for)
  auto bp = std::make_shared<Builder>(*e, flags); Builder& b = *bp;
  struct Hand { std::shared_ptr<Builder>& p; ~Hand() { Reaper::get().take(std::move(p)); } } hand{bp};      // (on every way out)
  if (!cache || off) { b.run(); return e; }
  const bool timing = getenv("KSH_TIMING") != nullptr; auto t0 = std::chrono::steady_clock::now();
  b.dedupe_specs(); b.specs_done = true; b.collect_active(); b.active_done = true;
  const std::string sig = b.active_signature();
  if (timing) fprintf(stderr, "  encode %-24s %8.2f ms\n", "dedupe + signature", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
  std::shared_ptr<const EnvBase> eb;
  { std::lock_guard<std::mutex> g(cache->mu); if (cache->base && cache->base->sig == sig && cache->base->flags == flags) eb = cache->base; }
  if (!eb) {
    // first batch with this signature: flatten completely (this batch is as good as any to close the universes over) and keep the result;
    // the batch itself then takes the same road every later one takes, so that a hit and a miss produce the same flat problem
    auto nb = std::make_shared<EnvBase>(); nb->sig = sig; nb->flags = flags; nb->enc = std::make_shared<Encoded>(); nb->enc->src = e->src; nb->enc->batch = e->batch;
    nb->builder = std::make_unique<Builder>(*nb->enc, flags); nb->builder->run();
    eb = nb; std::lock_guard<std::mutex> g(cache->mu); cache->base = nb;
  }
  b.base = eb->builder.get(); b.env_mode = true; e->shared = eb->enc;
  b.env_removed.assign(e->src->nodes.size(), 0); for (size_t i = 0; i < e->src->nodes.size(); ++i) b.env_removed[i] = e->src->nodes[i].in_state ? 0 : 1;
  b.removed = &b.env_removed;
  b.run_env();
  return e;
}
std::unique_ptr<Encoded> encode(std::shared_ptr<const ksp::Problem> pr, uint32_t flags, EnvCache* cache) {
  for (int attempt = 0;; ++attempt) {
    auto e = std::make_unique<Encoded>(); e->src = pr;
    struct Flag { bool was = tl_sync_confirm; ~Flag() { tl_sync_confirm = was; } } flag; if (attempt) tl_sync_confirm = true;
    try { return encode_cached(std::move(e), flags, cache); } catch (const ConfirmFailed&) { if (attempt) throw std::logic_error("spec confirmation failed twice"); }
  }
}

std::unique_ptr<Encoded> encode(std::shared_ptr<const ksp::Problem> env, std::shared_ptr<const ksp::PodBatch> batch, uint32_t flags, EnvCache* cache) {
  if (!env->pods.empty()) throw ksp::Error("the environment of a binary pod batch must carry no pods of its own (PODS 0)");
  auto e = std::make_unique<Encoded>(); e->src = std::move(env); e->batch = std::move(batch);
  return encode_cached(std::move(e), flags, cache);
}

// Binary pod ingress (include/kshost.h, kspb.hpp).  Per block: hash every record (all host threads), partition the block's pods by
// record equality (word for word -- the block's strings are interned), decode the DISTINCT records only, merge those across blocks
// field by field (a second block has another string table).  Spec ids follow the first occurrence in the concatenated batch.
std::shared_ptr<const ksp::PodBatch> ingest_pod_blocks(const ksh_pod_block* blocks, uint32_t nb) {
  auto out = std::make_shared<ksp::PodBatch>();
  auto tl = std::chrono::steady_clock::now(); const bool timing = getenv("KSH_TIMING") != nullptr;
  auto lap = [&](const char* what) { if (!timing) return; auto t1 = std::chrono::steady_clock::now(); fprintf(stderr, "  ingest %-24s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(t1 - tl).count()); tl = t1; };
  std::vector<size_t> pod0(nb + 1, 0);
  for (uint32_t b = 0; b < nb; ++b) {
    const ksh_pod_block& B = blocks[b];
    if (B.n_pods && (!B.spec_off || !B.spec_words || !B.uid || !B.creation_ts)) throw ksp::Error("pod block: null array");
    if (B.n_strings && (!B.str_off || !B.str_bytes)) throw ksp::Error("pod block: null string table");
    for (uint32_t i = 0; i < B.n_strings; ++i) if (B.str_off[i] > B.str_off[i + 1]) throw ksp::Error("pod block: string offsets must ascend");
    if (B.n_strings && B.str_off[B.n_strings] > B.str_bytes_len) throw ksp::Error("pod block: string offsets reach beyond str_bytes_len");
    for (uint32_t i = 0; i < B.n_pods; ++i) if (B.spec_off[i] > B.spec_off[i + 1]) throw ksp::Error("pod block: record offsets must ascend");
    if (B.n_pods && B.spec_off[B.n_pods] > B.spec_words_len) throw ksp::Error("pod block: record offsets reach beyond spec_words_len");
    pod0[b + 1] = pod0[b] + B.n_pods;
  }
  const size_t P = pod0[nb];
  if (P >= (1ull << 31)) throw ksp::Error("pod block: too many pods");
  auto block_of = [&](size_t i) { uint32_t b = 0; while (i >= pod0[b + 1]) ++b; return b; };
  // 1. hash every record
  std::vector<Hash128> hs(P); std::atomic<int> bad{0};
  parallel_chunks(P, [&](size_t lo, size_t hi, uint32_t) {
    if (lo >= hi) return; uint32_t b = block_of(lo);
    for (size_t i = lo; i < hi; ++i) {
      while (i >= pod0[b + 1]) ++b;
      const ksh_pod_block& B = blocks[b]; const size_t l = i - pod0[b];
      if (B.spec_off[l] > B.spec_off[l + 1] || B.uid[l] >= B.n_strings) { bad = 1; continue; }
      Hash128 h; h.bytes((const char*)(B.spec_words + B.spec_off[l]), 4 * (size_t)(B.spec_off[l + 1] - B.spec_off[l])); h.finish(); hs[i] = h;
    } });
  if (bad) throw ksp::Error("pod block: record offsets must ascend and uid ids must name a string");
  lap("hash records");
  // 2. per block: partition by record equality (first occurrences in pod order)
  struct Local { std::vector<uint32_t> first; std::vector<uint32_t> of; std::vector<uint32_t> global; };
  std::vector<Local> loc(nb); std::vector<std::exception_ptr> errs(nb);
  run_threads(nb, [&](uint32_t b) { try {
    const ksh_pod_block& B = blocks[b]; Local& L = loc[b]; const uint32_t n = B.n_pods; L.of.resize(n);
    uint64_t cap = 64; while (cap < 4ull * n) cap <<= 1;
    std::vector<int32_t> tab(cap, -1);
    for (uint32_t i = 0; i < n; ++i) {
      const Hash128& h = hs[pod0[b] + i]; const uint32_t len = B.spec_off[i + 1] - B.spec_off[i]; int32_t found = -1;
      uint64_t j = h.a & (cap - 1);
      for (;; j = (j + 1) & (cap - 1)) {
        const int32_t sidx = tab[j]; if (sidx < 0) break;
        const uint32_t f = L.first[sidx]; const Hash128& o = hs[pod0[b] + f];
        if (o.a == h.a && o.b == h.b && B.spec_off[f + 1] - B.spec_off[f] == len && memcmp(B.spec_words + B.spec_off[f], B.spec_words + B.spec_off[i], 4 * (size_t)len) == 0) { found = sidx; break; }
      }
      if (found < 0) { found = (int32_t)L.first.size(); tab[j] = found; L.first.push_back(i); }
      L.of[i] = (uint32_t)found;
    } } catch (...) { errs[b] = std::current_exception(); } });
  for (auto& e : errs) if (e) std::rethrow_exception(e);
  lap("partition blocks");
  // 3. decode the distinct records (all blocks, in parallel), then merge them in batch order by content
  std::vector<size_t> d0(nb + 1, 0); for (uint32_t b = 0; b < nb; ++b) d0[b + 1] = d0[b] + loc[b].first.size();
  const size_t D = d0[nb]; std::vector<Pod> dec(D); std::vector<Hash128> dh(D);
  parallel_chunks(D, [&](size_t lo, size_t hi, uint32_t) {
    for (size_t x = lo; x < hi; ++x) {
      uint32_t b = 0; while (x >= d0[b + 1]) ++b;
      const ksh_pod_block& B = blocks[b]; const uint32_t f = loc[b].first[x - d0[b]];
      dec[x] = ksp::SpecReader(B, B.spec_words + B.spec_off[f], B.spec_words + B.spec_off[f + 1]).read();
      dh[x] = spec_hash(dec[x]);
      Hash128& h = dh[x]; h.u(dec[x].volume_error ? 1 : 0); h.u(dec[x].volumes.size()); for (auto& v : dec[x].volumes) { h.str(v.driver); h.str(v.pvc); } h.finish();
    } }, 64);
  lap("decode distinct");
  auto same_volumes = [](const Pod& a, const Pod& b) { if (a.volume_error != b.volume_error || a.volumes.size() != b.volumes.size()) return false; for (size_t i = 0; i < a.volumes.size(); ++i) if (a.volumes[i].driver != b.volumes[i].driver || a.volumes[i].pvc != b.volumes[i].pvc) return false; return true; };
  { uint64_t cap = 64; while (cap < 4ull * D) cap <<= 1; std::vector<int32_t> tab(cap, -1); std::vector<size_t> rep;      // rep[g] = index into dec of global spec g
    for (uint32_t b = 0; b < nb; ++b) { loc[b].global.resize(loc[b].first.size());
      for (size_t l = 0; l < loc[b].first.size(); ++l) {
        const size_t x = d0[b] + l; int32_t found = -1; uint64_t j = dh[x].a & (cap - 1);
        for (;; j = (j + 1) & (cap - 1)) { const int32_t g = tab[j]; if (g < 0) break; const size_t y = rep[g]; if (dh[y].a == dh[x].a && dh[y].b == dh[x].b && same_spec(dec[y], dec[x]) && same_volumes(dec[y], dec[x])) { found = g; break; } }
        if (found < 0) { found = (int32_t)rep.size(); tab[j] = found; rep.push_back(x); }
        loc[b].global[l] = (uint32_t)found;
      } }
    out->specs.reserve(rep.size()); for (size_t x : rep) out->specs.push_back(std::move(dec[x])); }
  lap("merge across blocks");
  // 4. per pod: spec id, timestamp, uid bytes
  out->pod_spec.resize(P); out->ts.resize(P); out->uid_off.resize(P + 1); out->uid_off[0] = 0;
  { size_t off = 0; for (uint32_t b = 0; b < nb; ++b) { const ksh_pod_block& B = blocks[b]; for (uint32_t i = 0; i < B.n_pods; ++i) { off += B.str_off[B.uid[i] + 1] - B.str_off[B.uid[i]]; if (off >= (1ull << 32)) throw ksp::Error("pod block: uids exceed 4 GiB"); out->uid_off[pod0[b] + i + 1] = (uint32_t)off; } }
    out->uid_bytes.resize(off); }
  parallel_chunks(P, [&](size_t lo, size_t hi, uint32_t) {
    if (lo >= hi) return; uint32_t b = block_of(lo);
    for (size_t i = lo; i < hi; ++i) {
      while (i >= pod0[b + 1]) ++b;
      const ksh_pod_block& B = blocks[b]; const size_t l = i - pod0[b];
      out->pod_spec[i] = loc[b].global[loc[b].of[l]]; out->ts[i] = B.creation_ts[l];
      memcpy(&out->uid_bytes[out->uid_off[i]], B.str_bytes + B.str_off[B.uid[l]], out->uid_off[i + 1] - out->uid_off[i]);
    } });
  lap("per-pod arrays");
  return out;
}

struct SnapshotBase {
  bool continued = false;      // the flattening continued the one before (ksh_env_apply) instead of starting over
  std::shared_ptr<const ksp::Problem> snapshot; std::shared_ptr<Encoded> enc; std::unique_ptr<Builder> builder;
  std::vector<std::vector<uint32_t>> by_node;      // pods bound to each node, in pod order
  std::vector<int32_t> node_row, node_tmpl; std::vector<int64_t> node_cap; bool delta_ok = false; std::string delta_why;      // (delta_inputs)
  // snapshots with topology groups: what a what-if's NewTopology takes from its candidate set (ksolve.h ks_whatif_topo)
  std::vector<int32_t> t_node_cnt, t_node_dom, t_tot, t_extra_tot, t_grph_base; std::vector<uint64_t> t_node_own; ks_whatif_topo topo{}; bool has_topo = false;
};
// The per-node tables behind ks_whatif_topo.  A what-if's groups are the snapshot's (its pods' specs are a subset); what depends on the candidate set is
// which groups a pod of the batch owns from the start, and which cluster pods countDomains still sees: a bound pod's cluster-pod record counts exactly
// while its node stays.  Returns "" or what stands in the way.
static std::string build_topo_tables(SnapshotBase& sb, const int32_t* pod_node) {
  const Builder& b = *sb.builder; const Encoded& E = *sb.enc; const ksp::Problem& pr = *sb.snapshot;
  const uint32_t G = E.prob.G, GH = E.prob.GH, NE = E.prob.E, NT = E.prob.n_topologies; const size_t NN = pr.nodes.size();
  if (G > 1024) return "more than 1024 topology groups";
  const uint32_t GW = (G + 63) / 64;
  if (b.shared_filter_differs) return "two pods share a spread group while their node filters differ: the group's filter is that of the first pod of each batch";
  // A derived what-if runs on the SNAPSHOT's classes, whose record lists name every group of the snapshot that selects the pod -- a what-if flattened by
  // itself lists only its own.  The kernel's per-class limit (KS_MAX_REC = 24 recorded groups) must hold for the longer lists.
  for (uint32_t c = 0; c + 1 < E.cls_sel_off.size(); ++c)
    if ((E.cls_sel_off[c + 1] - E.cls_sel_off[c]) + (E.cls_iown_off[c + 1] - E.cls_iown_off[c]) > 24) return "a pod is selected by more than 24 of the snapshot's topology groups (the kernel's per-class record list)";
  auto grp = [&](uint32_t gi) -> const Group& { return *b.groups[b.group_order[gi]]; };
  // Inverse groups (required anti-affinity, topology.go:181-199,202-229) EXIST only while an owner is in the batch or stays bound outside it.  A hostname-keyed
  // one whose counts are all zero constrains nothing (every hostname is registered with 0, nothing is narrowed), so it may simply exist in every what-if;
  // a value-keyed one narrows the node's requirement to its registered domains by existing: ks_derive_topology decides per what-if whether it does
  // (an owner in the batch: node_own; an owner that stays: the counts) and the evaluation skips a group that does not.
  if (NT < G) for (auto& n : pr.nodes) { auto hl = n.labels.find(ksp::kHostname); if (hl != n.labels.end() && hl->second.empty()) return "a node with an empty hostname label under hostname-keyed anti-affinity"; }
  sb.t_node_cnt.assign((size_t)G * NN, 0); sb.t_node_dom.assign((size_t)G * NN, -1); sb.t_node_own.assign(NN * GW, 0); sb.t_tot.assign((size_t)G * 64, 0);
  sb.t_extra_tot.assign(GH, 0); sb.t_grph_base.assign((size_t)GH * NE, 0);
  for (size_t i = 0; i < pr.pods.size(); ++i) {
    const auto& sg = b.specs[b.pod_spec[i]].stages[0].sg;      // (required anti-affinity terms survive every relaxation: the first stage owns what all stages own)
    if (pod_node[i] < 0) continue;      // (bound nowhere any more)
    for (const auto* l : {&sg.own, &sg.iown}) for (int g : *l) { const uint32_t gi = (uint32_t)b.group_remap[g]; sb.t_node_own[(size_t)pod_node[i] * GW + (gi >> 6)] |= 1ull << (gi & 63u); }
  }
  // does group g count pods on node n at all (key present, node filter), and under which domain
  std::vector<uint8_t> counts_on((size_t)G * NN, 0); std::vector<int32_t> key_of(G, -1);
  for (uint32_t g = 0; g < G; ++g) if (grp(g).key != ksp::kHostname) key_of[g] = b.key_id.at(grp(g).key);
  for (size_t n = 0; n < NN; ++n) {
    const Requirements nr = Requirements::FromLabels(pr.nodes[n].labels);
    for (uint32_t g = 0; g < G; ++g) {
      const Group& gr = grp(g); const bool host = gr.key == ksp::kHostname;
      auto lt = pr.nodes[n].labels.find(gr.key);
      if (g >= NT) { if (lt == pr.nodes[n].labels.end()) continue; }      // updateInverseAntiAffinity reads the node's label, nothing else
      else { if (!host && lt == pr.nodes[n].labels.end()) continue; if (!FilterMatches(gr.filter, nr, b.wellKnown)) continue; }
      counts_on[g * NN + n] = 1;
      if (!host) sb.t_node_dom[g * NN + n] = b.value_id(key_of[g], lt->second); else sb.t_node_dom[g * NN + n] = 0;      // (hostname-keyed: the count of the pods that are never in a batch)
    }
  }
  std::map<std::string, size_t> node_index; for (size_t n = 0; n < NN; ++n) node_index.emplace(pr.nodes[n].name, n);
  std::unordered_map<std::string, std::vector<uint32_t>> selects;      // (namespace, labels) -> groups that list such a pod (TopologyListOptions: a nil selector lists everything)
  auto count_one = [&](uint32_t g, size_t n, bool in_a_batch) -> const char* {
    if (!counts_on[g * NN + n]) return nullptr;
    const bool host = key_of[g] < 0;
    if (in_a_batch) { sb.t_node_cnt[g * NN + n]++; if (!host) { const int32_t d = sb.t_node_dom[g * NN + n]; if (d < 0 || d >= 64) return "a counted topology domain is missing from the universe"; sb.t_tot[(size_t)g * 64 + d]++; } }
    else if (host) sb.t_node_dom[g * NN + n]++;      // (value-keyed groups: the snapshot's own grp_count already holds the pods that are never in a batch)
    return nullptr;
  };
  for (auto& cp : pr.cluster_pods) {
    const int64_t pi = b.batch_uids.find(cp.uid);
    auto ni = node_index.find(cp.node_name); if (ni == node_index.end()) continue;      // countDomains skips pods whose node it cannot find
    const size_t n = ni->second;
    if (pi >= 0 && (size_t)pod_node[pi] != n) return "a cluster pod record names another node than the one its pod is bound to";
    std::string sig = cp.ns; sig += '\3'; sig_map(sig, cp.labels);
    auto it = selects.find(sig);
    if (it == selects.end()) {
      std::vector<uint32_t> m;
      for (uint32_t g = 0; g < NT; ++g) { const Group& gr = grp(g); if (gr.namespaces.count(cp.ns) && (gr.selector.nil || SelectorMatches(gr.selector, cp.labels))) m.push_back(g); }
      it = selects.emplace(std::move(sig), std::move(m)).first;
    }
    for (uint32_t g : it->second) if (const char* why = count_one(g, n, pi >= 0)) return why;
    for (auto& t : cp.anti_required) {      // the inverse group its required anti-affinity owns while the pod stays bound (topology.go:181-199)
      auto gi = b.inverse_by_id.find(Builder::group_id(2, t.topology_key, Builder::ns_list(cp.ns, t.namespaces), t.selector, INT32_MAX, Filter{}));
      if (gi == b.inverse_by_id.end()) return "a cluster pod's required anti-affinity names a group none of the bound pods owns";
      if (const char* why = count_one((uint32_t)b.group_remap[gi->second], n, pi >= 0)) return why;
    }
  }
  for (uint32_t g = 0; g < G; ++g) {
    const int32_t hs = E.grp_hslot[g]; if (hs < 0) continue;
    const Group& gr = grp(g);
    for (size_t n = 0; n < NN; ++n) {
      const int32_t c = sb.t_node_cnt[g * NN + n] + sb.t_node_dom[g * NN + n], e = sb.node_row[n];
      if (e < 0) { if (c > 0) sb.t_extra_tot[hs]++; continue; }
      auto hl = pr.nodes[n].labels.find(ksp::kHostname); const std::string& hostname = (hl == pr.nodes[n].labels.end() || hl->second.empty()) ? pr.nodes[n].name : hl->second;
      sb.t_grph_base[(size_t)hs * NE + e] = c > 0 ? c : ((g >= NT || gr.counts.count(hostname)) ? 0 : -2);      // -2: registered only if a pod of the batch owns the group (existingnode.go:73)
    }
  }
  sb.topo.node_cnt = sb.t_node_cnt.data(); sb.topo.node_dom = sb.t_node_dom.data(); sb.topo.node_own = sb.t_node_own.data(); sb.topo.tot = sb.t_tot.data();
  sb.topo.extra_tot = sb.t_extra_tot.data(); sb.topo.grph_base = sb.t_grph_base.data(); sb.has_topo = true;
  return "";
}
std::shared_ptr<const SnapshotBase> make_snapshot_base(std::shared_ptr<const ksp::Problem> snapshot, const int32_t* pod_node, uint32_t flags, const SnapshotBase* before) {
  auto sb = std::make_shared<SnapshotBase>(); sb->snapshot = snapshot;
  sb->by_node.resize(snapshot->nodes.size());
  // pod_node[i] = -1: a pod that is bound nowhere any more (ksh_env_apply keeps it in place: nothing that points into the problem moves); it is in no what-if's batch
  for (size_t i = 0; i < snapshot->pods.size(); ++i) { if (pod_node[i] < 0) continue; if ((size_t)pod_node[i] >= snapshot->nodes.size()) throw ksp::Error("pod_node out of range"); sb->by_node[pod_node[i]].push_back((uint32_t)i); }
  sb->enc = std::make_shared<Encoded>(); sb->enc->src = snapshot;
  sb->builder = std::make_unique<Builder>(*sb->enc, flags); sb->builder->keep_warm_state = true;
  if (before && before->snapshot.get() == snapshot.get() && !getenv("KSH_NO_WARM_SNAPSHOT")) sb->builder->prev = before->builder.get();
  sb->builder->run(); sb->continued = sb->builder->warm;
  sb->builder->prev = nullptr;      // (this flattening now stands alone: `before` may go)
  {   // what deriving what-ifs on the device needs (delta_inputs)
    const Builder& b = *sb->builder; const Encoded& E = *sb->enc; const uint32_t R = b.R, M = (uint32_t)E.templates.size(); const size_t NN = snapshot->nodes.size();
    sb->node_row.assign(b.base_existing_of.begin(), b.base_existing_of.end()); sb->node_row.resize(NN, -1);
    sb->node_tmpl.assign(NN, -1); sb->node_cap.assign(NN * R, 0);
    for (size_t i = 0; i < NN; ++i) if (b.base_existing_of[i] >= 0) {
      auto pl = snapshot->nodes[i].labels.find(ksp::kProvisionerName);
      for (uint32_t m = 0; m < M; ++m) if (E.templates[m]->has_limits && E.templates[m]->name == pl->second) {
        sb->node_tmpl[i] = (int32_t)m;
        for (auto& kv : E.templates[m]->limits) { auto c = snapshot->nodes[i].capacity.find(kv.first); if (c != snapshot->nodes[i].capacity.end()) sb->node_cap[i * R + b.res_id.at(kv.first)] = c->second; }
      }
    }
    sb->delta_ok = true;
    if (b.any_volume_limits || b.pods_have_volumes) { sb->delta_ok = false; sb->delta_why = "volume limits / claims: the shared-claim partition depends on the candidate set"; }
    if (sb->delta_ok && !b.groups.empty()) { const std::string why = build_topo_tables(*sb, pod_node); if (!why.empty()) { sb->delta_ok = false; sb->delta_why = why; } }
  }
  return sb;
}
bool snapshot_continued(const SnapshotBase& sb) { return sb.continued; }
uint64_t snapshot_fingerprint(const SnapshotBase& sb) {
  uint64_t h = 1469598103934665603ull;
  auto mix = [&](const void* p, size_t bytes) { const unsigned char* c = (const unsigned char*)p; for (size_t i = 0; i < bytes; ++i) { h ^= c[i]; h *= 1099511628211ull; } };
  auto vec = [&](const auto& v) { uint64_t n = v.size(); mix(&n, 8); if (n) mix(v.data(), n * sizeof(v[0])); };
  vec(sb.node_row); vec(sb.node_tmpl); vec(sb.node_cap); vec(sb.builder->pod_rank); vec(sb.builder->pod_spec); for (auto& l : sb.by_node) vec(l);
  const uint8_t ok = sb.delta_ok, topo = sb.has_topo; mix(&ok, 1); mix(&topo, 1); mix(sb.delta_why.data(), sb.delta_why.size());
  vec(sb.t_node_cnt); vec(sb.t_node_dom); vec(sb.t_tot); vec(sb.t_extra_tot); vec(sb.t_grph_base); vec(sb.t_node_own);
  return h;
}
DeltaInputs delta_inputs(const SnapshotBase& sb) {
  DeltaInputs d; d.base = sb.enc; d.n_nodes = (uint32_t)sb.snapshot->nodes.size(); d.node_row = sb.node_row.data(); d.by_node = &sb.by_node; d.pod_rank = sb.builder->pod_rank.data();
  d.node_cap = sb.node_cap.data(); d.node_tmpl = sb.node_tmpl.data(); d.eligible = sb.delta_ok; d.why = sb.delta_why; d.topo = sb.has_topo ? &sb.topo : nullptr; return d;
}
std::unique_ptr<Encoded> encode_whatif(const SnapshotBase& sb, const uint32_t* cand, uint32_t ncand, uint32_t flags) {
  auto e = std::make_unique<Encoded>(); e->src = sb.snapshot; e->shared = sb.enc;
  std::vector<uint8_t> removed(sb.snapshot->nodes.size(), 0);
  Builder b(*e, flags); b.base = sb.builder.get(); b.removed = &removed;
  for (uint32_t i = 0; i < ncand; ++i) { if (cand[i] >= removed.size()) throw ksp::Error("candidate node out of range"); removed[cand[i]] = 1; for (uint32_t p : sb.by_node[cand[i]]) b.podp.push_back(&sb.snapshot->pods[p]); }
  b.run();
  return e;
}

// CPU self-check of the device derivation (tests): what ks_derive_topology computes from the per-node tables for ONE candidate set -- restated here in
// plain loops -- against the what-if flattened by itself (encode_whatif), group by group (matched by identity), domain by domain, row by row.  "" or the
// first difference.  Not on any solving path.
std::string check_derived_topology(const SnapshotBase& sb, const uint32_t* cand, uint32_t ncand, uint32_t flags) {
  if (!sb.delta_ok) return "not derivable: " + sb.delta_why;
  const Builder& bb = *sb.builder; const Encoded& EB = *sb.enc; const uint32_t G = EB.prob.G, GH = EB.prob.GH, NE = EB.prob.E, NT = EB.prob.n_topologies, GW = (G + 63) / 64;
  const size_t NN = sb.snapshot->nodes.size();
  auto e = std::make_unique<Encoded>(); e->src = sb.snapshot; e->shared = sb.enc;
  std::vector<uint8_t> removed(NN, 0);
  Builder wb(*e, flags); wb.base = sb.builder.get(); wb.removed = &removed;
  for (uint32_t i = 0; i < ncand; ++i) { if (cand[i] >= NN) throw ksp::Error("candidate node out of range"); removed[cand[i]] = 1; for (uint32_t p : sb.by_node[cand[i]]) wb.podp.push_back(&sb.snapshot->pods[p]); }
  wb.run();
  const Encoded& EW = *e;
  if (!G) return EW.prob.G ? "the what-if has groups the snapshot lacks" : "";
  // ---- the derivation (ksolve.hip ks_derive_topology + ks_host_count0) ----
  std::vector<uint8_t> active(G, 0); std::vector<int32_t> count((size_t)G * 64, -1), extra(GH, 0); std::vector<uint8_t> exists(G, 1);
  std::vector<uint64_t> own(GW, 0); for (uint32_t i = 0; i < ncand; ++i) for (uint32_t w = 0; w < GW; ++w) own[w] |= sb.t_node_own[(size_t)cand[i] * GW + w];
  for (uint32_t g = 0; g < G; ++g) {
    const bool ownbit = (own[g >> 6] >> (g & 63u)) & 1ull;
    active[g] = g >= NT ? 1 : (ownbit ? 1 : 0);
    const int32_t hs = EB.grp_hslot[g];
    if (hs >= 0) {
      int32_t ex = sb.t_extra_tot[hs];
      for (uint32_t i = 0; i < ncand; ++i) { const uint32_t nd = cand[i]; const int32_t stay = sb.t_node_dom[g * NN + nd], all = stay + sb.t_node_cnt[g * NN + nd]; if (sb.node_row[nd] < 0) ex -= (all > 0) - (stay > 0); else ex += stay > 0; }
      extra[hs] = ex; continue;
    }
    int32_t c[64]; for (int v = 0; v < 64; ++v) c[v] = sb.t_tot[(size_t)g * 64 + v];
    for (uint32_t i = 0; i < ncand; ++i) { const int32_t dm = sb.t_node_dom[g * NN + cand[i]]; if (dm >= 0) c[dm] -= sb.t_node_cnt[g * NN + cand[i]]; }
    bool owners = ownbit;
    for (int v = 0; v < 64; ++v) { const int32_t r = EB.grp_count[(size_t)g * 64 + v]; const int32_t x = (r >= 0 || c[v] > 0) ? (r > 0 ? r : 0) + c[v] : -1; count[(size_t)g * 64 + v] = x; if (x > 0) owners = true; }
    if (g >= NT && !owners) exists[g] = 0;
  }
  auto row = [&](uint32_t hs, uint32_t g, uint32_t eb) -> int32_t { int32_t c = sb.t_grph_base[(size_t)hs * NE + eb]; if (c == -2) c = active[g] ? 0 : -1; return c; };
  // ---- against the what-if's own flattening ----
  std::vector<uint8_t> seen(G, 0);
  auto where = [&](uint32_t gb) { return " (snapshot group " + std::to_string(gb) + ", key " + bb.groups[bb.group_order[gb]]->key + ")"; };
  for (int inv = 0; inv < 2; ++inv) for (auto& kv : (inv ? wb.inverse_by_id : wb.topo_by_id)) {
    const auto& idx = inv ? bb.inverse_by_id : bb.topo_by_id; auto it = idx.find(kv.first);
    if (it == idx.end()) return "the what-if owns a group the snapshot does not have: " + kv.first;
    const uint32_t gw = (uint32_t)wb.group_remap[kv.second], gb = (uint32_t)bb.group_remap[it->second]; seen[gb] = 1;
    if (!exists[gb]) return "a group the what-if has does not exist in the derivation" + where(gb);
    if ((EW.grp_active[gw] != 0) != (active[gb] != 0)) return "group activity differs" + where(gb);
    const int32_t hw = EW.grp_hslot[gw], hb = EB.grp_hslot[gb];
    if ((hw >= 0) != (hb >= 0)) return "hostname slot kind differs" + where(gb);
    if (hb < 0) { for (int v = 0; v < 64; ++v) if (EW.grp_count[(size_t)gw * 64 + v] != count[(size_t)gb * 64 + v]) return "count of domain " + std::to_string(v) + " differs: " + std::to_string(EW.grp_count[(size_t)gw * 64 + v]) + " flattened, " + std::to_string(count[(size_t)gb * 64 + v]) + " derived" + where(gb); continue; }
    if (EW.grph_extra_pos[hw] != extra[hb]) return "count of positive hostnames that are no existing node differs: " + std::to_string(EW.grph_extra_pos[hw]) + " flattened, " + std::to_string(extra[hb]) + " derived" + where(gb);
    for (uint32_t ew = 0; ew < EW.prob.E; ++ew) { const int32_t eb = sb.node_row[EW.existing[ew]]; const int32_t a = EW.grph_count[(size_t)hw * EW.prob.E + ew], d = row((uint32_t)hb, gb, (uint32_t)eb); if (a != d) return "hostname row of node " + std::to_string(EW.existing[ew]) + " differs: " + std::to_string(a) + " flattened, " + std::to_string(d) + " derived" + where(gb); }
  }
  for (uint32_t gb = 0; gb < G; ++gb) if (!seen[gb]) {      // groups of the snapshot this what-if does not have: they must be inert in it
    if (gb < NT) { if (active[gb]) return "a group no pod of the batch owns is active" + where(gb); continue; }
    const int32_t hb = EB.grp_hslot[gb];
    if (hb < 0) { if (exists[gb]) return "an inverse group without owners exists" + where(gb); continue; }
    for (uint32_t ew = 0; ew < EW.prob.E; ++ew) if (row((uint32_t)hb, gb, (uint32_t)sb.node_row[EW.existing[ew]]) != 0) return "an inverse group without owners counts on node " + std::to_string(EW.existing[ew]) + where(gb);
  }
  return "";
}

std::unique_ptr<Encoded::ResultBuf> Encoded::make_result() const {
  auto rb = std::make_unique<ResultBuf>(); const ks_problem& p = prob; const size_t N = p.max_new_nodes, TW = (p.T + 63) / 64;
  rb->pod_node.resize(p.P + 1); rb->pod_stage.resize(p.P + 1); rb->pod_seq.resize(p.P + 1); rb->unscheduled.resize(p.P + 1); rb->pod_reason.resize(p.P + 1);
  rb->node_tmpl.resize(N); rb->node_types.resize(N * TW); rb->node_requests.resize(N * p.R); rb->node_requests_present.resize(N);
  rb->node_present.resize(N); rb->node_complement.resize(N); rb->node_mask.resize(N * p.K + 1); rb->node_gt.resize(N * p.K + 1); rb->node_lt.resize(N * p.K + 1); rb->node_it_state.resize(N);
  ks_result& r = rb->r;
  r.pod_node = rb->pod_node.data(); r.pod_stage = rb->pod_stage.data(); r.pod_seq = rb->pod_seq.data(); r.unscheduled = rb->unscheduled.data(); r.pod_reason = rb->pod_reason.data();
  r.node_tmpl = rb->node_tmpl.data(); r.node_types = rb->node_types.data(); r.node_requests = rb->node_requests.data(); r.node_requests_present = rb->node_requests_present.data();
  r.node_present = rb->node_present.data(); r.node_complement = rb->node_complement.data(); r.node_mask = rb->node_mask.data(); r.node_gt = rb->node_gt.data(); r.node_lt = rb->node_lt.data(); r.node_it_state = rb->node_it_state.data();
  return rb;
}

static std::string tokq(const std::string& s) { return s.empty() ? "~" : s; }

std::string Encoded::decode(const ks_result& r, double solve_seconds) const {
  if (view && shared) {      // a what-if derived on the device: every naming table is the snapshot's; the dimensions are this what-if's
    Encoded tmp; tmp.src = shared->src; tmp.key_names = shared->key_names; tmp.key_values = shared->key_values; tmp.key_members = shared->key_members; tmp.key_ints = shared->key_ints; tmp.res_names = shared->res_names; tmp.templates = shared->templates;
    tmp.existing = shared->existing; tmp.shared = shared; tmp.shared_lattice = true; tmp.prob = prob;
    return tmp.decode(r, solve_seconds);
  }
  const ks_problem& p = prob; const uint32_t TW = (p.T + 63) / 64, NE = p.E;
  std::vector<std::vector<std::pair<int32_t, int32_t>>> pods_of(NE + r.n_new);   // (seq, pod)
  for (uint32_t i = 0; i < p.P; ++i) if (r.pod_node[i] >= 0) pods_of[r.pod_node[i]].push_back({r.pod_seq[i], (int32_t)i});
  for (auto& v : pods_of) std::sort(v.begin(), v.end());
  std::ostringstream o;
  o << "KSR1\nNEWNODES " << r.n_new << "\n";
  for (uint32_t j = 0; j < r.n_new; ++j) {
    const auto& prov = *templates[r.node_tmpl[j]];
    o << "NODE " << tokq(prov.name) << " " << pods_of[NE + j].size(); for (auto& sp : pods_of[NE + j]) o << " " << sp.second;
    std::vector<const std::string*> names;
    for (int idx : prov.instance_types) if ((r.node_types[(size_t)j * TW + idx / 64] >> (idx % 64)) & 1ull) names.push_back(&src->instance_types[idx].name);   // order-preserving filter (lo.Filter, node.go:138)
    o << " " << names.size(); for (auto* n : names) o << " " << tokq(*n);
    const uint32_t pm = r.node_requests_present[j]; std::map<std::string, int64_t> req;
    for (uint32_t rr = 0; rr < p.R; ++rr) if ((pm >> rr) & 1u) req[res_names[rr]] = r.node_requests[(size_t)j * p.R + rr];
    o << " " << req.size(); for (auto& kv : req) o << " " << kv.first << " " << kv.second;
    struct Out { bool c; std::vector<std::string> vals; std::string gt, lt; };
    std::map<std::string, Out> reqs;
    for (uint32_t k = 0; k < p.K; ++k) if ((r.node_present[j] >> k) & 1u) {
      Out x; x.c = (r.node_complement[j] >> k) & 1u; const uint64_t m = r.node_mask[(size_t)j * p.K + k];
      for (size_t v = 0; v < key_values[k].size(); ++v) if ((m >> v) & 1ull) {
        if (k < key_members.size() && !key_members[k].empty()) x.vals.insert(x.vals.end(), key_members[k][v].begin(), key_members[k][v].end());      // a value class: every member
        else x.vals.push_back(key_values[k][v]);
      }
      const int32_t gt = r.node_gt[(size_t)j * p.K + k], lt = r.node_lt[(size_t)j * p.K + k];
      const bool ranks = k < key_ints.size() && !key_ints[k].empty();      // (bounds carried as ranks: back to the integers)
      x.gt = gt == KS_NO_BOUND_GT ? "-" : std::to_string(ranks ? key_ints[k].at((size_t)gt) : (long long)gt); x.lt = lt == KS_NO_BOUND_LT ? "-" : std::to_string(ranks ? key_ints[k].at((size_t)lt) : (long long)lt);
      reqs[key_names[k]] = std::move(x);
    }
    if (r.node_it_state[j] > 0) {
      const Requirement& q = lattice().it_states[r.node_it_state[j]]; Out x; x.c = q.complement; x.vals.assign(q.values.begin(), q.values.end());
      x.gt = q.greaterThan ? std::to_string(*q.greaterThan) : "-"; x.lt = q.lessThan ? std::to_string(*q.lessThan) : "-"; reqs[ksp::kInstanceType] = std::move(x);
    }
    o << " " << reqs.size();
    for (auto& kv : reqs) { o << " " << kv.first << " " << (kv.second.c ? 1 : 0) << " " << kv.second.vals.size(); for (auto& v : kv.second.vals) o << " " << tokq(v); o << " " << kv.second.gt << " " << kv.second.lt; }
    o << "\n";
  }
  o << "EXISTING " << NE << "\n";
  for (uint32_t e = 0; e < NE; ++e) { o << "ENODE " << tokq(src->nodes[existing[e]].name) << " " << pods_of[e].size(); for (auto& sp : pods_of[e]) o << " " << sp.second; o << "\n"; }
  o << "UNSCHEDULED " << r.n_unscheduled; for (uint32_t i = 0; i < r.n_unscheduled; ++i) o << " " << r.unscheduled[i]; o << "\n";
  o << "STAGES " << p.P; for (uint32_t i = 0; i < p.P; ++i) o << " " << r.pod_stage[i]; o << "\n";
  o << "REASONS " << r.n_unscheduled; for (uint32_t i = 0; i < r.n_unscheduled; ++i) o << " " << r.unscheduled[i] << " " << r.pod_reason[r.unscheduled[i]]; o << "\n";
  o << "STATS 33 p20 " << r.stats[20] << " eq_pods " << r.stats[8] << " reuse_exhausted " << r.stats[9] << " reuse_seeds " << r.stats[10] << " reuse_hits " << r.stats[11] << " cyc_kind0 " << r.stats[27] << " cyc_kind1 " << r.stats[28] << " cyc_kind2 " << r.stats[29] << " n_kind1 " << r.stats[30] << " n_kind2 " << r.stats[31] << " p22 " << r.stats[22] << " p23 " << r.stats[23] << " p24 " << r.stats[24] << " p25 " << r.stats[25] << " p26 " << r.stats[26] << " cyc_pop " << r.stats[12] << " cyc_stage " << r.stats[13] << " cyc_scan " << r.stats[14] << " cyc_evalout " << r.stats[15] << " cyc_full " << r.stats[16]
    << " cyc_commit " << r.stats[17] << " cyc_order " << r.stats[18] << " cyc_new " << r.stats[19] << " scan_chunks " << r.stats[21]
    << " queue_pops " << r.stats[KS_STAT_POPS] << " relaxations " << r.stats[KS_STAT_RELAX] << " full_checks " << r.stats[KS_STAT_FULLCHECKS] << " full_fails " << r.stats[KS_STAT_FULLFAILS]
    << " attempts " << r.stats[KS_STAT_REF_ATTEMPTS] << " types_scanned " << r.stats[KS_STAT_REF_TYPES] << " kernel_cycles " << r.stats[KS_STAT_CYCLES]
    << " classes " << p.C << " solve_ns " << (int64_t)(solve_seconds * 1e9) << "\n";
  o << "END\n";
  return o.str();
}

}  // namespace ksh
