// oracle.cpp -- CPU restatement of karpenter-core's provisioning scheduler hot path.
//
// *** TEST INFRASTRUCTURE ONLY.  Nothing in the product (karpenter_core_amd/, include/) may link,
// *** import or execute this file.  Only tests/, __graft_entry__.smoke() and bench.py's
// *** cpu_baseline leg use it -- as the checker / the CPU baseline, never as the thing shipped.
//
// What it restates (all paths relative to /root/reference/pkg; the Go toolchain is absent in this
// image, so the reference cannot be compiled -- SURVEY.md 8c):
//   controllers/provisioning/scheduling/{scheduler,node,existingnode,queue,preferences,topology,
//     topologygroup,topologynodefilter}.go, scheduling/{requirement,requirements,taints,
//     hostportusage}.go, utils/resources/resources.go, cloudprovider/types.go:72-145,
//     controllers/provisioning/provisioner.go:237-296 (NewScheduler assembly).
// Third-party pieces restated from their published behaviour (not vendored in the reference):
//   k8s.io/api v0.25.4 core/v1 Toleration.ToleratesTaint / MatchToleration;
//   k8s.io/apimachinery v0.25.4 resource.Quantity (exact decimal -> int64 milli-units here),
//     sets.String, labels.Selector / metav1.LabelSelectorAsSelector;
//   github.com/mitchellh/hashstructure/v2 v2.0.2 (group identity only; see group_identity()).
//
// Pinning: the Requirement algebra is checked against every cell of the reference's own truth
// tables (scheduling/requirement_test.go:81-463, requirements_test.go:50-290) by
// tests/test_oracle_golden.py.  Whole-Solve behaviour is pinned only at the level the reference's
// envtest suites pin it (invariants: node counts, skew multisets, ... -- tests/test_scenarios.py);
// tie-breaks that depend on Go map order / unstable sort.Slice are canonicalised per SURVEY.md
// App. B/C and are therefore "parity unpinned" by the reference itself:
//   * topology domains iterate in ascending byte-wise string order (hostnames: existing nodes in
//     given order, then placeholders by creation id);
//   * sort.Slice(newNodes, by len(Pods)) (scheduler.go:183) is taken to be STABLE;
//   * provisioners order by weight descending, stably (provisioner.go:132-136);
//   * hostname placeholder ids restart at 1 per Solve (node.go:42 is a process-global counter).
//
// Data structures are deliberately reference-shaped (maps keyed by label key, sets of values,
// per-attempt recomputation) -- this file doubles as the "reference CPU path" stand-in that
// bench.py times (cpu_baseline.kind = "port").  Strings are interned to ints at parse time; all
// set algebra is on value sets, not on the bitmask encoding the GPU path uses.
#include <algorithm>
#include <chrono>
#include <climits>
#include <deque>
#include <cstdio>
#include <fstream>
#include <functional>
#include <iostream>
#include <memory>
#include <optional>
#include <set>
#include <sstream>
#include <unordered_map>

#include "../karpenter_core_amd/host/ksp.hpp"

namespace oracle {

using ksp::Expr; using ksp::Op; using ksp::ResList; using ksp::StrMap;

// ------------------------------------------------------------------------------------------------
// string interning (values and keys share one table); Sym order is NOT string order
// ------------------------------------------------------------------------------------------------
using Sym = int32_t;
struct Interner {
  std::unordered_map<std::string, Sym> ids; std::vector<std::string> strs;
  Sym get(const std::string& s) { auto it = ids.find(s); if (it != ids.end()) return it->second; Sym id = (Sym)strs.size(); ids.emplace(s, id); strs.push_back(s); return id; }
  const std::string& str(Sym s) const { return strs[s]; }
};
static thread_local Interner* g_in = nullptr;
static inline Sym S(const std::string& s) { return g_in->get(s); }
static inline const std::string& STR(Sym s) { return g_in->str(s); }

// sets.String (k8s.io/apimachinery/pkg/util/sets) as a sorted vector of symbols
struct SymSet {
  std::vector<Sym> v;
  bool has(Sym s) const { return std::binary_search(v.begin(), v.end(), s); }
  void insert(Sym s) { auto it = std::lower_bound(v.begin(), v.end(), s); if (it == v.end() || *it != s) v.insert(it, s); }
  size_t size() const { return v.size(); }
  bool empty() const { return v.empty(); }
  static SymSet uni(const SymSet& a, const SymSet& b) { SymSet r; std::set_union(a.v.begin(), a.v.end(), b.v.begin(), b.v.end(), std::back_inserter(r.v)); return r; }
  static SymSet inter(const SymSet& a, const SymSet& b) { SymSet r; std::set_intersection(a.v.begin(), a.v.end(), b.v.begin(), b.v.end(), std::back_inserter(r.v)); return r; }
  static SymSet diff(const SymSet& a, const SymSet& b) { SymSet r; std::set_difference(a.v.begin(), a.v.end(), b.v.begin(), b.v.end(), std::back_inserter(r.v)); return r; }
  bool operator==(const SymSet& o) const { return v == o.v; }
};

// strconv.Atoi restated: optional sign, decimal digits only, must fit int64
static bool atoi_go(const std::string& s, int64_t* out) {
  if (s.empty()) return false;
  size_t i = 0; bool neg = false;
  if (s[0] == '+' || s[0] == '-') { neg = s[0] == '-'; i = 1; if (s.size() == 1) return false; }
  __int128 v = 0;
  for (; i < s.size(); ++i) { if (s[i] < '0' || s[i] > '9') return false; v = v * 10 + (s[i] - '0'); if (v > (__int128)INT64_MAX + 1) return false; }
  if (neg) v = -v;
  if (v > INT64_MAX || v < INT64_MIN) return false;
  *out = (int64_t)v; return true;
}

// ------------------------------------------------------------------------------------------------
// Requirement (scheduling/requirement.go:36-269)
// ------------------------------------------------------------------------------------------------
static const int64_t kMaxInt64 = INT64_MAX;

struct Req {
  Sym key = -1; bool complement = false; SymSet values;
  bool has_gt = false, has_lt = false; int64_t gt = 0, lt = 0;
};

// withinIntPtrs, requirement.go:227-243
static bool within(Sym value, bool has_gt, int64_t gt, bool has_lt, int64_t lt) {
  if (!has_gt && !has_lt) return true;
  int64_t v; if (!atoi_go(STR(value), &v)) return false;
  if (has_gt && gt >= v) return false;
  if (has_lt && lt <= v) return false;
  return true;
}

// NewRequirement, requirement.go:44-68
static Req new_req(const std::string& key_in, Op op, const std::vector<std::string>& values) {
  Req r; r.key = S(ksp::normalize_key(key_in)); r.complement = true;
  if (op == Op::In || op == Op::DoesNotExist) r.complement = false;
  if (op == Op::In || op == Op::NotIn) for (auto& v : values) r.values.insert(S(v));
  if (op == Op::Gt) { int64_t v = 0; atoi_go(values.at(0), &v); r.has_gt = true; r.gt = v; }
  if (op == Op::Lt) { int64_t v = 0; atoi_go(values.at(0), &v); r.has_lt = true; r.lt = v; }
  return r;
}
static Req new_req_sym(Sym key, Op op) { Req r; r.key = key; r.complement = !(op == Op::In || op == Op::DoesNotExist); return r; }

// Len, requirement.go:199-204
static int64_t req_len(const Req& r) { return r.complement ? kMaxInt64 - (int64_t)r.values.size() : (int64_t)r.values.size(); }
// Operator, requirement.go:186-197
static Op req_operator(const Req& r) {
  if (r.complement) return req_len(r) < kMaxInt64 ? Op::NotIn : Op::Exists;
  return req_len(r) > 0 ? Op::In : Op::DoesNotExist;
}
// Has, requirement.go:171-176
static bool req_has(const Req& r, Sym v) {
  if (r.complement) return !r.values.has(v) && within(v, r.has_gt, r.gt, r.has_lt, r.lt);
  return r.values.has(v) && within(v, r.has_gt, r.gt, r.has_lt, r.lt);
}
// Intersection, requirement.go:117-150   (receiver r, argument q)
static Req req_intersection(const Req& r, const Req& q) {
  bool complement = r.complement && q.complement;
  bool has_gt = r.has_gt || q.has_gt, has_lt = r.has_lt || q.has_lt; int64_t gt = 0, lt = 0;
  if (r.has_gt && q.has_gt) gt = std::max(r.gt, q.gt); else if (r.has_gt) gt = r.gt; else if (q.has_gt) gt = q.gt;   // maxIntPtr :258
  if (r.has_lt && q.has_lt) lt = std::min(r.lt, q.lt); else if (r.has_lt) lt = r.lt; else if (q.has_lt) lt = q.lt;   // minIntPtr :245
  if (has_gt && has_lt && gt >= lt) return new_req_sym(r.key, Op::DoesNotExist);
  SymSet values;
  if (r.complement && q.complement) values = SymSet::uni(r.values, q.values);
  else if (r.complement && !q.complement) values = SymSet::diff(q.values, r.values);
  else if (!r.complement && q.complement) values = SymSet::diff(r.values, q.values);
  else values = SymSet::inter(r.values, q.values);
  SymSet kept; for (Sym v : values.v) if (within(v, has_gt, gt, has_lt, lt)) kept.v.push_back(v);
  Req out; out.key = r.key; out.complement = complement; out.values = kept;
  if (complement) { out.has_gt = has_gt; out.gt = gt; out.has_lt = has_lt; out.lt = lt; }   // bounds dropped for concrete sets :145
  return out;
}
static bool op_is_notin_or_dne(const Req& r) { Op o = req_operator(r); return o == Op::NotIn || o == Op::DoesNotExist; }

// ------------------------------------------------------------------------------------------------
// Requirements (scheduling/requirements.go:32-223): map key -> Requirement
// ------------------------------------------------------------------------------------------------
struct Reqs {
  std::map<Sym, Req> m;
  bool has(Sym k) const { return m.count(k) != 0; }
  Req get(Sym k) const { auto it = m.find(k); if (it == m.end()) return new_req_sym(k, Op::Exists); return it->second; }   // :114-120
  void add(const Req& q) {   // :87-94  requirement = requirement.Intersection(existing)
    auto it = m.find(q.key);
    if (it != m.end()) it->second = req_intersection(q, it->second); else m.emplace(q.key, q);
  }
  void add_all(const Reqs& o) { for (auto& kv : o.m) add(kv.second); }
};
static Reqs reqs_from_exprs(const std::vector<Expr>& es) { Reqs r; for (auto& e : es) r.add(new_req(e.key, e.op, e.values)); return r; }   // :43-49
static Reqs reqs_from_labels(const StrMap& labels) { Reqs r; for (auto& kv : labels) r.add(new_req(kv.first, Op::In, {kv.second})); return r; }   // :52-58

struct Ctx {   // per-Solve globals
  std::set<Sym> well_known;   // v1alpha5.WellKnownLabels (labels.go:84-92) + provider additions (fake/instancetype.go:40-46)
};

// Intersects, requirements.go:189-206; returns true when NO error
static bool reqs_intersects(const Reqs& r, const Reqs& in) {
  for (auto& kv : r.m) {
    auto it = in.m.find(kv.first); if (it == in.m.end()) continue;
    const Req& existing = kv.second; const Req& incoming = it->second;
    if (req_len(req_intersection(existing, incoming)) == 0) {
      if (op_is_notin_or_dne(incoming) && op_is_notin_or_dne(existing)) continue;
      return false;
    }
  }
  return true;
}
// Compatible, requirements.go:123-133; returns true when NO error
static bool reqs_compatible(const Ctx& cx, const Reqs& r, const Reqs& in) {
  bool ok = true;
  for (auto& kv : in.m) {
    if (cx.well_known.count(kv.first)) continue;
    if (r.has(kv.first) || op_is_notin_or_dne(kv.second)) continue;
    ok = false;   // "label %q does not have known values"
  }
  return reqs_intersects(r, in) && ok;
}

// ------------------------------------------------------------------------------------------------
// resources (utils/resources/resources.go)
// ------------------------------------------------------------------------------------------------
static ResList res_merge(const ResList& a, const ResList& b) { ResList r = a; for (auto& kv : b) r[kv.first] += kv.second; return r; }   // Merge :47-60
static ResList res_subtract(const ResList& lhs, const ResList& rhs) {   // Subtract :62-75 (only keys of lhs)
  ResList r = lhs; for (auto& kv : r) { auto it = rhs.find(kv.first); if (it != rhs.end()) kv.second -= it->second; } return r;
}
static ResList res_max(const ResList& a, const ResList& b) {   // MaxResources :92-102
  ResList r; for (auto* l : {&a, &b}) for (auto& kv : *l) { auto it = r.find(kv.first); if (it == r.end() || kv.second > it->second) r[kv.first] = kv.second; } return r;
}
static ResList limits_into_requests(const ksp::Container& c) {   // MergeResourceLimitsIntoRequests :105-119
  ResList r = c.requests; for (auto& kv : c.limits) if (!r.count(kv.first)) r[kv.first] = kv.second; return r;
}
static ResList pod_ceiling_requests(const ksp::Pod& p) {   // Ceiling :78-89
  ResList r; for (auto& c : p.containers) r = res_merge(r, limits_into_requests(c));
  for (auto& c : p.init_containers) r = res_max(r, limits_into_requests(c));
  return r;
}
static ResList requests_for_pods(const std::vector<const ksp::Pod*>& pods) {   // RequestsForPods :25-33
  ResList merged; for (auto* p : pods) merged = res_merge(merged, pod_ceiling_requests(*p));
  merged["pods"] = (int64_t)pods.size() * 1000;
  return merged;
}
static bool res_fits(const ResList& candidate, const ResList& total) {   // Fits :138-145
  for (auto& kv : candidate) { auto it = total.find(kv.first); int64_t t = it == total.end() ? 0 : it->second; if (kv.second > t) return false; }
  return true;
}

// ------------------------------------------------------------------------------------------------
// taints (scheduling/taints.go:28-40) + k8s.io/api core/v1 Toleration.ToleratesTaint
// ------------------------------------------------------------------------------------------------
static bool tolerates_taint(const ksp::Toleration& t, const ksp::Taint& taint) {
  if (!t.effect.empty() && t.effect != taint.effect) return false;
  if (!t.key.empty() && t.key != taint.key) return false;
  if (t.op.empty() || t.op == "Equal") return t.value == taint.value;
  if (t.op == "Exists") return true;
  return false;
}
static bool taints_tolerates(const std::vector<ksp::Taint>& ts, const ksp::Pod& pod) {
  bool ok = true;
  for (auto& taint : ts) { bool tol = false; for (auto& t : pod.tolerations) tol = tol || tolerates_taint(t, taint); if (!tol) ok = false; }
  return ok;
}

// ------------------------------------------------------------------------------------------------
// host ports (scheduling/hostportusage.go)
// ------------------------------------------------------------------------------------------------
struct PortEntry { std::string ip; int32_t port; std::string proto; };
// net.ParseIP / IP.Equal / IsUnspecified restated for the textual forms the tests use; an IP that
// does not parse is nil: nil.Equal(nil) is true, nil is not "unspecified".
static std::string canon_ip(const std::string& s) {
  // IPv4 dotted quad -> canonical "a.b.c.d"; anything else kept verbatim lower-cased (IPv6 forms
  // are compared textually except the unspecified address "::").
  int a, b, c, d; char tail;
  if (sscanf(s.c_str(), "%d.%d.%d.%d%c", &a, &b, &c, &d, &tail) == 4 && a >= 0 && a < 256 && b >= 0 && b < 256 && c >= 0 && c < 256 && d >= 0 && d < 256) {
    char buf[32]; snprintf(buf, sizeof buf, "%d.%d.%d.%d", a, b, c, d); return buf;
  }
  std::string r = s; for (auto& ch : r) ch = (char)tolower(ch);
  if (r == "0:0:0:0:0:0:0:0") r = "::";
  return r;
}
static bool ip_unspecified(const std::string& c) { return c == "0.0.0.0" || c == "::"; }
static bool port_matches(const PortEntry& e, const PortEntry& rhs) {   // entry.matches :45-57
  if (e.proto != rhs.proto) return false;
  if (e.port != rhs.port) return false;
  if (e.ip != rhs.ip && !ip_unspecified(e.ip) && !ip_unspecified(rhs.ip)) return false;
  return true;
}
static std::vector<PortEntry> get_host_ports(const ksp::Pod& pod) {   // getHostPorts :122-144
  std::vector<PortEntry> u;
  for (auto& c : pod.containers) for (auto& p : c.ports) {
    if (p.port == 0) continue;
    std::string ip = p.ip.empty() ? "0.0.0.0" : p.ip;
    u.push_back({canon_ip(ip), p.port, p.proto});
  }
  return u;
}
struct HostPortUsage {
  std::map<std::string, std::vector<PortEntry>> reserved;   // "ns/name" -> entries (pod uid stands in for the name)
  bool validate(const ksp::Pod& pod, std::vector<PortEntry>* out) const {   // validate :81-93
    auto nu = get_host_ports(pod); std::string me = pod.ns + "/" + pod.uid;
    for (auto& ne : nu) for (auto& kv : reserved) for (auto& ex : kv.second) if (port_matches(ne, ex) && kv.first != me) return false;
    if (out) *out = nu; return true;
  }
  void add(const ksp::Pod& pod) { std::vector<PortEntry> nu; validate(pod, &nu); reserved[pod.ns + "/" + pod.uid] = get_host_ports(pod); }   // Add :66-72
};

// ------------------------------------------------------------------------------------------------
// NewPodRequirements (requirements.go:61-78).  The in-place sort of the preferred terms (:69) is an
// unstable sort.Slice on <= 12 elements == insertion sort == stable.
// ------------------------------------------------------------------------------------------------
static Reqs new_pod_requirements(ksp::Pod& pod) {
  Reqs r = reqs_from_labels(pod.node_selector);
  if (!pod.preferred_affinity.empty()) {
    std::stable_sort(pod.preferred_affinity.begin(), pod.preferred_affinity.end(), [](const ksp::PreferredTerm& a, const ksp::PreferredTerm& b) { return a.weight > b.weight; });
    r.add_all(reqs_from_exprs(pod.preferred_affinity[0].exprs));
  }
  if (!pod.required_affinity.empty()) r.add_all(reqs_from_exprs(pod.required_affinity[0]));
  return r;
}

// ------------------------------------------------------------------------------------------------
// label selectors (metav1.LabelSelectorAsSelector + labels.Selector.Matches)
// ------------------------------------------------------------------------------------------------
static bool selector_matches(const ksp::Selector& sel, const StrMap& labels) {
  if (sel.nil) return false;   // nil selector -> labels.Nothing()
  for (auto& kv : sel.match_labels) { auto it = labels.find(kv.first); if (it == labels.end() || it->second != kv.second) return false; }
  for (auto& e : sel.match_exprs) {
    auto it = labels.find(e.key); bool has = it != labels.end();
    bool in = has && std::find(e.values.begin(), e.values.end(), it->second) != e.values.end();
    switch (e.op) {
      case Op::In: if (!in) return false; break;
      case Op::NotIn: if (has && in) return false; break;   // labels.Requirement NotIn: key absent matches
      case Op::Exists: if (!has) return false; break;
      case Op::DoesNotExist: if (has) return false; break;
      default: return false;   // invalid operator -> LabelSelectorAsSelector error -> labels.Nothing() (topologygroup.go:247-250)
    }
  }
  return true;
}

// ------------------------------------------------------------------------------------------------
// TopologyNodeFilter (topologynodefilter.go:28-70)
// ------------------------------------------------------------------------------------------------
struct NodeFilter { bool always = true; std::vector<Reqs> terms; };
static NodeFilter make_node_filter(const ksp::Pod& p) {
  NodeFilter f; f.always = false;
  Reqs sel = reqs_from_labels(p.node_selector);
  if (p.required_affinity.empty()) { f.terms.push_back(sel); return f; }   // nil RequiredDuringScheduling... == no terms here
  for (auto& term : p.required_affinity) { Reqs r; r.add_all(sel); r.add_all(reqs_from_exprs(term)); f.terms.push_back(r); }
  return f;
}
static bool filter_matches_reqs(const Ctx& cx, const NodeFilter& f, const Reqs& requirements) {   // MatchesRequirements :57-70
  if (f.always || f.terms.empty()) return true;
  for (auto& t : f.terms) if (reqs_compatible(cx, requirements, t)) return true;
  return false;
}

// ------------------------------------------------------------------------------------------------
// TopologyGroup (topologygroup.go)
// ------------------------------------------------------------------------------------------------
enum TopoType { kSpread = 0, kAffinity = 1, kAntiAffinity = 2 };

struct DomKey {   // canonical domain order: strings byte-wise, then placeholders by id
  int kind; std::string s; int64_t id; Sym sym = -1;   // sym: interned name (not part of the order)
  bool operator<(const DomKey& o) const { if (kind != o.kind) return kind < o.kind; if (kind == 0) return s < o.s; return id < o.id; }
};

struct Stats { int64_t attempts = 0, types_scanned = 0, domains_scanned = 0, relaxations = 0, queue_pops = 0; };

struct TopologyGroup {
  Sym key; TopoType type; int32_t max_skew; std::set<std::string> namespaces; ksp::Selector selector; NodeFilter filter;
  std::set<std::string> owners;                 // pod UIDs
  std::map<DomKey, int32_t> domains;            // domain -> count
  bool is_hostname = false;
};

struct Hostnames {   // maps a hostname symbol to its canonical DomKey
  std::unordered_map<Sym, DomKey> keys;
  DomKey of(Sym s) const { auto it = keys.find(s); if (it != keys.end()) return it->second; return DomKey{0, STR(s), 0, s}; }
};

struct Topology;
static bool tg_selects(const TopologyGroup& t, const ksp::Pod& pod) { return t.namespaces.count(pod.ns) && selector_matches(t.selector, pod.labels); }   // :246-252

// Canonical identity standing in for TopologyGroup.Hash() (topologygroup.go:137-153).
// hashstructure v2.0.2 walks exported struct fields only, so for NodeFilter ([]Requirements ==
// []map[string]*Requirement) only the map keys and Requirement.Key contribute: two pods whose node
// selectors / required affinity terms mention the same KEYS (any values/operators) hash to the same
// spread group, and the group keeps the filter of whichever pod created it first.
static std::string selector_identity(const ksp::Selector& s) {
  if (s.nil) return "nil";
  std::string r = "sel{";
  for (auto& kv : s.match_labels) r += kv.first + "=" + kv.second + ",";
  std::vector<std::string> es;
  for (auto& e : s.match_exprs) { std::vector<std::string> vs = e.values; std::sort(vs.begin(), vs.end()); std::string x = e.key + ":" + std::to_string((int)e.op) + "["; for (auto& v : vs) x += v + ","; es.push_back(x + "]"); }
  std::sort(es.begin(), es.end()); for (auto& e : es) r += e + ";";
  return r + "}";
}
static std::string group_identity(const TopologyGroup& t) {
  std::string r = STR(t.key) + "|" + std::to_string((int)t.type) + "|";
  for (auto& n : t.namespaces) r += n + ",";
  r += "|" + selector_identity(t.selector) + "|" + std::to_string(t.max_skew) + "|";
  if (t.filter.always) r += "nofilter"; else {
    std::vector<std::string> terms;
    for (auto& term : t.filter.terms) { std::string x = "("; for (auto& kv : term.m) x += STR(kv.first) + ","; terms.push_back(x + ")"); }
    std::sort(terms.begin(), terms.end()); for (auto& x : terms) r += x;
  }
  return r;
}

// ------------------------------------------------------------------------------------------------
// Topology (topology.go)
// ------------------------------------------------------------------------------------------------
struct ClusterView {   // what the kube client / state.Cluster would answer
  const ksp::Problem* pr; std::map<std::string, const ksp::StateNode*> node_by_name;
};

struct Topology {
  bool inert = false;                                        // &scheduling.Topology{} of the reference benchmark
  const Ctx* cx = nullptr; const ClusterView* cv = nullptr; Hostnames* hn = nullptr; Stats* st = nullptr;
  std::vector<std::unique_ptr<TopologyGroup>> topologies, inverse;      // creation order == canonical iteration order
  std::map<std::string, int> topo_by_id, inverse_by_id;
  std::map<Sym, std::set<std::string>> domains;              // universe by topology key (provisioner.go:267-276)
  std::set<std::string> excluded;                            // UIDs of the batch (topology.go:66-70)

  TopologyGroup* new_group(TopoType type, const std::string& key, const ksp::Pod& pod, const std::set<std::string>& ns, const ksp::Selector& sel, int32_t max_skew) {   // NewTopologyGroup :64-86
    auto* g = new TopologyGroup(); g->type = type; g->key = S(key);   // NB: topology keys are NOT normalised by the reference (only requirement keys are)
    g->namespaces = ns; g->selector = sel; g->max_skew = max_skew; g->is_hostname = (key == ksp::kHostname);
    if (type == kSpread) g->filter = make_node_filter(pod);
    auto it = domains.find(g->key);
    if (it != domains.end()) for (auto& d : it->second) g->domains[hn->of(S(d))] = 0;
    return g;
  }
  std::set<std::string> namespace_list(const std::string& ns, const std::vector<std::string>& namespaces) {   // buildNamespaceList :324-347 (selector pre-resolved)
    if (namespaces.empty()) return {ns};
    return std::set<std::string>(namespaces.begin(), namespaces.end());
  }
  void tg_record(TopologyGroup& g, const DomKey& d) { g.domains[d]++; }                                   // Record :101-105
  void tg_register(TopologyGroup& g, const DomKey& d) { if (!g.domains.count(d)) g.domains[d] = 0; }      // Register :114-120

  // updateInverseAntiAffinity :202-227   (node_labels == nullptr for batch pods)
  void update_inverse_anti(const std::string& uid, const std::string& ns, const std::vector<ksp::AffinityTerm>& terms, const ksp::Pod& podForFilter, const StrMap* node_labels) {
    for (auto& term : terms) {
      auto nss = namespace_list(ns, term.namespaces);
      std::unique_ptr<TopologyGroup> tg(new_group(kAntiAffinity, term.topology_key, podForFilter, nss, term.selector, INT32_MAX));
      std::string id = group_identity(*tg); TopologyGroup* g;
      auto it = inverse_by_id.find(id);
      if (it == inverse_by_id.end()) { inverse_by_id[id] = (int)inverse.size(); inverse.push_back(std::move(tg)); g = inverse.back().get(); } else g = inverse[it->second].get();
      if (node_labels) { auto lt = node_labels->find(STR(g->key)); if (lt != node_labels->end()) tg_record(*g, hn->of(S(lt->second))); }
      g->owners.insert(uid);
    }
  }
  // updateInverseAffinities :181-199
  void update_inverse_affinities() {
    for (auto& cp : cv->pr->cluster_pods) {
      if (cp.anti_required.empty()) continue;
      if (excluded.count(cp.uid)) continue;
      auto nit = cv->node_by_name.find(cp.node_name); if (nit == cv->node_by_name.end()) continue;
      ksp::Pod dummy; dummy.ns = cp.ns;
      update_inverse_anti(cp.uid, cp.ns, cp.anti_required, dummy, &nit->second->labels);
    }
  }
  // countDomains :231-276
  void count_domains(TopologyGroup& tg) {
    for (auto& cp : cv->pr->cluster_pods) {
      if (!tg.namespaces.count(cp.ns)) continue;
      if (!selector_matches_list(tg.selector, cp.labels)) continue;
      if (excluded.count(cp.uid)) continue;
      auto nit = cv->node_by_name.find(cp.node_name); if (nit == cv->node_by_name.end()) continue;   // Get(node) error path not modelled
      const ksp::StateNode& node = *nit->second;
      auto lt = node.labels.find(STR(tg.key)); bool ok = lt != node.labels.end(); std::string domain = ok ? lt->second : "";
      if (!ok && tg.is_hostname) { domain = node.name; ok = true; }
      if (!ok) continue;
      if (!filter_matches_reqs(*cx, tg.filter, reqs_from_labels(node.labels))) continue;   // nodeFilter.Matches(node) :53-55
      tg_record(tg, hn->of(S(domain)));
    }
  }
  // TopologyListOptions :366-386: nil selector lists EVERYTHING in the namespace (labels.Everything()),
  // unlike selects() where nil matches nothing.
  static bool selector_matches_list(const ksp::Selector& sel, const StrMap& labels) {
    if (sel.nil) return true;
    return selector_matches(sel, labels);
  }
  // Update :86-117
  void update(ksp::Pod& p) {
    if (inert) return;
    for (auto& t : topologies) t->owners.erase(p.uid);
    if (!p.anti_required.empty() || !p.anti_preferred.empty()) update_inverse_anti(p.uid, p.ns, p.anti_required, p, nullptr);   // HasPodAntiAffinity
    std::vector<std::unique_ptr<TopologyGroup>> fresh;
    for (auto& cs : p.spread) fresh.emplace_back(new_group(kSpread, cs.key, p, {p.ns}, cs.selector, cs.max_skew));   // newForTopologies :278-284
    // newForAffinities :287-322 (canonical: affinity terms before anti-affinity terms; required before preferred)
    for (auto& t : p.affinity_required) fresh.emplace_back(new_group(kAffinity, t.topology_key, p, namespace_list(p.ns, t.namespaces), t.selector, INT32_MAX));
    for (auto& w : p.affinity_preferred) fresh.emplace_back(new_group(kAffinity, w.term.topology_key, p, namespace_list(p.ns, w.term.namespaces), w.term.selector, INT32_MAX));
    for (auto& t : p.anti_required) fresh.emplace_back(new_group(kAntiAffinity, t.topology_key, p, namespace_list(p.ns, t.namespaces), t.selector, INT32_MAX));
    for (auto& w : p.anti_preferred) fresh.emplace_back(new_group(kAntiAffinity, w.term.topology_key, p, namespace_list(p.ns, w.term.namespaces), w.term.selector, INT32_MAX));
    for (auto& tg : fresh) {
      std::string id = group_identity(*tg); TopologyGroup* g;
      auto it = topo_by_id.find(id);
      if (it == topo_by_id.end()) { count_domains(*tg); topo_by_id[id] = (int)topologies.size(); topologies.push_back(std::move(tg)); g = topologies.back().get(); }
      else g = topologies[it->second].get();
      g->owners.insert(p.uid);
    }
  }
  // Register :170-181
  void register_domain(Sym key, const DomKey& d) {
    if (inert) return;
    for (auto& t : topologies) if (t->key == key) tg_register(*t, d);
    for (auto& t : inverse) if (t->key == key) tg_register(*t, d);
  }
  bool tg_counts(const TopologyGroup& t, const ksp::Pod& pod, const Reqs& reqs) { return tg_selects(t, pod) && filter_matches_reqs(*cx, t.filter, reqs); }   // Counts :109-111

  // Requirement.Values() on a domain requirement -> domain keys (requirement.go:178-180: for a
  // complement set this is the EXCLUDED values -- restated literally).
  // Record :120-143
  void record(const ksp::Pod& p, const Reqs& requirements) {
    if (inert) return;
    for (auto& tc : topologies) {
      if (tg_counts(*tc, p, requirements)) {
        Req domains = requirements.get(tc->key);
        if (tc->type == kAntiAffinity) { for (Sym v : domains.values.v) tg_record(*tc, hn->of(v)); }
        else if (req_len(domains) == 1) tg_record(*tc, hn->of(domains.values.v[0]));
      }
    }
    for (auto& tc : inverse) if (tc->owners.count(p.uid)) { Req d = requirements.get(tc->key); for (Sym v : d.values.v) tg_record(*tc, hn->of(v)); }
  }

  // nextDomainTopologySpread :155-182 / domainMinCount :184-200
  Req next_spread(TopologyGroup& t, const ksp::Pod& pod, const Req& podDomains, const Req& nodeDomains) {
    int32_t min = INT32_MAX;
    if (t.is_hostname) min = 0;
    else for (auto& kv : t.domains) { st->domains_scanned++; if (req_has(podDomains, dom_sym(kv.first))) if (kv.second < min) min = kv.second; }
    bool self = tg_selects(t, pod);
    bool found = false; DomKey minDomain{}; int32_t minCount = INT32_MAX;
    for (auto& kv : t.domains) {
      st->domains_scanned++;
      if (req_has(nodeDomains, dom_sym(kv.first))) {
        int32_t count = kv.second; if (self) count++;
        if ((int64_t)count - (int64_t)min <= (int64_t)t.max_skew && count < minCount) { minDomain = kv.first; minCount = count; found = true; }
      }
    }
    if (!found) {
      if (&t == relax_group) {      // (model check of the kernel's round resolver only: "accepts but for the skew", see Scheduler::solve_spec_v2)
        relax_hit = true; Req r = new_req_sym(podDomains.key, Op::In);
        for (auto& kv : t.domains) if (req_has(nodeDomains, dom_sym(kv.first))) r.values.insert(dom_sym(kv.first));
        return r;
      }
      return new_req_sym(podDomains.key, Op::DoesNotExist);
    }
    Req r = new_req_sym(podDomains.key, Op::In); r.values.insert(dom_sym(minDomain)); return r;
  }
  const TopologyGroup* relax_group = nullptr; bool relax_hit = false;
  // nextDomainAffinity :202-233
  Req next_affinity(TopologyGroup& t, const ksp::Pod& pod, const Req& podDomains, const Req& nodeDomains) {
    Req options = new_req_sym(podDomains.key, Op::DoesNotExist);
    for (auto& kv : t.domains) { st->domains_scanned++; if (req_has(podDomains, dom_sym(kv.first)) && kv.second > 0) options.values.insert(dom_sym(kv.first)); }
    if (req_len(options) == 0 && tg_selects(t, pod)) {
      Req intersected = req_intersection(podDomains, nodeDomains);
      for (auto& kv : t.domains) if (req_has(intersected, dom_sym(kv.first))) { options.values.insert(dom_sym(kv.first)); break; }
      for (auto& kv : t.domains) if (req_has(podDomains, dom_sym(kv.first))) { options.values.insert(dom_sym(kv.first)); break; }
    }
    return options;
  }
  // nextDomainAntiAffinity :235-243
  Req next_anti(TopologyGroup& t, const Req& domains) {
    Req options = new_req_sym(domains.key, Op::DoesNotExist);
    for (auto& kv : t.domains) { st->domains_scanned++; if (req_has(domains, dom_sym(kv.first)) && kv.second == 0) options.values.insert(dom_sym(kv.first)); }
    return options;
  }
  Sym dom_sym(const DomKey& d) const { return d.sym; }
  static std::string placeholder_name(int64_t id) { char b[64]; snprintf(b, sizeof b, "hostname-placeholder-%04lld", (long long)id); return b; }

  // AddRequirements :149-167 ; returns false on "unsatisfiable topology constraint"
  bool add_requirements(const Reqs& podReqs, const Reqs& nodeReqs, const ksp::Pod& p, Reqs* out) {
    Reqs requirements = nodeReqs;
    if (!inert) {
      // getMatchingTopologies :351-364
      std::vector<TopologyGroup*> matching;
      for (auto& tc : topologies) if (tc->owners.count(p.uid)) matching.push_back(tc.get());
      for (auto& tc : inverse) if (tg_counts(*tc, p, nodeReqs)) matching.push_back(tc.get());
      for (auto* t : matching) {
        Req podDomains = podReqs.has(t->key) ? podReqs.get(t->key) : new_req_sym(t->key, Op::Exists);
        Req nodeDomains = nodeReqs.has(t->key) ? nodeReqs.get(t->key) : new_req_sym(t->key, Op::Exists);
        Req domains;
        switch (t->type) {   // Get :88-99
          case kSpread: domains = next_spread(*t, p, podDomains, nodeDomains); break;
          case kAffinity: domains = next_affinity(*t, p, podDomains, nodeDomains); break;
          default: domains = next_anti(*t, podDomains); break;
        }
        if (req_len(domains) == 0) return false;
        requirements.add(domains);
      }
    }
    *out = requirements; return true;
  }
};

// ------------------------------------------------------------------------------------------------
// instance types as the scheduler sees them
// ------------------------------------------------------------------------------------------------
struct IType {
  const ksp::InstanceType* src; int index; Reqs requirements;
  ResList allocatable() const { return res_subtract(src->capacity, src->overhead); }   // types.go:87-89 -- recomputed per call like the reference
};

struct MachineTemplate {   // machinetemplate.go:32-62
  const ksp::Provisioner* prov; Reqs requirements; std::vector<ksp::Taint> taints;
};

// compatible / fits / hasOffering, node.go:143-159
static bool it_compatible(const IType& it, const Reqs& reqs) { return reqs_intersects(it.requirements, reqs); }
static bool it_fits(const IType& it, const ResList& requests) { return res_fits(requests, it.allocatable()); }
static bool it_has_offering(const IType& it, const Reqs& reqs, Sym zoneKey, Sym ctKey) {
  for (auto& o : it.src->offerings) {
    if (!o.available) continue;
    if ((!reqs.has(zoneKey) || req_has(reqs.get(zoneKey), S(o.zone))) && (!reqs.has(ctKey) || req_has(reqs.get(ctKey), S(o.capacity_type)))) return true;
  }
  return false;
}

// ------------------------------------------------------------------------------------------------
// Go's sort.Slice (go 1.19+: pdqsort_func, $GOROOT/src/sort/zsortfunc.go + sort.go), restated from SURVEY App. C.1.
// UNVERIFIED against a Go toolchain (none in this image): it exists to QUANTIFY how far the reference's unstable sort of the open nodes
// (scheduler.go:183) can move results away from the canonical stable order, not to define parity.  `less(i, j)` / `swap(i, j)` act on positions.
// ------------------------------------------------------------------------------------------------
template <class Less, class Swap> struct GoSort {
  Less less; Swap swp;
  static int bits_len(unsigned long long x) { int n = 0; while (x) { ++n; x >>= 1; } return n; }
  void insertion(int a, int b) { for (int i = a + 1; i < b; ++i) for (int j = i; j > a && less(j, j - 1); --j) swp(j, j - 1); }
  void sift_down(int lo, int hi, int first) { int root = lo; for (;;) { int child = 2 * root + 1; if (child >= hi) return; if (child + 1 < hi && less(first + child, first + child + 1)) ++child; if (!less(first + root, first + child)) return; swp(first + root, first + child); root = child; } }
  void heap(int a, int b) { int first = a, lo = 0, hi = b - a; for (int i = (hi - 1) / 2; i >= 0; --i) sift_down(i, hi, first); for (int i = hi - 1; i >= 0; --i) { swp(first, first + i); sift_down(lo, i, first); } }
  void order2(int& a, int& b, int& swaps) { if (less(b, a)) { ++swaps; std::swap(a, b); } }
  int median(int a, int b, int c, int& swaps) { order2(a, b, swaps); order2(b, c, swaps); order2(a, b, swaps); return b; }
  void choose_pivot(int a, int b, int& pivot, int& hint) {      // hint: 0 unknown, 1 increasing, 2 decreasing
    const int l = b - a; int swaps = 0, i = a + l / 4 * 1, j = a + l / 4 * 2, k = a + l / 4 * 3;
    if (l >= 8) { if (l >= 50) { i = median(i - 1, i, i + 1, swaps); j = median(j - 1, j, j + 1, swaps); k = median(k - 1, k, k + 1, swaps); } j = median(i, j, k, swaps); }
    pivot = j; hint = swaps == 0 ? 1 : (swaps == 12 ? 2 : 0);
  }
  void reverse_range(int a, int b) { int i = a, j = b - 1; while (i < j) { swp(i, j); ++i; --j; } }
  bool partial_insertion(int a, int b) {
    int i = a + 1;
    for (int step = 0; step < 5; ++step) {
      while (i < b && !less(i, i - 1)) ++i;
      if (i == b) return true;
      if (b - a < 50) return false;
      swp(i, i - 1);
      if (i - a >= 2) for (int j = i - 1; j >= 1; --j) { if (!less(j, j - 1)) break; swp(j, j - 1); }
      if (b - i >= 2) for (int j = i + 1; j < b; ++j) { if (!less(j, j - 1)) break; swp(j, j - 1); }
    }
    return false;
  }
  void break_patterns(int a, int b) {
    const int l = b - a;
    if (l >= 8) {
      unsigned long long r = (unsigned long long)l; const unsigned long long mod = 1ull << bits_len((unsigned long long)l); const int idx = a + (l / 4) * 2 - 1;
      for (int t = 0; t < 3; ++t) { r ^= r << 13; r ^= r >> 17; r ^= r << 5; int other = (int)(r & (mod - 1)); if (other >= l) other -= l; swp(idx - 1 + t, a + other); }
    }
  }
  int partition_equal(int a, int b, int pivot) { swp(a, pivot); int i = a + 1, j = b - 1; for (;;) { while (i <= j && !less(a, i)) ++i; while (i <= j && less(a, j)) --j; if (i > j) break; swp(i, j); ++i; --j; } return i; }
  int partition(int a, int b, int pivot, bool& already) {
    swp(a, pivot); int i = a + 1, j = b - 1;
    while (i <= j && less(i, a)) ++i;
    while (i <= j && !less(j, a)) --j;
    if (i > j) { swp(j, a); already = true; return j; }
    swp(i, j); ++i; --j;
    for (;;) { while (i <= j && less(i, a)) ++i; while (i <= j && !less(j, a)) --j; if (i > j) break; swp(i, j); ++i; --j; }
    swp(j, a); already = false; return j;
  }
  void pdq(int a, int b, int limit) {
    bool was_balanced = true, was_partitioned = true;
    for (;;) {
      const int len = b - a;
      if (len <= 12) { insertion(a, b); return; }
      if (limit == 0) { heap(a, b); return; }
      if (!was_balanced) { break_patterns(a, b); --limit; }
      int pivot, hint; choose_pivot(a, b, pivot, hint);
      if (hint == 2) { reverse_range(a, b); pivot = (b - 1) - (pivot - a); hint = 1; }
      if (was_balanced && was_partitioned && hint == 1) if (partial_insertion(a, b)) return;
      if (a > 0 && !less(a - 1, pivot)) { a = partition_equal(a, b, pivot); continue; }
      bool already; const int mid = partition(a, b, pivot, already); was_partitioned = already;
      const int left = mid - a, right = b - mid, thr = len / 8;
      if (left < right) { was_balanced = left >= thr; pdq(a, mid, limit); a = mid + 1; }
      else { was_balanced = right >= thr; pdq(mid + 1, b, limit); b = mid; }
    }
  }
  void sort(int n) { pdq(0, n, bits_len((unsigned long long)n)); }
};
template <class Less, class Swap> static void go_sort_slice(int n, Less less, Swap swp) { GoSort<Less, Swap> g{less, swp}; g.sort(n); }

struct PodState { ksp::Pod spec; int index; int stage = 0; uint32_t reasons = 0; };   // reasons: why the last add() failed, 4 bits per template (scheduler.go:193-217)

struct Node {   // scheduling.Node, node.go:34-107
  const MachineTemplate* tmpl; Reqs requirements; std::vector<const IType*> options; ResList requests;
  std::vector<int> pods; HostPortUsage ports; int64_t placeholder_id; size_t seq;
};
struct ExistingNode {   // existingnode.go:28-130
  const ksp::StateNode* sn; Reqs requirements; ResList requests, available; std::vector<ksp::Taint> taints;
  std::vector<int> pods; HostPortUsage ports;
  std::map<std::string, std::set<std::string>> volumes;   // VolumeUsage.volumes: driver -> claim ids (volumeusage.go:33-40)
  std::map<std::string, int> volume_limits;                // VolumeCount from the CSINode (cluster.go:292-304); absent = unlimited
};

struct Scheduler {
  Ctx cx; ClusterView cv; Hostnames hn; Stats st; Topology topo;
  std::vector<IType> itypes; std::vector<MachineTemplate> templates;   // weight order
  std::map<std::string, std::vector<const IType*>> instance_types;     // provisioner -> types
  std::map<const MachineTemplate*, ResList> daemon_overhead;
  std::map<std::string, ResList> remaining;                            // remainingResources (provisioner name -> list)
  std::vector<std::unique_ptr<Node>> new_nodes; std::vector<std::unique_ptr<ExistingNode>> existing;
  std::vector<PodState> pods; bool tolerate_prefer_no_schedule = false;
  int64_t node_id = 0; Sym zoneKey, ctKey, hostnameKey;
  std::vector<int> unscheduled;
  bool gosort = false;      // order the open nodes with the restated Go sort.Slice instead of the canonical stable sort (quantifies the unpinned region)

  // ---- Preferences.Relax, preferences.go:36-145 ----
  bool relax(ksp::Pod& pod) {
    // removeRequiredNodeAffinityTerm :75-89
    if (pod.required_affinity.size() > 1) { pod.required_affinity.erase(pod.required_affinity.begin()); return true; }
    // removePreferredPodAffinityTerm :103-116 (SliceStable by weight desc, drop the first)
    if (!pod.affinity_preferred.empty()) { std::stable_sort(pod.affinity_preferred.begin(), pod.affinity_preferred.end(), [](const ksp::WeightedTerm& a, const ksp::WeightedTerm& b) { return a.weight > b.weight; }); pod.affinity_preferred.erase(pod.affinity_preferred.begin()); return true; }
    // removePreferredPodAntiAffinityTerm :118-131
    if (!pod.anti_preferred.empty()) { std::stable_sort(pod.anti_preferred.begin(), pod.anti_preferred.end(), [](const ksp::WeightedTerm& a, const ksp::WeightedTerm& b) { return a.weight > b.weight; }); pod.anti_preferred.erase(pod.anti_preferred.begin()); return true; }
    // removePreferredNodeAffinityTerm :58-73
    if (!pod.preferred_affinity.empty()) { std::stable_sort(pod.preferred_affinity.begin(), pod.preferred_affinity.end(), [](const ksp::PreferredTerm& a, const ksp::PreferredTerm& b) { return a.weight > b.weight; }); pod.preferred_affinity.erase(pod.preferred_affinity.begin()); return true; }
    // removeTopologySpreadScheduleAnyway :91-101 (swap-with-last delete of the first ScheduleAnyway)
    for (size_t i = 0; i < pod.spread.size(); ++i) if (pod.spread[i].schedule_anyway) { pod.spread[i] = pod.spread.back(); pod.spread.pop_back(); return true; }
    // toleratePreferNoScheduleTaints :133-145
    if (tolerate_prefer_no_schedule) {
      for (auto& t : pod.tolerations) if (t.key.empty() && t.effect == "PreferNoSchedule" && t.op == "Exists" && t.value.empty()) return false;   // MatchToleration
      pod.tolerations.push_back({"", "Exists", "", "PreferNoSchedule"}); return true;
    }
    return false;
  }

  // ---- filterInstanceTypesByRequirements, node.go:137-141 ----
  std::vector<const IType*> filter_types(const std::vector<const IType*>& in, const Reqs& reqs, const ResList& requests) {
    std::vector<const IType*> out;
    for (auto* it : in) { st.types_scanned++; if (it_compatible(*it, reqs) && it_fits(*it, requests) && it_has_offering(*it, reqs, zoneKey, ctKey)) out.push_back(it); }
    return out;
  }

  // ---- Node.Add, node.go:62-107 ----
  // which step of Node.Add refused the pod (node.go:62-107), for the per-pod failure reasons the boundary reports:
  // 2 taints, 3 host ports, 4 incompatible requirements, 5 unsatisfiable topology, 6 topology requirements incompatible, 7 no instance type
  int fail_step = 0;
  bool node_add(Node& m, PodState& ps) {
    st.attempts++;
    ksp::Pod& pod = ps.spec;
    if (!taints_tolerates(m.tmpl->taints, pod)) { fail_step = 2; return false; }
    if (!m.ports.validate(pod, nullptr)) { fail_step = 3; return false; }
    Reqs nodeReqs = m.requirements;
    Reqs podReqs = new_pod_requirements(pod);
    if (!reqs_compatible(cx, nodeReqs, podReqs)) { fail_step = 4; return false; }
    nodeReqs.add_all(podReqs);
    Reqs topoReqs;
    if (!topo.add_requirements(podReqs, nodeReqs, pod, &topoReqs)) { fail_step = 5; return false; }
    if (!reqs_compatible(cx, nodeReqs, topoReqs)) { fail_step = 6; return false; }
    nodeReqs.add_all(topoReqs);
    ResList requests = res_merge(m.requests, requests_for_pods({&pod}));
    auto its = filter_types(m.options, nodeReqs, requests);
    if (its.empty()) { fail_step = 7; return false; }
    m.pods.push_back(ps.index); m.options = its; m.requests = requests; m.requirements = nodeReqs;
    topo.record(pod, nodeReqs); m.ports.add(pod);
    return true;
  }
  // ---- ExistingNode.Add, existingnode.go:77-130 (incl. the volume limits of :87-94; PVC -> driver lookups arrive resolved) ----
  bool existing_add(ExistingNode& n, PodState& ps) {
    st.attempts++;
    ksp::Pod& pod = ps.spec;
    if (!taints_tolerates(n.taints, pod)) return false;
    if (!n.ports.validate(pod, nullptr)) return false;
    // volumeUsage.Validate + VolumeCount.Exceeds, existingnode.go:87-94 / volumeusage.go:102-143
    if (pod.volume_error) return false;                                     // a lookup failed: Validate returns the error
    {
      std::map<std::string, std::set<std::string>> u = n.volumes;           // volumes.union(podVolumes)
      for (auto& v : pod.volumes) u[v.driver].insert(v.pvc);
      for (auto& kv : u) { auto lim = n.volume_limits.find(kv.first); if (lim != n.volume_limits.end() && (int)kv.second.size() > lim->second) return false; }   // "would exceed node volume limits"
    }
    ResList requests = res_merge(n.requests, requests_for_pods({&pod}));
    if (!res_fits(requests, n.available)) return false;
    Reqs nodeReqs = n.requirements;
    Reqs podReqs = new_pod_requirements(pod);
    if (!reqs_compatible(cx, nodeReqs, podReqs)) return false;
    nodeReqs.add_all(podReqs);
    Reqs topoReqs;
    if (!topo.add_requirements(podReqs, nodeReqs, pod, &topoReqs)) return false;
    if (!reqs_compatible(cx, nodeReqs, topoReqs)) return false;
    nodeReqs.add_all(topoReqs);
    n.pods.push_back(ps.index); n.requests = requests; n.requirements = nodeReqs;
    topo.record(pod, nodeReqs); n.ports.add(pod);
    for (auto& v : pod.volumes) n.volumes[v.driver].insert(v.pvc);          // volumeUsage.Add, volumeusage.go:94-100
    return true;
  }

  // ---- NewNode, node.go:44-60 ----
  std::unique_ptr<Node> new_node(const MachineTemplate& t, const ResList& daemon, const std::vector<const IType*>& its) {
    auto n = std::make_unique<Node>();
    n->placeholder_id = ++node_id;
    std::string hostname = Topology::placeholder_name(n->placeholder_id);
    hn.keys[S(hostname)] = DomKey{1, "", n->placeholder_id, S(hostname)};
    topo.register_domain(hostnameKey, hn.of(S(hostname)));
    n->tmpl = &t; n->requirements.add_all(t.requirements);
    n->requirements.add(new_req(ksp::kHostname, Op::In, {hostname}));
    n->options = its; n->requests = daemon;
    return n;
  }

  // ---- filterByRemainingResources :293-309 / subtractMax :273-290 ----
  static std::vector<const IType*> filter_by_remaining(const std::vector<const IType*>& its, const ResList& remaining) {
    std::vector<const IType*> out;
    for (auto* it : its) {
      bool viable = true;
      for (auto& kv : remaining) { auto c = it->src->capacity.find(kv.first); int64_t cap = c == it->src->capacity.end() ? 0 : c->second; if (cap > kv.second) viable = false; }
      if (viable) out.push_back(it);
    }
    return out;
  }
  static ResList subtract_max(const ResList& remaining, const std::vector<const IType*>& its) {
    if (its.empty()) return remaining;
    ResList mx; for (auto* it : its) mx = res_max(mx, it->src->capacity);
    ResList result; for (auto& kv : remaining) { auto m = mx.find(kv.first); result[kv.first] = kv.second - (m == mx.end() ? 0 : m->second); }
    return result;
  }

  // ---- Scheduler.add, scheduler.go:174-219 ----
  FILE* trace = getenv("KO_TRACE") ? fopen(getenv("KO_TRACE"), "w") : nullptr;   // debugging aid: per add() the winner's position in the visiting order
  bool add(PodState& ps) {
    for (auto& n : existing) if (existing_add(*n, ps)) { ps.reasons = 0; return true; }
    // sort.Slice(newNodes, len(Pods) asc) -- canonical: stable
    if (gosort) go_sort_slice((int)new_nodes.size(), [&](int i, int j) { return new_nodes[i]->pods.size() < new_nodes[j]->pods.size(); }, [&](int i, int j) { std::swap(new_nodes[i], new_nodes[j]); });
    else std::stable_sort(new_nodes.begin(), new_nodes.end(), [](const std::unique_ptr<Node>& a, const std::unique_ptr<Node>& b) { return a->pods.size() < b->pods.size(); });
    { int pos = 0; for (auto& n : new_nodes) { if (node_add(*n, ps)) { ps.reasons = 0; if (trace) fprintf(trace, "%d %d %zu %zu %d\n", ps.index, pos, new_nodes.size(), n->pods.size(), (int)n->seq); return true; } ++pos; } }
    if (trace) fprintf(trace, "%d -1 %zu 0 -1\n", ps.index, new_nodes.size());
    ps.reasons = 0; uint32_t ti = 0;
    for (auto& t : templates) {
      const uint32_t shift = 4 * ti++;
      std::vector<const IType*> its = instance_types[t.prov->name];
      auto rem = remaining.find(t.prov->name);
      if (rem != remaining.end()) { its = filter_by_remaining(instance_types[t.prov->name], rem->second); if (its.empty()) { if (shift < 32) ps.reasons |= 1u << shift; continue; } }   // "all available instance types exceed provisioner limits"
      auto node = new_node(t, daemon_overhead[&t], its);
      if (!node_add(*node, ps)) { if (shift < 32) ps.reasons |= (uint32_t)fail_step << shift; continue; }                                        // "incompatible with provisioner ..., <step>"
      ps.reasons = 0;
      node->seq = new_nodes.size();
      new_nodes.push_back(std::move(node));
      // NB scheduler.go:215 assigns remainingResources[name] even when the provisioner has no limits
      // (subtractMax of a nil list is an empty list); an empty list filters nothing, so it is inert.
      remaining[t.prov->name] = subtract_max(rem != remaining.end() ? rem->second : ResList{}, new_nodes.back()->options);
      return true;
    }
    return false;
  }

  // ---- Queue, queue.go:29-110 ; Solve, scheduler.go:96-133 ----
  void solve() {
    std::vector<int> q(pods.size()); for (size_t i = 0; i < pods.size(); ++i) q[i] = (int)i;
    std::vector<ResList> rq(pods.size()); for (size_t i = 0; i < pods.size(); ++i) rq[i] = requests_for_pods({&pods[i].spec});
    auto get = [](const ResList& r, const char* k) { auto it = r.find(k); return it == r.end() ? (int64_t)0 : it->second; };
    // byCPUAndMemoryDescending :74-110 -- a strict total order when UIDs are unique (required by the input contract)
    std::sort(q.begin(), q.end(), [&](int a, int b) {
      int64_t ca = get(rq[a], "cpu"), cb = get(rq[b], "cpu"); if (ca != cb) return ca > cb;
      int64_t ma = get(rq[a], "memory"), mb = get(rq[b], "memory"); if (ma != mb) return ma > mb;
      if (pods[a].spec.creation_ts != pods[b].spec.creation_ts) return pods[a].spec.creation_ts < pods[b].spec.creation_ts;
      return pods[a].spec.uid < pods[b].spec.uid;
    });
    std::deque<int> queue(q.begin(), q.end());
    std::unordered_map<int, size_t> lastLen;   // keyed by pod (UIDs unique)
    for (;;) {
      if (queue.empty()) break;
      int pi = queue.front();
      auto ll = lastLen.find(pi);
      if (ll != lastLen.end() && ll->second == queue.size()) break;   // Pop :44-58 (absent key reads 0, never equals a non-empty length)
      queue.pop_front(); st.queue_pops++;
      PodState& ps = pods[pi];
      if (add(ps)) continue;
      bool relaxed = relax(ps.spec);
      queue.push_back(pi);                                           // Push :61-68
      if (relaxed) { lastLen.clear(); ps.stage++; st.relaxations++; topo.update(ps.spec); } else lastLen[pi] = queue.size();
    }
    unscheduled.assign(queue.begin(), queue.end());
    for (auto& n : new_nodes) n->requirements.m.erase(hostnameKey);   // FinalizeScheduling, node.go:111-115
  }
  // ------------------------------------------------------------------------------------------------
  // Round-speculation model check (tests only).  The HIP kernel evaluates up to W queued pods against ONE
  // snapshot of the nodes (the first 64 candidates in visiting order), then a resolver picks each pod's node
  // from the fit bitmaps with the rules below, without re-evaluating.  This restates those rules on top of the
  // sequential algorithm and checks every prediction against what add() then really does.
  //   out[0] pods placed through a prediction, out[1] rounds, out[2] pods handled sequentially,
  //   out[3] rule violations (must be 0), out[4] predictions cut by the topology rule, out[5] cut by the order rule
  // ------------------------------------------------------------------------------------------------
  bool dry_new(const Node& m, PodState& ps) {
    ksp::Pod& pod = ps.spec;
    if (!taints_tolerates(m.tmpl->taints, pod)) return false;
    if (!m.ports.validate(pod, nullptr)) return false;
    Reqs nodeReqs = m.requirements; Reqs podReqs = new_pod_requirements(pod);
    if (!reqs_compatible(cx, nodeReqs, podReqs)) return false;
    nodeReqs.add_all(podReqs);
    Reqs topoReqs;
    if (!topo.add_requirements(podReqs, nodeReqs, pod, &topoReqs)) return false;
    if (!reqs_compatible(cx, nodeReqs, topoReqs)) return false;
    nodeReqs.add_all(topoReqs);
    ResList requests = res_merge(m.requests, requests_for_pods({&pod}));
    return !filter_types(m.options, nodeReqs, requests).empty();
  }
  bool dry_existing(const ExistingNode& n, PodState& ps) {
    ksp::Pod& pod = ps.spec;
    if (!taints_tolerates(n.taints, pod)) return false;
    if (!n.ports.validate(pod, nullptr)) return false;
    // volumeUsage.Validate + VolumeCount.Exceeds, existingnode.go:87-94 / volumeusage.go:102-143
    if (pod.volume_error) return false;                                     // a lookup failed: Validate returns the error
    {
      std::map<std::string, std::set<std::string>> u = n.volumes;           // volumes.union(podVolumes)
      for (auto& v : pod.volumes) u[v.driver].insert(v.pvc);
      for (auto& kv : u) { auto lim = n.volume_limits.find(kv.first); if (lim != n.volume_limits.end() && (int)kv.second.size() > lim->second) return false; }   // "would exceed node volume limits"
    }
    ResList requests = res_merge(n.requests, requests_for_pods({&pod}));
    if (!res_fits(requests, n.available)) return false;
    Reqs nodeReqs = n.requirements; Reqs podReqs = new_pod_requirements(pod);
    if (!reqs_compatible(cx, nodeReqs, podReqs)) return false;
    nodeReqs.add_all(podReqs);
    Reqs topoReqs;
    if (!topo.add_requirements(podReqs, nodeReqs, pod, &topoReqs)) return false;
    return reqs_compatible(cx, nodeReqs, topoReqs);
  }
  // flags bit 0: hostname-keyed anti-affinity groups, and hostname-keyed spread groups that existed before the Solve, never cut a
  //              round through the topology rule (a record only touches the winner's own hostname counter, which no other
  //              candidate's evaluation reads; a taken node is covered by the order rule)
  // maxcls > 0:  a round holds at most `maxcls` distinct evaluation classes (one wave evaluates one class for all its pods)
  // out[6] pods in rounds of >= 8, out[7] largest round
  std::string eval_signature(PodState& ps) {
    ksp::Pod& pod = ps.spec; std::string s;
    Reqs pr = new_pod_requirements(pod);
    for (auto& kv : pr.m) { s += STR(kv.first); s += kv.second.complement ? '!' : '='; for (Sym v : kv.second.values.v) { s += STR(v); s += ','; } s += kv.second.has_gt ? std::to_string(kv.second.gt) : "-"; s += kv.second.has_lt ? std::to_string(kv.second.lt) : "-"; s += ';'; }
    ResList rq = requests_for_pods({&pod}); for (auto& kv : rq) { s += kv.first; s += std::to_string(kv.second); s += ','; }
    for (auto& t : pod.tolerations) { s += t.key + "/" + t.op + "/" + t.value + "/" + t.effect + ";"; }
    for (auto& c : pod.containers) for (auto& hp : c.ports) { s += hp.ip + ":" + std::to_string(hp.port) + hp.proto + ";"; }
    for (auto& tc : topo.topologies) if (tc->owners.count(pod.uid)) { s += "T" + std::to_string((uintptr_t)tc.get()) + (tg_selects(*tc, pod) ? "s" : "n"); }
    for (auto& tc : topo.inverse) if (tg_selects(*tc, pod)) { s += "I" + std::to_string((uintptr_t)tc.get()); }
    return s;
  }
  void solve_spec(int W, long long* out, int flags = 0, int maxcls = 0) {
    std::set<const TopologyGroup*> initial; for (auto& tc : topo.topologies) initial.insert(tc.get());
    std::vector<int> q(pods.size()); for (size_t i = 0; i < pods.size(); ++i) q[i] = (int)i;
    std::vector<ResList> rq(pods.size()); for (size_t i = 0; i < pods.size(); ++i) rq[i] = requests_for_pods({&pods[i].spec});
    auto get = [](const ResList& r, const char* k) { auto it = r.find(k); return it == r.end() ? (int64_t)0 : it->second; };
    std::sort(q.begin(), q.end(), [&](int a, int b) {
      int64_t ca = get(rq[a], "cpu"), cb = get(rq[b], "cpu"); if (ca != cb) return ca > cb;
      int64_t ma = get(rq[a], "memory"), mb = get(rq[b], "memory"); if (ma != mb) return ma > mb;
      if (pods[a].spec.creation_ts != pods[b].spec.creation_ts) return pods[a].spec.creation_ts < pods[b].spec.creation_ts;
      return pods[a].spec.uid < pods[b].spec.uid;
    });
    std::deque<int> queue(q.begin(), q.end());
    std::unordered_map<int, size_t> lastLen;
    for (int i = 0; i < 8; ++i) out[i] = 0;
    auto sequential_step = [&]() -> bool {   // one iteration of solve(); false == stop
      int pi = queue.front();
      auto ll = lastLen.find(pi);
      if (ll != lastLen.end() && ll->second == queue.size()) return false;
      queue.pop_front(); st.queue_pops++;
      PodState& ps = pods[pi];
      if (add(ps)) return true;
      bool relaxed = relax(ps.spec);
      queue.push_back(pi);
      if (relaxed) { lastLen.clear(); ps.stage++; st.relaxations++; topo.update(ps.spec); } else lastLen[pi] = queue.size();
      return true;
    };
    while (!queue.empty()) {
      // ---- snapshot: the first 64 candidates in visiting order ----
      std::stable_sort(new_nodes.begin(), new_nodes.end(), [](const std::unique_ptr<Node>& a, const std::unique_ptr<Node>& b) { return a->pods.size() < b->pods.size(); });
      const size_t E = existing.size(), total = E + new_nodes.size(), nc = std::min<size_t>(64, total);
      std::vector<size_t> cnt(nc, 0); for (size_t i = 0; i < nc; ++i) if (i >= E) cnt[i] = new_nodes[i - E]->pods.size();
      std::vector<Node*> snapN(nc, nullptr); std::vector<ExistingNode*> snapE(nc, nullptr);
      for (size_t i = 0; i < nc; ++i) { if (i < E) snapE[i] = existing[i].get(); else snapN[i] = new_nodes[i - E].get(); }
      // ---- pods of the round: never-requeued queue entries only ----
      size_t n = 0; while (n < (size_t)W && n < queue.size() && lastLen.find(queue[n]) == lastLen.end()) ++n;
      if (maxcls > 0) { std::set<std::string> seen; size_t k = 0; for (; k < n; ++k) { std::string e = eval_signature(pods[queue[k]]); if (!seen.count(e)) { if ((int)seen.size() == maxcls) break; seen.insert(e); } } n = k; }
      std::vector<uint64_t> m(n, 0); std::vector<std::set<const TopologyGroup*>> T(n), R(n);
      Stats keep = st;
      for (size_t k = 0; k < n; ++k) {
        PodState& ps = pods[queue[k]];
        for (size_t i = 0; i < nc; ++i) if (i < E ? dry_existing(*snapE[i], ps) : dry_new(*snapN[i], ps)) m[k] |= 1ull << i;
        if (!topo.inert) {
          for (auto& tc : topo.topologies) {
            const bool own_counter_only = (flags & 1) && tc->is_hostname && (tc->type == kAntiAffinity || (tc->type == kSpread && initial.count(tc.get())));
            if (tc->owners.count(ps.spec.uid) && !own_counter_only) T[k].insert(tc.get());
            if (tg_selects(*tc, ps.spec)) R[k].insert(tc.get());
          }
          for (auto& tc : topo.inverse) { if (tg_selects(*tc, ps.spec)) T[k].insert(tc.get()); if (tc->owners.count(ps.spec.uid)) R[k].insert(tc.get()); }
        }
      }
      st = keep;
      // ---- resolver ----
      std::vector<int> win; uint64_t taken = 0; std::set<const TopologyGroup*> Rall;
      for (size_t k = 0; k < n; ++k) {
        bool topo_hit = false; for (auto* g : T[k]) if (Rall.count(g)) topo_hit = true;
        if (topo_hit) { out[4]++; break; }
        const uint64_t cand = m[k] & ~taken; if (!cand) break;
        const int u = __builtin_ctzll(cand);
        bool ok = true;
        for (int wj : win) if ((m[k] >> wj) & 1ull) {         // an earlier winner that accepted this pod at the snapshot: does it precede u now?
          if ((size_t)wj < E) { if (wj < u) ok = false; }     // existing nodes keep their place
          else if ((size_t)u >= E && cnt[wj] + 1 <= cnt[u]) ok = false;   // it moved to the front of bucket cnt+1
        }
        if (!ok) { out[5]++; break; }
        win.push_back(u); taken |= 1ull << u; for (auto* g : R[k]) Rall.insert(g);
      }
      if (win.empty()) { out[2]++; if (!sequential_step()) break; continue; }
      out[1]++; if (win.size() >= 8) out[6] += (long long)win.size(); if ((long long)win.size() > out[7]) out[7] = (long long)win.size();
      // ---- commit the predictions through the real algorithm and compare ----
      for (size_t k = 0; k < win.size(); ++k) {
        int pi = queue.front(); queue.pop_front(); st.queue_pops++;
        PodState& ps = pods[pi];
        const size_t before_nodes = new_nodes.size();
        bool placed = add(ps);
        bool match = placed && new_nodes.size() == before_nodes;
        if (match) { const int u = win[k]; match = (size_t)u < E ? (!snapE[u]->pods.empty() && snapE[u]->pods.back() == ps.index) : (!snapN[u]->pods.empty() && snapN[u]->pods.back() == ps.index); }
        if (!match) { out[3]++; fprintf(stderr, "speculation rule violated: pod %d (round position %zu) predicted candidate %d\n", pi, k, win[k]); }
        if (!placed) { bool relaxed = relax(ps.spec); queue.push_back(pi); if (relaxed) { lastLen.clear(); ps.stage++; st.relaxations++; topo.update(ps.spec); } else lastLen[pi] = queue.size(); break; }
        out[0]++;
      }
    }
    unscheduled.assign(queue.begin(), queue.end());
    for (auto& n : new_nodes) n->requirements.m.erase(hostnameKey);
  }

  // ------------------------------------------------------------------------------------------------
  // Model check of the kernel's WATERMARK over the existing nodes (ksolve.hip ClsPlan::mono, tests only).  The kernel starts a pod's scan where the last
  // pod of its class stopped, on the claim that an existing node which refused the class once refuses it for the rest of the Solve -- for classes whose
  // own requirements are on well-known keys and whose topology items, if any, are all anti-affinity (own or inverse).  Here every pod of such a class is
  // dry-run against EVERY existing node right before the sequential algorithm places it: a node on record as having refused the class must refuse again.
  //   out[0] violations (must be 0), out[1] dry runs compared, out[2] pods of watermark classes, out[3] of them with anti-affinity items
  //   mutate: also treat classes with spread / affinity items as watermark classes (the claim is false for them: the check must be able to tell)
  // ------------------------------------------------------------------------------------------------
  // the part of ExistingNode.Add before any topology step (existingnode.go:77-103): taints, host ports, volume limits, resources, the pod's own requirements
  bool dry_existing_pre(const ExistingNode& n, PodState& ps) {
    ksp::Pod& pod = ps.spec;
    if (!taints_tolerates(n.taints, pod)) return false;
    if (!n.ports.validate(pod, nullptr)) return false;
    if (pod.volume_error) return false;
    {
      std::map<std::string, std::set<std::string>> u = n.volumes;
      for (auto& v : pod.volumes) u[v.driver].insert(v.pvc);
      for (auto& kv : u) { auto lim = n.volume_limits.find(kv.first); if (lim != n.volume_limits.end() && (int)kv.second.size() > lim->second) return false; }
    }
    ResList requests = res_merge(n.requests, requests_for_pods({&pod}));
    if (!res_fits(requests, n.available)) return false;
    Reqs nodeReqs = n.requirements; Reqs podReqs = new_pod_requirements(pod);
    return reqs_compatible(cx, nodeReqs, podReqs);
  }
  // mode 2 (a lead for the next round): EVERY class whose own requirements are on well-known keys -- spread and affinity classes too -- keeps
  // refusals monotone as long as the refusal happens BEFORE the topology steps.  out[4] counts the pre-topology refusals put on record, out[5] the later dry
  // runs of a recorded (class, node) pair; a recorded node that accepts is a violation (out[0]).
  void solve_watermark_check(long long* out, bool mutate, bool pre_only = false) {
    for (int i = 0; i < 8; ++i) out[i] = 0;
    std::vector<int> q(pods.size()); for (size_t i = 0; i < pods.size(); ++i) q[i] = (int)i;
    std::vector<ResList> rq(pods.size()); for (size_t i = 0; i < pods.size(); ++i) rq[i] = requests_for_pods({&pods[i].spec});
    auto get = [](const ResList& r, const char* k) { auto it = r.find(k); return it == r.end() ? (int64_t)0 : it->second; };
    std::sort(q.begin(), q.end(), [&](int a, int b) {
      int64_t ca = get(rq[a], "cpu"), cb = get(rq[b], "cpu"); if (ca != cb) return ca > cb;
      int64_t ma = get(rq[a], "memory"), mb = get(rq[b], "memory"); if (ma != mb) return ma > mb;
      if (pods[a].spec.creation_ts != pods[b].spec.creation_ts) return pods[a].spec.creation_ts < pods[b].spec.creation_ts;
      return pods[a].spec.uid < pods[b].spec.uid;
    });
    std::deque<int> queue(q.begin(), q.end());
    std::unordered_map<int, size_t> lastLen;
    std::map<std::string, std::vector<uint8_t>> refused;      // class signature -> per existing node: refused it before
    for (;;) {
      if (queue.empty()) break;
      int pi = queue.front();
      auto ll = lastLen.find(pi);
      if (ll != lastLen.end() && ll->second == queue.size()) break;
      queue.pop_front(); st.queue_pops++;
      PodState& ps = pods[pi];
      {
        bool eligible = true, anti = false;
        Reqs pr = new_pod_requirements(ps.spec);
        for (auto& kv : pr.m) if (!cx.well_known.count(kv.first)) eligible = false;
        for (auto& tc : topo.topologies) if (tc->owners.count(ps.spec.uid)) { if (tc->type == kAntiAffinity) anti = true; else if (!mutate && !pre_only) eligible = false; }
        for (auto& tc : topo.inverse) if (tg_selects(*tc, ps.spec)) anti = true;
        if (eligible && !existing.empty()) {
          out[2]++; if (anti) out[3]++;
          auto& rf = refused[eval_signature(ps)]; rf.resize(existing.size(), 0);
          Stats keep = st;
          for (size_t i = 0; i < existing.size(); ++i) {
            const bool ok = dry_existing(*existing[i], ps); out[1]++;
            if (rf[i]) out[5]++;
            if (ok && rf[i]) { out[0]++; if (out[0] <= 5) fprintf(stderr, "watermark rule violated: pod %d accepted by existing node %zu that refused its class before\n", pi, i); }
            if (pre_only ? !dry_existing_pre(*existing[i], ps) : !ok) { if (!rf[i]) out[4]++; rf[i] = 1; }
          }
          st = keep;
        }
      }
      if (add(ps)) continue;
      bool relaxed = relax(ps.spec);
      queue.push_back(pi);
      if (relaxed) { lastLen.clear(); ps.stage++; st.relaxations++; topo.update(ps.spec); } else lastLen[pi] = queue.size();
    }
    unscheduled.assign(queue.begin(), queue.end());
    for (auto& n : new_nodes) n->requirements.m.erase(hostnameKey);
  }

  // ------------------------------------------------------------------------------------------------
  // Model check of the round resolver as the kernel runs it since round 2 (ksolve.hip "P2: the leader resolves the round"): candidates that
  // already took pods of the round stay in play (exact resources with what the round put on them, hostname-keyed items followed through the
  // round's certain records: exact reject / slack), runs of equivalent pods (SWEEP / CLIMB), zonal spread followed exactly on pinned nodes
  // against counts kept for the round (dynamic spread: dd / rdyn / unknown candidates), closed candidates, the window-incomplete rule.
  // The RULES are restated on the reference-shaped state; every prediction is then committed through the real add() and compared.
  //   flags bit 2 (mutation, tests only): leave out the rdyn rule -- the model must then report violations (it would have caught that bug).
  //   out[] as solve_spec.
  // ------------------------------------------------------------------------------------------------
  struct Dry { bool ok = false, skew_failed = false, changes = false; };
  static bool req_equal(const Req& a, const Req& b) { return a.complement == b.complement && a.values == b.values && a.has_gt == b.has_gt && a.has_lt == b.has_lt && (!a.has_gt || a.gt == b.gt) && (!a.has_lt || a.lt == b.lt); }
  static bool reqs_equal(const Reqs& a, const Reqs& b) { if (a.m.size() != b.m.size()) return false; auto x = a.m.begin(); auto y = b.m.begin(); for (; x != a.m.end(); ++x, ++y) if (x->first != y->first || !req_equal(x->second, y->second)) return false; return true; }
  Dry dry_v2(Node* m, ExistingNode* e, PodState& ps, const ResList& extra, const TopologyGroup* relax) {
    Dry d; ksp::Pod& pod = ps.spec; const Stats keep = st;
    topo.relax_group = relax; topo.relax_hit = false;
    auto done = [&](bool ok, bool changes) { d.ok = ok; d.changes = changes; d.skew_failed = topo.relax_hit; topo.relax_group = nullptr; st = keep; return d; };
    const std::vector<ksp::Taint>& taints = m ? m->tmpl->taints : e->taints;
    if (!taints_tolerates(taints, pod)) return done(false, false);
    if (!(m ? m->ports.validate(pod, nullptr) : e->ports.validate(pod, nullptr))) return done(false, false);
    const Reqs& base = m ? m->requirements : e->requirements;
    ResList requests = res_merge(res_merge(m ? m->requests : e->requests, extra), requests_for_pods({&pod}));
    if (e) { if (pod.volume_error || !pod.volumes.empty()) return done(false, false); if (!res_fits(requests, e->available)) return done(false, false); }
    Reqs nodeReqs = base; Reqs podReqs = new_pod_requirements(pod);
    if (!reqs_compatible(cx, nodeReqs, podReqs)) return done(false, false);
    nodeReqs.add_all(podReqs);
    Reqs topoReqs;
    if (!topo.add_requirements(podReqs, nodeReqs, pod, &topoReqs)) return done(false, false);
    if (!reqs_compatible(cx, nodeReqs, topoReqs)) return done(false, false);
    nodeReqs.add_all(topoReqs);
    if (m && filter_types(m->options, nodeReqs, requests).empty()) return done(false, false);
    return done(true, !reqs_equal(nodeReqs, base));
  }
  void solve_spec_v2(int W, long long* out, int flags = 0, int maxcls = 0) {
    typedef const TopologyGroup* G; typedef std::set<G> GS;
    auto meets = [](const GS& a, const GS& b) { for (G g : a) if (b.count(g)) return true; return false; };
    auto trivial_filter = [](const TopologyGroup& t) { if (t.filter.always || t.filter.terms.empty()) return true; for (auto& term : t.filter.terms) if (term.m.empty()) return true; return false; };
    const bool no_rdyn = (flags & 4) != 0;
    GS initial; for (auto& tc : topo.topologies) initial.insert(tc.get());
    // groups the resolver follows exactly: unfiltered spread groups present from the start on ONE narrow key (the one with the most of them)
    GS dyn_groups;
    { std::map<Sym, std::vector<G>> per; for (auto& tc : topo.topologies) if (tc->type == kSpread && !tc->is_hostname && trivial_filter(*tc) && tc->domains.size() <= 8) per[tc->key].push_back(tc.get());
      const std::vector<G>* best = nullptr; for (auto& kv : per) if (!best || kv.second.size() > best->size()) best = &kv.second;
      if (best) for (size_t i = 0; i < best->size() && i < 16; ++i) dyn_groups.insert((*best)[i]); }
    const Sym dyn_key = dyn_groups.empty() ? (Sym)-1 : (*dyn_groups.begin())->key;
    std::vector<int> q(pods.size()); for (size_t i = 0; i < pods.size(); ++i) q[i] = (int)i;
    std::vector<ResList> rq(pods.size()); for (size_t i = 0; i < pods.size(); ++i) rq[i] = requests_for_pods({&pods[i].spec});
    auto get = [](const ResList& r, const char* k) { auto it = r.find(k); return it == r.end() ? (int64_t)0 : it->second; };
    std::sort(q.begin(), q.end(), [&](int a, int b) {
      int64_t ca = get(rq[a], "cpu"), cb = get(rq[b], "cpu"); if (ca != cb) return ca > cb;
      int64_t ma = get(rq[a], "memory"), mb = get(rq[b], "memory"); if (ma != mb) return ma > mb;
      if (pods[a].spec.creation_ts != pods[b].spec.creation_ts) return pods[a].spec.creation_ts < pods[b].spec.creation_ts;
      return pods[a].spec.uid < pods[b].spec.uid;
    });
    std::deque<int> queue(q.begin(), q.end());
    std::unordered_map<int, size_t> lastLen;
    for (int i = 0; i < 8; ++i) out[i] = 0;
    auto sequential_step = [&]() -> bool {
      int pi = queue.front();
      auto ll = lastLen.find(pi);
      if (ll != lastLen.end() && ll->second == queue.size()) return false;
      queue.pop_front(); st.queue_pops++;
      PodState& ps = pods[pi];
      if (add(ps)) return true;
      bool relaxed = relax(ps.spec);
      queue.push_back(pi);
      if (relaxed) { lastLen.clear(); ps.stage++; st.relaxations++; topo.update(ps.spec); } else lastLen[pi] = queue.size();
      return true;
    };
    struct Info { std::vector<TopologyGroup*> narrow, host; GS tmask, tfull, rmask, rsure, zmask; bool eligible = true; int dyn = 0; TopologyGroup* dg = nullptr; bool self = false; std::string ev; ResList req; };
    struct Cand { Node* n = nullptr; ExistingNode* e = nullptr; size_t cnt0 = 0, cnt = 0; bool moved = false, closed = false; int np = 0, last = -1; ResList extra; GS racc, rsure, unsure; std::map<G, int> hrec; bool pinned = false; DomKey zone; uint64_t key = 0; };
    while (!queue.empty()) {
      std::stable_sort(new_nodes.begin(), new_nodes.end(), [](const std::unique_ptr<Node>& a, const std::unique_ptr<Node>& b) { return a->pods.size() < b->pods.size(); });
      const size_t E = existing.size(), total = E + new_nodes.size(), nc = std::min<size_t>(64, total);
      const bool window_complete = total <= 64;
      std::vector<Cand> C(nc);
      for (size_t i = 0; i < nc; ++i) {
        Cand& c = C[i]; if (i < E) c.e = existing[i].get(); else { c.n = new_nodes[i - E].get(); c.cnt0 = c.cnt = c.n->pods.size(); }
        c.key = i < E ? (uint64_t)i : (((uint64_t)c.cnt << 8) | (64u + (uint64_t)i));
        if (dyn_key != (Sym)-1) { const Reqs& rq2 = c.n ? c.n->requirements : c.e->requirements; if (rq2.has(dyn_key)) { const Req& z = rq2.m.at(dyn_key); if (!z.complement && z.values.size() == 1) { c.pinned = true; c.zone = hn.of(z.values.v[0]); } } }
      }
      const size_t cnt_last = nc ? C[nc - 1].cnt : 0;
      // ---- the round's pods ----
      size_t n = 0; std::vector<Info> I;
      { std::set<std::string> seen;
        while (n < (size_t)W && n < queue.size() && lastLen.find(queue[n]) == lastLen.end()) {
          PodState& ps = pods[queue[n]]; ksp::Pod& pod = ps.spec; Info in; in.ev = eval_signature(ps); in.req = requests_for_pods({&pod});
          for (auto& c : pod.containers) if (!c.ports.empty()) in.eligible = false;
          if (pod.volume_error || !pod.volumes.empty()) in.eligible = false;
          if (!topo.inert) {
            Reqs podReqs = new_pod_requirements(pod);
            for (auto& tc : topo.topologies) if (tc->owners.count(pod.uid)) { (tc->is_hostname ? in.host : in.narrow).push_back(tc.get()); }
            for (auto& tc : topo.inverse) if (tg_selects(*tc, pod)) { (tc->is_hostname ? in.host : in.narrow).push_back(tc.get()); }
            for (auto* g : in.narrow) { in.tmask.insert(g); in.tfull.insert(g); }
            for (auto* g : in.host) {
              in.tfull.insert(g); const bool self = tg_selects(*g, pod);
              const bool own_counter_only = g->type == kAntiAffinity || (g->type == kSpread && initial.count(g));
              if (!own_counter_only) in.tmask.insert(g);
              if (g->type == kAntiAffinity || (g->type == kSpread && (int64_t)g->max_skew - (self ? 1 : 0) <= 0)) in.zmask.insert(g);
            }
            for (auto& tc : topo.topologies) if (tg_selects(*tc, pod)) {
              in.rmask.insert(tc.get());
              if (tc->is_hostname) { if (initial.count(tc.get()) && trivial_filter(*tc)) in.rsure.insert(tc.get()); if (!initial.count(tc.get())) in.eligible = false; }
            }
            for (auto& tc : topo.inverse) if (tc->owners.count(pod.uid)) { in.rmask.insert(tc.get()); if (tc->is_hostname) in.rsure.insert(tc.get()); }
            if (in.narrow.size() == 1 && in.host.empty() && in.narrow[0]->type == kSpread && dyn_groups.count(in.narrow[0]) && !podReqs.has(in.narrow[0]->key) && in.narrow[0]->max_skew >= 0 && in.narrow[0]->max_skew < (1 << 24)) { in.dyn = 1; in.dg = in.narrow[0]; in.self = tg_selects(*in.dg, pod); }
            else if (in.narrow.empty() && in.host.size() == 1 && in.host[0]->type != kAffinity) { in.dyn = 2; in.dg = in.host[0]; in.self = tg_selects(*in.dg, pod); }
          }
          if (!in.eligible) break;
          if (maxcls > 0 && !seen.count(in.ev)) { if ((int)seen.size() == maxcls) break; seen.insert(in.ev); }
          I.push_back(std::move(in)); ++n;
        } }
      if (n < 2) { out[2]++; if (!sequential_step()) break; continue; }
      // ---- one evaluation per evaluation class against the snapshot: strict, "but for the skew", changes-the-node ----
      std::vector<uint64_t> m(n, 0), mo(n, 0), chg(n, 0);
      { std::map<std::string, size_t> firstof;
        for (size_t k = 0; k < n; ++k) {
          auto it = firstof.find(I[k].ev); if (it != firstof.end()) { m[k] = m[it->second]; mo[k] = mo[it->second]; chg[k] = chg[it->second]; continue; }
          firstof[I[k].ev] = k; PodState& ps = pods[queue[k]];
          for (size_t i = 0; i < nc; ++i) {
            Dry d = dry_v2(C[i].n, C[i].e, ps, ResList{}, I[k].dyn == 1 ? I[k].dg : nullptr);
            if (d.ok) { mo[k] |= 1ull << i; if (!d.skew_failed) m[k] |= 1ull << i; if (d.changes) chg[k] |= 1ull << i; }
          }
        } }
      // ---- the resolver ----
      std::vector<int> win(n, -1); size_t k = 0; GS rall, rdyn; std::map<G, std::map<DomKey, int>> dd; uint64_t moved = 0, closed = 0;
      auto hostdom = [&](const Cand& c) { const Reqs& r2 = c.n ? c.n->requirements : c.e->requirements; return hn.of(r2.m.at(hostnameKey).values.v[0]); };
      auto scaled = [](const ResList& r, int64_t t) { ResList o; for (auto& kv : r) o[kv.first] = kv.second * t; return o; };
      auto same_run = [&](size_t a, size_t b) { return I[a].tfull.empty() && I[b].tfull.empty() && I[a].ev == I[b].ev; };
      // what one pod records on candidate c (track_records / track_one + count_host): returns the part of its record set that goes into rall
      auto place_records = [&](Cand& c, const Info& in, bool chg_c) -> GS {
        for (G g : in.rsure) if (g->is_hostname) c.hrec[g]++;
        GS inx;
        for (G g : in.rmask) {
          if (!dyn_groups.count(g)) { inx.insert(g); continue; }
          if (c.pinned) { dd[g][c.zone]++; rdyn.insert(g); }
          else if (chg_c) inx.insert(g);
        }
        return inx;
      };
      bool cut = false;
      while (k < n && !cut) {
        const Info& in = I[k]; PodState& ps = pods[queue[k]];
        const bool follows = in.dyn == 1;
        if (meets(in.tmask, rall) || (!follows && !no_rdyn && meets(in.tmask, rdyn))) { out[4]++; break; }
        uint64_t mk = m[k], unk = 0; GS tfk = in.tfull;
        if (follows) {
          TopologyGroup* g = in.dg; auto& ddg = dd[g]; bool anyd = false; for (auto& kv : ddg) if (kv.second) anyd = true;
          int64_t mn = INT64_MAX; for (auto& kv : g->domains) { const int64_t c2 = (int64_t)kv.second + (ddg.count(kv.first) ? ddg[kv.first] : 0); if (c2 < mn) mn = c2; }
          uint64_t pinned = 0, okz = 0;
          for (size_t i = 0; i < nc; ++i) if (C[i].pinned) {
            pinned |= 1ull << i; auto it = g->domains.find(C[i].zone);
            if (it != g->domains.end()) { const int64_t c2 = (int64_t)it->second + (ddg.count(C[i].zone) ? ddg[C[i].zone] : 0) + (in.self ? 1 : 0); if (c2 - mn <= (int64_t)g->max_skew) okz |= 1ull << i; }
          }
          if (anyd) unk = mo[k] & ~pinned;
          mk = (mo[k] & okz) | (anyd ? unk : (m[k] & ~pinned));
          tfk.erase(g);
        }
        const bool hsk = in.dyn == 2;
        size_t r = 1; if (tfk.empty()) while (k + r < n && same_run(k + r - 1, k + r)) ++r;
        // candidates the round already used: exact resources with what it put on them; hostname counters
        uint64_t A = mk & ~moved;
        for (size_t i = 0; i < nc; ++i) if ((mk & moved) >> i & 1ull) {
          bool ok = dry_v2(C[i].n, C[i].e, ps, C[i].extra, follows ? in.dg : nullptr).ok;
          if (hsk) {
            const TopologyGroup* h = in.dg; const int extra = C[i].hrec.count(h) ? C[i].hrec[h] : 0; int64_t slack = 0;
            if (h->type == kSpread) { auto it = h->domains.find(hostdom(C[i])); const int64_t c0 = it == h->domains.end() ? 0 : it->second; slack = (int64_t)h->max_skew - (in.self ? 1 : 0) - c0; }
            if (slack < 0) slack = 0; if (slack > 255) slack = 255;
            if (extra > slack) ok = false;
          } else { GS z; for (G g : tfk) if (in.zmask.count(g)) z.insert(g); if (meets(z, C[i].rsure)) ok = false; }
          if (ok) A |= 1ull << i;
        }
        if (!A) { out[5]++; break; }
        int bu = -1; for (size_t i = 0; i < nc; ++i) if ((A >> i) & 1ull) if (bu < 0 || C[i].key < C[bu].key) bu = (int)i;
        if ((unk >> bu) & 1ull) { out[4]++; break; }
        const bool bu_moved = (moved >> bu) & 1ull, bu_new = (size_t)bu >= E;
        if (bu_moved) {
          if (bu_new && !window_complete && C[bu].cnt > cnt_last) { out[5]++; break; }
          if ((closed >> bu) & 1ull) { out[5]++; break; }
          if (meets(tfk, hsk ? C[bu].unsure : C[bu].racc)) { out[4]++; break; }
        }
        if (r >= 2 && !bu_moved && bu_new) {      // SWEEP
          std::vector<size_t> S0; for (size_t i = 0; i < nc; ++i) if (((A & ~moved) >> i) & 1ull) if (C[i].cnt == C[bu].cnt) S0.push_back(i);
          const size_t sN = std::min(r, S0.size());
          if (sN >= 2) {
            for (size_t x = 0; x < sN; ++x) {
              const size_t i = S0[x], j = k + x; Cand& c = C[i]; const Info& ij = I[j]; const bool chg_i = (chg[k] >> i) & 1ull;
              c.extra = res_merge(c.extra, ij.req); for (G g : ij.rmask) { c.racc.insert(g); if (!ij.rsure.count(g)) c.unsure.insert(g); } for (G g : ij.rsure) c.rsure.insert(g);
              c.np = 1; c.last = (int)j; c.cnt++; c.key = ((uint64_t)c.cnt << 8) | (uint64_t)(63 - j);
              for (G g : place_records(c, ij, chg_i)) rall.insert(g);
              win[j] = (int)i; moved |= 1ull << i; if (chg_i) closed |= 1ull << i;
            }
            k += sN; continue;
          }
        }
        size_t t = 1;
        if (r >= 2 && !((chg[k] >> bu) & 1ull)) {   // CLIMB
          size_t t_order = r;
          if (bu_new) {
            uint64_t other = UINT64_MAX; for (size_t i = 0; i < nc; ++i) if (((A >> i) & 1ull) && (int)i != bu && C[i].key < other) other = C[i].key;
            uint64_t oc = other == UINT64_MAX ? 0x00FFFFFFull : (other >> 8);
            if (!window_complete) oc = std::min<uint64_t>(oc, cnt_last);
            t_order = oc >= C[bu].cnt ? (size_t)(oc - C[bu].cnt + 1) : 1;
          }
          size_t t_res = 1; while (t_res < r && dry_v2(C[bu].n, C[bu].e, ps, res_merge(C[bu].extra, scaled(in.req, (int64_t)t_res)), nullptr).ok) ++t_res;
          t = std::max<size_t>(1, std::min(std::min(r, t_order), t_res));
        }
        { Cand& c = C[bu]; const bool chg_bu = (chg[k] >> bu) & 1ull;
          for (size_t x = 0; x < t; ++x) {
            const Info& ij = I[k + x];
            c.extra = res_merge(c.extra, ij.req); for (G g : ij.rmask) { c.racc.insert(g); if (!ij.rsure.count(g)) c.unsure.insert(g); } for (G g : ij.rsure) c.rsure.insert(g);
            for (G g : place_records(c, ij, chg_bu)) rall.insert(g);
            win[k + x] = bu;
          }
          c.np += (int)t; c.last = (int)(k + t - 1); if (bu_new) { c.cnt += t; c.key = ((uint64_t)c.cnt << 8) | (uint64_t)(63 - (k + t - 1)); }
          moved |= 1ull << bu; if (chg_bu) closed |= 1ull << bu; }
        k += t;
      }
      const size_t n_ok = k;
      if (n_ok == 0) { out[2]++; if (!sequential_step()) break; continue; }
      out[1]++; if (n_ok >= 8) out[6] += (long long)n_ok; if ((long long)n_ok > out[7]) out[7] = (long long)n_ok;
      // ---- commit the predictions through the real algorithm and compare ----
      bool stop = false;
      for (size_t j = 0; j < n_ok; ++j) {
        int pi = queue.front(); queue.pop_front(); st.queue_pops++;
        PodState& ps = pods[pi];
        const size_t before_nodes = new_nodes.size();
        bool placed = add(ps);
        bool match = placed && new_nodes.size() == before_nodes;
        if (match) { const Cand& c = C[win[j]]; match = c.e ? (!c.e->pods.empty() && c.e->pods.back() == ps.index) : (!c.n->pods.empty() && c.n->pods.back() == ps.index); }
        if (!match) { out[3]++; if (out[3] <= 5) fprintf(stderr, "resolver rule violated: pod %d (round position %zu of %zu) predicted candidate %d\n", pi, j, n_ok, win[j]); }
        if (!placed) { bool relaxed = relax(ps.spec); queue.push_back(pi); if (relaxed) { lastLen.clear(); ps.stage++; st.relaxations++; topo.update(ps.spec); } else lastLen[pi] = queue.size(); stop = true; break; }
        out[0]++;
      }
      (void)stop;
    }
    unscheduled.assign(queue.begin(), queue.end());
    for (auto& n : new_nodes) n->requirements.m.erase(hostnameKey);
  }
};

// ------------------------------------------------------------------------------------------------
// NewScheduler assembly: provisioner.go:237-296 + scheduler.go:42-94,221-267
// ------------------------------------------------------------------------------------------------
static std::unique_ptr<Scheduler> build(const ksp::Problem& pr, bool inert_topology) {
  auto s = std::make_unique<Scheduler>();
  s->zoneKey = S(ksp::kZone); s->ctKey = S(ksp::kCapacityType); s->hostnameKey = S(ksp::kHostname);
  for (const char* k : {ksp::kProvisionerName, ksp::kZone, ksp::kRegion, ksp::kInstanceType, ksp::kArch, ksp::kOS, ksp::kCapacityType}) s->cx.well_known.insert(S(k));
  for (auto& k : pr.extra_well_known) s->cx.well_known.insert(S(k));
  s->cv.pr = &pr; for (auto& n : pr.nodes) s->cv.node_by_name[n.name] = &n;

  s->itypes.reserve(pr.instance_types.size());
  for (size_t i = 0; i < pr.instance_types.size(); ++i) { IType it; it.src = &pr.instance_types[i]; it.index = (int)i; it.requirements = reqs_from_exprs(pr.instance_types[i].requirements); s->itypes.push_back(std::move(it)); }

  // OrderByWeight (apis/v1alpha5/provisioner.go:132-136), canonical: stable
  std::vector<const ksp::Provisioner*> provs; for (auto& p : pr.provisioners) provs.push_back(&p);
  std::stable_sort(provs.begin(), provs.end(), [](const ksp::Provisioner* a, const ksp::Provisioner* b) { return a->weight > b->weight; });
  s->templates.reserve(provs.size());
  for (auto* p : provs) {
    MachineTemplate t; t.prov = p;   // NewMachineTemplate machinetemplate.go:46-62
    StrMap labels = p->labels; labels[ksp::kProvisionerName] = p->name;
    t.requirements.add_all(reqs_from_exprs(p->requirements)); t.requirements.add_all(reqs_from_labels(labels)); t.taints = p->taints;
    s->templates.push_back(std::move(t));
    auto& lst = s->instance_types[p->name];
    for (int idx : p->instance_types) lst.push_back(&s->itypes.at(idx));
    // domain universe, provisioner.go:267-276 (Requirement.Values(): for complement sets the excluded values)
    for (int idx : p->instance_types) for (auto& kv : s->itypes[idx].requirements.m) for (Sym v : kv.second.values.v) s->topo.domains[kv.first].insert(STR(v));
    Reqs preq = reqs_from_exprs(p->requirements);
    for (auto& kv : preq.m) if (req_operator(kv.second) == Op::In) for (Sym v : kv.second.values.v) s->topo.domains[kv.first].insert(STR(v));
    for (auto& tt : p->taints) if (tt.effect == "PreferNoSchedule") s->tolerate_prefer_no_schedule = true;   // scheduler.go:49-56
    if (p->has_limits) s->remaining[p->name] = p->limits;                                                      // :71-75
  }

  // pods (working copies: Relax mutates specs)
  s->pods.reserve(pr.pods.size());
  for (size_t i = 0; i < pr.pods.size(); ++i) { PodState ps; ps.spec = pr.pods[i]; ps.index = (int)i; s->pods.push_back(std::move(ps)); }

  // NewTopology, topology.go:56-80
  s->topo.inert = inert_topology; s->topo.cx = &s->cx; s->topo.cv = &s->cv; s->topo.hn = &s->hn; s->topo.st = &s->st;
  if (!inert_topology) {
    for (auto& p : pr.pods) s->topo.excluded.insert(p.uid);
    s->topo.update_inverse_affinities();
    for (auto& ps : s->pods) s->topo.update(ps.spec);
  }

  // getDaemonOverhead, scheduler.go:250-267
  for (auto& t : s->templates) {
    std::vector<const ksp::Pod*> daemons;
    for (auto& d : pr.daemons) { ksp::Pod dp = d; if (!taints_tolerates(t.taints, dp)) continue; if (!reqs_compatible(s->cx, t.requirements, new_pod_requirements(dp))) continue; daemons.push_back(&d); }
    s->daemon_overhead[&t] = requests_for_pods(daemons);
  }
  // calculateExistingMachines, scheduler.go:221-248 + NewExistingNode, existingnode.go:41-75
  for (auto& n : pr.nodes) {
    if (!n.in_state) continue;
    if (!n.owned()) continue;
    std::vector<const ksp::Pod*> daemons;
    for (auto& d : pr.daemons) { ksp::Pod dp = d; if (!taints_tolerates(n.taints, dp)) continue; if (!reqs_compatible(s->cx, reqs_from_labels(n.labels), new_pod_requirements(dp))) continue; daemons.push_back(&d); }
    ResList remainingDaemon = res_subtract(requests_for_pods(daemons), n.daemonset_requests);
    for (auto& kv : remainingDaemon) if (kv.second < 0) kv.second = 0;
    auto en = std::make_unique<ExistingNode>();
    en->sn = &n; en->available = n.available; en->taints = n.taints; en->requests = remainingDaemon; en->requirements = reqs_from_labels(n.labels);
    for (auto& hp : n.host_ports) { std::string ip = hp.ip.empty() ? "0.0.0.0" : hp.ip; en->ports.reserved["~existing~/" + std::to_string(en->ports.reserved.size())].push_back({canon_ip(ip), hp.port, hp.proto}); }
    for (auto& v : n.volumes) en->volumes[v.driver].insert(v.pvc);
    for (auto& kv : n.volume_limits) en->volume_limits[kv.first] = kv.second;
    auto hl = n.labels.find(ksp::kHostname); std::string hostname = (hl == n.labels.end() || hl->second.empty()) ? n.name : hl->second;
    en->requirements.add(new_req(ksp::kHostname, Op::In, {hostname}));
    s->topo.register_domain(s->hostnameKey, s->hn.of(S(hostname)));
    s->existing.push_back(std::move(en));
    auto pl = n.labels.find(ksp::kProvisionerName);
    auto rem = s->remaining.find(pl->second);
    if (rem != s->remaining.end()) rem->second = res_subtract(rem->second, n.capacity);   // scheduler.go:244-246
  }
  return s;
}

// ------------------------------------------------------------------------------------------------
// KSR1 result text
// ------------------------------------------------------------------------------------------------
static std::string tokq(const std::string& s) { return s.empty() ? "~" : s; }
static std::string result_text(Scheduler& s, double solve_seconds) {
  std::ostringstream o;
  o << "KSR1\nNEWNODES " << s.new_nodes.size() << "\n";
  // report new nodes in creation order (the Go slice order at return is the last sort's order; the
  // creation order is the canonical, sort-independent listing)
  std::vector<Node*> ns; for (auto& n : s.new_nodes) ns.push_back(n.get());
  std::sort(ns.begin(), ns.end(), [](Node* a, Node* b) { return a->seq < b->seq; });
  for (auto* n : ns) {
    o << "NODE " << tokq(n->tmpl->prov->name) << " " << n->pods.size(); for (int p : n->pods) o << " " << p;
    o << " " << n->options.size(); for (auto* it : n->options) o << " " << tokq(it->src->name);
    o << " " << n->requests.size(); for (auto& kv : n->requests) o << " " << kv.first << " " << kv.second;
    o << " " << n->requirements.m.size();
    std::map<std::string, const Req*> byname; for (auto& kv : n->requirements.m) byname[STR(kv.first)] = &kv.second;
    for (auto& kv : byname) {
      const Req& r = *kv.second; o << " " << kv.first << " " << (r.complement ? 1 : 0) << " " << r.values.size();
      std::vector<std::string> vs; for (Sym v : r.values.v) vs.push_back(STR(v)); std::sort(vs.begin(), vs.end()); for (auto& v : vs) o << " " << tokq(v);
      if (r.has_gt) o << " " << r.gt; else o << " -"; if (r.has_lt) o << " " << r.lt; else o << " -";
    }
    o << "\n";
  }
  o << "EXISTING " << s.existing.size() << "\n";
  for (auto& e : s.existing) { o << "ENODE " << tokq(e->sn->name) << " " << e->pods.size(); for (int p : e->pods) o << " " << p; o << "\n"; }
  o << "UNSCHEDULED " << s.unscheduled.size(); for (int p : s.unscheduled) o << " " << p; o << "\n";
  o << "STAGES " << s.pods.size(); for (auto& p : s.pods) o << " " << p.stage; o << "\n";
  o << "REASONS " << s.unscheduled.size(); for (int p : s.unscheduled) o << " " << p << " " << s.pods[p].reasons; o << "\n";
  o << "STATS 6 attempts " << s.st.attempts << " types_scanned " << s.st.types_scanned << " domains_scanned " << s.st.domains_scanned
    << " relaxations " << s.st.relaxations << " queue_pops " << s.st.queue_pops << " solve_ns " << (int64_t)(solve_seconds * 1e9) << "\n";
  o << "END\n";
  return o.str();
}

}  // namespace oracle

// ------------------------------------------------------------------------------------------------
// C entry points (ctypes) + CLI
// ------------------------------------------------------------------------------------------------
extern "C" {

// Solve one KSP1 problem; returns 0 and a malloc'd KSR1 text, or <0 and a malloc'd error message.
// flags bit0: inert topology (the reference benchmark's &scheduling.Topology{}, scheduling_benchmark_test.go:123)
//       bit1: order the open nodes with the restated Go sort.Slice (pdqsort) instead of the canonical stable sort
int ko_solve(const char* ksp_text, size_t len, int flags, char** out_text) {
  oracle::Interner in; oracle::g_in = &in;
  try {
    ksp::Problem pr = ksp::Parser(ksp_text, len).parse();
    auto s = oracle::build(pr, (flags & 1) != 0);
    s->gosort = (flags & 2) != 0;
    auto t0 = std::chrono::steady_clock::now();
    s->solve();
    double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::string r = oracle::result_text(*s, dt);
    *out_text = strdup(r.c_str()); oracle::g_in = nullptr; return 0;
  } catch (const std::exception& e) { *out_text = strdup(e.what()); oracle::g_in = nullptr; return -1; }
}
void ko_free(char* p) { free(p); }
// The restated sort.Slice on plain keys: perm_out[i] = index (into keys) of the element at sorted position i.
void ko_gosort_order(const int* keys, int n, int* perm_out) {
  std::vector<int> perm(n); for (int i = 0; i < n; ++i) perm[i] = i;
  oracle::go_sort_slice(n, [&](int i, int j) { return keys[perm[i]] < keys[perm[j]]; }, [&](int i, int j) { std::swap(perm[i], perm[j]); });
  for (int i = 0; i < n; ++i) perm_out[i] = perm[i];
}

// Model check of the kernel's round speculation (see Scheduler::solve_spec): solves through predictions, returns the
// KSR1 text (must equal ko_solve's) and the counters.
int ko_solve_spec2(const char* ksp_text, size_t len, int W, int flags, int maxcls, long long* counters, char** out_text) {
  oracle::Interner in; oracle::g_in = &in;
  try {
    ksp::Problem pr = ksp::Parser(ksp_text, len).parse();
    auto s = oracle::build(pr, false);
    if (flags & 8) s->solve_watermark_check(counters, (flags & 16) != 0, (flags & 32) != 0);
    else if (flags & 2) s->solve_spec_v2(W, counters, flags, maxcls); else s->solve_spec(W, counters, flags, maxcls);
    std::string r = oracle::result_text(*s, 0.0);
    *out_text = strdup(r.c_str()); oracle::g_in = nullptr; return 0;
  } catch (const std::exception& e) { *out_text = strdup(e.what()); oracle::g_in = nullptr; return -1; }
}
int ko_solve_spec(const char* ksp_text, size_t len, int W, long long* counters, char** out_text) {
  oracle::Interner in; oracle::g_in = &in;
  try {
    ksp::Problem pr = ksp::Parser(ksp_text, len).parse();
    auto s = oracle::build(pr, false);
    s->solve_spec(W, counters);
    std::string r = oracle::result_text(*s, 0.0);
    *out_text = strdup(r.c_str()); oracle::g_in = nullptr; return 0;
  } catch (const std::exception& e) { *out_text = strdup(e.what()); oracle::g_in = nullptr; return -1; }
}

// Requirement algebra probes for the reference truth tables.  A requirement is given as
// "<op> <nvals> <val>*"; the result of Intersection is rendered "c=<0|1> vals=[a,b] gt=<n|-> lt=<n|->".
static oracle::Req parse_req_spec(const char* spec) {
  std::istringstream is(spec); std::string op; int n; is >> op >> n; std::vector<std::string> vals(n); for (auto& v : vals) is >> v;
  return oracle::new_req("key", ksp::parse_op(op), vals);
}
static std::string render_req(const oracle::Req& r) {
  std::vector<std::string> vs; for (auto v : r.values.v) vs.push_back(oracle::STR(v)); std::sort(vs.begin(), vs.end());
  std::string s = std::string("c=") + (r.complement ? "1" : "0") + " vals=[";
  for (size_t i = 0; i < vs.size(); ++i) s += (i ? "," : "") + vs[i];
  s += "] gt=" + (r.has_gt ? std::to_string(r.gt) : std::string("-")) + " lt=" + (r.has_lt ? std::to_string(r.lt) : std::string("-"));
  return s;
}
int ko_req_intersection(const char* a, const char* b, char* out, size_t cap) {
  oracle::Interner in; oracle::g_in = &in;
  std::string s = render_req(oracle::req_intersection(parse_req_spec(a), parse_req_spec(b)));
  snprintf(out, cap, "%s", s.c_str()); oracle::g_in = nullptr; return 0;
}
int ko_req_has(const char* a, const char* value) { oracle::Interner in; oracle::g_in = &in; int r = oracle::req_has(parse_req_spec(a), oracle::S(value)); oracle::g_in = nullptr; return r; }
// returns the operator name index: 0 In 1 NotIn 2 Exists 3 DoesNotExist
int ko_req_operator(const char* a) { oracle::Interner in; oracle::g_in = &in; int r = (int)oracle::req_operator(parse_req_spec(a)); oracle::g_in = nullptr; return r; }
long long ko_req_len(const char* a) { oracle::Interner in; oracle::g_in = &in; long long r = oracle::req_len(parse_req_spec(a)); oracle::g_in = nullptr; return r; }
// Requirements.Compatible on one key; spec "-" means the unconstrained (empty) Requirements.
int ko_reqs_compatible(const char* key, int key_is_well_known, const char* a, const char* b) {
  oracle::Interner in; oracle::g_in = &in; oracle::Ctx cx; if (key_is_well_known) cx.well_known.insert(oracle::S(key));
  oracle::Reqs ra, rb;
  if (std::string(a) != "-") { auto r = parse_req_spec(a); r.key = oracle::S(key); ra.add(r); }
  if (std::string(b) != "-") { auto r = parse_req_spec(b); r.key = oracle::S(key); rb.add(r); }
  int ok = oracle::reqs_compatible(cx, ra, rb); oracle::g_in = nullptr; return ok;
}
long long ko_parse_quantity_milli(const char* s, int* err) { try { *err = 0; return ksp::parse_quantity_milli(s); } catch (...) { *err = 1; return 0; } }

}  // extern "C"

#ifdef ORACLE_MAIN
int main(int argc, char** argv) {
  if (argc < 2) { fprintf(stderr, "usage: %s problem.ksp [--inert-topology]\n", argv[0]); return 2; }
  std::ifstream f(argv[1], std::ios::binary); std::stringstream ss; ss << f.rdbuf(); std::string text = ss.str();
  int flags = (argc > 2 && std::string(argv[2]) == "--inert-topology") ? 1 : 0;
  char* out = nullptr; int rc = ko_solve(text.data(), text.size(), flags, &out);
  if (rc != 0) { fprintf(stderr, "error: %s\n", out); return 1; }
  fputs(out, stdout); ko_free(out); return 0;
}
#endif
