// hreq.hpp -- host-side mirror of the reference's scheduling.Requirement / Requirements API
// (pkg/scheduling/requirement.go, requirements.go) plus the small host-only predicates the flattening
// needs (tolerations, label selectors, resource lists).  Same names and argument meaning as the Go
// types so host code reads like the reference; this is the part of the path that stays on the host
// (NewScheduler / NewTopology assembly).  The hot loop itself never runs here.
#pragma once
#include <algorithm>
#include <climits>
#include <map>
#include <optional>
#include <set>
#include <string>
#include <vector>

#include "ksp.hpp"

namespace ksh {

using ksp::Op;

inline bool Atoi(const std::string& s, long long* out) {   // strconv.Atoi
  if (s.empty()) return false;
  size_t i = 0; bool neg = false;
  if (s[0] == '+' || s[0] == '-') { neg = s[0] == '-'; i = 1; if (s.size() == 1) return false; }
  __int128 v = 0;
  for (; i < s.size(); ++i) { if (s[i] < '0' || s[i] > '9') return false; v = v * 10 + (s[i] - '0'); if (v > ((__int128)1 << 64)) return false; }
  if (neg) v = -v;
  if (v > LLONG_MAX || v < LLONG_MIN) return false;
  *out = (long long)v; return true;
}

// scheduling.Requirement (requirement.go:36-42)
class Requirement {
 public:
  std::string Key; bool complement = false; std::set<std::string> values; std::optional<long long> greaterThan, lessThan;

  static Requirement New(const std::string& key, Op op, const std::vector<std::string>& vals = {}) {   // NewRequirement :44-68
    Requirement r; r.Key = ksp::normalize_key(key); r.complement = !(op == Op::In || op == Op::DoesNotExist);
    if (op == Op::In || op == Op::NotIn) r.values.insert(vals.begin(), vals.end());
    long long v = 0;
    if (op == Op::Gt) { Atoi(vals.at(0), &v); r.greaterThan = v; }
    if (op == Op::Lt) { Atoi(vals.at(0), &v); r.lessThan = v; }
    return r;
  }
  static bool within(const std::string& value, const std::optional<long long>& gt, const std::optional<long long>& lt) {   // withinIntPtrs :227-243
    if (!gt && !lt) return true;
    long long v; if (!Atoi(value, &v)) return false;
    if (gt && *gt >= v) return false;
    if (lt && *lt <= v) return false;
    return true;
  }
  bool Has(const std::string& v) const { return (complement ? !values.count(v) : values.count(v) != 0) && within(v, greaterThan, lessThan); }   // :171-176
  long long Len() const { return complement ? LLONG_MAX - (long long)values.size() : (long long)values.size(); }                              // :199-204
  Op Operator() const { if (complement) return Len() < LLONG_MAX ? Op::NotIn : Op::Exists; return Len() > 0 ? Op::In : Op::DoesNotExist; }  // :186-197
  bool IsNotInOrDoesNotExist() const { Op o = Operator(); return o == Op::NotIn || o == Op::DoesNotExist; }
  Requirement Intersection(const Requirement& q) const {   // :117-150
    Requirement out; out.Key = Key; out.complement = complement && q.complement;
    std::optional<long long> gt = greaterThan, lt = lessThan;
    if (q.greaterThan && (!gt || *q.greaterThan > *gt)) gt = q.greaterThan;
    if (q.lessThan && (!lt || *q.lessThan < *lt)) lt = q.lessThan;
    if (gt && lt && *gt >= *lt) { out.complement = false; return out; }
    std::set<std::string> vals;
    if (complement && q.complement) { vals = values; vals.insert(q.values.begin(), q.values.end()); }
    else if (complement && !q.complement) { for (auto& v : q.values) if (!values.count(v)) vals.insert(v); }
    else if (!complement && q.complement) { for (auto& v : values) if (!q.values.count(v)) vals.insert(v); }
    else { for (auto& v : values) if (q.values.count(v)) vals.insert(v); }
    for (auto& v : vals) if (within(v, gt, lt)) out.values.insert(v);
    if (out.complement) { out.greaterThan = gt; out.lessThan = lt; }
    return out;
  }
  std::string identity() const {
    std::string s = complement ? "!" : "="; for (auto& v : values) s += v + ","; s += "|";
    s += greaterThan ? std::to_string(*greaterThan) : "-"; s += "|"; s += lessThan ? std::to_string(*lessThan) : "-"; return s;
  }
};

// scheduling.Requirements (requirements.go:32-223)
class Requirements {
 public:
  std::map<std::string, Requirement> m;
  bool Has(const std::string& k) const { return m.count(k) != 0; }
  Requirement Get(const std::string& k) const { auto it = m.find(k); return it == m.end() ? Requirement::New(k, Op::Exists) : it->second; }   // :114-120
  void Add(const Requirement& r) { auto it = m.find(r.Key); if (it != m.end()) it->second = r.Intersection(it->second); else m.emplace(r.Key, r); }   // :87-94
  void Add(const Requirements& o) { for (auto& kv : o.m) Add(kv.second); }
  static Requirements FromExprs(const std::vector<ksp::Expr>& es) { Requirements r; for (auto& e : es) r.Add(Requirement::New(e.key, e.op, e.values)); return r; }   // NewNodeSelectorRequirements :43-49
  static Requirements FromLabels(const ksp::StrMap& l) { Requirements r; for (auto& kv : l) r.Add(Requirement::New(kv.first, Op::In, {kv.second})); return r; }     // NewLabelRequirements :52-58
  bool Intersects(const Requirements& in) const {   // :189-206, true == no error
    for (auto& kv : m) { auto it = in.m.find(kv.first); if (it == in.m.end()) continue;
      if (kv.second.Intersection(it->second).Len() == 0) { if (it->second.IsNotInOrDoesNotExist() && kv.second.IsNotInOrDoesNotExist()) continue; return false; } }
    return true;
  }
  bool Compatible(const Requirements& in, const std::set<std::string>& wellKnown) const {   // :123-133, true == no error
    for (auto& kv : in.m) { if (wellKnown.count(kv.first)) continue; if (Has(kv.first) || kv.second.IsNotInOrDoesNotExist()) continue; return false; }
    return Intersects(in);
  }
};

// NewPodRequirements (requirements.go:61-78); sorts the preferred terms in place like the reference.
inline Requirements NewPodRequirements(ksp::Pod& pod) {
  Requirements r = Requirements::FromLabels(pod.node_selector);
  if (!pod.preferred_affinity.empty()) {
    std::stable_sort(pod.preferred_affinity.begin(), pod.preferred_affinity.end(), [](const ksp::PreferredTerm& a, const ksp::PreferredTerm& b) { return a.weight > b.weight; });
    r.Add(Requirements::FromExprs(pod.preferred_affinity[0].exprs));
  }
  if (!pod.required_affinity.empty()) r.Add(Requirements::FromExprs(pod.required_affinity[0]));
  return r;
}

// v1.Toleration.ToleratesTaint (k8s.io/api v0.25.4 core/v1/toleration.go)
inline bool ToleratesTaint(const ksp::Toleration& t, const ksp::Taint& taint) {
  if (!t.effect.empty() && t.effect != taint.effect) return false;
  if (!t.key.empty() && t.key != taint.key) return false;
  if (t.op.empty() || t.op == "Equal") return t.value == taint.value;
  return t.op == "Exists";
}
inline bool Tolerates(const std::vector<ksp::Taint>& taints, const ksp::Pod& pod) {   // Taints.Tolerates, taints.go:28-40
  for (auto& taint : taints) { bool ok = false; for (auto& t : pod.tolerations) ok = ok || ToleratesTaint(t, taint); if (!ok) return false; }
  return true;
}

// labels.Selector.Matches via metav1.LabelSelectorAsSelector; nil selector == labels.Nothing()
inline bool SelectorMatches(const ksp::Selector& sel, const ksp::StrMap& labels) {
  if (sel.nil) return false;
  for (auto& kv : sel.match_labels) { auto it = labels.find(kv.first); if (it == labels.end() || it->second != kv.second) return false; }
  for (auto& e : sel.match_exprs) {
    auto it = labels.find(e.key); const bool has = it != labels.end();
    const bool in = has && std::find(e.values.begin(), e.values.end(), it->second) != e.values.end();
    if (e.op == Op::In) { if (!in) return false; }
    else if (e.op == Op::NotIn) { if (in) return false; }
    else if (e.op == Op::Exists) { if (!has) return false; }
    else if (e.op == Op::DoesNotExist) { if (has) return false; }
    else return false;
  }
  return true;
}

// utils/resources
using ksp::ResList;
inline ResList Merge(const ResList& a, const ResList& b) { ResList r = a; for (auto& kv : b) r[kv.first] += kv.second; return r; }
inline ResList Subtract(const ResList& lhs, const ResList& rhs) { ResList r = lhs; for (auto& kv : r) { auto it = rhs.find(kv.first); if (it != rhs.end()) kv.second -= it->second; } return r; }
inline ResList MaxResources(const ResList& a, const ResList& b) { ResList r = a; for (auto& kv : b) { auto it = r.find(kv.first); if (it == r.end() || kv.second > it->second) r[kv.first] = kv.second; } return r; }
inline ResList ContainerRequests(const ksp::Container& c) { ResList r = c.requests; for (auto& kv : c.limits) if (!r.count(kv.first)) r[kv.first] = kv.second; return r; }   // MergeResourceLimitsIntoRequests
inline ResList RequestsForPod(const ksp::Pod& p) {   // RequestsForPods(pod), resources.go:25-33 + Ceiling :78-89
  ResList r; for (auto& c : p.containers) r = Merge(r, ContainerRequests(c));
  for (auto& c : p.init_containers) r = MaxResources(r, ContainerRequests(c));
  r["pods"] = 1000; return r;
}

}  // namespace ksh
