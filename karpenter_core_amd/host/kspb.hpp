// kspb.hpp -- binary pod ingress: the caller's pending pods as flat u32 / i64 arrays instead of KSP1 text.
//
// What the caller of provisioner.go:301-307 holds when it calls NewScheduler / Solve is a []*v1.Pod.  A cgo shim cannot hand Go
// pointers to C beyond the call, and printing 100k pods as text (16 MB) then parsing them costs more than the Solve.  The binary form
// is what a shim can fill in one pass over its pods (one goroutine per block): per BLOCK a table of interned strings, per pod a
// record of u32 words -- string ids, counts, int64 quantities in milli-units as two words -- plus the uid (a string id) and the
// creation timestamp.  The record grammar mirrors the KSP1 POD record (karpenter_core_amd/model.py) field for field:
//
//   spec      := ns:S  labels:MAP  node_selector:MAP
//                N { N {expr} }                                  required node-affinity terms
//                N { weight:I N {expr} }                         preferred node-affinity terms
//                N { key:S op:S value:S effect:S }               tolerations
//                N { reslist reslist N { ip:S port:I proto:S } } containers (requests, limits, host ports)
//                N { reslist reslist }                           init containers
//                N { max_skew:I key:S schedule_anyway:I selector }   topology spread constraints
//                N {term}  N { weight:I term }  N {term}  N { weight:I term }      pod affinity required / preferred, anti-affinity required / preferred
//                V { driver:S claim:S }                          resolved volumes; V = 0xFFFFFFFF: VolumeUsage.validate failed (no entries follow)
//   MAP       := N { key:S value:S }        (ascending keys: equal specs must be equal WORD FOR WORD to be merged; see below)
//   reslist   := N { name:S lo:U hi:U }     int64 milli-units, exact (resource.Quantity.MilliValue of a value with no sub-milli part)
//   expr      := key:S op:I N {value:S}     op: 0 In 1 NotIn 2 Exists 3 DoesNotExist 4 Gt 5 Lt
//   selector  := 1 | 0 MAP N {expr}         1: nil selector (selects nothing)
//   term      := topology_key:S N {namespace:S} selector
//
// Ingest never builds 100k pod objects: inside a block two pods share a spec iff their records are equal word for word (a block's
// string table is interned, so equal ids <=> equal strings; a caller that lists map entries in another order merely gets more
// specs, never a wrong result -- classes are merged on their semantics later).  Only the DISTINCT records are decoded into
// ksp::Pod objects; across blocks those few are merged field by field.  The rest of a pod is 16 bytes: spec id, timestamp, uid span.
#pragma once
#include <cstdint>
#include <cstring>
#include <string>
#include <string_view>
#include <vector>

#include "../../include/kshost.h"
#include "ksp.hpp"

namespace ksp {

// The pending batch in compact form: distinct specs + per-pod (spec, timestamp, uid).
struct PodBatch {
  std::vector<Pod> specs;                // distinct specs in order of first occurrence in the batch (uid empty, creation_ts 0)
  std::vector<uint32_t> pod_spec;        // [P]
  std::vector<int64_t> ts;               // [P] creationTimestamp
  std::vector<uint32_t> uid_off;         // [P+1] into uid_bytes
  std::string uid_bytes;
  size_t size() const { return pod_spec.size(); }
  std::string_view uid(size_t i) const { return std::string_view(uid_bytes.data() + uid_off[i], uid_off[i + 1] - uid_off[i]); }
};

// One spec record -> ksp::Pod (uid / creation_ts left empty).  Throws ksp::Error on a malformed record.
class SpecReader {
 public:
  SpecReader(const ksh_pod_block& b, const uint32_t* w, const uint32_t* e) : b_(b), w_(w), e_(e) {}
  Pod read() {
    Pod p; p.ns = s(); p.labels = map(); p.node_selector = map();
    for (uint32_t n = cnt(); n; --n) { std::vector<Expr> t; for (uint32_t m = cnt(); m; --m) t.push_back(expr()); p.required_affinity.push_back(std::move(t)); }
    for (uint32_t n = cnt(); n; --n) { PreferredTerm t; t.weight = i(); for (uint32_t m = cnt(); m; --m) t.exprs.push_back(expr()); p.preferred_affinity.push_back(std::move(t)); }
    for (uint32_t n = cnt(); n; --n) { Toleration t; t.key = s(); t.op = s(); t.value = s(); t.effect = s(); p.tolerations.push_back(std::move(t)); }
    for (uint32_t n = cnt(); n; --n) { Container c; c.requests = res(); c.limits = res(); for (uint32_t m = cnt(); m; --m) { HostPort h; h.ip = s(); h.port = i(); h.proto = s(); c.ports.push_back(std::move(h)); } p.containers.push_back(std::move(c)); }
    for (uint32_t n = cnt(); n; --n) { Container c; c.requests = res(); c.limits = res(); p.init_containers.push_back(std::move(c)); }
    for (uint32_t n = cnt(); n; --n) { Spread t; t.max_skew = i(); t.key = s(); t.schedule_anyway = u() != 0; t.selector = selector(); p.spread.push_back(std::move(t)); }
    for (uint32_t n = cnt(); n; --n) p.affinity_required.push_back(term());
    for (uint32_t n = cnt(); n; --n) { WeightedTerm t; t.weight = i(); t.term = term(); p.affinity_preferred.push_back(std::move(t)); }
    for (uint32_t n = cnt(); n; --n) p.anti_required.push_back(term());
    for (uint32_t n = cnt(); n; --n) { WeightedTerm t; t.weight = i(); t.term = term(); p.anti_preferred.push_back(std::move(t)); }
    const uint32_t nv = u();
    if (nv == 0xFFFFFFFFu) p.volume_error = true;
    else { if (nv > (uint32_t)(e_ - w_)) throw Error("pod block: volume count runs past the record"); for (uint32_t n = nv; n; --n) { Volume v; v.driver = s(); v.pvc = s(); p.volumes.push_back(std::move(v)); } }
    if (w_ != e_) throw Error("pod block: trailing words in a spec record");
    return p;
  }
 protected:
  const ksh_pod_block& b_; const uint32_t* w_; const uint32_t* e_;
  uint32_t u() { if (w_ >= e_) throw Error("pod block: spec record ends early"); return *w_++; }
  int32_t i() { return (int32_t)u(); }
  uint32_t cnt() { const uint32_t n = u(); if (n > (uint32_t)(e_ - w_)) throw Error("pod block: count runs past the record"); return n; }      // every element takes at least one word
  std::string s() { const uint32_t id = u(); if (id >= b_.n_strings) throw Error("pod block: string id out of range"); return std::string(b_.str_bytes + b_.str_off[id], b_.str_off[id + 1] - b_.str_off[id]); }
  StrMap map() { StrMap m; for (uint32_t n = cnt(); n; --n) { std::string k = s(); m[std::move(k)] = s(); } return m; }
  ResList res() { ResList m; for (uint32_t n = cnt(); n; --n) { std::string k = s(); const uint64_t lo = u(), hi = u(); m[std::move(k)] = (int64_t)(lo | (hi << 32)); } return m; }
  Expr expr() { Expr x; x.key = s(); const uint32_t op = u(); if (op > 5) throw Error("pod block: bad operator"); x.op = (Op)op; for (uint32_t n = cnt(); n; --n) x.values.push_back(s()); return x; }
  Selector selector() { Selector x; const uint32_t nil = u(); if (nil > 1) throw Error("pod block: bad selector tag"); x.nil = nil == 1; if (!x.nil) { x.match_labels = map(); for (uint32_t n = cnt(); n; --n) x.match_exprs.push_back(expr()); } return x; }
  AffinityTerm term() { AffinityTerm t; t.topology_key = s(); for (uint32_t n = cnt(); n; --n) t.namespaces.push_back(s()); t.selector = selector(); return t; }
};

// The ENVIRONMENT through the same kind of door (include/kshost.h `ksh_env_block`; round 5): everything NewScheduler reads besides the pending pods -- what a shim holds as
// []*cloudprovider.InstanceType (types.go:72-145), []v1alpha5.Provisioner, []*state.Node, the cluster's pods with required anti-affinity (topology.go:231-276) and the
// daemonset pods -- as ONE stream of u32 words over ONE string table (the record grammar of the pod blocks: S string id, I int32, U uint32, MAP, reslist, expr, term):
//
//   env        := N {well_known:S}  N {instance_type}  N {provisioner}  N {state_node}  N {cluster_pod}  N {daemon}  simulation_mode:U
//   instance_type := name:S N {expr} N { capacity_type:S zone:S price_lo:U price_hi:U available:U } reslist reslist          (price: the IEEE-754 bits of the float64; capacity, overhead)
//   provisioner   := name:S weight:I MAP N {expr} N {taint} has_limits:U reslist N {instance_type_index:U}                     (labels, requirements, taints, limits, its instance types)
//   taint         := key:S value:S effect:S
//   state_node    := name:S in_state:U MAP N {taint} reslist reslist reslist N { ip:S port:I proto:S } N { driver:S count:I } N { driver:S claim:S }
//                                                                                   (labels, taints, available, capacity, daemonset requests, host ports, volume limits, volumes in use)
//   cluster_pod   := uid:S ns:S node_name:S MAP N {term}
//   daemon        := uid:S ts_lo:U ts_hi:U nwords:U spec                           (spec: the pod blocks' record, nwords words)
//
// The result is the ksp::Problem ksh_parse builds from the KSP1 text of the same objects with `PODS 0` (tests/test_env_block.py: equal flat problems, equal results).
class EnvReader : public SpecReader {
 public:
  EnvReader(const ksh_pod_block& strings, const uint32_t* w, const uint32_t* e) : SpecReader(strings, w, e) {}
  Problem read_env() {
    Problem pr;
    for (uint32_t n = cnt(); n; --n) pr.extra_well_known.push_back(s());
    { const uint32_t n = cnt(); pr.instance_types.reserve(n);
      for (uint32_t k = 0; k < n; ++k) {
        InstanceType it; it.name = s();
        for (uint32_t m = cnt(); m; --m) it.requirements.push_back(expr());
        for (uint32_t m = cnt(); m; --m) { Offering o; o.capacity_type = s(); o.zone = s(); const uint64_t lo = u(), hi = u(), bits = lo | (hi << 32); std::memcpy(&o.price, &bits, 8); o.available = u() != 0; it.offerings.push_back(std::move(o)); }
        it.capacity = res(); it.overhead = res();
        pr.instance_types.push_back(std::move(it));
      } }
    for (uint32_t n = cnt(); n; --n) {
      Provisioner pv; pv.name = s(); pv.weight = i(); pv.labels = map();
      for (uint32_t m = cnt(); m; --m) pv.requirements.push_back(expr());
      for (uint32_t m = cnt(); m; --m) pv.taints.push_back(taint());
      pv.has_limits = u() != 0; pv.limits = res(); if (!pv.has_limits && !pv.limits.empty()) throw Error("env block: limits listed for a provisioner without limits");
      for (uint32_t m = cnt(); m; --m) { const uint32_t ix = u(); if (ix >= pr.instance_types.size()) throw Error("env block: instance type index out of range"); pv.instance_types.push_back((int32_t)ix); }
      pr.provisioners.push_back(std::move(pv));
    }
    for (uint32_t n = cnt(); n; --n) {
      StateNode sn; sn.name = s(); sn.in_state = u() != 0; sn.labels = map();
      for (uint32_t m = cnt(); m; --m) sn.taints.push_back(taint());
      sn.available = res(); sn.capacity = res(); sn.daemonset_requests = res();
      for (uint32_t m = cnt(); m; --m) { HostPort h; h.ip = s(); h.port = i(); h.proto = s(); sn.host_ports.push_back(std::move(h)); }
      for (uint32_t m = cnt(); m; --m) { std::string d = s(); sn.volume_limits.emplace_back(std::move(d), i()); }
      for (uint32_t m = cnt(); m; --m) { Volume v; v.driver = s(); v.pvc = s(); sn.volumes.push_back(std::move(v)); }
      pr.nodes.push_back(std::move(sn));
    }
    for (uint32_t n = cnt(); n; --n) {
      ClusterPod cp; cp.uid = s(); cp.ns = s(); cp.node_name = s(); cp.labels = map();
      for (uint32_t m = cnt(); m; --m) cp.anti_required.push_back(term());
      pr.cluster_pods.push_back(std::move(cp));
    }
    for (uint32_t n = cnt(); n; --n) {
      std::string uid = s(); const uint64_t lo = u(), hi = u(); const uint32_t nw = cnt();
      Pod d = SpecReader(b_, w_, w_ + nw).read(); w_ += nw;
      d.uid = std::move(uid); d.creation_ts = (int64_t)(lo | (hi << 32));
      pr.daemons.push_back(std::move(d));
    }
    pr.simulation_mode = u() != 0;
    if (w_ != e_) throw Error("env block: trailing words");
    return pr;
  }
 private:
  Taint taint() { Taint t; t.key = s(); t.value = s(); t.effect = s(); return t; }
};

}  // namespace ksp
