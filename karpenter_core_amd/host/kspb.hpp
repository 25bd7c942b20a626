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
 private:
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

}  // namespace ksp
