// ksp.hpp -- semantic problem model of one Solve() call + KSP1 text parser + quantity parser.
//
// This header carries DATA TYPES ONLY (the analogue of k8s.io/api's v1.Pod, karpenter's
// v1alpha5.Provisioner and cloudprovider.InstanceType): no scheduling logic lives here.  It is the
// input closure of reference provisioner.go:237-296 (NewScheduler assembly), topology.go:56-80
// (NewTopology) and scheduler.go:96-133 (Solve).  Grammar: see karpenter_core_amd/model.py.
//
// Quantities are exact: every resource.Quantity is converted to int64 *milli-units*
// (reference arithmetic is exact decimal, k8s.io/apimachinery v0.25.4 pkg/api/resource; the
// scheduling tests use "1.8G", "100M", "10Mi", "1.1": suite_test.go:1084,1125).  Sub-milli
// precision is rejected loudly instead of being rounded.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

namespace ksp {

struct Error : std::runtime_error { using std::runtime_error::runtime_error; };

// ---- label constants (reference pkg/apis/v1alpha5/labels.go:26-110) ----
static const char* const kZone = "topology.kubernetes.io/zone";
static const char* const kRegion = "topology.kubernetes.io/region";
static const char* const kHostname = "kubernetes.io/hostname";
static const char* const kInstanceType = "node.kubernetes.io/instance-type";
static const char* const kArch = "kubernetes.io/arch";
static const char* const kOS = "kubernetes.io/os";
static const char* const kCapacityType = "karpenter.sh/capacity-type";
static const char* const kProvisionerName = "karpenter.sh/provisioner-name";

// NormalizedLabels, labels.go:103-109
inline std::string normalize_key(const std::string& k) {
  if (k == "failure-domain.beta.kubernetes.io/zone") return kZone;
  if (k == "beta.kubernetes.io/arch") return kArch;
  if (k == "beta.kubernetes.io/os") return kOS;
  if (k == "beta.kubernetes.io/instance-type") return kInstanceType;
  if (k == "failure-domain.beta.kubernetes.io/region") return kRegion;
  return k;
}

enum class Op { In, NotIn, Exists, DoesNotExist, Gt, Lt };

inline Op parse_op(const std::string& s) {
  if (s == "In") return Op::In;
  if (s == "NotIn") return Op::NotIn;
  if (s == "Exists") return Op::Exists;
  if (s == "DoesNotExist") return Op::DoesNotExist;
  if (s == "Gt") return Op::Gt;
  if (s == "Lt") return Op::Lt;
  throw Error("KSP1: bad operator " + s);
}

struct Expr { std::string key; Op op; std::vector<std::string> values; };
using StrMap = std::map<std::string, std::string>;
using ResList = std::map<std::string, int64_t>;   // resource name -> milli-units (presence matters)

struct Selector {           // metav1.LabelSelector; nil => selects nothing
  bool nil = true;
  StrMap match_labels;
  std::vector<Expr> match_exprs;
};
struct AffinityTerm { std::string topology_key; std::vector<std::string> namespaces; Selector selector; };
struct WeightedTerm { int32_t weight; AffinityTerm term; };
struct Spread { int32_t max_skew; std::string key; bool schedule_anyway; Selector selector; };
struct Toleration { std::string key, op, value, effect; };
struct Taint { std::string key, value, effect; };
struct HostPort { std::string ip; int32_t port; std::string proto; };
struct Container { ResList requests, limits; std::vector<HostPort> ports; };
struct PreferredTerm { int32_t weight; std::vector<Expr> exprs; };
// One mounted volume after the lookups of VolumeUsage.validate (volumeusage.go:145-195): the CSI driver the claim resolves to (PVC -> bound
// PV's Spec.CSI.Driver, else StorageClass.Provisioner) and the claim's id ("<namespace>/<claim>", or "<namespace>/<pod>-<volume>" for a
// generic ephemeral volume).  Volumes that resolve to no CSI driver are not listed (":185-188 might be a non-CSI driver").
struct Volume { std::string driver, pvc; };

struct Pod {
  std::string uid, ns;
  int64_t creation_ts = 0;
  StrMap labels, node_selector;
  std::vector<std::vector<Expr>> required_affinity;
  std::vector<PreferredTerm> preferred_affinity;
  std::vector<Toleration> tolerations;
  std::vector<Container> containers, init_containers;
  std::vector<Spread> spread;
  std::vector<AffinityTerm> affinity_required, anti_required;
  std::vector<WeightedTerm> affinity_preferred, anti_preferred;
  std::vector<Volume> volumes;
  bool volume_error = false;   // a lookup of VolumeUsage.validate failed (claim / storage class / volume not found): ExistingNode.Add returns that error
};

struct Offering { std::string capacity_type, zone; double price; bool available; };
struct InstanceType {
  std::string name;
  std::vector<Expr> requirements;
  std::vector<Offering> offerings;
  ResList capacity, overhead;
};
struct Provisioner {
  std::string name; int32_t weight = 0;
  StrMap labels;
  std::vector<Expr> requirements;
  std::vector<Taint> taints;
  bool has_limits = false; ResList limits;
  std::vector<int32_t> instance_types;
};
struct StateNode {
  std::string name; bool in_state = true;
  StrMap labels; std::vector<Taint> taints;
  ResList available, capacity, daemonset_requests;
  std::vector<HostPort> host_ports;
  std::vector<std::pair<std::string, int32_t>> volume_limits;   // state.Node.VolumeLimits(): CSINode allocatable count per driver (cluster.go:292-304)
  std::vector<Volume> volumes;                                   // state.Node.VolumeUsage(): volumes of the pods bound to the node
  bool owned() const { auto it = labels.find(kProvisionerName); return it != labels.end() && !it->second.empty(); }
};
struct ClusterPod { std::string uid, ns, node_name; StrMap labels; std::vector<AffinityTerm> anti_required; };

// One event of a cluster's life between two snapshots (ksh_env_apply; state.Cluster's UpdateNode / DeleteNode / UpdatePod / DeletePod, cluster.go)
struct DeltaEvent { enum Kind { NodeAdd, NodeRemove, PodBind, PodUnbind } kind = NodeAdd; StateNode node; Pod pod; std::string name; };

struct Problem {
  std::vector<std::string> extra_well_known;
  std::vector<InstanceType> instance_types;
  std::vector<Provisioner> provisioners;
  std::vector<StateNode> nodes;
  std::vector<ClusterPod> cluster_pods;
  std::vector<Pod> daemons;
  std::vector<Pod> pods;
  bool simulation_mode = false;
};

// ---- quantity parsing: text -> exact int64 milli-units ----
// Grammar follows resource.Quantity: <sign><digits>[.<digits>][<suffix>] with suffix in
// m | "" | k M G T P E | Ki Mi Gi Ti Pi Ei | e<N>/E<N> (decimal exponent).
inline int64_t parse_quantity_milli(const std::string& s) {
  if (s.empty()) throw Error("empty quantity");
  size_t i = 0; bool neg = false;
  if (s[i] == '+' || s[i] == '-') { neg = s[i] == '-'; ++i; }
  __int128 mant = 0; int frac_digits = 0; bool seen_digit = false, seen_dot = false;
  for (; i < s.size(); ++i) {
    char c = s[i];
    if (c >= '0' && c <= '9') { mant = mant * 10 + (c - '0'); if (seen_dot) ++frac_digits; seen_digit = true;
      if (mant > ((__int128)1 << 100)) throw Error("quantity too large: " + s); }
    else if (c == '.' && !seen_dot) seen_dot = true;
    else break;
  }
  if (!seen_digit) throw Error("bad quantity: " + s);
  std::string suf = s.substr(i);
  __int128 mul_num = 1000, mul_den = 1;  // value in milli = mant * mul_num / (mul_den * 10^frac_digits)
  if (suf == "") {}
  else if (suf == "m") mul_num = 1;
  else if (suf == "k") mul_num = (__int128)1000 * 1000;
  else if (suf == "M") mul_num = (__int128)1000 * 1000000;
  else if (suf == "G") mul_num = (__int128)1000 * 1000000000LL;
  else if (suf == "T") mul_num = (__int128)1000 * 1000000000000LL;
  else if (suf == "P") mul_num = (__int128)1000 * 1000000000000000LL;
  else if (suf == "E") mul_num = (__int128)1000 * 1000000000000000000LL;
  else if (suf == "Ki") mul_num = (__int128)1000 << 10;
  else if (suf == "Mi") mul_num = (__int128)1000 << 20;
  else if (suf == "Gi") mul_num = (__int128)1000 << 30;
  else if (suf == "Ti") mul_num = (__int128)1000 << 40;
  else if (suf == "Pi") mul_num = (__int128)1000 << 50;
  else if (suf == "Ei") mul_num = (__int128)1000 << 60;
  else if (suf.size() >= 2 && (suf[0] == 'e' || suf[0] == 'E')) {
    char* end = nullptr; long ex = std::strtol(suf.c_str() + 1, &end, 10);
    if (*end != 0 || ex < -30 || ex > 30) throw Error("bad quantity exponent: " + s);
    for (long k = 0; k < ex; ++k) mul_num *= 10;
    for (long k = 0; k > ex; --k) mul_den *= 10;
  } else throw Error("bad quantity suffix: " + s);
  for (int k = 0; k < frac_digits; ++k) mul_den *= 10;
  __int128 num = mant * mul_num;
  if (num % mul_den != 0) throw Error("quantity finer than 1 milli-unit is not representable exactly: " + s);
  __int128 v = num / mul_den;
  if (v > (__int128)INT64_MAX / 4) throw Error("quantity overflows int64 milli-units: " + s);
  return neg ? -(int64_t)v : (int64_t)v;
}

// ---- tokenizer / parser ----
class Parser {
 public:
  Parser(const char* text, size_t len) : p_(text), e_(text + len) {}

  // What state.Cluster hears between two passes over the cluster (cluster.go UpdateNode / DeleteNode / UpdatePod / DeletePod), as KSD1 text:
  //   KSD1 <n events>  { NODE+ <NODE record> | NODE- <node name> | BIND <node name> POD <pod record> | UNBIND <pod uid> }*  END
  // The records are KSP1's own (the NODE record without its leading keyword, the POD record with it).
  std::vector<DeltaEvent> parse_delta() {
    std::vector<DeltaEvent> ev;
    expect("KSD1");
    for (int n = count(); n > 0; --n) {
      DeltaEvent e; const std::string k = tok();
      if (k == "NODE+") { e.kind = DeltaEvent::NodeAdd; e.node = node(); }
      else if (k == "NODE-") { e.kind = DeltaEvent::NodeRemove; e.name = str(); }
      else if (k == "BIND") { e.kind = DeltaEvent::PodBind; e.name = str(); expect("POD"); e.pod = pod(); }
      else if (k == "UNBIND") { e.kind = DeltaEvent::PodUnbind; e.name = str(); }
      else throw Error("KSD1: expected NODE+|NODE-|BIND|UNBIND got " + k);
      ev.push_back(std::move(e));
    }
    expect("END");
    return ev;
  }
  // (a snapshot is patched in place by ksh_env_apply: the vectors it appends to keep room, so that nothing that points into them moves)
  static size_t spare(size_t n, size_t least) { return n + std::max(least, n / 4); }

  Problem parse() {
    Problem pr;
    expect("KSP1");
    expect("WELLKNOWN");
    for (int n = count(); n > 0; --n) pr.extra_well_known.push_back(str());
    expect("ITS");
    int nit = count(); pr.instance_types.reserve(nit);
    for (int i = 0; i < nit; ++i) {
      expect("IT");
      InstanceType it; it.name = str();
      for (int n = count(); n > 0; --n) it.requirements.push_back(expr());
      for (int n = count(); n > 0; --n) {
        Offering o; o.capacity_type = str(); o.zone = str(); o.price = std::strtod(tok().c_str(), nullptr); o.available = count() != 0;
        it.offerings.push_back(o);
      }
      it.capacity = reslist(); it.overhead = reslist();
      pr.instance_types.push_back(std::move(it));
    }
    expect("PROVS");
    for (int np = count(); np > 0; --np) {
      expect("PROV");
      Provisioner pv; pv.name = str(); pv.weight = (int32_t)integer();
      pv.labels = strmap();
      for (int n = count(); n > 0; --n) pv.requirements.push_back(expr());
      for (int n = count(); n > 0; --n) pv.taints.push_back(taint());
      int64_t nl = integer();
      if (nl >= 0) { pv.has_limits = true; for (; nl > 0; --nl) { std::string k = str(); pv.limits[k] = parse_quantity_milli(str()); } }
      for (int n = count(); n > 0; --n) pv.instance_types.push_back((int32_t)integer());
      pr.provisioners.push_back(std::move(pv));
    }
    expect("NODES");
    { const int nn = count(); pr.nodes.reserve(spare(nn, 256));
      for (int i = 0; i < nn; ++i) { expect("NODE"); pr.nodes.push_back(node()); } }
    expect("CPODS");
    for (int nc = count(); nc > 0; --nc) {
      expect("CPOD");
      ClusterPod cp; cp.uid = str(); cp.ns = str(); cp.node_name = str(); cp.labels = strmap();
      for (int n = count(); n > 0; --n) cp.anti_required.push_back(term());
      pr.cluster_pods.push_back(std::move(cp));
    }
    expect("DAEMONS");
    for (int nd = count(); nd > 0; --nd) { expect("POD"); pr.daemons.push_back(pod()); }
    expect("SIM"); pr.simulation_mode = count() != 0;
    expect("PODS");
    int npods = count(); pr.pods.reserve(spare(npods, 4096));
    for (int i = 0; i < npods; ++i) { expect("POD"); pr.pods.push_back(pod()); }
    expect("END");
    return pr;
  }

 private:
  const char* p_; const char* e_;

  std::string tok() {
    while (p_ < e_ && (*p_ == ' ' || *p_ == '\n' || *p_ == '\t' || *p_ == '\r')) ++p_;
    if (p_ >= e_) throw Error("KSP1: unexpected end of input");
    const char* b = p_;
    while (p_ < e_ && !(*p_ == ' ' || *p_ == '\n' || *p_ == '\t' || *p_ == '\r')) ++p_;
    return std::string(b, p_);
  }
  std::string str() { std::string t = tok(); return t == "~" ? std::string() : t; }
  // optional trailing sections (volumes): look at the next token without consuming it.  Unambiguous: every record of the grammar opens with a
  // keyword (POD / NODE / CPOD / ...), so the token after a pod's ANP list or a node's host ports is a keyword, never a free-form uid or name.
  bool next_is(const char* s) {
    const char* save = p_; bool is = false;
    if (p_ < e_) { try { is = tok() == s; } catch (const Error&) { is = false; } }
    p_ = save; return is;
  }
  int64_t integer() {
    std::string t = tok(); char* end = nullptr; long long v = std::strtoll(t.c_str(), &end, 10);
    if (*end != 0 || t.empty()) throw Error("KSP1: expected integer, got " + t);
    return v;
  }
  int count() { int64_t v = integer(); if (v < 0 || v > (1 << 30)) throw Error("KSP1: bad count"); return (int)v; }
  void expect(const char* s) { std::string t = tok(); if (t != s) throw Error(std::string("KSP1: expected ") + s + " got " + t); }
  StrMap strmap() { StrMap m; for (int n = count(); n > 0; --n) { std::string k = str(); m[k] = str(); } return m; }
  ResList reslist() { ResList m; for (int n = count(); n > 0; --n) { std::string k = str(); m[k] = parse_quantity_milli(str()); } return m; }
  Expr expr() { Expr e; e.key = str(); e.op = parse_op(tok()); for (int n = count(); n > 0; --n) e.values.push_back(str()); return e; }
  Taint taint() { Taint t; t.key = str(); t.value = str(); t.effect = str(); return t; }
  HostPort hostport() { HostPort h; h.ip = str(); h.port = (int32_t)integer(); h.proto = str(); return h; }
  Selector selector() {
    Selector s; std::string t = tok();
    if (t == "NIL") { s.nil = true; return s; }
    if (t != "SEL") throw Error("KSP1: expected NIL|SEL got " + t);
    s.nil = false; s.match_labels = strmap();
    for (int n = count(); n > 0; --n) s.match_exprs.push_back(expr());
    return s;
  }
  AffinityTerm term() {
    AffinityTerm t; t.topology_key = str();
    for (int n = count(); n > 0; --n) t.namespaces.push_back(str());
    t.selector = selector(); return t;
  }
  StateNode node() {
    StateNode sn; sn.name = str(); sn.in_state = count() != 0;
    sn.labels = strmap();
    for (int n = count(); n > 0; --n) sn.taints.push_back(taint());
    sn.available = reslist(); sn.capacity = reslist(); sn.daemonset_requests = reslist();
    for (int n = count(); n > 0; --n) sn.host_ports.push_back(hostport());
    if (next_is("VL")) { expect("VL"); for (int n = count(); n > 0; --n) { std::string d = str(); sn.volume_limits.emplace_back(d, (int32_t)integer()); } }
    if (next_is("VU")) { expect("VU"); for (int n = count(); n > 0; --n) { Volume v; v.driver = str(); v.pvc = str(); sn.volumes.push_back(std::move(v)); } }
    return sn;
  }
  Pod pod() {
    Pod p; p.uid = str(); p.ns = str(); p.creation_ts = integer();
    expect("L"); p.labels = strmap();
    expect("NS"); p.node_selector = strmap();
    expect("RA");
    for (int n = count(); n > 0; --n) { std::vector<Expr> t; for (int m = count(); m > 0; --m) t.push_back(expr()); p.required_affinity.push_back(std::move(t)); }
    expect("PA");
    for (int n = count(); n > 0; --n) { PreferredTerm t; t.weight = (int32_t)integer(); for (int m = count(); m > 0; --m) t.exprs.push_back(expr()); p.preferred_affinity.push_back(std::move(t)); }
    expect("TOL");
    for (int n = count(); n > 0; --n) { Toleration t; t.key = str(); t.op = str(); t.value = str(); t.effect = str(); p.tolerations.push_back(t); }
    expect("C");
    for (int n = count(); n > 0; --n) { Container c; c.requests = reslist(); c.limits = reslist(); for (int m = count(); m > 0; --m) c.ports.push_back(hostport()); p.containers.push_back(std::move(c)); }
    expect("I");
    for (int n = count(); n > 0; --n) { Container c; c.requests = reslist(); c.limits = reslist(); p.init_containers.push_back(std::move(c)); }
    expect("TS");
    for (int n = count(); n > 0; --n) {
      Spread s; s.max_skew = (int32_t)integer(); s.key = str(); std::string wu = tok();
      if (wu == "ScheduleAnyway") s.schedule_anyway = true; else if (wu == "DoNotSchedule") s.schedule_anyway = false; else throw Error("KSP1: bad whenUnsatisfiable " + wu);
      s.selector = selector(); p.spread.push_back(std::move(s));
    }
    expect("AFR"); for (int n = count(); n > 0; --n) p.affinity_required.push_back(term());
    expect("AFP"); for (int n = count(); n > 0; --n) { WeightedTerm w; w.weight = (int32_t)integer(); w.term = term(); p.affinity_preferred.push_back(std::move(w)); }
    expect("ANR"); for (int n = count(); n > 0; --n) p.anti_required.push_back(term());
    expect("ANP"); for (int n = count(); n > 0; --n) { WeightedTerm w; w.weight = (int32_t)integer(); w.term = term(); p.anti_preferred.push_back(std::move(w)); }
    if (next_is("VOL")) {
      expect("VOL"); int64_t n = integer();
      if (n < 0) p.volume_error = true;
      else for (; n > 0; --n) { Volume v; v.driver = str(); v.pvc = str(); p.volumes.push_back(std::move(v)); }
    }
    return p;
  }
};

inline Problem parse(const std::string& text) { return Parser(text.data(), text.size()).parse(); }

}  // namespace ksp
