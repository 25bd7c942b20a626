// ksolve.hip -- HIP/CDNA4 (gfx950) implementation of the libksolve C ABI (include/ksolve.h).
//
// Kernels (all integer / bitmask work -- no MFMA: there is no dense contraction on this path):
//   ks_build_type_tables  transposes the instance-type requirement table into per-(key,value)
//                         T-bit masks ("which types admit value v of key k"), one wave per row,
//                         64 types per __ballot.
//   ks_grid_mc            template x pod-class merge: Taints.Tolerates + Requirements.Compatible +
//                         Requirements.Add for a fresh node (node.go:62-80 with m.Requirements = template).
//   ks_grid_types         the pod-class x instance-type feasibility grid: every lane owns one
//                         instance type (its requirement masks / allocatable vector live in registers,
//                         loaded once, coalesced SoA), streams over (template,class) records that are
//                         wave-uniform, and emits one 64-bit word per __ballot.  HBM-bound.
//   ks_pack               one persistent workgroup per Solve(): the first-fit-decreasing loop of
//                         scheduler.go:96-219.  Per pod: all open nodes are screened in parallel (one
//                         lane per node: taints, host ports, requirement intersection, topology domain
//                         choice, resource screen), a wave-shuffle + LDS arg-min picks the first feasible
//                         node in the reference's visiting order, the instance-type filter runs
//                         word-parallel on T-bit masks, and one lane commits.
//
// The reference functions each device function restates are cited inline (paths relative to
// aws/karpenter-core pkg/).  There is deliberately NO CPU fallback in this library: if no gfx950
// device is present every entry point returns KS_ERR_DEVICE.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/ksolve.h"
#include "ks_algebra.h"

#define KS_NT 1024           // threads per pack workgroup (16 waves on one CU)
#define KS_MAX_TOPO 24       // topology groups evaluated per pod class
#define KS_MAX_TOUCH 12      // distinct narrow keys a class may touch (own requirements + topology keys)

typedef uint64_t u64; typedef uint32_t u32; typedef int64_t i64; typedef int32_t i32; typedef uint8_t u8;

static thread_local std::string g_err;
static int fail(int code, const std::string& msg) { g_err = msg; return code; }
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return fail(KS_ERR_DEVICE, std::string(#x) + ": " + hipGetErrorString(e_)); } while (0)

// ------------------------------------------------------------------------------------------------
// device-side views
// ------------------------------------------------------------------------------------------------
struct ReqSetsD { u32 n; const u32* present; const u32* complement; const u64* mask; const i32* gt; const i32* lt; const i32* it_state; };

struct DevProb {
  u32 P, C, T, TW, M, E, K, R, G, GH, S, NMAX, flags, n_topologies;
  u32 wellknown_mask; const u32* key_nvalues; const i32* value_int; i32 key_zone, key_ct; u32 n_ct;
  const u32* it_present; const u32* it_complement; const u64* it_mask; const i64* it_alloc; const i64* it_cap; const u64* it_offer;
  const u8* its_inter; const u8* its_fail; const u8* its_nidne; const u64* its_types;
  ReqSetsD tmpl; const u64* tmpl_taints; const i64* tmpl_daemon; const u32* tmpl_daemon_present; const u64* tmpl_types;
  const u32* tmpl_limit_present; const i64* tmpl_remaining;
  ReqSetsD en; const u64* en_taints; const i64* en_avail; const i64* en_requests; const u32* en_requests_present; const u32* en_port_off;
  ReqSetsD cls; const u8* cls_hn_mode; const u32* cls_hn_off; const u32* hn_list; const i64* cls_requests; const u32* cls_requests_present;
  const u64* cls_tolerated; const u32* cls_port_off; const u64* ports;
  const u32* cls_own_off; const u32* own_list; const u32* cls_sel_off; const u32* sel_list;
  const u32* cls_isel_off; const u32* isel_list; const u32* cls_iown_off; const u32* iown_list;
  const u32* pod_stage_off; const u32* stage_cls; const u32* queue;
  const u8* grp_type; const i32* grp_key; const i32* grp_max_skew; const u8* grp_active; const u32* grp_filter_off; ReqSetsD flt;
  const i32* grp_count; const i32* grp_hslot; const i32* grph_count; const i32* grph_extra_pos;
  // derived static tables (built on the device by ks_build_type_tables / ks_grid_*)
  u64* kv_types;     // [K*64*TW] types lacking key k or whose requirement on k Has(value v)
  u64* cmplx_types;  // [K*TW]    types lacking key k or with a complement requirement on k
  u64* nidnex_types; // [K*TW]    types lacking key k or with operator in {NotIn, DoesNotExist} on k
  u64* pair_types;   // [64*TW]   types with an available offering for (zone,capacity-type) pair
  u8* mc_ok;         // [M*C]
  u32* mc_present; u32* mc_complement; u64* mc_mask; i32* mc_gt; i32* mc_lt; i32* mc_it;   // template ∩ class
  u64* grid;         // [M*C*TW]
};

// Mutable state of one Solve (device memory, one allocation).
struct DevState {
  // queue (queue.go:29-72)
  u32* q; u32* lastlen; u32* lastgen; i32* pod_stage; i32* pod_node; i32* pod_seq;
  // slots: [0,E) existing nodes, [E,E+NMAX) new nodes
  u32* s_present; u32* s_complement; u64* s_mask; i32* s_gt; i32* s_lt; i32* s_it;
  i64* s_req; u32* s_reqmask; i64* s_cap; u64* s_taints; i32* s_porthead;
  i32* n_tmpl; u32* n_count; u64* n_key; u64* n_alive;   // new nodes only, indexed by j = slot-E
  // topology
  i32* gcnt; u64* g_reg; u64* g_pos; u8* g_active; i32* hcnt; i32* g_hpos;
  // provisioner limits
  i64* remaining;
  // host-port pool
  u64* pp_entry; i32* pp_next; u32 pp_cap;
  // outputs
  u64* stats; u32* out_counts;   // out_counts: [0]=n_new [1]=n_unscheduled
  i32* unscheduled;
};

// ------------------------------------------------------------------------------------------------
// small device helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ KReq load_req(const u32 present, const u32 complement, const u64* mask, const i32* gt, const i32* lt, int k) {
  KReq r; r.present = (present >> k) & 1u; r.complement = (complement >> k) & 1u; r.mask = mask[k]; r.gt = gt[k]; r.lt = lt[k]; return r;
}
__device__ __forceinline__ KReq type_req(const DevProb& P, u32 t, int k) {   // instance types never carry bounds (encoder enforces)
  KReq r; r.present = (P.it_present[t] >> k) & 1u; r.complement = (P.it_complement[t] >> k) & 1u; r.mask = P.it_mask[(size_t)k * P.T + t]; r.gt = KS_NOGT; r.lt = KS_NOLT; return r;
}
__device__ __forceinline__ u64 ballot64(bool p) { return __ballot(p); }

// ------------------------------------------------------------------------------------------------
// ks_build_type_tables: one wave per table row
// rows: [0, K*64) -> kv_types[k][v]; [K*64, K*64+K) -> cmplx[k]; [.., +K) -> nidnex[k]; [.., +64) -> pair_types
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ks_build_type_tables(DevProb P) {
  const int lane = threadIdx.x & 63;
  const u32 wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const u32 nrows = P.K * 64 + 2 * P.K + 64;
  if (wave >= nrows) return;
  for (u32 w = 0; w < P.TW; ++w) {
    const u32 t = w * 64 + lane; const bool valid = t < P.T; bool bit = false;
    if (wave < P.K * 64) {
      const int k = wave >> 6, v = wave & 63;
      if (valid && (u32)v < P.key_nvalues[k]) {
        KReq a = type_req(P, t, k);
        bit = !a.present || ((a.complement ? ~a.mask : a.mask) >> v & 1ull);       // Has(v), requirement.go:171-176
      }
      u64 m = ballot64(bit); if (lane == 0) P.kv_types[(size_t)wave * P.TW + w] = m;
    } else if (wave < P.K * 64 + P.K) {
      const int k = wave - P.K * 64;
      if (valid) { KReq a = type_req(P, t, k); bit = !a.present || a.complement; }
      u64 m = ballot64(bit); if (lane == 0) P.cmplx_types[(size_t)k * P.TW + w] = m;
    } else if (wave < P.K * 64 + 2 * P.K) {
      const int k = wave - P.K * 64 - P.K;
      if (valid) { KReq a = type_req(P, t, k); bit = !a.present || kreq_nidne(a); }
      u64 m = ballot64(bit); if (lane == 0) P.nidnex_types[(size_t)k * P.TW + w] = m;
    } else {
      const int pair = wave - P.K * 64 - 2 * P.K;
      if (valid) bit = (P.it_offer[t] >> pair) & 1ull;
      u64 m = ballot64(bit); if (lane == 0) P.pair_types[(size_t)pair * P.TW + w] = m;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// ks_grid_mc: one thread per (template m, class c)
// node.go:62-80 for a fresh node: Taints.Tolerates, nodeRequirements.Compatible(podRequirements),
// nodeRequirements.Add(podRequirements).  A fresh node's hostname is a placeholder no pod can name
// (node.go:46), so a pod class with a concrete hostname requirement can never use a new node.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ks_grid_mc(DevProb P) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)P.M * P.C) return;
  const u32 m = idx / P.C, c = idx % P.C;
  bool ok = (P.tmpl_taints[m] & ~P.cls_tolerated[c]) == 0;
  if (P.cls_hn_mode[c] == 1) ok = false;
  const u32 tp = P.tmpl.present[m], tc = P.tmpl.complement[m], cp = P.cls.present[c], cc = P.cls.complement[c];
  u32 present = tp | cp, complement = 0;
  for (u32 k = 0; k < P.K; ++k) {
    KReq a = load_req(tp, tc, P.tmpl.mask + (size_t)m * P.K, P.tmpl.gt + (size_t)m * P.K, P.tmpl.lt + (size_t)m * P.K, k);
    KReq b = load_req(cp, cc, P.cls.mask + (size_t)c * P.K, P.cls.gt + (size_t)c * P.K, P.cls.lt + (size_t)c * P.K, k);
    const i32* vi = P.value_int + k * 64; const u32 nv = P.key_nvalues[k];
    if (kreq_compatible_fail(a, b, (P.wellknown_mask >> k) & 1u, vi, nv)) ok = false;
    KReq r = kreq_add(a, b, vi, nv);
    if (r.complement) complement |= 1u << k;
    P.mc_mask[idx * P.K + k] = r.mask; P.mc_gt[idx * P.K + k] = r.gt; P.mc_lt[idx * P.K + k] = r.lt;
  }
  const i32 sa = P.tmpl.it_state[m], sb = P.cls.it_state[c];
  if (P.its_fail[sa * P.S + sb]) ok = false;
  P.mc_it[idx] = P.its_inter[sa * P.S + sb];
  P.mc_present[idx] = present; P.mc_complement[idx] = complement; P.mc_ok[idx] = ok ? 1 : 0;
}

// ------------------------------------------------------------------------------------------------
// ks_grid_types: the feasibility grid.  filterInstanceTypesByRequirements (node.go:137-159) for a
// fresh node of template m receiving a pod of class c:
//   compatible  = instanceType.Requirements.Intersects(nodeRequirements) == nil     (node.go:143)
//   fits        = resources.Fits(daemonOverhead + podRequests, Allocatable())      (node.go:147)
//   hasOffering = some available offering whose zone / capacity-type the node allows (node.go:151)
// Wave (w, chunk): lane owns type t = 64*w + lane for the whole kernel; (m,c) records are wave-uniform.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ks_grid_types(DevProb P, u32 chunks) {
  const int lane = threadIdx.x & 63;
  const u32 wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const u32 w = wave % P.TW, chunk = wave / P.TW;
  if (chunk >= chunks) return;
  const u32 t = w * 64 + lane; const bool valid = t < P.T;
  // lane-private type record (registers)
  u32 tpres = 0, tcomp = 0; u64 tmask[KS_MAX_KEYS]; i64 talloc[KS_MAX_RES]; u64 toffer = 0;
  if (valid) { tpres = P.it_present[t]; tcomp = P.it_complement[t]; toffer = P.it_offer[t]; }
#pragma unroll
  for (int k = 0; k < KS_MAX_KEYS; ++k) tmask[k] = (valid && (u32)k < P.K) ? P.it_mask[(size_t)k * P.T + t] : 0;
#pragma unroll
  for (int r = 0; r < KS_MAX_RES; ++r) talloc[r] = (valid && (u32)r < P.R) ? P.it_alloc[(size_t)r * P.T + t] : 0;
  const size_t MC = (size_t)P.M * P.C;
  for (size_t mc = chunk; mc < MC; mc += chunks) {
    const u32 m = mc / P.C, c = mc % P.C;
    bool ok = valid && P.mc_ok[mc] && ((P.tmpl_types[(size_t)m * P.TW + w] >> lane) & 1ull);
    if (__builtin_amdgcn_readfirstlane(P.mc_ok[mc]) == 0) { if (lane == 0) P.grid[mc * P.TW + w] = 0; continue; }
    const u32 np = P.mc_present[mc], nc = P.mc_complement[mc];
    // compatible: per key present on both sides
    u32 both = np & tpres;
#pragma unroll
    for (int k = 0; k < KS_MAX_KEYS; ++k) {
      if (!((np >> k) & 1u)) continue;                  // wave-uniform
      KReq nreq; nreq.present = true; nreq.complement = (nc >> k) & 1u; nreq.mask = P.mc_mask[mc * P.K + k]; nreq.gt = P.mc_gt[mc * P.K + k]; nreq.lt = P.mc_lt[mc * P.K + k];
      if ((both >> k) & 1u) {
        KReq a; a.present = true; a.complement = (tcomp >> k) & 1u; a.mask = tmask[k]; a.gt = KS_NOGT; a.lt = KS_NOLT;
        if (kreq_intersects_fail(a, nreq, P.value_int + k * 64, P.key_nvalues[k])) ok = false;
      }
    }
    // instance-type key
    ok = ok && ((P.its_types[(size_t)P.mc_it[mc] * P.TW + w] >> lane) & 1ull);
    // fits
    const u32 rp = P.tmpl_daemon_present[m] | P.cls_requests_present[c];
#pragma unroll
    for (int r = 0; r < KS_MAX_RES; ++r) {
      if (!((rp >> r) & 1u)) continue;
      const i64 need = P.tmpl_daemon[(size_t)m * P.R + r] + P.cls_requests[(size_t)c * P.R + r];
      if (need > talloc[r]) ok = false;
    }
    // hasOffering
    u64 allowZ = ~0ull, allowC = ~0ull;
    if (P.key_zone >= 0 && ((np >> P.key_zone) & 1u)) { KReq z; z.present = true; z.complement = (nc >> P.key_zone) & 1u; z.mask = P.mc_mask[mc * P.K + P.key_zone]; z.gt = P.mc_gt[mc * P.K + P.key_zone]; z.lt = P.mc_lt[mc * P.K + P.key_zone]; allowZ = kreq_has_mask(z, P.value_int + P.key_zone * 64, P.key_nvalues[P.key_zone]); }
    if (P.key_ct >= 0 && ((np >> P.key_ct) & 1u)) { KReq z; z.present = true; z.complement = (nc >> P.key_ct) & 1u; z.mask = P.mc_mask[mc * P.K + P.key_ct]; z.gt = P.mc_gt[mc * P.K + P.key_ct]; z.lt = P.mc_lt[mc * P.K + P.key_ct]; allowC = kreq_has_mask(z, P.value_int + P.key_ct * 64, P.key_nvalues[P.key_ct]); }
    u64 pairs = 0;
    for (u64 zz = allowZ; zz; zz &= zz - 1) { const int z = __builtin_ctzll(zz); if ((u32)z * P.n_ct >= 64) break; pairs |= (allowC & ((1ull << P.n_ct) - 1)) << (z * P.n_ct); }
    if (P.n_ct == 0) pairs = ~0ull;
    ok = ok && ((toffer & pairs) != 0);
    const u64 word = ballot64(ok);
    if (lane == 0) P.grid[mc * P.TW + w] = word;
  }
}

// ------------------------------------------------------------------------------------------------
// pack kernel
// ------------------------------------------------------------------------------------------------
struct TopoItem {
  i32 g; i32 key; i32 hslot; i32 maxskew; i32 minc; u8 type; u8 self; u8 inverse; u8 pod_has; u64 PD; u64 reg; u64 pos;
};
struct ClsL {      // the popped pod's class, staged in LDS once per pod
  u32 c; u32 present, complement; i32 it_state; u32 hn_mode, hn_off, hn_cnt;
  u64 mask[KS_MAX_KEYS]; i32 gt[KS_MAX_KEYS]; i32 lt[KS_MAX_KEYS];
  i64 req[KS_MAX_RES]; u32 reqmask; u64 tol; u32 port_off, port_cnt;
  int ntopo; TopoItem topo[KS_MAX_TOPO];
};
struct ReqOut {    // requirement set of the winning node after the pod is added (written by lane 0)
  u32 present, complement; i32 it_state; u64 mask[KS_MAX_KEYS]; i32 gt[KS_MAX_KEYS]; i32 lt[KS_MAX_KEYS]; u32 changed; u32 topo_narrowed;
};
struct NodeView {  // where a candidate node's state lives (an open slot, or a template∩class record for a fresh node)
  u32 present, complement; i32 it_state; const u64* mask; const i32* gt; const i32* lt;
  u64 taints; i32 porthead; const i64* req; u32 reqmask; const i64* cap; i32 slot; bool existing; bool fresh;
};

__device__ __forceinline__ NodeView slot_view(const DevProb& P, const DevState& S, u32 s) {
  NodeView v; v.present = S.s_present[s]; v.complement = S.s_complement[s]; v.it_state = S.s_it[s];
  v.mask = S.s_mask + (size_t)s * P.K; v.gt = S.s_gt + (size_t)s * P.K; v.lt = S.s_lt + (size_t)s * P.K;
  v.taints = S.s_taints[s]; v.porthead = S.s_porthead[s]; v.req = S.s_req + (size_t)s * P.R; v.reqmask = S.s_reqmask[s];
  v.cap = S.s_cap + (size_t)s * P.R; v.slot = (i32)s; v.existing = s < P.E; v.fresh = false; return v;
}

__device__ __forceinline__ bool cls_allows_hostname(const DevProb& P, const ClsL& c, const NodeView& v) {
  if (c.hn_mode == 0) return true;
  bool inlist = false;
  if (v.existing) for (u32 i = 0; i < c.hn_cnt; ++i) if (P.hn_list[c.hn_off + i] == (u32)v.slot) { inlist = true; break; }
  return c.hn_mode == 1 ? inlist : !inlist;
}

// HostPortUsage.validate, hostportusage.go:81-93 / entry.matches :45-57
__device__ __forceinline__ bool ports_conflict(const DevProb& P, const DevState& S, const ClsL& c, i32 head) {
  for (u32 i = 0; i < c.port_cnt; ++i) {
    const u64 a = P.ports[c.port_off + i];
    for (i32 e = head; e >= 0; e = S.pp_next[e]) {
      const u64 b = S.pp_entry[e];
      if ((a >> 32) != (b >> 32)) continue;                                   // protocol + port
      const u32 ia = (u32)a, ib = (u32)b;
      if (ia == ib || ia == 0 || ib == 0) return true;                        // equal, or either unspecified
    }
  }
  return false;
}

// One attempt of Node.Add / ExistingNode.Add up to (not including) the instance-type filter.
// Returns 0: fails before the filter; 1: reaches the filter but fails the resource screen;
// 2: passes everything evaluated here.  With `out` != nullptr also writes the node's requirement set
// after Add (nodeRequirements after :80 and :90 of node.go / :105 and :115 of existingnode.go).
__device__ int eval_node(const DevProb& P, const DevState& S, const ClsL& c, const NodeView& v, ReqOut* out, bool merged = false) {
  // Taints.Tolerates, taints.go:28-40
  if (v.taints & ~c.tol) return 0;
  // hostname requirement of the pod against the node's `hostname In [own]`
  if (!merged && !cls_allows_hostname(P, c, v)) return 0;
  // HostPortUsage.Validate
  if (c.port_cnt && v.porthead >= 0 && ports_conflict(P, S, c, v.porthead)) return 0;
  // ExistingNode: resources.Fits(requests, available), existingnode.go:99-103 (exact, final)
  if (v.existing) {
    const u32 rp = v.reqmask | c.reqmask;
    for (u32 r = 0; r < P.R; ++r) if ((rp >> r) & 1u) { if (v.req[r] + c.req[r] > v.cap[r]) return 0; }
  }
  // Compatible + Add on the pod's keys
  int nt = 0; i32 tkey[KS_MAX_TOUCH]; KReq treq[KS_MAX_TOUCH];
  for (u32 bits = merged ? 0u : c.present; bits; bits &= bits - 1) {
    const int k = __builtin_ctz(bits);
    KReq a = load_req(v.present, v.complement, v.mask, v.gt, v.lt, k);
    KReq b = load_req(c.present, c.complement, c.mask, c.gt, c.lt, k);
    const i32* vi = P.value_int + k * 64; const u32 nv = P.key_nvalues[k];
    if (kreq_compatible_fail(a, b, (P.wellknown_mask >> k) & 1u, vi, nv)) return 0;
    tkey[nt] = k; treq[nt] = kreq_add(a, b, vi, nv); ++nt;
  }
  i32 it_state = v.it_state;
  if (c.it_state && !merged) { if (P.its_fail[v.it_state * P.S + c.it_state]) return 0; it_state = P.its_inter[v.it_state * P.S + c.it_state]; }
  const int n_own = nt;   // entries [0,n_own) come from the pod's own requirements

  // Topology.AddRequirements, topology.go:149-167, then Compatible + Add of the result (node.go:83-90)
  u32 topo_keys = 0; bool host_ok = true;
  u64 dom[KS_MAX_TOUCH]; u32 domset = 0;   // accumulated In-sets per touched entry
  for (int i = 0; i < c.ntopo; ++i) {
    const TopoItem& t = c.topo[i];
    if (t.key == KS_KEY_HOSTNAME) {
      const bool allowed = true;   // cls_allows_hostname already held above
      i32 cnt;
      if (v.fresh) cnt = S.g_active[t.g] ? 0 : -1;           // NewNode registers the placeholder first (node.go:47)
      else cnt = S.hcnt[(size_t)t.hslot * (P.E + P.NMAX) + v.slot];
      bool ok;
      if (t.type == 0) ok = cnt >= 0 && (i64)cnt + t.self <= (i64)t.maxskew;                         // nextDomainTopologySpread, min==0 for hostname (topologygroup.go:184-188)
      else if (t.type == 2) ok = allowed && cnt == 0;                                                 // nextDomainAntiAffinity :235-243
      else { const bool anypos = S.g_hpos[t.hslot] > 0; ok = anypos ? (cnt > 0) : (t.self && cnt >= 0); }   // nextDomainAffinity :202-233
      if (!ok) host_ok = false;
      continue;
    }
    const int k = t.key; const i32* vi = P.value_int + k * 64; const u32 nv = P.key_nvalues[k];
    // nodeDomains = nodeRequirements[key] (after adding the pod) or Exists
    int e = -1; for (int j = 0; j < nt; ++j) if (tkey[j] == k) { e = j; break; }
    KReq nodeD;
    if (e >= 0) nodeD = treq[e];
    else { nodeD = load_req(v.present, v.complement, v.mask, v.gt, v.lt, k); if (nt >= KS_MAX_TOUCH) return 0; e = nt; tkey[nt] = k; treq[nt] = nodeD; ++nt; }
    KReq nd = nodeD.present ? nodeD : kreq_exists();
    const u64 ND = kreq_has_mask(nd, vi, nv);
    u64 options = 0;
    if (t.type == 0) {                                        // spread
      i32 best = INT32_MAX; int bestv = -1;
      for (u64 bits = t.reg & ND; bits; bits &= bits - 1) {
        const int d = __builtin_ctzll(bits);
        i32 cnt = S.gcnt[(size_t)t.g * 64 + d] + t.self;
        if ((i64)cnt - (i64)t.minc <= (i64)t.maxskew && cnt < best) { best = cnt; bestv = d; }
      }
      if (bestv >= 0) options = 1ull << bestv;
    } else if (t.type == 1) {                                 // affinity
      options = t.reg & t.PD & t.pos;
      if (!options && t.self) {
        KReq pd; pd.present = true; pd.complement = true; pd.mask = 0; pd.gt = KS_NOGT; pd.lt = KS_NOLT;
        if (t.pod_has) pd = load_req(c.present, c.complement, c.mask, c.gt, c.lt, k);
        const u64 I = kreq_has_mask(kreq_intersect(pd, nd, vi, nv), vi, nv);
        const u64 a = t.reg & I, b = t.reg & t.PD;
        if (a) options |= a & (~a + 1);
        if (b) options |= b & (~b + 1);
      }
    } else {                                                  // anti-affinity
      options = t.reg & t.PD & ~t.pos;
    }
    if (!options) return 0;                                   // "unsatisfiable topology constraint"
    if ((domset >> e) & 1u) dom[e] &= options; else { dom[e] = options; domset |= 1u << e; }
    topo_keys |= 1u << e;
  }
  if (!host_ok) return 0;
  u32 narrowed = 0;
  for (int e = 0; e < nt; ++e) if ((topo_keys >> e) & 1u) {
    const int k = tkey[e]; const i32* vi = P.value_int + k * 64; const u32 nv = P.key_nvalues[k];
    const KReq before = treq[e];
    KReq in = kreq_in(dom[e]);
    // nodeRequirements.Compatible(topologyRequirements) on this key: the topology requirement is
    // node ∩ In[options]; see DESIGN.md "topology compatibility" for the reduction used here.
    if (!before.present) {
      if (!((P.wellknown_mask >> k) & 1u)) return 0;          // custom key the node does not define
      treq[e] = in;
    } else {
      KReq merged = kreq_intersect(in, before, vi, nv);
      if (kreq_len0(merged) && !kreq_nidne(before)) return 0;
      treq[e] = merged;
    }
    if (treq[e].mask != before.mask || treq[e].complement != before.complement || treq[e].present != before.present) narrowed |= 1u << k;
  }
  if (out) {
    out->present = v.present; out->complement = v.complement; out->it_state = it_state; out->changed = 0; out->topo_narrowed = narrowed;
    for (u32 k = 0; k < P.K; ++k) { out->mask[k] = v.mask[k]; out->gt[k] = v.gt[k]; out->lt[k] = v.lt[k]; }
    for (int e = 0; e < nt; ++e) {
      const int k = tkey[e]; const KReq& r = treq[e];
      if (!r.present) continue;
      const bool was = (v.present >> k) & 1u;
      if (!was || r.mask != v.mask[k] || r.complement != (bool)((v.complement >> k) & 1u) || r.gt != v.gt[k] || r.lt != v.lt[k]) out->changed |= 1u << k;
      out->present |= 1u << k; out->complement = r.complement ? (out->complement | (1u << k)) : (out->complement & ~(1u << k));
      out->mask[k] = r.mask; out->gt[k] = r.gt; out->lt[k] = r.lt;
    }
    (void)n_own;
  }
  // new nodes: necessary resource screen against the per-resource maximum over the surviving types
  if (!v.existing && !v.fresh) {
    for (u32 bits = c.reqmask; bits; bits &= bits - 1) { const int r = __builtin_ctz(bits); if (v.req[r] + c.req[r] > v.cap[r]) return 1; }
  }
  return 2;
}

// T-bit mask word of the types that pass `instanceType.Requirements.Intersects` on key k against node
// requirement B (derivation in DESIGN.md): types lacking the key always pass.
__device__ __forceinline__ u64 pass_types_word(const DevProb& P, int k, const KReq& B, u32 w) {
  const i32* vi = P.value_int + k * 64; const u32 nv = P.key_nvalues[k];
  u64 acc = 0;
  for (u64 bits = kreq_has_mask(B, vi, nv); bits; bits &= bits - 1) acc |= P.kv_types[((size_t)k * 64 + __builtin_ctzll(bits)) * P.TW + w];
  if (B.complement) acc |= P.cmplx_types[(size_t)k * P.TW + w];
  if (kreq_nidne(B)) acc |= P.nidnex_types[(size_t)k * P.TW + w];
  return acc;
}
// hasOffering (node.go:151-159) as a T-bit mask word
__device__ __forceinline__ u64 offer_types_word(const DevProb& P, const ReqOut& rq, u32 w) {
  u64 allowZ = ~0ull, allowC = ~0ull;
  if (P.key_zone >= 0 && ((rq.present >> P.key_zone) & 1u)) { KReq z = load_req(rq.present, rq.complement, rq.mask, rq.gt, rq.lt, P.key_zone); allowZ = kreq_has_mask(z, P.value_int + P.key_zone * 64, P.key_nvalues[P.key_zone]); }
  if (P.key_ct >= 0 && ((rq.present >> P.key_ct) & 1u)) { KReq z = load_req(rq.present, rq.complement, rq.mask, rq.gt, rq.lt, P.key_ct); allowC = kreq_has_mask(z, P.value_int + P.key_ct * 64, P.key_nvalues[P.key_ct]); }
  if (P.n_ct == 0) return ~0ull;
  u64 acc = 0; const u64 cm = allowC & ((1ull << P.n_ct) - 1);
  for (u64 zz = allowZ; zz; zz &= zz - 1) {
    const int z = __builtin_ctzll(zz); if ((u32)z * P.n_ct >= 64) break;
    for (u64 cb = cm; cb; cb &= cb - 1) acc |= P.pair_types[((size_t)z * P.n_ct + __builtin_ctzll(cb)) * P.TW + w];
  }
  return acc;
}

// TopologyNodeFilter.MatchesRequirements, topologynodefilter.go:57-70
__device__ bool filter_matches(const DevProb& P, int g, const ReqOut& rq) {
  const u32 b = P.grp_filter_off[g], e = P.grp_filter_off[g + 1];
  if (b == e) return true;
  for (u32 f = b; f < e; ++f) {
    bool ok = true;
    const u32 fp = P.flt.present[f], fc = P.flt.complement[f];
    for (u32 bits = fp; bits && ok; bits &= bits - 1) {
      const int k = __builtin_ctz(bits);
      KReq a = load_req(rq.present, rq.complement, rq.mask, rq.gt, rq.lt, k);
      KReq in = load_req(fp, fc, P.flt.mask + (size_t)f * P.K, P.flt.gt + (size_t)f * P.K, P.flt.lt + (size_t)f * P.K, k);
      if (kreq_compatible_fail(a, in, (P.wellknown_mask >> k) & 1u, P.value_int + k * 64, P.key_nvalues[k])) ok = false;
    }
    if (ok && P.flt.it_state[f] && P.its_fail[rq.it_state * P.S + P.flt.it_state[f]]) ok = false;
    if (ok) return true;
  }
  return false;
}

__device__ __forceinline__ void grp_record(const DevProb& P, const DevState& S, int g, int d) {   // TopologyGroup.Record, topologygroup.go:101-105
  i32& c = S.gcnt[(size_t)g * 64 + d]; c = c < 0 ? 1 : c + 1; S.g_reg[g] |= 1ull << d; S.g_pos[g] |= 1ull << d;
}
__device__ __forceinline__ void grp_record_host(const DevProb& P, const DevState& S, int g, u32 slot) {
  const i32 h = P.grp_hslot[g]; i32& c = S.hcnt[(size_t)h * (P.E + P.NMAX) + slot];
  if (c <= 0) S.g_hpos[h]++;
  c = c < 0 ? 1 : c + 1;
}
// Topology.Record, topology.go:120-143 (lane 0)
__device__ void topology_record(const DevProb& P, const DevState& S, const ClsL& c, const ReqOut& rq, u32 slot) {
  for (u32 i = P.cls_sel_off[c.c]; i < P.cls_sel_off[c.c + 1]; ++i) {
    const int g = P.sel_list[i];
    if (!S.g_active[g]) continue;
    if (!filter_matches(P, g, rq)) continue;                       // TopologyGroup.Counts, topologygroup.go:109-111
    const i32 k = P.grp_key[g];
    if (k == KS_KEY_HOSTNAME) { grp_record_host(P, S, g, slot); continue; }   // node requirement is `hostname In [own]`
    if (!((rq.present >> k) & 1u)) continue;                       // Get() of a missing key is Exists: no values, Len != 1
    const bool comp = (rq.complement >> k) & 1u; const u64 m = rq.mask[k];
    if (P.grp_type[g] == 2) { for (u64 b = m; b; b &= b - 1) grp_record(P, S, g, __builtin_ctzll(b)); }     // Values(): for a complement set the excluded values
    else if (!comp && __builtin_popcountll(m) == 1) grp_record(P, S, g, __builtin_ctzll(m));
  }
  for (u32 i = P.cls_iown_off[c.c]; i < P.cls_iown_off[c.c + 1]; ++i) {
    const int g = P.iown_list[i]; const i32 k = P.grp_key[g];
    if (k == KS_KEY_HOSTNAME) { grp_record_host(P, S, g, slot); continue; }
    if (!((rq.present >> k) & 1u)) continue;
    for (u64 b = rq.mask[k]; b; b &= b - 1) grp_record(P, S, g, __builtin_ctzll(b));
  }
}

struct PackShared {
  ClsL cls; ReqOut rq;
  u64 red_key[KS_NT / 64]; u32 red_slot[KS_NT / 64];
  i64 red_i64[KS_NT / 64][KS_MAX_RES];
  u32 any_flag; u32 pod; i32 bcast_i; u64 bcast_key; u32 bcast_slot; u32 limit_any;
  u64 floor_key;
};

// block-wide arg-min of (key, slot); every thread gets the result
__device__ __forceinline__ void block_argmin(PackShared& sh, u64& key, u32& slot) {
  for (int off = 32; off > 0; off >>= 1) {
    const u64 ok = __shfl_xor(key, off); const u32 os = __shfl_xor(slot, off);
    if (ok < key) { key = ok; slot = os; }
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) { sh.red_key[wave] = key; sh.red_slot[wave] = slot; }
  __syncthreads();
  if (threadIdx.x == 0) {
    u64 bk = sh.red_key[0]; u32 bs = sh.red_slot[0];
    for (int i = 1; i < KS_NT / 64; ++i) if (sh.red_key[i] < bk) { bk = sh.red_key[i]; bs = sh.red_slot[i]; }
    sh.bcast_key = bk; sh.bcast_slot = bs;
  }
  __syncthreads();
  key = sh.bcast_key; slot = sh.bcast_slot;
  __syncthreads();
}

// Load the pod's class into LDS and pre-evaluate the per-pod part of every matching topology group
// (getMatchingTopologies, topology.go:351-364; domainMinCount, topologygroup.go:184-200).
__device__ void stage_class(const DevProb& P, const DevState& S, PackShared& sh, u32 c) {
  ClsL& L = sh.cls;
  if (threadIdx.x < P.K) { const u32 k = threadIdx.x; L.mask[k] = P.cls.mask[(size_t)c * P.K + k]; L.gt[k] = P.cls.gt[(size_t)c * P.K + k]; L.lt[k] = P.cls.lt[(size_t)c * P.K + k]; }
  if (threadIdx.x >= 64 && threadIdx.x < 64 + P.R) { const u32 r = threadIdx.x - 64; L.req[r] = P.cls_requests[(size_t)c * P.R + r]; }
  if (threadIdx.x == 128) {
    L.c = c; L.present = P.cls.present[c]; L.complement = P.cls.complement[c]; L.it_state = P.cls.it_state[c];
    L.hn_mode = P.cls_hn_mode[c]; L.hn_off = P.cls_hn_off[c]; L.hn_cnt = P.cls_hn_off[c + 1] - P.cls_hn_off[c];
    L.reqmask = P.cls_requests_present[c]; L.tol = P.cls_tolerated[c]; L.port_off = P.cls_port_off[c]; L.port_cnt = P.cls_port_off[c + 1] - P.cls_port_off[c];
  }
  __syncthreads();
  // topology items: owned groups first, then inverse groups selecting the pod
  const u32 ob = P.cls_own_off[c], oe = P.cls_own_off[c + 1], ib = P.cls_isel_off[c], ie = P.cls_isel_off[c + 1];
  const u32 n = (oe - ob) + (ie - ib);
  if (threadIdx.x < n && threadIdx.x < KS_MAX_TOPO) {
    const u32 i = threadIdx.x; TopoItem t;
    u32 ent; if (i < oe - ob) { ent = P.own_list[ob + i]; t.inverse = 0; } else { ent = P.isel_list[ib + (i - (oe - ob))]; t.inverse = 1; }
    t.g = ent & 0x7FFFFFFFu; t.self = ent >> 31; t.type = P.grp_type[t.g]; t.key = P.grp_key[t.g]; t.hslot = P.grp_hslot[t.g]; t.maxskew = P.grp_max_skew[t.g];
    t.minc = 0; t.PD = ~0ull; t.pod_has = 0; t.reg = 0; t.pos = 0;
    if (t.key >= 0) {
      const int k = t.key; KReq pd = kreq_exists();
      if ((L.present >> k) & 1u) { pd = load_req(L.present, L.complement, L.mask, L.gt, L.lt, k); t.pod_has = 1; }
      t.PD = kreq_has_mask(pd, P.value_int + k * 64, P.key_nvalues[k]);
      t.reg = S.g_reg[t.g]; t.pos = S.g_pos[t.g];
      i32 mn = INT32_MAX;
      for (u64 b = t.reg & t.PD; b; b &= b - 1) { const i32 cn = S.gcnt[(size_t)t.g * 64 + __builtin_ctzll(b)]; if (cn < mn) mn = cn; }
      t.minc = mn;
    }
    L.topo[i] = t;
  }
  if (threadIdx.x == 0) L.ntopo = n < KS_MAX_TOPO ? (int)n : KS_MAX_TOPO;
  __syncthreads();
}

// Instance-type filter for the node described by (alive words, requests) after adding the pod:
// alive' = alive & passTypes(changed keys) & offerings & fits.  Block-wide; returns any(alive').
// `alive_out` (global, TW words) receives alive'.  Also reduces the new per-resource maxima of
// Allocatable over alive' into sh.red_i64 (used as the resource screen of later pods).
__device__ bool filter_types(const DevProb& P, PackShared& sh, const u64* alive_in, u64* alive_out, const i64* req_new, u32 reqmask_new,
                             u32 changed_keys, bool check_offer, bool check_it, i64* cap_out) {
  const ReqOut& rq = sh.rq;
  __shared__ u64 words[KS_NT];      // up to 1024 words == 65536 types per pass
  const u32 tid = threadIdx.x;
  if (tid == 0) sh.any_flag = 0;
  __syncthreads();
  i64 mx[KS_MAX_RES];
  for (u32 r = 0; r < P.R; ++r) mx[r] = INT64_MIN;
  bool any = false;
  for (u32 wbase = 0; wbase < P.TW; wbase += KS_NT) {
    // step 1: requirement / offering masks, one word per thread
    const u32 w = wbase + tid; u64 a = 0;
    if (w < P.TW) {
      a = alive_in[w];
      for (u32 bits = changed_keys; bits && a; bits &= bits - 1) { const int k = __builtin_ctz(bits); a &= pass_types_word(P, k, load_req(rq.present, rq.complement, rq.mask, rq.gt, rq.lt, k), w); }
      if (check_it && a) a &= P.its_types[(size_t)rq.it_state * P.TW + w];
      if (check_offer && a) a &= offer_types_word(P, rq, w);
    }
    words[tid] = a;
    __syncthreads();
    // step 2: fits, one type per lane
    const u32 nwords = min((u32)KS_NT, P.TW - wbase);
    for (u32 tl = tid; tl < nwords * 64; tl += KS_NT) {
      const u32 wl = tl >> 6; const u64 aw = words[wl]; bool ok = (aw >> (tl & 63)) & 1ull;
      const u32 t = (wbase + wl) * 64 + (tl & 63);
      if (ok) { for (u32 bits = reqmask_new; bits; bits &= bits - 1) { const int r = __builtin_ctz(bits); if (req_new[r] > P.it_alloc[(size_t)r * P.T + t]) { ok = false; break; } } }
      if (ok) for (u32 r = 0; r < P.R; ++r) { const i64 al = P.it_alloc[(size_t)r * P.T + t]; if (al > mx[r]) mx[r] = al; }
      const u64 bw = ballot64(ok);
      if ((tl & 63) == 0) { alive_out[wbase + wl] = bw; if (bw) any = true; }
    }
    __syncthreads();
  }
  if (any) sh.any_flag = 1;
  // reduce maxima
  for (u32 r = 0; r < P.R; ++r) { i64 v = mx[r]; for (int off = 32; off > 0; off >>= 1) { const i64 o = __shfl_xor(v, off); if (o > v) v = o; } if ((tid & 63) == 0) sh.red_i64[tid >> 6][r] = v; }
  __syncthreads();
  if (tid < P.R) { i64 v = sh.red_i64[0][tid]; for (int i = 1; i < KS_NT / 64; ++i) if (sh.red_i64[i][tid] > v) v = sh.red_i64[i][tid]; cap_out[tid] = v; }
  __syncthreads();
  return sh.any_flag != 0;
}

__global__ __launch_bounds__(KS_NT) void ks_pack(const DevProb* probs, const DevState* states) {
  const DevProb& P = probs[blockIdx.x];
  const DevState& S = states[blockIdx.x];
  __shared__ PackShared sh;
  __shared__ i64 req_new[KS_MAX_RES];
  __shared__ i64 cap_new[KS_MAX_RES];
  __shared__ u32 s_head, s_len, s_gen, s_nnew, s_seq, s_stop, s_err, s_reqmask_new;
  const u32 tid = threadIdx.x;
  const u32 NS = P.E + P.NMAX;
  const u64 t_start = __builtin_readcyclecounter();

  // ---------------- initialise state ----------------
  for (u32 i = tid; i < P.P; i += KS_NT) { S.q[i] = P.queue[i]; S.lastgen[i] = 0xFFFFFFFFu; S.lastlen[i] = 0; S.pod_stage[i] = 0; S.pod_node[i] = -1; S.pod_seq[i] = -1; }
  for (u32 e = tid; e < P.E; e += KS_NT) {
    S.s_present[e] = P.en.present[e]; S.s_complement[e] = P.en.complement[e]; S.s_it[e] = P.en.it_state[e];
    for (u32 k = 0; k < P.K; ++k) { S.s_mask[(size_t)e * P.K + k] = P.en.mask[(size_t)e * P.K + k]; S.s_gt[(size_t)e * P.K + k] = P.en.gt[(size_t)e * P.K + k]; S.s_lt[(size_t)e * P.K + k] = P.en.lt[(size_t)e * P.K + k]; }
    for (u32 r = 0; r < P.R; ++r) { S.s_req[(size_t)e * P.R + r] = P.en_requests[(size_t)e * P.R + r]; S.s_cap[(size_t)e * P.R + r] = P.en_avail[(size_t)e * P.R + r]; }
    S.s_reqmask[e] = P.en_requests_present[e]; S.s_taints[e] = P.en_taints[e];
    // existing host ports: chain the node's initial entries
    i32 head = -1; for (u32 i = P.en_port_off[e]; i < P.en_port_off[e + 1]; ++i) { S.pp_entry[i] = P.ports[i]; S.pp_next[i] = head; head = (i32)i; }
    S.s_porthead[e] = head;
  }
  for (u32 i = tid; i < P.G * 64; i += KS_NT) S.gcnt[i] = P.grp_count[i];
  for (u32 g = tid; g < P.G; g += KS_NT) {
    u64 reg = 0, pos = 0; for (int d = 0; d < 64; ++d) { const i32 c = P.grp_count[(size_t)g * 64 + d]; if (c >= 0) reg |= 1ull << d; if (c > 0) pos |= 1ull << d; }
    S.g_reg[g] = reg; S.g_pos[g] = pos; S.g_active[g] = P.grp_active[g];
  }
  for (u32 h = tid; h < P.GH; h += KS_NT) {
    i32 np = P.grph_extra_pos[h];
    for (u32 e = 0; e < P.E; ++e) { const i32 c = P.grph_count[(size_t)h * P.E + e]; S.hcnt[(size_t)h * NS + e] = c; if (c > 0) ++np; }
    S.g_hpos[h] = np;
  }
  for (u32 i = tid; i < P.M * P.R; i += KS_NT) S.remaining[i] = P.tmpl_remaining[i];
  if (tid == 0) { s_head = 0; s_len = P.P; s_gen = 0; s_nnew = 0; s_seq = 0; s_stop = 0; s_err = 0; for (int i = 0; i < 16; ++i) S.stats[i] = 0; }
  __syncthreads();
  u32 pp_used = P.E ? P.en_port_off[P.E] : 0;     // uniform across threads
  u64 st_pops = 0, st_relax = 0, st_full = 0, st_fullfail = 0, st_ref_attempts = 0, st_ref_types = 0;   // lane-0 counters
  const bool want_stats = (P.flags & KS_FLAG_STATS) != 0;

  // ---------------- Solve loop, scheduler.go:104-124 ----------------
  for (;;) {
    // Queue.Pop, queue.go:44-58
    if (tid == 0) {
      if (s_len == 0) s_stop = 1;
      else {
        const u32 p = S.q[s_head];
        if (S.lastgen[p] == s_gen && S.lastlen[p] == s_len) s_stop = 1;
        else { sh.pod = p; s_head = (s_head + 1 == P.P) ? 0 : s_head + 1; s_len--; ++st_pops; }
      }
    }
    __syncthreads();
    if (s_stop || s_err) break;
    const u32 pod = sh.pod;
    const u32 cidx = P.stage_cls[P.pod_stage_off[pod] + S.pod_stage[pod]];
    stage_class(P, S, sh, cidx);
    const ClsL& c = sh.cls;
    bool placed = false;

    // ---- 1. existing nodes in the caller's order: first success wins (scheduler.go:176-180) ----
    if (P.E) {
      u64 key = ~0ull; u32 slot = 0xFFFFFFFFu;
      for (u32 e = tid; e < P.E; e += KS_NT) {
        NodeView v = slot_view(P, S, e);
        if (eval_node(P, S, c, v, nullptr) == 2) { if ((u64)e < key) { key = e; slot = e; } }
      }
      block_argmin(sh, key, slot);
      if (want_stats && tid == 0) st_ref_attempts += (slot == 0xFFFFFFFFu) ? P.E : slot + 1;
      if (slot != 0xFFFFFFFFu) {
        if (tid == 0) {
          NodeView v = slot_view(P, S, slot);
          eval_node(P, S, c, v, &sh.rq);
          // commit, existingnode.go:122-129
          S.s_present[slot] = sh.rq.present; S.s_complement[slot] = sh.rq.complement; S.s_it[slot] = sh.rq.it_state;
          for (u32 k = 0; k < P.K; ++k) { S.s_mask[(size_t)slot * P.K + k] = sh.rq.mask[k]; S.s_gt[(size_t)slot * P.K + k] = sh.rq.gt[k]; S.s_lt[(size_t)slot * P.K + k] = sh.rq.lt[k]; }
          for (u32 r = 0; r < P.R; ++r) S.s_req[(size_t)slot * P.R + r] += c.req[r];
          S.s_reqmask[slot] |= c.reqmask;
          topology_record(P, S, c, sh.rq, slot);
          for (u32 i = 0; i < c.port_cnt; ++i) { S.pp_entry[pp_used + i] = P.ports[c.port_off + i]; S.pp_next[pp_used + i] = S.s_porthead[slot]; S.s_porthead[slot] = (i32)(pp_used + i); }
          S.pod_node[pod] = (i32)slot; S.pod_seq[pod] = (i32)s_seq++;
        }
        pp_used += c.port_cnt;
        placed = true;
        __threadfence_block();
        __syncthreads();
      }
    }

    // ---- 2. open new nodes in `sort.Slice(newNodes, len(Pods))` order (scheduler.go:183-190) ----
    if (!placed && s_nnew) {
      u64 floor = 0;   // candidates must have key > floor (keys are unique); 0 is below every key
      for (;;) {
        u64 key = ~0ull; u32 slot = 0xFFFFFFFFu; u64 my_types = 0; u32 my_attempts = 0;
        const u32 nn = s_nnew;
        for (u32 j = tid; j < nn; j += KS_NT) {
          const u64 kk = S.n_key[j];
          if (kk <= floor && floor) continue;
          NodeView v = slot_view(P, S, P.E + j);
          const int rc = eval_node(P, S, c, v, nullptr);
          if (rc == 2 && kk < key) { key = kk; slot = P.E + j; }
        }
        block_argmin(sh, key, slot);
        if (slot == 0xFFFFFFFFu) break;
        // full check of the candidate: instance-type filter
        const u32 j = slot - P.E;
        if (tid == 0) {
          NodeView v = slot_view(P, S, slot);
          eval_node(P, S, c, v, &sh.rq);
          u32 rm = v.reqmask | c.reqmask; s_reqmask_new = rm;
          for (u32 r = 0; r < P.R; ++r) req_new[r] = v.req[r] + c.req[r];
          ++st_full;
        }
        __syncthreads();
        const bool zc = (P.key_zone >= 0 && ((sh.rq.changed >> P.key_zone) & 1u)) || (P.key_ct >= 0 && ((sh.rq.changed >> P.key_ct) & 1u));
        const bool itc = sh.rq.it_state != S.s_it[slot];
        u64* alive = S.n_alive + (size_t)j * P.TW;
        u64* scratch = S.n_alive + (size_t)P.NMAX * P.TW;     // one spare row
        const bool ok = filter_types(P, sh, alive, scratch, req_new, s_reqmask_new, sh.rq.changed, zc, itc, cap_new);
        if (ok) {
          // commit, node.go:100-105
          for (u32 w = tid; w < P.TW; w += KS_NT) alive[w] = scratch[w];
          if (tid == 0) {
            S.s_present[slot] = sh.rq.present; S.s_complement[slot] = sh.rq.complement; S.s_it[slot] = sh.rq.it_state;
            for (u32 k = 0; k < P.K; ++k) { S.s_mask[(size_t)slot * P.K + k] = sh.rq.mask[k]; S.s_gt[(size_t)slot * P.K + k] = sh.rq.gt[k]; S.s_lt[(size_t)slot * P.K + k] = sh.rq.lt[k]; }
            for (u32 r = 0; r < P.R; ++r) { S.s_req[(size_t)slot * P.R + r] = req_new[r]; S.s_cap[(size_t)slot * P.R + r] = cap_new[r]; }
            S.s_reqmask[slot] = s_reqmask_new;
            const u32 cnt = ++S.n_count[j];
            const u32 seq = s_seq++;
            S.n_key[j] = ((u64)cnt << 32) | (u64)(0xFFFFFFFFu - seq);     // moved to the FRONT of the next count bucket (stable sort)
            topology_record(P, S, c, sh.rq, slot);
            for (u32 i = 0; i < c.port_cnt; ++i) { S.pp_entry[pp_used + i] = P.ports[c.port_off + i]; S.pp_next[pp_used + i] = S.s_porthead[slot]; S.s_porthead[slot] = (i32)(pp_used + i); }
            S.pod_node[pod] = (i32)slot; S.pod_seq[pod] = (i32)seq;
          }
          pp_used += c.port_cnt;
          placed = true;
          __threadfence_block();
          __syncthreads();
          break;
        }
        if (tid == 0) ++st_fullfail;
        floor = key;       // next candidate strictly after this one in visiting order
        __syncthreads();
        (void)my_types; (void)my_attempts;
      }
    }

    // ---- 3. a new node from the first template that works (scheduler.go:193-217) ----
    if (!placed) {
      for (u32 m = 0; m < P.M && !placed; ++m) {
        const size_t mc = (size_t)m * P.C + cidx;
        if (!P.mc_ok[mc]) continue;
        if (s_nnew >= P.NMAX) { if (tid == 0) s_err = (u32)(-KS_ERR_CAPACITY); __syncthreads(); break; }
        const u32 j = s_nnew; const u32 slot = P.E + j;
        u64* alive = S.n_alive + (size_t)j * P.TW;
        u64* scratch = S.n_alive + (size_t)P.NMAX * P.TW;
        const u32 lim = P.tmpl_limit_present[m];
        // filterByRemainingResources, scheduler.go:293-309 (only when the provisioner has limits)
        if (tid == 0) sh.limit_any = 0;
        __syncthreads();
        bool lany = false;
        for (u32 tl = tid; tl < P.TW * 64; tl += KS_NT) {
          const u32 t = tl; bool ok = t < P.T && ((P.tmpl_types[(size_t)m * P.TW + (t >> 6)] >> (t & 63)) & 1ull);
          if (ok && lim != 0xFFFFFFFFu) for (u32 bits = lim; bits; bits &= bits - 1) { const int r = __builtin_ctz(bits); if (P.it_cap[(size_t)r * P.T + t] > S.remaining[(size_t)m * P.R + r]) { ok = false; break; } }
          const u64 bw = ballot64(ok);
          if ((tl & 63) == 0) { scratch[tl >> 6] = bw & P.grid[mc * P.TW + (tl >> 6)]; if (bw) lany = true; }
        }
        if (lany) sh.limit_any = 1;
        __syncthreads();
        if (!sh.limit_any) continue;            // "all available instance types exceed provisioner limits"
        // NewNode + Node.Add on the fresh node
        if (tid == 0) {
          NodeView v; v.present = P.mc_present[mc]; v.complement = P.mc_complement[mc]; v.it_state = P.mc_it[mc];
          v.mask = P.mc_mask + mc * P.K; v.gt = P.mc_gt + mc * P.K; v.lt = P.mc_lt + mc * P.K;
          v.taints = 0; v.porthead = -1; v.req = P.tmpl_daemon + (size_t)m * P.R; v.reqmask = P.tmpl_daemon_present[m]; v.cap = nullptr; v.slot = (i32)slot; v.existing = false; v.fresh = true;
          // the class's own requirements are already merged into mc_*: evaluate topology on top of them
          const int rc = eval_node(P, S, c, v, &sh.rq, true);
          sh.bcast_i = rc;
          s_reqmask_new = P.tmpl_daemon_present[m] | c.reqmask;
          for (u32 r = 0; r < P.R; ++r) req_new[r] = P.tmpl_daemon[(size_t)m * P.R + r] + c.req[r];
          ++st_full;
          if (want_stats) { st_ref_attempts += 1; }
        }
        __syncthreads();
        if (sh.bcast_i != 2) continue;
        // topology may have narrowed keys beyond template∩class: re-filter those keys (+ offerings)
        const u32 nk = sh.rq.topo_narrowed;
        const bool zc = (P.key_zone >= 0 && ((nk >> P.key_zone) & 1u)) || (P.key_ct >= 0 && ((nk >> P.key_ct) & 1u));
        const bool ok = filter_types(P, sh, scratch, alive, req_new, s_reqmask_new, nk, zc, false, cap_new);
        if (!ok) { if (tid == 0) ++st_fullfail; continue; }
        // commit the new node (scheduler.go:214-216)
        for (u32 h = tid; h < P.G; h += KS_NT) if (P.grp_hslot[h] >= 0) S.hcnt[(size_t)P.grp_hslot[h] * NS + slot] = S.g_active[h] ? 0 : -1;   // Topology.Register(hostname), node.go:47
        __syncthreads();
        // subtractMax, scheduler.go:273-290
        if (lim != 0xFFFFFFFFu) {
          i64 mx[KS_MAX_RES]; for (u32 r = 0; r < P.R; ++r) mx[r] = INT64_MIN;
          for (u32 t = tid; t < P.T; t += KS_NT) if ((alive[t >> 6] >> (t & 63)) & 1ull) for (u32 r = 0; r < P.R; ++r) { const i64 cp = P.it_cap[(size_t)r * P.T + t]; if (cp > mx[r]) mx[r] = cp; }
          for (u32 r = 0; r < P.R; ++r) { i64 v = mx[r]; for (int off = 32; off > 0; off >>= 1) { const i64 o = __shfl_xor(v, off); if (o > v) v = o; } if ((tid & 63) == 0) sh.red_i64[tid >> 6][r] = v; }
          __syncthreads();
          if (tid < P.R && ((lim >> tid) & 1u)) { i64 v = sh.red_i64[0][tid]; for (int i = 1; i < KS_NT / 64; ++i) if (sh.red_i64[i][tid] > v) v = sh.red_i64[i][tid]; S.remaining[(size_t)m * P.R + tid] -= v; }
          __syncthreads();
        }
        if (tid == 0) {
          S.s_present[slot] = sh.rq.present; S.s_complement[slot] = sh.rq.complement; S.s_it[slot] = sh.rq.it_state;
          for (u32 k = 0; k < P.K; ++k) { S.s_mask[(size_t)slot * P.K + k] = sh.rq.mask[k]; S.s_gt[(size_t)slot * P.K + k] = sh.rq.gt[k]; S.s_lt[(size_t)slot * P.K + k] = sh.rq.lt[k]; }
          for (u32 r = 0; r < P.R; ++r) { S.s_req[(size_t)slot * P.R + r] = req_new[r]; S.s_cap[(size_t)slot * P.R + r] = cap_new[r]; }
          S.s_reqmask[slot] = s_reqmask_new; S.s_taints[slot] = P.tmpl_taints[m]; S.s_porthead[slot] = -1;
          S.n_tmpl[j] = (i32)m; S.n_count[j] = 1;
          const u32 seq = s_seq++;
          S.n_key[j] = (1ull << 32) | (u64)seq;                         // appended: BACK of the count-1 bucket
          topology_record(P, S, c, sh.rq, slot);
          for (u32 i = 0; i < c.port_cnt; ++i) { S.pp_entry[pp_used + i] = P.ports[c.port_off + i]; S.pp_next[pp_used + i] = S.s_porthead[slot]; S.s_porthead[slot] = (i32)(pp_used + i); }
          S.pod_node[pod] = (i32)slot; S.pod_seq[pod] = (i32)seq;
          s_nnew = j + 1;
        }
        pp_used += c.port_cnt;
        placed = true;
        __threadfence_block();
        __syncthreads();
      }
    }

    // ---- 4. failure: Preferences.Relax + Queue.Push + Topology.Update (scheduler.go:116-123) ----
    if (!placed && !s_err) {
      if (tid == 0) {
        const u32 nst = P.pod_stage_off[pod + 1] - P.pod_stage_off[pod];
        const bool relaxed = (u32)S.pod_stage[pod] + 1 < nst;
        u32 tail = s_head + s_len; if (tail >= P.P) tail -= P.P;
        S.q[tail] = pod; s_len++;
        if (relaxed) {
          S.pod_stage[pod]++; s_gen++; ++st_relax;
          const u32 nc = P.stage_cls[P.pod_stage_off[pod] + S.pod_stage[pod]];
          for (u32 i = P.cls_own_off[nc]; i < P.cls_own_off[nc + 1]; ++i) S.g_active[P.own_list[i] & 0x7FFFFFFFu] = 1;   // Topology.Update creates the group
        } else { S.lastlen[pod] = s_len; S.lastgen[pod] = s_gen; }
      }
      __threadfence_block();
      __syncthreads();
    }
  }

  // ---------------- results ----------------
  __syncthreads();
  for (u32 i = tid; i < s_len; i += KS_NT) { u32 idx = s_head + i; if (idx >= P.P) idx -= P.P; S.unscheduled[i] = (i32)S.q[idx]; }
  if (tid == 0) {
    S.out_counts[0] = s_nnew; S.out_counts[1] = s_len;
    S.stats[KS_STAT_POPS] = st_pops; S.stats[KS_STAT_RELAX] = st_relax; S.stats[KS_STAT_FULLCHECKS] = st_full; S.stats[KS_STAT_FULLFAILS] = st_fullfail;
    S.stats[KS_STAT_REF_ATTEMPTS] = st_ref_attempts; S.stats[KS_STAT_REF_TYPES] = st_ref_types;
    S.stats[KS_STAT_CYCLES] = __builtin_readcyclecounter() - t_start; S.stats[KS_STAT_ERR] = s_err;
  }
}

// ------------------------------------------------------------------------------------------------
// probe kernels (truth-table checks on the device)
// ------------------------------------------------------------------------------------------------
__global__ void ks_probe_kernel(ks_req1 a, ks_req1 b, const i32* vint, u32 nv, int wk, ks_req1* out, int* okout) {
  KReq A{a.mask, a.gt, a.lt, (bool)a.present, (bool)a.complement}, B{b.mask, b.gt, b.lt, (bool)b.present, (bool)b.complement};
  KReq r = kreq_intersect(A, B, vint, nv);
  out->mask = r.mask; out->gt = r.gt; out->lt = r.lt; out->present = r.present; out->complement = r.complement;
  *okout = !kreq_compatible_fail(A, B, wk != 0, vint, nv);
}

// ================================================================================================
// host side
// ================================================================================================
struct ks_dev_problem {
  int device = 0;
  DevProb h{};                     // host copy of the device view (pointers are device pointers)
  DevProb* d_prob = nullptr;       // the same struct in device memory (for ks_pack)
  DevState hs{}; DevState* d_state = nullptr;
  std::vector<void*> allocs;
  hipStream_t stream = nullptr;
  bool tables_built = false;
  u32 pp_cap = 0;
};

template <typename T> static int dev_copy(ks_dev_problem* d, const T* src, size_t n, const T** dst) {
  *dst = nullptr;
  size_t bytes = (n ? n : 1) * sizeof(T);
  void* p = nullptr;
  HIPCHK(hipMalloc(&p, bytes));
  d->allocs.push_back(p);
  if (n) { if (!src) return fail(KS_ERR_INVALID, "null array in ks_problem"); HIPCHK(hipMemcpy(p, src, n * sizeof(T), hipMemcpyHostToDevice)); }
  *dst = (const T*)p; return KS_OK;
}
template <typename T> static int dev_alloc(ks_dev_problem* d, size_t n, T** dst, int fill = -2) {
  size_t bytes = (n ? n : 1) * sizeof(T); void* p = nullptr;
  HIPCHK(hipMalloc(&p, bytes)); d->allocs.push_back(p);
  if (fill != -2) HIPCHK(hipMemset(p, fill, bytes));
  *dst = (T*)p; return KS_OK;
}
#define TRY(x) do { int rc_ = (x); if (rc_ != KS_OK) return rc_; } while (0)

static int copy_reqsets(ks_dev_problem* d, const ks_reqsets& s, u32 n, u32 K, ReqSetsD* out) {
  out->n = n;
  TRY(dev_copy(d, s.present, n, &out->present)); TRY(dev_copy(d, s.complement, n, &out->complement));
  TRY(dev_copy(d, s.mask, (size_t)n * K, &out->mask)); TRY(dev_copy(d, s.gt, (size_t)n * K, &out->gt)); TRY(dev_copy(d, s.lt, (size_t)n * K, &out->lt));
  TRY(dev_copy(d, s.it_state, n, &out->it_state)); return KS_OK;
}

extern "C" int ks_device_count(void) {
  int n = 0; if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  int ok = 0;
  for (int i = 0; i < n; ++i) { hipDeviceProp_t pr; if (hipGetDeviceProperties(&pr, i) == hipSuccess && std::strstr(pr.gcnArchName, "gfx950")) ++ok; }
  return ok;
}

static int validate(const ks_problem* p) {
  if (!p) return fail(KS_ERR_INVALID, "null problem");
  if (p->K > KS_MAX_KEYS) return fail(KS_ERR_UNSUPPORTED, "more than 32 narrow label keys");
  if (p->R > KS_MAX_RES || p->R < 3) return fail(KS_ERR_INVALID, "R must be in [3,8]");
  if (p->S == 0 || p->S > KS_MAX_ITSTATES) return fail(KS_ERR_INVALID, "S must be in [1,256]");
  if (p->M == 0) return fail(KS_ERR_INVALID, "no provisioners found");   // provisioner.go:278-280
  if (p->T == 0) return fail(KS_ERR_INVALID, "no instance types");
  if (p->max_new_nodes == 0 && p->P) return fail(KS_ERR_INVALID, "max_new_nodes == 0");
  for (u32 k = 0; k < p->K; ++k) if (p->key_nvalues[k] > 64) return fail(KS_ERR_UNSUPPORTED, "label key with more than 64 distinct values");
  if (p->key_zone >= 0 && p->key_ct >= 0 && (u64)p->key_nvalues[p->key_zone] * p->n_ct > 64) return fail(KS_ERR_UNSUPPORTED, "more than 64 zone x capacity-type pairs");
  return KS_OK;
}

extern "C" void ks_problem_free(ks_dev_problem* d) {
  if (!d) return;
  hipSetDevice(d->device);
  for (void* p : d->allocs) hipFree(p);
  if (d->stream) hipStreamDestroy(d->stream);
  delete d;
}

extern "C" int ks_problem_upload(const ks_problem* p, int device, ks_dev_problem** out) {
  *out = nullptr;
  TRY(validate(p));
  if (ks_device_count() <= 0) return fail(KS_ERR_DEVICE, "no gfx950 (MI355X) device visible; libksolve has no CPU path");
  HIPCHK(hipSetDevice(device));
  ks_dev_problem* d = new ks_dev_problem(); d->device = device;
  struct Guard { ks_dev_problem* d; bool ok = false; ~Guard() { if (!ok) ks_problem_free(d); } } guard{d};
  HIPCHK(hipStreamCreate(&d->stream));
  DevProb& h = d->h;
  h.P = p->P; h.C = p->C; h.T = p->T; h.TW = (p->T + 63) / 64; h.M = p->M; h.E = p->E; h.K = p->K; h.R = p->R; h.G = p->G; h.GH = p->GH; h.S = p->S;
  h.NMAX = p->max_new_nodes ? p->max_new_nodes : 1; h.flags = p->flags; h.n_topologies = p->n_topologies;
  h.wellknown_mask = p->wellknown_mask; h.key_zone = p->key_zone; h.key_ct = p->key_ct; h.n_ct = p->n_ct;
  const u32 K = h.K, R = h.R, T = h.T, TW = h.TW, C = h.C, M = h.M, E = h.E, G = h.G, P = h.P;
  TRY(dev_copy(d, p->key_nvalues, K, &h.key_nvalues)); TRY(dev_copy(d, p->value_int, (size_t)K * 64, &h.value_int));
  TRY(dev_copy(d, p->it_present, T, &h.it_present)); TRY(dev_copy(d, p->it_complement, T, &h.it_complement));
  TRY(dev_copy(d, p->it_mask, (size_t)K * T, &h.it_mask)); TRY(dev_copy(d, p->it_alloc, (size_t)R * T, &h.it_alloc));
  TRY(dev_copy(d, p->it_cap, (size_t)R * T, &h.it_cap)); TRY(dev_copy(d, p->it_offer, T, &h.it_offer));
  TRY(dev_copy(d, p->its_inter, (size_t)h.S * h.S, &h.its_inter)); TRY(dev_copy(d, p->its_fail, (size_t)h.S * h.S, &h.its_fail));
  TRY(dev_copy(d, p->its_nidne, h.S, &h.its_nidne)); TRY(dev_copy(d, p->its_types, (size_t)h.S * TW, &h.its_types));
  TRY(copy_reqsets(d, p->tmpl, M, K, &h.tmpl)); TRY(dev_copy(d, p->tmpl_taints, M, &h.tmpl_taints));
  TRY(dev_copy(d, p->tmpl_daemon, (size_t)M * R, &h.tmpl_daemon)); TRY(dev_copy(d, p->tmpl_daemon_present, M, &h.tmpl_daemon_present));
  TRY(dev_copy(d, p->tmpl_types, (size_t)M * TW, &h.tmpl_types)); TRY(dev_copy(d, p->tmpl_limit_present, M, &h.tmpl_limit_present));
  TRY(dev_copy(d, p->tmpl_remaining, (size_t)M * R, &h.tmpl_remaining));
  TRY(copy_reqsets(d, p->en, E, K, &h.en)); TRY(dev_copy(d, p->en_taints, E, &h.en_taints)); TRY(dev_copy(d, p->en_avail, (size_t)E * R, &h.en_avail));
  TRY(dev_copy(d, p->en_requests, (size_t)E * R, &h.en_requests)); TRY(dev_copy(d, p->en_requests_present, E, &h.en_requests_present));
  TRY(dev_copy(d, p->en_port_off, (size_t)E + 1, &h.en_port_off));
  TRY(copy_reqsets(d, p->cls, C, K, &h.cls)); TRY(dev_copy(d, p->cls_hn_mode, C, &h.cls_hn_mode)); TRY(dev_copy(d, p->cls_hn_off, (size_t)C + 1, &h.cls_hn_off));
  TRY(dev_copy(d, p->hn_list, C ? p->cls_hn_off[C] : 0, &h.hn_list));
  TRY(dev_copy(d, p->cls_requests, (size_t)C * R, &h.cls_requests)); TRY(dev_copy(d, p->cls_requests_present, C, &h.cls_requests_present));
  TRY(dev_copy(d, p->cls_tolerated, C, &h.cls_tolerated)); TRY(dev_copy(d, p->cls_port_off, (size_t)C + 1, &h.cls_port_off));
  const u32 nports_static = C ? p->cls_port_off[C] : (E ? p->en_port_off[E] : 0);
  TRY(dev_copy(d, p->ports, nports_static, &h.ports));
  TRY(dev_copy(d, p->cls_own_off, (size_t)C + 1, &h.cls_own_off)); TRY(dev_copy(d, p->own_list, C ? p->cls_own_off[C] : 0, &h.own_list));
  TRY(dev_copy(d, p->cls_sel_off, (size_t)C + 1, &h.cls_sel_off)); TRY(dev_copy(d, p->sel_list, C ? p->cls_sel_off[C] : 0, &h.sel_list));
  TRY(dev_copy(d, p->cls_isel_off, (size_t)C + 1, &h.cls_isel_off)); TRY(dev_copy(d, p->isel_list, C ? p->cls_isel_off[C] : 0, &h.isel_list));
  TRY(dev_copy(d, p->cls_iown_off, (size_t)C + 1, &h.cls_iown_off)); TRY(dev_copy(d, p->iown_list, C ? p->cls_iown_off[C] : 0, &h.iown_list));
  TRY(dev_copy(d, p->pod_stage_off, (size_t)P + 1, &h.pod_stage_off)); TRY(dev_copy(d, p->stage_cls, P ? p->pod_stage_off[P] : 0, &h.stage_cls));
  TRY(dev_copy(d, p->queue, P, &h.queue));
  TRY(dev_copy(d, p->grp_type, G, &h.grp_type)); TRY(dev_copy(d, p->grp_key, G, &h.grp_key)); TRY(dev_copy(d, p->grp_max_skew, G, &h.grp_max_skew));
  TRY(dev_copy(d, p->grp_active, G, &h.grp_active)); TRY(dev_copy(d, p->grp_filter_off, (size_t)G + 1, &h.grp_filter_off));
  TRY(copy_reqsets(d, p->flt, p->flt.n, K, &h.flt));
  TRY(dev_copy(d, p->grp_count, (size_t)G * 64, &h.grp_count)); TRY(dev_copy(d, p->grp_hslot, G, &h.grp_hslot));
  TRY(dev_copy(d, p->grph_count, (size_t)p->GH * E, &h.grph_count)); TRY(dev_copy(d, p->grph_extra_pos, p->GH, &h.grph_extra_pos));
  // derived tables
  TRY(dev_alloc(d, (size_t)K * 64 * TW, &h.kv_types, 0)); TRY(dev_alloc(d, (size_t)K * TW, &h.cmplx_types, 0)); TRY(dev_alloc(d, (size_t)K * TW, &h.nidnex_types, 0));
  TRY(dev_alloc(d, (size_t)64 * TW, &h.pair_types, 0));
  const size_t MC = (size_t)M * C;
  TRY(dev_alloc(d, MC, &h.mc_ok, 0)); TRY(dev_alloc(d, MC, &h.mc_present)); TRY(dev_alloc(d, MC, &h.mc_complement));
  TRY(dev_alloc(d, MC * K, &h.mc_mask)); TRY(dev_alloc(d, MC * K, &h.mc_gt)); TRY(dev_alloc(d, MC * K, &h.mc_lt)); TRY(dev_alloc(d, MC, &h.mc_it));
  TRY(dev_alloc(d, MC * TW, &h.grid, 0));
  // state
  DevState& s = d->hs; const size_t NS = (size_t)E + h.NMAX;
  TRY(dev_alloc(d, P, &s.q)); TRY(dev_alloc(d, P, &s.lastlen)); TRY(dev_alloc(d, P, &s.lastgen)); TRY(dev_alloc(d, P, &s.pod_stage)); TRY(dev_alloc(d, P, &s.pod_node)); TRY(dev_alloc(d, P, &s.pod_seq));
  TRY(dev_alloc(d, NS, &s.s_present)); TRY(dev_alloc(d, NS, &s.s_complement)); TRY(dev_alloc(d, NS * K, &s.s_mask)); TRY(dev_alloc(d, NS * K, &s.s_gt)); TRY(dev_alloc(d, NS * K, &s.s_lt)); TRY(dev_alloc(d, NS, &s.s_it));
  TRY(dev_alloc(d, NS * R, &s.s_req)); TRY(dev_alloc(d, NS, &s.s_reqmask)); TRY(dev_alloc(d, NS * R, &s.s_cap)); TRY(dev_alloc(d, NS, &s.s_taints)); TRY(dev_alloc(d, NS, &s.s_porthead));
  TRY(dev_alloc(d, (size_t)h.NMAX, &s.n_tmpl)); TRY(dev_alloc(d, (size_t)h.NMAX, &s.n_count)); TRY(dev_alloc(d, (size_t)h.NMAX, &s.n_key)); TRY(dev_alloc(d, ((size_t)h.NMAX + 1) * TW, &s.n_alive));
  TRY(dev_alloc(d, (size_t)G * 64, &s.gcnt)); TRY(dev_alloc(d, G, &s.g_reg)); TRY(dev_alloc(d, G, &s.g_pos)); TRY(dev_alloc(d, G, &s.g_active));
  TRY(dev_alloc(d, (size_t)p->GH * NS, &s.hcnt, 0xFF)); TRY(dev_alloc(d, p->GH, &s.g_hpos)); TRY(dev_alloc(d, (size_t)M * R, &s.remaining));
  // host-port pool: existing entries + one batch-worth of pod ports (max over stages)
  size_t pool = E ? p->en_port_off[E] : 0;
  for (u32 i = 0; i < P; ++i) { u32 mx = 0; for (u32 st = p->pod_stage_off[i]; st < p->pod_stage_off[i + 1]; ++st) { const u32 c = p->stage_cls[st]; const u32 n = p->cls_port_off[c + 1] - p->cls_port_off[c]; if (n > mx) mx = n; } pool += mx; }
  s.pp_cap = (u32)pool; TRY(dev_alloc(d, pool, &s.pp_entry)); TRY(dev_alloc(d, pool, &s.pp_next));
  TRY(dev_alloc(d, 16, &s.stats, 0)); TRY(dev_alloc(d, 4, &s.out_counts, 0)); TRY(dev_alloc(d, P, &s.unscheduled));
  TRY(dev_alloc(d, 1, &d->d_prob)); TRY(dev_alloc(d, 1, &d->d_state));
  HIPCHK(hipMemcpy(d->d_prob, &d->h, sizeof(DevProb), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(d->d_state, &d->hs, sizeof(DevState), hipMemcpyHostToDevice));
  guard.ok = true; *out = d; return KS_OK;
}

// Build the derived tables + the feasibility grid (idempotent).  Returns the grid kernels' time.
static int build_static(ks_dev_problem* d, float* grid_ms) {
  HIPCHK(hipSetDevice(d->device));
  const DevProb& h = d->h;
  hipEvent_t e0, e1; HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
  const u32 rows = h.K * 64 + 2 * h.K + 64;
  hipLaunchKernelGGL(ks_build_type_tables, dim3((rows * 64 + 255) / 256), dim3(256), 0, d->stream, h);
  const size_t MC = (size_t)h.M * h.C;
  HIPCHK(hipEventRecord(e0, d->stream));
  if (MC) {
    hipLaunchKernelGGL(ks_grid_mc, dim3((u32)((MC + 255) / 256)), dim3(256), 0, d->stream, h);
    // waves = TW * chunks; aim at >= 8 waves per SIMD on 256 CUs (8192 waves) without exceeding the work
    u32 chunks = (u32)std::min<size_t>(MC, std::max<size_t>(1, (8192 + h.TW - 1) / h.TW));
    const size_t waves = (size_t)h.TW * chunks;
    hipLaunchKernelGGL(ks_grid_types, dim3((u32)((waves * 64 + 255) / 256)), dim3(256), 0, d->stream, h, chunks);
  }
  HIPCHK(hipEventRecord(e1, d->stream));
  HIPCHK(hipStreamSynchronize(d->stream));
  HIPCHK(hipGetLastError());
  if (grid_ms) HIPCHK(hipEventElapsedTime(grid_ms, e0, e1));
  hipEventDestroy(e0); hipEventDestroy(e1);
  d->tables_built = true; return KS_OK;
}

extern "C" int ks_feasibility_grid(ks_dev_problem* d, uint64_t* out_grid, float* kernel_ms) {
  if (!d) return fail(KS_ERR_INVALID, "null device problem");
  TRY(build_static(d, kernel_ms));
  if (out_grid) HIPCHK(hipMemcpy(out_grid, d->h.grid, (size_t)d->h.M * d->h.C * d->h.TW * sizeof(u64), hipMemcpyDeviceToHost));
  return KS_OK;
}

static int download(ks_dev_problem* d, ks_result* out) {
  const DevProb& h = d->h; const DevState& s = d->hs;
  u32 counts[4]; HIPCHK(hipMemcpy(counts, s.out_counts, sizeof counts, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(out->stats, s.stats, 16 * sizeof(u64), hipMemcpyDeviceToHost));
  out->n_new = counts[0]; out->n_unscheduled = counts[1];
  if (out->stats[KS_STAT_ERR]) return fail(-(int)out->stats[KS_STAT_ERR], "device-side error (more new nodes than max_new_nodes?)");
  const u32 P = h.P, K = h.K, R = h.R, TW = h.TW, E = h.E, N = out->n_new;
  if (P) {
    HIPCHK(hipMemcpy(out->pod_node, s.pod_node, P * sizeof(i32), hipMemcpyDeviceToHost)); HIPCHK(hipMemcpy(out->pod_stage, s.pod_stage, P * sizeof(i32), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(out->pod_seq, s.pod_seq, P * sizeof(i32), hipMemcpyDeviceToHost)); HIPCHK(hipMemcpy(out->unscheduled, s.unscheduled, P * sizeof(i32), hipMemcpyDeviceToHost));
  }
  if (N) {
    HIPCHK(hipMemcpy(out->node_tmpl, s.n_tmpl, N * sizeof(i32), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(out->node_types, s.n_alive, (size_t)N * TW * sizeof(u64), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(out->node_requests, s.s_req + (size_t)E * R, (size_t)N * R * sizeof(i64), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(out->node_requests_present, s.s_reqmask + E, N * sizeof(u32), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(out->node_present, s.s_present + E, N * sizeof(u32), hipMemcpyDeviceToHost)); HIPCHK(hipMemcpy(out->node_complement, s.s_complement + E, N * sizeof(u32), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(out->node_mask, s.s_mask + (size_t)E * K, (size_t)N * K * sizeof(u64), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(out->node_gt, s.s_gt + (size_t)E * K, (size_t)N * K * sizeof(i32), hipMemcpyDeviceToHost)); HIPCHK(hipMemcpy(out->node_lt, s.s_lt + (size_t)E * K, (size_t)N * K * sizeof(i32), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(out->node_it_state, s.s_it + E, N * sizeof(i32), hipMemcpyDeviceToHost));
  }
  return KS_OK;
}

extern "C" int ks_solve_batch_dev(ks_dev_problem* const* ds, uint32_t n, ks_result* const* outs, float* kernel_ms) {
  if (!n) return KS_OK;
  if (!ds || !outs) return fail(KS_ERR_INVALID, "null batch");
  const int device = ds[0]->device;
  HIPCHK(hipSetDevice(device));
  for (u32 i = 0; i < n; ++i) { if (ds[i]->device != device) return fail(KS_ERR_INVALID, "batch spans devices"); if (!ds[i]->tables_built) TRY(build_static(ds[i], nullptr)); }
  std::vector<DevProb> hp(n); std::vector<DevState> hs(n);
  for (u32 i = 0; i < n; ++i) { hp[i] = ds[i]->h; hs[i] = ds[i]->hs; }
  DevProb* dp = nullptr; DevState* dsv = nullptr;
  if (n == 1) { dp = ds[0]->d_prob; dsv = ds[0]->d_state; }
  else {
    HIPCHK(hipMalloc((void**)&dp, n * sizeof(DevProb))); HIPCHK(hipMalloc((void**)&dsv, n * sizeof(DevState)));
    HIPCHK(hipMemcpy(dp, hp.data(), n * sizeof(DevProb), hipMemcpyHostToDevice)); HIPCHK(hipMemcpy(dsv, hs.data(), n * sizeof(DevState), hipMemcpyHostToDevice));
  }
  hipStream_t st = ds[0]->stream;
  hipEvent_t e0, e1; HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
  HIPCHK(hipEventRecord(e0, st));
  hipLaunchKernelGGL(ks_pack, dim3(n), dim3(KS_NT), 0, st, dp, dsv);
  HIPCHK(hipEventRecord(e1, st));
  HIPCHK(hipStreamSynchronize(st));
  HIPCHK(hipGetLastError());
  if (kernel_ms) HIPCHK(hipEventElapsedTime(kernel_ms, e0, e1));
  hipEventDestroy(e0); hipEventDestroy(e1);
  if (n > 1) { hipFree(dp); hipFree(dsv); }
  for (u32 i = 0; i < n; ++i) TRY(download(ds[i], outs[i]));
  return KS_OK;
}

extern "C" int ks_solve_dev(ks_dev_problem* d, ks_result* out, float* kernel_ms) {
  if (!d || !out) return fail(KS_ERR_INVALID, "null argument");
  ks_dev_problem* arr[1] = {d}; ks_result* outs[1] = {out};
  return ks_solve_batch_dev(arr, 1, outs, kernel_ms);
}

extern "C" int ks_solve(const ks_problem* p, ks_result* out) {
  ks_dev_problem* d = nullptr;
  TRY(ks_problem_upload(p, 0, &d));
  int rc = ks_solve_dev(d, out, nullptr);
  ks_problem_free(d); return rc;
}

extern "C" int ks_solve_batch(const ks_problem* const* ps, uint32_t n, ks_result* const* outs) {
  std::vector<ks_dev_problem*> ds(n, nullptr); int rc = KS_OK;
  for (u32 i = 0; i < n && rc == KS_OK; ++i) rc = ks_problem_upload(ps[i], 0, &ds[i]);
  if (rc == KS_OK) rc = ks_solve_batch_dev(ds.data(), n, outs, nullptr);
  for (auto* d : ds) ks_problem_free(d);
  return rc;
}

static KReq to_k(const ks_req1* a) { KReq r; r.mask = a->mask; r.gt = a->gt; r.lt = a->lt; r.present = a->present; r.complement = a->complement; return r; }
static void from_k(const KReq& r, ks_req1* o) { o->mask = r.mask; o->gt = r.gt; o->lt = r.lt; o->present = r.present; o->complement = r.complement; }

static int probe_device(const ks_req1* a, const ks_req1* b, int wk, const int32_t* vint, uint32_t nv, ks_req1* out, int* ok) {
  if (ks_device_count() <= 0) return fail(KS_ERR_DEVICE, "no gfx950 device");
  i32* dv; ks_req1* dout; int* dok;
  HIPCHK(hipMalloc((void**)&dv, 64 * sizeof(i32))); HIPCHK(hipMalloc((void**)&dout, sizeof(ks_req1))); HIPCHK(hipMalloc((void**)&dok, sizeof(int)));
  i32 tmp[64]; for (int i = 0; i < 64; ++i) tmp[i] = (u32)i < nv ? vint[i] : INT32_MIN;
  HIPCHK(hipMemcpy(dv, tmp, sizeof tmp, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(ks_probe_kernel, dim3(1), dim3(1), 0, 0, *a, *b, dv, nv, wk, dout, dok);
  HIPCHK(hipDeviceSynchronize());
  if (out) HIPCHK(hipMemcpy(out, dout, sizeof(ks_req1), hipMemcpyDeviceToHost));
  if (ok) HIPCHK(hipMemcpy(ok, dok, sizeof(int), hipMemcpyDeviceToHost));
  hipFree(dv); hipFree(dout); hipFree(dok); return KS_OK;
}

extern "C" int ks_probe_intersection(const ks_req1* a, const ks_req1* b, const int32_t* value_int, uint32_t nvalues, int on_device, ks_req1* out) {
  if (on_device) return probe_device(a, b, 1, value_int, nvalues, out, nullptr);
  from_k(kreq_intersect(to_k(a), to_k(b), value_int, nvalues), out); return KS_OK;
}
extern "C" int ks_probe_compatible(const ks_req1* a, const ks_req1* b, int well_known, const int32_t* value_int, uint32_t nvalues, int on_device, int* ok) {
  if (on_device) return probe_device(a, b, well_known, value_int, nvalues, nullptr, ok);
  *ok = !kreq_compatible_fail(to_k(a), to_k(b), well_known != 0, value_int, nvalues); return KS_OK;
}

extern "C" const char* ks_last_error(void) { return g_err.c_str(); }
extern "C" const char* ks_version(void) { return "ksolve 0.1.0 (gfx950)"; }
