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
//   ks_pack               one persistent single-wavefront workgroup per Solve(): the first-fit-decreasing
//                         loop of scheduler.go:96-219 with no barriers.  Per pod the next 64 open nodes in
//                         the reference's visiting order are screened one-per-lane (taints, host ports,
//                         requirement intersection, topology domain choice, resource screen), __ballot +
//                         count-trailing-zeros is the first-fit pick, the instance-type filter runs on
//                         T-bit masks (word per lane, then type per lane), lane-parallel stores commit.
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

// Mutable state of one Solve (device memory).
struct DevState {
  // queue (queue.go:29-72)
  u32* q; u32* lastlen; u32* lastgen; i32* pod_stage; i32* pod_node; i32* pod_seq;
  // node records (AoS, see Rec): slots [0,E) existing nodes, [E,E+NMAX) new nodes
  u8* rec; u32 rec_stride;
  i32* n_tmpl; u64* n_alive;          // new nodes only, indexed by j = slot-E; n_alive has one spare row
  u32* bstart;                        // [P+3] count-bucket boundaries of the visiting-order array
  u32* order_g;                       // [NMAX] global-memory home of the visiting order once it outgrows LDS
  // topology
  i32* gcnt; u64* g_reg; u64* g_pos; u8* g_active; i32* hcnt /* [slot][GH] */; i32* g_hpos;
  // provisioner limits
  i64* remaining;
  // host-port pool
  u64* pp_entry; i32* pp_next; u32 pp_cap;
  // outputs
  u64* stats; u32* out_counts;   // out_counts: [0]=n_new [1]=n_unscheduled
  i32* unscheduled;
  u32* o_present; u32* o_complement; u64* o_mask; i32* o_gt; i32* o_lt; i32* o_it; i64* o_req; u32* o_reqmask;
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
// pack kernel: ONE wavefront per Solve(), no barriers.
//
// The Solve() loop is a serial dependency chain (pod k sees the state pods 0..k-1 left behind), so
// the only parallelism inside one pod step is across candidate nodes and across instance types.  One
// 64-lane wave owns the whole Solve:
//   * open nodes are kept in an array sorted in the reference's visiting order -- the stable
//     `sort.Slice(newNodes, len(Pods))` of scheduler.go:183 -- which is maintained incrementally (a node
//     whose pod count grows moves to the FRONT of the next count bucket; a new node is appended to the
//     BACK of the count-1 bucket);
//   * per step the wave takes the next 64 nodes in that order, one lane per node, evaluates
//     Node.Add up to the instance-type filter in registers, and __ballot + count-trailing-zeros IS the
//     first-fit pick (the reference's "first node whose Add succeeds");
//   * the instance-type filter runs on T-bit masks (one 64-bit word per lane, then one type per lane
//     for resources.Fits), and the commit is a handful of lane-parallel stores.
// Node state is an array of fixed-stride records (AoS: a lane touches one or two cache lines per
// candidate); instance-type allocatable vectors and the visiting order live in LDS.
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
struct ReqOut {    // requirement set of the winning node after the pod is added (written by the winning lane)
  u32 present, complement; i32 it_state; u64 mask[KS_MAX_KEYS]; i32 gt[KS_MAX_KEYS]; i32 lt[KS_MAX_KEYS]; u32 changed; u32 topo_narrowed;
};

// ---- slot record (AoS).  Offsets in bytes; stride = ks_rec_stride(R,K) ----
//   0 u64 taints | 8 u32 present | 12 u32 complement | 16 i32 it_state | 20 u32 reqmask | 24 i32 porthead | 28 u32 count
//   32 i64 req[R] | 32+8R i64 cap[R] | 32+16R u64 mask[K] | +8K i32 gt[K] | +4K i32 lt[K]
__host__ __device__ inline u32 ks_rec_stride(u32 R, u32 K) { return (32 + 16 * R + 16 * K + 15) & ~15u; }
struct Rec {
  u8* p; u32 R, K;
  __device__ __forceinline__ u64& taints() const { return *(u64*)p; }
  __device__ __forceinline__ u32& present() const { return *(u32*)(p + 8); }
  __device__ __forceinline__ u32& complement() const { return *(u32*)(p + 12); }
  __device__ __forceinline__ i32& it_state() const { return *(i32*)(p + 16); }
  __device__ __forceinline__ u32& reqmask() const { return *(u32*)(p + 20); }
  __device__ __forceinline__ i32& porthead() const { return *(i32*)(p + 24); }
  __device__ __forceinline__ u32& count() const { return *(u32*)(p + 28); }
  __device__ __forceinline__ i64* req() const { return (i64*)(p + 32); }
  __device__ __forceinline__ i64* cap() const { return (i64*)(p + 32 + 8 * R); }
  __device__ __forceinline__ u64* mask() const { return (u64*)(p + 32 + 16 * R); }
  __device__ __forceinline__ i32* gt() const { return (i32*)(p + 32 + 16 * R + 8 * K); }
  __device__ __forceinline__ i32* lt() const { return (i32*)(p + 32 + 16 * R + 12 * K); }
};
__device__ __forceinline__ Rec slot_rec(const DevProb& P, const DevState& S, u32 s) { Rec r; r.p = S.rec + (size_t)s * S.rec_stride; r.R = P.R; r.K = P.K; return r; }

struct NodeView {  // where a candidate node's state lives (an open slot, or a template∩class record for a fresh node)
  u32 present, complement; i32 it_state; const u64* mask; const i32* gt; const i32* lt;
  u64 taints; i32 porthead; const i64* req; u32 reqmask; const i64* cap; i32 slot; bool existing; bool fresh;
};
__device__ __forceinline__ NodeView slot_view(const DevProb& P, const DevState& S, u32 s) {
  const Rec r = slot_rec(P, S, s);
  NodeView v; v.present = r.present(); v.complement = r.complement(); v.it_state = r.it_state();
  v.mask = r.mask(); v.gt = r.gt(); v.lt = r.lt(); v.taints = r.taints(); v.porthead = r.porthead(); v.req = r.req(); v.reqmask = r.reqmask();
  v.cap = r.cap(); v.slot = (i32)s; v.existing = s < P.E; v.fresh = false; return v;
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
// `merged`: the pod's own requirements are already folded into the view (fresh node built from the
// template∩class record), only topology is evaluated on top.
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

  // Topology.AddRequirements, topology.go:149-167, then Compatible + Add of the result (node.go:83-90)
  u32 topo_keys = 0; bool host_ok = true;
  u64 dom[KS_MAX_TOUCH]; u32 domset = 0;   // accumulated In-sets per touched entry
  for (int i = 0; i < c.ntopo; ++i) {
    const TopoItem& t = c.topo[i];
    if (t.key == KS_KEY_HOSTNAME) {
      i32 cnt;
      if (v.fresh) cnt = S.g_active[t.g] ? 0 : -1;           // NewNode registers the placeholder first (node.go:47)
      else cnt = S.hcnt[(size_t)v.slot * P.GH + t.hslot];
      bool ok;
      if (t.type == 0) ok = cnt >= 0 && (i64)cnt + t.self <= (i64)t.maxskew;                         // nextDomainTopologySpread, min==0 for hostname (topologygroup.go:184-188)
      else if (t.type == 2) ok = cnt == 0;                                                            // nextDomainAntiAffinity :235-243
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
        KReq pd = kreq_exists();
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
      KReq merged_req = kreq_intersect(in, before, vi, nv);
      if (kreq_len0(merged_req) && !kreq_nidne(before)) return 0;
      treq[e] = merged_req;
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
  const i32 h = P.grp_hslot[g]; i32& c = S.hcnt[(size_t)slot * P.GH + h];
  if (c <= 0) S.g_hpos[h]++;
  c = c < 0 ? 1 : c + 1;
}
// Topology.Record, topology.go:120-143 (one lane)
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

struct WaveShared {
  ClsL cls; ReqOut rq;
  i64 req_new[KS_MAX_RES]; i64 cap_new[KS_MAX_RES];
};

#define WSYNC() __syncthreads()     /* single-wave workgroup: an LDS/global ordering point, not a real barrier */

__device__ __forceinline__ i64 wave_max_i64(i64 v) { for (int off = 32; off > 0; off >>= 1) { const i64 o = __shfl_xor(v, off); if (o > v) v = o; } return v; }

// Load the pod's class into LDS and pre-evaluate the per-pod part of every matching topology group
// (getMatchingTopologies, topology.go:351-364; domainMinCount, topologygroup.go:184-200).
__device__ void stage_class(const DevProb& P, const DevState& S, WaveShared& sh, u32 c, int lane) {
  ClsL& L = sh.cls;
  if ((u32)lane < P.K) { const u32 k = lane; L.mask[k] = P.cls.mask[(size_t)c * P.K + k]; L.gt[k] = P.cls.gt[(size_t)c * P.K + k]; L.lt[k] = P.cls.lt[(size_t)c * P.K + k]; }
  if (lane >= 32 && (u32)lane < 32 + P.R) { const u32 r = lane - 32; L.req[r] = P.cls_requests[(size_t)c * P.R + r]; }
  if (lane == 63) {
    L.c = c; L.present = P.cls.present[c]; L.complement = P.cls.complement[c]; L.it_state = P.cls.it_state[c];
    L.hn_mode = P.cls_hn_mode[c]; L.hn_off = P.cls_hn_off[c]; L.hn_cnt = P.cls_hn_off[c + 1] - P.cls_hn_off[c];
    L.reqmask = P.cls_requests_present[c]; L.tol = P.cls_tolerated[c]; L.port_off = P.cls_port_off[c]; L.port_cnt = P.cls_port_off[c + 1] - P.cls_port_off[c];
  }
  WSYNC();
  const u32 ob = P.cls_own_off[c], oe = P.cls_own_off[c + 1], ib = P.cls_isel_off[c], ie = P.cls_isel_off[c + 1];
  const u32 n = (oe - ob) + (ie - ib);
  if ((u32)lane < n && lane < KS_MAX_TOPO) {
    const u32 i = lane; TopoItem t;
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
  if (lane == 0) L.ntopo = n < KS_MAX_TOPO ? (int)n : KS_MAX_TOPO;
  WSYNC();
}

// Instance-type filter (filterInstanceTypesByRequirements, node.go:137-141) on T-bit masks, one wave:
//   alive' = alive & passTypes(changed keys) & its_types(state) & offerings & fits(requests)
// Lane w owns word w of the mask; for every non-empty word all 64 lanes then test one type each against
// the request vector (resources.Fits) and __ballot rebuilds the word.  Also returns, per resource, the
// maximum Allocatable over alive' (the resource screen of later pods) in sh.cap_new.
__device__ bool filter_types(const DevProb& P, WaveShared& sh, const i64* alloc, const u64* alive_in, u64* alive_out, u32 reqmask_new,
                             u32 changed_keys, bool check_offer, bool check_it, int lane) {
  const ReqOut& rq = sh.rq;
  i64 mx[KS_MAX_RES];
#pragma unroll
  for (int r = 0; r < KS_MAX_RES; ++r) mx[r] = INT64_MIN;
  bool any = false;
  for (u32 wbase = 0; wbase < P.TW; wbase += 64) {
    const u32 w = wbase + lane; u64 a = 0;
    if (w < P.TW) {
      a = alive_in[w];
      for (u32 bits = changed_keys; bits && a; bits &= bits - 1) { const int k = __builtin_ctz(bits); a &= pass_types_word(P, k, load_req(rq.present, rq.complement, rq.mask, rq.gt, rq.lt, k), w); }
      if (check_it && a) a &= P.its_types[(size_t)rq.it_state * P.TW + w];
      if (check_offer && a) a &= offer_types_word(P, rq, w);
    }
    for (u64 nz = ballot64(a != 0); nz; nz &= nz - 1) {
      const int b = __builtin_ctzll(nz);
      const u64 aw = __shfl(a, b);
      const u32 t = (wbase + b) * 64 + lane;
      bool ok = (aw >> lane) & 1ull;
      if (ok) for (u32 bits = reqmask_new; bits; bits &= bits - 1) { const int r = __builtin_ctz(bits); if (sh.req_new[r] > alloc[(size_t)r * P.T + t]) { ok = false; break; } }
      if (ok) for (u32 r = 0; r < P.R; ++r) { const i64 al = alloc[(size_t)r * P.T + t]; if (al > mx[r]) mx[r] = al; }
      const u64 bw = ballot64(ok);
      if (lane == b) a = bw;
    }
    if (w < P.TW) alive_out[w] = a;
    if (ballot64(a != 0)) any = true;
  }
  for (u32 r = 0; r < P.R; ++r) { const i64 v = wave_max_i64(mx[r]); if (lane == 0) sh.cap_new[r] = v; }
  WSYNC();
  return any;
}

// Write the winning node's record after Add (lane-parallel stores).
__device__ __forceinline__ void write_record(const DevProb& P, const Rec& r, const WaveShared& sh, u32 reqmask_new, bool write_cap, int lane) {
  if ((u32)lane < P.K) { r.mask()[lane] = sh.rq.mask[lane]; r.gt()[lane] = sh.rq.gt[lane]; r.lt()[lane] = sh.rq.lt[lane]; }
  if (lane >= 32 && (u32)lane < 32 + P.R) { const int rr = lane - 32; r.req()[rr] = sh.req_new[rr]; if (write_cap) r.cap()[rr] = sh.cap_new[rr]; }
  if (lane == 63) { r.present() = sh.rq.present; r.complement() = sh.rq.complement; r.it_state() = sh.rq.it_state; r.reqmask() = reqmask_new; }
}

extern __shared__ __attribute__((aligned(16))) unsigned char ks_dyn_lds[];

__global__ __launch_bounds__(64) void ks_pack(const DevProb* probs, const DevState* states, u32 lds_bytes) {
  const DevProb& P = probs[blockIdx.x];
  const DevState& S = states[blockIdx.x];
  __shared__ WaveShared sh;
  const int lane = threadIdx.x;
  const u32 NS = P.E + P.NMAX;
  const u64 t_start = __builtin_readcyclecounter();

  // ---- dynamic LDS: instance-type Allocatable table (if it fits in half), then the visiting-order array ----
  const size_t alloc_bytes = (size_t)P.R * P.T * sizeof(i64);
  const i64* alloc = P.it_alloc; u32 lds_used = 0;
  if (alloc_bytes <= lds_bytes / 2) {
    i64* a = (i64*)ks_dyn_lds;
    for (u32 i = lane; i < P.R * P.T; i += 64) a[i] = P.it_alloc[i];
    alloc = a; lds_used = (u32)((alloc_bytes + 15) & ~(size_t)15);
  }
  u32* ord = (u32*)(ks_dyn_lds + lds_used);            // ord[pos] = new-node index j, sorted in visiting order
  const u32 ord_cap = (lds_bytes - lds_used) / 4;
  bool ord_in_lds = true;

  // ---------------- initialise state ----------------
  for (u32 i = lane; i < P.P; i += 64) { S.q[i] = P.queue[i]; S.lastgen[i] = 0xFFFFFFFFu; S.lastlen[i] = 0; S.pod_stage[i] = 0; S.pod_node[i] = -1; S.pod_seq[i] = -1; }
  for (u32 e = lane; e < P.E; e += 64) {
    const Rec r = slot_rec(P, S, e);
    r.taints() = P.en_taints[e]; r.present() = P.en.present[e]; r.complement() = P.en.complement[e]; r.it_state() = P.en.it_state[e];
    r.reqmask() = P.en_requests_present[e]; r.count() = 0;
    for (u32 k = 0; k < P.K; ++k) { r.mask()[k] = P.en.mask[(size_t)e * P.K + k]; r.gt()[k] = P.en.gt[(size_t)e * P.K + k]; r.lt()[k] = P.en.lt[(size_t)e * P.K + k]; }
    for (u32 rr = 0; rr < P.R; ++rr) { r.req()[rr] = P.en_requests[(size_t)e * P.R + rr]; r.cap()[rr] = P.en_avail[(size_t)e * P.R + rr]; }
    i32 head = -1; for (u32 i = P.en_port_off[e]; i < P.en_port_off[e + 1]; ++i) { S.pp_entry[i] = P.ports[i]; S.pp_next[i] = head; head = (i32)i; }
    r.porthead() = head;
    for (u32 h = 0; h < P.GH; ++h) S.hcnt[(size_t)e * P.GH + h] = P.grph_count[(size_t)h * P.E + e];
  }
  for (u32 i = lane; i < P.G * 64; i += 64) S.gcnt[i] = P.grp_count[i];
  for (u32 g = lane; g < P.G; g += 64) {
    u64 reg = 0, pos = 0; for (int d = 0; d < 64; ++d) { const i32 c = P.grp_count[(size_t)g * 64 + d]; if (c >= 0) reg |= 1ull << d; if (c > 0) pos |= 1ull << d; }
    S.g_reg[g] = reg; S.g_pos[g] = pos; S.g_active[g] = P.grp_active[g];
  }
  for (u32 h = lane; h < P.GH; h += 64) { i32 np = P.grph_extra_pos[h]; for (u32 e = 0; e < P.E; ++e) if (P.grph_count[(size_t)h * P.E + e] > 0) ++np; S.g_hpos[h] = np; }
  for (u32 i = lane; i < P.M * P.R; i += 64) S.remaining[i] = P.tmpl_remaining[i];
  WSYNC();

  // wave-uniform loop state lives in registers (SGPRs)
  u32 q_head = 0, q_len = P.P, q_gen = 0, nnew = 0, seq = 0, err = 0, maxc = 0;
  u32 pp_used = P.E ? P.en_port_off[P.E] : 0;
  u64 st_pops = 0, st_relax = 0, st_full = 0, st_fullfail = 0, st_ref_attempts = 0, st_ref_types = 0;
  const bool want_stats = (P.flags & KS_FLAG_STATS) != 0;
  // S.bstart[c] (1 <= c <= maxc+1): first position in `ord` whose node has >= c pods; bstart[maxc+1] == nnew

  // ---------------- Solve loop, scheduler.go:104-124 ----------------
  for (;;) {
    // Queue.Pop, queue.go:44-58
    if (q_len == 0) break;
    const u32 pod = S.q[q_head];
    if (S.lastgen[pod] == q_gen && S.lastlen[pod] == q_len) break;
    q_head = (q_head + 1 == P.P) ? 0 : q_head + 1; q_len--; ++st_pops;
    const u32 cidx = P.stage_cls[P.pod_stage_off[pod] + S.pod_stage[pod]];
    stage_class(P, S, sh, cidx, lane);
    const ClsL& c = sh.cls;
    bool placed = false;

    // ---- 1. existing nodes in the caller's order: first success wins (scheduler.go:176-180) ----
    for (u32 base = 0; base < P.E && !placed; base += 64) {
      const u32 e = base + lane; int rc = 0;
      if (e < P.E) { NodeView v = slot_view(P, S, e); rc = eval_node(P, S, c, v, nullptr); }
      const u64 m = ballot64(rc == 2);
      if (!m) { if (want_stats) st_ref_attempts += min(64u, P.E - base); continue; }
      const int win = __builtin_ctzll(m); const u32 slot = base + win;
      if (want_stats) st_ref_attempts += win + 1;
      if (lane == win) { NodeView v = slot_view(P, S, slot); eval_node(P, S, c, v, &sh.rq); }
      WSYNC();
      const Rec r = slot_rec(P, S, slot);
      if ((u32)lane < P.R) sh.req_new[lane] = r.req()[lane] + c.req[lane];
      const u32 rm = r.reqmask() | c.reqmask;
      WSYNC();
      write_record(P, r, sh, rm, false, lane);                       // commit, existingnode.go:122-129
      if (lane == 0) {
        topology_record(P, S, c, sh.rq, slot);
        for (u32 i = 0; i < c.port_cnt; ++i) { S.pp_entry[pp_used + i] = P.ports[c.port_off + i]; S.pp_next[pp_used + i] = r.porthead(); r.porthead() = (i32)(pp_used + i); }
        S.pod_node[pod] = (i32)slot; S.pod_seq[pod] = (i32)seq;
      }
      pp_used += c.port_cnt; ++seq; placed = true;
      WSYNC();
    }

    // ---- 2. open new nodes in `sort.Slice(newNodes, len(Pods))` order (scheduler.go:183-190) ----
    for (u32 base = 0; base < nnew && !placed; base += 64) {
      const u32 pos = base + lane; u32 j = 0xFFFFFFFFu; int rc = 0;
      if (pos < nnew) { j = ord[pos]; NodeView v = slot_view(P, S, P.E + j); rc = eval_node(P, S, c, v, nullptr); }
      u64 m = ballot64(rc == 2);
      const u64 reach = ballot64(rc >= 1);
      u32 my_alive = 0;
      if (want_stats) { if (rc >= 1) for (u32 w = 0; w < P.TW; ++w) my_alive += __builtin_popcountll(S.n_alive[(size_t)j * P.TW + w]); }
      u32 last_lane = min(64u, nnew - base);     // lanes the reference would have visited in this chunk (all, unless one succeeds)
      while (m) {
        const int win = __builtin_ctzll(m);
        const u32 jw = __shfl(j, win); const u32 slot = P.E + jw;
        if (lane == win) { NodeView v = slot_view(P, S, slot); eval_node(P, S, c, v, &sh.rq); }
        const Rec r = slot_rec(P, S, slot);
        if ((u32)lane < P.R) sh.req_new[lane] = r.req()[lane] + c.req[lane];
        const u32 rm = r.reqmask() | c.reqmask;
        WSYNC();
        ++st_full;
        const bool zc = (P.key_zone >= 0 && ((sh.rq.changed >> P.key_zone) & 1u)) || (P.key_ct >= 0 && ((sh.rq.changed >> P.key_ct) & 1u));
        const bool itc = sh.rq.it_state != r.it_state();
        u64* alive = S.n_alive + (size_t)jw * P.TW;
        u64* scratch = S.n_alive + (size_t)P.NMAX * P.TW;     // one spare row
        const bool ok = filter_types(P, sh, alloc, alive, scratch, rm, sh.rq.changed, zc, itc, lane);
        if (!ok) { ++st_fullfail; m &= m - 1; continue; }
        // commit, node.go:100-105
        last_lane = win + 1;
        for (u32 w = lane; w < P.TW; w += 64) alive[w] = scratch[w];
        write_record(P, r, sh, rm, true, lane);
        const u32 cnt = r.count();                                   // pods on the node before this one
        if (lane == 0) {
          r.count() = cnt + 1;
          topology_record(P, S, c, sh.rq, slot);
          for (u32 i = 0; i < c.port_cnt; ++i) { S.pp_entry[pp_used + i] = P.ports[c.port_off + i]; S.pp_next[pp_used + i] = r.porthead(); r.porthead() = (i32)(pp_used + i); }
          S.pod_node[pod] = (i32)slot; S.pod_seq[pod] = (i32)seq;
        }
        // visiting order: the node leaves position p of bucket `cnt` for the FRONT of bucket cnt+1
        {
          const u32 p = base + win;
          const u32 endc = S.bstart[cnt + 1];                        // one past the last node with `cnt` pods
          for (u32 i = p + 1; i < endc; i += 64) { const u32 ii = i + lane; u32 v = 0; if (ii < endc) v = ord[ii]; WSYNC(); if (ii < endc) ord[ii - 1] = v; }
          WSYNC();
          if (lane == 0) { ord[endc - 1] = jw; S.bstart[cnt + 1] = endc - 1; if (cnt + 1 > maxc) S.bstart[cnt + 2] = nnew; }
          if (cnt + 1 > maxc) maxc = cnt + 1;
        }
        pp_used += c.port_cnt; ++seq; placed = true;
        WSYNC();
        break;
      }
      if (want_stats) {
        st_ref_attempts += last_lane;
        u32 ty = ((u32)lane < last_lane && ((reach >> lane) & 1ull)) ? my_alive : 0;
        for (int off = 32; off > 0; off >>= 1) ty += __shfl_xor(ty, off);
        st_ref_types += ty;
      }
    }

    // ---- 3. a new node from the first template that works (scheduler.go:193-217) ----
    for (u32 m = 0; m < P.M && !placed && !err; ++m) {
      const size_t mc = (size_t)m * P.C + cidx;
      const u32 lim = P.tmpl_limit_present[m];
      const u32 j = nnew; const u32 slot = P.E + j;
      if (nnew >= P.NMAX) { err = (u32)(-KS_ERR_CAPACITY); break; }
      u64* alive = S.n_alive + (size_t)j * P.TW;
      u64* scratch = S.n_alive + (size_t)P.NMAX * P.TW;
      // filterByRemainingResources, scheduler.go:293-309 (only when the provisioner has limits)
      bool lany = false; u32 ltypes = 0;
      for (u32 wbase = 0; wbase < P.TW; wbase += 64) {
        const u32 w = wbase + lane; u64 a = 0;
        if (w < P.TW) a = P.tmpl_types[(size_t)m * P.TW + w];
        if (lim != 0xFFFFFFFFu) {
          for (u64 nz = ballot64(a != 0); nz; nz &= nz - 1) {
            const int b = __builtin_ctzll(nz); const u64 aw = __shfl(a, b); const u32 t = (wbase + b) * 64 + lane;
            bool ok = (aw >> lane) & 1ull;
            if (ok) for (u32 bits = lim; bits; bits &= bits - 1) { const int r = __builtin_ctz(bits); if (P.it_cap[(size_t)r * P.T + t] > S.remaining[(size_t)m * P.R + r]) { ok = false; break; } }
            const u64 bw = ballot64(ok); if (lane == b) a = bw;
          }
        }
        if (ballot64(a != 0)) lany = true;
        if (want_stats) { u32 pc = __builtin_popcountll(a); for (int off = 32; off > 0; off >>= 1) pc += __shfl_xor(pc, off); ltypes += pc; }
        if (w < P.TW) scratch[w] = a & P.grid[mc * P.TW + w];
      }
      if (!lany) continue;                      // "all available instance types exceed provisioner limits" (before NewNode)
      WSYNC();
      if (want_stats) ++st_ref_attempts;        // NewNode + node.Add is attempted for this template
      if (!P.mc_ok[mc]) continue;               // taints / Compatible fail inside Add
      // NewNode + Node.Add on the fresh node (its own requirements are already merged in mc_*)
      int rc = 0;
      if (lane == 0) {
        NodeView v; v.present = P.mc_present[mc]; v.complement = P.mc_complement[mc]; v.it_state = P.mc_it[mc];
        v.mask = P.mc_mask + mc * P.K; v.gt = P.mc_gt + mc * P.K; v.lt = P.mc_lt + mc * P.K;
        v.taints = 0; v.porthead = -1; v.req = P.tmpl_daemon + (size_t)m * P.R; v.reqmask = P.tmpl_daemon_present[m]; v.cap = nullptr; v.slot = (i32)slot; v.existing = false; v.fresh = true;
        rc = eval_node(P, S, c, v, &sh.rq, true);
      }
      rc = __shfl(rc, 0);
      if ((u32)lane < P.R) sh.req_new[lane] = P.tmpl_daemon[(size_t)m * P.R + lane] + c.req[lane];
      const u32 rm = P.tmpl_daemon_present[m] | c.reqmask;
      WSYNC();
      if (rc != 2) continue;
      ++st_full; if (want_stats) st_ref_types += ltypes;
      // topology may have narrowed keys beyond template∩class: re-filter those keys (+ offerings); fits again for the maxima
      const u32 nk = sh.rq.topo_narrowed;
      const bool zc = (P.key_zone >= 0 && ((nk >> P.key_zone) & 1u)) || (P.key_ct >= 0 && ((nk >> P.key_ct) & 1u));
      const bool ok = filter_types(P, sh, alloc, scratch, alive, rm, nk, zc, false, lane);
      if (!ok) { ++st_fullfail; continue; }
      // commit the new node (scheduler.go:214-216); Topology.Register(hostname), node.go:47
      for (u32 g = lane; g < P.G; g += 64) if (P.grp_hslot[g] >= 0) S.hcnt[(size_t)slot * P.GH + P.grp_hslot[g]] = S.g_active[g] ? 0 : -1;
      // subtractMax, scheduler.go:273-290
      if (lim != 0xFFFFFFFFu) {
        i64 mx[KS_MAX_RES];
#pragma unroll
        for (int r = 0; r < KS_MAX_RES; ++r) mx[r] = INT64_MIN;
        for (u32 t = lane; t < P.T; t += 64) if ((alive[t >> 6] >> (t & 63)) & 1ull) for (u32 r = 0; r < P.R; ++r) { const i64 cp = P.it_cap[(size_t)r * P.T + t]; if (cp > mx[r]) mx[r] = cp; }
        for (u32 r = 0; r < P.R; ++r) { const i64 v = wave_max_i64(mx[r]); if (lane == 0 && ((lim >> r) & 1u)) S.remaining[(size_t)m * P.R + r] -= v; }
      }
      const Rec r = slot_rec(P, S, slot);
      write_record(P, r, sh, rm, true, lane);
      WSYNC();
      if (lane == 0) {
        r.taints() = P.tmpl_taints[m]; r.porthead() = -1; r.count() = 1; S.n_tmpl[j] = (i32)m;
        topology_record(P, S, c, sh.rq, slot);
        for (u32 i = 0; i < c.port_cnt; ++i) { S.pp_entry[pp_used + i] = P.ports[c.port_off + i]; S.pp_next[pp_used + i] = r.porthead(); r.porthead() = (i32)(pp_used + i); }
        S.pod_node[pod] = (i32)slot; S.pod_seq[pod] = (i32)seq;
      }
      // visiting order: appended -> BACK of the count-1 bucket, i.e. position bstart[2]; everything after shifts right
      {
        if (ord_in_lds && nnew + 1 > ord_cap) {                       // spill the order array to global memory
          for (u32 i = lane; i < nnew; i += 64) S.order_g[i] = ord[i];
          WSYNC(); ord = S.order_g; ord_in_lds = false;
        }
        if (maxc == 0) { if (lane == 0) { S.bstart[1] = 0; S.bstart[2] = 1; ord[0] = j; } maxc = 1; }
        else {
          const u32 ins = S.bstart[2];
          for (u32 hi = nnew; hi > ins; ) { const u32 lo = hi > ins + 64 ? hi - 64 : ins; const u32 ii = lo + lane; u32 v = 0; if (ii < hi) v = ord[ii]; WSYNC(); if (ii < hi) ord[ii + 1] = v; WSYNC(); hi = lo; }
          if (lane == 0) ord[ins] = j;
          for (u32 cc = 2 + lane; cc <= maxc + 1; cc += 64) S.bstart[cc] += 1;
        }
      }
      nnew = j + 1; pp_used += c.port_cnt; ++seq; placed = true;
      WSYNC();
    }
    if (err) break;

    // ---- 4. failure: Preferences.Relax + Queue.Push + Topology.Update (scheduler.go:116-123) ----
    if (!placed) {
      const u32 nst = P.pod_stage_off[pod + 1] - P.pod_stage_off[pod];
      const i32 stg = S.pod_stage[pod];
      const bool relaxed = (u32)stg + 1 < nst;
      u32 tail = q_head + q_len; if (tail >= P.P) tail -= P.P;
      q_len++;
      if (lane == 0) {
        S.q[tail] = pod;
        if (relaxed) {
          S.pod_stage[pod] = stg + 1;
          const u32 nc = P.stage_cls[P.pod_stage_off[pod] + stg + 1];
          for (u32 i = P.cls_own_off[nc]; i < P.cls_own_off[nc + 1]; ++i) S.g_active[P.own_list[i] & 0x7FFFFFFFu] = 1;   // Topology.Update creates the group
        } else { S.lastlen[pod] = q_len; S.lastgen[pod] = q_gen; }
      }
      if (relaxed) { q_gen++; ++st_relax; }
      WSYNC();
    }
  }

  // ---------------- results ----------------
  WSYNC();
  for (u32 i = lane; i < q_len; i += 64) { u32 idx = q_head + i; if (idx >= P.P) idx -= P.P; S.unscheduled[i] = (i32)S.q[idx]; }
  for (u32 j = lane; j < nnew; j += 64) {        // de-interleave the new nodes' records into the SoA result arrays
    const Rec r = slot_rec(P, S, P.E + j);
    S.o_present[j] = r.present(); S.o_complement[j] = r.complement(); S.o_it[j] = r.it_state(); S.o_reqmask[j] = r.reqmask();
    for (u32 k = 0; k < P.K; ++k) { S.o_mask[(size_t)j * P.K + k] = r.mask()[k]; S.o_gt[(size_t)j * P.K + k] = r.gt()[k]; S.o_lt[(size_t)j * P.K + k] = r.lt()[k]; }
    for (u32 rr = 0; rr < P.R; ++rr) S.o_req[(size_t)j * P.R + rr] = r.req()[rr];
  }
  if (lane == 0) {
    S.out_counts[0] = nnew; S.out_counts[1] = q_len;
    S.stats[KS_STAT_POPS] = st_pops; S.stats[KS_STAT_RELAX] = st_relax; S.stats[KS_STAT_FULLCHECKS] = st_full; S.stats[KS_STAT_FULLFAILS] = st_fullfail;
    S.stats[KS_STAT_REF_ATTEMPTS] = st_ref_attempts; S.stats[KS_STAT_REF_TYPES] = st_ref_types;
    S.stats[KS_STAT_CYCLES] = __builtin_readcyclecounter() - t_start; S.stats[KS_STAT_ERR] = err;
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
  s.rec_stride = ks_rec_stride(R, K);
  TRY(dev_alloc(d, NS * s.rec_stride, &s.rec, 0));
  TRY(dev_alloc(d, (size_t)h.NMAX, &s.n_tmpl)); TRY(dev_alloc(d, ((size_t)h.NMAX + 1) * TW, &s.n_alive));
  TRY(dev_alloc(d, (size_t)P + 4, &s.bstart, 0)); TRY(dev_alloc(d, (size_t)h.NMAX, &s.order_g));
  TRY(dev_alloc(d, (size_t)G * 64, &s.gcnt)); TRY(dev_alloc(d, G, &s.g_reg)); TRY(dev_alloc(d, G, &s.g_pos)); TRY(dev_alloc(d, G, &s.g_active));
  TRY(dev_alloc(d, (size_t)p->GH * NS, &s.hcnt, 0xFF)); TRY(dev_alloc(d, p->GH, &s.g_hpos)); TRY(dev_alloc(d, (size_t)M * R, &s.remaining));
  const size_t NM = h.NMAX;
  TRY(dev_alloc(d, NM, &s.o_present)); TRY(dev_alloc(d, NM, &s.o_complement)); TRY(dev_alloc(d, NM * K, &s.o_mask)); TRY(dev_alloc(d, NM * K, &s.o_gt)); TRY(dev_alloc(d, NM * K, &s.o_lt));
  TRY(dev_alloc(d, NM, &s.o_it)); TRY(dev_alloc(d, NM * R, &s.o_req)); TRY(dev_alloc(d, NM, &s.o_reqmask));
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
  const u32 P = h.P, K = h.K, R = h.R, TW = h.TW, N = out->n_new;
  if (P) {
    HIPCHK(hipMemcpy(out->pod_node, s.pod_node, P * sizeof(i32), hipMemcpyDeviceToHost)); HIPCHK(hipMemcpy(out->pod_stage, s.pod_stage, P * sizeof(i32), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(out->pod_seq, s.pod_seq, P * sizeof(i32), hipMemcpyDeviceToHost)); HIPCHK(hipMemcpy(out->unscheduled, s.unscheduled, P * sizeof(i32), hipMemcpyDeviceToHost));
  }
  if (N) {
    HIPCHK(hipMemcpy(out->node_tmpl, s.n_tmpl, N * sizeof(i32), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(out->node_types, s.n_alive, (size_t)N * TW * sizeof(u64), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(out->node_requests, s.o_req, (size_t)N * R * sizeof(i64), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(out->node_requests_present, s.o_reqmask, N * sizeof(u32), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(out->node_present, s.o_present, N * sizeof(u32), hipMemcpyDeviceToHost)); HIPCHK(hipMemcpy(out->node_complement, s.o_complement, N * sizeof(u32), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(out->node_mask, s.o_mask, (size_t)N * K * sizeof(u64), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(out->node_gt, s.o_gt, (size_t)N * K * sizeof(i32), hipMemcpyDeviceToHost)); HIPCHK(hipMemcpy(out->node_lt, s.o_lt, (size_t)N * K * sizeof(i32), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(out->node_it_state, s.o_it, N * sizeof(i32), hipMemcpyDeviceToHost));
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
  // dynamic LDS: Allocatable table + visiting-order array.  One Solve gets most of the CU's 160 KiB;
  // batched what-ifs take 64 KiB each so two workgroups share a CU.
  const u32 lds_bytes = n == 1 ? 144u * 1024u : 64u * 1024u;
  static bool attr_set = false;
  if (!attr_set) { HIPCHK(hipFuncSetAttribute((const void*)ks_pack, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024)); attr_set = true; }
  hipLaunchKernelGGL(ks_pack, dim3(n), dim3(64), lds_bytes, st, dp, dsv, lds_bytes);
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
