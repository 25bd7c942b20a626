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
//                         scheduler.go:96-219.  Per pod the next 64 open nodes in the reference's visiting
//                         order are screened one-per-lane (taints, host ports, requirement intersection,
//                         topology domain choice, resource screen), __ballot + count-trailing-zeros is the
//                         first-fit pick, the instance-type filter runs on T-bit masks (a word per lane),
//                         lane-parallel stores commit.  A batch of what-ifs runs one single-wave workgroup
//                         each; a single Solve gets 8 (or 4) waves: wave 0 carries the sequential state, the
//                         others evaluate the next queued pods against the same snapshot (speculation rounds,
//                         resolved exactly) and scan ahead for pods whose first fit lies deep in the order.
//   ks_price_filter       consolidation price stage (filterByPrice / worstLaunchPrice) on device-resident results.
//   ks_gather             batched read-back of a what-if batch's results.
//
// The reference functions each device function restates are cited inline (paths relative to
// aws/karpenter-core pkg/).  There is deliberately NO CPU fallback in this library: if no gfx950
// device is present every entry point returns KS_ERR_DEVICE.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <string>
#include <vector>
#include <type_traits>
#include <system_error>

#include "../../include/ksolve.h"
#include "ks_algebra.h"

#define KS_FLAG_NOFOLD 0x10000u   /* internal (KS_NO_FOLD=1 at upload): the rounds do not fold node-opening pods in (A/B and parity of both ways) */
#define KS_MAX_TOPO 24       // topology groups evaluated per pod class
#define KS_MAX_TOUCH 12      // distinct narrow keys a class may touch (own requirements + topology + recorded keys)

#ifdef KS_SIM      /* tests/sim/hip_sim.h: the kernels of this file run on the host, a fibre per lane (test infrastructure; hipcc never defines it) */
struct u32x4 { unsigned int x, y, z, w; };
#else
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#endif
#define RL(v, l) ((u32)__builtin_amdgcn_readlane((int)(v), (l)))   /* broadcast from a wave-uniform lane: v_readlane, no LDS round trip */
/* a value the program knows to be wave-uniform but the compiler cannot (it came out of LDS / global memory): move it to
   SGPRs so that branches on it are scalar branches and address arithmetic is scalar */
#define UF(x) ((u32)__builtin_amdgcn_readfirstlane((int)(x)))
#define UF64(x) ((u64)UF((u32)(u64)(x)) | ((u64)UF((u32)((u64)(x) >> 32)) << 32))
#define RL64(v, l) ((u64)RL((u32)(u64)(v), (l)) | ((u64)RL((u32)((u64)(v) >> 32), (l)) << 32))
#ifdef KS_CUTSTATS  /* experiment builds: why speculation rounds end (slots 12..17 of the statistics instead of the sequential phase probes) */
#define CUT(i) do { if (lane == 0) ls.ctr[(i)] += 1; } while (0)
#else
#define CUT(i) do { } while (0)
#endif
#ifdef KS_P2PROBES   /* experiment builds: where the resolver's time goes (slots 12..19 instead of the sequential phase probes) */
#define P2T(i) do { const u64 now_ = __builtin_readcyclecounter(); if (lane == 0) ls.ctr[(i)] += now_ - t2p; t2p = now_; } while (0)
#define P2C(i, v) do { if (lane == 0) ls.ctr[(i)] += (v); } while (0)
#else
#define P2T(i) do { } while (0)
#define P2C(i, v) do { } while (0)
#endif
#if defined(KS_PROBES) && !defined(KS_CUTSTATS) && !defined(KS_P2PROBES)   /* fine-grained cycle probes (tools/phase_profile.py --probes builds with -DKS_PROBES) */
#define PROBE(i) do { const u64 now_ = __builtin_readcyclecounter(); if (lane == 0 && wv == 0) ls.ctr[(i)] += now_ - tprobe; tprobe = now_; } while (0)
#else
#define PROBE(i) do { (void)tprobe; } while (0)
#endif
#ifdef KS_SIM
#define GA
#else
#define GA __attribute__((address_space(1)))   /* global address space: loads become global_load, not flat_load */
#endif
typedef uint64_t u64; typedef uint32_t u32; typedef uint16_t u16; typedef int64_t i64; typedef int32_t i32; typedef uint8_t u8;

static thread_local std::string g_err;
static int fail(int code, const std::string& msg) { g_err = msg; return code; }
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return fail(KS_ERR_DEVICE, std::string(#x) + ": " + hipGetErrorString(e_)); } while (0)

// ------------------------------------------------------------------------------------------------
// device-side views
// ------------------------------------------------------------------------------------------------
struct ReqSetsD { u32 n; const u32* present; const u32* complement; const u64* mask; const i32* gt; const i32* lt; const i32* it_state; };

struct DevProb {
  u32 P, C, T, TW, M, E, K, R, G, GH, S, SC, NMAX, flags, n_topologies, ge_max;   // ge_max: largest ge_cnt[r]
  u32 wellknown_mask; const u32* key_nvalues; const i32* value_int; i32 key_zone, key_ct; u32 n_ct;
  const u32* it_present; const u32* it_complement; const u64* it_mask; const i64* it_alloc; const i64* it_cap; const u64* it_offer;
  const double* it_price; const double* it_price_lo; i32 ct_spot, ct_ondemand;     // consolidation price stage (ks_price_filter), launch pick (ks_launch_pick)
  const u16* its_inter; const u8* its_fail; const u8* its_nidne; const u64* its_types;
  ReqSetsD tmpl; const u64* tmpl_taints; const i64* tmpl_daemon; const u32* tmpl_daemon_present; const u64* tmpl_types;
  const u32* tmpl_limit_present; const i64* tmpl_remaining;
  ReqSetsD en; const u64* en_taints; const i64* en_avail; const i64* en_requests; const u32* en_requests_present; const u32* en_port_off;
  ReqSetsD cls; const u8* cls_hn_mode; const u32* cls_hn_off; const u32* hn_list; const i64* cls_requests; const u32* cls_requests_present;
  const u64* cls_tolerated; const u32* cls_port_off; const u64* ports;
  u32 ND, SW; const i32* en_vol_limit; const i32* en_vol_count; const u64* en_vol_set; const u32* cls_vol_off; const u32* vol_list;   // volume limits of existing nodes
  const u32* cls_own_off; const u32* own_list; const u32* cls_sel_off; const u32* sel_list;
  const u32* cls_isel_off; const u32* isel_list; const u32* cls_iown_off; const u32* iown_list;
  const u32* pod_stage_off; const u32* stage_cls; const u32* queue;
  // A what-if DERIVED on the device from a resident cluster snapshot (ks_whatifs_open): the batch is a subset of the snapshot's pods.  Pod i of
  // the what-if (its position in the snapshot's queue order: queue == nullptr reads as the identity) is the snapshot's pod pod_gid[i], whose
  // relaxation chain lives in the snapshot's pod_stage_off / stage_cls; en_removed: bit e = existing node e left the cluster (a candidate).
  const u32* pod_gid; const u64* en_removed;
  const i32* hgrp_of;      // derived what-if over a snapshot WITH topology groups: hostname row h -> its group (grp_active is then this what-if's: a group no pod of the batch owns yet)
  const u8* grp_type; const i32* grp_key; const i32* grp_max_skew; const u8* grp_active; const u32* grp_filter_off; ReqSetsD flt;
  const i32* grp_count; const i32* grp_hslot; const i32* grph_count; const i32* grph_extra_pos;
  // derived static tables (built on the device by ks_build_type_tables / ks_grid_*)
  u64* kv_types;     // [K*64*TW] types lacking key k or whose requirement on k Has(value v)
  u64* cmplx_types;  // [K*TW]    types lacking key k or with a complement requirement on k
  u64* nidnex_types; // [K*TW]    types lacking key k or with operator in {NotIn, DoesNotExist} on k
  u64* pair_types;   // [64*TW]   types with an available offering for (zone,capacity-type) pair
  i64* ge_vals;      // [R*T]     ascending distinct Allocatable values of resource r (ge_cnt[r] of them)
  u32* ge_cnt;       // [R]
  u64* ge_rows;      // [(r*T+i)*TW] types whose Allocatable[r] >= ge_vals[r*T+i]   (resources.Fits as a mask)
  void* plans;       // [C] ClsPlan: per-class plan records (ks_build_plans)
  void* briefs;      // [C] ClsBrief: what the round planner / resolver reads of a class
  // Spread groups the round resolver follows exactly (ks_pack, "dynamic spread"): unfiltered, initially active spread groups on ONE narrow key
  // (the key with the most of them, <= 8 values): bit g of dyn_groups, slot(g) = popcount(dyn_groups below g), at most 16.
  u64 dyn_groups; i32 dyn_key; u32 dyn_nz;
  u32* ev_tab; u32 ev_tab_size, derived_shared;   // open-addressing table that interns evaluation classes (ks_link_ev); derived_shared: kv_types .. ge_rows belong to another problem
  u8* mc_ok;         // [M*C]
  u8* mc_why;        // [M*C]  KS_WHY_* of a fresh node of template m refusing class c before the topology step (0 if mc_ok)
  u32* mc_present; u32* mc_complement; u64* mc_mask; i32* mc_gt; i32* mc_lt; i32* mc_it;   // template ∩ class
  u64* grid;         // [M*C*TW]
  // register-resident pack kernel (ks_pack_rr.inc): per-class briefs (ks_build_rr), per-class cached answers, template x class -> nrc, the nrcs' type rows, the node hand-over
  void* rr_briefs; u32* rr_memo; u8* rr_mcnrc; u64* rr_types; u64* rr_nodes; u32* rr_tab; u32* rr_tab2; u32* rr_mi; u8* rr_mcch; u32* rr_hot;      // rr_hot[c][8]: what the head window's straight-line loop reads of a class, unpacked once (ks_build_rr) | rr_tab: interns requirement classes (ks_link_rr); rr_mi[c]: the class whose cached answers class c shares; rr_mcch: template x class -> nrc of a fresh node with the pod on it
};

// Mutable state of one Solve (device memory).
struct DevState {
  // queue (queue.go:29-72)
  u64* q; u32* lastlen; u32* lastgen; i32* pod_stage; i32* pod_node; i32* pod_seq; u32* pod_reason;   // q entry: pod | class<<32 | requeued-unrelaxed<<63
  // node records (AoS, see Rec): slots [0,E) existing nodes, [E,E+NMAX) new nodes
  u8* rec; u32 rec_stride;
  i32* n_tmpl; u64* n_alive;          // new nodes only, indexed by j = slot-E; n_alive has one spare row
  u64* lowi;                          // [E+NMAX][2] per node: index of Rec::low[r] in resource r's Allocatable ladder, 16 bits each (0xFFFF: none yet; word r/4) -- the rounds' filters resume from it
  u64* round_scratch;                 // [KS_MAX_WAVES*64][TW]: InstanceTypeOptions rows as they were before a round's filters (restored if the round is cancelled)
  u32* bstart;                        // [P+3] count-bucket boundaries of the visiting-order array
  u32* order_g;                       // [NMAX] global-memory home of the visiting order once it outgrows LDS
  // topology
  i32* gcnt; u64* g_reg; u64* g_pos; u8* g_active; i32* hcnt /* [slot][GH] */; i32* g_hpos; i32* g_hzero;   // g_hzero[h]: registered hostnames of group h whose count is 0
  // provisioner limits
  i64* remaining;
  // host-port pool
  u64* pp_entry; i32* pp_next; u32 pp_cap;
  // volumes mounted on the existing nodes (VolumeUsage.volumes as counts + the shared-claim set)
  u32 vol_pad; i32* vol_cnt; u64* vol_set;
  // per class (index eq-1 for evaluation-equivalent classes): every existing node below it has refused the class (ClsPlan::mono)
  u32* wm;
  // outputs
  u64* stats; u32* out_counts;   // out_counts: [0]=n_new [1]=n_unscheduled
  u64* batch_meta;               // batched launches: [0]=n_new [1]=n_unscheduled [2..33]=stats of this problem in one batch-wide array (one read-back)
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
#ifdef KS_BALLOT_BUILTIN   /* experiment builds only (tools/mkvariant.sh t_ballot -DKS_BALLOT_BUILTIN): see DESIGN.md "open items" */
__device__ __forceinline__ u64 ballot64(bool p) { return __builtin_amdgcn_ballot_w64(p); }
#else
__device__ __forceinline__ u64 ballot64(bool p) { return __ballot(p); }      // (__builtin_amdgcn_ballot_w64 on the bool itself saves two VALU ops per ballot; in round 2 a build with it faulted -- re-examined in round 3)
#endif

// The count existing node e starts with in hostname-keyed group (row) h.  A derived what-if (DevProb::en_removed / hgrp_of) reads the SNAPSHOT's table --
// the pods of the cluster that match the group on that node, as countDomains would find them were none of them in the batch (topology.go:231-276) --
// and adjusts it to its candidate set: a node that left registers nowhere, and in a group no pod of the batch owns yet (created later by a relaxation,
// topology.go:86-117) a hostname is registered only where pods were counted (NewTopologyGroup registers the universe, which holds no hostnames).
__device__ __forceinline__ i32 ks_host_count0(const DevProb& P, u32 h, u32 e) {
  i32 c = P.grph_count[(size_t)h * P.E + e];
  if (P.en_removed && ((P.en_removed[e >> 6] >> (e & 63u)) & 1ull)) return -1;
  if (c == -2) c = (P.hgrp_of && P.grp_active[P.hgrp_of[h]]) ? 0 : -1;      // (ks_whatif_topo::grph_base: a hostname only NewExistingNode would register)
  return c;
}

// ------------------------------------------------------------------------------------------------
// ks_build_type_tables: one wave per table row
// rows: [0, K*64) -> kv_types[k][v]; [K*64, K*64+K) -> cmplx[k]; [.., +K) -> nidnex[k]; [.., +64) -> pair_types
// ------------------------------------------------------------------------------------------------
// (Every static-table kernel takes the descriptor array of a batch; blockIdx.y picks the problem.  One Solve launches with grid.y = 1.)
__global__ __launch_bounds__(256) void ks_build_type_tables(const DevProb* probs) {
  const DevProb& P = probs[blockIdx.y];
  if (P.derived_shared) return;      // the catalogue-derived rows are another problem's (ks_problem_upload_shared)
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
// ks_build_ge_rows: one wave per (resource r, distinct value i): the T-bit mask of instance types whose
// Allocatable()[r] >= value.  resources.Fits(requests, allocatable) (resources.go:138-145) for a whole
// catalogue then is the AND of one row per requested resource.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ks_build_ge_rows(const DevProb* probs) {
  const DevProb& P = probs[blockIdx.y];
  if (P.derived_shared) return;
  const int lane = threadIdx.x & 63;
  const u32 wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (wave >= P.R * P.T) return;
  const u32 r = wave / P.T, i = wave % P.T;
  if (i >= P.ge_cnt[r]) return;
  const i64 v = P.ge_vals[(size_t)r * P.T + i];
  for (u32 w = 0; w < P.TW; ++w) {
    const u32 t = w * 64 + lane;
    const bool bit = t < P.T && P.it_alloc[(size_t)r * P.T + t] >= v;
    const u64 m = ballot64(bit); if (lane == 0) P.ge_rows[(size_t)wave * P.TW + w] = m;
  }
}

// ------------------------------------------------------------------------------------------------
// ks_grid_mc: one thread per (template m, class c)
// node.go:62-80 for a fresh node: Taints.Tolerates, nodeRequirements.Compatible(podRequirements),
// nodeRequirements.Add(podRequirements).  A fresh node's hostname is a placeholder no pod can name
// (node.go:46), so a pod class with a concrete hostname requirement can never use a new node.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ks_grid_mc(const DevProb* probs) {
  const DevProb& P = probs[blockIdx.y];
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)P.M * P.C) return;
  const u32 m = idx / P.C, c = idx % P.C;
  bool ok = (P.tmpl_taints[m] & ~P.cls_tolerated[c]) == 0;
  const bool taints_ok = ok;
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
  if (P.its_fail[sa * P.SC + sb]) ok = false;
  P.mc_it[idx] = P.its_inter[sa * P.SC + sb];
  P.mc_present[idx] = present; P.mc_complement[idx] = complement; P.mc_ok[idx] = ok ? 1 : 0;
  P.mc_why[idx] = ok ? KS_WHY_NONE : (!taints_ok ? KS_WHY_TAINTS : KS_WHY_REQUIREMENTS);
}

// ------------------------------------------------------------------------------------------------
// waves = TW * chunks: aim at `wave_target` waves (>= 8 per SIMD on 256 CUs for one Solve) without exceeding the work
__host__ __device__ inline u32 ks_grid_chunks(size_t MC, u32 TW, u32 wave_target) { const size_t want = (wave_target + TW - 1) / TW; const size_t c = MC < want ? MC : want; return (u32)(c ? c : 1); }

// ks_grid_types: the feasibility grid.  filterInstanceTypesByRequirements (node.go:137-159) for a
// fresh node of template m receiving a pod of class c:
//   compatible  = instanceType.Requirements.Intersects(nodeRequirements) == nil     (node.go:143)
//   fits        = resources.Fits(daemonOverhead + podRequests, Allocatable())      (node.go:147)
//   hasOffering = some available offering whose zone / capacity-type the node allows (node.go:151)
// Wave (w, chunk): lane owns type t = 64*w + lane for the whole kernel; (m,c) records are wave-uniform.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ks_grid_types(const DevProb* probs, u32 wave_target, u32 row_lo, u32 row_hi) {      // rows [row_lo, row_hi) of the grid (the rows split over GPUs; everything: 0, ~0)
  const DevProb& P = probs[blockIdx.y];
  const size_t MC0 = (size_t)P.M * P.C; if (MC0 == 0) return;
  const u32 chunks = ks_grid_chunks(MC0, P.TW, wave_target);
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
  const size_t MCall = (size_t)P.M * P.C, MC = MCall < (size_t)row_hi ? MCall : (size_t)row_hi;
  for (size_t mc = (size_t)row_lo + chunk; mc < MC; mc += chunks) {
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
// pack kernel: ONE workgroup per Solve().
//
// The Solve() loop is a serial dependency chain (pod k sees the state pods 0..k-1 left behind), so
// the only parallelism inside one pod step is across candidate nodes and across instance types.  Wave 0
// owns the Solve's sequential state (the other waves of a multi-wave workgroup only help, see ks_pack):
//   * open nodes are kept in an array sorted in the reference's visiting order -- the stable
//     `sort.Slice(newNodes, len(Pods))` of scheduler.go:183 -- which is maintained incrementally (a node
//     whose pod count grows moves to the FRONT of the next count bucket; a new node is appended to the
//     BACK of the count-1 bucket);
//   * per step the wave takes the next 64 nodes in that order, one lane per node, evaluates
//     Node.Add up to the instance-type filter in registers, and __ballot + count-trailing-zeros IS the
//     first-fit pick (the reference's "first node whose Add succeeds");
//   * the instance-type filter runs on T-bit masks (one 64-bit word per lane; resources.Fits is one
//     precomputed row per requested resource), and the commit is a handful of lane-parallel stores.
// Latency, not bandwidth, bounds this kernel, so the data layout is built to make every per-pod load
// independent: the pod's class is ONE pre-built plan record (a single coalesced load), a node is one
// fixed-stride record (a lane touches one or two cache lines per candidate), and everything small and
// hot (topology counters, key tables, sorted Allocatable values, the visiting order) lives in LDS.
// ------------------------------------------------------------------------------------------------
#define KS_MAX_HOST 3
#define KS_MAX_REC 24
#define KS_BST_LDS 1024
#ifndef KS_FIRST_WIDTH
#define KS_FIRST_WIDTH 64    // candidates evaluated in a pod's first step (measured: a narrower first step is a wash -- the 11-15 % of pods that need a second step pay a whole extra evaluation)
#endif
#define KS_FAST_G 64        // FAST variant: topology groups (and hostname groups) whose counters live in LDS
#define KS_FAST_S 16        // FAST variant: instance-type-key states
#define KS_FAST_RT 6144     // FAST variant: R*T sorted Allocatable values in LDS (48 KiB)

struct PlanTouch {   // one narrow key the class touches: its own requirement (if any) and the topology items on that key
  u64 mask; i32 gt, lt; i32 key; u8 own; u8 complement; u8 topo_begin, topo_end;
};
struct PlanTopo {    // static part of one matching topology group (getMatchingTopologies, topology.go:351-364)
  u64 PD;            // podDomains.Has(value) over the key's universe
  i32 g; i32 maxskew; u8 type; u8 self; u8 pod_has; u8 hslot; u32 pad;
};
struct PlanRec { i32 g; i32 key; u8 type; u8 owned_inverse; u16 hslot; u8 tidx; u8 filtered; u16 pad; };   // one group Topology.Record must visit (tidx: touch entry holding its key, 0xFF none)
struct alignas(16) ClsPlan {
  u32 c, present, complement; i32 it_state;
  u32 hn_mode, hn_off, hn_cnt, reqmask;
  u64 tol; u32 port_off, port_cnt;
  u32 vol_off, vol_cnt, mono, dyn;     // mono: an existing node that refused this class once refuses it for the rest of the Solve (see the watermark in ks_pack); dyn: see ClsBrief::dyn
  u32 ntouch, ntopo, nhost, nrec;
  i64 req[KS_MAX_RES];
  PlanTouch touch[KS_MAX_TOUCH];
  PlanTopo topo[KS_MAX_TOPO];        // narrow-key items, grouped by touch entry
  PlanTopo host[KS_MAX_HOST];        // hostname-key items
  PlanRec rec[KS_MAX_REC];
  u64 tmask, rmask;                        // groups (bit g & 63) the evaluation reads / Topology.Record may update: round speculation (ks_pack) needs them disjoint
  u32 overflow; u32 eq; u64 tkeys;         // tkeys: touch[i].key packed 5 bits each, so the per-key bit arithmetic of the commit needs no LDS reads
                                           // eq: evaluation-equivalence id (ks_link_plans), 0 = none
};
static_assert(sizeof(ClsPlan) % 16 == 0, "plan records are copied with 16-byte loads");
// What the round planner and resolver (ks_pack, speculation rounds) read of a class.
struct alignas(16) ClsBrief {
  u64 tmask;    // groups (bit g & 63) whose counters the evaluation of ANOTHER node can depend on: a record into one of them by an earlier pod of the round ends the round
  u64 tfull;    // every group the evaluation reads (adds the hostname-keyed groups whose record only touches the winner's own counter)
  u64 rmask;    // groups Topology.Record may update
  u32 ev;       // evaluation class: classes with equal ids are evaluated identically by eval_node (they may differ in what they record)
  u32 flags;    // bit 0: may take part in rounds (plan fits the kernel's limits, no host ports, no volumes)
  u32 reqmask;
  u32 dyn;      // bit 1 (2 | hslot << 8 | (g & 63) << 16): the evaluation's only topology item is a hostname-keyed spread / anti-affinity (row hslot < 16
                // of the hostname tables): whether a candidate still takes the pod depends on how many pods the round has recorded into that group
                // on it, which the resolver counts.
                // bit 0: the evaluation's only topology item is a spread over a group of DevProb::dyn_groups and the pod has no requirement of its own on that
                // key: 1 | g << 8 | selfSelecting << 16.  On a node whose requirement on the key is a single value the item then depends on that value
                // and the group's counts alone, so the resolver can apply it against counts it keeps for the round.
  i64 req[KS_MAX_RES];
  u64 zmask;    // hostname-keyed groups whose item accepts a node only while the node's own counter is 0 (anti-affinity; spread with maxSkew - self == 0)
  u64 rsure;    // hostname-keyed groups this class records into for certain (group present from the start, no node filter): subset of rmask
  i32 dyn_maxskew; u32 dyn_pd;      // dyn: maxSkew and the pod's domains (PlanTopo::PD) over the key's <= 8 values
};
static_assert(sizeof(ClsBrief) == 128, "ClsBrief layout");

// ------------------------------------------------------------------------------------------------
// ks_build_plans: one thread per pod class; everything about a class that does not depend on the
// Solve state is resolved here once (host/encode.cpp produced the CSR lists).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void ks_build_plans(const DevProb* probs) {
  const DevProb& P = probs[blockIdx.y]; ClsPlan* const plans = (ClsPlan*)P.plans; ClsBrief* const briefs = (ClsBrief*)P.briefs;
  const u32 c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= P.C) return;
  ClsPlan pl; memset(&pl, 0, sizeof pl);
  pl.c = c; pl.present = P.cls.present[c]; pl.complement = P.cls.complement[c]; pl.it_state = P.cls.it_state[c];
  pl.hn_mode = P.cls_hn_mode[c]; pl.hn_cnt = P.cls_hn_off[c + 1] - P.cls_hn_off[c]; pl.hn_off = pl.hn_cnt ? P.cls_hn_off[c] : 0;      // (an empty list has no position: classes that differ only there evaluate alike)
  pl.reqmask = P.cls_requests_present[c]; pl.tol = P.cls_tolerated[c]; pl.port_cnt = P.cls_port_off[c + 1] - P.cls_port_off[c]; pl.port_off = pl.port_cnt ? P.cls_port_off[c] : 0;
  pl.vol_cnt = P.cls_vol_off[c + 1] - P.cls_vol_off[c]; pl.vol_off = pl.vol_cnt ? P.cls_vol_off[c] : 0;
  for (u32 r = 0; r < P.R; ++r) pl.req[r] = P.cls_requests[(size_t)c * P.R + r];
  // own requirement keys, ascending
  for (u32 k = 0; k < P.K; ++k) if ((pl.present >> k) & 1u) {
    if (pl.ntouch >= KS_MAX_TOUCH) { pl.overflow = 1; break; }
    PlanTouch& t = pl.touch[pl.ntouch++]; t.key = (i32)k; t.own = 1; t.complement = (pl.complement >> k) & 1u;
    t.mask = P.cls.mask[(size_t)c * P.K + k]; t.gt = P.cls.gt[(size_t)c * P.K + k]; t.lt = P.cls.lt[(size_t)c * P.K + k];
  }
  // matching topology groups: owned first, then inverse groups selecting the pod
  struct Tmp { PlanTopo it; int tidx; } tmp[KS_MAX_TOPO]; int ntmp = 0;
  const u32 ob = P.cls_own_off[c], oe = P.cls_own_off[c + 1], ib = P.cls_isel_off[c], ie = P.cls_isel_off[c + 1];
  for (u32 i = 0; i < (oe - ob) + (ie - ib); ++i) {
    const u32 ent = i < oe - ob ? P.own_list[ob + i] : P.isel_list[ib + (i - (oe - ob))];
    PlanTopo it; it.g = (i32)(ent & 0x7FFFFFFFu); it.self = ent >> 31; it.type = P.grp_type[it.g]; it.maxskew = P.grp_max_skew[it.g]; it.pod_has = 0; it.PD = ~0ull; it.hslot = 0; it.pad = 0;
    const i32 k = P.grp_key[it.g];
    if (k == KS_KEY_HOSTNAME) {
      if (pl.nhost >= KS_MAX_HOST) { pl.overflow = 1; continue; }
      it.hslot = (u8)P.grp_hslot[it.g]; pl.host[pl.nhost++] = it; continue;
    }
    KReq pd = kreq_exists();
    if ((pl.present >> k) & 1u) { pd = load_req(pl.present, pl.complement, P.cls.mask + (size_t)c * P.K, P.cls.gt + (size_t)c * P.K, P.cls.lt + (size_t)c * P.K, k); it.pod_has = 1; }
    it.PD = kreq_has_mask(pd, P.value_int + k * 64, P.key_nvalues[k]);
    int ti = -1; for (u32 j = 0; j < pl.ntouch; ++j) if (pl.touch[j].key == k) { ti = (int)j; break; }
    if (ti < 0) { if (pl.ntouch >= KS_MAX_TOUCH) { pl.overflow = 1; continue; } ti = (int)pl.ntouch; PlanTouch& t = pl.touch[pl.ntouch++]; t.key = k; t.own = 0; t.complement = 0; t.mask = 0; t.gt = KS_NOGT; t.lt = KS_NOLT; }
    if (ntmp >= KS_MAX_TOPO) { pl.overflow = 1; continue; }
    tmp[ntmp].it = it; tmp[ntmp].tidx = ti; ++ntmp;
  }
  for (u32 j = 0; j < pl.ntouch; ++j) {   // group the narrow items by touch entry (order inside an entry is irrelevant: the In-sets intersect)
    pl.touch[j].topo_begin = (u8)pl.ntopo;
    for (int i = 0; i < ntmp; ++i) if (tmp[i].tidx == (int)j) pl.topo[pl.ntopo++] = tmp[i].it;
    pl.touch[j].topo_end = (u8)pl.ntopo;
  }
  // groups Topology.Record visits: non-inverse groups selecting the pod, then inverse groups it owns
  const u32 sb = P.cls_sel_off[c], se = P.cls_sel_off[c + 1], wb = P.cls_iown_off[c], we = P.cls_iown_off[c + 1];
  for (u32 i = 0; i < (se - sb) + (we - wb); ++i) {
    if (pl.nrec >= KS_MAX_REC) { pl.overflow = 1; break; }
    PlanRec& r = pl.rec[pl.nrec++];
    const bool inv = i >= se - sb;
    r.g = (i32)(inv ? P.iown_list[wb + (i - (se - sb))] : P.sel_list[sb + i]);
    r.key = P.grp_key[r.g]; r.type = P.grp_type[r.g]; r.owned_inverse = inv ? 1 : 0; r.hslot = (u16)(P.grp_hslot[r.g] >= 0 ? P.grp_hslot[r.g] : 0);
    if (r.key >= 0 && pl.ntouch < KS_MAX_TOUCH) {   // gather the recorded key with the rest so Topology.Record needs no extra loads
      bool seen = false; for (u32 j = 0; j < pl.ntouch; ++j) if (pl.touch[j].key == r.key) seen = true;
      if (!seen) { PlanTouch& t = pl.touch[pl.ntouch]; t.key = r.key; t.own = 0; t.complement = 0; t.mask = 0; t.gt = KS_NOGT; t.lt = KS_NOLT; t.topo_begin = t.topo_end = (u8)pl.ntopo; ++pl.ntouch; }
    }
  }
  for (u32 i = 0; i < pl.nrec; ++i) {
    PlanRec& r = pl.rec[i]; r.tidx = 0xFF; r.pad = 0;
    r.filtered = (!r.owned_inverse && P.grp_filter_off[r.g] != P.grp_filter_off[r.g + 1]) ? 1 : 0;
    for (u32 j = 0; j < pl.ntouch; ++j) if (pl.touch[j].key == r.key) r.tidx = (u8)j;
    if (r.filtered) {   // a single empty filter term ({}: the pod has no node selector / affinity) always matches
      const u32 fb = P.grp_filter_off[r.g], fe = P.grp_filter_off[r.g + 1];
      bool trivial = false; for (u32 f = fb; f < fe; ++f) if (P.flt.present[f] == 0 && P.flt.it_state[f] == 0) trivial = true;
      if (trivial) r.filtered = 0;
    }
  }
  pl.tkeys = 0; for (u32 j = 0; j < pl.ntouch; ++j) pl.tkeys |= (u64)(u32)pl.touch[j].key << (5 * j);
  pl.tmask = 0; pl.rmask = 0; u64 tfull = 0;
  for (u32 j = 0; j < pl.ntopo; ++j) pl.tmask |= 1ull << (pl.topo[j].g & 63);
  tfull = pl.tmask; for (u32 j = 0; j < pl.nhost; ++j) tfull |= 1ull << (pl.host[j].g & 63);
  for (u32 j = 0; j < pl.nhost; ++j) {
    // A record into a hostname-keyed anti-affinity group, or a spread group every node registered with, changes the winner's own
    // hostname counter only; no other candidate's evaluation reads it, and the winner itself is covered by the order rule of the
    // rounds.  (A spread group created by a later relaxation has unregistered hostnames that a record can turn into accepting ones.)
    const bool own_counter_only = pl.host[j].type == 2 || (pl.host[j].type == 0 && P.grp_active[pl.host[j].g] != 0);
    if (!own_counter_only) pl.tmask |= 1ull << (pl.host[j].g & 63);
  }
  for (u32 j = 0; j < pl.nrec; ++j) pl.rmask |= 1ull << (pl.rec[j].g & 63);
  pl.eq = 0;
  // Rejections by an EXISTING node are monotone for a class that consults no topology group and whose own requirements are all on
  // well-known keys: taints are static, host ports / volumes / requests only accumulate, requirement sets only narrow.  (A custom
  // label the node does not define is the exception -- "label does not have known values" until another pod's NotIn defines it.)
  // Anti-affinity items (own or inverse) keep it monotone: they need a count of 0, counts only grow within a Solve, and every domain such a group will ever
  // know of an existing node is registered before the first pod (topology.go:56-84, existingnode.go:73).  Spread (the minimum moves) and affinity (a domain
  // becomes eligible once it counts) do not.
  bool only_anti = true;
  for (u32 j = 0; j < pl.ntopo; ++j) if (pl.topo[j].type != 2) only_anti = false;
  for (u32 j = 0; j < pl.nhost; ++j) if (pl.host[j].type != 2) only_anti = false;
  pl.mono = (!pl.overflow && only_anti && (pl.present & ~P.wellknown_mask) == 0) ? 1u : 0u;
  pl.dyn = 0;
  if (!pl.overflow && pl.ntopo == 1 && pl.nhost == 0 && pl.topo[0].type == 0 && !pl.topo[0].pod_has && pl.topo[0].g < 64 && ((P.dyn_groups >> pl.topo[0].g) & 1ull) &&
      pl.topo[0].maxskew >= 0 && pl.topo[0].maxskew < (1 << 24) /* RoundCtl::dynq packs it into 24 bits */) pl.dyn = 1u | ((u32)pl.topo[0].g << 8) | ((u32)pl.topo[0].self << 16);
  else if (!pl.overflow && pl.ntopo == 0 && pl.nhost == 1 && pl.host[0].type != 1 && pl.host[0].hslot < 24 && pl.host[0].g >= 0 && pl.host[0].g < 64 && P.G <= 64 && P.GH <= 24)
    pl.dyn = 2u | ((u32)pl.host[0].hslot << 8) | ((u32)pl.host[0].g << 16);
  plans[c] = pl;
  u64 zmask = 0;
  for (u32 j = 0; j < pl.nhost; ++j) if (pl.host[j].type == 2 || (pl.host[j].type == 0 && (i64)pl.host[j].maxskew - (i64)pl.host[j].self <= 0)) zmask |= 1ull << (pl.host[j].g & 63);
  for (u32 j = 0; j < pl.nhost; ++j) if (!(pl.host[j].type == 2 || (pl.host[j].type == 0 && (i64)pl.host[j].maxskew - (i64)pl.host[j].self <= 0))) zmask &= ~(1ull << (pl.host[j].g & 63));   // (two items hashing to one bit: keep the careful answer)
  for (u32 j = 0; j < pl.ntopo; ++j) zmask &= ~(1ull << (pl.topo[j].g & 63));
  u64 rsure = 0;
  for (u32 j = 0; j < pl.nrec; ++j) { const PlanRec& r = pl.rec[j]; if (r.key == KS_KEY_HOSTNAME && (r.owned_inverse || (P.grp_active[r.g] != 0 && !r.filtered))) rsure |= 1ull << (r.g & 63); }
  ClsBrief b; b.zmask = zmask; b.rsure = rsure; b.tmask = pl.tmask; b.tfull = tfull; b.rmask = pl.rmask; bool late_host = false;       // a record into a hostname-keyed group a relaxation creates later: such hostnames may be unregistered
  for (u32 j = 0; j < pl.nrec; ++j) if (pl.rec[j].key == KS_KEY_HOSTNAME && !pl.rec[j].owned_inverse && P.grp_active[pl.rec[j].g] == 0) late_host = true;
  b.ev = 0; b.flags = (!pl.overflow && pl.port_cnt == 0 && pl.vol_cnt == 0 && !late_host) ? 1u : 0u; b.reqmask = pl.reqmask; b.dyn = pl.dyn; b.dyn_maxskew = (pl.dyn & 1u) ? pl.topo[0].maxskew : 0; b.dyn_pd = (pl.dyn & 1u) ? (u32)pl.topo[0].PD : 0u;
  for (u32 r = 0; r < KS_MAX_RES; ++r) b.req[r] = pl.req[r];
  briefs[c] = b;
}

// Evaluation classes: two classes get the same id when eval_node reads identical inputs for both (requests, requirements, tolerations,
// hostname selector, ports, every topology item incl. its self-selecting flag) -- they may still differ in the groups Topology.Record
// visits.  Interned through an open-addressing table: hash of the evaluation part of the plan, confirmed word by word.
__device__ inline bool ks_plan_eval_equal(const ClsPlan& a, const ClsPlan& b) {
  const u32* x = (const u32*)&a; const u32* y = (const u32*)&b;
  constexpr u32 w_present = offsetof(ClsPlan, present) / 4, w_nrec = offsetof(ClsPlan, nrec) / 4, w_req = offsetof(ClsPlan, req) / 4, w_rec = offsetof(ClsPlan, rec) / 4;
  for (u32 i = w_present; i < w_nrec; ++i) if (x[i] != y[i]) return false;
  for (u32 i = w_req; i < w_rec; ++i) if (x[i] != y[i]) return false;
  return a.overflow == b.overflow && a.tkeys == b.tkeys;
}
__device__ inline u64 ks_plan_eval_hash(const ClsPlan& a) {
  const u32* x = (const u32*)&a; u64 h = 0x9E3779B97F4A7C15ull;
  constexpr u32 w_present = offsetof(ClsPlan, present) / 4, w_nrec = offsetof(ClsPlan, nrec) / 4, w_req = offsetof(ClsPlan, req) / 4, w_rec = offsetof(ClsPlan, rec) / 4;
  for (u32 i = w_present; i < w_nrec; ++i) { h ^= x[i]; h *= 0x9FB21C651E98DF25ull; h ^= h >> 29; }
  for (u32 i = w_req; i < w_rec; ++i) { h ^= x[i]; h *= 0x9FB21C651E98DF25ull; h ^= h >> 29; }
  h ^= a.overflow; h *= 0xD6E8FEB86659FD93ull; h ^= a.tkeys; h *= 0xD6E8FEB86659FD93ull; h ^= h >> 32;
  return h;
}
__global__ __launch_bounds__(64) void ks_link_ev(const DevProb* probs) {
  const DevProb& P = probs[blockIdx.y]; const ClsPlan* const plans = (const ClsPlan*)P.plans; ClsBrief* const briefs = (ClsBrief*)P.briefs; u32* const tab = P.ev_tab; const u32 tab_size = P.ev_tab_size, C = P.C;
  const u32 c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const ClsPlan& me = plans[c];
  u32 slot = (u32)ks_plan_eval_hash(me) & (tab_size - 1), ev = 0;
  for (u32 probes = 0; probes < tab_size; ++probes, slot = (slot + 1) & (tab_size - 1)) {
    const u32 prev = atomicCAS(&tab[slot], 0u, c + 1);
    if (prev == 0) { ev = c + 1; break; }                        // first of its kind: it represents the evaluation class
    if (ks_plan_eval_equal(plans[prev - 1], me)) { ev = prev; break; }
  }
  briefs[c].ev = ev ? ev : c + 1;
}

// Two classes are evaluation-equivalent when Node.Add reads exactly the same inputs for both and neither
// consults the topology (they may still differ in the groups Topology.Record updates, e.g. replicas that
// differ only in labels).  For a run of equivalent pods the fit bitmap of one candidate step stays valid:
// only the node that just received a pod changed, and it moved behind the rest of its count bucket.
__device__ inline bool ks_plan_eval_eligible(const ClsPlan& p) { return !p.overflow && p.ntopo == 0 && p.nhost == 0 && p.port_cnt == 0 && p.vol_cnt == 0 && p.hn_mode == 0; }
__global__ __launch_bounds__(64) void ks_link_plans(const DevProb* probs) {
  const DevProb& P = probs[blockIdx.y]; ClsPlan* const plans = (ClsPlan*)P.plans; const ClsBrief* const briefs = (const ClsBrief*)P.briefs;
  const u32 c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= P.C) return;
  // For plans that consult no topology, carry no ports / volumes / hostname selector, "evaluated identically" (ks_link_ev: every word of the
  // evaluation part equal) IS "Node.Add reads the same inputs": the evaluation class interned there -- one hash-table probe per class instead
  // of a scan over every earlier class -- names the set (which member represents it is immaterial: the id only indexes the watermark).
  if (ks_plan_eval_eligible(plans[c])) plans[c].eq = briefs[c].ev;
}

// Hot, small tables and scalars of one Solve, held in registers.  In the FAST kernel variant every
// pointer below derives directly from a __shared__ array, so the compiler emits ds_read/ds_write for
// them (a generic pointer loaded from memory would become a FLAT access, and every FLAT access waits on
// vmcnt(0) AND lgkmcnt(0), serialising all outstanding global loads).
struct Tabs {
  const u32* key_nvalues; const i32* value_int; const u8* its_fail; const u16* its_inter;
  i32* gcnt; u64* g_reg; u64* g_pos; u8* g_active; i32* g_hpos; i32* g_hzero;
  const i64* ge_vals; const u32* ge_cnt;
  u32 K, R, T, TW, GH, E, S, SC, n_ct, wellknown, ge_stride; i32 key_zone, key_ct;
  // hot global arrays, typed with the global address space (pointers loaded from a descriptor in memory
  // would otherwise be generic and every access a FLAT instruction)
  // Only what every pod step touches lives here (SGPRs are scarce: 102 per wave); cold pointers are read
  // from the LDS descriptor at their use site through the G_* accessors below.
  GA u64* q; GA i32* pod_node; GA i32* pod_seq;
  GA u8* rec; u32 rec_stride; GA i32* hcnt; GA u64* n_alive; const GA u64* ge_rows;
};
#define G_lastgen ((GA u32*)S.lastgen)
#define G_lastlen ((GA u32*)S.lastlen)
#define G_pod_stage ((GA i32*)S.pod_stage)
#define G_pod_reason ((GA u32*)S.pod_reason)
#define G_n_tmpl ((GA i32*)S.n_tmpl)
#define G_wm ((GA u32*)S.wm)
#define G_remaining ((GA i64*)S.remaining)
#define G_bstart ((GA u32*)S.bstart)
#define GC(T, p) ((const GA T*)(p))   /* a descriptor pointer read as global memory */
#define G_stage_cls ((const GA u32*)P.stage_cls)
#define G_pod_stage_off ((const GA u32*)P.pod_stage_off)
#define G_grp_filter_off ((const GA u32*)P.grp_filter_off)
#define G_kv_types ((const GA u64*)P.kv_types)
#define G_cmplx_types ((const GA u64*)P.cmplx_types)
#define G_nidnex_types ((const GA u64*)P.nidnex_types)
#define G_pair_types ((const GA u64*)P.pair_types)
#define G_its_types ((const GA u64*)P.its_types)
#define G_grid ((const GA u64*)P.grid)

struct ReqOut {    // per-key requirement of the winning node after the pod is added (entries in Pub::valid only)
  u64 mask[KS_MAX_KEYS]; i32 gt[KS_MAX_KEYS]; i32 lt[KS_MAX_KEYS];
};
// The winner's evaluation, broadcast to the whole wave with v_readlane: wave-uniform registers, so the
// filter and the commit branch on scalars instead of waiting on LDS round trips.
// RM: compile-time bound on the resource count (4 in the LEAN kernel variant, KS_MAX_RES otherwise) -- every loop over
// resources is fully unrolled, so the bound is paid in instructions and registers whether or not R reaches it.
template <int RM> struct PubT {
  u32 slot, present, complement, changed, narrowed, valid, rm, count; i32 it_state, it_before; bool need;
  i64 req_new[RM], room_new[RM];
};
struct TopoDyn { u64 reg, pos; i32 minc; i32 pad; };
struct alignas(16) WaveShared {      // one per wave of the workgroup
  ClsPlan cls; ReqOut rq;
  TopoDyn dyn[KS_MAX_TOPO]; i32 host_anypos[KS_MAX_HOST]; i32 host_zero[KS_MAX_HOST]; i32 pad_hz[2];
  i64 low_new[KS_MAX_RES]; u32 low_idx[KS_MAX_RES];
  u64 la_mask[KS_MAX_TOUCH][64];                                                            // per-lane requirement slots of eval_node
};
struct WaveBounds { i32 la_gt[KS_MAX_TOUCH][64]; i32 la_lt[KS_MAX_TOUCH][64]; };            // ... their Gt/Lt halves (BOUNDS variants only)
struct LeaderShared {                // owned by wave 0, which carries the Solve's sequential state
  u32 hard[8];          // 256-bit hashed set of classes whose last pod found nothing in the first candidate window (heuristic only)
  u8 hslot_of[64];      // group g (< 64) -> row of the hostname tables (grp_hslot), 0xFF if its key is not the hostname
  u32 bstart[KS_BST_LDS];
  u64 ctr[32];          // statistics + (KS_PROBES builds) per-phase cycle counters; slot numbers = ks_result.stats[]
};
#define KS_MAX_WAVES 8
template <int RM> struct RoundCtlT {   // speculation-round hand-off between the leader (wave 0) and the worker waves
  u32 mode2[2], n, nnew, seq0, n_ok, ord_in_lds, nwk;   // mode2: double-buffered by step parity (the leader may plan the next step before a slow wave has read this one's)
  u32 cmd, scan_base, scan_total, scan_cidx;     // scan-ahead service of the sequential path (waves 1.. evaluate the windows after the leader's)
  u32 par, fail_at, pad0, pad1;                  // fail_at: first round pod placed on a node whose instance-type filter came back empty (0xFFFFFFFF: none)
  u64 scanm[KS_MAX_WAVES];                       // scan-ahead: fit bitmap of each helper's window
  u64 qe[2][64];                                 // the round's pods (queue entries); double-buffered like mode2
  u8 pw[2][64];                                  // worker (evaluation class) of round pod i
  u32 wcls[2][KS_MAX_WAVES];                     // class evaluated by worker j
  u64 m[KS_MAX_WAVES], chg[KS_MAX_WAVES];        // per worker: candidates that accept its class / whose requirements a commit of that class would change
  u32 cnt[64], rmsk[64]; i64 room[RM][64], req0[RM][64], low0[RM][64];   // per candidate, class independent (published by worker 0): pods, requested-resource mask, headroom, requests, filter thresholds (Rec::low)
  u8 win[64];                                    // candidate (window lane) of round pod i
  u8 lastpod[64], firstpod[64], npods[64];       // per candidate: last / first round pod placed on it, how many
  u32 rmsk_new[64]; i64 roomrem[RM][64];         // per candidate after the round's pods: requested-resource mask, headroom
  // dynamic spread (DevProb::dyn_groups)
  u64 mo[KS_MAX_WAVES];                          // per worker: candidates that accept its class if the skew test is left aside (superset of m)
  u8 zone[64];                                   // per candidate: the single value its requirement on dyn_key allows (In [v]), 0xFF if it is not of that form
  u32 dynq[2][64];                               // per round pod of a ClsBrief::dyn class: maxSkew | PD << 24 (the dyn word itself rides in the leader's b_flags)
  i32 dd[16][8];                                 // what the round's pods have recorded so far: [slot of the group in dyn_groups][domain]
  // hostname-keyed items that tolerate more than zero pods (ClsBrief::dyn tag 2): the round's certain records per candidate and group
  u32 hrec32[64][6];                             // [candidate][hslot / 4]: 8-bit counters, hslot < 24
  u8 hslack[KS_MAX_WAVES][64];                   // per worker: how many more pods of its group candidate i takes (maxSkew - self - count at the snapshot)
  u64 hz0[KS_MAX_WAVES];                         // per worker (ClsBrief::dyn tag 2): candidates whose own counter of the item's group was 0 at the snapshot
};

// ---- slot record (AoS).  Offsets in bytes; stride = ks_rec_stride(R,K) ----
//   0 u64 taints | 8 u32 present | 12 u32 complement | 16 i32 it_state | 20 u32 reqmask | 24 i32 porthead | 28 u32 count
//   32 i64 room[R] | +8R i64 req[R] | +8R i64 low[R] | 32+24R u64 mask[K] | +8K i32 gt[K] | +4K i32 lt[K]
//   room: resource screen = cap - req, cap being Available() for an existing node (exact) or the max Allocatable
//         over the surviving types for a new one (lazily tightened); header + room[<=4] is one 64-byte line
//   low: the Allocatable value the last instance-type filter rounded each request up to; while the requests
//        stay <= low the filter would select the same ge_rows, i.e. leave InstanceTypeOptions unchanged
__host__ __device__ inline u32 ks_rec_stride(u32 R, u32 K) { return (32 + 24 * R + 16 * K + 15) & ~15u; }
struct Rec {
  GA u8* p; u32 R, K;
  __device__ __forceinline__ GA u64& taints() const { return *(GA u64*)p; }
  __device__ __forceinline__ GA u32& present() const { return *(GA u32*)(p + 8); }
  __device__ __forceinline__ GA u32& complement() const { return *(GA u32*)(p + 12); }
  __device__ __forceinline__ GA i32& it_state() const { return *(GA i32*)(p + 16); }
  __device__ __forceinline__ GA u32& reqmask() const { return *(GA u32*)(p + 20); }
  __device__ __forceinline__ GA i32& porthead() const { return *(GA i32*)(p + 24); }
  __device__ __forceinline__ GA u32& count() const { return *(GA u32*)(p + 28); }
  __device__ __forceinline__ GA i64* room() const { return (GA i64*)(p + 32); }
  __device__ __forceinline__ GA i64* req() const { return (GA i64*)(p + 32 + 8 * R); }
  __device__ __forceinline__ GA i64* low() const { return (GA i64*)(p + 32 + 16 * R); }
  __device__ __forceinline__ GA u64* mask() const { return (GA u64*)(p + 32 + 24 * R); }
  __device__ __forceinline__ GA i32* gt() const { return (GA i32*)(p + 32 + 24 * R + 8 * K); }
  __device__ __forceinline__ GA i32* lt() const { return (GA i32*)(p + 32 + 24 * R + 12 * K); }
};
__device__ __forceinline__ Rec slot_rec(const DevState& S, const Tabs& tb, u32 s) { Rec r; r.p = tb.rec + (size_t)s * tb.rec_stride; r.R = tb.R; r.K = tb.K; return r; }
// BOUNDS == false: no requirement anywhere in the problem carries Gt/Lt, so no node can ever acquire bounds;
// the sentinels become compile-time constants and every within-bounds computation folds away.
template <bool BOUNDS>
__device__ __forceinline__ KReq rec_req(const Rec& r, u32 present, u32 complement, int k) {
  KReq q; q.present = (present >> k) & 1u; q.complement = (complement >> k) & 1u; q.mask = r.mask()[k];
  if constexpr (BOUNDS) { q.gt = r.gt()[k]; q.lt = r.lt()[k]; } else { q.gt = KS_NOGT; q.lt = KS_NOLT; }
  return q;
}

// HostPortUsage.validate, hostportusage.go:81-93 / entry.matches :45-57
__device__ __forceinline__ bool ports_conflict(const DevProb& P, const DevState& S, const ClsPlan& c, i32 head) {
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

// volumeUsage.Validate + VolumeCount.Exceeds for existing node e (existingnode.go:87-94, volumeusage.go:102-143): the class's volumes are
// ordered by driver; per driver, the claims the node does not mount yet are added to its count and compared with its limit.
// COMMIT: volumeUsage.Add (volumeusage.go:94-100) instead of the comparison.
template <bool COMMIT>
__device__ __forceinline__ bool volumes_walk(const DevProb& P, const DevState& S, const ClsPlan& c, u32 e) {
  const u32 ND = P.ND, SW = P.SW; u32 cur = 0xFFFFFFFFu; i32 run = 0;
  for (u32 i = 0; i <= c.vol_cnt; ++i) {
    const u32 ent = i < c.vol_cnt ? P.vol_list[c.vol_off + i] : 0u;
    if (i < c.vol_cnt && ent == 0xFFFFFFFFu) return true;                     // VolumeUsage.validate returned an error
    const u32 d = i < c.vol_cnt ? (ent >> 24) & 63u : 0xFFFFFFFEu;
    if (d != cur) {
      if (cur != 0xFFFFFFFFu) {
        if (COMMIT) S.vol_cnt[(size_t)e * ND + cur] += run;
        else if ((i64)S.vol_cnt[(size_t)e * ND + cur] + run > (i64)P.en_vol_limit[(size_t)e * ND + cur]) return true;   // "would exceed node volume limits"
      }
      cur = d; run = 0;
    }
    if (i == c.vol_cnt) break;
    if (ent >> 31) run += (i32)(ent & 0xFFFFFFu);
    else {
      const u32 id = ent & 0xFFFFFFu; u64& w = S.vol_set[(size_t)e * SW + (id >> 6)];
      if (!((w >> (id & 63u)) & 1ull)) { ++run; if (COMMIT) w |= 1ull << (id & 63u); }
    }
  }
  return false;
}

__device__ __forceinline__ bool kreq_differs(const KReq& x, const KReq& y) { return x.present != y.present || x.mask != y.mask || x.complement != y.complement || x.gt != y.gt || x.lt != y.lt; }

// The popped pod's class scalars, hoisted into wave-uniform registers once per pod (a class field read from
// LDS costs a ~60-cycle dependent round trip at every use inside eval_node).
template <int RM> struct ClsRT { u64 tol, tkeys; u32 reqmask, ntouch, nhost, hn_mode, port_cnt, vol_cnt, eq; i32 it_state; i64 req[RM]; };

// Result of evaluating one node for the current pod: scalars in registers, the per-key requirements in
// per-lane LDS slots (sh.la_*[touch index][lane]) so the algebra below is ONE dynamic loop body instead
// of an unrolled copy per key, and so every lane can read the winner's slots directly.
template <int RM> struct EvT {
  int rc;                       // 0: fails before the instance-type filter; 1: reaches it but fails the resource screen; 2: passes
  u32 present, complement, count, reqmask; i32 it_state, it0;
  u32 tpres, tcomp, tchg, tnar; // per touch index: requirement present / complement after Add; changed; narrowed by topology
  i64 room[RM];                 // the node's resource headroom (Rec::room)
  i64 req[RM], low[RM];           // Rec::req / Rec::low, fetched with the rest so the winner publishes without another round trip
};

// One attempt of Node.Add / ExistingNode.Add up to (not including) the instance-type filter
// (node.go:62-90 / existingnode.go:77-115).  Every lane evaluates its own node.
// `merged`: the pod's own requirements are already folded into the record (a fresh node materialised
// from the template∩class record), only topology is evaluated on top.
template <bool BOUNDS, bool LEAN, int RM>
__device__ __forceinline__ void eval_node(const DevProb& P, const DevState& S, const Tabs& tb, WaveShared& sh, WaveBounds& wb, u32 slot, bool existing, bool merged, EvT<RM>& ev, int lane, u64& tprobe, const ClsRT<RM>& cr, const bool relax_skew = false) {
  // relax_skew (round evaluation of a ClsBrief::dyn class only): a spread item no domain of the node satisfies does not end the evaluation -- it
  // goes on as if every registered domain of the node were allowed and the result is flagged (rc |= 4): "accepts but for the skew", which the
  // resolver re-decides against the counts of the moment.
  const ClsPlan& c = sh.cls; bool skew_failed = false;
  const Rec r = slot_rec(S, tb, slot);
  ev.rc = 0; ev.tpres = 0; ev.tcomp = 0; ev.tchg = 0; ev.tnar = 0;
  // ---- gather: header, requests/capacity, first touched key, hostname counters (independent loads) ----
  const u32x4 h0 = *(const GA u32x4*)r.p, h1 = *(const GA u32x4*)(r.p + 16);
  const u64 taints = (u64)h0.x | ((u64)h0.y << 32); const u32 present = h0.z, complement = h0.w;
  const i32 it0 = (i32)h1.x; const u32 reqmask = h1.y; const i32 porthead = (i32)h1.z;
  ev.present = present; ev.complement = complement; ev.it_state = it0; ev.it0 = it0; ev.reqmask = reqmask; ev.count = h1.w;
#pragma unroll
  for (int i = 0; i < RM; ++i) { ev.room[i] = 0; ev.req[i] = 0; ev.low[i] = INT64_MIN; if ((u32)i < tb.R) { ev.room[i] = r.room()[i]; ev.req[i] = r.req()[i]; ev.low[i] = r.low()[i]; } }
  const u32 ntouch = cr.ntouch;
  KReq nxt = kreq_absent();
  if (ntouch) nxt = rec_req<BOUNDS>(r, present, complement, (int)((u32)cr.tkeys & 31u));
  i32 hc0 = -1, hc1 = -1, hc2 = -1;
  if (cr.nhost > 0) hc0 = tb.hcnt[(size_t)slot * tb.GH + UF(c.host[0].hslot)];
  if (cr.nhost > 1) hc1 = tb.hcnt[(size_t)slot * tb.GH + UF(c.host[1].hslot)];
  if (cr.nhost > 2) hc2 = tb.hcnt[(size_t)slot * tb.GH + UF(c.host[2].hslot)];

  // ---- Taints.Tolerates, taints.go:28-40 ----
  if (taints & ~cr.tol) return;
  // ---- the pod's hostname requirement against the node's `hostname In [own]` ----
  if (!LEAN && !merged && cr.hn_mode != 0) {
    bool inlist = false;
    if (existing) for (u32 i = 0; i < c.hn_cnt; ++i) if (P.hn_list[c.hn_off + i] == slot) { inlist = true; break; }
    if (cr.hn_mode == 1 ? !inlist : inlist) return;
  }
  // ---- HostPortUsage.Validate ----
  if (!LEAN && cr.port_cnt && porthead >= 0 && ports_conflict(P, S, c, porthead)) return;
  // ---- volumeUsage.Validate / VolumeCount.Exceeds (existing nodes only) ----
  if (!LEAN && existing && cr.vol_cnt && volumes_walk<false>(P, S, c, slot)) return;
  // ---- resources: exact for existing nodes (existingnode.go:99-103), a necessary screen for new ones ----
  bool fit = true;
#pragma unroll
  for (int i = 0; i < RM; ++i) if (((reqmask | cr.reqmask) >> i) & 1u) { if (cr.req[i] > ev.room[i]) fit = false; }
  if (existing && !fit) return;
  if (!LEAN && !merged && cr.it_state) { if (tb.its_fail[it0 * tb.SC + cr.it_state]) return; ev.it_state = tb.its_inter[it0 * tb.SC + cr.it_state]; }
  // ---- Topology.AddRequirements on hostname-keyed groups: the node's only hostname domain is its own ----
  for (u32 i = 0; i < cr.nhost; ++i) {
    const PlanTopo& th = c.host[i]; const i32 cnt = i == 0 ? hc0 : (i == 1 ? hc1 : hc2); bool ok;
    struct { u32 type, self; i32 maxskew; } t; { const u32 f = UF(*(const u32*)&th.type); t.type = f & 0xFF; t.self = (f >> 8) & 0xFF; t.maxskew = (i32)UF(th.maxskew); }
    if (t.type == 0) ok = cnt >= 0 && (i64)cnt + t.self <= (i64)t.maxskew;                        // nextDomainTopologySpread, min==0 for hostname (topologygroup.go:184-188)
    else if (t.type == 2) ok = cnt == 0;                                                           // nextDomainAntiAffinity :235-243
    else ok = UF(sh.host_anypos[i]) ? (cnt > 0) : (t.self && cnt >= 0);                                // nextDomainAffinity :202-233
    if (!ok) {
      // which error the reference would raise (only read for fresh nodes): a spread picks among the NODE's domains, so nothing viable is an
      // empty domain set ("unsatisfiable topology constraint", topology.go:160-163); affinity / anti-affinity pick among ALL domains
      // (topologygroup.go:202-243), so a non-empty choice that excludes this node fails one step later, at Compatible (node.go:87)
      const bool elsewhere = t.type == 1 ? UF(sh.host_anypos[i]) != 0 : (t.type == 2 && (i32)UF(sh.host_zero[i]) > 0);
      ev.rc = elsewhere ? -KS_WHY_TOPOLOGY_REQS : -KS_WHY_TOPOLOGY; return;
    }
  }
  // ---- per touched key: Compatible + Add of the pod's own requirement (requirements.go:123-133, :87-94; one
  //      Intersection serves both), then Topology.AddRequirements (topology.go:149-167) and the Compatible + Add
  //      of its result (node.go:83-90).  The next key's node requirement is loaded while this one is processed. ----
  for (u32 i = 0; i < ntouch; ++i) {
    const PlanTouch& tl = c.touch[i]; const int k = (int)((u32)(cr.tkeys >> (5 * i)) & 31u);
    struct { u64 mask; i32 gt, lt; u32 own, complement, topo_begin, topo_end; } t;
    { const u32 f = UF(*(const u32*)&tl.own); t.own = f & 0xFF; t.complement = (f >> 8) & 0xFF; t.topo_begin = (f >> 16) & 0xFF; t.topo_end = f >> 24; t.mask = tl.mask; t.gt = tl.gt; t.lt = tl.lt; }
    KReq a = nxt;
    if (i + 1 < ntouch) nxt = rec_req<BOUNDS>(r, present, complement, (int)((u32)(cr.tkeys >> (5 * (i + 1))) & 31u));
    const KReq orig = a;
    const i32* vi = tb.value_int + k * 64; const u32 nv = tb.key_nvalues[k];
    if (t.own && !merged) {
      KReq b; b.present = true; b.complement = t.complement; b.mask = t.mask; b.gt = BOUNDS ? t.gt : KS_NOGT; b.lt = BOUNDS ? t.lt : KS_NOLT;
      if (!a.present) { if (!((tb.wellknown >> k) & 1u) && !kreq_nidne(b)) return; a = b; }     // "label does not have known values"
      else {
        const KReq mg = kreq_intersect(b, a, vi, nv);
        if (kreq_len0(mg) && !(kreq_nidne(b) && kreq_nidne(a))) return;
        a = mg;
      }
    }
    if (t.topo_end > t.topo_begin) {
      const KReq before = a;
      const KReq nd = before.present ? before : kreq_exists();
      const u64 ND = kreq_has_mask(nd, vi, nv);
      u64 dom = ~0ull; bool constrained = false;
      for (u32 j = t.topo_begin; j < t.topo_end; ++j) {
        const PlanTopo& tg = c.topo[j]; const TopoDyn& d = sh.dyn[j]; u64 options = 0;
        if (d.reg == 0 && d.pos == ~0ull) continue;               // (derived what-if) an inverse anti-affinity group none of whose owners is around: it does not exist
        constrained = true;
        struct { u64 PD; i32 g, maxskew; u32 type, self, pod_has; } tt;
        { const u32 f = UF(*(const u32*)&tg.type); tt.type = f & 0xFF; tt.self = (f >> 8) & 0xFF; tt.pod_has = (f >> 16) & 0xFF; tt.g = (i32)UF(tg.g); tt.maxskew = (i32)UF(tg.maxskew); tt.PD = tg.PD; }
        if (tt.type == 0) {                                       // spread: nextDomainTopologySpread :155-182
          i32 best = INT32_MAX; int bestv = -1;
          for (u64 bits = d.reg & ND; bits; bits &= bits - 1) {
            const int dd = __builtin_ctzll(bits);
            const i32 cnt = tb.gcnt[(size_t)tt.g * 64 + dd] + tt.self;
            if ((i64)cnt - (i64)d.minc <= (i64)tt.maxskew && cnt < best) { best = cnt; bestv = dd; }
          }
          if (bestv >= 0) options = 1ull << bestv;
          else if (relax_skew) { skew_failed = true; options = d.reg & ND; }
        } else if (tt.type == 1) {                                // affinity: nextDomainAffinity :202-233
          options = d.reg & tt.PD & d.pos;
          if (!options && tt.self) {
            KReq pd = kreq_exists();
            if (tt.pod_has) { pd.present = true; pd.complement = t.complement; pd.mask = t.mask; pd.gt = BOUNDS ? t.gt : KS_NOGT; pd.lt = BOUNDS ? t.lt : KS_NOLT; }
            const u64 I = kreq_has_mask(kreq_intersect(pd, nd, vi, nv), vi, nv);
            const u64 x = d.reg & I, y = d.reg & tt.PD;
            if (x) options |= x & (~x + 1);
            if (y) options |= y & (~y + 1);
          }
        } else options = d.reg & tt.PD & ~d.pos;                  // anti-affinity :235-243
        if (!options) { ev.rc = -KS_WHY_TOPOLOGY; return; }        // "unsatisfiable topology constraint"
        dom &= options;
      }
      // nodeRequirements.Compatible(topologyRequirements) on this key: the topology requirement is
      // node ∩ In[dom]; see DESIGN.md "topology compatibility" for the reduction used here.
      if (constrained) {
      const KReq in = kreq_in(dom);
      if (!before.present) { if (!((tb.wellknown >> k) & 1u)) { ev.rc = -KS_WHY_TOPOLOGY_REQS; return; } a = in; }
      else { const KReq mg = kreq_intersect(in, before, vi, nv); if (kreq_len0(mg) && !kreq_nidne(before)) { ev.rc = -KS_WHY_TOPOLOGY_REQS; return; } a = mg; }
      if (kreq_differs(a, before)) ev.tnar |= 1u << i;
      }
    }
    if (a.present) ev.tpres |= 1u << i;
    if (a.complement) ev.tcomp |= 1u << i;
    if (kreq_differs(a, orig)) ev.tchg |= 1u << i;
    sh.la_mask[i][lane] = a.mask; if constexpr (BOUNDS) { wb.la_gt[i][lane] = a.gt; wb.la_lt[i][lane] = a.lt; }
  }
  ev.rc = (fit ? 2 : 1) | (skew_failed ? 4 : 0);
}

// Synchronisation.  The sequential path runs on wave 0 alone:
//   LSYNC: cross-lane hand-off through LDS.  LDS instructions of one wave execute in program order, so only
//          the compiler must be stopped from reordering -- no instruction is emitted.
//   GSYNC: cross-lane hand-off through GLOBAL memory: the writer's stores must have completed
//          (s_waitcnt vmcnt(0)) before another lane's load is issued; costs a store round trip, so it is used
//          once per pod (before the candidate scan re-reads node records) and on rare paths.
//   __syncthreads(): hand-off between the waves of a multi-wave workgroup (speculation rounds).
#define CTR(i, v) do { if (lane == 0 && wv == 0) ls.ctr[(i)] += (v); } while (0)
#ifdef KS_SIM
#define LSYNC() do { uint64_t pm_; (void)ks_sim::exchange(0, &pm_); } while (0)
#define GSYNC() do { uint64_t pm_; (void)ks_sim::exchange(0, &pm_); } while (0)
#else
#define LSYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
#define GSYNC() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory")   /* wave-local: only wave 0 runs the sequential path */
#endif

// Publish the winning lane's evaluation: scalars by v_readlane into wave-uniform registers (Pub), the per-key
// requirements by one parallel LDS copy (lane i moves touch entry i).  Also decides whether the instance-type
// filter can change anything at all (see Rec::low).
template <bool BOUNDS, int RM>
__device__ __forceinline__ void publish_eval(const Tabs& tb, WaveShared& sh, const WaveBounds& wb, const EvT<RM>& ev, u32 slot, bool fresh, int lane, int win, const ClsRT<RM>& cr, PubT<RM>& p) {
  p.slot = RL(slot, win);
  const u32 tpres = RL(ev.tpres, win), tcomp = RL(ev.tcomp, win), tchg = RL(ev.tchg, win), tnar = RL(ev.tnar, win);
  u32 np = RL(ev.present, win), nc = RL(ev.complement, win), changed = 0, narrowed = 0, valid = 0;
  for (u32 i = 0; i < cr.ntouch; ++i) {
    const u32 kb = 1u << ((u32)(cr.tkeys >> (5 * i)) & 31u);
    if ((tpres >> i) & 1u) { np |= kb; nc = ((tcomp >> i) & 1u) ? (nc | kb) : (nc & ~kb); }
    if ((tchg >> i) & 1u) changed |= kb;
    if ((tnar >> i) & 1u) narrowed |= kb;
    valid |= kb;
  }
  if ((u32)lane < cr.ntouch) {
    const u32 k = (u32)(cr.tkeys >> (5 * lane)) & 31u;
    sh.rq.mask[k] = sh.la_mask[lane][win];
    if constexpr (BOUNDS) { sh.rq.gt[k] = wb.la_gt[lane][win]; sh.rq.lt[k] = wb.la_lt[lane][win]; } else { sh.rq.gt[k] = KS_NOGT; sh.rq.lt[k] = KS_NOLT; }
  }
  p.present = np; p.complement = nc; p.changed = changed; p.narrowed = narrowed; p.valid = valid;
  p.it_state = (i32)RL(ev.it_state, win); p.it_before = (i32)RL(ev.it0, win); p.count = RL(ev.count, win);
  p.rm = RL(ev.reqmask, win) | cr.reqmask;
  bool need = fresh || changed != 0 || p.it_state != p.it_before;
#pragma unroll
  for (int i = 0; i < RM; ++i) {
    p.req_new[i] = 0; p.room_new[i] = 0;
    if ((u32)i < tb.R) {
      const i64 v = (i64)RL64(ev.req[i], win) + cr.req[i]; p.req_new[i] = v; p.room_new[i] = (i64)RL64(ev.room[i], win) - cr.req[i];
      if (((p.rm >> i) & 1u) && v > (i64)RL64(ev.low[i], win)) need = true;
    }
  }
  p.need = need;
  LSYNC();
}

// The node's requirement on key k after the Add that is being committed (published entries, else the record).
template <class PUB>
__device__ __forceinline__ KReq new_req(const PUB& pb, const WaveShared& sh, const Rec& r, int k) {
  const ReqOut& o = sh.rq; KReq q; q.present = (pb.present >> k) & 1u; q.complement = (pb.complement >> k) & 1u;
  if ((pb.valid >> k) & 1u) { q.mask = o.mask[k]; q.gt = o.gt[k]; q.lt = o.lt[k]; } else { q.mask = r.mask()[k]; q.gt = r.gt()[k]; q.lt = r.lt()[k]; }
  return q;
}

// T-bit mask word of the types that pass `instanceType.Requirements.Intersects` on key k against node
// requirement B (derivation in DESIGN.md): types lacking the key always pass.
__device__ __forceinline__ u64 pass_types_word(const DevProb& P, const Tabs& tb, int k, const KReq& B, u32 w) {
  const i32* vi = tb.value_int + k * 64; const u32 nv = tb.key_nvalues[k];
  u64 acc = 0;
  for (u64 bits = kreq_has_mask(B, vi, nv); bits; bits &= bits - 1) acc |= G_kv_types[((size_t)k * 64 + __builtin_ctzll(bits)) * tb.TW + w];
  if (B.complement) acc |= G_cmplx_types[(size_t)k * tb.TW + w];
  if (kreq_nidne(B)) acc |= G_nidnex_types[(size_t)k * tb.TW + w];
  return acc;
}
// hasOffering (node.go:151-159) as a T-bit mask word
template <class PUB>
__device__ __forceinline__ u64 offer_types_word(const DevProb& P, const Tabs& tb, const PUB& pb, const WaveShared& sh, const Rec& r, u32 w) {
  u64 allowZ = ~0ull, allowC = ~0ull;
  if (tb.key_zone >= 0 && ((pb.present >> tb.key_zone) & 1u)) allowZ = kreq_has_mask(new_req(pb, sh, r, tb.key_zone), tb.value_int + tb.key_zone * 64, tb.key_nvalues[tb.key_zone]);
  if (tb.key_ct >= 0 && ((pb.present >> tb.key_ct) & 1u)) allowC = kreq_has_mask(new_req(pb, sh, r, tb.key_ct), tb.value_int + tb.key_ct * 64, tb.key_nvalues[tb.key_ct]);
  if (tb.n_ct == 0) return ~0ull;
  u64 acc = 0; const u64 cm = allowC & ((1ull << tb.n_ct) - 1);
  for (u64 zz = allowZ; zz; zz &= zz - 1) {
    const int z = __builtin_ctzll(zz); if ((u32)z * tb.n_ct >= 64) break;
    for (u64 cb = cm; cb; cb &= cb - 1) acc |= G_pair_types[((size_t)z * tb.n_ct + __builtin_ctzll(cb)) * tb.TW + w];
  }
  return acc;
}

// TopologyNodeFilter.MatchesRequirements, topologynodefilter.go:57-70
template <class PUB>
__device__ __forceinline__ bool filter_matches(const DevProb& P, const Tabs& tb, int g, const PUB& pb, const WaveShared& sh, const Rec& r) {
  const u32 b = G_grp_filter_off[g], e = G_grp_filter_off[g + 1];
  if (b == e) return true;
  for (u32 f = b; f < e; ++f) {
    bool ok = true;
    const u32 fp = P.flt.present[f], fc = P.flt.complement[f];
    for (u32 bits = fp; bits && ok; bits &= bits - 1) {
      const int k = __builtin_ctz(bits);
      const KReq a = new_req(pb, sh, r, k);
      const KReq in = load_req(fp, fc, P.flt.mask + (size_t)f * tb.K, P.flt.gt + (size_t)f * tb.K, P.flt.lt + (size_t)f * tb.K, k);
      if (kreq_compatible_fail(a, in, (tb.wellknown >> k) & 1u, tb.value_int + k * 64, tb.key_nvalues[k])) ok = false;
    }
    if (ok && P.flt.it_state[f] && tb.its_fail[pb.it_state * tb.SC + P.flt.it_state[f]]) ok = false;
    if (ok) return true;
  }
  return false;
}

// ATOMIC: several waves commit pods of one speculation round at the same time; two of them may count into the same
// domain.  max(c, 0) is idempotent and the increments commute, so the result does not depend on the interleaving.
template <bool ATOMIC>
__device__ __forceinline__ void grp_record(const Tabs& tb, int g, int d) {   // TopologyGroup.Record, topologygroup.go:101-105
  i32& c = tb.gcnt[(size_t)g * 64 + d];
  if constexpr (ATOMIC) { atomicMax(&c, 0); atomicAdd(&c, 1); atomicOr((unsigned long long*)&tb.g_reg[g], 1ull << d); atomicOr((unsigned long long*)&tb.g_pos[g], 1ull << d); }
  else { c = c < 0 ? 1 : c + 1; tb.g_reg[g] |= 1ull << d; tb.g_pos[g] |= 1ull << d; }
}
template <bool ATOMIC>
__device__ __forceinline__ void grp_record_host(const DevState& S, const Tabs& tb, int h, u32 slot) {
  GA i32& c = tb.hcnt[(size_t)slot * tb.GH + h];
  if constexpr (ATOMIC) {      // several pods of one round may land on the same node: every transition old -> new is taken exactly once
    // Rounds only admit classes whose hostname-keyed records go to groups that exist from the start (ClsBrief.flags), and every node registers
    // its hostname with those (NewNode / NewExistingNode): the counter is never the "unregistered" -1 here, one atomic add does it.
    const i32 old = __hip_atomic_fetch_add(&c, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (old == 0) { atomicAdd(&tb.g_hpos[h], 1); atomicSub(&tb.g_hzero[h], 1); }
    return;
  }
  if (c <= 0) tb.g_hpos[h]++;
  if (c == 0) tb.g_hzero[h]--;      // one registered zero-count hostname fewer
  c = c < 0 ? 1 : c + 1;
}
// Topology.Record, topology.go:120-143: lane i handles the i-th group of the class's record list
// (distinct groups, so the lanes never touch the same counters).
template <bool ATOMIC, class PUB>
__device__ __forceinline__ void topology_record(const DevProb& P, const DevState& S, const Tabs& tb, const PUB& pb, const WaveShared& sh, const Rec& r, u32 slot, int lane) {
  const ClsPlan& c = sh.cls;
  if ((u32)lane >= c.nrec) return;
  const PlanRec& pr = c.rec[lane]; const int g = pr.g;
  if (!pr.owned_inverse) {
    if (!tb.g_active[g]) return;
    if (pr.filtered && !filter_matches(P, tb, g, pb, sh, r)) return;         // TopologyGroup.Counts, topologygroup.go:109-111 (filtered == 0: no filter, or one that matches every node)
  }
  if (pr.key == KS_KEY_HOSTNAME) { grp_record_host<ATOMIC>(S, tb, pr.hslot, slot); return; }   // the node requirement is `hostname In [own]`
  const KReq q = new_req(pb, sh, r, pr.key);
  if (!q.present) return;                                            // Get() of a missing key is Exists: no values, Len != 1
  if (pr.owned_inverse || pr.type == 2) { for (u64 b = q.mask; b; b &= b - 1) grp_record<ATOMIC>(tb, g, __builtin_ctzll(b)); }   // Values(): for a complement set the excluded values
  else if (!q.complement && __builtin_popcountll(q.mask) == 1) grp_record<ATOMIC>(tb, g, __builtin_ctzll(q.mask));
}

// Minimum of a 32-bit value over the wave, returned wave-uniform: four row shifts and two row broadcasts on the data-parallel-primitive
// path (no LDS round trips), the result sits in lane 63.  Every lane takes part (callers pass 0xFFFFFFFF for lanes that do not count).
#ifdef KS_SIM
__device__ __forceinline__ u32 wave_min_u32(u32 v) { return ks_sim::wave_min_u32(v); }
__device__ __forceinline__ u32 wave_or_u32(u32 v) { return ks_sim::wave_or_u32(v); }
__device__ __forceinline__ i64 wave_max_i64(i64 v) { return ks_sim::wave_max_i64(v); }
__device__ __forceinline__ u32 lanes8_min_u32(u32 v) { return ks_sim::wave_min_u32((threadIdx.x & 63) < 8 ? v : 0xFFFFFFFFu); }      // (every lane takes part in the exchange; lanes 0..7 count)
#else
__device__ __forceinline__ u32 wave_min_u32(u32 v) {
  const int id = (int)0xFFFFFFFFu;
  v = min(v, (u32)__builtin_amdgcn_update_dpp(id, (int)v, 0x111, 0xF, 0xF, false));   // row_shr:1
  v = min(v, (u32)__builtin_amdgcn_update_dpp(id, (int)v, 0x112, 0xF, 0xF, false));   // row_shr:2
  v = min(v, (u32)__builtin_amdgcn_update_dpp(id, (int)v, 0x114, 0xF, 0xF, false));   // row_shr:4
  v = min(v, (u32)__builtin_amdgcn_update_dpp(id, (int)v, 0x118, 0xF, 0xF, false));   // row_shr:8  -> lane 15 of every row holds the row's minimum
  v = min(v, (u32)__builtin_amdgcn_update_dpp(id, (int)v, 0x142, 0xA, 0xF, false));   // row_bcast:15 into rows 1 and 3
  v = min(v, (u32)__builtin_amdgcn_update_dpp(id, (int)v, 0x143, 0xC, 0xF, false));   // row_bcast:31 into rows 2 and 3
  return (u32)__builtin_amdgcn_readlane((int)v, 63);
}
// Bitwise OR of a 32-bit value over the wave, wave-uniform (same path; lanes that do not count pass 0).
__device__ __forceinline__ u32 wave_or_u32(u32 v) {
  v |= (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false);
  v |= (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false);
  v |= (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false);
  v |= (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, false);
  v |= (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);
  v |= (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);
  return (u32)__builtin_amdgcn_readlane((int)v, 63);
}
// ... over lanes 0..7 only (three row shifts; the result sits in lane 7)
__device__ __forceinline__ u32 lanes8_min_u32(u32 v) {
  const int id = (int)0xFFFFFFFFu;
  v = min(v, (u32)__builtin_amdgcn_update_dpp(id, (int)v, 0x111, 0xF, 0xF, false));
  v = min(v, (u32)__builtin_amdgcn_update_dpp(id, (int)v, 0x112, 0xF, 0xF, false));
  v = min(v, (u32)__builtin_amdgcn_update_dpp(id, (int)v, 0x114, 0xF, 0xF, false));
  return (u32)__builtin_amdgcn_readlane((int)v, 7);
}
__device__ __forceinline__ i64 wave_max_i64(i64 v) { for (int off = 32; off > 0; off >>= 1) { const i64 o = __shfl_xor(v, off); if (o > v) v = o; } return v; }
#endif

// Stage the pod's class plan in LDS (one coalesced copy) and evaluate the per-pod, node-independent part
// of its topology groups (domainMinCount, topologygroup.go:184-200).
// The plan itself reaches LDS from registers: it was prefetched while the previous pod was being placed.
constexpr u32 KS_PLAN_V = sizeof(ClsPlan) / 16;
static_assert(KS_PLAN_V <= 128, "a class plan is prefetched as two 16-byte registers per lane");
__device__ __forceinline__ void stage_class(const Tabs& tb, WaveShared& sh, int lane) {
  LSYNC();
  const ClsPlan& L = sh.cls;
  if ((u32)lane < L.ntopo) {
    const PlanTopo& t = L.topo[lane]; TopoDyn d; d.reg = tb.g_reg[t.g]; d.pos = tb.g_pos[t.g]; d.pad = 0;
    i32 mn = INT32_MAX;
    for (u64 b = d.reg & t.PD; b; b &= b - 1) { const i32 cn = tb.gcnt[(size_t)t.g * 64 + __builtin_ctzll(b)]; if (cn < mn) mn = cn; }
    d.minc = mn; sh.dyn[lane] = d;
  }
  if (lane >= 32 && (u32)(lane - 32) < L.nhost) { sh.host_anypos[lane - 32] = tb.g_hpos[L.host[lane - 32].hslot] > 0; sh.host_zero[lane - 32] = tb.g_hzero[L.host[lane - 32].hslot]; }
  LSYNC();
}

// lower_bound over the ascending distinct Allocatable values of every requested resource with a 64-ary
// search: every lane probes one pivot, __ballot narrows the interval (two rounds cover 4096 values).  All
// resources advance together so their LDS reads overlap.  idx[r] == ge_cnt[r] means "no type has that much".
template <int RM>
__device__ __forceinline__ void ge_row_indices(const Tabs& tb, const PubT<RM>& pb, u32 reqmask, int lane, u32 (&idx)[RM]) {
  u32 lo[RM], hi[RM];
#pragma unroll
  for (int r = 0; r < RM; ++r) { lo[r] = 0; hi[r] = ((reqmask >> r) & 1u) ? tb.ge_cnt[r] : 0; }
  for (;;) {
    bool busy = false;
#pragma unroll
    for (int r = 0; r < RM; ++r) if (lo[r] < hi[r]) {
      busy = true;
      const u32 span = hi[r] - lo[r], step = (span + 63) >> 6;
      const u32 p = lo[r] + (u32)lane * step;
      const bool in = p < hi[r];
      const bool ge = in && tb.ge_vals[(size_t)r * tb.ge_stride + p] >= pb.req_new[r];
      const u64 b = ballot64(ge); const u32 npiv = __builtin_popcountll(ballot64(in));
      if (!b) lo[r] = lo[r] + (npiv - 1) * step + 1;
      else {
        const u32 f = __builtin_ctzll(b);
        if (step == 1) { lo[r] = hi[r] = lo[r] + f; }
        else { hi[r] = lo[r] + f * step; if (f > 0) lo[r] = lo[r] + (f - 1) * step + 1; }
      }
    }
    if (!busy) break;
  }
#pragma unroll
  for (int r = 0; r < RM; ++r) idx[r] = lo[r];
}

// Instance-type filter (filterInstanceTypesByRequirements, node.go:137-141) on T-bit masks, one wave:
//   alive' = alive & passTypes(changed keys) & its_types(state) & offerings & AND_r ge_rows[r][row(requests[r])]
// Lane w owns word w; there is no per-type loop: resources.Fits is one precomputed row per requested resource.
template <int RM>
__device__ __forceinline__ bool filter_types(const DevProb& P, const Tabs& tb, const PubT<RM>& pb, WaveShared& sh, const Rec& r, const GA u64* alive_in, GA u64* alive_out, u32 reqmask_new,
                             u32 changed_keys, bool check_offer, bool check_it, int lane, u64& tprobe, u64 (&word)[2]) {   // alive_out == nullptr (TW <= 128 only): the result stays in `word` (lane l: words l and 64+l)
  const GA u64* rows[RM]; u32 ridx[RM];
  ge_row_indices(tb, pb, reqmask_new, lane, ridx);
  bool none = false;
#pragma unroll
  for (int i = 0; i < RM; ++i) {
    rows[i] = tb.ge_rows;      // (a resource that is not requested reads row 0 and ignores it: every row read is unconditional, so they are in flight together)
    if ((reqmask_new >> i) & 1u) { if (ridx[i] >= tb.ge_cnt[i]) none = true; else rows[i] = tb.ge_rows + ((size_t)i * tb.T + ridx[i]) * tb.TW; }
  }
  word[0] = 0; word[1] = 0;
  if (none) { if (alive_out) for (u32 w = lane; w < tb.TW; w += 64) alive_out[w] = 0; LSYNC(); return false; }   // nothing has that much of some resource
#pragma unroll
  for (int i = 0; i < RM; ++i) if (lane == 0) { if ((reqmask_new >> i) & 1u) { sh.low_new[i] = tb.ge_vals[(size_t)i * tb.ge_stride + ridx[i]]; sh.low_idx[i] = ridx[i]; } else sh.low_idx[i] = 0xFFFFu; }
  bool any = false;
  for (u32 wbase = 0; wbase < tb.TW; wbase += 64) {
    const u32 w = wbase + lane; u64 a = 0;
    if (w < tb.TW) {
      // every load below is independent of the others: one memory round trip, not one per term
      u64 gr[RM];
#pragma unroll
      for (int i = 0; i < RM; ++i) gr[i] = rows[i][w];
      a = alive_in[w];
#pragma unroll
      for (int i = 0; i < RM; ++i) a &= ((reqmask_new >> i) & 1u) ? gr[i] : ~0ull;
      u64 x = ~0ull;
      for (u32 bits = changed_keys; bits; bits &= bits - 1) { const int k = __builtin_ctz(bits); x &= pass_types_word(P, tb, k, new_req(pb, sh, r, k), w); }
      if (check_it) x &= G_its_types[(size_t)pb.it_state * tb.TW + w];
      if (check_offer) x &= offer_types_word(P, tb, pb, sh, r, w);
      a &= x;
      if (alive_out) alive_out[w] = a;
    }
    if (wbase == 0) word[0] = a; else if (wbase == 64) word[1] = a;
    if (ballot64(a != 0)) any = true;
  }
  LSYNC();
  return any;
}

// Per-resource maximum Allocatable over a node's surviving types: the (necessary) resource screen of
// eval_node.  Recomputed lazily -- only after a candidate passed the screen but failed the filter.
template <int RM>
__device__ __forceinline__ void recompute_cap(const DevProb& P, const Tabs& tb, const GA u64* alive, const Rec& rec, int lane) {
  i64 mx[RM];
#pragma unroll
  for (int r = 0; r < RM; ++r) mx[r] = INT64_MIN;
  for (u32 t = lane; t < tb.TW * 64; t += 64) {
    const bool on = t < tb.T && ((alive[t >> 6] >> (t & 63)) & 1ull);
#pragma unroll
    for (int r = 0; r < RM; ++r) if ((u32)r < tb.R && on) { const i64 al = P.it_alloc[(size_t)r * tb.T + t]; if (al > mx[r]) mx[r] = al; }
  }
#pragma unroll
  for (int r = 0; r < RM; ++r) if ((u32)r < tb.R) { const i64 v = wave_max_i64(mx[r]); if (lane == 0) rec.room()[r] = v - rec.req()[r]; }
  LSYNC();
}

// Write the winning node's record after Add: only what changed (lane-parallel stores).
template <bool BOUNDS, int RM>
__device__ __forceinline__ void write_record(const Tabs& tb, const Rec& r, const PubT<RM>& pb, const WaveShared& sh, u32 reqmask_new, int lane) {
  if ((u32)lane < tb.K && ((pb.changed >> lane) & 1u)) { r.mask()[lane] = sh.rq.mask[lane]; if constexpr (BOUNDS) { r.gt()[lane] = sh.rq.gt[lane]; r.lt()[lane] = sh.rq.lt[lane]; } }
  if (lane >= 32 && (u32)lane < 32 + tb.R) {
    i64 rq = 0, ro = 0;
#pragma unroll
    for (int i = 0; i < RM; ++i) if (lane - 32 == i) { rq = pb.req_new[i]; ro = pb.room_new[i]; }
    r.req()[lane - 32] = rq; r.room()[lane - 32] = ro;
  }
  if (lane == 63) { r.present() = pb.present; r.complement() = pb.complement; r.it_state() = pb.it_state; r.reqmask() = reqmask_new; }
}

// FAST variant: the small hot tables of one Solve as true LDS arrays
struct alignas(16) FastTabs {
  u64 g_reg[KS_FAST_G]; u64 g_pos[KS_FAST_G]; i32 g_hpos[KS_FAST_G]; i32 g_hzero[KS_FAST_G]; u32 key_nvalues[KS_MAX_KEYS]; u32 ge_cnt[KS_MAX_RES];
  u16 its_inter[KS_FAST_S * KS_FAST_S]; u8 its_fail[KS_FAST_S * KS_FAST_S]; u8 g_active[KS_FAST_G];
  i32 gcnt[KS_FAST_G * 64]; i32 value_int[KS_MAX_KEYS * 64];
};
struct alignas(16) NoTabs { u32 pad[4]; };
template <bool FAST, bool BOUNDS, int NW, int RM> struct alignas(16) PackLds {
  alignas(16) unsigned char rc_raw[NW > 1 ? sizeof(RoundCtlT<RM>) : 16]; LeaderShared ls; typename std::conditional<FAST, FastTabs, NoTabs>::type ft; DevProb P; DevState S;
  alignas(16) unsigned char wbs_raw[BOUNDS ? sizeof(WaveBounds) * NW : 16]; WaveShared shw[NW];      // (no Gt/Lt anywhere in the problem: the bounds slots are never touched)
};
#ifdef KS_SIM
static unsigned char ks_dyn_lds[160 * 1024] __attribute__((aligned(16)));
#else
extern __shared__ __attribute__((aligned(16))) unsigned char ks_dyn_lds[];
#endif

#if !defined(KS_SIM) || defined(KS_SIM_PACK)      /* (the emulator's default build runs the register-resident kernel only; -DKS_SIM_PACK compiles ks_pack for it too: round 5) */
// LEAN: no class has host ports, a hostname selector or an instance-type requirement, no provisioner has limits,
// R <= 4 and no statistics are requested -- the code for all of that (and half of every unrolled resource loop) is
// compiled out.  One wave issues ~1 instruction per 5 cycles, so instructions, not bytes, are what a Solve costs.
template <bool FAST, bool BOUNDS, bool LEAN, int NW>
__global__ __launch_bounds__(64 * NW) void ks_pack(const DevProb* probs, const DevState* states, u32 lds_bytes) {
  constexpr int RM = LEAN ? 4 : KS_MAX_RES;
  using ClsR = ClsRT<RM>; using Ev = EvT<RM>; using Pub = PubT<RM>;
  static_assert(NW >= 1 && NW <= KS_MAX_WAVES, "wave count");
  // descriptors are copied to LDS: loads from them can then be CSE'd across global stores (no aliasing)
  // ONE static LDS object, hot lane-indexed arrays first: a ds instruction carries a 16-bit immediate offset, so everything in the first
  // 64 KiB is addressed as lane*stride + immediate; an array beyond that needs its base in a register of its own, which the compiler hoists
  // out of the Solve loop and -- the 8-wave kernel sits at its 256-VGPR budget -- spills to scratch (a memory round trip per reload).
  __shared__ PackLds<FAST, BOUNDS, NW, RM> L;
  DevProb& P_lds = L.P; DevState& S_lds = L.S;
  WaveShared (&shw)[NW] = L.shw; WaveBounds* const wbs = (WaveBounds*)L.wbs_raw; LeaderShared& ls = L.ls;
  RoundCtlT<RM>& rc = *(RoundCtlT<RM>*)L.rc_raw;      // (single-wave kernels have no rounds: nothing of rc is touched)
  const int lane = threadIdx.x & 63; const u32 wv = NW > 1 ? UF(threadIdx.x >> 6) : 0u;
  WaveShared& sh = shw[wv]; WaveBounds& wb = wbs[BOUNDS ? wv : 0];
#ifdef KS_CHECK   /* debug builds: LDS starts out as garbage from whatever ran before -- make that garbage deterministic and hostile */
  { u32* z = (u32*)&L; for (u32 i = threadIdx.x; i < sizeof(L) / 4; i += 64 * NW) z[i] = 0xA5A5A5A5u; u32* y = (u32*)ks_dyn_lds; for (u32 i = threadIdx.x; i < lds_bytes / 4; i += 64 * NW) y[i] = 0xA5A5A5A5u; }
  __syncthreads();
#endif
  { const u32* src = (const u32*)&probs[blockIdx.x]; u32* dst = (u32*)&P_lds; for (u32 i = threadIdx.x; i < sizeof(DevProb) / 4; i += 64 * NW) dst[i] = src[i]; }
  { const u32* src = (const u32*)&states[blockIdx.x]; u32* dst = (u32*)&S_lds; for (u32 i = threadIdx.x; i < sizeof(DevState) / 4; i += 64 * NW) dst[i] = src[i]; }
  __syncthreads();
  const u64 t_start = __builtin_readcyclecounter();
  const DevProb& P = P_lds;
  const DevState& S = S_lds;
  Tabs tb;
  tb.K = UF(P.K); tb.R = UF(P.R); tb.T = UF(P.T); tb.TW = UF(P.TW); tb.GH = UF(P.GH); tb.E = UF(P.E); tb.S = UF(P.S); tb.SC = UF(P.SC); tb.n_ct = UF(P.n_ct); tb.wellknown = UF(P.wellknown_mask); tb.key_zone = (i32)UF(P.key_zone); tb.key_ct = (i32)UF(P.key_ct);
  tb.key_nvalues = P.key_nvalues; tb.value_int = P.value_int; tb.its_fail = P.its_fail; tb.its_inter = P.its_inter;
  tb.q = (GA u64*)UF64((u64)S.q); tb.pod_node = (GA i32*)UF64((u64)S.pod_node); tb.pod_seq = (GA i32*)UF64((u64)S.pod_seq);
  tb.rec = (GA u8*)UF64((u64)S.rec); tb.rec_stride = UF(S.rec_stride); tb.hcnt = (GA i32*)UF64((u64)S.hcnt); tb.n_alive = (GA u64*)UF64((u64)S.n_alive); tb.ge_rows = (const GA u64*)UF64((u64)P.ge_rows);
  const u32 nP = UF(P.P), nM = UF(P.M), nC = UF(P.C), nG = UF(P.G), nMAX = UF(P.NMAX);
  tb.gcnt = S.gcnt; tb.g_reg = S.g_reg; tb.g_pos = S.g_pos; tb.g_active = S.g_active; tb.g_hpos = S.g_hpos; tb.g_hzero = S.g_hzero; tb.ge_vals = P.ge_vals; tb.ge_cnt = P.ge_cnt; tb.ge_stride = tb.T;

  // ---------------- initialise state (global memory); wave 0 alone, it is a one-off ----------------
  if (wv == 0) {
  for (u32 i = lane; i < P.P; i += 64) { const u32 pd = P.queue ? P.queue[i] : i; const u32 gp = P.pod_gid ? P.pod_gid[pd] : pd; tb.q[i] = (u64)pd | ((u64)P.stage_cls[P.pod_stage_off[gp]] << 32); G_lastgen[i] = 0xFFFFFFFFu; G_lastlen[i] = 0; G_pod_stage[i] = 0; tb.pod_node[i] = -1; tb.pod_seq[i] = -1; G_pod_reason[i] = 0; }
  for (u32 e = lane; e < tb.E; e += 64) {
    const Rec r = slot_rec(S, tb, e);
    // (derived what-if) a node that left the cluster keeps its row -- the others keep their place in the visiting order -- but takes nothing:
    // every pod requests `pods`, and no request fits a room of INT64_MIN / 2 (existingnode.go:99-103 is the first resource test); its taints
    // are all set besides, so that a class that tolerated everything would still be refused by the resource test
    const bool gone = P.en_removed && ((P.en_removed[e >> 6] >> (e & 63u)) & 1ull);
    r.taints() = gone ? ~0ull : P.en_taints[e]; r.present() = P.en.present[e]; r.complement() = P.en.complement[e]; r.it_state() = P.en.it_state[e];
    r.reqmask() = gone ? ((1u << tb.R) - 1u) : P.en_requests_present[e]; r.count() = 0;
    for (u32 k = 0; k < tb.K; ++k) { r.mask()[k] = P.en.mask[(size_t)e * tb.K + k]; r.gt()[k] = P.en.gt[(size_t)e * tb.K + k]; r.lt()[k] = P.en.lt[(size_t)e * tb.K + k]; }
    for (u32 rr = 0; rr < tb.R; ++rr) { r.req()[rr] = P.en_requests[(size_t)e * tb.R + rr]; r.room()[rr] = gone ? INT64_MIN / 2 : P.en_avail[(size_t)e * tb.R + rr] - P.en_requests[(size_t)e * tb.R + rr]; r.low()[rr] = INT64_MIN; }
    i32 head = -1; for (u32 i = P.en_port_off[e]; i < P.en_port_off[e + 1]; ++i) { S.pp_entry[i] = P.ports[i]; S.pp_next[i] = head; head = (i32)i; }
    r.porthead() = head;
    for (u32 h = 0; h < tb.GH; ++h) tb.hcnt[(size_t)e * tb.GH + h] = ks_host_count0(P, h, e);
  }
  for (u32 i = lane; i < P.C; i += 64) G_wm[i] = 0;
  for (u32 i = lane; i < tb.E * P.ND; i += 64) S.vol_cnt[i] = P.en_vol_count[i];
  for (u32 i = lane; i < tb.E * P.SW; i += 64) S.vol_set[i] = P.en_vol_set[i];
  for (u32 i = lane; i < P.M * tb.R; i += 64) G_remaining[i] = P.tmpl_remaining[i];
  }

  // ---------------- small hot tables: true LDS arrays in the FAST variant, global memory otherwise ----------------
  u32 lds_used = 0;
  if constexpr (FAST) {
    FastTabs& ft = L.ft;
    u32 (&sm_key_nvalues)[KS_MAX_KEYS] = ft.key_nvalues; i32 (&sm_value_int)[KS_MAX_KEYS * 64] = ft.value_int;
    u8 (&sm_its_fail)[KS_FAST_S * KS_FAST_S] = ft.its_fail; u16 (&sm_its_inter)[KS_FAST_S * KS_FAST_S] = ft.its_inter;
    i32 (&sm_gcnt)[KS_FAST_G * 64] = ft.gcnt; u64 (&sm_g_reg)[KS_FAST_G] = ft.g_reg; u64 (&sm_g_pos)[KS_FAST_G] = ft.g_pos; u8 (&sm_g_active)[KS_FAST_G] = ft.g_active;
    i32 (&sm_g_hpos)[KS_FAST_G] = ft.g_hpos; i32 (&sm_g_hzero)[KS_FAST_G] = ft.g_hzero; u32 (&sm_ge_cnt)[KS_MAX_RES] = ft.ge_cnt;
    i64* ge = (i64*)ks_dyn_lds;
    const u32 gs = UF(P.ge_max);            // the Allocatable ladders are stored with the longest one's stride
    if (wv == 0) {
    for (u32 i = lane; i < tb.K; i += 64) sm_key_nvalues[i] = P.key_nvalues[i];
    for (u32 i = lane; i < tb.K * 64; i += 64) sm_value_int[i] = P.value_int[i];
    // SC == 1: no class / filter constrains the instance-type key, the tables are never read
    if (tb.SC > 1) for (u32 i = lane; i < tb.S * tb.SC; i += 64) { sm_its_fail[i] = P.its_fail[i]; sm_its_inter[i] = P.its_inter[i]; }
    for (u32 i = lane; i < P.G * 64; i += 64) sm_gcnt[i] = P.grp_count[i];
    for (u32 g = lane; g < P.G; g += 64) {
      u64 reg = 0, pos = 0; for (int d = 0; d < 64; ++d) { const i32 c = P.grp_count[(size_t)g * 64 + d]; if (c >= 0) reg |= 1ull << d; if (c > 0) pos |= 1ull << d; }
      if (P.grp_count[(size_t)g * 64] == INT32_MIN) { reg = 0; pos = ~0ull; }      // (ks_derive_topology) the group does not exist in this what-if: eval_node skips it
      sm_g_reg[g] = reg; sm_g_pos[g] = pos; sm_g_active[g] = P.grp_active[g];
    }
    for (u32 h = lane; h < tb.GH; h += 64) { i32 np = P.grph_extra_pos[h], nz = 0; for (u32 e = 0; e < tb.E; ++e) { const i32 c = ks_host_count0(P, h, e); if (c > 0) ++np; if (c == 0) ++nz; } sm_g_hpos[h] = np; sm_g_hzero[h] = nz; }
    for (u32 i = lane; i < tb.R; i += 64) sm_ge_cnt[i] = P.ge_cnt[i];
    for (u32 r = 0; r < tb.R; ++r) for (u32 i = lane; i < gs; i += 64) ge[(size_t)r * gs + i] = i < P.ge_cnt[r] ? P.ge_vals[(size_t)r * tb.T + i] : INT64_MAX;
    }
    lds_used = (u32)(((size_t)tb.R * gs * sizeof(i64) + 15) & ~(size_t)15);
    tb.key_nvalues = sm_key_nvalues; tb.value_int = sm_value_int; tb.its_fail = sm_its_fail; tb.its_inter = sm_its_inter;
    tb.gcnt = sm_gcnt; tb.g_reg = sm_g_reg; tb.g_pos = sm_g_pos; tb.g_active = sm_g_active; tb.g_hpos = sm_g_hpos; tb.g_hzero = sm_g_hzero; tb.ge_cnt = sm_ge_cnt; tb.ge_vals = ge; tb.ge_stride = gs;
  } else if (wv == 0) {
    for (u32 i = lane; i < P.G * 64; i += 64) S.gcnt[i] = P.grp_count[i];
    for (u32 g = lane; g < P.G; g += 64) {
      u64 reg = 0, pos = 0; for (int d = 0; d < 64; ++d) { const i32 c = P.grp_count[(size_t)g * 64 + d]; if (c >= 0) reg |= 1ull << d; if (c > 0) pos |= 1ull << d; }
      if (P.grp_count[(size_t)g * 64] == INT32_MIN) { reg = 0; pos = ~0ull; }
      S.g_reg[g] = reg; S.g_pos[g] = pos; S.g_active[g] = P.grp_active[g];
    }
    for (u32 h = lane; h < tb.GH; h += 64) { i32 np = P.grph_extra_pos[h], nz = 0; for (u32 e = 0; e < tb.E; ++e) { const i32 c = ks_host_count0(P, h, e); if (c > 0) ++np; if (c == 0) ++nz; } S.g_hpos[h] = np; S.g_hzero[h] = nz; }
  }
  if (wv == 0) ls.hslot_of[lane] = ((u32)lane < P.G && P.grp_hslot[lane] >= 0 && P.grp_hslot[lane] < 255) ? (u8)P.grp_hslot[lane] : (u8)0xFF;
  if (wv == 0 && lane < 32) ls.ctr[lane] = 0;
  if (wv == 0 && lane < 8) ls.hard[lane] = 0;
  if (lane == 0) sh.cls.c = 0xFFFFFFFFu;      // no plan staged yet
  __threadfence_block();
  __syncthreads();
  const GA ClsPlan* plans = (const GA ClsPlan*)UF64((u64)P.plans);
  u32* const ord_l = (u32*)(ks_dyn_lds + lds_used);    // ord[pos] = new-node index j, sorted in visiting order (LDS home)
  GA u32* const ord_g = (GA u32*)S.order_g;            // ... its global-memory home once it outgrows LDS
#define ORD_RD(i) (ord_in_lds ? ord_l[(i)] : ord_g[(i)])
#define ORD_WR(i, v) do { if (ord_in_lds) ord_l[(i)] = (v); else ord_g[(i)] = (v); } while (0)
  const u32 ord_cap = (lds_bytes - lds_used) / 4;
  bool ord_in_lds = true;
  // count-bucket boundaries: bstart[c] (1 <= c <= maxc+1) = first position in `ord` whose node has >= c pods
  auto bst_rd = [&](u32 c) -> u32 { u32 v; if (c < KS_BST_LDS) v = ls.bstart[c]; else v = G_bstart[c]; return v; };      // (a pointer chosen between LDS and global memory would be generic: FLAT accesses wait on both counters)
  auto bst_wr = [&](u32 c, u32 v) { if (c < KS_BST_LDS) ls.bstart[c] = v; else G_bstart[c] = v; };

  // The Solve's sequential state lives in wave 0's registers (SGPRs); the other waves of the workgroup only take
  // part in speculation rounds (below) and otherwise wait at the barriers.
  u32 q_head = 0, q_len = nP, q_gen = 0, nnew = 0, seq = 0, err = 0, maxc = 0;
  u32 pp_used = tb.E ? P.en_port_off[tb.E] : 0;
  const bool want_stats = !LEAN && (UF(P.flags) & KS_FLAG_STATS) != 0;
  GA u64* const scratch = tb.n_alive + (size_t)nMAX * tb.TW;      // one spare row of the alive table
  u64 tprobe = __builtin_readcyclecounter();
  u64 qe_a = 0, qe_b = 0; u32x4 pf0 = {0, 0, 0, 0}, pf1 = {0, 0, 0, 0}; bool pf_ok = false;
  // Fit-bitmap reuse across a run of evaluation-equivalent pods (ks_link_plans): every lane keeps its last
  // evaluation (ev, slot, the sh.la_* slots); r_mask = lanes that passed and have not been used, valid for lanes
  // below r_lim (the rest of the winner's count bucket); r_removed = nodes that left the step's window since.
  Ev ev; ev.rc = 0; u32 slot = 0xFFFFFFFFu;
  bool r_valid = false; u32 r_eq = 0, r_base = 0, r_removed = 0, r_lim = 0; u64 r_mask = 0;
  bool done = false; u32 seq_credit = 0, iters = 0;
  u64 pq_e = 0; bool pq_ok = false;      // leader: the next 64 queue entries (lane i: entry i), requested at the end of the previous step

  // ---------------- Solve loop, scheduler.go:104-124 ----------------
  // Planning (wave 0): what the next step of the loop is -- 0 done, 1 one pod sequentially, 2 speculation round.  It runs at
  // the END of a step, so the barrier that ends the step also publishes the plan.
  // A round takes the next queue entries (up to 64, up to the first requeued one: its staleness test needs the sequential state)
  // as long as they belong to at most NW-1 distinct evaluation classes: worker wave j evaluates class j ONCE for all its pods.
  const GA ClsBrief* briefs = (const GA ClsBrief*)UF64((u64)P.briefs);
  u64 b_e = 0, b_tmask = 0, b_tfull = 0, b_rmask = 0, b_zmask = 0, b_rsure = 0; u32 b_flags = 0, b_reqmask = 0, b_w = 0xFFu; i64 b_req[RM];      // leader, lane i: round pod i
#pragma unroll
  for (int i = 0; i < RM; ++i) b_req[i] = 0;
  u32 round_lim = 64;              // after a cancelled round: the next one stops short of the pod whose node failed
  u32 plan_par = 0, stepc = 0;     // stepc: loop iterations started (every wave counts them alike)
  u32 spec_mode = 0;               // what the plan made during a round's filter phase chose (undone if the round is cancelled)
  auto plan = [&](const u32 q_head_, const u32 q_len_, const u32 seq_) {      // (the queue as it will be when the planned step starts)
        const u32 q_head = UF(q_head_), q_len = UF(q_len_), seq = UF(seq_);
        plan_par = UF(plan_par); round_lim = UF(round_lim); seq_credit = UF(seq_credit); nnew = UF(nnew);
        u32 mode = 1; const u32 wpar = plan_par ^ 1u;
        if (++iters > 8u * nP + 4096u) err = (u32)(-KS_ERR_INTERNAL);      // watchdog: a Solve needs at most a few steps per pod
        if (done || err || q_len == 0) mode = 0;
        else if (seq_credit == 0 && q_len >= 2 && round_lim >= 2) {
          const u32 cap = min(min(64u, q_len), round_lim); u64 e = pq_e;
          if (!pq_ok) { u32 idx = q_head + lane; if (idx >= nP) idx -= nP; if (idx >= nP) idx = 0; e = tb.q[idx]; }      // every slot of q always holds a valid class index
          const u32 cls = (u32)(e >> 32) & 0x7FFFFFFFu;
          const GA u32x4* bp = (const GA u32x4*)(briefs + cls);
          const u32x4 v0 = bp[0], v1 = bp[1]; const u32 rqm = *(const GA u32*)((const GA u8*)bp + 32);
          const GA i64* rqp = (const GA i64*)((const GA u8*)bp + 40);
#pragma unroll
          for (int i = 0; i < RM; ++i) b_req[i] = rqp[i];
          u32 dynw_l = 0;
          { const u32 dynw = *(const GA u32*)((const GA u8*)bp + 36); const u32 dms = *(const GA u32*)((const GA u8*)bp + 120), dpd = *(const GA u32*)((const GA u8*)bp + 124);
            rc.dynq[wpar][lane] = (dms & 0xFFFFFFu) | (dpd << 24); dynw_l = dynw; }
          b_e = e; b_tmask = (u64)v0.x | ((u64)v0.y << 32); b_tfull = (u64)v0.z | ((u64)v0.w << 32); b_rmask = (u64)v1.x | ((u64)v1.y << 32); b_flags = (v1.w & 1u) | (dynw_l << 1) /* bit 0: round-eligible; bits 1..: ClsBrief::dyn */; b_reqmask = rqm; b_zmask = *(const GA u64*)((const GA u8*)bp + 104); b_rsure = *(const GA u64*)((const GA u8*)bp + 112);
          const u32 evc = v1.z;
          u32 rn = cap;
          { const u64 rq_bits = ballot64((u32)lane < cap && (e >> 63) != 0); if (rq_bits) rn = min(rn, (u32)__builtin_ctzll(rq_bits)); }
          { const u64 ne = ballot64((u32)lane < rn && !(b_flags & 1u)); if (ne) rn = min(rn, (u32)__builtin_ctzll(ne)); }      // host ports / plans over the kernel's limits: sequential path
          // a class whose last pod had to look past the window (or open a node) will most likely do so again: a round
          // would evaluate it for nothing -- take it sequentially right away
          { const u32 c0 = RL(cls, 0); if ((UF(ls.hard[(c0 >> 5) & 7u]) >> (c0 & 31u)) & 1u) rn = 0; }
          u32 nw = 0; b_w = 0xFFu;
          u64 rem = rn >= 64 ? ~0ull : ((1ull << rn) - 1ull);
          while (rem && nw < (u32)NW - 1u) {
            const int l = __builtin_ctzll(rem); const u32 v = RL(evc, l);
            const u64 same = ballot64(evc == v) & rem;
            if ((same >> lane) & 1ull) b_w = nw;
            { const u32 c_l = RL(cls, l); if (lane == 0) rc.wcls[wpar][nw] = c_l; }      // (the lane read outside the predicate: the emulator's readlane is a wave collective)
            rem &= ~same; ++nw;
          }
          if (rem) rn = (u32)__builtin_ctzll(rem);            // the first pod of one class too many ends the round
          if (rn >= 2) { mode = 2; rc.qe[wpar][lane] = e; rc.pw[wpar][lane] = (u8)b_w; if (lane == 0) { rc.par = wpar; rc.n = rn; rc.nwk = nw; rc.nnew = nnew; rc.seq0 = seq; rc.ord_in_lds = ord_in_lds ? 1u : 0u; } }
        }
        round_lim = 64;
        if (mode == 1 && seq_credit) --seq_credit;
        if (mode == 2) plan_par = wpar;
        if (lane == 0) rc.mode2[stepc & 1u] = mode;
        spec_mode = mode;
  };
  if constexpr (NW > 1) { if (wv == 0) plan(q_head, q_len, seq); __syncthreads(); }
  for (;;) {
    u32 mode = 1;   // 0 done, 1 one pod sequentially (wave 0), 2 speculation round
    if constexpr (NW > 1) { mode = UF(rc.mode2[stepc & 1u]); ++stepc; if (mode == 0) break; }
#ifdef KS_CHECK   /* debug builds: the visiting order must list every new node once, by nondecreasing pod count, inside its bucket */
    if (NW > 1 && wv == 0 && !err) {
          GSYNC();
          u32 bad = 0;
          for (u32 i = lane; i < nnew; i += 64) {
            const u32 j = ORD_RD(i);
            if (j >= nnew) { bad = 1; continue; }
            const u32 cc = slot_rec(S, tb, tb.E + j).count();
            if (cc == 0 || cc > maxc) { bad = 3; continue; }
            if (i + 1 < nnew) { const u32 j2 = ORD_RD(i + 1); if (j2 < nnew && slot_rec(S, tb, tb.E + j2).count() < cc) bad = 2; }
            if (i < bst_rd(cc) || i >= bst_rd(cc + 1)) bad = 4;
          }
          const u64 bb = ballot64(bad != 0);
          if (bb) err = 100u + RL(bad, __builtin_ctzll(bb));
    }
#endif
    if (mode == 1) {
    if (wv == 0) do {
    // The leader's sequential state is wave-uniform by construction; say so (the compiler's uniformity analysis gives up on the Solve loop,
    // and a state it takes for divergent lives in vector registers, every use behind a v_readfirstlane and an exec-masked branch).
    q_head = UF(q_head); q_len = UF(q_len); q_gen = UF(q_gen); nnew = UF(nnew); seq = UF(seq); maxc = UF(maxc); err = UF(err); pp_used = UF(pp_used);
    // Queue.Pop, queue.go:44-58
    if (q_len == 0) { done = true; break; }
    PROBE(20);
#ifdef KS_PROBES
    const u64 t_pod = __builtin_readcyclecounter();
#endif
    // The queue entry and the class plan of this pod were requested one pod ago (qe_a, pf0/pf1), the entry after
    // it two pods ago (qe_b); a Push invalidates the pipeline (the pushed entry may be one of the prefetched slots).
    if (NW > 1 || !pf_ok) {
      qe_a = tb.q[q_head]; qe_b = tb.q[(q_head + 1 == nP) ? 0 : q_head + 1];
      const GA u32x4* src = (const GA u32x4*)(plans + ((u32)(qe_a >> 32) & 0x7FFFFFFFu));
      pf0 = src[lane]; if ((u32)lane + 64 < KS_PLAN_V) pf1 = src[lane + 64];
    }
    const u64 qe = UF64(qe_a);
    const u32 pod = (u32)qe, cidx = (u32)(qe >> 32) & 0x7FFFFFFFu;
    if ((qe >> 63) && UF(G_lastgen[pod]) == q_gen && UF(G_lastlen[pod]) == q_len) { done = true; break; }   // only a requeued, unrelaxed pod can be stale
    q_head = (q_head + 1 == nP) ? 0 : q_head + 1; q_len--; CTR(KS_STAT_POPS, 1);
    PROBE(12);
    { u32x4* dst = (u32x4*)&sh.cls; dst[lane] = pf0; if ((u32)lane + 64 < KS_PLAN_V) dst[lane + 64] = pf1; }
    stage_class(tb, sh, lane);
    const ClsPlan& c = sh.cls;
    if (UF(c.overflow)) { err = (u32)(-KS_ERR_UNSUPPORTED); break; }
    ClsR cr; cr.tol = UF64(c.tol); cr.reqmask = UF(c.reqmask); cr.ntouch = UF(c.ntouch); cr.nhost = UF(c.nhost); cr.hn_mode = UF(c.hn_mode); cr.port_cnt = UF(c.port_cnt); cr.vol_cnt = UF(c.vol_cnt); cr.it_state = (i32)UF(c.it_state); cr.tkeys = UF64(c.tkeys); cr.eq = UF(c.eq);
#pragma unroll
    for (int i = 0; i < RM; ++i) cr.req[i] = (i64)UF64(c.req[i]);
    if constexpr (LEAN) { cr.port_cnt = 0; cr.vol_cnt = 0; cr.hn_mode = 0; cr.it_state = 0; }
    bool placed = false;
    PROBE(13);

    // Candidates in the reference's visiting order (scheduler.go:174-217): existing nodes in the caller's
    // order, then open new nodes in `sort.Slice(newNodes, len(Pods))` order -- 64 per step, one per lane --
    // then one fresh node per machine template (a single-lane "chunk").  One code path evaluates, filters
    // and commits all three kinds.
    GSYNC();                       // the previous pod's record / counter stores are complete before they are re-read
    if constexpr (NW == 1) {   // (after the barrier: its vmcnt(0) would otherwise wait for these loads) request the next pod's plan and the queue entry after it (every slot of q always holds a valid class index)
      qe_a = qe_b; qe_b = tb.q[(q_head + 1 == nP) ? 0 : q_head + 1];
      const GA u32x4* src = (const GA u32x4*)(plans + ((u32)(qe_a >> 32) & 0x7FFFFFFFu));
      pf0 = src[lane]; if ((u32)lane + 64 < KS_PLAN_V) pf1 = src[lane + 64];
      pf_ok = true;
    }
    u32 pos_base = 0, tm = 0, width = KS_FIRST_WIDTH, why = 0;       // why: one KS_WHY_* per template this pod could not use (scheduler.go:193-217)
    // Watermark: the existing nodes come first in the visiting order and never move; for a `mono` class every node that refused it
    // stays refused, so the scan starts where the last pod of the class (or of an evaluation-equivalent one) had to start looking.
    const bool mono = tb.E != 0 && UF(c.mono) != 0 && !want_stats;
    const u32 wmi = cr.eq ? cr.eq - 1u : cidx;
    u32 scan_lo = 0;
    if (mono) { scan_lo = UF(G_wm[wmi]); pos_base = scan_lo; }
    const u32 wm0 = scan_lo;
    bool reuse = NW == 1 && r_valid && !want_stats && cr.eq != 0 && cr.eq == r_eq;     // (single-wave kernel only: in the multi-wave one rounds take the runs of equivalent pods)
    if (!want_stats && cr.nhost) {
      // anti-affinity (count == 0) and spread with maxSkew - self == 0 accept a node only if its own hostname counts 0; with no such
      // hostname registered every candidate would reject the pod: go straight to the templates
      bool dead = false;
      for (u32 i = 0; i < cr.nhost; ++i) {
        const PlanTopo& th = c.host[i]; const u32 f = UF(*(const u32*)&th.type); const u32 ty = f & 0xFF, self = (f >> 8) & 0xFF;
        if ((ty == 2 || (ty == 0 && (i64)UF(th.maxskew) - (i64)self <= 0)) && (i32)UF(sh.host_zero[i]) <= 0) dead = true;
      }
      if (dead) { pos_base = tb.E + nnew; reuse = false; r_valid = false; CUT(26); }
    }
    if (NW == 1 && cr.eq != 0) CTR(8, 1);
    r_valid = reuse;               // any other path re-evaluates (or moves nodes in ways the window does not track)
    while (!placed && !err) {
      const u32 total = tb.E + nnew;
      const bool fresh = !reuse && pos_base >= total;
      u32 m_t = 0, lim = 0xFFFFFFFFu, ltypes = 0; size_t mc = 0;
      u64 m = 0, reach = 0;
      // Scan ahead: while wave 0 evaluates this window, waves 1..NW-1 evaluate the NW-1 windows after it for the same pod;
      // if this window has no candidate the leader jumps straight to the first window that has one.
      const bool scanning = NW > 1 && !reuse && !fresh && pos_base + 64 < total;
      if constexpr (NW > 1) if (scanning) {
        if (lane == 0) { rc.scan_base = pos_base + 64; rc.scan_total = total; rc.scan_cidx = cidx; rc.ord_in_lds = ord_in_lds ? 1u : 0u; rc.cmd = 1; }
        __syncthreads();
      }
      if (reuse) {
        // the kept evaluations are still exact: no node of the window changed except the ones taken out of it
        pos_base = r_base - r_removed;
        m = r_mask & (r_lim >= 64 ? ~0ull : ((1ull << r_lim) - 1ull));
      } else {
      slot = 0xFFFFFFFFu;
      if (!fresh) {
        const u32 pos = pos_base + lane;
        if ((u32)lane < width && pos < total) slot = pos < tb.E ? pos : tb.E + ORD_RD(pos - tb.E);
      } else {
        // ---- a new node from the next template that survives the pre-checks (scheduler.go:193-213) ----
        PROBE(20);
        bool have = false;
        for (; tm < nM && !have; ++tm) {
          m_t = tm; mc = (size_t)m_t * nC + cidx; lim = UF(P.tmpl_limit_present[m_t]);
          if constexpr (LEAN) lim = 0xFFFFFFFFu;
          if (nnew >= nMAX) { err = (u32)(-KS_ERR_CAPACITY); break; }
          // filterByRemainingResources, scheduler.go:293-309 (only when the provisioner has limits)
          bool lany = false; ltypes = 0;
          for (u32 wbase = 0; wbase < tb.TW; wbase += 64) {
            const u32 w = wbase + lane; u64 a = 0;
            if (w < tb.TW) a = GC(u64, P.tmpl_types)[(size_t)m_t * tb.TW + w];
            if (lim != 0xFFFFFFFFu) {
              for (u64 nz = ballot64(a != 0); nz; nz &= nz - 1) {
                const int b = __builtin_ctzll(nz); const u64 aw = __shfl(a, b); const u32 t = (wbase + b) * 64 + lane;
                bool ok = (aw >> lane) & 1ull;
                if (ok) for (u32 bits = lim; bits; bits &= bits - 1) { const int r = __builtin_ctz(bits); if (P.it_cap[(size_t)r * tb.T + t] > G_remaining[(size_t)m_t * tb.R + r]) { ok = false; break; } }
                const u64 bw = ballot64(ok); if (lane == b) a = bw;
              }
            }
            if (ballot64(a != 0)) lany = true;
            if (want_stats) { u32 pc = __builtin_popcountll(a); for (int off = 32; off > 0; off >>= 1) pc += __shfl_xor(pc, off); ltypes += pc; }
            if (w < tb.TW) scratch[w] = a & G_grid[mc * tb.TW + w];
          }
          if (!lany) { if (m_t < 8) why |= (u32)KS_WHY_LIMITS << (4 * m_t); continue; }                  // "all available instance types exceed provisioner limits" (before NewNode)
          if (want_stats) CTR(KS_STAT_REF_ATTEMPTS, 1);    // NewNode + node.Add is attempted for this template
          if (!UF(GC(u8, P.mc_ok)[mc])) { if (m_t < 8) why |= (u32)UF(GC(u8, P.mc_why)[mc]) << (4 * m_t); continue; }           // taints / Compatible fail inside Add
          have = true;
        }
        if (err) break;
        if (!have) { PROBE(19); break; }     // every template failed -> relax / requeue
        // NewNode (node.go:44-60): materialise the fresh node's record from template∩class, register its hostname
        const u32 fs = tb.E + nnew; const Rec fr = slot_rec(S, tb, fs);
        if ((u32)lane < tb.K) { fr.mask()[lane] = GC(u64, P.mc_mask)[mc * tb.K + lane]; fr.gt()[lane] = GC(i32, P.mc_gt)[mc * tb.K + lane]; fr.lt()[lane] = GC(i32, P.mc_lt)[mc * tb.K + lane]; }
        if (lane >= 32 && (u32)lane < 32 + tb.R) { fr.req()[lane - 32] = GC(i64, P.tmpl_daemon)[(size_t)m_t * tb.R + lane - 32]; fr.room()[lane - 32] = INT64_MAX / 2; fr.low()[lane - 32] = INT64_MIN; }
        if (lane == 63) { fr.taints() = GC(u64, P.tmpl_taints)[m_t]; fr.present() = GC(u32, P.mc_present)[mc]; fr.complement() = GC(u32, P.mc_complement)[mc]; fr.it_state() = GC(i32, P.mc_it)[mc]; fr.reqmask() = GC(u32, P.tmpl_daemon_present)[m_t]; fr.porthead() = -1; fr.count() = 0; }
        for (u32 g = lane; g < nG; g += 64) { const i32 hs = GC(i32, P.grp_hslot)[g]; if (hs >= 0) tb.hcnt[(size_t)fs * tb.GH + hs] = tb.g_active[g] ? 0 : -1; }   // Topology.Register(hostname), node.go:47
        __threadfence_block();
        GSYNC();
        if (lane == 0) slot = fs;
        PROBE(19);
      }

      // ---- Node.Add / ExistingNode.Add up to the instance-type filter, one node per lane ----
      ev.rc = 0;
      if constexpr (NW > 1) {      // nothing of the last evaluation is needed: do not carry it in registers
        ev.count = 0; ev.reqmask = 0; ev.tchg = 0; ev.tnar = 0; ev.tpres = 0; ev.tcomp = 0; ev.present = 0; ev.complement = 0; ev.it_state = 0; ev.it0 = 0;
#pragma unroll
        for (int i = 0; i < RM; ++i) { ev.room[i] = 0; ev.req[i] = 0; ev.low[i] = INT64_MIN; }
      }
      if (slot != 0xFFFFFFFFu) eval_node<BOUNDS, LEAN, RM>(P, S, tb, sh, wb, slot, slot < tb.E, fresh, ev, lane, tprobe, cr);
      PROBE(26);
      m = ballot64(ev.rc == 2);
      reach = ballot64(ev.rc >= 1);
      if (fresh && m == 0 && m_t < 8) { const i32 rc0 = (i32)RL((u32)ev.rc, 0); why |= (u32)(rc0 < 0 ? -rc0 : (rc0 == 1 ? KS_WHY_NO_INSTANCE_TYPE : KS_WHY_REQUIREMENTS)) << (4 * m_t); }
      }
      if constexpr (NW > 1) if (scanning) {
        __syncthreads();
        if (m == 0) {
          const u32 ns = min((u32)NW - 1u, (total - (pos_base + 64) + 63) / 64);       // windows the helpers covered
          const u64 hm = (u32)lane < ns ? rc.scanm[lane & (KS_MAX_WAVES - 1)] : 0ull;
          const u64 hb = ballot64(hm != 0);
          pos_base += 64 * (hb ? (u32)__builtin_ctzll(hb) : ns);                          // the loop's own `+= width` completes the jump
        }
      }
      if (!fresh) { PROBE(14); CTR(21, 1); }
      u32 my_alive = 0; u32 visited = fresh ? 0 : min(width, total - pos_base);   // lanes the reference would have visited (all, unless one succeeds)
      if (want_stats && !fresh && slot != 0xFFFFFFFFu && slot >= tb.E && ev.rc >= 1) for (u32 w = 0; w < tb.TW; ++w) my_alive += __builtin_popcountll(tb.n_alive[(size_t)(slot - tb.E) * tb.TW + w]);

      while (m) {
        const int win = __builtin_ctzll(m);
        Pub pb; publish_eval<BOUNDS, RM>(tb, sh, wb, ev, slot, fresh, lane, win, cr, pb);
        const u32 sw = pb.slot; const bool ex = sw < tb.E; const u32 jw = sw - tb.E;
        const Rec r = slot_rec(S, tb, sw);
        const u32 rm = pb.rm;
        PROBE(15);
        if (!ex) {
          // filterInstanceTypesByRequirements (node.go:94-98): existing nodes have no instance-type step
          CTR(KS_STAT_FULLCHECKS, 1); if (want_stats && fresh) CTR(KS_STAT_REF_TYPES, ltypes);
          GA u64* const alive = tb.n_alive + (size_t)jw * tb.TW;
          const u32 keys = fresh ? pb.narrowed : pb.changed;   // a fresh node's own keys are already in the grid row
          const bool zc = (tb.key_zone >= 0 && ((keys >> tb.key_zone) & 1u)) || (tb.key_ct >= 0 && ((keys >> tb.key_ct) & 1u));
          const bool itc = !fresh && pb.it_state != pb.it_before;
          if (pb.need) {     // otherwise the filter would pick the same rows as last time: InstanceTypeOptions unchanged
            const bool inreg = !fresh && tb.TW <= 128;  // the surviving-type words stay in registers: no scratch round trip
            u64 aw[2];
            const bool ok = filter_types(P, tb, pb, sh, r, fresh ? scratch : alive, fresh ? alive : (inreg ? (GA u64*)nullptr : scratch), rm, keys, zc, itc, lane, tprobe, aw);
            PROBE(16);
            if (!ok) { CTR(KS_STAT_FULLFAILS, 1); if (!fresh) recompute_cap<RM>(P, tb, alive, r, lane); else if (m_t < 8) why |= (u32)KS_WHY_NO_INSTANCE_TYPE << (4 * m_t); m &= m - 1; continue; }
            if (inreg) { if ((u32)lane < tb.TW) alive[lane] = aw[0]; if ((u32)lane + 64 < tb.TW) alive[lane + 64] = aw[1]; }
            else if (!fresh) for (u32 w = lane; w < tb.TW; w += 64) alive[w] = scratch[w];
            if ((u32)lane < tb.R && ((rm >> lane) & 1u)) r.low()[lane] = sh.low_new[lane];
            if constexpr (NW > 1) if (lane < 2 && 4 * lane < RM) { u64 li = 0; for (int i = 0; i < 4; ++i) li |= (u64)((u32)(4 * lane + i) < tb.R ? (sh.low_idx[4 * lane + i] & 0xFFFFu) : 0xFFFFu) << (16 * i); ((GA u64*)S.lowi)[2 * (size_t)sw + lane] = li; }
          }
        }
        // ---- commit: node.go:100-105 / existingnode.go:122-129 / scheduler.go:214-216 ----
        visited = win + 1;
        if (mono && !reuse) scan_lo = ex ? pos_base + (u32)win : tb.E;
        // Run commit (existing node): the queue entries that follow and are evaluation-equivalent to this pod (same ClsBrief::ev, nothing to
        // record, never requeued) would each rescan the same refusing nodes and land here while the node has room -- resources.Fits on the
        // accumulated requests (existingnode.go:99-103) is the only thing that changes between them.  Lane k tests "k+1 more fit".
        u32 t_extra = 0; u64 run_e = 0;
        if (ex && mono && cr.eq != 0 && UF(c.nrec) == 0 && q_len != 0) {
          const u32 avail = min(q_len, 63u), my_ev = UF(briefs[cidx].ev);
          bool okl = false;
          if ((u32)lane < avail) {
            u32 idx = q_head + (u32)lane; if (idx >= nP) idx -= nP;
            run_e = tb.q[idx];
            const GA ClsBrief* bq = briefs + ((u32)(run_e >> 32) & 0x7FFFFFFFu);
            okl = (run_e >> 63) == 0 && bq->ev == my_ev && bq->rmask == 0;
          }
          const u64 nok = ~ballot64(okl); const u32 run = nok ? (u32)__builtin_ctzll(nok) : 64u;
          if (run) {
            bool fitl = true;
#pragma unroll
            for (int i = 0; i < RM; ++i) if (cr.req[i] > 0 && (i64)(lane + 1) * cr.req[i] > pb.room_new[i]) fitl = false;
            const u64 nfit = ~ballot64(fitl); t_extra = min(run, nfit ? (u32)__builtin_ctzll(nfit) : 64u);
#pragma unroll
            for (int i = 0; i < RM; ++i) { pb.req_new[i] += (i64)t_extra * cr.req[i]; pb.room_new[i] -= (i64)t_extra * cr.req[i]; }
          }
        }
        if (fresh && lim != 0xFFFFFFFFu) {          // subtractMax, scheduler.go:273-290
          GA u64* const alive = tb.n_alive + (size_t)jw * tb.TW;
          i64 mx[RM];
#pragma unroll
          for (int rr = 0; rr < RM; ++rr) mx[rr] = INT64_MIN;
          for (u32 t = lane; t < tb.TW * 64; t += 64) {
            const bool on = t < tb.T && ((alive[t >> 6] >> (t & 63)) & 1ull);
#pragma unroll
            for (int rr = 0; rr < RM; ++rr) if ((u32)rr < tb.R && on) { const i64 cp = P.it_cap[(size_t)rr * tb.T + t]; if (cp > mx[rr]) mx[rr] = cp; }
          }
#pragma unroll
          for (int rr = 0; rr < RM; ++rr) if ((u32)rr < tb.R) { const i64 v = wave_max_i64(mx[rr]); if (lane == 0 && ((lim >> rr) & 1u)) G_remaining[(size_t)m_t * tb.R + rr] -= v; }
        }
        if (fresh) {        // the node exists from here on: its registered hostnames join the zero-count census (before this pod is recorded)
          for (u32 g = lane; g < nG; g += 64) { const i32 hs = GC(i32, P.grp_hslot)[g]; if (hs >= 0 && tb.g_active[g]) tb.g_hzero[hs]++; }      // (its counters were set a moment ago: 0 for the groups that exist, see Topology.Register above)
          LSYNC();
        }
        topology_record<false>(P, S, tb, pb, sh, r, sw, lane);      // (the sequential path commits alone: nothing else records meanwhile)
        const u32 cnt = pb.count;                                   // pods on the node before this one
        LSYNC();
        write_record<BOUNDS, RM>(tb, r, pb, sh, rm, lane);
        if (lane == 0) {
          if (!ex) r.count() = cnt + 1;
          if (fresh) G_n_tmpl[jw] = (i32)m_t;
          for (u32 i = 0; i < cr.port_cnt; ++i) { S.pp_entry[pp_used + i] = P.ports[c.port_off + i]; S.pp_next[pp_used + i] = r.porthead(); r.porthead() = (i32)(pp_used + i); }
          if (!LEAN && ex && cr.vol_cnt) volumes_walk<true>(P, S, c, sw);
          tb.pod_node[pod] = (i32)sw; tb.pod_seq[pod] = (i32)seq; G_pod_reason[pod] = 0;
        }
        if (t_extra) {
          if ((u32)lane < t_extra) { const u32 pd = (u32)run_e; tb.pod_node[pd] = (i32)sw; tb.pod_seq[pd] = (i32)(seq + 1u + (u32)lane); G_pod_reason[pd] = 0; }
          q_head += t_extra; if (q_head >= nP) q_head -= nP;
          q_len -= t_extra; seq += t_extra; pf_ok = false;
        }
        PROBE(17);
        if (!ex && !fresh) {
          // visiting order: the node leaves position p of bucket `cnt` for the FRONT of bucket cnt+1
          const u32 p = pos_base + win - tb.E;
          const u32 endc = UF(bst_rd(cnt + 1));                         // one past the last node with `cnt` pods
          for (u32 i = p + 1; i < endc; i += 256) {       // shift left by one: reads may run ahead of the writes (four chunks in flight)
            u32 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { const u32 ii = i + 64u * (u32)u + (u32)lane; v[u] = 0; if (ii < endc) v[u] = ORD_RD(ii); }
            if (ord_in_lds) LSYNC(); else GSYNC();
#pragma unroll
            for (int u = 0; u < 4; ++u) { const u32 ii = i + 64u * (u32)u + (u32)lane; if (ii < endc) ORD_WR(ii - 1, v[u]); }
          }
          if (ord_in_lds) LSYNC(); else GSYNC();
          if (lane == 0) { ORD_WR(endc - 1, jw); bst_wr(cnt + 1, endc - 1); if (cnt + 1 > maxc) bst_wr(cnt + 2, nnew); }
          if (cnt + 1 > maxc) maxc = cnt + 1;
          // keep the step's remaining fit bits for the next pod if it is evaluation-equivalent: valid for the lanes
          // whose nodes share the winner's count bucket (they now precede it in the visiting order)
          if (reuse) { r_mask = m & (m - 1); ++r_removed; if (NW == 1) CTR(11, 1); }
          else if (NW == 1 && cr.eq != 0 && !want_stats) {
            CTR(10, 1);
            const u32 lim_abs = tb.E + endc;                          // one past the bucket, in this step's coordinates
            r_valid = true; r_eq = cr.eq; r_mask = m & (m - 1); r_base = pos_base; r_removed = 1; r_lim = lim_abs > pos_base ? min(64u, lim_abs - pos_base) : 0;
          }
        } else if (fresh) {
          // visiting order: appended -> BACK of the count-1 bucket, i.e. position bstart[2]; everything after shifts right
          if (ord_in_lds && nnew + 1 > ord_cap) {                     // spill the order array to global memory
            for (u32 i = lane; i < nnew; i += 64) ord_g[i] = ord_l[i];
            GSYNC(); ord_in_lds = false;
          }
          if (maxc == 0) { if (lane == 0) { bst_wr(1, 0); bst_wr(2, 1); ORD_WR(0, jw); } maxc = 1; }
          else {
            const u32 ins = UF(bst_rd(2));
            for (u32 hi = nnew; hi > ins; ) {            // shift right by one, from the top down: four chunks in flight (the reads run ahead towards lower positions, the writes go up)
              const u32 lo = hi > ins + 256 ? hi - 256 : ins; u32 v[4];
#pragma unroll
              for (int u = 0; u < 4; ++u) { const u32 ii = lo + 64u * (u32)u + (u32)lane; v[u] = 0; if (ii < hi) v[u] = ORD_RD(ii); }
              if (ord_in_lds) LSYNC(); else GSYNC();
#pragma unroll
              for (int u = 0; u < 4; ++u) { const u32 ii = lo + 64u * (u32)u + (u32)lane; if (ii < hi) ORD_WR(ii + 1, v[u]); }
              if (ord_in_lds) LSYNC(); else GSYNC();
              hi = lo;
            }
            if (lane == 0) ORD_WR(ins, jw);
            for (u32 cc = 2 + lane; cc <= maxc + 1; cc += 64) bst_wr(cc, bst_rd(cc) + 1u);
          }
          nnew = jw + 1;
        }
        pp_used += cr.port_cnt; ++seq; placed = true;
        if (NW > 1) { if (fresh) CUT(29); else if (pos_base == wm0) CUT(12); else CUT(20); }      // (experiment builds) sequential pods: opened a node / first window / deeper
        if constexpr (NW > 1) { if (lane == 0) { const u32 hb = 1u << (cidx & 31u); if (fresh || (!reuse && pos_base != 0)) ls.hard[(cidx >> 5) & 7u] |= hb; else ls.hard[(cidx >> 5) & 7u] &= ~hb; } }
        LSYNC();
        PROBE(18);
        break;
      }
      if (reuse && !placed) { if (NW == 1) CTR(9, 1); reuse = false; r_valid = false; pos_base = mono ? scan_lo : 0u; continue; }   // window exhausted: evaluate from the top
      if (want_stats && !fresh) {
        CTR(KS_STAT_REF_ATTEMPTS, visited);
        u32 ty = ((u32)lane < visited && ((reach >> lane) & 1ull)) ? my_alive : 0;
        for (int off = 32; off > 0; off >>= 1) ty += __shfl_xor(ty, off);
        CTR(KS_STAT_REF_TYPES, ty);
      }
      if (!fresh) { pos_base += width; width = 64; if (mono && !reuse && !placed) scan_lo = min(pos_base, tb.E); }
    }
    if (err) break;
    if (mono && scan_lo > wm0 && lane == 0) G_wm[wmi] = scan_lo;
#ifdef KS_PROBES
    if (NW == 1) { const u32 kind = cr.nhost ? 2 : (c.ntopo ? 1 : 0); CTR(27 + kind, __builtin_readcyclecounter() - t_pod); if (kind) CTR(29 + kind, 1); }
    else { CTR(28, __builtin_readcyclecounter() - t_pod); CTR(30, 1); }
#endif

    // ---- failure: Preferences.Relax + Queue.Push + Topology.Update (scheduler.go:116-123) ----
    if (!placed) {
      CUT(31);
      const u32 gpod = P.pod_gid ? UF(GC(u32, P.pod_gid)[pod]) : pod;      // (derived what-if: the chain is the snapshot pod's)
      const u32 nst = UF(G_pod_stage_off[gpod + 1] - G_pod_stage_off[gpod]);
      const i32 stg = (i32)UF(G_pod_stage[pod]);
      const bool relaxed = (u32)stg + 1 < nst;
      u32 tail = q_head + q_len; if (tail >= nP) tail -= nP;
      q_len++; pf_ok = false;
      LSYNC();      // (every lane has read the pod's stage before lane 0 moves it on: in lockstep that is the order anyway; the lane-fibre emulator runs lane 0 ahead to the next barrier and models readfirstlane as the lane's own read)
      if (lane == 0) {
        G_pod_reason[pod] = why;
        const u32 ncls = relaxed ? G_stage_cls[G_pod_stage_off[gpod] + stg + 1] : cidx;
        tb.q[tail] = (u64)pod | ((u64)ncls << 32) | (relaxed ? 0ull : (1ull << 63));
        if (relaxed) {
          G_pod_stage[pod] = stg + 1;
          const u32 nc = ncls;
          for (u32 i = P.cls_own_off[nc]; i < P.cls_own_off[nc + 1]; ++i) tb.g_active[P.own_list[i] & 0x7FFFFFFFu] = 1;   // Topology.Update creates the group
        } else { G_lastlen[pod] = q_len; G_lastgen[pod] = q_gen; }
      }
      if (relaxed) { q_gen++; CTR(KS_STAT_RELAX, 1); }
      __threadfence_block();
      GSYNC();
    }
    } while (0);
    if constexpr (NW == 1) { if (done || err) break; }
    else {
      if (wv == 0) {
        if (lane == 0) rc.cmd = 0;
        pq_ok = false; plan(q_head, q_len, seq);
        __syncthreads();                                                   // release the scan-ahead helpers; publishes the plan
      }
      else {
        bool staged = false; ClsR cr;
        for (;;) {
          __syncthreads();
          if (UF(rc.cmd) == 0) break;
          const u32 base = UF(rc.scan_base) + 64u * (wv - 1), total = UF(rc.scan_total);
          u64 m = 0;
          if (base < total) {
            if (!staged) {      // once per pod: nothing the evaluation reads changes while the leader looks for a node
              const u32 cidx = UF(rc.scan_cidx);
              { const GA u32x4* src = (const GA u32x4*)(plans + cidx); u32x4* dst = (u32x4*)&sh.cls; dst[lane] = src[lane]; if ((u32)lane + 64 < KS_PLAN_V) dst[lane + 64] = src[lane + 64]; }
              stage_class(tb, sh, lane);
              const ClsPlan& c = sh.cls;
              cr.tol = UF64(c.tol); cr.reqmask = UF(c.reqmask); cr.ntouch = UF(c.ntouch); cr.nhost = UF(c.nhost); cr.hn_mode = UF(c.hn_mode); cr.port_cnt = UF(c.port_cnt); cr.vol_cnt = UF(c.vol_cnt); cr.it_state = (i32)UF(c.it_state); cr.tkeys = UF64(c.tkeys); cr.eq = UF(c.eq);
#pragma unroll
              for (int i = 0; i < RM; ++i) cr.req[i] = (i64)UF64(c.req[i]);
              if constexpr (LEAN) { cr.port_cnt = 0; cr.vol_cnt = 0; cr.hn_mode = 0; cr.it_state = 0; }
              staged = true;
            }
            const bool ol = UF(rc.ord_in_lds) != 0;
            const u32 pos = base + lane;
            slot = 0xFFFFFFFFu;
            if (pos < total) slot = pos < tb.E ? pos : tb.E + (ol ? ord_l[pos - tb.E] : ord_g[pos - tb.E]);
            ev.rc = 0;
            if (slot != 0xFFFFFFFFu) eval_node<BOUNDS, LEAN, RM>(P, S, tb, sh, wb, slot, slot < tb.E, false, ev, lane, tprobe, cr);
            m = ballot64(ev.rc == 2);
          }
          if (lane == 0) rc.scanm[wv - 1] = m;
          __syncthreads();
        }
      }
    }
    continue;
    }

    // =====================================================================================================
    // Speculation round.  The round's pods belong to at most NW-1 evaluation classes; worker wave j evaluates class j ONCE
    // against a snapshot of the first 64 candidates in visiting order (lane = candidate).  The leader then walks the pods in
    // queue order and gives each the node the sequential algorithm would give it, from the fit bitmaps alone:
    //   * it keeps the visiting order of the window as the round changes it -- a node that takes a pod goes to the FRONT of
    //     its next count bucket (existing nodes keep their place) -- and pod k takes the first candidate in THAT order which
    //     accepts it.  A candidate the round has not touched accepts iff it did at the snapshot.  A candidate that already
    //     took pods of this round accepts iff it did at the snapshot AND the resource screen still passes with what the round
    //     put on it, provided none of those pods changed its requirements and none recorded into a topology group pod k's
    //     evaluation reads (otherwise the answer is unknown: the round ends before pod k).  Nodes that rejected stay rejecting
    //     (a commit only narrows a node); a moved node can only be trusted to be next if the window still covers its place.
    //   * (a) the round ends before a pod whose evaluation reads a topology counter of ANOTHER node that an earlier pod of the
    //     round records into (ClsBrief.tmask / rmask); hostname-keyed spread / anti-affinity records only touch the winner.
    // Commit is per NODE: the wave that evaluated the last pod placed on a node (lane = that candidate) adds up what the round
    // put there, runs the instance-type filter once with the totals (filters are monotone in the requests, so every intermediate
    // state is non-empty if the final one is) and writes the record; Topology.Record and the pod results are per pod.  A filter
    // that comes back empty cancels the round (nothing was written but InstanceTypeOptions rows, which are restored); the next
    // round stops short of the offending pod.
    // =====================================================================================================
    if constexpr (NW > 1) {
      const u32 rn = UF(rc.n), par = UF(rc.par), seq0 = UF(rc.seq0), nwk = UF(rc.nwk);
#ifdef KS_PROBES
      const u64 t_round = __builtin_readcyclecounter();
#endif
      const bool ord_lds_r = UF(rc.ord_in_lds) != 0;
      const u32 kw = wv - 1;                 // this wave's evaluation class within the round (wave 0: none)
      const u32 total = tb.E + UF(rc.nnew), nwin = min(64u, total);
      ClsR cr;
      // ---- P1: one evaluation per class ----
      if (wv != 0 && kw < nwk) {
        const u32 cidx = UF(rc.wcls[par][kw]);
        if (UF(sh.cls.c) != cidx) { const GA u32x4* src = (const GA u32x4*)(plans + cidx); u32x4* dst = (u32x4*)&sh.cls; dst[lane] = src[lane]; if ((u32)lane + 64 < KS_PLAN_V) dst[lane + 64] = src[lane + 64]; }      // (usually staged at the end of the last round, below)
        stage_class(tb, sh, lane);
        const ClsPlan& c = sh.cls;
        cr.tol = UF64(c.tol); cr.reqmask = UF(c.reqmask); cr.ntouch = UF(c.ntouch); cr.nhost = UF(c.nhost); cr.hn_mode = UF(c.hn_mode); cr.port_cnt = UF(c.port_cnt); cr.vol_cnt = UF(c.vol_cnt); cr.it_state = (i32)UF(c.it_state); cr.tkeys = UF64(c.tkeys); cr.eq = UF(c.eq);
#pragma unroll
        for (int i = 0; i < RM; ++i) cr.req[i] = (i64)UF64(c.req[i]);
        if constexpr (LEAN) { cr.port_cnt = 0; cr.vol_cnt = 0; cr.hn_mode = 0; cr.it_state = 0; }
        slot = 0xFFFFFFFFu;
        if ((u32)lane < nwin) slot = (u32)lane < tb.E ? (u32)lane : tb.E + (ord_lds_r ? ord_l[lane - tb.E] : ord_g[lane - tb.E]);
        ev.rc = 0; ev.count = 0; ev.reqmask = 0; ev.tchg = 0; ev.tpres = 0; ev.tcomp = 0; ev.present = 0; ev.complement = 0; ev.it_state = 0; ev.it0 = 0;
#pragma unroll
        for (int i = 0; i < RM; ++i) { ev.room[i] = 0; ev.req[i] = 0; ev.low[i] = INT64_MIN; }
        const bool dync = (UF(c.dyn) & 1u) != 0;
        // (what this wave publishes besides the evaluation is requested BEFORE it and used after: the loads travel with the evaluation's own)
        u32 zl = 0xFFu;        // (worker 0) this candidate's single value on dyn_key, if its requirement is In [v]
        const bool want_zl = kw == 0 && (i32)UF(P.dyn_key) >= 0 && slot != 0xFFFFFFFFu; const u32 dk = want_zl ? UF((u32)P.dyn_key) : 0u;
        u64 zm_pre = 0; if (want_zl) zm_pre = slot_rec(S, tb, slot).mask()[dk];
        const u32 dyn2 = UF(c.dyn) & 2u; u32 h0f = 0, h0slot = 0; i32 h0max = 0, hc_pre = 0;
        if (dyn2) { const PlanTopo& th = c.host[0]; h0f = UF(*(const u32*)&th.type); h0slot = UF(th.hslot); h0max = (i32)UF(th.maxskew); if (slot != 0xFFFFFFFFu) hc_pre = tb.hcnt[(size_t)slot * tb.GH + h0slot]; }
        if (slot != 0xFFFFFFFFu) eval_node<BOUNDS, LEAN, RM>(P, S, tb, sh, wb, slot, slot < tb.E, false, ev, lane, tprobe, cr, dync);
        if (want_zl && ((ev.present >> dk) & 1u) && !((ev.complement >> dk) & 1u) && __builtin_popcountll(zm_pre) == 1) zl = (u32)__builtin_ctzll(zm_pre);
        const u64 mo = ballot64((ev.rc & 3) == 2 && ev.rc > 0);
        const u64 m = ballot64(ev.rc == 2);
        const u64 chgb = ballot64((ev.rc & 3) == 2 && ev.rc > 0 && (ev.tchg != 0 || ev.it_state != ev.it0));
        if (lane == 0) { rc.m[kw] = m; rc.chg[kw] = chgb; rc.mo[kw] = mo; }
        if (kw == 0) rc.zone[lane] = (u8)zl;
        if (dyn2) {      // how many more pods of the item's group this candidate takes
          const u32 ty = h0f & 0xFF, self = (h0f >> 8) & 0xFF;
          i32 slack = 0;
          if (slot != 0xFFFFFFFFu && ty == 0) slack = h0max - (i32)self - hc_pre;
          rc.hslack[kw][lane] = (u8)(slack < 0 ? 0 : (slack > 255 ? 255 : slack));
          const u64 hz = ballot64(slot != 0xFFFFFFFFu && hc_pre == 0); if (lane == 0) rc.hz0[kw] = hz;
        }
        if (kw == 0) {
          rc.cnt[lane] = ev.count; rc.rmsk[lane] = ev.reqmask;
#pragma unroll
          for (int i = 0; i < RM; ++i) { rc.room[i][lane] = ev.room[i]; rc.req0[i][lane] = ev.req[i]; rc.low0[i][lane] = ev.low[i]; }
        }
      }
      if (wv == 0 && lane == 0) rc.fail_at = 0xFFFFFFFFu;
      __syncthreads();
#ifdef KS_PROBES
      u64 t_ph = __builtin_readcyclecounter(); CTR(22, t_ph - t_round);
#endif
      // ---- P2: the leader resolves the round.  Lane i plays two parts: round pod i (b_*) and window candidate i (c_*). ----
      u32 c_cnt = 0, c_cnt0 = 0, c_rm = 0, c_np = 0, c_last = 0, c_first = 0, c_key = 0xFFFFFFFFu; u64 c_racc = 0, c_rsure = 0; i64 c_room[RM];
      u64 movedmask = 0; u32 n_ok = 0, nocand_cls = 0xFFFFFFFFu, fold_k = 0xFFFFFFFFu;
      const bool getenv_nofold = (UF(P.flags) & KS_FLAG_NOFOLD) != 0;      // (test hook: KS_NO_FOLD=1 at upload time)
      // Workers: what the filter phase and the commit read from global memory is requested NOW and arrives while the leader resolves --
      // lane l as window candidate l: its slot and the ladder indices of its filter thresholds (DevState::lowi);
      // lane l as round pod l (if this wave evaluated its class): the first groups Topology.Record will visit.
      u32 wslot = 0xFFFFFFFFu; u64 li[2] = {~0ull, ~0ull}; u32 pr_nrec = 0, pr_w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      if (wv != 0) {
        if ((u32)lane < nwin) wslot = (u32)lane < tb.E ? (u32)lane : tb.E + (ord_lds_r ? ord_l[lane - tb.E] : ord_g[lane - tb.E]);
        if (wslot != 0xFFFFFFFFu && wslot >= tb.E) { li[0] = ((const GA u64*)S.lowi)[2 * (size_t)wslot]; if constexpr (RM > 4) li[1] = ((const GA u64*)S.lowi)[2 * (size_t)wslot + 1]; }
        if (kw < nwk && (u32)lane < rn && rc.pw[par][lane] == kw) {
          const GA ClsPlan* pl = plans + ((u32)(rc.qe[par][lane] >> 32) & 0x7FFFFFFFu);
          pr_nrec = pl->nrec; const GA u32* rp = (const GA u32*)&pl->rec[0];
#pragma unroll
          for (int i = 0; i < 8; ++i) pr_w[i] = rp[i];
        }
      }
      if (wv == 0) {
#ifdef KS_P2PROBES
        u64 t2p = __builtin_readcyclecounter();
#endif
        c_cnt = c_cnt0 = rc.cnt[lane]; c_rm = rc.rmsk[lane];
#pragma unroll
        for (int i = 0; i < RM; ++i) c_room[i] = rc.room[i][lane];
        // visiting-order key: existing nodes in the caller's order, then new nodes by pod count; within a count the nodes moved
        // by this round come first, the most recently moved one in front, then the untouched ones in window order
        if ((u32)lane < nwin) c_key = (u32)lane < tb.E ? (u32)lane : ((c_cnt << 8) | (64u + (u32)lane));
        const u64 mk_l = (u32)lane < rn ? rc.m[b_w & (KS_MAX_WAVES - 1)] : 0ull, chg_l = (u32)lane < rn ? rc.chg[b_w & (KS_MAX_WAVES - 1)] : 0ull;
        const u32 cnt_last = nwin ? RL(c_cnt, (int)(nwin - 1)) : 0u; const bool window_complete = total <= 64;
        u64 closedmask = 0, rall = 0;
        const bool exact_masks = nG <= 64;      // group bits (g & 63) do not alias: a set bit names one group
        // OR of two per-lane masks over the lanes of `in`: row shifts and row broadcasts on the data-parallel-primitive path (no LDS round trips);
        // with at most 32 groups the upper halves are zero
        const bool groups_hi = nG > 32;
        auto or_masks = [&](bool in, u64 v_r, u64 v_s, u64& o_r, u64& o_s) {
          const u32 rl = wave_or_u32(in ? (u32)v_r : 0u), sl = wave_or_u32(in ? (u32)v_s : 0u); u32 rh = 0, sh_ = 0;
          if (groups_hi) { rh = wave_or_u32(in ? (u32)(v_r >> 32) : 0u); sh_ = wave_or_u32(in ? (u32)(v_s >> 32) : 0u); }
          o_r = (u64)rl | ((u64)rh << 32); o_s = (u64)sl | ((u64)sh_ << 32);
        };
        // Dynamic spread: records into a group of dyn_groups are followed exactly where that is possible.  `mine`: this lane applies the
        // record masks `rm` of one pod placed on a candidate whose single value on dyn_key is `zb` (0xFF: not of that form) and whose
        // requirements the pod changes iff `chg`.  Returns the part of rm that was NOT accounted for exactly (it goes into `rall`).
        const u64 dyn_groups = exact_masks ? UF64(P.dyn_groups) : 0ull; const bool dyn_on = dyn_groups != 0;
        u32 c_zone = 0xFFu;
        auto dslot = [&](u32 g) -> u32 { return (u32)__builtin_popcountll(dyn_groups & ((1ull << g) - 1ull)) & 15u; };
        const bool hrec_on = exact_masks && tb.GH <= 24 && tb.GH != 0;
        if (dyn_on) { c_zone = rc.zone[lane]; ((i32*)rc.dd)[lane] = 0; ((i32*)rc.dd)[lane + 64] = 0; }
        if (hrec_on) { for (int i = 0; i < 6; ++i) rc.hrec32[lane][i] = 0; }
        if (dyn_on || hrec_on) LSYNC();
        u64 c_unsure = 0;      // groups an earlier pod of the round MAY have recorded into on this candidate (a record that is not certain: node filter, late group)
        // the round's certain hostname records on candidate `cand` (wave-uniform): lane g carries group g
        const u32 my_hs = hrec_on ? (u32)ls.hslot_of[lane] : 0xFFu;      // group `lane`'s row of the hostname tables
        auto count_host_one = [&](u64 sure, u32 cand) {
          if (hrec_on && ((sure >> lane) & 1ull) && my_hs < 24u) atomicAdd(&rc.hrec32[cand][my_hs >> 2], 1u << (8u * (my_hs & 3u)));
        };
        // ... of one pod per lane (`mine`), on candidate `cand` (per lane)
        auto count_host_lanes = [&](bool mine, u64 sure, u32 cand) {
          if (hrec_on && mine) for (u64 b = sure; b; b &= b - 1) { const u32 hs = ls.hslot_of[__builtin_ctzll(b)]; if (hs < 24u) atomicAdd(&rc.hrec32[cand & 63u][hs >> 2], 1u << (8u * (hs & 3u))); }
        };
        auto track_records = [&](bool mine, u64 rm, u32 zb, bool chg) -> u64 {
          const u64 db = rm & dyn_groups;
          if (!mine || !db) return rm;
          if (zb != 0xFFu) { for (u64 b = db; b; b &= b - 1) atomicAdd(&rc.dd[dslot((u32)__builtin_ctzll(b))][zb & 7u], 1); return rm & ~dyn_groups; }
          return chg ? rm : (rm & ~dyn_groups);      // a node whose requirement on the key is not In [v] and stays as it is: nothing is counted (topology.go:120-133)
        };
        // the same for ONE pod (wave-uniform arguments): lane g carries group g
        const u32 my_ds = dslot((u32)lane);
        auto track_one = [&](u64 rm, u32 zb) { if (zb != 0xFFu && ((rm & dyn_groups) >> lane) & 1ull) atomicAdd(&rc.dd[my_ds][zb & 7u], 1); };
        // runs: bit i of run_next = round pods i and i+1 belong to one evaluation class and neither evaluation reads a topology counter
        u64 run_next;
        { const u32 w_nx = (u32)__shfl_down((int)b_w, 1); const u64 tf_nx = (u64)(u32)__shfl_down((int)(u32)b_tfull, 1) | ((u64)(u32)__shfl_down((int)(u32)(b_tfull >> 32), 1) << 32);
          run_next = ballot64((u32)lane + 1u < rn && b_tfull == 0 && tf_nx == 0 && w_nx == b_w); }
        u32 k = 0;
        // ---- Scalar selection (KS_FASTSEL).  Most single pods of a round take a candidate the round has not touched yet that stands in front of
        // every candidate it HAS touched: then the pod's node is the first set bit of its snapshot bitmap minus the touched candidates -- base rules
        // (a) / (b) of the rounds, decided on the scalar unit (a wave's vector instructions issue once per 4 cycles, its scalar ones every cycle) --
        // and nothing of the per-candidate bookkeeping below is needed to CHOOSE.  The bookkeeping of the candidates taken that way (`pend`, the
        // pod of each in `pod_of`) is deferred to ONE vector pass (flush) before the general path next needs it.
        //   mc_min: lower bound of the pod counts of the NEW nodes the round has moved (every move event's count; counts only grow)
        u64 pend = 0; u32 mc_min = 0xFFFFFFFFu, pod_of = 0;
        // rdyn: groups of dyn_groups into which the round has recorded EXACTLY (counted in rc.dd, kept out of `rall`).  Only a pod whose class the
        // resolver follows against rc.dd (ClsBrief::dyn tag 1) may go on reading such a group; any other reader -- a class with a second topology
        // item, or a requirement of its own on the key -- has a snapshot evaluation the counts have left behind: the round ends before it.
        u64 rdyn = 0;
        const u64 emask = tb.E >= 64u ? ~0ull : ((1ull << tb.E) - 1ull);      // window lanes holding existing nodes (they keep their place when they take a pod)
#ifndef KS_NO_FASTSEL
        const bool fast_on = true;
#else
        const bool fast_on = false;
#endif
        auto shfl64 = [&](u64 v, int src) -> u64 { return (u64)(u32)__shfl((int)(u32)v, src) | ((u64)(u32)__shfl((int)(u32)(v >> 32), src) << 32); };
        auto flush = [&]() {
          const bool inT = (pend >> lane) & 1ull;
          const int src = inT ? (int)pod_of : lane;
          i64 rq[RM];
#pragma unroll
          for (int i = 0; i < RM; ++i) rq[i] = (i64)shfl64((u64)b_req[i], src);
          const u32 rmq = (u32)__shfl((int)b_reqmask, src);
          const u64 prm = shfl64(b_rmask, src), psu = shfl64(b_rsure, src);
          if (inT) {
#pragma unroll
            for (int i = 0; i < RM; ++i) c_room[i] -= rq[i];
            c_rm |= rmq; c_racc |= prm; c_rsure |= psu; c_unsure |= prm & ~psu; c_np = 1; c_first = pod_of; c_last = pod_of;
            if ((u32)lane >= tb.E) { ++c_cnt; c_key = (c_cnt << 8) | (63u - pod_of); }
            rc.win[pod_of & 63u] = (u8)lane;
          }
          count_host_lanes(inT, psu, (u32)lane);
          pend = 0;
        };
        P2T(19);
        while (k < rn) {
          P2C(15, 1);
          { const u64 tmk = RL64(b_tmask, k); const bool follows = dyn_on && ((RL(b_flags, k) >> 1) & 1u) != 0;
            if (tmk & (follows ? rall : (rall | rdyn))) { CUT(13); CUT(17); break; } }
          u64 mk = RL64(mk_l, k), tfk = RL64(b_tfull, k); const u64 chgk = RL64(chg_l, k); const u32 rmk = RL(b_reqmask, k);
          u64 unk = 0;      // candidates whose answer under the counts of the moment is not known (their requirement on dyn_key is not In [v])
          const u32 hsw = hrec_on ? (RL(b_flags, k) >> 1) : 0u; const bool hsk = (hsw & 2u) != 0;      // ClsBrief::dyn tag 2: hslot = bits 8.., group = bits 16..
          if (dyn_on) {
            const u32 dynk = RL(b_flags, k) >> 1;
            if (dynk & 1u) {
              P2T(13);
              // nextDomainTopologySpread (topologygroup.go:155-182) for a node with ONE domain z: z registered and count(z) + self - min <= maxSkew,
              // with the counts as the round has left them; min over the registered domains the pod allows.
              const u32 g = (dynk >> 8) & 0xFFu, self = (dynk >> 16) & 1u, q = UF(rc.dynq[par][k]), pd = q >> 24; const i32 maxskew = (i32)(q & 0xFFFFFFu);
              const u64 reg = tb.g_reg[g];
              LSYNC();
              i32 cz = 0, dz = 0; const bool zin = (u32)lane < 8u && ((reg >> lane) & 1ull);
              if ((u32)lane < 8u) { dz = rc.dd[dslot(g)][lane]; cz = tb.gcnt[(size_t)g * 64 + lane] + dz; }
              const u32 minc = lanes8_min_u32((zin && ((pd >> lane) & 1u)) ? (u32)cz : 0xFFFFFFFFu);
              const u64 VZ = ballot64(zin && (i64)cz + (i64)self - (i64)minc <= (i64)maxskew);
              const bool anyd = ballot64((u32)lane < 8u && dz != 0) != 0;
              const u64 mok = UF64(rc.mo[RL(b_w, k) & (KS_MAX_WAVES - 1)]);
              const u64 pinned = ballot64(c_zone != 0xFFu);
              const u64 okz = ballot64(c_zone != 0xFFu && ((VZ >> (c_zone & 63u)) & 1ull));
              if (anyd) unk = mok & ~pinned;
              mk = (mok & okz) | (anyd ? unk : (mk & ~pinned));
              tfk &= ~(1ull << g);
              P2T(21);
            }
          }
          P2T(13);
          if (fast_on && !((run_next >> k) & 1ull)) {      // (a run of equivalent pods is SWEEP's / CLIMB's: one step for the lot)
            const u64 un = mk & ~movedmask;
            if (un) {
              const u32 u = (u32)__builtin_ctzll(un); const u64 ubit = 1ull << u;
              bool ok = (movedmask & emask & (ubit - 1ull)) == 0;      // no touched existing node stands before it (they keep their place)
              u32 cu = 0;
              if (u >= tb.E) { cu = RL(c_cnt0, u); ok = ok && mc_min > cu; }      // every touched new node holds more pods: it stands behind
              if (ok) {
                if ((unk >> u) & 1ull) { CUT(13); CUT(18); break; }
                const bool chgb = (chgk & ubit) != 0; const u64 rmk64 = RL64(b_rmask, k); u64 inx = rmk64;
                if (rmk64 & dyn_groups) {      // (track_records / track_one for one pod on candidate u)
                  const u32 zb = RL(c_zone, u);
                  if (zb != 0xFFu) { track_one(rmk64, zb); inx = rmk64 & ~dyn_groups; rdyn = UF64(rdyn | (rmk64 & dyn_groups)); }
                  else if (!chgb) inx = rmk64 & ~dyn_groups;
                }
                rall = UF64(rall | inx); movedmask = UF64(movedmask | ubit); pend = UF64(pend | ubit); if (chgb) closedmask = UF64(closedmask | ubit);
                if (u >= tb.E) mc_min = UF(min(mc_min, cu + 1u));
                if ((u32)lane == u) pod_of = k;
                k = UF(k + 1u); n_ok = k;
                P2C(29, 1);
                P2T(26);
                continue;
              }
            }
          }
          if (pend) flush();
          i64 rqk[RM];
#pragma unroll
          for (int i = 0; i < RM; ++i) rqk[i] = (i64)RL64(b_req[i], k);
          // r: consecutive pods of this evaluation class whose evaluation reads no topology counter -- they differ at most in what
          // they record, so where one goes is decided by the same bitmap and the same requests
          u32 r = 1;
          if (tfk == 0) { const u64 nx = ~(run_next >> k); r = 1u + (nx ? (u32)__builtin_ctzll(nx) : 63u); }
          u64 A = mk;
          if (mk & movedmask) {      // candidates the round already used
            const u64 zm = RL64(b_zmask, k);
            // the resource screen again, with what the round put on them; and their own hostname counters: an item that needs the
            // counter at 0 (anti-affinity, spread with maxSkew - self == 0) is certain to fail once a pod of the round recorded into its group
            bool ok = true;
#pragma unroll
            for (int i = 0; i < RM; ++i) if ((((c_rm | rmk) >> i) & 1u) && rqk[i] > c_room[i]) ok = false;
            if (hsk) {       // the candidate took `extra` certain pods of the item's group since the snapshot: it accepts while that stays within its slack
              const u32 hsl = (hsw >> 8) & 31u; const u32 extra = (rc.hrec32[lane][hsl >> 2] >> (8u * (hsl & 3u))) & 255u;
              if (extra > (u32)rc.hslack[RL(b_w, k) & (KS_MAX_WAVES - 1)][lane]) ok = false;
            } else if (exact_masks && (tfk & zm & c_rsure)) ok = false;
            A = mk & (~movedmask | ballot64(ok));
          }
          P2T(18);
          if (!A) {
            // nothing in the window takes this pod (and a window only loses acceptors as the round goes on): it needs a deeper scan or a new
            // node -- remember its class so that the next plan hands it to the sequential path instead of opening a round that ends at once
            nocand_cls = RL((u32)(b_e >> 32), k) & 0x7FFFFFFFu;
            // Fold (LEAN kernel): the pod needs a node whose own counter of ONE hostname-keyed group is 0 (anti-affinity, spread with maxSkew - self == 0),
            // every registered hostname that still counts 0 for that group stands in this window (census == the snapshot's zero-count candidates), and
            // none of them takes the pod with what the round did to them: no node anywhere takes it -- scheduler.add goes to the templates
            // (scheduler.go:193-213).  The leader prepares that node while the workers filter and commits it after the round's order moves: the step that
            // would have opened it, its plan and its barriers fall away.
            if constexpr (LEAN) if (k >= 1 && hsk && !getenv_nofold) {
              const u32 hsl = (hsw >> 8) & 31u, gk = (hsw >> 16) & 63u;
              if (((RL64(b_zmask, k) >> gk) & 1ull) && (u32)__builtin_popcountll(UF64(rc.hz0[RL(b_w, k) & (KS_MAX_WAVES - 1)])) == (u32)UF((u32)tb.g_hzero[hsl])) fold_k = k;
            }
            CUT(14); break;
          }
          const u64 un = A & ~movedmask;
          // the first acceptor in the visiting order as the round has changed it: the smallest key (keys are unique)
          const bool inA = (A >> lane) & 1ull;
          const u32 best = wave_min_u32(inA ? c_key : 0xFFFFFFFFu);
          const int bu = __builtin_ctzll(ballot64(inA && c_key == best));
          const bool bu_moved = (movedmask >> bu) & 1ull;
          if ((unk >> bu) & 1ull) { CUT(13); CUT(18); break; }
          if (bu_moved) {
            if ((u32)bu >= tb.E && !window_complete && (best >> 8) > cnt_last) { CUT(15); break; }        // nodes beyond the window may precede it
            if ((closedmask >> bu) & 1ull) { CUT(15); break; }                                              // its requirements changed in this round
            if (tfk & (hsk ? RL64(c_unsure, bu) : RL64(c_racc, bu))) { CUT(13); CUT(19); break; }                                                 // a counter of that node the evaluation reads may have changed (the certain cases were answered above)
          }
          P2T(19);
          const u32 cnt_bu = RL(c_cnt, bu);
          if (r >= 2 && !bu_moved && (u32)bu >= tb.E) {
            // SWEEP: the untouched acceptors that share bu's pod count follow it in window order, and a node that takes a pod goes BEHIND them
            // (front of the next bucket): the next pods of the run take them one each.
            const u64 S0 = un & ballot64(c_cnt == cnt_bu);
            const u32 sN = min(r, (u32)__builtin_popcountll(S0));
            if (sN >= 2) {
              const u32 rank = (u32)__builtin_popcountll(S0 & ((1ull << lane) - 1ull));
              const bool inS = ((S0 >> lane) & 1ull) && rank < sN;
              const u32 pidx = k + rank;
              const int src = inS ? (int)pidx : lane;
              const u64 prm = (u64)(u32)__shfl((int)(u32)b_rmask, src) | ((u64)(u32)__shfl((int)(u32)(b_rmask >> 32), src) << 32);
              const u64 psu = (u64)(u32)__shfl((int)(u32)b_rsure, src) | ((u64)(u32)__shfl((int)(u32)(b_rsure >> 32), src) << 32);
              if (inS) {
#pragma unroll
                for (int i = 0; i < RM; ++i) c_room[i] -= rqk[i];
                c_rm |= rmk; c_racc |= prm; c_rsure |= psu; c_unsure |= prm & ~psu; c_np = 1; c_first = pidx; c_last = pidx; ++c_cnt; c_key = (c_cnt << 8) | (63u - pidx);
                rc.win[pidx] = (u8)lane;
              }
              const u64 S = ballot64(inS);
              count_host_lanes(inS, psu, (u32)lane);
              const u64 inx = track_records(inS, prm, c_zone, (chgk >> lane) & 1ull);
              u64 orr, ors; or_masks(inS, inx, psu, orr, ors);
              if (dyn_on) { u64 od, ou; or_masks(inS && c_zone != 0xFFu, prm & dyn_groups, 0ull, od, ou); rdyn = UF64(rdyn | od); }
              movedmask = UF64(movedmask | S); closedmask = UF64(closedmask | (S & chgk)); rall = UF64(rall | orr);
              mc_min = UF(min(mc_min, cnt_bu + 1u));
              k = UF(k + sN); n_ok = k;
              P2T(26);
              continue;
            }
          }
          // CLIMB: how many pods of the run does bu take in a row?  It stays first while its count does not exceed the next acceptor's
          // (a node that just took a pod stands in FRONT of its new bucket) and while the cumulative requests pass the resource screen.
          u32 t = 1;
          if (r >= 2 && !((chgk >> bu) & 1ull)) {
            u32 t_order = r;
            if ((u32)bu >= tb.E) {
              const u32 other = wave_min_u32((inA && lane != bu) ? c_key : 0xFFFFFFFFu);      // the next acceptor in line
              u32 oc = other == 0xFFFFFFFFu ? 0x00FFFFFFu : (other >> 8);
              if (!window_complete) oc = min(oc, cnt_last);
              t_order = oc >= cnt_bu ? oc - cnt_bu + 1u : 1u;
            }
            const u32 rm_bu = RL(c_rm, bu) | rmk;
            bool okj = true;
#pragma unroll
            for (int i = 0; i < RM; ++i) { const i64 room_bu = (i64)RL64(c_room[i], bu); if (((rm_bu >> i) & 1u) && (i64)(lane + 1) * rqk[i] > room_bu) okj = false; }
            const u64 bal = ballot64(okj);
            const u32 t_res = ~bal ? (u32)__builtin_ctzll(~bal) : 64u;
            t = max(1u, min(min(r, t_order), t_res));
          }
          u64 orr = RL64(b_rmask, k), ors = RL64(b_rsure, k), inx = orr, uns = 0;
          {
            const u32 zb = RL(c_zone, bu); const bool chg_bu = (chgk >> bu) & 1ull;
            if (t >= 2) {
              const bool inrun = (u32)lane >= k && (u32)lane < k + t;
              u64 o_full, o_s; or_masks(inrun, b_rmask, b_rsure, o_full, o_s);
              if (hrec_on && o_full) { u64 o_u, o_d; or_masks(inrun, b_rmask & ~b_rsure, 0ull, o_u, o_d); uns = o_u; count_host_lanes(inrun, b_rsure, (u32)bu); }
              orr = o_full; ors = o_s; inx = o_full;
              if (dyn_on && (o_full & dyn_groups)) { (void)track_records(inrun, b_rmask, zb, chg_bu); inx = (zb != 0xFFu || !chg_bu) ? (o_full & ~dyn_groups) : o_full; if (zb != 0xFFu) rdyn = UF64(rdyn | (o_full & dyn_groups)); }
            } else {
              uns = orr & ~ors; count_host_one(ors, (u32)bu);
              if (dyn_on && (orr & dyn_groups)) { track_one(orr, zb); inx = (zb != 0xFFu || !chg_bu) ? (orr & ~dyn_groups) : orr; if (zb != 0xFFu) rdyn = UF64(rdyn | (orr & dyn_groups)); }
            }
          }
          if (lane == bu) {
#pragma unroll
            for (int i = 0; i < RM; ++i) c_room[i] -= (i64)t * rqk[i];
            c_rm |= rmk; c_racc |= orr; c_rsure |= ors; c_unsure |= uns; if (c_np == 0) c_first = k; c_np += t; c_last = k + t - 1;
            if ((u32)bu >= tb.E) { c_cnt += t; c_key = (c_cnt << 8) | (63u - (k + t - 1)); }
          }
          if ((u32)lane >= k && (u32)lane < k + t) rc.win[lane] = (u8)bu;
          if ((u32)bu >= tb.E) mc_min = UF(min(mc_min, cnt_bu + 1u));
          movedmask = UF64(movedmask | (1ull << bu));
          if ((chgk >> bu) & 1ull) closedmask = UF64(closedmask | (1ull << bu));
          rall = UF64(rall | inx);
          k = UF(k + t); n_ok = k;
          P2T(26);
        }
        if (pend) flush();
        P2T(13);
        if (n_ok == rn) CUT(16);
        rc.npods[lane] = (u8)c_np; rc.lastpod[lane] = (u8)c_last; rc.firstpod[lane] = (u8)c_first; rc.rmsk_new[lane] = c_rm;
#pragma unroll
        for (int i = 0; i < RM; ++i) rc.roomrem[i][lane] = c_room[i];
        if (lane == 0) rc.n_ok = n_ok;
        P2T(14);
      }
      __syncthreads();
#ifdef KS_PROBES
      { const u64 t2 = __builtin_readcyclecounter(); CTR(23, t2 - t_ph); t_ph = t2; }
#endif
      // ---- P3: per node, the instance-type filter with the totals.  A node whose last pod changes its requirements is filtered by the wave
      //      that evaluated that pod (the new requirement sits in its LDS slots); the others only need what the resolver published, so
      //      they are dealt out over ALL worker waves (a run of equivalent pods would otherwise leave six of seven waves idle) ----
      n_ok = UF(rc.n_ok);
      // worker lane u as candidate u:
      bool committer = false, filtered = false, my_chg = false; u32 n_np = 0, n_rm = 0, n_pres = 0, n_comp = 0, n_chgkeys = 0; u32 n_idx[RM];
      u64 filtmask = 0, ranmask = 0, failedmask = 0;     // (little state crosses the barriers: requests / thresholds are re-read from LDS where they are needed)
      if (wv != 0 && n_ok) {
        n_np = (u32)lane < nwin ? rc.npods[lane] : 0u;
        const u32 lastw = rc.pw[par][rc.lastpod[lane] & 63];
        const bool changing = n_np != 0 && ((rc.chg[lastw & (KS_MAX_WAVES - 1)] >> lane) & 1ull);
        my_chg = changing && lastw == kw;
        const u64 plain = ballot64(n_np != 0 && !changing);
        committer = my_chg || (n_np != 0 && !changing && (u32)__builtin_popcountll(plain & ((1ull << lane) - 1ull)) % (u32)(NW - 1) == kw);
        n_rm = rc.rmsk_new[lane];
        // the node's requirement bits after this class was added (publish_eval's arithmetic, per lane); only an evaluating wave has them
        if (kw < nwk) {
          n_pres = ev.present; n_comp = ev.complement;
          for (u32 i = 0; i < cr.ntouch; ++i) {
            const u32 kb = 1u << ((u32)(cr.tkeys >> (5 * i)) & 31u);
            if ((ev.tpres >> i) & 1u) { n_pres |= kb; n_comp = ((ev.tcomp >> i) & 1u) ? (n_comp | kb) : (n_comp & ~kb); }
            if ((ev.tchg >> i) & 1u) n_chgkeys |= kb;
          }
        }
        bool need = my_chg && wslot >= tb.E;
        bool none = false;
#pragma unroll
        for (int i = 0; i < RM; ++i) n_idx[i] = 0;
#ifdef KS_P2PROBES
        u64 t3p = __builtin_readcyclecounter();
        if (wv == 1 && lane == 0) ls.ctr[17] += t3p - t_ph;        // worker 1: up to the totals
#endif
        {
          // Totals per candidate, then lower_bound over the ascending distinct Allocatable values of every requested resource.  A resource whose
          // total stays under its threshold keeps its ladder index; one that crosses it resumes from that index: three probes in flight (one
          // LDS round trip) settle almost every case, a branch-free binary search with a wave-uniform trip count the rest.
          // (Every read below is unconditional -- rows past R are clamped to row R-1 and their results ignored -- so that the reads of a step are
          // issued together: a wave-uniform branch around a read ends the scheduling region and costs one LDS round trip per read.)
          i64 tot[RM]; u32 lo[RM], hi[RM]; const bool cw = committer && wslot >= tb.E; bool more = false; const u32 Rm1 = tb.R - 1u;
          i64 rq0[RM], rm0[RM], rr0[RM], lw0[RM];
#pragma unroll
          for (int i = 0; i < RM; ++i) { rq0[i] = rc.req0[i][lane]; rm0[i] = rc.room[i][lane]; rr0[i] = rc.roomrem[i][lane]; lw0[i] = rc.low0[i][lane]; }
#pragma unroll
          for (int i = 0; i < RM; ++i) { tot[i] = rq0[i] + (rm0[i] - rr0[i]); lo[i] = 0; hi[i] = 0; if ((u32)i < tb.R && cw && ((n_rm >> i) & 1u) && tot[i] > lw0[i]) { need = true; hi[i] = 1; } }
          if (ballot64(need)) {
            u32 gc[RM], gmax = 0, row[RM];
#pragma unroll
            for (int i = 0; i < RM; ++i) { row[i] = min((u32)i, Rm1) * tb.ge_stride; gc[i] = tb.ge_cnt[min((u32)i, Rm1)]; }
#pragma unroll
            for (int i = 0; i < RM; ++i) { gc[i] = UF(gc[i]); gmax = max(gmax, gc[i]); }
            i64 pv[RM][3]; u32 ixs[RM][3];
#pragma unroll
            for (int i = 0; i < RM; ++i) {
              const u32 old = (u32)(li[i >> 2] >> (16 * (i & 3))) & 0xFFFFu; const bool crossing = hi[i] != 0;
              lo[i] = crossing ? (old == 0xFFFFu ? 0u : old + 1u) : old; hi[i] = crossing ? gc[i] : lo[i];      // (lo == hi: settled)
#pragma unroll
              for (int j = 0; j < 3; ++j) { ixs[i][j] = lo[i] + (u32)j; pv[i][j] = tb.ge_vals[(size_t)(row[i] + (ixs[i][j] < gc[i] ? ixs[i][j] : 0u))]; }
            }
#pragma unroll
            for (int i = 0; i < RM; ++i) {
#pragma unroll
              for (int j = 0; j < 3; ++j) if (ixs[i][j] >= gc[i]) pv[i][j] = INT64_MAX;
              const bool act = lo[i] < hi[i];
              const u32 adv = pv[i][0] >= tot[i] ? 0u : (pv[i][1] >= tot[i] ? 1u : (pv[i][2] >= tot[i] ? 2u : 3u));
              if (act) { if (adv < 3u) { lo[i] += adv; hi[i] = lo[i]; } else { lo[i] = min(lo[i] + 3u, hi[i]); if (lo[i] < hi[i]) more = true; } }
            }
            if (ballot64(more)) {
              const u32 steps = 32u - (u32)__builtin_clz(gmax | 1u);
              for (u32 st = 0; st < steps; ++st) {
                i64 v[RM]; u32 mid[RM];
#pragma unroll
                for (int i = 0; i < RM; ++i) { mid[i] = (lo[i] + hi[i]) >> 1; v[i] = tb.ge_vals[(size_t)(row[i] + (lo[i] < hi[i] ? mid[i] : 0u))]; }
#pragma unroll
                for (int i = 0; i < RM; ++i) if (lo[i] < hi[i]) { if (v[i] >= tot[i]) hi[i] = mid[i]; else lo[i] = mid[i] + 1; }
              }
            }
#pragma unroll
            for (int i = 0; i < RM; ++i) if (need && (u32)i < tb.R && ((n_rm >> i) & 1u)) { n_idx[i] = lo[i]; if (lo[i] >= gc[i]) none = true; }
          }
        }
        filtmask = ballot64(need);
#ifdef KS_P2PROBES
        { const u64 now_ = __builtin_readcyclecounter(); if (wv == 1 && lane == 0) ls.ctr[20] += now_ - t3p; }      // worker 1: totals + lower bounds
#endif
        u64 failed = ballot64(need && none);
        ranmask = filtmask & ~failed;
        u32 nf = 0;
        GA u64* const scr = (GA u64*)S.round_scratch + (size_t)kw * 64 * tb.TW;
        for (u64 fm = ranmask; fm; fm &= fm - 1, ++nf) {
          const int u = __builtin_ctzll(fm);
          const bool chg_u = RL((u32)my_chg, u) != 0;
          const u32 su = RL(wslot, u), rmu = RL(n_rm, u), pres_u = RL(n_pres, u), comp_u = RL(n_comp, u), chk = chg_u ? RL(n_chgkeys, u) : 0u;
          const i32 its_u = chg_u ? (i32)RL((u32)ev.it_state, u) : 0; const bool itc = chg_u && its_u != (i32)RL((u32)ev.it0, u);
          u32 idx_u[RM];
#pragma unroll
          for (int i = 0; i < RM; ++i) idx_u[i] = RL(n_idx[i], u);
          const Rec ru = slot_rec(S, tb, su);
          GA u64* const alive = tb.n_alive + (size_t)(su - tb.E) * tb.TW;
          // the node's requirement on key k after the add: from the evaluation's slots if the class touches k, else the record
          auto node_req = [&](int k) {
            KReq q; q.present = (pres_u >> k) & 1u; q.complement = (comp_u >> k) & 1u; q.gt = KS_NOGT; q.lt = KS_NOLT; q.mask = 0; bool hit = false;
            for (u32 i = 0; i < cr.ntouch; ++i) if ((int)((u32)(cr.tkeys >> (5 * i)) & 31u) == k) { q.mask = sh.la_mask[i][u]; if constexpr (BOUNDS) { q.gt = wb.la_gt[i][u]; q.lt = wb.la_lt[i][u]; } hit = true; }
            if (!hit) { q.mask = ru.mask()[k]; if constexpr (BOUNDS) { q.gt = ru.gt()[k]; q.lt = ru.lt()[k]; } }
            return q;
          };
          const bool zc = (tb.key_zone >= 0 && ((chk >> tb.key_zone) & 1u)) || (tb.key_ct >= 0 && ((chk >> tb.key_ct) & 1u));
          u64 allowZ = ~0ull, allowC = ~0ull;
          if (zc) {
            if (tb.key_zone >= 0 && ((pres_u >> tb.key_zone) & 1u)) allowZ = kreq_has_mask(node_req(tb.key_zone), tb.value_int + tb.key_zone * 64, tb.key_nvalues[tb.key_zone]);
            if (tb.key_ct >= 0 && ((pres_u >> tb.key_ct) & 1u)) allowC = kreq_has_mask(node_req(tb.key_ct), tb.value_int + tb.key_ct * 64, tb.key_nvalues[tb.key_ct]);
          }
          bool any = false;
          for (u32 wbase = 0; wbase < tb.TW; wbase += 64) {
            const u32 w = wbase + lane; u64 a = 0;
            if (w < tb.TW) {
              // (the rows are read unconditionally and together -- a branch around each read would cost one memory round trip per row)
              u64 gr[RM];
#pragma unroll
              for (int i = 0; i < RM; ++i) gr[i] = tb.ge_rows[((size_t)min((u32)i, tb.R - 1u) * tb.T + idx_u[i]) * tb.TW + w];
              const u64 old = alive[w]; a = old;
#pragma unroll
              for (int i = 0; i < RM; ++i) a &= ((u32)i < tb.R && ((rmu >> i) & 1u)) ? gr[i] : ~0ull;
              u64 x = ~0ull;
              for (u32 bits = chk; bits; bits &= bits - 1) { const int k = __builtin_ctz(bits); x &= pass_types_word(P, tb, k, node_req(k), w); }
              if (itc) x &= G_its_types[(size_t)its_u * tb.TW + w];
              if (zc && tb.n_ct != 0) {
                u64 acc = 0; const u64 cm = allowC & ((1ull << tb.n_ct) - 1);
                for (u64 zz = allowZ; zz; zz &= zz - 1) { const int z = __builtin_ctzll(zz); if ((u32)z * tb.n_ct >= 64) break; for (u64 cb = cm; cb; cb &= cb - 1) acc |= G_pair_types[((size_t)z * tb.n_ct + __builtin_ctzll(cb)) * tb.TW + w]; }
                x &= acc;
              }
              a &= x;
              scr[(size_t)nf * tb.TW + w] = old; alive[w] = a;
            }
            if (ballot64(a != 0)) any = true;
          }
#ifdef KS_CHECK   /* debug builds declare pseudo-random filters empty: the cancel / restore / shorter-round path must not change results */
          if (((stepc * 2654435761u + (u32)u * 40503u) >> 24) < 6u) any = false;
#endif
          if (!any) failed |= 1ull << u;
        }
        filtered = (filtmask >> lane) & 1ull; failedmask = failed;
        if ((failed >> lane) & 1ull) atomicMin(&rc.fail_at, (u32)rc.firstpod[lane]);
#ifdef KS_P2PROBES
        if (wv == 1) { const u64 now_ = __builtin_readcyclecounter(); if (lane == 0) ls.ctr[12] += now_ - t_ph; }      // worker 1's filters
#endif
      }
      u32 sp_head = q_head, sp_len = q_len, sp_seq = seq;
      bool fold_ok = false; Pub f_pb; u32 f_pod = 0, f_cidx = 0, f_mt = 0; f_pb.slot = 0; f_pb.present = 0; f_pb.complement = 0; f_pb.changed = 0; f_pb.narrowed = 0; f_pb.valid = 0; f_pb.rm = 0; f_pb.count = 0; f_pb.it_state = 0; f_pb.it_before = 0; f_pb.need = false;
#pragma unroll
      for (int i = 0; i < RM; ++i) { f_pb.req_new[i] = 0; f_pb.room_new[i] = 0; }
      if (wv == 0) {
        // While the workers filter, plan the step after this round as if the round commits (a filter comes back empty a handful of times
        // per Solve; the plan is then redone): queue entries and class briefs of the next pods are requested a whole phase early.
        if ((u32)lane < n_ok) { const u32 ck = (u32)(b_e >> 32) & 0x7FFFFFFFu; atomicAnd(&ls.hard[(ck >> 5) & 7u], ~(1u << (ck & 31u))); }
        if (nocand_cls != 0xFFFFFFFFu) { LSYNC(); if (lane == 0) ls.hard[(nocand_cls >> 5) & 7u] |= 1u << (nocand_cls & 31u); }      // (nocand_cls is wave-uniform: the hand-off outside the lane predicate)
        sp_head = q_head + n_ok; if (sp_head >= nP) sp_head -= nP; sp_len = q_len - n_ok; sp_seq = seq + n_ok;
        { u32 idx = sp_head + lane; if (idx >= nP) idx -= nP; if (idx >= nP) idx = 0; pq_e = tb.q[idx]; pq_ok = true; }
        if (n_ok == 0) seq_credit = 1;                 // the head pod needs more than the window offers: take it sequentially
        if (fold_k == 0xFFFFFFFFu) plan(sp_head, sp_len, sp_seq);
        else if constexpr (LEAN) {
          // ---- fold, first half (while the workers filter): NewNode + Node.Add up to and including the instance-type filter for round pod fold_k on
          // a fresh node -- everything that only WRITES the node's own slot (record, hostname counters, alive row).  Nothing of it is visible before
          // the second half commits it; a round that is cancelled, or a template that does not work out at once, simply drops it. ----
          if (lane == 0) rc.mode2[stepc & 1u] = 0xFFu;      // (no plan yet: the workers must not stage a class for the next step)
          spec_mode = 0; pq_ok = false;
          const u64 fe = RL64(b_e, fold_k); f_pod = (u32)fe; f_cidx = (u32)(fe >> 32) & 0x7FFFFFFFu;
          { const GA u32x4* src = (const GA u32x4*)(plans + f_cidx); u32x4* dst = (u32x4*)&sh.cls; dst[lane] = src[lane]; if ((u32)lane + 64 < KS_PLAN_V) dst[lane + 64] = src[lane + 64]; }
          stage_class(tb, sh, lane);
          const ClsPlan& c = sh.cls;
          ClsR fcr; fcr.tol = UF64(c.tol); fcr.reqmask = UF(c.reqmask); fcr.ntouch = UF(c.ntouch); fcr.nhost = UF(c.nhost); fcr.hn_mode = 0; fcr.port_cnt = 0; fcr.vol_cnt = 0; fcr.it_state = 0; fcr.tkeys = UF64(c.tkeys); fcr.eq = UF(c.eq);
#pragma unroll
          for (int i = 0; i < RM; ++i) fcr.req[i] = (i64)UF64(c.req[i]);
          const u32 fnn = UF(nnew);
          if (!UF(c.overflow) && fnn < nMAX) {
            // the first template that survives the pre-checks (scheduler.go:193-213; LEAN: no provisioner has limits)
            bool have = false; u32 m_t = 0; size_t mc = 0;
            for (u32 tmf = 0; tmf < nM && !have; ++tmf) {
              m_t = tmf; mc = (size_t)m_t * nC + f_cidx; bool lany = false;
              for (u32 wbase = 0; wbase < tb.TW; wbase += 64) {
                const u32 w = wbase + lane; u64 a = 0;
                if (w < tb.TW) { a = GC(u64, P.tmpl_types)[(size_t)m_t * tb.TW + w]; scratch[w] = a & G_grid[mc * tb.TW + w]; }
                if (ballot64(a != 0)) lany = true;
              }
              if (!lany) continue;
              if (!UF(GC(u8, P.mc_ok)[mc])) continue;
              have = true;
            }
            if (have) {
              const u32 fs = tb.E + fnn; const Rec fr = slot_rec(S, tb, fs);
              if ((u32)lane < tb.K) { fr.mask()[lane] = GC(u64, P.mc_mask)[mc * tb.K + lane]; fr.gt()[lane] = GC(i32, P.mc_gt)[mc * tb.K + lane]; fr.lt()[lane] = GC(i32, P.mc_lt)[mc * tb.K + lane]; }
              if (lane >= 32 && (u32)lane < 32 + tb.R) { fr.req()[lane - 32] = GC(i64, P.tmpl_daemon)[(size_t)m_t * tb.R + lane - 32]; fr.room()[lane - 32] = INT64_MAX / 2; fr.low()[lane - 32] = INT64_MIN; }
              if (lane == 63) { fr.taints() = GC(u64, P.tmpl_taints)[m_t]; fr.present() = GC(u32, P.mc_present)[mc]; fr.complement() = GC(u32, P.mc_complement)[mc]; fr.it_state() = GC(i32, P.mc_it)[mc]; fr.reqmask() = GC(u32, P.tmpl_daemon_present)[m_t]; fr.porthead() = -1; fr.count() = 0; }
              for (u32 g = lane; g < nG; g += 64) { const i32 hs = GC(i32, P.grp_hslot)[g]; if (hs >= 0) tb.hcnt[(size_t)fs * tb.GH + hs] = tb.g_active[g] ? 0 : -1; }   // Topology.Register(hostname), node.go:47
              __threadfence_block();
              GSYNC();
              u32 fslot = 0xFFFFFFFFu; if (lane == 0) fslot = fs;
              Ev fev; fev.rc = 0; fev.count = 0; fev.reqmask = 0; fev.tchg = 0; fev.tnar = 0; fev.tpres = 0; fev.tcomp = 0; fev.present = 0; fev.complement = 0; fev.it_state = 0; fev.it0 = 0;
#pragma unroll
              for (int i = 0; i < RM; ++i) { fev.room[i] = 0; fev.req[i] = 0; fev.low[i] = INT64_MIN; }
              if (fslot != 0xFFFFFFFFu) eval_node<BOUNDS, LEAN, RM>(P, S, tb, sh, wb, fslot, false, true, fev, lane, tprobe, fcr);
              if (ballot64(fev.rc == 2) & 1ull) {
                publish_eval<BOUNDS, RM>(tb, sh, wb, fev, fslot, true, lane, 0, fcr, f_pb);
                GA u64* const alive = tb.n_alive + (size_t)fnn * tb.TW;
                const u32 keys = f_pb.narrowed;
                const bool zc = (tb.key_zone >= 0 && ((keys >> tb.key_zone) & 1u)) || (tb.key_ct >= 0 && ((keys >> tb.key_ct) & 1u));
                u64 aw[2];
                if (filter_types(P, tb, f_pb, sh, fr, scratch, alive, f_pb.rm, keys, zc, false, lane, tprobe, aw)) {
                  if ((u32)lane < tb.R && ((f_pb.rm >> lane) & 1u)) fr.low()[lane] = sh.low_new[lane];
                  if (lane < 2 && 4 * lane < RM) { u64 li2 = 0; for (int i = 0; i < 4; ++i) li2 |= (u64)((u32)(4 * lane + i) < tb.R ? (sh.low_idx[4 * lane + i] & 0xFFFFu) : 0xFFFFu) << (16 * i); ((GA u64*)S.lowi)[2 * (size_t)fs + lane] = li2; }
                  fold_ok = true; f_mt = m_t;
                }
              }
            }
          }
        }
#ifdef KS_P2PROBES
        { const u64 now_ = __builtin_readcyclecounter(); if (lane == 0) ls.ctr[14] += now_ - t_ph; }      // the leader's own share of the filter phase
#endif
      }
      __syncthreads();
#ifdef KS_PROBES
      { const u64 t2 = __builtin_readcyclecounter(); CTR(24, t2 - t_ph); t_ph = t2; }
#endif
      // ---- P4: commit (or cancel) ----
      const u32 fail_at = UF(rc.fail_at);
      const bool cancelled = fail_at != 0xFFFFFFFFu;
      if (wv != 0 && n_ok) {
        if (cancelled) {      // put the InstanceTypeOptions rows back
          GA u64* const scr = (GA u64*)S.round_scratch + (size_t)kw * 64 * tb.TW; u32 nf = 0;
          for (u64 fm = ranmask; fm; fm &= fm - 1, ++nf) {
            const u32 su = RL(wslot, __builtin_ctzll(fm)); GA u64* const alive = tb.n_alive + (size_t)(su - tb.E) * tb.TW;
            for (u32 w = lane; w < tb.TW; w += 64) alive[w] = scr[(size_t)nf * tb.TW + w];
          }
          GSYNC();
          // the screen let through what no single surviving type can hold: tighten it to the exact per-resource maxima, so the next
          // round (and the sequential path) stops before the pod that does not fit
          for (u64 fm = failedmask; fm; fm &= fm - 1) {
            const u32 su = RL(wslot, __builtin_ctzll(fm));
            recompute_cap<RM>(P, tb, tb.n_alive + (size_t)(su - tb.E) * tb.TW, slot_rec(S, tb, su), lane);
          }
        } else if (kw < nwk) {
          // per pod (lane l = round pod l, if this wave evaluated its class): results + Topology.Record against the node as it was for THAT pod
          const bool mine = (u32)lane < n_ok && rc.pw[par][lane] == kw;
          const int u = mine ? (int)rc.win[lane] : lane;         // (the cross-lane reads below must run with every lane active: an inactive source lane reads as 0)
          const u32 su = (u32)__shfl((int)wslot, u), pres_u = (u32)__shfl((int)n_pres, u), comp_u = (u32)__shfl((int)n_comp, u); const i32 its_u = __shfl(ev.it_state, u);
          if (mine) {
            const u64 qe = rc.qe[par][lane]; const u32 pod = (u32)qe, cidx = (u32)(qe >> 32) & 0x7FFFFFFFu;
            tb.pod_node[pod] = (i32)su; tb.pod_seq[pod] = (i32)(seq0 + (u32)lane); G_pod_reason[pod] = 0;
            const Rec ru = slot_rec(S, tb, su);
            auto node_req = [&](int k) {
              KReq q; q.present = (pres_u >> k) & 1u; q.complement = (comp_u >> k) & 1u; q.gt = KS_NOGT; q.lt = KS_NOLT; q.mask = 0; bool hit = false;
              for (u32 i = 0; i < cr.ntouch; ++i) if ((int)((u32)(cr.tkeys >> (5 * i)) & 31u) == k) { q.mask = sh.la_mask[i][u]; if constexpr (BOUNDS) { q.gt = wb.la_gt[i][u]; q.lt = wb.la_lt[i][u]; } hit = true; }
              if (!hit) { q.mask = ru.mask()[k]; if constexpr (BOUNDS) { q.gt = ru.gt()[k]; q.lt = ru.lt()[k]; } }
              return q;
            };
            const GA ClsPlan* pl = plans + cidx; const u32 nrec = pr_nrec;
            for (u32 t = 0; t < nrec; ++t) {      // Topology.Record, topology.go:120-143
              u32 rv0, rv1, rv2, rv3;              // PlanRec is 16 bytes: g, key, {type, owned_inverse, hslot}, {tidx, filtered, pad}; the first two came in early
              if (t == 0) { rv0 = pr_w[0]; rv1 = pr_w[1]; rv2 = pr_w[2]; rv3 = pr_w[3]; }
              else if (t == 1) { rv0 = pr_w[4]; rv1 = pr_w[5]; rv2 = pr_w[6]; rv3 = pr_w[7]; }
              else { const GA u32* rp = (const GA u32*)&pl->rec[t]; rv0 = rp[0]; rv1 = rp[1]; rv2 = rp[2]; rv3 = rp[3]; }
              const int g = (int)rv0, key = (int)rv1; const u32 type = rv2 & 0xFF, owned_inverse = (rv2 >> 8) & 0xFF, hslot = rv2 >> 16, filt = (rv3 >> 8) & 0xFF;
              if (!owned_inverse) {
                if (!tb.g_active[g]) continue;
                if (filt) {          // TopologyGroup.Counts: TopologyNodeFilter.MatchesRequirements, topologynodefilter.go:57-70
                  const u32 fb = G_grp_filter_off[g], fe = G_grp_filter_off[g + 1]; bool match = fb == fe;
                  for (u32 f = fb; f < fe && !match; ++f) {
                    bool ok = true; const u32 fp = P.flt.present[f], fc = P.flt.complement[f];
                    for (u32 bits = fp; bits && ok; bits &= bits - 1) {
                      const int k = __builtin_ctz(bits);
                      const KReq in = load_req(fp, fc, P.flt.mask + (size_t)f * tb.K, P.flt.gt + (size_t)f * tb.K, P.flt.lt + (size_t)f * tb.K, k);
                      if (kreq_compatible_fail(node_req(k), in, (tb.wellknown >> k) & 1u, tb.value_int + k * 64, tb.key_nvalues[k])) ok = false;
                    }
                    if (ok && P.flt.it_state[f] && tb.its_fail[its_u * tb.SC + P.flt.it_state[f]]) ok = false;
                    if (ok) match = true;
                  }
                  if (!match) continue;
                }
              }
              if (key == KS_KEY_HOSTNAME) { grp_record_host<true>(S, tb, (int)hslot, su); continue; }
              const KReq q = node_req(key);
              if (!q.present) continue;
              if (owned_inverse || type == 2) { for (u64 b = q.mask; b; b &= b - 1) grp_record<true>(tb, g, __builtin_ctzll(b)); }
              else if (!q.complement && __builtin_popcountll(q.mask) == 1) grp_record<true>(tb, g, __builtin_ctzll(q.mask));
            }
          }
        }
      }
      u64 M_moved = 0;
      if (wv == 0 && !cancelled && n_ok) {      // (while the workers record: the order array and the bucket starts are the leader's alone)
#ifdef KS_P2PROBES
          u64 t2p = __builtin_readcyclecounter();
#endif
          // ---- visiting order: every moved node leaves its place and enters the FRONT of the bucket of its final count (the
          // most recently moved one in front); untouched nodes keep their relative order.  One pass over the affected range. ----
          const bool mvd = (u32)lane >= tb.E && (u32)lane < nwin && c_np != 0;
          const u64 M = ballot64(mvd); M_moved = M;                       // by window lane
          if (M) {
            const u32 nM = (u32)__builtin_popcountll(M);
            const u64 Mpos = tb.E >= 64 ? 0ull : (M >> tb.E);            // by position in `ord`
            u32 jw_l = 0; if (mvd) jw_l = ORD_RD((u32)lane - tb.E);
            const u32 cmin = wave_min_u32(mvd ? c_cnt0 : 0xFFFFFFFFu), cmax = ~wave_min_u32(mvd ? ~c_cnt : 0xFFFFFFFFu);      // lowest count a moved node had, highest one has now
            const u32 pmin = (u32)__builtin_ctzll(Mpos);
            auto oldstart = [&](u32 b) -> u32 { return b <= maxc + 1 ? UF(bst_rd(b)) : nnew; };
            // Relocate the untouched elements of [pmin, end of bucket cmax): an element of bucket b moves left by (moved nodes that stood before it) -
            // (moved nodes that end up at or before bucket b's front).  Every net shift is <= 0 -- a moved node that lands at or before b's front
            // came from a lower bucket, i.e. from before b -- so reading ahead of the writes is safe: four chunks of 64 are in flight at a time
            // (one LDS round trip per four chunks instead of one each; the LDS unit executes a wave's instructions in order).
            const u32 nb = cmax - cmin + 2;                                  // buckets cmin .. cmax+1 (the last one only bounds the range)
            u32 ins_keep = 0;                                                // (nb <= 64) lane j: moved nodes whose final count is <= cmin+j
            if (nb > 64) {        // (more count buckets than lanes -- a window spanning very different pod counts: bucket by bucket, one chunk at a time)
              for (u32 b = cmin; b <= cmax; ++b) {
                const u32 s0 = max(oldstart(b), pmin), s1 = oldstart(b + 1);
                u32 ins = 0; for (u64 q = M; q; q &= q - 1) { const int x = __builtin_ctzll(q); if (RL(c_cnt, x) <= b) ++ins; }
                for (u32 base = s0; base < s1; base += 64) {
                  const u32 ii = base + lane; u32 v = 0; const bool in = ii < s1;
                  if (in) v = ORD_RD(ii);
                  if (ord_in_lds) LSYNC(); else GSYNC();
                  if (in) {
                    const bool is_m = ii < 64 && ((Mpos >> ii) & 1ull);
                    const u32 rem_before = ii >= 64 ? nM : (u32)__builtin_popcountll(Mpos & ((1ull << ii) - 1ull));
                    if (!is_m) ORD_WR(ii - rem_before + ins, v);
                  }
                  if (ord_in_lds) LSYNC(); else GSYNC();
                }
              }
            } else {
            u32 st_l = 0xFFFFFFFFu, ins_l = 0;                               // lane j < nb: old start of bucket cmin+j, moved nodes whose final count is <= cmin+j
            if ((u32)lane < nb) { const u32 bb = cmin + (u32)lane; st_l = bb <= maxc + 1 ? bst_rd(bb) : nnew; }      // (per lane: oldstart() is the wave-uniform form)
            if (nb <= nM) { for (u32 j = 0; j < nb; ++j) { const u32 cj = (u32)__builtin_popcountll(ballot64(mvd && c_cnt <= cmin + j)); if ((u32)lane == j) ins_l = cj; } }      // (one ballot per bucket ...)
            else for (u64 q = M; q; q &= q - 1) { const u32 fc = RL(c_cnt, __builtin_ctzll(q)); if ((u32)lane < nb && fc <= cmin + (u32)lane) ++ins_l; }                          // (... or one step per moved node)
            ins_keep = ins_l;
            const u32 endp = RL(st_l, (int)(nb - 1));
            if (pmin < 64) {      // the window's own positions: a moved node leaves a hole, the others close up
              const u32 ii = (u32)lane; const bool in = ii >= pmin && ii < endp; u32 v = 0;
              if (in) v = ORD_RD(ii);
              if (ord_in_lds) LSYNC(); else GSYNC();
              u32 ins = 0; for (u32 j = 1; j + 1 < nb; ++j) { const u32 sj = RL(st_l, (int)j), ij = RL(ins_l, (int)j); if (ii >= sj) ins = ij; }      // (both lane reads by every lane)
              const bool is_m = (Mpos >> ii) & 1ull;
              const u32 rem_before = (u32)__builtin_popcountll(Mpos & ((1ull << ii) - 1ull));
              if (in && !is_m) ORD_WR(ii - rem_before + ins, v);
            }
            // beyond the window every moved node stood before: inside bucket b the shift is the same for all, nM - ins_b (nothing to do where it is 0)
            for (u32 j = 0; j + 1 < nb; ++j) {
              const u32 lo = max(max(RL(st_l, (int)j), 64u), pmin), hi = RL(st_l, (int)(j + 1)), d = nM - RL(ins_l, (int)j);
              if (d == 0 || lo >= hi) continue;
              for (u32 base = lo; base < hi; base += 256) {
                u32 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { const u32 ii = base + 64u * (u32)u + (u32)lane; v[u] = 0; if (ii < hi) v[u] = ORD_RD(ii); }
                if (ord_in_lds) LSYNC(); else GSYNC();
#pragma unroll
                for (int u = 0; u < 4; ++u) { const u32 ii = base + 64u * (u32)u + (u32)lane; if (ii < hi) ORD_WR(ii - d, v[u]); }
              }
            }
            if (ord_in_lds) LSYNC(); else GSYNC();
            }
            // new bucket starts for the counts in (cmin, cmax + 1]
            for (u32 b0 = cmin + 1; b0 <= cmax + 1; b0 += 64) {      // (lane l takes bucket b0 + l; the loop itself is wave-uniform: lane reads inside it)
              const u32 b = b0 + (u32)lane; const bool inb = b <= cmax + 1;
              const u32 os = (inb && b <= maxc + 1) ? bst_rd(b) : nnew;
              const u32 rem_before = os >= 64 ? nM : (u32)__builtin_popcountll(Mpos & ((1ull << os) - 1ull));
              u32 below = 0;      // moved nodes whose final count is < b
              if (nb <= 64) below = ins_keep;       // (b = cmin + 1 + lane: the count of final counts <= cmin + lane is this lane's ins_l)
              else for (u64 q = M; q; q &= q - 1) { const int x = __builtin_ctzll(q); if (RL(c_cnt, x) < b) ++below; }
              if (inb) bst_wr(b, os - rem_before + below);
            }
            if (ord_in_lds) LSYNC(); else GSYNC();
            if (cmax > maxc) maxc = cmax;
            // the moved nodes: front of their final bucket, the most recent move first
            {     // (the lane reads by EVERY lane, the moved ones use them: the emulator's readlane is a wave collective, and M is wave-uniform)
              u32 rank = 0; for (u64 q = M; q; q &= q - 1) { const int x = __builtin_ctzll(q); const u32 cx = RL(c_cnt, x), lx = RL(c_last, x); if (cx == c_cnt && lx > c_last) ++rank; }
              if (mvd) ORD_WR(bst_rd(c_cnt) + rank, jw_l);
            }
            if (ord_in_lds) LSYNC(); else GSYNC();
          }
          P2T(16);
      }
      __syncthreads();      // records read node records; the node commits below rewrite them
      if (wv != 0 && n_ok && !cancelled && committer) {
        const Rec r = slot_rec(S, tb, wslot);
#pragma unroll
        for (int i = 0; i < RM; ++i) if ((u32)i < tb.R) {
          const i64 rem = rc.roomrem[i][lane];
          r.req()[i] = rc.req0[i][lane] + (rc.room[i][lane] - rem); r.room()[i] = rem;
          if (filtered && ((n_rm >> i) & 1u)) r.low()[i] = tb.ge_vals[(size_t)i * tb.ge_stride + n_idx[i]];
        }
        if (filtered) {
          u64 lw[2] = {0, 0};
#pragma unroll
          for (int i = 0; i < RM; ++i) lw[i >> 2] |= (u64)(((u32)i < tb.R && ((n_rm >> i) & 1u)) ? (n_idx[i] & 0xFFFFu) : 0xFFFFu) << (16 * (i & 3));
          ((GA u64*)S.lowi)[2 * (size_t)wslot] = lw[0]; if constexpr (RM > 4) ((GA u64*)S.lowi)[2 * (size_t)wslot + 1] = lw[1];
        }
        r.reqmask() = n_rm;
        if (wslot >= tb.E) r.count() = rc.cnt[lane] + n_np;
        if (my_chg) {
          r.present() = n_pres; r.complement() = n_comp; r.it_state() = ev.it_state;
          for (u32 i = 0; i < cr.ntouch; ++i) if ((ev.tchg >> i) & 1u) {
            const u32 k = (u32)(cr.tkeys >> (5 * i)) & 31u; r.mask()[k] = sh.la_mask[i][lane];
            if constexpr (BOUNDS) { r.gt()[k] = wb.la_gt[i][lane]; r.lt()[k] = wb.la_lt[i][lane]; }
          }
        }
      }
#ifdef KS_P2PROBES
      u64 t2p = __builtin_readcyclecounter();
#endif
      if (wv == 0) {
        if (!cancelled && n_ok) {
          q_head = sp_head; q_len = sp_len; seq = sp_seq; CTR(KS_STAT_POPS, n_ok); CTR(21, 1);
          CTR(KS_STAT_FULLCHECKS, (u32)__builtin_popcountll(M_moved));
        }
        const u32 n_commit = cancelled ? 0u : n_ok;
        if constexpr (LEAN) if (!cancelled && fold_k != 0xFFFFFFFFu) {
          if (fold_ok) {
            // ---- fold, second half (the workers write their nodes' records meanwhile; counters, order array and bucket starts are the leader's):
            // node.go:100-105 + scheduler.go:214-216 for the node prepared during the filter phase; the pod comes right after the round's pods ----
            nnew = UF(nnew); seq = UF(seq); q_head = UF(q_head); q_len = UF(q_len); maxc = UF(maxc);
            const u32 jw = nnew, fs = tb.E + jw; const Rec r = slot_rec(S, tb, fs);
            for (u32 g = lane; g < nG; g += 64) { const i32 hs = GC(i32, P.grp_hslot)[g]; if (hs >= 0 && tb.g_active[g]) atomicAdd(&tb.g_hzero[hs], 1); }      // its registered hostnames join the zero-count census
            LSYNC();
            topology_record<true>(P, S, tb, f_pb, sh, r, fs, lane);
            LSYNC();
            write_record<BOUNDS, RM>(tb, r, f_pb, sh, f_pb.rm, lane);
            if (lane == 0) { r.count() = 1; G_n_tmpl[jw] = (i32)f_mt; tb.pod_node[f_pod] = (i32)fs; tb.pod_seq[f_pod] = (i32)seq; G_pod_reason[f_pod] = 0; }
            // visiting order: appended -> BACK of the count-1 bucket, i.e. position bstart[2]; everything after shifts right
            if (ord_in_lds && nnew + 1 > ord_cap) { for (u32 i = lane; i < nnew; i += 64) ord_g[i] = ord_l[i]; GSYNC(); ord_in_lds = false; }
            if (maxc == 0) { if (lane == 0) { bst_wr(1, 0); bst_wr(2, 1); ORD_WR(0, jw); } maxc = 1; }
            else {
              const u32 ins = UF(bst_rd(2));
              for (u32 hi = nnew; hi > ins; ) {
                const u32 lo = hi > ins + 256 ? hi - 256 : ins; u32 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { const u32 ii = lo + 64u * (u32)u + (u32)lane; v[u] = 0; if (ii < hi) v[u] = ORD_RD(ii); }
                if (ord_in_lds) LSYNC(); else GSYNC();
#pragma unroll
                for (int u = 0; u < 4; ++u) { const u32 ii = lo + 64u * (u32)u + (u32)lane; if (ii < hi) ORD_WR(ii + 1, v[u]); }
                if (ord_in_lds) LSYNC(); else GSYNC();
                hi = lo;
              }
              if (lane == 0) ORD_WR(ins, jw);
              for (u32 cc = 2 + lane; cc <= maxc + 1; cc += 64) bst_wr(cc, bst_rd(cc) + 1u);
            }
            nnew = jw + 1; ++seq; q_head = (q_head + 1 == nP) ? 0 : q_head + 1; q_len--;
            CTR(KS_STAT_POPS, 1); CTR(KS_STAT_FULLCHECKS, 1); CUT(31);
            if (lane == 0) ls.hard[(f_cidx >> 5) & 7u] |= 1u << (f_cidx & 31u);
            __threadfence_block();
            GSYNC();
          }
          pq_ok = false; plan(q_head, q_len, seq);      // (none was made during the filter phase)
        }
#ifdef KS_PROBES
        CTR(8, rn); CTR(9, 1); CTR(10, n_commit); CTR(11, n_ok); CTR(27, __builtin_readcyclecounter() - t_round); CTR(25, __builtin_readcyclecounter() - t_ph);
#endif
        pf_ok = false; r_valid = false;
        if (cancelled) {      // nothing was committed: plan again from the queue as it is
          if (spec_mode == 2) plan_par ^= 1u;
          if (fail_at >= 2) round_lim = fail_at; else seq_credit = fail_at + 1;
          CTR(KS_STAT_FULLFAILS, 1);
          pq_ok = false; plan(q_head, q_len, seq);
        }
      }
      else if (!cancelled && UF(rc.mode2[stepc & 1u]) == 2u && kw < UF(rc.nwk)) {
        // the plan made during the filter phase stands: stage the class this wave evaluates next while the leader finishes the order moves
        const u32 cnx = UF(rc.wcls[UF(rc.par) & 1u][kw]);
        if (UF(sh.cls.c) != cnx) { const GA u32x4* src = (const GA u32x4*)(plans + cnx); u32x4* dst = (u32x4*)&sh.cls; dst[lane] = src[lane]; if ((u32)lane + 64 < KS_PLAN_V) dst[lane + 64] = src[lane + 64]; }
      }
      __syncthreads();
    }
  }

  // ---------------- results ----------------
  if (wv != 0) return;
  GSYNC();
  // (the leader's state is wave-uniform on the GPU; the lane-fibre emulator keeps a copy per lane, and a requeue on the sequential path updates lane 0's: re-read from it)
  q_len = UF(q_len); q_head = UF(q_head); nnew = UF(nnew);
  for (u32 i = lane; i < q_len; i += 64) { u32 idx = q_head + i; if (idx >= nP) idx -= nP; S.unscheduled[i] = (i32)tb.q[idx]; }
  for (u32 j = lane; j < nnew; j += 64) {        // de-interleave the new nodes' records into the SoA result arrays
    const Rec r = slot_rec(S, tb, tb.E + j);
    S.o_present[j] = r.present(); S.o_complement[j] = r.complement(); S.o_it[j] = r.it_state(); S.o_reqmask[j] = r.reqmask();
    for (u32 k = 0; k < tb.K; ++k) { S.o_mask[(size_t)j * tb.K + k] = r.mask()[k]; S.o_gt[(size_t)j * tb.K + k] = r.gt()[k]; S.o_lt[(size_t)j * tb.K + k] = r.lt()[k]; }
    for (u32 rr = 0; rr < tb.R; ++rr) S.o_req[(size_t)j * tb.R + rr] = r.req()[rr];
  }
  if (lane == 0) {
    S.out_counts[0] = nnew; S.out_counts[1] = q_len;
    for (int i = 0; i < 32; ++i) S.stats[i] = ls.ctr[i];
    S.stats[KS_STAT_CYCLES] = __builtin_readcyclecounter() - t_start; S.stats[KS_STAT_ERR] = err;
    if (S.batch_meta) { S.batch_meta[0] = nnew; S.batch_meta[1] = q_len; for (int i = 0; i < 32; ++i) S.batch_meta[2 + i] = S.stats[i]; }
  }
}
#endif      // !KS_SIM || KS_SIM_PACK

#include "ks_pack_rr.inc"

// ------------------------------------------------------------------------------------------------
// probe kernels (truth-table checks on the device)
// ------------------------------------------------------------------------------------------------
__global__ void ks_probe_kernel(ks_req1 a, ks_req1 b, const i32* vint, u32 nv, int wk, ks_req1* out, int* okout) {
  KReq A{a.mask, a.gt, a.lt, (bool)a.present, (bool)a.complement}, B{b.mask, b.gt, b.lt, (bool)b.present, (bool)b.complement};
  KReq r = kreq_intersect(A, B, vint, nv);
  out->mask = r.mask; out->gt = r.gt; out->lt = r.lt; out->present = r.present; out->complement = r.complement;
  *okout = !kreq_compatible_fail(A, B, wk != 0, vint, nv);
}

__device__ __host__ inline void ks_req_facts_of(const KReq& A, const i32* vint, u32 nv, ks_req_facts* o) {
  o->has_mask = kreq_has_mask(A, vint, nv); o->len = kreq_len(A); o->op = kreq_operator(A); o->nidne = kreq_nidne(A) ? 1 : 0; o->len0 = kreq_len0(A) ? 1 : 0;
}
__global__ void ks_probe_has_kernel(ks_req1 a, const i32* vint, u32 nv, ks_req_facts* out) {
  KReq A{a.mask, a.gt, a.lt, (bool)a.present, (bool)a.complement};
  ks_req_facts_of(A, vint, nv, out);
}

// ================================================================================================
// host side
// ================================================================================================
// Device / pinned-host buffer cache.  A controller solves every few seconds with problems of similar size: hipMalloc / hipFree /
// hipHostMalloc per Solve would cost more than the upload itself, so freed blocks are kept (per device, by power-of-two size
// class, a few of each) and handed out again.  Thread-safe; blocks never migrate between devices.
#include <mutex>
#include <thread>
#include <map>
namespace {
struct DevPool {
  std::mutex mu; std::map<std::pair<int, size_t>, std::vector<void*>> dev_free, host_free; std::map<int, std::vector<hipStream_t>> streams;
  size_t dev_cached = 0, host_cached = 0;       // bytes held: a consolidation pass keeps hundreds of what-if problems alive at once, so the cache is bounded by bytes, not entries
  static constexpr size_t kDevCap = (size_t)16 << 30, kHostCap = (size_t)2 << 30;
  int get_stream(int device, hipStream_t* out) {
    { std::lock_guard<std::mutex> g(mu); auto& v = streams[device]; if (!v.empty()) { *out = v.back(); v.pop_back(); return KS_OK; } }
    HIPCHK(hipStreamCreate(out)); return KS_OK;
  }
  void put_stream(int device, hipStream_t st) { if (!st) return; std::lock_guard<std::mutex> g(mu); streams[device].push_back(st); }
  static size_t klass(size_t bytes) { size_t c = 4096; while (c < bytes) c <<= 1; return c; }
  int get(int device, size_t bytes, bool host, void** out) {
    const size_t c = klass(bytes);
    { std::lock_guard<std::mutex> g(mu); auto& v = (host ? host_free : dev_free)[{device, c}]; if (!v.empty()) { *out = v.back(); v.pop_back(); (host ? host_cached : dev_cached) -= c; return KS_OK; } }
    if (host) HIPCHK(hipHostMalloc(out, c, hipHostMallocDefault)); else HIPCHK(hipMalloc(out, c));
    return KS_OK;
  }
  void put(int device, size_t bytes, bool host, void* p) {
    if (!p) return;
    const size_t c = klass(bytes);
    { std::lock_guard<std::mutex> g(mu); size_t& held = host ? host_cached : dev_cached; if (held + c <= (host ? kHostCap : kDevCap)) { (host ? host_free : dev_free)[{device, c}].push_back(p); held += c; return; } }
    if (host) hipHostFree(p); else hipFree(p);
  }
};
DevPool& pool() { static DevPool* p = new DevPool(); return *p; }      // never destroyed: the HIP runtime may already be gone at exit
// A temporary device block for the duration of one call (returned to the pool on every exit path).
struct TmpDev {
  int device; size_t bytes = 0; void* p = nullptr;
  explicit TmpDev(int dev) : device(dev) {}
  int alloc(size_t n) { bytes = n ? n : 1; return pool().get(device, bytes, false, &p); }
  ~TmpDev() { pool().put(device, bytes, false, p); }
  template <class T> T* as() const { return (T*)p; }
};
}  // namespace

struct ks_dev_problem {
  int device = 0;
  DevProb h{};                     // host copy of the device view (pointers are device pointers)
  DevProb* d_prob = nullptr;       // the same struct in device memory (for ks_pack)
  DevState hs{}; DevState* d_state = nullptr;
  // one device arena: [copied from the host | zero-filled | 0xFF-filled | uninitialised]; sub-allocations are 256-byte aligned.
  // ks_problem_upload lays the problem out twice: a measuring pass sizes the regions, a placing pass hands out pointers and
  // packs the host data into ONE pinned staging buffer, which goes over in ONE transfer.
  bool measure = true; size_t sz[4] = {0, 0, 0, 0}, off[4] = {0, 0, 0, 0}; u8* base[4] = {nullptr, nullptr, nullptr, nullptr};
  u8* arena = nullptr; size_t arena_bytes = 0; u8* stage = nullptr; size_t stage_bytes = 0;
  hipStream_t stream = nullptr;
  bool tables_built = false;
  ks_problem src{};              // the host arrays this problem was uploaded from (pointer identity decides what a what-if can share with it)
  bool any_bounds = false;       // some requirement carries Gt/Lt -> the BOUNDS kernel variant
  bool lean_ok = false;          // none of the rarely used features is present -> the LEAN kernel variant (see ks_pack)
  u32 pp_cap = 0;
  bool view = false;             // a what-if derived from a resident snapshot (ks_whatifs_open): memory and stream belong to its ks_whatif_batch
  int rr_started = 0, rr_code = 0; // the last solve: ks_pack_rr was launched | why it declined (0: it took the Solve; the codes are in ks_pack_rr.inc)
  bool no_multi = false;         // ... over a snapshot with topology groups: the class briefs (round eligibility, certain records) were built for the snapshot's group activity, not this what-if's -- single-wave kernel only
};

static inline size_t ks_align256(size_t b) { return (b + 255) & ~(size_t)255; }
static int arena_take(ks_dev_problem* d, int region, size_t bytes, void** out) {
  bytes = ks_align256(bytes ? bytes : 1);
  if (d->measure) { d->sz[region] += bytes; *out = nullptr; return KS_OK; }
  if (d->off[region] + bytes > d->sz[region]) return fail(KS_ERR_INTERNAL, "upload: the two layout passes disagree");
  *out = d->base[region] + d->off[region]; d->off[region] += bytes; return KS_OK;
}
template <typename T> static int dev_copy(ks_dev_problem* d, const T* src, size_t n, const T** dst) {
  if (n && !src) return fail(KS_ERR_INVALID, "null array in ks_problem");
  void* p = nullptr; const size_t at = d->off[0];
  int rc = arena_take(d, 0, n * sizeof(T), &p); if (rc != KS_OK) return rc;
  if (!d->measure && n) memcpy(d->stage + at, src, n * sizeof(T));
  *dst = (const T*)p; return KS_OK;
}
template <typename T> static int dev_alloc(ks_dev_problem* d, size_t n, T** dst, int fill = -2) {
  void* p = nullptr; int rc = arena_take(d, fill == 0 ? 1 : (fill == -2 ? 3 : 2), n * sizeof(T), &p); if (rc != KS_OK) return rc;
  *dst = (T*)p; return KS_OK;
}
#define TRY(x) do { int rc_ = (x); if (rc_ != KS_OK) return rc_; } while (0)

static int copy_reqsets(ks_dev_problem* d, const ks_reqsets& s, u32 n, u32 K, ReqSetsD* out) {
  out->n = n;
  for (size_t i = 0; i < (size_t)n * K; ++i) if (s.gt[i] != KS_NO_BOUND_GT || s.lt[i] != KS_NO_BOUND_LT) d->any_bounds = true;
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

extern "C" int ks_current_device(void) { int d = 0; if (hipGetDevice(&d) != hipSuccess) return 0; return d; }
extern "C" int ks_problem_device(const ks_dev_problem* d) { return d ? d->device : -1; }
extern "C" int ks_problem_rr_status(const ks_dev_problem* d, int* started, int* decline_code) { if (!d) return fail(KS_ERR_INVALID, "null problem"); if (started) *started = d->rr_started; if (decline_code) *decline_code = d->rr_code; return KS_OK; }

static int validate(const ks_problem* p) {
  if (!p) return fail(KS_ERR_INVALID, "null problem");
  if (p->K > KS_MAX_KEYS) return fail(KS_ERR_UNSUPPORTED, "more than 32 narrow label keys");
  if (p->R > KS_MAX_RES || p->R < 3) return fail(KS_ERR_INVALID, "R must be in [3,8]");
  if (p->S == 0 || p->S > KS_MAX_ITSTATES || p->SC == 0 || p->SC > KS_MAX_ITSTATES) return fail(KS_ERR_INVALID, "S and SC must be in [1,65535]");
  if (p->M == 0) return fail(KS_ERR_INVALID, "no provisioners found");   // provisioner.go:278-280
  if (p->T == 0) return fail(KS_ERR_INVALID, "no instance types");
  if (p->max_new_nodes == 0 && p->P) return fail(KS_ERR_INVALID, "max_new_nodes == 0");
  for (u32 k = 0; k < p->K; ++k) if (p->key_nvalues[k] > 64) return fail(KS_ERR_UNSUPPORTED, "label key with more than 64 distinct values");
  if (p->key_zone >= 0 && p->key_ct >= 0 && (u64)p->key_nvalues[p->key_zone] * p->n_ct > 64) return fail(KS_ERR_UNSUPPORTED, "more than 64 zone x capacity-type pairs");
  if (p->C && !p->cls_vol_off) return fail(KS_ERR_INVALID, "cls_vol_off is required (all zeros when no class mounts volumes)");
  if (p->ND > 64) return fail(KS_ERR_UNSUPPORTED, "more than 64 CSI drivers with volume limits");
  if (p->C && p->cls_vol_off[p->C] && p->ND == 0) for (u32 i = 0; i < p->cls_vol_off[p->C]; ++i) if (p->vol_list[i] != 0xFFFFFFFFu) return fail(KS_ERR_INVALID, "volume entries without volume drivers");
  return KS_OK;
}

extern "C" void ks_problem_free(ks_dev_problem* d) {
  if (!d) return;
  if (d->view) { delete d; return; }
  hipSetDevice(d->device);
  if (d->stream) { hipStreamSynchronize(d->stream); pool().put_stream(d->device, d->stream); }
  pool().put(d->device, d->arena_bytes, false, d->arena);
  pool().put(d->device, d->stage_bytes, true, d->stage);
  delete d;
}

// `base` (optional): a resident problem flattened from the same cluster snapshot.  Catalogue arrays whose HOST pointers are the ones `base`
// was uploaded from are not copied again, and if the whole catalogue is shared so are the tables derived from it (kv_types .. ge_rows).
static int upload_impl(const ks_problem* p, int device, const ks_dev_problem* base, ks_dev_problem** out) {
  *out = nullptr;
  TRY(validate(p));
  if (ks_device_count() <= 0) return fail(KS_ERR_DEVICE, "no gfx950 (MI355X) device visible; libksolve has no CPU path");
  HIPCHK(hipSetDevice(device));
  if (base && (base->device != device || !base->tables_built)) return fail(KS_ERR_INVALID, "the shared problem must be resident on the same device with its tables built (ks_problem_prepare)");
  ks_dev_problem* d = new ks_dev_problem(); d->device = device; d->src = *p;
  // share X: same host array as the base's -> the base's device copy
#define SHARED(field) (base && p->field && p->field == base->src.field)
#define COPY_OR_SHARE(field, count, dst) do { if (SHARED(field)) (dst) = base->h.field; else TRY(dev_copy(d, p->field, (count), &(dst))); } while (0)
  const bool share_cat = base && p->K == base->h.K && p->T == base->h.T && p->R == base->h.R && SHARED(it_present) && SHARED(it_complement) && SHARED(it_mask) && SHARED(it_alloc) && SHARED(it_offer) &&
                         memcmp(p->key_nvalues, base->src.key_nvalues, p->K * sizeof(u32)) == 0;
  struct Guard { ks_dev_problem* d; bool ok = false; ~Guard() { if (!ok) ks_problem_free(d); } } guard{d};
  TRY(pool().get_stream(device, &d->stream));
  auto layout = [&]() -> int {
  DevProb& h = d->h;
  h.P = p->P; h.C = p->C; h.T = p->T; h.TW = (p->T + 63) / 64; h.M = p->M; h.E = p->E; h.K = p->K; h.R = p->R; h.G = p->G; h.GH = p->GH; h.S = p->S; h.SC = p->SC;
  h.NMAX = p->max_new_nodes ? p->max_new_nodes : 1; h.flags = p->flags | (getenv("KS_NO_FOLD") ? KS_FLAG_NOFOLD : 0u); h.n_topologies = p->n_topologies;
  h.wellknown_mask = p->wellknown_mask; h.key_zone = p->key_zone; h.key_ct = p->key_ct; h.n_ct = p->n_ct;
  {   // groups the round resolver may follow exactly (DevProb::dyn_groups)
    h.dyn_groups = 0; h.dyn_key = -1; h.dyn_nz = 0;
    if (p->G <= 64 && !getenv("KS_NO_DYN")) {
      u64 per_key[KS_MAX_KEYS]; for (u32 k = 0; k < KS_MAX_KEYS; ++k) per_key[k] = 0;
      for (u32 g = 0; g < p->n_topologies && g < p->G; ++g) {
        const i32 k = p->grp_key[g];
        if (p->grp_type[g] != 0 || k < 0 || (u32)k >= p->K || !p->grp_active[g] || p->key_nvalues[k] > 8) continue;
        bool unfiltered = p->grp_filter_off[g] == p->grp_filter_off[g + 1];
        for (u32 f = p->grp_filter_off[g]; f < p->grp_filter_off[g + 1]; ++f) if (p->flt.present[f] == 0 && p->flt.it_state[f] == 0) unfiltered = true;      // an empty term matches every node
        if (unfiltered) per_key[k] |= 1ull << g;
      }
      int best = -1; for (u32 k = 0; k < p->K; ++k) if (per_key[k] && (best < 0 || __builtin_popcountll(per_key[k]) > __builtin_popcountll(per_key[best]))) best = (int)k;
      if (best >= 0) {
        u64 m = per_key[best]; while (__builtin_popcountll(m) > 16) m &= ~(1ull << (63 - __builtin_clzll(m)));
        h.dyn_groups = m; h.dyn_key = best; h.dyn_nz = p->key_nvalues[best];
      }
    }
  }
  const u32 K = h.K, R = h.R, T = h.T, TW = h.TW, C = h.C, M = h.M, E = h.E, G = h.G, P = h.P;
  {   // LEAN kernel variant eligibility (see ks_pack)
    bool lean = R <= 4 && h.SC == 1 && !(p->flags & KS_FLAG_STATS);
    for (u32 m = 0; m < M && lean; ++m) lean = p->tmpl_limit_present[m] == 0xFFFFFFFFu || p->tmpl_limit_present[m] == 0;
    for (u32 c = 0; c < C && lean; ++c) lean = p->cls_hn_mode[c] == 0 && p->cls_port_off[c + 1] == p->cls_port_off[c] && p->cls_vol_off[c + 1] == p->cls_vol_off[c];
    d->lean_ok = lean;
  }
  TRY(dev_copy(d, p->key_nvalues, K, &h.key_nvalues)); TRY(dev_copy(d, p->value_int, (size_t)K * 64, &h.value_int));
  COPY_OR_SHARE(it_present, T, h.it_present); COPY_OR_SHARE(it_complement, T, h.it_complement);
  COPY_OR_SHARE(it_mask, (size_t)K * T, h.it_mask); COPY_OR_SHARE(it_alloc, (size_t)R * T, h.it_alloc);
  COPY_OR_SHARE(it_cap, (size_t)R * T, h.it_cap); COPY_OR_SHARE(it_offer, T, h.it_offer);
  h.it_price = nullptr; h.ct_spot = p->ct_spot; h.ct_ondemand = p->ct_ondemand;
  const bool same_pairs = base && p->key_zone == base->src.key_zone && p->key_ct == base->src.key_ct && p->n_ct == base->src.n_ct;
  if (p->it_price && p->key_zone >= 0 && p->key_ct >= 0) { if (same_pairs && SHARED(it_price) && base->h.it_price) h.it_price = base->h.it_price; else TRY(dev_copy(d, p->it_price, (size_t)T * p->key_nvalues[p->key_zone] * p->n_ct, &h.it_price)); }
  h.it_price_lo = h.it_price;
  if (p->it_price_lo && p->key_zone >= 0 && p->key_ct >= 0) { if (same_pairs && SHARED(it_price_lo) && base->h.it_price_lo) h.it_price_lo = base->h.it_price_lo; else TRY(dev_copy(d, p->it_price_lo, (size_t)T * p->key_nvalues[p->key_zone] * p->n_ct, &h.it_price_lo)); }
  const bool same_lattice = base && h.S == base->h.S && h.SC == base->h.SC;
  if (same_lattice && SHARED(its_inter)) h.its_inter = base->h.its_inter; else TRY(dev_copy(d, p->its_inter, (size_t)h.S * h.SC, &h.its_inter));
  if (same_lattice && SHARED(its_fail)) h.its_fail = base->h.its_fail; else TRY(dev_copy(d, p->its_fail, (size_t)h.S * h.SC, &h.its_fail));
  if (same_lattice && SHARED(its_nidne)) h.its_nidne = base->h.its_nidne; else TRY(dev_copy(d, p->its_nidne, h.S, &h.its_nidne));
  if (same_lattice && SHARED(its_types)) h.its_types = base->h.its_types; else TRY(dev_copy(d, p->its_types, (size_t)h.S * TW, &h.its_types));
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
  h.ND = p->ND; h.SW = p->SW;
  TRY(dev_copy(d, p->en_vol_limit, (size_t)E * p->ND, &h.en_vol_limit)); TRY(dev_copy(d, p->en_vol_count, (size_t)E * p->ND, &h.en_vol_count)); TRY(dev_copy(d, p->en_vol_set, (size_t)E * p->SW, &h.en_vol_set));
  TRY(dev_copy(d, p->cls_vol_off, (size_t)C + 1, &h.cls_vol_off)); TRY(dev_copy(d, p->vol_list, C ? p->cls_vol_off[C] : 0, &h.vol_list));
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
  h.derived_shared = share_cat ? 1u : 0u;
  if (share_cat) {
    h.kv_types = base->h.kv_types; h.cmplx_types = base->h.cmplx_types; h.nidnex_types = base->h.nidnex_types; h.pair_types = base->h.pair_types;
    h.ge_vals = base->h.ge_vals; h.ge_cnt = base->h.ge_cnt; h.ge_rows = base->h.ge_rows; h.ge_max = base->h.ge_max;
  } else {
  TRY(dev_alloc(d, (size_t)K * 64 * TW, &h.kv_types, 0)); TRY(dev_alloc(d, (size_t)K * TW, &h.cmplx_types, 0)); TRY(dev_alloc(d, (size_t)K * TW, &h.nidnex_types, 0));
  TRY(dev_alloc(d, (size_t)64 * TW, &h.pair_types, 0));
  {   // ascending distinct Allocatable values per resource (host sort; the rows are built on the device)
    std::vector<i64> vals((size_t)R * T, 0); std::vector<u32> cnt(R, 0);
    for (u32 r = 0; r < R; ++r) { std::vector<i64> v(p->it_alloc + (size_t)r * T, p->it_alloc + (size_t)(r + 1) * T); std::sort(v.begin(), v.end()); v.erase(std::unique(v.begin(), v.end()), v.end()); cnt[r] = (u32)v.size(); std::copy(v.begin(), v.end(), vals.begin() + (size_t)r * T); }
    const i64* dv; const u32* dc; TRY(dev_copy(d, vals.data(), vals.size(), &dv)); TRY(dev_copy(d, cnt.data(), cnt.size(), &dc)); h.ge_vals = (i64*)dv; h.ge_cnt = (u32*)dc;
    h.ge_max = 1; for (u32 r = 0; r < R; ++r) h.ge_max = std::max(h.ge_max, cnt[r]);
    TRY(dev_alloc(d, (size_t)R * T * TW, &h.ge_rows, 0));
  }
  }
  { u8* pl = nullptr; TRY(dev_alloc(d, (size_t)C * sizeof(ClsPlan), &pl, 0)); h.plans = pl; }
  { u8* br = nullptr; TRY(dev_alloc(d, (size_t)C * sizeof(ClsBrief), &br, 0)); h.briefs = br;
    h.ev_tab_size = 64; while (h.ev_tab_size < 2 * C) h.ev_tab_size <<= 1; TRY(dev_alloc(d, (size_t)h.ev_tab_size, &h.ev_tab, 0)); }
  const size_t MC = (size_t)M * C;
  TRY(dev_alloc(d, MC, &h.mc_ok, 0)); TRY(dev_alloc(d, MC, &h.mc_why, 0)); TRY(dev_alloc(d, MC, &h.mc_present)); TRY(dev_alloc(d, MC, &h.mc_complement));
  TRY(dev_alloc(d, MC * K, &h.mc_mask)); TRY(dev_alloc(d, MC * K, &h.mc_gt)); TRY(dev_alloc(d, MC * K, &h.mc_lt)); TRY(dev_alloc(d, MC, &h.mc_it));
  TRY(dev_alloc(d, MC * TW, &h.grid, 0));
  { u8* rb = nullptr; TRY(dev_alloc(d, (size_t)C * sizeof(RRBrief), &rb)); h.rr_briefs = rb; TRY(dev_alloc(d, (size_t)C * (sizeof(RRMemo) / 4), &h.rr_memo)); TRY(dev_alloc(d, MC, &h.rr_mcnrc)); TRY(dev_alloc(d, MC, &h.rr_mcch)); TRY(dev_alloc(d, (size_t)h.ev_tab_size, &h.rr_tab, 0)); TRY(dev_alloc(d, (size_t)h.ev_tab_size, &h.rr_tab2, 0)); TRY(dev_alloc(d, (size_t)C, &h.rr_mi)); TRY(dev_alloc(d, (size_t)C * 8, &h.rr_hot));
    TRY(dev_alloc(d, (size_t)RR_NRC * TW, &h.rr_types)); TRY(dev_alloc(d, (size_t)RR_NODES * 5, &h.rr_nodes)); }
  // state
  DevState& s = d->hs; const size_t NS = (size_t)E + h.NMAX;
  TRY(dev_alloc(d, P, &s.q)); TRY(dev_alloc(d, P, &s.lastlen)); TRY(dev_alloc(d, P, &s.lastgen)); TRY(dev_alloc(d, P, &s.pod_stage)); TRY(dev_alloc(d, P, &s.pod_node)); TRY(dev_alloc(d, P, &s.pod_seq)); TRY(dev_alloc(d, P, &s.pod_reason));
  s.rec_stride = ks_rec_stride(R, K);
  TRY(dev_alloc(d, NS * s.rec_stride, &s.rec, 0));
  TRY(dev_alloc(d, (size_t)h.NMAX, &s.n_tmpl)); TRY(dev_alloc(d, ((size_t)h.NMAX + 1) * TW, &s.n_alive)); TRY(dev_alloc(d, (size_t)8 * 64 * TW, &s.round_scratch)); TRY(dev_alloc(d, 2 * NS, &s.lowi));
  TRY(dev_alloc(d, (size_t)P + 4, &s.bstart, 0)); TRY(dev_alloc(d, (size_t)h.NMAX, &s.order_g));
  TRY(dev_alloc(d, (size_t)G * 64, &s.gcnt)); TRY(dev_alloc(d, G, &s.g_reg)); TRY(dev_alloc(d, G, &s.g_pos)); TRY(dev_alloc(d, G, &s.g_active));
  TRY(dev_alloc(d, (size_t)p->GH * NS, &s.hcnt, 0xFF)); TRY(dev_alloc(d, p->GH, &s.g_hpos)); TRY(dev_alloc(d, p->GH, &s.g_hzero)); TRY(dev_alloc(d, (size_t)M * R, &s.remaining));
  const size_t NM = h.NMAX;
  TRY(dev_alloc(d, NM, &s.o_present)); TRY(dev_alloc(d, NM, &s.o_complement)); TRY(dev_alloc(d, NM * K, &s.o_mask)); TRY(dev_alloc(d, NM * K, &s.o_gt)); TRY(dev_alloc(d, NM * K, &s.o_lt));
  TRY(dev_alloc(d, NM, &s.o_it)); TRY(dev_alloc(d, NM * R, &s.o_req)); TRY(dev_alloc(d, NM, &s.o_reqmask));
  // host-port pool: existing entries + one batch-worth of pod ports (max over stages)
  size_t pp_pool = E ? p->en_port_off[E] : 0;
  for (u32 i = 0; i < P; ++i) { u32 mx = 0; for (u32 st = p->pod_stage_off[i]; st < p->pod_stage_off[i + 1]; ++st) { const u32 c = p->stage_cls[st]; const u32 n = p->cls_port_off[c + 1] - p->cls_port_off[c]; if (n > mx) mx = n; } pp_pool += mx; }
  s.pp_cap = (u32)pp_pool; TRY(dev_alloc(d, pp_pool, &s.pp_entry)); TRY(dev_alloc(d, pp_pool, &s.pp_next));
  s.vol_pad = 0; TRY(dev_alloc(d, (size_t)E * p->ND, &s.vol_cnt)); TRY(dev_alloc(d, (size_t)E * p->SW, &s.vol_set)); TRY(dev_alloc(d, C, &s.wm));
  TRY(dev_alloc(d, 32, &s.stats, 0)); TRY(dev_alloc(d, 4, &s.out_counts, 0)); TRY(dev_alloc(d, P, &s.unscheduled));
  // the two descriptors go last: by now (placing pass) every pointer in them is final
  { const DevProb* dp; const DevState* ds; TRY(dev_copy(d, &d->h, 1, &dp)); TRY(dev_copy(d, &d->hs, 1, &ds)); d->d_prob = (DevProb*)dp; d->d_state = (DevState*)ds; }
  return KS_OK;
  };
  d->measure = true; TRY(layout());
  d->arena_bytes = d->sz[0] + d->sz[1] + d->sz[2] + d->sz[3]; d->stage_bytes = d->sz[0];
  { void* a = nullptr; TRY(pool().get(device, d->arena_bytes, false, &a)); d->arena = (u8*)a; void* st = nullptr; TRY(pool().get(device, d->stage_bytes, true, &st)); d->stage = (u8*)st; }
  d->base[0] = d->arena; d->base[1] = d->base[0] + d->sz[0]; d->base[2] = d->base[1] + d->sz[1]; d->base[3] = d->base[2] + d->sz[2];
  d->measure = false; TRY(layout());
  HIPCHK(hipMemcpyAsync(d->base[0], d->stage, d->sz[0], hipMemcpyHostToDevice, d->stream));
  if (d->sz[1]) HIPCHK(hipMemsetAsync(d->base[1], 0, d->sz[1], d->stream));
  if (d->sz[2]) HIPCHK(hipMemsetAsync(d->base[2], 0xFF, d->sz[2], d->stream));
  // Diagnostic: KS_POISON=<byte> fills the region the kernels must initialise themselves before reading (queue, node records' tails, alive rows,
  // ladder indices, order array, ...).  Results must not depend on it (tests/test_parity.py::test_poisoned_arena, tools/stress_cold.py).
  if (const char* pz = getenv("KS_POISON")) { if (d->sz[3]) HIPCHK(hipMemsetAsync(d->base[3], (int)strtol(pz, nullptr, 0) & 0xFF, d->sz[3], d->stream)); }
  guard.ok = true; *out = d; return KS_OK;
#undef SHARED
#undef COPY_OR_SHARE
}
extern "C" int ks_problem_upload(const ks_problem* p, int device, ks_dev_problem** out) { return upload_impl(p, device, nullptr, out); }
extern "C" int ks_problem_upload_shared(const ks_problem* p, const ks_dev_problem* base, ks_dev_problem** out) {
  if (!base) return fail(KS_ERR_INVALID, "null shared problem");
  return upload_impl(p, base->device, base, out);
}

// ------------------------------------------------------------------------------------------------
// Consolidation what-ifs DERIVED on the device from a resident cluster snapshot (SURVEY 8b `ks_solve_batch(shared, deltas ...)`, App. D.5;
// deprovisioning/helpers.go:48-61,81-84: a what-if is the snapshot minus its candidate nodes plus their pods).  The snapshot -- every node
// an existing node, every bound pod in the batch, flattened and uploaded ONCE with its tables and grid built -- stays resident; a what-if is
// its candidate set.  ks_whatifs_open lays out the mutable state of all n what-ifs in ONE arena (one allocation, one host-to-device copy
// of the candidate masks / remaining resources / descriptors, one memset), and ks_derive_whatifs builds every what-if's batch -- the
// snapshot's queue order restricted to the pods of its candidate nodes -- on the device.  Each what-if is then an ordinary ks_dev_problem
// (a view: it owns nothing) and goes through ks_solve_batch_dev / ks_batch_records_dev / the price stage like any other.
// What the derivation cannot express falls to the caller: it requires a snapshot whose what-ifs differ in nothing but the pod subset,
// the removed nodes and remainingResources (no topology groups, no volume limits: libkshost checks, and flattens per what-if otherwise).
// ------------------------------------------------------------------------------------------------
struct DeriveDesc { const u64* cand_nodes; u32* pod_gid; u32 P_expected, pad; };
// one block per what-if: the snapshot's queue in order, 256 pods per step; a pod joins iff its node is a candidate; ballots give its place
__global__ __launch_bounds__(256) void ks_derive_whatifs(const u32* base_queue, const i32* pod_node, u32 P_base, const DeriveDesc* descs, u32* mismatch) {
  const DeriveDesc d = descs[blockIdx.x];
  __shared__ u32 wave_cnt[4]; __shared__ u32 base_off;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (threadIdx.x == 0) base_off = 0;
  __syncthreads();
  for (u32 i0 = 0; i0 < P_base; i0 += 256) {
    const u32 i = i0 + threadIdx.x; bool in = false; u32 pod = 0;
    if (i < P_base) { pod = base_queue[i]; const i32 nd = pod_node[pod]; in = nd >= 0 && ((d.cand_nodes[nd >> 6] >> (nd & 63)) & 1ull); }
    const u64 b = __ballot(in);
    if (lane == 0) wave_cnt[wv] = (u32)__builtin_popcountll(b);
    __syncthreads();
    u32 off = base_off; for (int w = 0; w < wv; ++w) off += wave_cnt[w];
    if (in) d.pod_gid[off + (u32)__builtin_popcountll(b & ((1ull << lane) - 1ull))] = pod;
    __syncthreads();
    if (threadIdx.x == 0) base_off += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
    __syncthreads();
  }
  if (threadIdx.x == 0 && base_off != d.P_expected) atomicAdd(mismatch, 1u);
}

// Topology of a derived what-if (snapshots whose bound pods carry spread / affinity terms): thread g = group g (G <= 1024).
//   active: some pod of the batch owns the group at its first relaxation stage (NewTopology's Update per pod, topology.go:72-78)
//   counts: NewTopologyGroup's registered domains + countDomains over the cluster pods that are NOT in the batch (topology.go:231-276) =
//           the snapshot-wide totals minus what the candidate nodes' pods contribute
struct TopoDesc { const u32* cand; u32 ncand, pad; u8* active; i32* count; i32* extra; };
__global__ __launch_bounds__(1024) void ks_derive_topology(const TopoDesc* descs, u32 G, u32 GH, u32 n_nodes, const i32* node_cnt, const i32* node_dom, const u64* node_own,
                                                        const i32* tot, const i32* reg, const i32* extra_tot, const i32* grp_hslot, const i32* node_row, u32 n_topologies) {
  const TopoDesc d = descs[blockIdx.x]; const u32 g = threadIdx.x, GW = (G + 63) / 64;      // (blockDim = 64 * GW: node_own holds GW words per node)
  if (g >= G) return;
  u64 own = 0; for (u32 i = 0; i < d.ncand; ++i) own |= node_own[(size_t)d.cand[i] * GW + (g >> 6)];
  d.active[g] = g >= n_topologies ? (u8)1 : (u8)((own >> (g & 63u)) & 1ull);      // (a hostname-keyed inverse group with zero counts constrains nothing: it may exist in every what-if)
  const i32 hs = grp_hslot[g];
  if (hs >= 0) {      // hostname key: the rows are the snapshot's (ks_host_count0); what moves is the number of positive domains that are no existing node
    i32 ex = extra_tot[hs];
    for (u32 i = 0; i < d.ncand; ++i) {      // (node_dom of a hostname-keyed group: the pods that are never in a batch -- they keep counting under a hostname that is no row any more)
      const u32 nd = d.cand[i]; const i32 stay = node_dom[(size_t)g * n_nodes + nd], all = stay + node_cnt[(size_t)g * n_nodes + nd];
      if (node_row[nd] < 0) ex -= (all > 0) - (stay > 0); else ex += stay > 0;
    }
    d.extra[hs] = ex;
    for (u32 v = 0; v < 64; ++v) d.count[(size_t)g * 64 + v] = reg[(size_t)g * 64 + v];
    return;
  }
  i32 c[64];
  for (u32 v = 0; v < 64; ++v) c[v] = tot[(size_t)g * 64 + v];
  for (u32 i = 0; i < d.ncand; ++i) { const u32 nd = d.cand[i]; const i32 dm = node_dom[(size_t)g * n_nodes + nd]; if (dm >= 0) c[dm] -= node_cnt[(size_t)g * n_nodes + nd]; }
  bool owners = (own >> (g & 63u)) & 1ull;      // an inverse group exists while an owner is in the batch, stays bound outside it, or is in no batch at all (topology.go:181-229)
  for (u32 v = 0; v < 64; ++v) { const i32 r = reg[(size_t)g * 64 + v]; const i32 x = (r >= 0 || c[v] > 0) ? (r > 0 ? r : 0) + c[v] : -1; d.count[(size_t)g * 64 + v] = x; if (x > 0) owners = true; }
  if (g >= n_topologies && !owners) d.count[(size_t)g * 64] = INT32_MIN;      // it does not exist in this what-if: the pack kernel's evaluation skips it (see its init)
}

struct ks_whatif_batch {
  int device = 0; u32 n = 0; hipStream_t stream = nullptr; bool own_stream = false;
  u8* arena = nullptr; size_t arena_bytes = 0; u8* stage = nullptr; size_t stage_bytes = 0;
  std::vector<ks_dev_problem*> views;
};
extern "C" void ks_whatifs_free(ks_whatif_batch* b) {
  if (!b) return;
  hipSetDevice(b->device);
  if (b->stream) { hipStreamSynchronize(b->stream); pool().put_stream(b->device, b->stream); }
  pool().put(b->device, b->arena_bytes, false, b->arena);
  pool().put(b->device, b->stage_bytes, true, b->stage);
  for (auto* v : b->views) delete v;
  delete b;
}
extern "C" ks_dev_problem* const* ks_whatifs_problems(ks_whatif_batch* b) { return b ? b->views.data() : nullptr; }
extern "C" uint32_t ks_whatifs_count(const ks_whatif_batch* b) { return b ? b->n : 0; }

// base: the resident snapshot (tables built).  pod_node[base P]: snapshot node index of every snapshot pod (-1: none); node_row[n_nodes]: the node's
// existing-node row in `base`, or -1 (a node no provisioner owns).  What-if w removes nodes cand[cand_off[w] .. cand_off[w+1]); n_pods[w] = pods
// bound to them (the caller knows; checked on the device); remaining[w][M][R] = remainingResources with those nodes gone (scheduler.go:71-75).
extern "C" int ks_whatifs_open(const ks_dev_problem* base, uint32_t n_nodes, const int32_t* pod_node, const int32_t* node_row, uint32_t n, const uint32_t* cand_off,
                               const uint32_t* cand, const uint32_t* n_pods, const int64_t* remaining, const ks_whatif_topo* topo, ks_whatif_batch** out) {
  if (out) *out = nullptr;
  if (!base || !out || (n && (!cand_off || !n_pods || !remaining)) || (n_nodes && (!node_row)) || (base->h.P && !pod_node)) return fail(KS_ERR_INVALID, "null argument");
  if (!base->tables_built) return fail(KS_ERR_INVALID, "the snapshot must be resident with its tables built (ks_problem_prepare)");
  if (base->h.ND || base->h.pod_gid) return fail(KS_ERR_UNSUPPORTED, "what-ifs cannot be derived from a snapshot with volume limits");
  const bool with_topo = base->h.G != 0;
  if (with_topo && (!topo || !topo->node_cnt || !topo->node_dom || !topo->node_own || !topo->tot || (base->h.GH && (!topo->extra_tot || (base->h.E && !topo->grph_base))))) return fail(KS_ERR_UNSUPPORTED, "the snapshot has topology groups: their per-node tables (ks_whatif_topo) are needed to derive what-ifs from it");
  if (with_topo && base->h.G > 1024) return fail(KS_ERR_UNSUPPORTED, "derived what-ifs: at most 1024 topology groups");
  const DevProb& bh = base->h; const u32 E = bh.E, M = bh.M, R = bh.R, K = bh.K, TW = bh.TW, C = bh.C, Pb = bh.P;
  HIPCHK(hipSetDevice(base->device));
  auto b = new ks_whatif_batch(); b->device = base->device; b->n = n;
  struct Guard { ks_whatif_batch* b; bool ok = false; ~Guard() { if (!ok) ks_whatifs_free(b); } } guard{b};
  TRY(pool().get_stream(b->device, &b->stream));
  const size_t NW = ((size_t)n_nodes + 63) / 64, EW = ((size_t)E + 63) / 64;
  // ---- layout: [copied | zeroed | uninitialised], 256-byte aligned pieces ----
  size_t sz[3] = {0, 0, 0};
  auto take = [&](int region, size_t bytes) { const size_t at = sz[region]; sz[region] += ks_align256(bytes ? bytes : 1); return at; };
  struct Lay { size_t cand_bits, removed, remaining, q, lastlen, lastgen, pod_stage, pod_node, pod_seq, pod_reason, rec, n_tmpl, n_alive, lowi, bstart, order_g, rem_state,
               o_present, o_complement, o_mask, o_gt, o_lt, o_it, o_req, o_reqmask, pp_entry, pp_next, wm, stats, out_counts, unscheduled, pod_gid, round_scratch,
               t_active, t_count, t_extra, gcnt, g_reg, g_pos, g_active, hcnt, g_hpos, g_hzero; u32 P, NMAX, pp_cap; };
  std::vector<Lay> L(n);
  const size_t pod_node_at = take(0, (size_t)Pb * sizeof(i32));
  const u32 G = bh.G, GH = bh.GH; const size_t GN = (size_t)G * n_nodes;
  size_t t_cnt_at = 0, t_dom_at = 0, t_own_at = 0, t_tot_at = 0, t_ext_at = 0, t_hbase_at = 0, t_row_at = 0, t_hg_at = 0, t_cand_at = 0, t_desc_at = 0;
  if (with_topo) {
    t_cnt_at = take(0, GN * 4); t_dom_at = take(0, GN * 4); t_own_at = take(0, (size_t)n_nodes * ((G + 63) / 64) * 8); t_tot_at = take(0, (size_t)G * 64 * 4); t_ext_at = take(0, (size_t)GH * 4);
    t_hbase_at = take(0, (size_t)GH * E * 4); t_row_at = take(0, (size_t)n_nodes * 4); t_hg_at = take(0, (size_t)GH * 4); t_cand_at = take(0, (size_t)cand_off[n] * 4); t_desc_at = take(0, (size_t)n * sizeof(TopoDesc));
  }
  const u32 rec_stride = ks_rec_stride(R, K);
  const size_t pp_static = E ? base->src.en_port_off[E] : 0;
  for (u32 w = 0; w < n; ++w) {
    Lay& l = L[w]; const u32 P = n_pods[w]; l.P = P; l.NMAX = P ? P : 1; const size_t NS = (size_t)E + l.NMAX, NM = l.NMAX;
    l.cand_bits = take(0, NW * 8); l.removed = take(0, EW * 8); l.remaining = take(0, (size_t)M * R * 8);
    l.rec = take(1, NS * rec_stride); l.bstart = take(1, ((size_t)P + 4) * 4); l.stats = take(1, 32 * 8); l.out_counts = take(1, 4 * 4);
    l.q = take(2, (size_t)P * 8); l.lastlen = take(2, (size_t)P * 4); l.lastgen = take(2, (size_t)P * 4); l.pod_stage = take(2, (size_t)P * 4); l.pod_node = take(2, (size_t)P * 4);
    l.pod_seq = take(2, (size_t)P * 4); l.pod_reason = take(2, (size_t)P * 4); l.n_tmpl = take(2, NM * 4); l.n_alive = take(2, (NM + 1) * TW * 8); l.lowi = take(2, 2 * NS * 8);
    l.order_g = take(2, NM * 4); l.rem_state = take(2, (size_t)M * R * 8);
    l.o_present = take(2, NM * 4); l.o_complement = take(2, NM * 4); l.o_mask = take(2, NM * K * 8); l.o_gt = take(2, NM * K * 4); l.o_lt = take(2, NM * K * 4); l.o_it = take(2, NM * 4);
    l.o_req = take(2, NM * R * 8); l.o_reqmask = take(2, NM * 4);
    // host-port pool: the existing reservations + at most the ports of every batch pod (an upper bound: 8 per pod would be unheard of; classes with ports carry their count)
    size_t pp = pp_static; if (C && base->src.cls_port_off[C] != pp_static) { u32 mx = 0; for (u32 c = 0; c < C; ++c) mx = std::max(mx, base->src.cls_port_off[c + 1] - base->src.cls_port_off[c]); pp += (size_t)mx * P; }
    l.pp_entry = take(2, pp * 8); l.pp_next = take(2, pp * 4); l.wm = take(2, (size_t)C * 4); l.unscheduled = take(2, (size_t)P * 4); l.pod_gid = take(2, (size_t)P * 4);
    l.pp_cap = (u32)pp; l.round_scratch = n == 1 ? take(2, (size_t)8 * 64 * TW * 8) : 0;      // (a batch of one runs the multi-wave kernel, whose rounds keep rows to restore)
    if (with_topo) {
      l.t_active = take(2, G); l.t_count = take(2, (size_t)G * 64 * 4); l.t_extra = take(2, (size_t)GH * 4);
      l.gcnt = take(2, (size_t)G * 64 * 4); l.g_reg = take(2, (size_t)G * 8); l.g_pos = take(2, (size_t)G * 8); l.g_active = take(2, G);
      l.hcnt = take(2, (size_t)GH * NS * 4); l.g_hpos = take(2, (size_t)GH * 4); l.g_hzero = take(2, (size_t)GH * 4);
    }
  }
  const size_t desc_at = take(0, (size_t)n * sizeof(DeriveDesc)), dprob_at = take(0, (size_t)n * sizeof(DevProb)), dstate_at = take(0, (size_t)n * sizeof(DevState));
  const size_t mismatch_at = take(1, 4);
  b->arena_bytes = sz[0] + sz[1] + sz[2]; b->stage_bytes = sz[0];
  { void* a = nullptr; TRY(pool().get(b->device, b->arena_bytes, false, &a)); b->arena = (u8*)a; void* st = nullptr; TRY(pool().get(b->device, b->stage_bytes, true, &st)); b->stage = (u8*)st; }
  u8* const r0 = b->arena; u8* const r1 = r0 + sz[0]; u8* const r2 = r1 + sz[1];
  // ---- the copied region: pod -> node, per what-if candidate / removed masks and remaining resources, then the descriptors ----
  memset(b->stage, 0, sz[0]);
  if (Pb) memcpy(b->stage + pod_node_at, pod_node, (size_t)Pb * sizeof(i32));
  b->views.resize(n, nullptr);
  DeriveDesc* hd = (DeriveDesc*)(b->stage + desc_at); DevProb* hp = (DevProb*)(b->stage + dprob_at); DevState* hsv = (DevState*)(b->stage + dstate_at);
  TopoDesc* htd = with_topo ? (TopoDesc*)(b->stage + t_desc_at) : nullptr;
  if (with_topo) {
    memcpy(b->stage + t_cnt_at, topo->node_cnt, GN * 4); memcpy(b->stage + t_dom_at, topo->node_dom, GN * 4); memcpy(b->stage + t_own_at, topo->node_own, (size_t)n_nodes * ((G + 63) / 64) * 8);
    memcpy(b->stage + t_tot_at, topo->tot, (size_t)G * 64 * 4); if (GH) { memcpy(b->stage + t_ext_at, topo->extra_tot, (size_t)GH * 4); if (E) memcpy(b->stage + t_hbase_at, topo->grph_base, (size_t)GH * E * 4); }
    memcpy(b->stage + t_row_at, node_row, (size_t)n_nodes * 4); if (cand_off[n]) memcpy(b->stage + t_cand_at, cand, (size_t)cand_off[n] * 4);
    i32* hg = (i32*)(b->stage + t_hg_at); for (u32 g = 0; g < G; ++g) { const i32 hs = base->src.grp_hslot[g]; if (hs >= 0 && (u32)hs < GH) hg[hs] = (i32)g; }
  }
  for (u32 w = 0; w < n; ++w) {
    const Lay& l = L[w];
    u64* cb = (u64*)(b->stage + l.cand_bits); u64* rb = (u64*)(b->stage + l.removed);
    for (u32 i = cand_off[w]; i < cand_off[w + 1]; ++i) {
      const u32 nd = cand[i]; if (nd >= n_nodes) return fail(KS_ERR_INVALID, "candidate node out of range");
      cb[nd >> 6] |= 1ull << (nd & 63u);
      const i32 row = node_row[nd]; if (row >= (i32)E) return fail(KS_ERR_INVALID, "node row out of range"); if (row >= 0) rb[row >> 6] |= 1ull << (row & 63);
    }
    memcpy(b->stage + l.remaining, remaining + (size_t)w * M * R, (size_t)M * R * 8);
    hd[w] = DeriveDesc{(const u64*)(r0 + l.cand_bits), (u32*)(r2 + l.pod_gid), l.P, 0};
    auto* v = new ks_dev_problem(); b->views[w] = v;
    v->device = b->device; v->view = true; v->stream = b->stream; v->tables_built = true; v->any_bounds = base->any_bounds; v->lean_ok = base->lean_ok; v->src = base->src;
    DevProb& h = v->h; h = bh;
    h.P = l.P; h.NMAX = l.NMAX; h.queue = nullptr; h.pod_gid = (const u32*)(r2 + l.pod_gid); h.en_removed = (const u64*)(r0 + l.removed); h.tmpl_remaining = (const i64*)(r0 + l.remaining);
    h.derived_shared = 1;
    if (with_topo) {
      h.grp_active = (const u8*)(r2 + l.t_active); h.grp_count = (const i32*)(r2 + l.t_count); h.grph_count = (const i32*)(r0 + t_hbase_at); h.grph_extra_pos = (const i32*)(r2 + l.t_extra);
      h.hgrp_of = (const i32*)(r0 + t_hg_at); v->no_multi = true;
      htd[w] = TopoDesc{(const u32*)(r0 + t_cand_at) + cand_off[w], cand_off[w + 1] - cand_off[w], 0, (u8*)(r2 + l.t_active), (i32*)(r2 + l.t_count), (i32*)(r2 + l.t_extra)};
    }
    DevState& st = v->hs; st = DevState{};
    st.q = (u64*)(r2 + l.q); st.lastlen = (u32*)(r2 + l.lastlen); st.lastgen = (u32*)(r2 + l.lastgen); st.pod_stage = (i32*)(r2 + l.pod_stage); st.pod_node = (i32*)(r2 + l.pod_node);
    st.pod_seq = (i32*)(r2 + l.pod_seq); st.pod_reason = (u32*)(r2 + l.pod_reason);
    st.rec = r1 + l.rec; st.rec_stride = rec_stride; st.n_tmpl = (i32*)(r2 + l.n_tmpl); st.n_alive = (u64*)(r2 + l.n_alive); st.lowi = (u64*)(r2 + l.lowi); st.round_scratch = n == 1 ? (u64*)(r2 + l.round_scratch) : nullptr;
    st.bstart = (u32*)(r1 + l.bstart); st.order_g = (u32*)(r2 + l.order_g);
    st.gcnt = nullptr; st.g_reg = nullptr; st.g_pos = nullptr; st.g_active = nullptr; st.hcnt = nullptr; st.g_hpos = nullptr; st.g_hzero = nullptr;
    if (with_topo) { st.gcnt = (i32*)(r2 + l.gcnt); st.g_reg = (u64*)(r2 + l.g_reg); st.g_pos = (u64*)(r2 + l.g_pos); st.g_active = (u8*)(r2 + l.g_active); st.hcnt = (i32*)(r2 + l.hcnt);
                     st.g_hpos = (i32*)(r2 + l.g_hpos); st.g_hzero = (i32*)(r2 + l.g_hzero); }
    st.remaining = (i64*)(r2 + l.rem_state);
    st.pp_entry = (u64*)(r2 + l.pp_entry); st.pp_next = (i32*)(r2 + l.pp_next); st.pp_cap = l.pp_cap;
    st.vol_pad = 0; st.vol_cnt = nullptr; st.vol_set = nullptr; st.wm = (u32*)(r2 + l.wm);
    st.stats = (u64*)(r1 + l.stats); st.out_counts = (u32*)(r1 + l.out_counts); st.batch_meta = nullptr; st.unscheduled = (i32*)(r2 + l.unscheduled);
    st.o_present = (u32*)(r2 + l.o_present); st.o_complement = (u32*)(r2 + l.o_complement); st.o_mask = (u64*)(r2 + l.o_mask); st.o_gt = (i32*)(r2 + l.o_gt); st.o_lt = (i32*)(r2 + l.o_lt);
    st.o_it = (i32*)(r2 + l.o_it); st.o_req = (i64*)(r2 + l.o_req); st.o_reqmask = (u32*)(r2 + l.o_reqmask);
    hp[w] = h; hsv[w] = st;
    v->d_prob = (DevProb*)(r0 + dprob_at) + w; v->d_state = (DevState*)(r0 + dstate_at) + w;
  }
  HIPCHK(hipMemcpyAsync(r0, b->stage, sz[0], hipMemcpyHostToDevice, b->stream));
  if (sz[1]) HIPCHK(hipMemsetAsync(r1, 0, sz[1], b->stream));
  if (const char* pz = getenv("KS_POISON")) { if (sz[2]) HIPCHK(hipMemsetAsync(r2, (int)strtol(pz, nullptr, 0) & 0xFF, sz[2], b->stream)); }
  if (n) hipLaunchKernelGGL(ks_derive_whatifs, dim3(n), dim3(256), 0, b->stream, bh.queue, (const i32*)(r0 + pod_node_at), Pb, (const DeriveDesc*)(r0 + desc_at), (u32*)(r1 + mismatch_at));
  if (n && with_topo) hipLaunchKernelGGL(ks_derive_topology, dim3(n), dim3(64 * ((G + 63) / 64)), 0, b->stream, (const TopoDesc*)(r0 + t_desc_at), G, GH, n_nodes, (const i32*)(r0 + t_cnt_at), (const i32*)(r0 + t_dom_at),
                                         (const u64*)(r0 + t_own_at), (const i32*)(r0 + t_tot_at), bh.grp_count, (const i32*)(r0 + t_ext_at), bh.grp_hslot, (const i32*)(r0 + t_row_at), bh.n_topologies);
  u32 mismatch = 0;
  HIPCHK(hipMemcpyAsync(&mismatch, r1 + mismatch_at, 4, hipMemcpyDeviceToHost, b->stream));
  HIPCHK(hipStreamSynchronize(b->stream)); HIPCHK(hipGetLastError());
  if (mismatch) return fail(KS_ERR_INVALID, "n_pods does not match the pods bound to the candidate nodes");
  guard.ok = true; *out = b; return KS_OK;
}
// the snapshot pods behind what-if i's batch, in ITS pod order (= the snapshot's queue order): out[n_pods]
extern "C" int ks_whatifs_pod_ids(ks_whatif_batch* b, uint32_t i, uint32_t* out) {
  if (!b || i >= b->n || !out) return fail(KS_ERR_INVALID, "bad argument");
  HIPCHK(hipSetDevice(b->device));
  if (b->views[i]->h.P) HIPCHK(hipMemcpy(out, b->views[i]->h.pod_gid, (size_t)b->views[i]->h.P * 4, hipMemcpyDeviceToHost));
  return KS_OK;
}

// Build the derived tables + the feasibility grid (idempotent).  Returns the grid kernels' time.
// Launch the static-table kernels for the problems behind `probs` (device descriptor array, n of them) on one stream.
struct StaticDims { u32 rows = 0, C = 0, RT = 0, TW = 1; size_t MC = 0; };
static void static_dims_of(const DevProb& h, StaticDims& a) {
  if (!h.derived_shared) { a.rows = std::max(a.rows, h.K * 64 + 2 * h.K + 64); a.RT = std::max(a.RT, h.R * h.T); }
  a.C = std::max(a.C, h.C); a.MC = std::max(a.MC, (size_t)h.M * h.C); a.TW = std::max(a.TW, h.TW);
}
static void launch_static(const DevProb* probs, u32 n, const StaticDims& a, u32 wave_target, hipStream_t st, hipEvent_t before_grid, u32 row_lo = 0, u32 row_hi = 0xFFFFFFFFu) {
  if (a.rows) hipLaunchKernelGGL(ks_build_type_tables, dim3((a.rows * 64 + 255) / 256, n), dim3(256), 0, st, probs);
  if (a.C) {
    hipLaunchKernelGGL(ks_build_plans, dim3((a.C + 63) / 64, n), dim3(64), 0, st, probs);
    hipLaunchKernelGGL(ks_link_ev, dim3((a.C + 63) / 64, n), dim3(64), 0, st, probs);      // (ev_tab comes zero-filled from the upload; a repeated build finds its own entries again)
    hipLaunchKernelGGL(ks_link_plans, dim3((a.C + 63) / 64, n), dim3(64), 0, st, probs);
    hipLaunchKernelGGL(ks_build_rr, dim3((a.C + 63) / 64, n), dim3(64), 0, st, probs);
    hipLaunchKernelGGL(ks_link_rr, dim3((a.C + 63) / 64, n), dim3(64), 0, st, probs);      // (rr_tab comes zero-filled from the upload, like ev_tab)
    hipLaunchKernelGGL(ks_link_rr2, dim3((a.C + 63) / 64, n), dim3(64), 0, st, probs);
  }
  if (a.RT) hipLaunchKernelGGL(ks_build_ge_rows, dim3((u32)(((size_t)a.RT * 64 + 255) / 256), n), dim3(256), 0, st, probs);
  if (before_grid) hipEventRecord(before_grid, st);
  if (a.MC) {
    hipLaunchKernelGGL(ks_grid_mc, dim3((u32)((a.MC + 255) / 256), n), dim3(256), 0, st, probs);
    const size_t waves = (size_t)a.TW * ks_grid_chunks(a.MC, a.TW, wave_target);      // (an upper bound over the batch: a problem's surplus waves return at once)
    hipLaunchKernelGGL(ks_grid_types, dim3((u32)((waves * 64 + 255) / 256), n), dim3(256), 0, st, probs, wave_target, row_lo, row_hi);
  }
}
// Build the derived tables + the feasibility grid (idempotent).  Returns the grid kernels' time.
static int build_static(ks_dev_problem* d, float* grid_ms, u32 row_lo = 0, u32 row_hi = 0xFFFFFFFFu) {
  HIPCHK(hipSetDevice(d->device));
  hipEvent_t e0 = nullptr, e1 = nullptr; HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
  StaticDims a; static_dims_of(d->h, a);
  launch_static(d->d_prob, 1, a, 8192, d->stream, e0, row_lo, row_hi);
  d->tables_built = row_lo == 0 && (size_t)row_hi >= (size_t)d->h.M * d->h.C;      // (a row range: the grid is whole once the other rows are installed)
  HIPCHK(hipEventRecord(e1, d->stream));
  HIPCHK(hipStreamSynchronize(d->stream));
  HIPCHK(hipGetLastError());
  if (grid_ms) HIPCHK(hipEventElapsedTime(grid_ms, e0, e1));
  hipEventDestroy(e0); hipEventDestroy(e1);
  return KS_OK;
}

extern "C" int ks_problem_prepare(ks_dev_problem* d) {
  if (!d) return fail(KS_ERR_INVALID, "null device problem");
  if (!d->tables_built) TRY(build_static(d, nullptr));
  return KS_OK;
}

extern "C" int ks_feasibility_grid(ks_dev_problem* d, uint64_t* out_grid, float* kernel_ms) {
  if (!d) return fail(KS_ERR_INVALID, "null device problem");
  TRY(build_static(d, kernel_ms));
  if (out_grid) HIPCHK(hipMemcpy(out_grid, d->h.grid, (size_t)d->h.M * d->h.C * d->h.TW * sizeof(u64), hipMemcpyDeviceToHost));
  return KS_OK;
}

// the static grid's rows (template x class pairs, TW words each) split over GPUs.  Rows [row_lo, row_hi) are computed HERE -- every other static table in
// full: they are per class, not per class x type -- and copied out, to host memory and / or into device memory of the caller's (a slice of the buffer an all-gather fills).
extern "C" int ks_feasibility_grid_rows(ks_dev_problem* d, uint32_t row_lo, uint32_t row_hi, uint64_t* out_rows, void* out_rows_dev, float* kernel_ms) {
  if (!d) return fail(KS_ERR_INVALID, "null device problem");
  const size_t MC = (size_t)d->h.M * d->h.C, TW = d->h.TW;
  if (row_lo > row_hi || row_hi > MC) return fail(KS_ERR_INVALID, "grid rows out of range");
  TRY(build_static(d, kernel_ms, row_lo, row_hi));
  const size_t bytes = (size_t)(row_hi - row_lo) * TW * sizeof(u64); const u64* src = d->h.grid + (size_t)row_lo * TW;
  if (out_rows && bytes) HIPCHK(hipMemcpy(out_rows, src, bytes, hipMemcpyDeviceToHost));
  if (out_rows_dev && bytes) HIPCHK(hipMemcpy(out_rows_dev, src, bytes, hipMemcpyDeviceToDevice));
  return KS_OK;
}
// ... and rows computed elsewhere installed (from host or device memory).  `complete` != 0: every row is in now -- the problem solves without building its grid again.
extern "C" int ks_feasibility_grid_install(ks_dev_problem* d, uint32_t row_lo, uint32_t row_hi, const uint64_t* rows, const void* rows_dev, int complete) {
  if (!d) return fail(KS_ERR_INVALID, "null device problem");
  const size_t MC = (size_t)d->h.M * d->h.C, TW = d->h.TW;
  if (row_lo > row_hi || row_hi > MC || (!rows && !rows_dev && row_hi > row_lo)) return fail(KS_ERR_INVALID, "grid rows out of range (or no source)");
  HIPCHK(hipSetDevice(d->device));
  const size_t bytes = (size_t)(row_hi - row_lo) * TW * sizeof(u64); u64* dst = d->h.grid + (size_t)row_lo * TW;
  if (bytes) { if (rows_dev) HIPCHK(hipMemcpy(dst, rows_dev, bytes, hipMemcpyDeviceToDevice)); else HIPCHK(hipMemcpy(dst, rows, bytes, hipMemcpyHostToDevice)); }
  if (complete) d->tables_built = true;
  return KS_OK;
}

// Batched read-back: one descriptor per problem, one block copies that problem's result arrays into a contiguous blob
// (segments in the order of ks_gather_layout), so the host needs three transfers per batch instead of ~16 per problem.
struct GatherDesc { u64 off; u32 P, K, R, TW, N, pad; };
struct GatherSeg { const void* src; u64 bytes; };
__device__ __host__ inline u64 ks_pad8(u64 b) { return (b + 7) & ~7ull; }
__global__ __launch_bounds__(256) void ks_gather(const DevState* states, const GatherDesc* descs, u8* blob) {
  const DevState& s = states[blockIdx.x]; const GatherDesc g = descs[blockIdx.x];
  const u64 P = g.P, N = g.N, K = g.K, R = g.R, TW = g.TW;
  const GatherSeg segs[15] = {{s.pod_node, P * 4}, {s.pod_stage, P * 4}, {s.pod_seq, P * 4}, {s.pod_reason, P * 4}, {s.unscheduled, P * 4}, {s.n_tmpl, N * 4}, {s.n_alive, N * TW * 8},
                              {s.o_req, N * R * 8}, {s.o_reqmask, N * 4}, {s.o_present, N * 4}, {s.o_complement, N * 4}, {s.o_mask, N * K * 8}, {s.o_gt, N * K * 4},
                              {s.o_lt, N * K * 4}, {s.o_it, N * 4}};
  u64 off = g.off;
  for (int i = 0; i < 15; ++i) {
    const u32* src = (const u32*)segs[i].src; u32* dst = (u32*)(blob + off);
    for (u64 j = threadIdx.x; j < segs[i].bytes / 4; j += blockDim.x) dst[j] = src[j];
    off += ks_pad8(segs[i].bytes);
  }
}
static u64 ks_gather_bytes(u64 P, u64 K, u64 R, u64 TW, u64 N) {
  return 5 * ks_pad8(P * 4) + ks_pad8(N * 4) + ks_pad8(N * TW * 8) + ks_pad8(N * R * 8) + 3 * ks_pad8(N * 4) + ks_pad8(N * K * 8) + 2 * ks_pad8(N * K * 4) + ks_pad8(N * 4);
}
static int ks_stats_error(const u64* stats) {
  if (!stats[KS_STAT_ERR]) return KS_OK;
  return fail(-(int)stats[KS_STAT_ERR], stats[KS_STAT_ERR] == (u64)(-KS_ERR_CAPACITY) ? "more new nodes than max_new_nodes" :
              stats[KS_STAT_ERR] == (u64)(-KS_ERR_INTERNAL) ? "pack kernel watchdog: step bound exceeded" :
              stats[KS_STAT_ERR] >= 100 ? "pack kernel self-check failed (KS_CHECK build): visiting order inconsistent" :
              "a pod class exceeds the kernel's per-class limits (12 touched keys / 24 topology groups / 3 hostname groups / 24 recorded groups)");
}
static int download_batch(ks_dev_problem* const* ds, u32 n, const DevState* dsv, const u64* d_meta, ks_result* const* outs) {
  std::vector<u64> meta((size_t)n * 34);
  HIPCHK(hipMemcpy(meta.data(), d_meta, meta.size() * sizeof(u64), hipMemcpyDeviceToHost));
  std::vector<GatherDesc> descs(n); u64 total = 0;
  for (u32 i = 0; i < n; ++i) {
    const DevProb& h = ds[i]->h; ks_result* out = outs[i];
    out->n_new = (u32)meta[(size_t)i * 34]; out->n_unscheduled = (u32)meta[(size_t)i * 34 + 1];
    memcpy(out->stats, &meta[(size_t)i * 34 + 2], 32 * sizeof(u64));
    TRY(ks_stats_error(out->stats));
    descs[i] = GatherDesc{total, h.P, h.K, h.R, h.TW, out->n_new, 0};
    total += ks_gather_bytes(h.P, h.K, h.R, h.TW, out->n_new);
  }
  TmpDev t_desc(ds[0]->device), t_blob(ds[0]->device);
  TRY(t_desc.alloc(n * sizeof(GatherDesc))); TRY(t_blob.alloc(total ? total : 8));
  GatherDesc* d_desc = t_desc.as<GatherDesc>(); u8* d_blob = t_blob.as<u8>();
  HIPCHK(hipMemcpy(d_desc, descs.data(), n * sizeof(GatherDesc), hipMemcpyHostToDevice));
  hipLaunchKernelGGL(ks_gather, dim3(n), dim3(256), 0, ds[0]->stream, dsv, d_desc, d_blob);
  std::vector<u8> blob(total ? total : 8);
  HIPCHK(hipMemcpyAsync(blob.data(), d_blob, total ? total : 8, hipMemcpyDeviceToHost, ds[0]->stream));
  HIPCHK(hipStreamSynchronize(ds[0]->stream));
  for (u32 i = 0; i < n; ++i) {
    const DevProb& h = ds[i]->h; ks_result* out = outs[i]; const u64 P = h.P, K = h.K, R = h.R, TW = h.TW, N = out->n_new;
    const u8* p = blob.data() + descs[i].off;
    auto take = [&](void* dst, u64 bytes) { if (bytes) memcpy(dst, p, bytes); p += ks_pad8(bytes); };
    take(out->pod_node, P * 4); take(out->pod_stage, P * 4); take(out->pod_seq, P * 4); take(out->pod_reason, P * 4); take(out->unscheduled, P * 4);
    take(out->node_tmpl, N * 4); take(out->node_types, N * TW * 8); take(out->node_requests, N * R * 8); take(out->node_requests_present, N * 4);
    take(out->node_present, N * 4); take(out->node_complement, N * 4); take(out->node_mask, N * K * 8); take(out->node_gt, N * K * 4); take(out->node_lt, N * K * 4);
    take(out->node_it_state, N * 4);
  }
  return KS_OK;
}

static int download(ks_dev_problem* d, ks_result* out) {
  const DevProb& h = d->h; const DevState& s = d->hs;
  u32 counts[4]; HIPCHK(hipMemcpy(counts, s.out_counts, sizeof counts, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(out->stats, s.stats, 32 * sizeof(u64), hipMemcpyDeviceToHost));
  out->n_new = counts[0]; out->n_unscheduled = counts[1];
  if (out->stats[KS_STAT_ERR]) return fail(-(int)out->stats[KS_STAT_ERR], out->stats[KS_STAT_ERR] == (u64)(-KS_ERR_CAPACITY) ? "more new nodes than max_new_nodes" :
                                           out->stats[KS_STAT_ERR] == (u64)(-KS_ERR_INTERNAL) ? "pack kernel watchdog: step bound exceeded" :
                                           out->stats[KS_STAT_ERR] >= 100 ? "pack kernel self-check failed (KS_CHECK build): visiting order inconsistent" :
                                           "a pod class exceeds the kernel's per-class limits (12 touched keys / 24 topology groups / 3 hostname groups / 24 recorded groups)");
  const u32 P = h.P, K = h.K, R = h.R, TW = h.TW, N = out->n_new;
  if (P) {
    HIPCHK(hipMemcpy(out->pod_node, s.pod_node, P * sizeof(i32), hipMemcpyDeviceToHost)); HIPCHK(hipMemcpy(out->pod_stage, s.pod_stage, P * sizeof(i32), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(out->pod_seq, s.pod_seq, P * sizeof(i32), hipMemcpyDeviceToHost)); HIPCHK(hipMemcpy(out->unscheduled, s.unscheduled, P * sizeof(i32), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(out->pod_reason, s.pod_reason, P * sizeof(u32), hipMemcpyDeviceToHost));
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
  if (!ds) return fail(KS_ERR_INVALID, "null batch");
  const int device = ds[0]->device;
  HIPCHK(hipSetDevice(device));
  u32 unbuilt = 0;
  for (u32 i = 0; i < n; ++i) { if (ds[i]->device != device) return fail(KS_ERR_INVALID, "batch spans devices"); if (!ds[i]->tables_built) ++unbuilt; }
  if (n == 1 && unbuilt) TRY(build_static(ds[0], nullptr));
  std::vector<DevProb> hp(n); std::vector<DevState> hs(n);
  for (u32 i = 0; i < n; ++i) { hp[i] = ds[i]->h; hs[i] = ds[i]->hs; }
  DevProb* dp = nullptr; DevState* dsv = nullptr; u64* d_meta = nullptr;
  TmpDev t_meta(device), t_dp(device), t_dsv(device);
  if (n == 1) { dp = ds[0]->d_prob; dsv = ds[0]->d_state; }
  else {
    TRY(t_meta.alloc((size_t)n * 34 * sizeof(u64))); d_meta = t_meta.as<u64>();
    for (u32 i = 0; i < n; ++i) hs[i].batch_meta = d_meta + (size_t)i * 34;
    TRY(t_dp.alloc(n * sizeof(DevProb))); TRY(t_dsv.alloc(n * sizeof(DevState))); dp = t_dp.as<DevProb>(); dsv = t_dsv.as<DevState>();
    HIPCHK(hipMemcpy(dp, hp.data(), n * sizeof(DevProb), hipMemcpyHostToDevice)); HIPCHK(hipMemcpy(dsv, hs.data(), n * sizeof(DevState), hipMemcpyHostToDevice));
  }
  hipStream_t st = ds[0]->stream;
  if (n > 1 && unbuilt) {
    // The static tables of the whole batch in seven launches (grid.y = what-if) instead of seven per what-if.  The uploads were queued on
    // the what-ifs' own streams: one wait for all of them first.  (Problems built earlier are rebuilt in place: the build is idempotent.)
    HIPCHK(hipDeviceSynchronize());
    StaticDims a; for (u32 i = 0; i < n; ++i) static_dims_of(ds[i]->h, a);
    launch_static(dp, n, a, std::max(64u, 16384u / n), st, nullptr);
    for (u32 i = 0; i < n; ++i) ds[i]->tables_built = true;
  }
  hipEvent_t e0, e1; HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
  HIPCHK(hipEventRecord(e0, st));
  // dynamic LDS: Allocatable table + visiting-order array.  One Solve gets most of the CU's 160 KiB;
  // batched what-ifs take 64 KiB each so two workgroups share a CU.
  const u32 lds_bytes = n == 1 ? 100u * 1024u : 64u * 1024u;
  bool fast = true;
  for (u32 i = 0; i < n; ++i) { const DevProb& q = ds[i]->h; if (q.G > KS_FAST_G || q.GH > KS_FAST_G || (q.SC > 1 && (size_t)q.S * q.SC > KS_FAST_S * KS_FAST_S) || (size_t)q.R * q.ge_max > KS_FAST_RT || (size_t)q.R * q.ge_max * 8 + 8192 > lds_bytes) fast = false; }
  bool bounds = false; for (u32 i = 0; i < n; ++i) bounds = bounds || ds[i]->any_bounds;
  bool lean = true; for (u32 i = 0; i < n; ++i) lean = lean && ds[i]->lean_ok;
  u32 any_flags = 0; for (u32 i = 0; i < n; ++i) any_flags |= ds[i]->h.flags;
  if (getenv("KS_NO_LEAN") || (any_flags & KS_FLAG_NO_LEAN)) lean = false;      // run the general variant on a problem the LEAN one would take (ksolve.h; the variable: test hook for the whole process)
  // The register-resident kernel (ks_pack_rr.inc) takes a single LEAN Solve without Gt/Lt bounds; it declines what it does not cover -- before
  // or during the run, without having touched the inputs -- and ks_pack below takes over.
  bool rr_done = false;
  for (u32 i = 0; i < n; ++i) { ds[i]->rr_started = 0; ds[i]->rr_code = 0; }
  const bool asked_one_wave = getenv("KS_ONE_WAVE") != nullptr || (any_flags & KS_FLAG_ONE_WAVE);      // KS_ONE_WAVE asks for ks_pack's single-wave variant
  const bool asked_no_rr = getenv("KS_NO_RR") != nullptr || (any_flags & KS_FLAG_NO_RR);                 // KS_NO_RR=1: ks_pack only (A/B, and the parity of both kernels)
#ifdef KS_SIM
  // The emulator runs ks_pack_rr and (-DKS_SIM_PACK, round 5) ks_pack's single-wave variants -- what a what-if batch runs, LEAN and general.  The multi-wave variants'
  // speculation rounds lean on hand-offs between lanes in lockstep that the fibre emulator does not model yet: their results differ THERE, not on the GPU.
  const bool one_wave = true;
#ifdef KS_SIM_PACK
  const bool rr_on = !asked_no_rr && !asked_one_wave;
#else
  const bool rr_on = true;                    // (no ks_pack in this build)
#endif
#else
  const bool one_wave = asked_one_wave;
  const bool rr_on = !asked_no_rr && !asked_one_wave;
#endif
  if (rr_on && n == 1 && lean && !bounds && fast && !ds[0]->view && !(ds[0]->h.flags & KS_FLAG_STATS) && ds[0]->h.rr_briefs) {
    // dynamic LDS: the Allocatable ladders (R x ge_max x 8 bytes), at most what the kernel's static LDS object leaves of the CU's 160 KiB (a problem whose ladders
    // do not fit is declined by the kernel itself: its eligibility test reads the size it was given)
    u32 lds_rr = 0;
    {
      static std::mutex rr_mu; static std::vector<u32> rr_attr;
      std::lock_guard<std::mutex> g(rr_mu);
      if ((size_t)device >= rr_attr.size()) rr_attr.resize(device + 1, 0);
      if (!rr_attr[device]) {
#ifdef KS_SIM
        const u32 room = 44u * 1024u;
#else
        hipFuncAttributes fa; HIPCHK(hipFuncGetAttributes(&fa, (const void*)ks_pack_rr));
        if (fa.sharedSizeBytes + 1024u > 160u * 1024u) return fail(KS_ERR_DEVICE, "ks_pack_rr: the static LDS object leaves no room for the ladders");
        const u32 room = std::min<u32>(44u * 1024u, (u32)(160u * 1024u - fa.sharedSizeBytes) & ~255u);
#endif
        HIPCHK(hipFuncSetAttribute((const void*)ks_pack_rr, hipFuncAttributeMaxDynamicSharedMemorySize, (int)room)); rr_attr[device] = room;
      }
      const u32 need = (u32)(((size_t)ds[0]->h.R * ds[0]->h.ge_max * 8 + 64 + 255) & ~(size_t)255);
      lds_rr = std::min(rr_attr[device], need);
    }
    hipLaunchKernelGGL(ks_pack_rr, dim3(1), dim3(64 * RR_NW), lds_rr, st, dp, dsv, lds_rr);
    u64 rr_err[8] = {0, 0, 0, 0, 0, 0, 0, 0};      // stats[7..14]: the error word ... the decline code
    HIPCHK(hipMemcpyAsync(rr_err, ds[0]->hs.stats + KS_STAT_ERR, sizeof rr_err, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    HIPCHK(hipGetLastError());
    rr_done = rr_err[0] != KS_RR_DECLINED;
    ds[0]->rr_started = 1; ds[0]->rr_code = rr_done ? 0 : (int)rr_err[14 - KS_STAT_ERR];
  }
#if defined(KS_SIM) && !defined(KS_SIM_PACK)
  if (!rr_done) return fail(KS_ERR_UNSUPPORTED, "emulator build: the problem is outside what ks_pack_rr covers (ks_pack is not emulated in this build)");
#else
  if (!rr_done) {
  typedef void (*pack_fn)(const DevProb*, const DevState*, u32);
  static const pack_fn variants[8] = {ks_pack<false, false, false, 1>, ks_pack<false, true, false, 1>, ks_pack<true, false, false, 1>, ks_pack<true, true, false, 1>,
                                      ks_pack<false, false, true, 1>, ks_pack<false, true, true, 1>, ks_pack<true, false, true, 1>, ks_pack<true, true, true, 1>};
  {   // the large dynamic-LDS opt-in is a per-device function attribute: set it once per device, race-free (two Solves may run concurrently)
    static std::mutex attr_mu; static std::vector<char> attr_done;
    std::lock_guard<std::mutex> g(attr_mu);
    if ((size_t)device >= attr_done.size()) attr_done.resize(device + 1, 0);
    if (!attr_done[device]) {
      for (int i = 0; i < 8; ++i) HIPCHK(hipFuncSetAttribute((const void*)variants[i], hipFuncAttributeMaxDynamicSharedMemorySize, 104 * 1024));
      HIPCHK(hipFuncSetAttribute((const void*)ks_pack<true, false, true, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 44 * 1024));
      HIPCHK(hipFuncSetAttribute((const void*)ks_pack<true, false, false, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 44 * 1024));
      HIPCHK(hipFuncSetAttribute((const void*)ks_pack<true, true, false, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 44 * 1024));
      attr_done[device] = 1;
    }
  }
  // A single Solve whose problem takes the LEAN, FAST, no-bounds kernel gets 8 waves: waves 1..7 join wave 0 for the
  // speculation rounds (see ks_pack).  T <= 8192 keeps a node's surviving-type mask in two registers per lane.
  bool multi = n == 1 && fast && ds[0]->h.TW <= 128 && !(ds[0]->h.flags & KS_FLAG_STATS) && !one_wave && !ds[0]->no_multi;
  if (multi) {
    const u32 lds_mw = 44u * 1024u;
    if ((size_t)ds[0]->h.R * ds[0]->h.ge_max * 8 + 8192 > lds_mw) multi = false;
    else {
      if (lean && !bounds) hipLaunchKernelGGL((ks_pack<true, false, true, 8>), dim3(1), dim3(512), lds_mw, st, dp, dsv, lds_mw);
      else if (bounds) hipLaunchKernelGGL((ks_pack<true, true, false, 4>), dim3(1), dim3(256), lds_mw, st, dp, dsv, lds_mw);
      else hipLaunchKernelGGL((ks_pack<true, false, false, 4>), dim3(1), dim3(256), lds_mw, st, dp, dsv, lds_mw);     // host ports / limits / selectors on hostname or instance type: the general code
                                                                                                                     // needs > 256 VGPRs, so 4 waves (one per SIMD): leader + 3 workers
    }
  }
  if (!multi) hipLaunchKernelGGL(variants[(lean ? 4 : 0) + (fast ? 2 : 0) + (bounds ? 1 : 0)], dim3(n), dim3(64), lds_bytes, st, dp, dsv, lds_bytes);
  }
#endif
  HIPCHK(hipEventRecord(e1, st));
  HIPCHK(hipStreamSynchronize(st));
  HIPCHK(hipGetLastError());
  if (kernel_ms) HIPCHK(hipEventElapsedTime(kernel_ms, e0, e1));
  hipEventDestroy(e0); hipEventDestroy(e1);
  int rc = KS_OK;
  if (!outs) {       // results stay on the device (ks_batch_records_dev / ks_price_filter_dev / ... read them there): only the error words come back
    if (n > 1) { std::vector<u64> meta((size_t)n * 34); HIPCHK(hipMemcpy(meta.data(), d_meta, meta.size() * sizeof(u64), hipMemcpyDeviceToHost)); for (u32 i = 0; i < n && rc == KS_OK; ++i) rc = ks_stats_error(&meta[(size_t)i * 34 + 2]); }
    else { u64 st[32]; HIPCHK(hipMemcpy(st, ds[0]->hs.stats, sizeof st, hipMemcpyDeviceToHost)); rc = ks_stats_error(st); }
    return rc;
  }
  if (n > 1) rc = download_batch(ds, n, dsv, d_meta, outs);
  else rc = download(ds[0], outs[0]);
  return rc;
}

// ------------------------------------------------------------------------------------------------
// Fixed-size result records of a batch, built ON the device into a caller-owned device buffer: what the ranks of a what-if fan-out exchange
// (one RCCL all-gather, no host hop).  Record i = [ids[i], n_new, n_unscheduled, InstanceTypeOptions of new node 0 (words x u64, zero if none)].
// ------------------------------------------------------------------------------------------------
struct RecordDesc { const u32* counts; const u64* alive; u64 id; u32 TW, pad; };
__global__ __launch_bounds__(64) void ks_records(const RecordDesc* descs, u64* out, u32 words) {
  const RecordDesc d = descs[blockIdx.x]; u64* row = out + (size_t)blockIdx.x * (3 + words);
  const u32 n_new = d.counts[0];
  if (threadIdx.x == 0) { row[0] = d.id; row[1] = n_new; row[2] = d.counts[1]; }
  for (u32 w = threadIdx.x; w < words; w += 64) row[3 + w] = (n_new && w < d.TW) ? d.alive[w] : 0ull;
}
extern "C" int ks_batch_records_dev(ks_dev_problem* const* ds, uint32_t n, const uint64_t* ids, uint32_t words, void* d_out) {
  if (!n) return KS_OK;
  if (!ds || !ids || !d_out) return fail(KS_ERR_INVALID, "null argument");
  const int device = ds[0]->device; HIPCHK(hipSetDevice(device));
  std::vector<RecordDesc> hd(n);
  for (u32 i = 0; i < n; ++i) {
    if (ds[i]->device != device) return fail(KS_ERR_INVALID, "batch spans devices");
    if (ds[i]->h.TW > words) return fail(KS_ERR_INVALID, "record row too short");
    hd[i] = RecordDesc{ds[i]->hs.out_counts, ds[i]->hs.n_alive, ids[i], ds[i]->h.TW, 0};
  }
  TmpDev t_desc(device); TRY(t_desc.alloc(n * sizeof(RecordDesc)));
  HIPCHK(hipMemcpyAsync(t_desc.p, hd.data(), n * sizeof(RecordDesc), hipMemcpyHostToDevice, ds[0]->stream));
  hipLaunchKernelGGL(ks_records, dim3(n), dim3(64), 0, ds[0]->stream, t_desc.as<RecordDesc>(), (u64*)d_out, words);
  HIPCHK(hipStreamSynchronize(ds[0]->stream)); HIPCHK(hipGetLastError());      // the buffer is complete when this returns: the caller's own stream may read it
  return KS_OK;
}

// ------------------------------------------------------------------------------------------------
// The what-if fan-out in the C ABI (SURVEY 8b `ks_solve_batch(shared, whatifs, n, out, ngpus)`, 8e row 1; deprovisioning/helpers.go:42-115,
// multinodeconsolidation.go:74-114): the caller's what-ifs are resident as SHARDS -- one list of device problems per GPU (ks_whatifs_open on that
// GPU's copy of the snapshot) -- and one call solves every shard in ONE batched launch on its own device and stream, builds the fixed-size decision
// records there, and gathers them into ONE host table ordered by id.  A thread per shard; no Python, no torch.  (bench.py's N-rank step does the same
// over processes, with one RCCL all-gather instead of the device-to-host copies: one process per GPU is the launch contract there.)
// ks_deal_lpt: which shard a what-if goes to -- longest predicted work first, each to the least loaded shard (ties: the lower index), so that a
// batch is not its longest what-if plus whatever i mod N happened to put beside it.
// ------------------------------------------------------------------------------------------------
extern "C" void ks_deal_lpt(const uint64_t* weight, uint32_t n, uint32_t nshards, uint32_t* shard_of) {
  if (!nshards) return;
  std::vector<u32> order(n); for (u32 i = 0; i < n; ++i) order[i] = i;
  std::stable_sort(order.begin(), order.end(), [&](u32 a, u32 b) { return weight[a] > weight[b]; });
  std::vector<u64> load(nshards, 0);
  for (u32 i : order) { u32 best = 0; for (u32 s = 1; s < nshards; ++s) if (load[s] < load[best]) best = s; shard_of[i] = best; load[best] += weight[i] ? weight[i] : 1; }
}
extern "C" int ks_solve_batch_sharded(ks_dev_problem* const* const* shards, const uint32_t* shard_n, const uint64_t* const* shard_ids, uint32_t nshards, uint32_t words,
                                      uint64_t* out_rows, float* kernel_ms_max) {
  if (!nshards) return KS_OK;
  if (!shards || !shard_n || !shard_ids || !out_rows) return fail(KS_ERR_INVALID, "null argument");
  const size_t width = 3 + (size_t)words;
  std::vector<size_t> off(nshards + 1, 0); for (u32 s = 0; s < nshards; ++s) off[s + 1] = off[s] + shard_n[s];
  std::vector<int> rcs(nshards, KS_OK); std::vector<std::string> msgs(nshards); std::vector<float> kms(nshards, 0.0f);
  std::vector<u64> rows(off[nshards] * width);
  auto run = [&](u32 s) {
    const u32 n = shard_n[s]; if (!n) return;
    float k = 0.0f;
    int rc = ks_solve_batch_dev(shards[s], n, nullptr, &k);
    if (rc == KS_OK) {
      const int device = shards[s][0]->device;
      TmpDev buf(device);
      rc = buf.alloc(n * width * sizeof(u64));
      if (rc == KS_OK) rc = ks_batch_records_dev(shards[s], n, shard_ids[s], words, buf.p);
      if (rc == KS_OK && hipMemcpy(rows.data() + off[s] * width, buf.p, n * width * sizeof(u64), hipMemcpyDeviceToHost) != hipSuccess) { rc = KS_ERR_DEVICE; g_err = "records: device to host"; }
    }
    rcs[s] = rc; if (rc != KS_OK) msgs[s] = g_err; kms[s] = k;
  };
  // shard 0 on the caller's thread -- whose current device is put back afterwards (a solve selects its shard's device) --, the others on threads of their own; a thread that
  // cannot be started is an error code, not an exception through the C boundary
  int caller_dev = -1; const bool have_dev = hipGetDevice(&caller_dev) == hipSuccess;
  std::vector<std::thread> th; bool spawn_failed = false;
  try { for (u32 s = 1; s < nshards; ++s) th.emplace_back(run, s); } catch (const std::system_error&) { spawn_failed = true; }
  if (!spawn_failed) run(0);
  for (auto& t : th) t.join();
  if (have_dev) (void)hipSetDevice(caller_dev);
  if (spawn_failed) return fail(KS_ERR_DEVICE, "a shard's thread could not be started");
  for (u32 s = 0; s < nshards; ++s) if (rcs[s] != KS_OK) return fail(rcs[s], "shard " + std::to_string(s) + ": " + msgs[s]);
  // one table, ordered by id (what the all-gather + sort of the N-rank path leaves)
  std::vector<size_t> idx(off[nshards]); for (size_t i = 0; i < idx.size(); ++i) idx[i] = i;
  std::sort(idx.begin(), idx.end(), [&](size_t a, size_t b) { return rows[a * width] < rows[b * width]; });
  for (size_t i = 0; i < idx.size(); ++i) memcpy(out_rows + i * width, rows.data() + idx[i] * width, width * sizeof(u64));
  if (kernel_ms_max) { float m = 0; for (float k : kms) m = std::max(m, k); *kernel_ms_max = m; }
  return KS_OK;
}

// ------------------------------------------------------------------------------------------------
// Consolidation price stage: filterByPrice over worstLaunchPrice (deprovisioning/helpers.go:148-157,
// :292-315) on a what-if's replacement node while its result is still on the device.  One block per problem,
// lane w owns word w of the node's InstanceTypeOptions; float64 compares only (prices are never added here).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void ks_price_filter(const DevProb* probs, const DevState* states, const u32* node, const double* max_price, const u32* spot_only, u64* const* outs, u32* counts) {
  const DevProb& P = probs[blockIdx.x]; const DevState& S = states[blockIdx.x];
  const u32 j = node[blockIdx.x]; const double maxp = max_price[blockIdx.x]; const int lane = threadIdx.x;
  const u32 pres = S.o_present[j], comp = S.o_complement[j];
  // reqs.Get(key): a missing key reads as Exists (requirements.go:114-120), which Has() every value
  const KReq zq = ((pres >> P.key_zone) & 1u) ? load_req(pres, comp, S.o_mask + (size_t)j * P.K, S.o_gt + (size_t)j * P.K, S.o_lt + (size_t)j * P.K, P.key_zone) : kreq_exists();
  const KReq cq = ((pres >> P.key_ct) & 1u) ? load_req(pres, comp, S.o_mask + (size_t)j * P.K, S.o_gt + (size_t)j * P.K, S.o_lt + (size_t)j * P.K, P.key_ct) : kreq_exists();
  const u64 allowZ = kreq_has_mask(zq, P.value_int + P.key_zone * 64, P.key_nvalues[P.key_zone]);
  u64 allowC = kreq_has_mask(cq, P.value_int + P.key_ct * 64, P.key_nvalues[P.key_ct]);
  if (spot_only && spot_only[blockIdx.x]) allowC &= P.ct_spot >= 0 ? (1ull << P.ct_spot) : 0ull;     // Requirements.Add(capacity-type In [spot])
  const u32 NP = P.key_nvalues[P.key_zone] * P.n_ct;
  const bool spot = P.ct_spot >= 0 && ((allowC >> P.ct_spot) & 1ull), od = P.ct_ondemand >= 0 && ((allowC >> P.ct_ondemand) & 1ull);
  u32 kept = 0;
  for (u32 wbase = 0; wbase < P.TW; wbase += 64) {
    const u32 w = wbase + lane; u64 out = 0;
    if (w < P.TW) {
      for (u64 bits = S.n_alive[(size_t)j * P.TW + w]; bits; bits &= bits - 1) {
        const u32 b = (u32)__builtin_ctzll(bits), t = w * 64 + b; const u64 offer = P.it_offer[t];
        double launch = 1.7976931348623157e308; bool got = false;      // math.MaxFloat64
        if (spot) {          // "we prefer to launch spot offerings": the worst (highest) spot price in the allowed zones
          double mx = 0.0;
          for (u64 zz = allowZ; zz; zz &= zz - 1) { const u32 pair = (u32)__builtin_ctzll(zz) * P.n_ct + (u32)P.ct_spot; if (pair < 64 && ((offer >> pair) & 1ull)) { const double pr = P.it_price[(size_t)t * NP + pair]; if (!got || pr > mx) mx = pr; got = true; } }
          if (got) launch = mx;
        }
        if (!got && od) {
          double mx = 0.0;
          for (u64 zz = allowZ; zz; zz &= zz - 1) { const u32 pair = (u32)__builtin_ctzll(zz) * P.n_ct + (u32)P.ct_ondemand; if (pair < 64 && ((offer >> pair) & 1ull)) { const double pr = P.it_price[(size_t)t * NP + pair]; if (!got || pr > mx) mx = pr; got = true; } }
          if (got) launch = mx;
        }
        if (launch < maxp) out |= 1ull << b;
      }
      outs[blockIdx.x][w] = out;
    }
    u32 pc = (u32)__builtin_popcountll(out); for (int off = 32; off > 0; off >>= 1) pc += __shfl_xor(pc, off);
    kept += pc;
  }
  if (lane == 0) counts[blockIdx.x] = kept;
}

extern "C" int ks_price_filter_dev(ks_dev_problem* const* ds, uint32_t n, const uint32_t* node, const double* max_price, const uint32_t* spot_only, uint64_t* const* out_types, uint32_t* out_counts) {
  if (!n) return KS_OK;
  if (!ds || !node || !max_price || !out_types || !out_counts) return fail(KS_ERR_INVALID, "null argument");
  const int device = ds[0]->device; HIPCHK(hipSetDevice(device));
  std::vector<DevProb> hp(n); std::vector<DevState> hs(n); size_t words = 0; std::vector<size_t> off(n);
  for (u32 i = 0; i < n; ++i) {
    if (ds[i]->device != device) return fail(KS_ERR_INVALID, "batch spans devices");
    if (!ds[i]->h.it_price || ds[i]->h.key_zone < 0 || ds[i]->h.key_ct < 0) return fail(KS_ERR_INVALID, "problem carries no offering prices");
    if (node[i] >= ds[i]->h.NMAX) return fail(KS_ERR_INVALID, "node index out of range");
    hp[i] = ds[i]->h; hs[i] = ds[i]->hs; off[i] = words; words += ds[i]->h.TW;
  }
  TmpDev t_dp(device), t_dsv(device), t_node(device), t_max(device), t_out(device), t_ptr(device), t_cnt(device), t_spot(device);
  u32* dspot = nullptr;
  if (spot_only) { TRY(t_spot.alloc(n * sizeof(u32))); dspot = t_spot.as<u32>(); HIPCHK(hipMemcpy(dspot, spot_only, n * sizeof(u32), hipMemcpyHostToDevice)); }
  TRY(t_dp.alloc(n * sizeof(DevProb))); TRY(t_dsv.alloc(n * sizeof(DevState))); TRY(t_node.alloc(n * sizeof(u32))); TRY(t_max.alloc(n * sizeof(double)));
  TRY(t_out.alloc((words ? words : 1) * sizeof(u64))); TRY(t_ptr.alloc(n * sizeof(u64*))); TRY(t_cnt.alloc(n * sizeof(u32)));
  DevProb* dp = t_dp.as<DevProb>(); DevState* dsv = t_dsv.as<DevState>(); u32* dnode = t_node.as<u32>(); double* dmax = t_max.as<double>(); u64* dout = t_out.as<u64>(); u64** dptr = t_ptr.as<u64*>(); u32* dcnt = t_cnt.as<u32>();
  std::vector<u64*> ptrs(n); for (u32 i = 0; i < n; ++i) ptrs[i] = dout + off[i];
  HIPCHK(hipMemcpy(dp, hp.data(), n * sizeof(DevProb), hipMemcpyHostToDevice)); HIPCHK(hipMemcpy(dsv, hs.data(), n * sizeof(DevState), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(dnode, node, n * sizeof(u32), hipMemcpyHostToDevice)); HIPCHK(hipMemcpy(dmax, max_price, n * sizeof(double), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(dptr, ptrs.data(), n * sizeof(u64*), hipMemcpyHostToDevice));
  hipLaunchKernelGGL(ks_price_filter, dim3(n), dim3(64), 0, ds[0]->stream, dp, dsv, dnode, dmax, (const u32*)dspot, dptr, dcnt);
  std::vector<u64> host(words ? words : 1);
  HIPCHK(hipMemcpyAsync(host.data(), dout, (words ? words : 1) * sizeof(u64), hipMemcpyDeviceToHost, ds[0]->stream));
  HIPCHK(hipMemcpyAsync(out_counts, dcnt, n * sizeof(u32), hipMemcpyDeviceToHost, ds[0]->stream));
  HIPCHK(hipStreamSynchronize(ds[0]->stream)); HIPCHK(hipGetLastError());
  for (u32 i = 0; i < n; ++i) memcpy(out_types[i], host.data() + off[i], ds[i]->h.TW * sizeof(u64));
  return KS_OK;
}

// ------------------------------------------------------------------------------------------------
// Launch-time pick (fake/cloudprovider.go:79-84) and instanceTypesAreSubset (helpers.go:118-122) on device-resident results.
// ks_launch_pick: one wave per problem; lane w walks word w (+64, ...) of the node's InstanceTypeOptions, keeps the type whose cheapest
// allowed available offering is cheapest, then the wave reduces (price, type index) lexicographically: the first cheapest in index order wins.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void ks_launch_pick(const DevProb* probs, const DevState* states, const u32* node, i32* out_type, i32* out_pair, double* out_price) {
  const DevProb& P = probs[blockIdx.x]; const DevState& S = states[blockIdx.x];
  const u32 j = node[blockIdx.x]; const int lane = threadIdx.x;
  const u32 pres = S.o_present[j], comp = S.o_complement[j];
  const KReq zq = ((pres >> P.key_zone) & 1u) ? load_req(pres, comp, S.o_mask + (size_t)j * P.K, S.o_gt + (size_t)j * P.K, S.o_lt + (size_t)j * P.K, P.key_zone) : kreq_exists();
  const KReq cq = ((pres >> P.key_ct) & 1u) ? load_req(pres, comp, S.o_mask + (size_t)j * P.K, S.o_gt + (size_t)j * P.K, S.o_lt + (size_t)j * P.K, P.key_ct) : kreq_exists();
  const u64 allowZ = kreq_has_mask(zq, P.value_int + P.key_zone * 64, P.key_nvalues[P.key_zone]);
  const u64 allowC = kreq_has_mask(cq, P.value_int + P.key_ct * 64, P.key_nvalues[P.key_ct]) & (P.n_ct >= 64 ? ~0ull : ((1ull << P.n_ct) - 1ull));
  const u32 NP = P.key_nvalues[P.key_zone] * P.n_ct;
  u64 pairs = 0;
  for (u64 zz = allowZ; zz; zz &= zz - 1) { const u32 z = (u32)__builtin_ctzll(zz); if (z * P.n_ct >= 64) break; pairs |= allowC << (z * P.n_ct); }
  double best = 1.7976931348623157e308; u32 bt = 0xFFFFFFFFu, bp = 0;
  for (u32 w = lane; w < P.TW; w += 64) {
    for (u64 bits = S.n_alive[(size_t)j * P.TW + w]; bits; bits &= bits - 1) {
      const u32 t = w * 64 + (u32)__builtin_ctzll(bits);
      for (u64 of = P.it_offer[t] & pairs; of; of &= of - 1) {          // Offerings.Available().Requirements(reqs)
        const u32 pr = (u32)__builtin_ctzll(of); const double c = P.it_price_lo[(size_t)t * NP + pr];
        if (c < best || (c == best && t < bt)) { best = c; bt = t; bp = pr; }
      }
    }
  }
  for (int off = 32; off > 0; off >>= 1) {                               // lexicographic (price, type) minimum over the wave
    const double ob = __shfl_xor(best, off); const u32 ot = (u32)__shfl_xor((int)bt, off), op = (u32)__shfl_xor((int)bp, off);
    if (ob < best || (ob == best && ot < bt)) { best = ob; bt = ot; bp = op; }
  }
  if (lane == 0) { out_type[blockIdx.x] = bt == 0xFFFFFFFFu ? -1 : (i32)bt; out_pair[blockIdx.x] = bt == 0xFFFFFFFFu ? -1 : (i32)bp; out_price[blockIdx.x] = best; }
}
__global__ __launch_bounds__(64) void ks_types_subset(const DevProb* probs, const DevState* states, const u32* node, const u64* lhs, u32 stride, u32* out) {
  const DevProb& P = probs[blockIdx.x]; const DevState& S = states[blockIdx.x]; const u32 j = node[blockIdx.x]; const int lane = threadIdx.x;
  bool bad = false;
  for (u32 w = lane; w < P.TW; w += 64) if (lhs[(size_t)blockIdx.x * stride + w] & ~S.n_alive[(size_t)j * P.TW + w]) bad = true;
  const u64 b = ballot64(bad);
  if (lane == 0) out[blockIdx.x] = b ? 0u : 1u;
}
static int batch_descriptors(ks_dev_problem* const* ds, u32 n, const u32* node, bool need_prices, TmpDev& t_dp, TmpDev& t_dsv, TmpDev& t_node) {
  if (!ds || !node) return fail(KS_ERR_INVALID, "null argument");
  const int device = ds[0]->device;
  std::vector<DevProb> hp(n); std::vector<DevState> hs(n);
  for (u32 i = 0; i < n; ++i) {
    if (ds[i]->device != device) return fail(KS_ERR_INVALID, "batch spans devices");
    if (need_prices && (!ds[i]->h.it_price || ds[i]->h.key_zone < 0 || ds[i]->h.key_ct < 0)) return fail(KS_ERR_INVALID, "problem carries no offering prices");
    if (node[i] >= ds[i]->h.NMAX) return fail(KS_ERR_INVALID, "node index out of range");
    hp[i] = ds[i]->h; hs[i] = ds[i]->hs;
  }
  TRY(t_dp.alloc(n * sizeof(DevProb))); TRY(t_dsv.alloc(n * sizeof(DevState))); TRY(t_node.alloc(n * sizeof(u32)));
  HIPCHK(hipMemcpy(t_dp.p, hp.data(), n * sizeof(DevProb), hipMemcpyHostToDevice)); HIPCHK(hipMemcpy(t_dsv.p, hs.data(), n * sizeof(DevState), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(t_node.p, node, n * sizeof(u32), hipMemcpyHostToDevice));
  return KS_OK;
}
extern "C" int ks_launch_pick_dev(ks_dev_problem* const* ds, uint32_t n, const uint32_t* node, int32_t* out_type, int32_t* out_pair, double* out_price) {
  if (!n) return KS_OK;
  if (!ds || !out_type || !out_pair || !out_price) return fail(KS_ERR_INVALID, "null argument");
  const int device = ds[0]->device; HIPCHK(hipSetDevice(device));
  TmpDev t_dp(device), t_dsv(device), t_node(device), t_type(device), t_pair(device), t_price(device);
  TRY(batch_descriptors(ds, n, node, true, t_dp, t_dsv, t_node));
  TRY(t_type.alloc(n * sizeof(i32))); TRY(t_pair.alloc(n * sizeof(i32))); TRY(t_price.alloc(n * sizeof(double)));
  hipLaunchKernelGGL(ks_launch_pick, dim3(n), dim3(64), 0, ds[0]->stream, t_dp.as<DevProb>(), t_dsv.as<DevState>(), t_node.as<u32>(), t_type.as<i32>(), t_pair.as<i32>(), t_price.as<double>());
  HIPCHK(hipMemcpyAsync(out_type, t_type.p, n * sizeof(i32), hipMemcpyDeviceToHost, ds[0]->stream)); HIPCHK(hipMemcpyAsync(out_pair, t_pair.p, n * sizeof(i32), hipMemcpyDeviceToHost, ds[0]->stream));
  HIPCHK(hipMemcpyAsync(out_price, t_price.p, n * sizeof(double), hipMemcpyDeviceToHost, ds[0]->stream));
  HIPCHK(hipStreamSynchronize(ds[0]->stream)); HIPCHK(hipGetLastError());
  return KS_OK;
}
extern "C" int ks_types_subset_dev(ks_dev_problem* const* ds, uint32_t n, const uint32_t* node, const uint64_t* lhs, uint32_t stride_words, uint32_t* out) {
  if (!n) return KS_OK;
  if (!ds || !lhs || !out) return fail(KS_ERR_INVALID, "null argument");
  const int device = ds[0]->device; HIPCHK(hipSetDevice(device));
  for (u32 i = 0; i < n; ++i) if (ds[i]->h.TW > stride_words) return fail(KS_ERR_INVALID, "mask row too short");
  TmpDev t_dp(device), t_dsv(device), t_node(device), t_lhs(device), t_out(device);
  TRY(batch_descriptors(ds, n, node, false, t_dp, t_dsv, t_node));
  TRY(t_lhs.alloc((size_t)n * stride_words * sizeof(u64))); TRY(t_out.alloc(n * sizeof(u32)));
  HIPCHK(hipMemcpy(t_lhs.p, lhs, (size_t)n * stride_words * sizeof(u64), hipMemcpyHostToDevice));
  hipLaunchKernelGGL(ks_types_subset, dim3(n), dim3(64), 0, ds[0]->stream, t_dp.as<DevProb>(), t_dsv.as<DevState>(), t_node.as<u32>(), t_lhs.as<u64>(), stride_words, t_out.as<u32>());
  HIPCHK(hipMemcpyAsync(out, t_out.p, n * sizeof(u32), hipMemcpyDeviceToHost, ds[0]->stream));
  HIPCHK(hipStreamSynchronize(ds[0]->stream)); HIPCHK(hipGetLastError());
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

extern "C" int ks_probe_has(const ks_req1* a, const int32_t* value_int, uint32_t nvalues, int on_device, ks_req_facts* out) {
  if (!a || !out || !value_int) return fail(KS_ERR_INVALID, "null argument");
  if (!on_device) { ks_req_facts_of(to_k(a), value_int, nvalues, out); return KS_OK; }
  if (ks_device_count() <= 0) return fail(KS_ERR_DEVICE, "no gfx950 device");
  int device = ks_current_device(); TmpDev tv(device), to(device);
  TRY(tv.alloc(64 * sizeof(i32))); TRY(to.alloc(sizeof(ks_req_facts)));
  i32 tmp[64]; for (int i = 0; i < 64; ++i) tmp[i] = (u32)i < nvalues ? value_int[i] : INT32_MIN;
  HIPCHK(hipMemcpy(tv.p, tmp, sizeof tmp, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(ks_probe_has_kernel, dim3(1), dim3(1), 0, 0, *a, tv.as<i32>(), nvalues, to.as<ks_req_facts>());
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(out, to.p, sizeof(ks_req_facts), hipMemcpyDeviceToHost));
  return KS_OK;
}

// Diagnostics: the class briefs (evaluation-class ids, conflict masks) and plan records as the kernels see them.
extern "C" int ks_debug_classes(ks_dev_problem* d, void* briefs_out, void* plans_out) {
  if (!d) return fail(KS_ERR_INVALID, "null device problem");
  TRY(build_static(d, nullptr));
  if (briefs_out) HIPCHK(hipMemcpy(briefs_out, d->h.briefs, (size_t)d->h.C * sizeof(ClsBrief), hipMemcpyDeviceToHost));
  if (plans_out) HIPCHK(hipMemcpy(plans_out, d->h.plans, (size_t)d->h.C * sizeof(ClsPlan), hipMemcpyDeviceToHost));
  return (int)sizeof(ClsPlan);
}

extern "C" const char* ks_last_error(void) { return g_err.c_str(); }
extern "C" const char* ks_version(void) { return "ksolve 0.1.0 (gfx950)"; }
