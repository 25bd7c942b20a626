// What would a pack loop cost that evaluates EVERY pod against EVERY node from scratch, straight-line, with no state carried from pod to pod?  
// A synthetic stand-in for that loop, shaped like ks_pack_rr's data: one workgroup of 8 waves, every lane holds 8 nodes in registers (key, requirement class + requested
// mask, four 64-bit headrooms, eight words of 8-bit hostname counters); per "pod": its parameters as scalars (s_load from a table in global memory, the accepting-class
// mask from LDS), all 8 slots evaluated (resources, class bit, two hostname items whose counter word is chosen by a select chain, not a switch), the lane's and the wave's
// least accepting key (one DPP reduction), publish + ONE barrier + an 8-lane minimum, and the commit by the lane that owns the winner (a ladder over the slot, taken by
// one wave).  Prints shader cycles per pod.  Not the product, not a test: a price tag for the design.       hipcc --offload-arch=gfx950 -O3 lean_loop.hip -o lean_loop
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned long long u64; typedef unsigned int u32; typedef long long i64; typedef int i32;
#ifdef R32
typedef int res_t;        /* requests normalised to 32 bits */
#else
typedef long long res_t;
#endif
#ifndef NW
#define NW 8        /* waves */
#endif
#ifndef NPL
#define NPL 8       /* slots evaluated per lane (the arrays always have 8) */
#endif
#define GW 8
struct PodRec { i64 rq[4]; u32 item[2]; u32 lim[2]; u32 cls; u32 pad[3]; };      // 64 bytes: one s_load_dwordx16

__device__ __forceinline__ u32 wave_min(u32 v) {
  v = min(v, (u32)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x111, 0xf, 0xf, false));
  v = min(v, (u32)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x112, 0xf, 0xf, false));
  v = min(v, (u32)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x114, 0xf, 0xf, false));
  v = min(v, (u32)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x118, 0xf, 0xf, false));
  v = min(v, (u32)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x142, 0xa, 0xf, false));
  v = min(v, (u32)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x143, 0xc, 0xf, false));
  return (u32)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ u32 min8(u32 v) {      // the minimum over lanes 0..7, in lane 7
  v = min(v, (u32)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x111, 0xf, 0xf, false));
  v = min(v, (u32)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x112, 0xf, 0xf, false));
  v = min(v, (u32)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x114, 0xf, 0xf, false));
  return (u32)__builtin_amdgcn_readlane((int)v, 7);
}

__global__ __launch_bounds__(64 * NW) void lean(const PodRec* __restrict__ pods, const u64* __restrict__ accept_of_class, u64* out, int n_pods, int n_nodes) {
  __shared__ u32 pub[2][NW];
  __shared__ u64 acc_lds[256];
  const int lane = threadIdx.x & 63; const u32 wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int i = threadIdx.x; i < 256; i += 64 * NW) acc_lds[i] = accept_of_class[i];
  // the nodes: node n -> slot n / 512, wave n % 8, lane (n % 512) / 8
  u32 key[8], meta[8], hc[8][GW]; res_t room[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const u32 n = (u32)i * 64u * NW + (u32)lane * NW + wv;
    const bool live = n < (u32)n_nodes && i < NPL;
    key[i] = live ? ((1u + (n & 3u)) << 22) | (n << 3) | (u32)i : 0xFFFFFFFFu;
    meta[i] = n % 23u;
#pragma unroll
    for (int r = 0; r < 4; ++r) room[i][r] = (res_t)(400000 + (i64)(n * 37u % 1000u) * (r + 1));
#pragma unroll
    for (int w = 0; w < GW; ++w) hc[i][w] = 0;
  }
#ifdef ITEM_MASKS
  // per hostname counter c = word * 4 + byte: which of this lane's slots hold a node whose counter c is >= 1 (z1) / >= 2 (z2): byte `byte` of z?[word], one bit per slot.
  // An item with limit 0 or 1 (every BASELINE workload) is then ONE select + shift per pod instead of a counter extraction per slot; the commit keeps the masks up to date.
  u32 z1[GW], z2[GW];
#pragma unroll
  for (int w = 0; w < GW; ++w) { z1[w] = 0; z2[w] = 0; }
#endif
  __syncthreads();
  u32 placed = 0; const u64 t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int p = 0; p < n_pods; ++p) {
    // ---- the pod's parameters, as scalars ----
#ifdef CONST_POD
    const PodRec& R = pods[2];      // (loop-invariant: what the per-pod scalar fetches cost is the difference)
#else
    const PodRec& R = pods[p & 1023];
#endif
    const res_t rq0 = (res_t)R.rq[0], rq1 = (res_t)R.rq[1], rq2 = (res_t)R.rq[2], rq3 = (res_t)R.rq[3];
    const u32 it0 = R.item[0], it1 = R.item[1], cls = R.cls; const i32 lim0 = (i32)R.lim[0], lim1 = (i32)R.lim[1];
    const u64 accv = acc_lds[cls & 255u];
    const u64 nacc = ~(((u64)(u32)__builtin_amdgcn_readfirstlane((int)(u32)accv)) | ((u64)(u32)__builtin_amdgcn_readfirstlane((int)(u32)(accv >> 32)) << 32));
    const u32 w0 = it0 & 7u, sh0 = (it0 >> 8) & 31u, w1 = it1 & 7u, sh1 = (it1 >> 8) & 31u; const bool has0 = it0 >> 31, has1 = it1 >> 31;
#ifdef ITEM_MASKS
    u32 rejm = 0;      // the slots the pod's hostname items refuse
    {
      u32 s1 = z1[0], s2 = z2[0], t1 = z1[0], t2 = z2[0];
#pragma unroll
      for (int w = 1; w < GW; ++w) { s1 = w0 == (u32)w ? z1[w] : s1; s2 = w0 == (u32)w ? z2[w] : s2; t1 = w1 == (u32)w ? z1[w] : t1; t2 = w1 == (u32)w ? z2[w] : t2; }
      const u32 m0 = ((lim0 <= 0 ? s1 : s2) >> (sh0 & 24u)) & 0xFFu, m1 = ((lim1 <= 0 ? t1 : t2) >> (sh1 & 24u)) & 0xFFu;      // (limits 0 / 1; larger ones would take the counter path)
      rejm = (has0 ? m0 : 0u) | (has1 ? m1 : 0u);
    }
#endif
    // ---- every slot, from scratch ----
    u32 best = 0xFFFFFFFFu;
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
      bool rj = ((nacc >> (meta[i] & 63u)) & 1ull) != 0;
      rj |= rq0 > room[i][0]; rj |= rq1 > room[i][1]; rj |= rq2 > room[i][2]; rj |= rq3 > room[i][3];
#ifdef ITEM_MASKS
      rj |= ((rejm >> i) & 1u) != 0;
#elif !defined(NO_ITEMS)
#ifdef ITEMS_BRANCH
      if (has0 || has1)      /* (wave-uniform: a pod without hostname items jumps over the item code) */
#endif
      {
#ifdef TREE
      const u32 a0 = (w0 & 1u) ? hc[i][1] : hc[i][0], a1 = (w0 & 1u) ? hc[i][3] : hc[i][2], a2 = (w0 & 1u) ? hc[i][5] : hc[i][4], a3 = (w0 & 1u) ? hc[i][7] : hc[i][6];
      const u32 b0 = (w0 & 2u) ? a1 : a0, b1 = (w0 & 2u) ? a3 : a2; const u32 c0 = (w0 & 4u) ? b1 : b0;
      const u32 d0 = (w1 & 1u) ? hc[i][1] : hc[i][0], d1 = (w1 & 1u) ? hc[i][3] : hc[i][2], d2 = (w1 & 1u) ? hc[i][5] : hc[i][4], d3 = (w1 & 1u) ? hc[i][7] : hc[i][6];
      const u32 e0 = (w1 & 2u) ? d1 : d0, e1 = (w1 & 2u) ? d3 : d2; const u32 c1 = (w1 & 4u) ? e1 : e0;
#else
      u32 c0 = hc[i][0], c1 = hc[i][0];
#pragma unroll
      for (int w = 1; w < GW; ++w) { c0 = w0 == (u32)w ? hc[i][w] : c0; c1 = w1 == (u32)w ? hc[i][w] : c1; }
#endif
      rj |= has0 && (i32)((c0 >> sh0) & 0xFFu) > lim0;
      rj |= has1 && (i32)((c1 >> sh1) & 0xFFu) > lim1;
      }
#endif
      best = min(best, rj ? 0xFFFFFFFFu : key[i]);
    }
    // ---- the wave's, then the workgroup's least key ----
    const u32 wk = wave_min(best);
#ifdef NO_XCHG
    const u32 pk = wk, win = wk; if (wv != 0) { placed += win & 1u; continue; }      /* (no exchange: wave 0 commits its own winner, the others only evaluate) */
#else
    if (lane == 0) pub[p & 1][wv] = wk;
    __syncthreads();
    const u32 pk = lane < NW ? pub[p & 1][lane] : 0xFFFFFFFFu;
    const u32 win = min8(pk);
#endif
    if (win == 0xFFFFFFFFu) continue;      // (nothing accepts: the leader's business in the real thing)
    // ---- commit, by the lane that holds the node; nothing is evaluated again ----
    const u32 oslot = win & 7u, owv = (u32)__builtin_ctzll(__ballot(pk == win));      // (keys are unique: bucket | place | slot)
    ++placed;
#ifdef NO_COMMIT
    placed += oslot + owv; continue;
#endif
#ifdef COMMIT_SELECT
    // every wave, every slot, under a data predicate: no ladder, no merge points at which the compiler copies the register arrays
    {
      const u32 olane = (u32)__builtin_ctzll(__ballot(best == wk));
      const bool own = owv == wv && (u32)lane == olane;
#pragma unroll
      for (int i = 0; i < NPL; ++i) {
        const bool m = own && oslot == (u32)i;
        room[i][0] -= m ? rq0 : (res_t)0; room[i][1] -= m ? rq1 : (res_t)0; room[i][2] -= m ? rq2 : (res_t)0; room[i][3] -= m ? rq3 : (res_t)0;
        key[i] = m ? (((key[i] >> 22) + 1u) << 22) | ((0x7FFFFu - (u32)p) << 3) | (key[i] & 7u) : key[i];
        meta[i] = m ? (meta[i] + cls) % 23u : meta[i];
#pragma unroll
        for (int w = 0; w < GW; ++w) {
          const bool hit = m && has0 && w0 == (u32)w;
#ifdef ITEM_MASKS
          const u32 cnt = (hc[i][w] >> sh0) & 0xFFu;
          z1[w] |= hit ? (1u << i) << (sh0 & 24u) : 0u; z2[w] |= hit && cnt >= 1u ? (1u << i) << (sh0 & 24u) : 0u;
#endif
          hc[i][w] += hit ? (1u << sh0) : 0u;
        }
      }
      continue;
    }
#endif
    if (owv != wv) continue;
    const u32 olane = (u32)__builtin_ctzll(__ballot(best == wk));
    const bool mine = (u32)lane == olane;
#ifdef ITEM_MASKS
#define MASKS_UPDATE(i, w) { const u32 cnt = (hc[i][w] >> sh0) & 0xFFu; z1[w] |= hit ? (1u << (i)) << (sh0 & 24u) : 0u; z2[w] |= hit && cnt >= 1u ? (1u << (i)) << (sh0 & 24u) : 0u; }
#else
#define MASKS_UPDATE(i, w)
#endif
#define COMMIT(i) if (oslot == (i)) { if (mine) { room[i][0] -= rq0; room[i][1] -= rq1; room[i][2] -= rq2; room[i][3] -= rq3; key[i] = (((key[i] >> 22) + 1u) << 22) | ((0x7FFFFu - (u32)p) << 3) | (key[i] & 7u); meta[i] = (meta[i] + cls) % 23u; } \
      _Pragma("unroll") for (int w = 0; w < GW; ++w) { const bool hit = mine && has0 && w0 == (u32)w; MASKS_UPDATE(i, w) hc[i][w] += hit ? (1u << sh0) : 0u; } }
    COMMIT(0) COMMIT(1)
#if NPL > 2
    COMMIT(2) COMMIT(3)
#endif
#if NPL > 4
    COMMIT(4) COMMIT(5) COMMIT(6) COMMIT(7)
#endif
  }
  const u64 t1 = __builtin_readcyclecounter();
  u32 sink = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) sink += key[i] + meta[i] + (u32)room[i][0] + hc[i][3];
#ifdef ITEM_MASKS
  for (int w = 0; w < GW; ++w) sink += z1[w] ^ z2[w];
#endif
  if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = placed; }
  if (sink == 0x12345678u) out[2] = sink;
}

#ifndef VARIANT
#define VARIANT ""
#endif
int main() {
  const int n_pods = 20000, n_nodes = 64 * NW * NPL < 2100 ? 64 * NW * NPL : 2100;
  std::vector<PodRec> h(1024); std::vector<u64> acc(256);
  for (int i = 0; i < 1024; ++i) {
    PodRec& r = h[i]; u32 s = i * 2654435761u;
    for (int k = 0; k < 4; ++k) r.rq[k] = 100 + (s >> (4 * k) & 255);
    const bool host = (i % 7) == 2 || (i % 7) == 3;
    r.item[0] = host ? 0x80000000u | (((s >> 9) & 3u) * 8u) << 8 | ((s >> 5) & 7u) : 0u; r.item[1] = 0; r.lim[0] = 1; r.lim[1] = 0; r.cls = s >> 24;
  }
  for (int i = 0; i < 256; ++i) acc[i] = ~0ull ^ (1ull << (i % 23));
  PodRec* dp; u64 *dacc, *dout;
  hipMalloc(&dp, h.size() * sizeof(PodRec)); hipMalloc(&dacc, 256 * 8); hipMalloc(&dout, 64);
  hipMemcpy(dp, h.data(), h.size() * sizeof(PodRec), hipMemcpyHostToDevice); hipMemcpy(dacc, acc.data(), 256 * 8, hipMemcpyHostToDevice);
  u64 o[3] = {0, 0, 0};
  for (int rep = 0; rep < 3; ++rep) { hipLaunchKernelGGL(lean, dim3(1), dim3(64 * NW), 0, 0, dp, dacc, dout, n_pods, n_nodes); hipDeviceSynchronize(); hipMemcpy(o, dout, 24, hipMemcpyDeviceToHost);
    printf("lean loop [%d waves, %d slots%s" VARIANT "]: %d pods over %d nodes in registers: %llu cycles, %.0f cycles per pod (%llu placed)\n", NW, NPL,
#ifdef NO_ITEMS
 ", no hostname items",
#else
 "",
#endif
 n_pods, n_nodes, o[0], (double)o[0] / n_pods, o[1]); }
  return 0;
}
