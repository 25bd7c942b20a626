// Single-wave cost model on gfx950: cycles per primitive when ONE wave64 runs alone on a CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned long long u64; typedef unsigned int u32;
#define N 4096
__global__ void k(u32* g, u64* out, int n_nodes) {
  __shared__ u32 lds[4096];
  const int lane = threadIdx.x;
  for (int i = lane; i < 4096; i += 64) lds[i] = (i * 7 + 1) & 4095;
  __syncthreads();
  u64 t0, t1; u32 x = lane + 1; int slot = 0;
  // 0: empty timer pair
  t0 = __builtin_readcyclecounter(); t1 = __builtin_readcyclecounter(); if (lane == 0) out[slot] = t1 - t0; slot++;
  // 1: dependent VALU chain (v_mad)
  t0 = __builtin_readcyclecounter();
#pragma unroll 64
  for (int i = 0; i < N; ++i) x = x * 3u + 1u;
  t1 = __builtin_readcyclecounter(); if (lane == 0) out[slot] = t1 - t0; slot++;
  // 2: dependent 64-bit add chain
  u64 y = x;
  t0 = __builtin_readcyclecounter();
#pragma unroll 64
  for (int i = 0; i < N; ++i) y = y + (y >> 3) + 1;
  t1 = __builtin_readcyclecounter(); if (lane == 0) out[slot] = t1 - t0; slot++;
  // 3: taken branches (loop not unrolled, tiny body)
  t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int i = 0; i < N; ++i) { x += 1; asm volatile("" ::: "memory"); }
  t1 = __builtin_readcyclecounter(); if (lane == 0) out[slot] = t1 - t0; slot++;
  // 4: dependent LDS reads (pointer chase)
  u32 p = lane;
  t0 = __builtin_readcyclecounter();
#pragma unroll 16
  for (int i = 0; i < N; ++i) p = lds[p];
  t1 = __builtin_readcyclecounter(); if (lane == 0) out[slot] = t1 - t0; slot++;
  // 5: dependent global loads, L2/L1-resident pointer chase over n_nodes*64 dwords
  u32 q = lane;
  t0 = __builtin_readcyclecounter();
#pragma unroll 8
  for (int i = 0; i < 1024; ++i) q = g[q] & (n_nodes * 64 - 1);
  t1 = __builtin_readcyclecounter(); if (lane == 0) out[slot] = t1 - t0; slot++;
  // 6: ballot + ctz + readlane chain
  u32 z = x;
  t0 = __builtin_readcyclecounter();
#pragma unroll 16
  for (int i = 0; i < N; ++i) { u64 m = __ballot((z >> (i & 7)) & 1); int w = m ? __builtin_ctzll(m) : 0; z += __builtin_amdgcn_readlane(z, w); }
  t1 = __builtin_readcyclecounter(); if (lane == 0) out[slot] = t1 - t0; slot++;
  // 7: divergent branch (half the lanes) with small bodies
  t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int i = 0; i < N; ++i) { if ((lane + i) & 1) x = x * 5 + 1; else x = x ^ 0x55; }
  t1 = __builtin_readcyclecounter(); if (lane == 0) out[slot] = t1 - t0; slot++;
  // 8: uniform branch skipping a block (s_cbranch) per iteration
  t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int i = 0; i < N; ++i) { if (__builtin_amdgcn_readfirstlane(x + i) & 1024) { x = x * 7 + 3; y += x; } x += 2; }
  t1 = __builtin_readcyclecounter(); if (lane == 0) out[slot] = t1 - t0; slot++;
  // 9: global store + __syncthreads (vmcnt(0)) round trip
  t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int i = 0; i < 256; ++i) { g[(lane + i * 64) & 4095] = x; __syncthreads(); }
  t1 = __builtin_readcyclecounter(); if (lane == 0) out[slot] = t1 - t0; slot++;
  // 10: independent global loads (8 in flight) then sum
  t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int i = 0; i < 256; ++i) { u32 s = 0; 
#pragma unroll
    for (int j = 0; j < 8; ++j) s += g[(q + j * 517 + i * 64) & (n_nodes * 64 - 1)]; q += s & 63; }
  t1 = __builtin_readcyclecounter(); if (lane == 0) out[slot] = t1 - t0; slot++;
  // 11: wave minimum (DPP reduce, 6 steps + readlane), dependent
  t0 = __builtin_readcyclecounter();
#pragma unroll 8
  for (int i = 0; i < N; ++i) {
    u32 v = x + lane;
    v = min(v, (u32)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x111, 0xf, 0xf, false));
    v = min(v, (u32)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x112, 0xf, 0xf, false));
    v = min(v, (u32)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x114, 0xf, 0xf, false));
    v = min(v, (u32)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x118, 0xf, 0xf, false));
    v = min(v, (u32)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x142, 0xa, 0xf, false));
    v = min(v, (u32)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x143, 0xc, 0xf, false));
    x += __builtin_amdgcn_readlane(v, 63);
  }
  t1 = __builtin_readcyclecounter(); if (lane == 0) out[slot] = t1 - t0; slot++;
  // 12: a branch on a VALU-made condition of a wave-uniform value (v_cmp -> s_cbranch_vccnz), taken half the time
  t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int i = 0; i < N; ++i) { const u32 u = __builtin_amdgcn_readfirstlane(x); if (((u + i) & 1u) == 0) { x = x * 3 + 1; } else { x = x ^ 0x1234u; } asm volatile("" ::: "memory"); }
  t1 = __builtin_readcyclecounter(); if (lane == 0) out[slot] = t1 - t0; slot++;
  // 13: LDS write then read of the same word (a round trip through LDS, as between two dependent steps)
  t0 = __builtin_readcyclecounter();
#pragma unroll 4
  for (int i = 0; i < N; ++i) { lds[lane] = x; asm volatile("" ::: "memory"); x = lds[(lane + 1) & 63] + 1; }
  t1 = __builtin_readcyclecounter(); if (lane == 0) out[slot] = t1 - t0; slot++;
  // 14: readlane with a lane index that was just computed (SGPR from VALU result)
  t0 = __builtin_readcyclecounter();
#pragma unroll 8
  for (int i = 0; i < N; ++i) { const int w = (int)(__builtin_amdgcn_readfirstlane(x) & 63u); x += __builtin_amdgcn_readlane(x, w); }
  t1 = __builtin_readcyclecounter(); if (lane == 0) out[slot] = t1 - t0; slot++;
  if (lane == 0) out[15] = x + y + p + q + z;
}
// More single-wave primitives: what the alternatives to wave-uniform control flow cost
typedef u32 u32x8 __attribute__((ext_vector_type(8)));
__global__ void k2(const u32* __restrict__ g, u64* out) {
  __shared__ u32 lds[256];
  const int lane = threadIdx.x;
  for (int i = lane; i < 256; i += 64) lds[i] = (i * 7 + 1) & 255;
  __syncthreads();
  u64 t0, t1; u32 x = lane + 1; int slot = 0;
  // 0: scalar compare + branch NOT taken (the body is skipped never: condition false every time)
  t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int i = 0; i < N; ++i) { if (__builtin_expect((i & 0x10000) != 0, 0)) { x = x * 7 + 3; } x += 2; asm volatile("" ::: "memory"); }
  t1 = __builtin_readcyclecounter(); if (lane == 0) out[slot] = t1 - t0; slot++;
  // 1: scalar compare + branch on the loop counter, alternately taken (no VGPR involved in the condition)
  t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int i = 0; i < N; ++i) { if (i & 1) { x = x * 7 + 3; } x += 2; asm volatile("" ::: "memory"); }
  t1 = __builtin_readcyclecounter(); if (lane == 0) out[slot] = t1 - t0; slot++;
  // 2: a ladder of eight `if (idx == k)` blocks with tiny bodies, idx scalar (the pack kernels' slot ladders)
  t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int i = 0; i < N; ++i) {
    const int idx = i & 7;
    if (idx == 0) x += 1; asm volatile("" ::: "memory"); if (idx == 1) x += 3; asm volatile("" ::: "memory"); if (idx == 2) x ^= 5; asm volatile("" ::: "memory"); if (idx == 3) x += 7; asm volatile("" ::: "memory");
    if (idx == 4) x ^= 9; asm volatile("" ::: "memory"); if (idx == 5) x += 11; asm volatile("" ::: "memory"); if (idx == 6) x ^= 13; asm volatile("" ::: "memory"); if (idx == 7) x += 15; asm volatile("" ::: "memory");
  }
  t1 = __builtin_readcyclecounter(); if (lane == 0) out[slot] = t1 - t0; slot++;
  // 3: the same selection as eight v_cndmask on scalar conditions (no branch)
  t0 = __builtin_readcyclecounter();
#pragma unroll 4
  for (int i = 0; i < N; ++i) {
    const int idx = i & 7; u32 d = 0;
    d = idx == 0 ? 1u : d; d = idx == 1 ? 3u : d; d = idx == 2 ? 5u : d; d = idx == 3 ? 7u : d; d = idx == 4 ? 9u : d; d = idx == 5 ? 11u : d; d = idx == 6 ? 13u : d; d = idx == 7 ? 15u : d;
    x += d + (x >> 7);
  }
  t1 = __builtin_readcyclecounter(); if (lane == 0) out[slot] = t1 - t0; slot++;
  // 4: a register vector indexed by a scalar (s_set_gpr_idx / v_movrel), read and written
  u32x8 vec = {x, x + 1, x + 2, x + 3, x + 4, x + 5, x + 6, x + 7};
  t0 = __builtin_readcyclecounter();
#pragma unroll 4
  for (int i = 0; i < N; ++i) { const int idx = __builtin_amdgcn_readfirstlane((int)(i * 5)) & 7; x += vec[idx]; vec[(idx + 3) & 7] = x; }
  t1 = __builtin_readcyclecounter(); if (lane == 0) out[slot] = t1 - t0; slot++;
  // 5: a dependent chain of scalar loads from global memory (s_load_dword: the index never touches a VGPR)
  u32 sq = 0;
  t0 = __builtin_readcyclecounter();
#pragma unroll 4
  for (int i = 0; i < 1024; ++i) sq = g[__builtin_amdgcn_readfirstlane(sq) & 0xFFFFu];
  t1 = __builtin_readcyclecounter(); if (lane == 0) out[slot] = t1 - t0; slot++;
  // 6: an LDS word every lane reads (broadcast) made scalar by readfirstlane, dependent (how the kernels fetch a header word)
  u32 hp = 0;
  t0 = __builtin_readcyclecounter();
#pragma unroll 4
  for (int i = 0; i < N; ++i) hp = __builtin_amdgcn_readfirstlane(lds[hp & 255]);
  t1 = __builtin_readcyclecounter(); if (lane == 0) out[slot] = t1 - t0; slot++;
  // 7: sixteen INDEPENDENT readlanes of one register (a class brief's fields), then their sum
  t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int i = 0; i < 1024; ++i) { u32 s = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) s += __builtin_amdgcn_readlane(x, j * 3);
    x += s; }
  t1 = __builtin_readcyclecounter(); if (lane == 0) out[slot] = t1 - t0; slot++;
  if (lane == 0) out[15] = x + sq + hp + vec[3];
}
// Eight waves (the pack kernels' workgroup): what a barrier costs when everyone arrives together, and an LDS atomic minimum + barrier + read (one pick)
__global__ __launch_bounds__(512) void k8(u64* out) {
  __shared__ unsigned long long win[2];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  u32 x = threadIdx.x + 1;
  if (threadIdx.x < 2) win[threadIdx.x] = ~0ull;
  __syncthreads();
  u64 t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int i = 0; i < 1024; ++i) { x = x * 3 + 1; __syncthreads(); }
  u64 t1 = __builtin_readcyclecounter(); if (threadIdx.x == 0) out[0] = t1 - t0;
  t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int i = 0; i < 1024; ++i) {
    if (lane == 0) atomicMin(&win[i & 1], ((unsigned long long)(x + wv) << 32) | (u32)wv);
    __syncthreads();
    x += (u32)(win[i & 1] >> 32);
    if (threadIdx.x == 0) win[(i + 1) & 1] = ~0ull;
  }
  t1 = __builtin_readcyclecounter(); if (threadIdx.x == 0) { out[1] = t1 - t0; out[2] = x; }
}
int main() {
  const int n_nodes = 2048 * 8; u32* g; u64* out; std::vector<u32> h(n_nodes * 64);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (u32)((i * 2654435761u) % h.size());
  hipMalloc(&g, h.size() * 4); hipMalloc(&out, 16 * 8); hipMemcpy(g, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, g, out, n_nodes); hipDeviceSynchronize(); }
  u64 o[16]; hipMemcpy(o, out, sizeof o, hipMemcpyDeviceToHost);
  const char* names[] = {"timer pair", "dep v_mad x4096", "dep 64-bit add x4096", "taken branch loop x4096", "dep LDS read x4096", "dep global load (4MB chase) x1024",
                         "ballot+ctz+readlane x4096", "divergent if/else x4096", "uniform skip branch x4096", "store+syncthreads x256", "8 indep global loads x256"};
  const char* names2[] = {"wave min (6 DPP + readlane) x4096", "branch on v_cmp of uniform x4096", "LDS write->read round trip x4096", "readfirstlane->readlane x4096"};
  const int div[] = {1, 4096, 4096, 4096, 4096, 1024, 4096, 4096, 4096, 256, 256};
  for (int i = 0; i < 11; ++i) printf("%-36s total %8llu  per-iter %8.1f cycles\n", names[i], o[i], (double)o[i] / div[i]);
  for (int i = 0; i < 4; ++i) printf("%-36s total %8llu  per-iter %8.1f cycles\n", names2[i], o[11 + i], (double)o[11 + i] / 4096);
  for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k2, dim3(1), dim3(64), 0, 0, g, out); hipDeviceSynchronize(); }
  hipMemcpy(o, out, sizeof o, hipMemcpyDeviceToHost);
  const char* names3[] = {"scalar cmp+branch, not taken x4096", "scalar cmp+branch, alternating x4096", "ladder of 8 if(idx==k) x4096", "8 selects on scalar conds x4096", "vector[scalar idx] read+write x4096",
                          "dep scalar global load x1024", "LDS broadcast->readfirstlane x4096", "16 indep readlanes + sum x1024"};
  const int div3[] = {4096, 4096, 4096, 4096, 4096, 1024, 4096, 1024};
  for (int i = 0; i < 8; ++i) printf("%-36s total %8llu  per-iter %8.1f cycles\n", names3[i], o[i], (double)o[i] / div3[i]);
  for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k8, dim3(1), dim3(512), 0, 0, out); hipDeviceSynchronize(); }
  hipMemcpy(o, out, sizeof o, hipMemcpyDeviceToHost);
  printf("%-36s total %8llu  per-iter %8.1f cycles\n", "8 waves: barrier x1024", o[0], (double)o[0] / 1024);
  printf("%-36s total %8llu  per-iter %8.1f cycles\n", "8 waves: atomicMin+barrier+read x1024", o[1], (double)o[1] / 1024);
  return 0;
}
