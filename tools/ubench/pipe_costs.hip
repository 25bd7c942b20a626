// What a two-wave pipeline through LDS costs on gfx950 (round 6): the figures behind the head window's split into a preparing wave and a placing wave.
//   build: /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w -o tools/ubench/pipe_costs tools/ubench/pipe_costs.hip      run on the GPU box: tools/ubench/pipe_costs
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned long long u64; typedef unsigned int u32;
#define N 2048
#define LD(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
#define ST(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
#define CB() asm volatile("" ::: "memory")

// one wave alone: LDS atomics / stores of all 64 lanes to ONE address against one address per lane; two dependent chains interleaved against one
__global__ __launch_bounds__(64) void k1(u64* out) {
  __shared__ u32 a[256]; __shared__ u64 b[128];
  const int lane = threadIdx.x; int slot = 0; u64 t0, t1;
  for (int i = lane; i < 256; i += 64) a[i] = 0; if (lane < 128) b[lane] = 0; b[lane + 64] = 0;
  __syncthreads();
  // 0: atomicAdd (no return), all lanes one address, then a dependent read
  u32 x = 0;
  t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int i = 0; i < N; ++i) { atomicAdd(&a[0], 1u); CB(); x += a[1 + (x & 1)]; }
  t1 = __builtin_readcyclecounter(); if (lane == 0) out[slot] = t1 - t0; slot++;
  // 1: atomicAdd, one address per lane
  t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int i = 0; i < N; ++i) { atomicAdd(&a[64 + lane], 1u); CB(); x += a[1 + (x & 1)]; }
  t1 = __builtin_readcyclecounter(); if (lane == 0) out[slot] = t1 - t0; slot++;
  // 2: the head window's narrow record as it is: four atomics (max, add on a u32; two ors on u64), 63 lanes to one sink
  t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int i = 0; i < N; ++i) {
    const bool mine = lane == (i & 63);
    int* cp = mine ? (int*)&a[8 + (i & 3)] : (int*)&a[200]; u64* rp = mine ? &b[4] : &b[100]; u64* pp = mine ? &b[5] : &b[101];
    atomicMax(cp, 0); atomicAdd(cp, 1); atomicOr(rp, 1ull << (i & 3)); atomicOr(pp, 1ull << (i & 3)); CB(); x += a[1 + (x & 1)]; }
  t1 = __builtin_readcyclecounter(); if (lane == 0) out[slot] = t1 - t0; slot++;
  // 3: the same with a sink per lane
  t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int i = 0; i < N; ++i) {
    const bool mine = lane == (i & 63);
    int* cp = mine ? (int*)&a[8 + (i & 3)] : (int*)&a[128 + lane]; u64* rp = mine ? &b[4] : &b[64 + lane]; u64* pp = mine ? &b[5] : &b[64 + lane];
    atomicMax(cp, 0); atomicAdd(cp, 1); atomicOr(rp, 1ull << (i & 3)); atomicOr(pp, 1ull << (i & 3)); CB(); x += a[1 + (x & 1)]; }
  t1 = __builtin_readcyclecounter(); if (lane == 0) out[slot] = t1 - t0; slot++;
  // 4: plain stores, all lanes one address, then a dependent read
  t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int i = 0; i < N; ++i) { a[3] = x + i; CB(); x += a[1 + (x & 1)]; }
  t1 = __builtin_readcyclecounter(); if (lane == 0) out[slot] = t1 - t0; slot++;
  // 5: plain stores, one address per lane
  t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int i = 0; i < N; ++i) { a[64 + lane] = x + i; CB(); x += a[1 + (x & 1)]; }
  t1 = __builtin_readcyclecounter(); if (lane == 0) out[slot] = t1 - t0; slot++;
  // 6: ONE dependent chain of 64-bit compare + select + subtract (the commit's arithmetic) ...
  long long r0 = lane * 1000 + 100000, r1 = r0 + 7, q = 3 + lane;
  t0 = __builtin_readcyclecounter();
#pragma unroll 8
  for (int i = 0; i < N; ++i) { const bool f = q > r0; r0 -= f ? 1 : q; q ^= (r0 & 1); }
  t1 = __builtin_readcyclecounter(); if (lane == 0) out[slot] = t1 - t0; slot++;
  // 7: ... and TWO independent ones interleaved (does a lone wave overlap independent work?)
  long long q2 = 5 + lane;
  t0 = __builtin_readcyclecounter();
#pragma unroll 8
  for (int i = 0; i < N; ++i) { const bool f = q > r0; r0 -= f ? 1 : q; q ^= (r0 & 1); const bool g = q2 > r1; r1 -= g ? 1 : q2; q2 ^= (r1 & 1); }
  t1 = __builtin_readcyclecounter(); if (lane == 0) out[slot] = t1 - t0; slot++;
  // 8: the same arithmetic on 32 bits, one chain
  int s0 = lane * 1000 + 100000, p = 3 + lane;
  t0 = __builtin_readcyclecounter();
#pragma unroll 8
  for (int i = 0; i < N; ++i) { const bool f = p > s0; s0 -= f ? 1 : p; p ^= (s0 & 1); }
  t1 = __builtin_readcyclecounter(); if (lane == 0) out[slot] = t1 - t0; slot++;
  if (lane == 0) out[15] = x + (u32)r0 + (u32)r1 + (u32)q + (u32)q2 + (u32)s0 + (u32)p;
}

// two (of eight) waves: a flag ping-pong through LDS (one hop = half a round trip), polling with s_sleep SLP
template <int SLP, int WB>
__global__ __launch_bounds__(512) void k2(u64* out, int o) {
  __shared__ u32 f[64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (threadIdx.x < 64) f[threadIdx.x] = 0;
  __syncthreads();
  u64 t0 = __builtin_readcyclecounter();
  if (wv == 0) {
#pragma unroll 1
    for (u32 i = 1; i <= N; ++i) { ST(&f[0], i); while (LD(&f[32]) != i) { if (SLP >= 0) __builtin_amdgcn_s_sleep(SLP < 0 ? 0 : SLP); } }
  } else if (wv == WB) {
#pragma unroll 1
    for (u32 i = 1; i <= N; ++i) { while (LD(&f[0]) != i) { if (SLP >= 0) __builtin_amdgcn_s_sleep(SLP < 0 ? 0 : SLP); } ST(&f[32], i); }
  }
  u64 t1 = __builtin_readcyclecounter(); if (threadIdx.x == 0) out[o] = t1 - t0;
  __syncthreads();
}

// a producer wave and a consumer wave over a ring of 8 records (64 lanes x 2 words + a tag written last): the producer's work per record is WA dependent
// operations, the consumer's WBK; the consumer publishes its position, the producer stays at most 4 ahead.  Cycles per record at the consumer.
template <int WA, int WBK>
__global__ __launch_bounds__(512) void k3(u64* out, int o) {
  __shared__ u32 ring[8][132]; __shared__ u32 pos;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (threadIdx.x < 8) ring[threadIdx.x][128] = 0xFFFFFFFFu; if (threadIdx.x == 0) pos = 0;
  __syncthreads();
  u64 t0 = __builtin_readcyclecounter(); u32 acc = lane;
  if (wv == 1) {
#pragma unroll 1
    for (u32 i = 0; i < N; ++i) {
      while (i >= LD(&pos) + 4u) __builtin_amdgcn_s_sleep(1);
      u32 v = acc;
#pragma unroll
      for (int j = 0; j < WA; ++j) v = v * 3u + (v >> 5);
      acc = v; ring[i & 7][lane] = v; ring[i & 7][64 + lane] = v ^ i; CB(); ST(&ring[i & 7][128], i);
    }
  } else if (wv == 0) {
#pragma unroll 1
    for (u32 i = 0; i < N; ++i) {
      ST(&pos, i);
      while (LD(&ring[i & 7][128]) != i) { }
      CB(); u32 v = ring[i & 7][lane] + ring[i & 7][64 + lane] + acc;
#pragma unroll
      for (int j = 0; j < WBK; ++j) v = v * 3u + (v >> 5);
      acc = v;
    }
  }
  u64 t1 = __builtin_readcyclecounter(); if (threadIdx.x == 0) { out[o] = t1 - t0; out[15] = acc; }
  __syncthreads();
}

int main() {
  u64* out; hipMalloc(&out, 32 * 8); u64 o[32];
  for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k1, dim3(1), dim3(64), 0, 0, out); hipDeviceSynchronize(); }
  hipMemcpy(o, out, sizeof o, hipMemcpyDeviceToHost);
  const char* n1[] = {"atomicAdd 64 lanes -> 1 address + dep read", "atomicAdd 1 address per lane + dep read", "narrow record: 4 atomics, 63 lanes -> one sink", "narrow record: 4 atomics, a sink per lane",
                      "store 64 lanes -> 1 address + dep read", "store 1 address per lane + dep read", "i64 cmp+select+sub chain", "two independent i64 chains interleaved", "i32 cmp+select+sub chain"};
  for (int i = 0; i < 9; ++i) printf("%-48s per-iter %8.1f cycles\n", n1[i], (double)o[i] / N);
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL((k2<-1, 1>), dim3(1), dim3(512), 0, 0, out, 0); hipLaunchKernelGGL((k2<0, 1>), dim3(1), dim3(512), 0, 0, out, 1); hipLaunchKernelGGL((k2<1, 1>), dim3(1), dim3(512), 0, 0, out, 2);
    hipLaunchKernelGGL((k2<4, 1>), dim3(1), dim3(512), 0, 0, out, 3); hipLaunchKernelGGL((k2<-1, 4>), dim3(1), dim3(512), 0, 0, out, 4); hipLaunchKernelGGL((k2<1, 4>), dim3(1), dim3(512), 0, 0, out, 5);
    hipDeviceSynchronize(); }
  hipMemcpy(o, out, sizeof o, hipMemcpyDeviceToHost);
  const char* n2[] = {"ping-pong waves 0,1 busy poll", "ping-pong waves 0,1 s_sleep 0", "ping-pong waves 0,1 s_sleep 1", "ping-pong waves 0,1 s_sleep 4", "ping-pong waves 0,4 (same SIMD?) busy poll", "ping-pong waves 0,4 s_sleep 1"};
  for (int i = 0; i < 6; ++i) printf("%-48s per HOP  %8.1f cycles\n", n2[i], (double)o[i] / N / 2);
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL((k3<0, 0>), dim3(1), dim3(512), 0, 0, out, 0); hipLaunchKernelGGL((k3<64, 0>), dim3(1), dim3(512), 0, 0, out, 1); hipLaunchKernelGGL((k3<0, 64>), dim3(1), dim3(512), 0, 0, out, 2);
    hipLaunchKernelGGL((k3<64, 64>), dim3(1), dim3(512), 0, 0, out, 3); hipLaunchKernelGGL((k3<128, 64>), dim3(1), dim3(512), 0, 0, out, 4); hipLaunchKernelGGL((k3<32, 64>), dim3(1), dim3(512), 0, 0, out, 5);
    hipDeviceSynchronize(); }
  hipMemcpy(o, out, sizeof o, hipMemcpyDeviceToHost);
  const char* n3[] = {"ring: producer 0 ops, consumer 0 ops", "ring: producer 64 ops (2 each), consumer 0", "ring: producer 0, consumer 64", "ring: producer 64, consumer 64", "ring: producer 128, consumer 64", "ring: producer 32, consumer 64"};
  for (int i = 0; i < 6; ++i) printf("%-48s per record %6.1f cycles\n", n3[i], (double)o[i] / N);
  return 0;
}
