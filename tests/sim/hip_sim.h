// hip_sim.h -- TEST INFRASTRUCTURE ONLY.  A lane-fiber emulator that lets g++ compile karpenter_core_amd/csrc/ksolve.hip
// and run its kernels on the host, so that the register-resident pack kernel (ks_pack_rr) can be checked against the CPU
// oracle in the build container, where there is no GPU.  It is never part of the product: libksolve.so is built by hipcc
// from the same source without this header, and nothing under karpenter_core_amd/, bench.py or __graft_entry__.py loads a
// library built with it (tests/test_cabi.py checks that).
//
// Model: every HIP thread of a workgroup is a fibre with its own stack; fibres of one workgroup run cooperatively on one
// OS thread.  A cross-lane operation (ballot, shuffle, readlane, wave reductions, the wave-level LDS hand-off LSYNC) is a
// rendezvous of the 64 lanes of a wave; __syncthreads() is a rendezvous of the workgroup.  Collectives must therefore be
// reached in wave-uniform control flow by every lane that has not left the kernel -- which is also what the hardware
// kernels assume wherever they use them; a divergent collective shows up here as a reported deadlock instead of silent
// garbage.  Memory is sequentially consistent, so missing barriers between waves are NOT found by this emulator (the
// waves are interleaved in a seed-dependent order, KS_SIM_SEED, to shake some of them out).
#pragma once
#define KS_SIM 1
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <sys/mman.h>
#include <functional>
#include <mutex>
#include <vector>
#include <map>
#include <algorithm>

struct dim3 { unsigned x, y, z; dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {} };
struct KsSimIdx { unsigned x, y, z; };
extern KsSimIdx threadIdx, blockIdx, blockDim, gridDim;

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__ static
#define __HIP_MEMORY_SCOPE_WORKGROUP 0
#define __hip_atomic_fetch_add(p, v, o, s) ks_sim_fetch_add((p), (v))
#define __hip_atomic_store(p, v, o, s) (*(volatile __typeof__(*(p))*)(p) = (v))
#define __hip_atomic_load(p, o, s) (*(volatile const __typeof__(*(p))*)(p))
#define __builtin_readcyclecounter() ((unsigned long long)__builtin_ia32_rdtsc())
#define __builtin_amdgcn_readfirstlane(x) (x)
#define __threadfence_block() do { } while (0)
#define __threadfence() do { } while (0)

typedef int hipError_t;
typedef void* hipStream_t;
typedef void* hipEvent_t;
enum { hipSuccess = 0 };
enum { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3 };
enum { hipHostMallocDefault = 0, hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
struct hipDeviceProp_t { char gcnArchName[64]; };
static inline const char* hipGetErrorString(hipError_t) { return "sim"; }
static inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
static inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) { strcpy(p->gcnArchName, "gfx950:sim"); return hipSuccess; }
static inline hipError_t hipMalloc(void** p, size_t n) { return posix_memalign(p, 256, n ? n : 256) ? 1 : hipSuccess; }
template <class T> static inline hipError_t hipMalloc(T** p, size_t n) { return hipMalloc((void**)p, n); }
static inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
static inline hipError_t hipHostMalloc(void** p, size_t n, int) { return hipMalloc(p, n); }
static inline hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, int) { if (n) memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, int, hipStream_t) { if (n) memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { if (n) memset(d, v, n); return hipSuccess; }
static inline hipError_t hipStreamCreate(hipStream_t* s) { *s = (void*)1; return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = nullptr; return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.0f; return hipSuccess; }
#define hipFuncSetAttribute(...) hipSuccess

namespace ks_sim {
void launch(dim3 grid, dim3 block, const std::function<void()>& body);
// rendezvous of the calling lane's wave: deposits v, returns the wave's 64 deposited values and the mask of lanes that took part
const uint64_t* exchange(uint64_t v, uint64_t* present);
void block_barrier();
void yield();
int lane_id();
static inline uint64_t ballot(bool p) { uint64_t pm; const uint64_t* a = exchange(p ? 1 : 0, &pm); uint64_t m = 0; for (int i = 0; i < 64; ++i) if (((pm >> i) & 1) && a[i]) m |= 1ull << i; return m; }
static inline uint64_t shfl64(uint64_t v, int src) { uint64_t pm; const uint64_t* a = exchange(v, &pm); src &= 63; return ((pm >> src) & 1) ? a[src] : v; }
static inline uint64_t shfl_down64(uint64_t v, int d) { uint64_t pm; const int me = lane_id(); const uint64_t* a = exchange(v, &pm); const int s = me + d; return (s < 64 && ((pm >> s) & 1)) ? a[s] : v; }
static inline uint64_t shfl_xor64(uint64_t v, int x) { uint64_t pm; const int me = lane_id(); const uint64_t* a = exchange(v, &pm); const int s = me ^ x; return (s < 64 && ((pm >> s) & 1)) ? a[s] : v; }
static inline uint32_t wave_min_u32(uint32_t v) { uint64_t pm; const uint64_t* a = exchange(v, &pm); uint32_t m = 0xFFFFFFFFu; for (int i = 0; i < 64; ++i) if ((pm >> i) & 1) m = std::min(m, (uint32_t)a[i]); return m; }
static inline uint32_t wave_or_u32(uint32_t v) { uint64_t pm; const uint64_t* a = exchange(v, &pm); uint32_t m = 0; for (int i = 0; i < 64; ++i) if ((pm >> i) & 1) m |= (uint32_t)a[i]; return m; }
static inline int64_t wave_max_i64(int64_t v) { uint64_t pm; const uint64_t* a = exchange((uint64_t)v, &pm); int64_t m = INT64_MIN; for (int i = 0; i < 64; ++i) if ((pm >> i) & 1) m = std::max(m, (int64_t)a[i]); return m; }
}  // namespace ks_sim

#define __syncthreads() ks_sim::block_barrier()
#define __ballot(p) ks_sim::ballot((p))
template <class T> static inline T __shfl(T v, int src) { static_assert(sizeof(T) <= 8, "shfl"); uint64_t b = 0; memcpy(&b, &v, sizeof(T)); b = ks_sim::shfl64(b, src); T o; memcpy(&o, &b, sizeof(T)); return o; }
template <class T> static inline T __shfl_down(T v, int d) { uint64_t b = 0; memcpy(&b, &v, sizeof(T)); b = ks_sim::shfl_down64(b, d); T o; memcpy(&o, &b, sizeof(T)); return o; }
template <class T> static inline T __shfl_xor(T v, int x) { uint64_t b = 0; memcpy(&b, &v, sizeof(T)); b = ks_sim::shfl_xor64(b, x); T o; memcpy(&o, &b, sizeof(T)); return o; }
#define __builtin_amdgcn_readlane(v, l) ((int)ks_sim::shfl64((uint64_t)(uint32_t)(v), (l)))

template <class T, class V> static inline T ks_sim_fetch_add(T* p, V v) { T o = *p; *p = (T)(o + (T)v); return o; }
template <class T, class V> static inline T atomicAdd(T* p, V v) { T o = *p; *p = (T)(o + (T)v); return o; }
template <class T, class V> static inline T atomicSub(T* p, V v) { T o = *p; *p = (T)(o - (T)v); return o; }
template <class T, class V> static inline T atomicOr(T* p, V v) { T o = *p; *p = (T)(o | (T)v); return o; }
template <class T, class V> static inline T atomicAnd(T* p, V v) { T o = *p; *p = (T)(o & (T)v); return o; }
template <class T, class V> static inline T atomicMax(T* p, V v) { T o = *p; if ((T)v > o) *p = (T)v; return o; }
template <class T, class V> static inline T atomicMin(T* p, V v) { T o = *p; if ((T)v < o) *p = (T)v; return o; }
template <class T, class A, class B> static inline T atomicCAS(T* p, A cmp, B val) { T o = *p; if (o == (T)cmp) *p = (T)val; return o; }

static inline uint32_t min(uint32_t a, uint32_t b) { return a < b ? a : b; }
static inline uint32_t max(uint32_t a, uint32_t b) { return a > b ? a : b; }
static inline int32_t min(int32_t a, int32_t b) { return a < b ? a : b; }
static inline int32_t max(int32_t a, int32_t b) { return a > b ? a : b; }
static inline uint64_t min(uint64_t a, uint64_t b) { return a < b ? a : b; }
static inline uint64_t max(uint64_t a, uint64_t b) { return a > b ? a : b; }
static inline int64_t min(int64_t a, int64_t b) { return a < b ? a : b; }
static inline int64_t max(int64_t a, int64_t b) { return a > b ? a : b; }

#define hipLaunchKernelGGL(kern, grid, block, lds, stream, ...) ks_sim::launch(dim3(grid), dim3(block), [=]() { kern(__VA_ARGS__); })
