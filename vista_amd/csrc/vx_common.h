// Shared host/device helpers of libvx (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/vx_abi.h"

void vx_set_error(const char* fmt, ...);

#define VX_HIP(call)                                                                   \
  do {                                                                                 \
    hipError_t e_ = (call);                                                            \
    if (e_ != hipSuccess) {                                                            \
      vx_set_error("%s:%d %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e_)); \
      return VX_ERR_HIP;                                                               \
    }                                                                                  \
  } while (0)

#define VX_REQUIRE(cond, ...)      \
  do {                             \
    if (!(cond)) {                 \
      vx_set_error(__VA_ARGS__);   \
      return VX_ERR_ARG;           \
    }                              \
  } while (0)

#define VX_LAUNCH_CHECK() VX_HIP(hipGetLastError())

static inline size_t vx_align(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }
static inline int64_t vx_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---- open-addressing hash table in HBM: 16-byte slots, linear probing, lock-free ------------
// key  : linear cell index (int64), VX_EMPTY when free
// a    : voxelizer: smallest point index seen (atomicMin), later the voxel id; rulebook: row
// b    : voxelizer: number of points in the cell (atomicAdd); rulebook: unused
struct __attribute__((aligned(16))) VxSlot {
  long long key;
  int a;
  int b;
};
#define VX_EMPTY (-1LL)

static inline int64_t vx_hash_capacity(int64_t n) {
  int64_t cap = 1024;
  while (cap < 2 * n) cap <<= 1;
  return cap;
}

#ifdef __HIPCC__
__device__ __forceinline__ unsigned long long vx_mix64(unsigned long long x) {
  x ^= x >> 33;
  x *= 0xff51afd7ed558ccdULL;
  x ^= x >> 33;
  x *= 0xc4ceb9fe1a85ec53ULL;
  x ^= x >> 33;
  return x;
}

// insert-or-find; returns the slot index
__device__ __forceinline__ long long vx_hash_insert(VxSlot* t, long long mask, long long key) {
  long long s = (long long)(vx_mix64((unsigned long long)key) & (unsigned long long)mask);
  for (;;) {
    unsigned long long prev = atomicCAS((unsigned long long*)&t[s].key,
                                        (unsigned long long)VX_EMPTY, (unsigned long long)key);
    if (prev == (unsigned long long)VX_EMPTY || prev == (unsigned long long)key) return s;
    s = (s + 1) & mask;
  }
}

// lookup in a finished table (no concurrent writers): returns slot.a or -1
__device__ __forceinline__ int vx_hash_find(const VxSlot* t, long long mask, long long key) {
  long long s = (long long)(vx_mix64((unsigned long long)key) & (unsigned long long)mask);
  for (;;) {
    // one 16-byte load per probe
    const int4 v = *reinterpret_cast<const int4*>(&t[s]);
    long long k = ((long long)(unsigned)v.y << 32) | (unsigned)v.x;
    if (k == key) return v.z;
    if (k == VX_EMPTY) return -1;
    s = (s + 1) & mask;
  }
}
#endif

// stream-ordered byte fill by a kernel (what hipMemsetAsync does, but safe under HIP graph capture: see vx_scan.hip)
int32_t vx_fill_async(void* d_ptr, int value, size_t bytes, hipStream_t stream);

// exclusive scan of int32 (device), total written to d_total (int64, optional)
size_t vx_scan_workspace_bytes(int64_t n);
int32_t vx_scan_exclusive_i32(const int32_t* d_in, int32_t* d_out, int64_t n, int64_t* d_total,
                              void* d_ws, size_t ws_bytes, hipStream_t stream);
