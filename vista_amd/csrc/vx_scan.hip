// Device-wide exclusive prefix sum of int32 (three-phase: block sums -> recursive scan -> apply).
// HBM-bound: reads n*4 twice, writes n*4 once. Wave64 shuffles for the intra-wave step.
#include <stdarg.h>

#include "vx_common.h"

static thread_local char g_err[512] = "";
void vx_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* vx_last_error(void) { return g_err; }
extern "C" int32_t vx_version(void) { return 1; }

namespace {
constexpr int kThreads = 256;
constexpr int kItems = 8;
constexpr int kTile = kThreads * kItems;  // 2048

__device__ __forceinline__ int wave_inclusive_scan(int v) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    int t = __shfl_up(v, d, 64);
    if (lane >= d) v += t;
  }
  return v;
}

// returns the exclusive prefix of `v` over the block and the block total
__device__ __forceinline__ int block_exclusive_scan(int v, int* total) {
  __shared__ int wsum[kThreads / 64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int inc = wave_inclusive_scan(v);
  if (lane == 63) wsum[w] = inc;
  __syncthreads();
  int base = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < kThreads / 64; ++i) {
    int s = wsum[i];
    if (i < w) base += s;
    tot += s;
  }
  __syncthreads();
  *total = tot;
  return base + inc - v;
}

__global__ __launch_bounds__(kThreads) void scan_reduce_kernel(const int* __restrict__ in,
                                                               int* __restrict__ sums, long long n) {
  const long long base = (long long)blockIdx.x * kTile + (long long)threadIdx.x * kItems;
  int s = 0;
  if (base + kItems <= n) {
    const int4 a = *reinterpret_cast<const int4*>(in + base);
    const int4 b = *reinterpret_cast<const int4*>(in + base + 4);
    s = a.x + a.y + a.z + a.w + b.x + b.y + b.z + b.w;
  } else {
    for (int i = 0; i < kItems; ++i)
      if (base + i < n) s += in[base + i];
  }
  int tot;
  block_exclusive_scan(s, &tot);
  if (threadIdx.x == 0) sums[blockIdx.x] = tot;
}

// out[i] = block_base[blockIdx] + exclusive prefix inside the block. block_base may be NULL
// (single block). Writes the grand total when total != NULL and this is the last block.
__global__ __launch_bounds__(kThreads) void scan_apply_kernel(const int* __restrict__ in,
                                                              int* __restrict__ out,
                                                              const int* __restrict__ block_base,
                                                              long long n, long long* total) {
  const long long base = (long long)blockIdx.x * kTile + (long long)threadIdx.x * kItems;
  int v[kItems];
  if (base + kItems <= n) {
    const int4 a = *reinterpret_cast<const int4*>(in + base);
    const int4 b = *reinterpret_cast<const int4*>(in + base + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
    v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  } else {
#pragma unroll
    for (int i = 0; i < kItems; ++i) v[i] = (base + i < n) ? in[base + i] : 0;
  }
  int s = 0;
#pragma unroll
  for (int i = 0; i < kItems; ++i) s += v[i];
  int tot;
  int ex = block_exclusive_scan(s, &tot);
  const int bb = block_base ? block_base[blockIdx.x] : 0;
  ex += bb;
  int o[kItems];
#pragma unroll
  for (int i = 0; i < kItems; ++i) {
    o[i] = ex;
    ex += v[i];
  }
  if (base + kItems <= n) {
    *reinterpret_cast<int4*>(out + base) = make_int4(o[0], o[1], o[2], o[3]);
    *reinterpret_cast<int4*>(out + base + 4) = make_int4(o[4], o[5], o[6], o[7]);
  } else {
#pragma unroll
    for (int i = 0; i < kItems; ++i)
      if (base + i < n) out[base + i] = o[i];
  }
  if (total && blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) *total = (long long)bb + tot;
}
}  // namespace

// ---- fill kernel used instead of hipMemsetAsync ---------------------------------------------------------------
// hipMemsetAsync nodes captured in a HIP graph ran out of order on every replay after the first (ROCm 7.2, gfx950;
// tools/graph_memset_probe.py: a kernel - memset - kernel chain left 19-100 % of the bytes wrong from the second replay
// on), which made the captured step fault on new inputs. A plain kernel node keeps its place in the chain, so every
// fill on a path that may be captured is this kernel: 16-byte vector stores over the aligned body, bytes at the ends.
namespace {
__global__ __launch_bounds__(256) void fill_kernel(unsigned char* __restrict__ p, unsigned value, size_t head,
                                                   size_t quads, size_t tail) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const unsigned w = value * 0x01010101u;
  uint4* body = reinterpret_cast<uint4*>(p + head);
  for (size_t q = i; q < quads; q += (size_t)gridDim.x * 256) body[q] = make_uint4(w, w, w, w);
  if (i < head) p[i] = (unsigned char)value;
  if (i < tail) p[head + 16 * quads + i] = (unsigned char)value;
}
}  // namespace

int32_t vx_fill_async(void* d_ptr, int value, size_t bytes, hipStream_t stream) {
  if (bytes == 0) return VX_OK;
  const uintptr_t a = reinterpret_cast<uintptr_t>(d_ptr);
  size_t head = (16 - (a & 15)) & 15;
  if (head > bytes) head = bytes;
  const size_t quads = (bytes - head) / 16, tail = bytes - head - 16 * quads;
  size_t blocks = (quads + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > 16384) blocks = 16384;          // grid-stride beyond that: 16384 x 256 lanes x 16 B = 64 MiB per sweep
  fill_kernel<<<(unsigned)blocks, 256, 0, stream>>>((unsigned char*)d_ptr, (unsigned)value & 0xffu, head, quads, tail);
  VX_LAUNCH_CHECK();
  return VX_OK;
}

// public form (include/vx_abi.h): the capturable clear a caller of the "_dn" entry points needs for its count vector
extern "C" int32_t vx_fill(void* d_ptr, int32_t value, size_t bytes, void* stream) {
  VX_REQUIRE(d_ptr != nullptr || bytes == 0, "vx_fill: null pointer");
  return vx_fill_async(d_ptr, value, bytes, (hipStream_t)stream);
}

size_t vx_scan_workspace_bytes(int64_t n) {
  size_t bytes = 0;
  int64_t m = vx_cdiv(n > 0 ? n : 1, kTile);
  while (m > 1) {
    bytes += vx_align((size_t)m * sizeof(int));
    m = vx_cdiv(m, kTile);
  }
  return bytes + 256;
}

// in/out must be 16-byte aligned; in == out allowed.
int32_t vx_scan_exclusive_i32(const int32_t* d_in, int32_t* d_out, int64_t n, int64_t* d_total,
                              void* d_ws, size_t ws_bytes, hipStream_t stream) {
  if (n <= 0) {
    if (d_total) return vx_fill_async(d_total, 0, sizeof(int64_t), stream);
    return VX_OK;
  }
  if (ws_bytes < vx_scan_workspace_bytes(n)) {
    vx_set_error("scan: workspace too small");
    return VX_ERR_WORKSPACE;
  }
  const int64_t nb = vx_cdiv(n, kTile);
  if (nb == 1) {
    scan_apply_kernel<<<1, kThreads, 0, stream>>>(d_in, d_out, nullptr, n, (long long*)d_total);
    VX_LAUNCH_CHECK();
    return VX_OK;
  }
  int* sums = (int*)d_ws;
  const size_t used = vx_align((size_t)nb * sizeof(int));
  scan_reduce_kernel<<<(unsigned)nb, kThreads, 0, stream>>>(d_in, sums, n);
  VX_LAUNCH_CHECK();
  int32_t rc = vx_scan_exclusive_i32(sums, sums, nb, nullptr, (char*)d_ws + used, ws_bytes - used,
                                     stream);
  if (rc) return rc;
  scan_apply_kernel<<<(unsigned)nb, kThreads, 0, stream>>>(d_in, d_out, sums, n,
                                                           (long long*)d_total);
  VX_LAUNCH_CHECK();
  return VX_OK;
}

// ---- diagnostic (tools/graph_stale_probe.py): what does a kernel see of a device word that a copy or an earlier
// kernel rewrote just before it -- through a plain (scalar-cache) load, a plain vector load and an agent-scope load?
__global__ void vx_echo_kernel(const long long* __restrict__ src, long long* __restrict__ out_scalar,
                               long long* __restrict__ out_vector, long long* __restrict__ out_agent) {
  const long long s = *src;                                   // uniform address: s_load through the scalar cache
  const long long* vp = src + (threadIdx.x >> 10);            // same address, lane-dependent form: vector load
  const long long v = *vp;
  const long long a = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (threadIdx.x == 0) {
    out_scalar[blockIdx.x] = s;
    out_vector[blockIdx.x] = v;
    out_agent[blockIdx.x] = a;
  }
}
extern "C" int32_t vx_debug_echo(const int64_t* d_src, int64_t* d_out_scalar, int64_t* d_out_vector,
                                 int64_t* d_out_agent, int32_t blocks, void* stream) {
  vx_echo_kernel<<<blocks, 64, 0, (hipStream_t)stream>>>((const long long*)d_src, (long long*)d_out_scalar,
                                                         (long long*)d_out_vector, (long long*)d_out_agent);
  VX_LAUNCH_CHECK();
  return VX_OK;
}

extern "C" int32_t vx_debug_memset(void* d_ptr, int32_t value, size_t bytes, void* stream) {
  VX_HIP(hipMemsetAsync(d_ptr, value, bytes, (hipStream_t)stream));
  return VX_OK;
}
