// Hard voxelization + fused mean VFE on gfx950.
//
// Semantics restated from det3d/ops/point_cloud/point_cloud_ops.py:138-221 (points_to_voxel_new):
// voxel id = rank of the voxel's first point in input order; the first max_points points of a
// voxel (input order) are kept; voxels past max_voxels are dropped with their points.
//
// MI355X mapping (all phases HBM/atomic-latency bound, no GEMM-shaped work):
//   1. every point hashes its cell into a lock-free open-addressing table in HBM (64-bit CAS on
//      the key, atomicMin on the smallest point index, atomicAdd on the population);
//   2. flag[i] = (i is the smallest index of its cell); exclusive scan of the flags in point
//      order = voxel id  -> exactly the reference's first-appearance order, without a sort;
//   3. CSR of the points of every voxel (scan of populations + atomic fill), then one lane per
//      voxel picks its first max_points point indices in ascending order and writes the mean
//      (and optionally the zero-padded voxels[M][T][F] tensor) -- the 27 MB voxels tensor of
//      the reference need not exist when only the mean is wanted.
// fp32 cell arithmetic is floor((p - lo) / vs) with IEEE subtraction and true division
// (point_cloud_ops.py:173-174); this file is compiled with -ffp-contract=off and no fast-math.
#include <math.h>

#include <mutex>

#include "vx_common.h"

namespace {
constexpr int kThreads = 256;

constexpr int kMaxFrames = 64;

struct VoxGeom {
  float lo[3];
  float vs[3];
  int grid[3];  // x, y, z
  int n_frames;
  int batch_idx0;             // frame f gets batch index batch_idx0 + f
  long long off[kMaxFrames + 1];  // point ranges of the frames (frame-major concatenation)
  // "_dn" form: the ranges live on the device (int64[n_frames + 1]) and `off` is unused; the point count is
  // d_off[n_frames], the host-side n only the capacity the grids were sized for
  const long long* d_off;
  long long row_cap;          // "_dn" form: rows the outputs hold (voxels past it are dropped, the count is not clamped)
};

__device__ __forceinline__ long long frame_off(const VoxGeom& g, int f) { return g.d_off ? g.d_off[f] : g.off[f]; }

__device__ __forceinline__ long long points_of(const VoxGeom& g, long long n) {
  if (g.d_off) {
    const long long d = g.d_off[g.n_frames];
    n = d < n ? (d > 0 ? d : 0) : n;
  }
  return n;
}

__device__ __forceinline__ int frame_of(const VoxGeom& g, long long i) {
  int lo = 0, hi = g.n_frames;  // off[lo] <= i < off[hi]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (i >= frame_off(g, mid)) lo = mid; else hi = mid;
  }
  return lo;
}

__global__ __launch_bounds__(kThreads) void vox_init_table(VxSlot* t, long long cap) {
  long long i = (long long)blockIdx.x * kThreads + threadIdx.x;
  if (i < cap) {
    int4 v;
    v.x = -1; v.y = -1;       // key = VX_EMPTY
    v.z = 0x7fffffff;         // a = smallest point index so far
    v.w = 0;                  // b = population
    *reinterpret_cast<int4*>(&t[i]) = v;
  }
}

__global__ __launch_bounds__(kThreads) void vox_insert(const float* __restrict__ pts, long long n,
                                                       int nf, VoxGeom g, VxSlot* t,
                                                       long long mask, int* __restrict__ slot_of) {
  long long i = (long long)blockIdx.x * kThreads + threadIdx.x;
  if (i >= points_of(g, n)) return;
  const float* p = pts + i * nf;
  float c[3];
  bool ok = true;
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    c[d] = floorf(__fdiv_rn(__fsub_rn(p[d], g.lo[d]), g.vs[d]));
    ok = ok && (c[d] >= 0.0f) && (c[d] < (float)g.grid[d]);
  }
  int s = -1;
  if (ok) {
    const int f = g.n_frames > 1 ? frame_of(g, i) : 0;
    long long key = (((long long)f * g.grid[2] + (int)c[2]) * g.grid[1] + (int)c[1]) * g.grid[0] + (int)c[0];
    long long sl = vx_hash_insert(t, mask, key);
    atomicMin(&t[sl].a, (int)i);
    atomicAdd(&t[sl].b, 1);
    s = (int)sl;
  }
  slot_of[i] = s;
}

__global__ __launch_bounds__(kThreads) void vox_flag(const VxSlot* __restrict__ t,
                                                     const int* __restrict__ slot_of, long long n,
                                                     VoxGeom g, int* __restrict__ flag) {
  long long i = (long long)blockIdx.x * kThreads + threadIdx.x;
  if (i >= n) return;
  if (i >= points_of(g, n)) {      // capacity beyond the device point count: the scan runs over it
    flag[i] = 0;
    return;
  }
  int s = slot_of[i];
  flag[i] = (s >= 0 && t[s].a == (int)i) ? 1 : 0;
}

// first point of every voxel publishes the voxel: id, coordinates, clipped count
__global__ __launch_bounds__(kThreads) void vox_assign(VxSlot* t, const int* __restrict__ slot_of,
                                                       const int* __restrict__ flag,
                                                       const int* __restrict__ pos, long long n,
                                                       VoxGeom g, int max_points,
                                                       long long max_voxels,
                                                       int* __restrict__ coords,
                                                       int* __restrict__ num_points,
                                                       int* __restrict__ cntv,
                                                       const long long* __restrict__ total,
                                                       long long* __restrict__ num_voxels) {
  long long i = (long long)blockIdx.x * kThreads + threadIdx.x;
  if (i == 0) *num_voxels = (*total < max_voxels) ? *total : max_voxels;
  if (i >= points_of(g, n) || !flag[i]) return;
  const int s = slot_of[i];
  const int v = pos[i];
  if (v >= max_voxels || v >= g.row_cap) {
    t[s].a = -1;
    return;
  }
  t[s].a = v;
  const long long key = t[s].key;
  const int cnt = t[s].b;
  const int x = (int)(key % g.grid[0]);
  const long long r = key / g.grid[0];
  const int y = (int)(r % g.grid[1]);
  const long long r2 = r / g.grid[1];
  const int z = (int)(r2 % g.grid[2]);
  const int f = (int)(r2 / g.grid[2]);
  *reinterpret_cast<int4*>(coords + 4ll * v) = make_int4(g.batch_idx0 + f, z, y, x);
  num_points[v] = cnt < max_points ? cnt : max_points;
  cntv[v] = cnt;
}

__global__ __launch_bounds__(kThreads) void vox_fill_lists(const VxSlot* __restrict__ t,
                                                           const int* __restrict__ slot_of,
                                                           long long n, VoxGeom g,
                                                           const int* __restrict__ off,
                                                           int* __restrict__ fill,
                                                           int* __restrict__ plist) {
  long long i = (long long)blockIdx.x * kThreads + threadIdx.x;
  if (i >= points_of(g, n)) return;
  const int s = slot_of[i];
  if (s < 0) return;
  const int v = t[s].a;
  if (v < 0) return;
  const int p = off[v] + atomicAdd(&fill[v], 1);
  plist[p] = (int)i;
}

// one lane per voxel: ascending selection of its first max_points point indices
__global__ __launch_bounds__(kThreads) void vox_gather(const float* __restrict__ pts, int nf,
                                                       const int* __restrict__ off,
                                                       const int* __restrict__ cntv,
                                                       const int* __restrict__ plist,
                                                       const long long* __restrict__ num_voxels,
                                                       long long row_cap,
                                                       int max_points, float* __restrict__ voxels,
                                                       float* __restrict__ mean) {
  const long long v = (long long)blockIdx.x * kThreads + threadIdx.x;
  if (v >= *num_voxels || v >= row_cap) return;
  const int b = off[v], cnt = cntv[v];
  const int keep = cnt < max_points ? cnt : max_points;
  float sum[8];
#pragma unroll
  for (int f = 0; f < 8; ++f) sum[f] = 0.0f;
  int last = -1;
  for (int t = 0; t < max_points; ++t) {
    float* dst = voxels ? voxels + ((long long)v * max_points + t) * nf : nullptr;
    if (t < keep) {
      int best = 0x7fffffff;
      for (int j = 0; j < cnt; ++j) {
        const int p = plist[b + j];
        if (p > last && p < best) best = p;
      }
      last = best;
      const float* src = pts + (long long)best * nf;
      for (int f = 0; f < nf; ++f) {
        const float x = src[f];
        if (f < 8) sum[f] = __fadd_rn(sum[f], x);
        if (dst) dst[f] = x;
      }
    } else if (dst) {
      for (int f = 0; f < nf; ++f) dst[f] = 0.0f;
    } else {
      break;
    }
  }
  if (mean) {
    const float d = (float)keep;
    for (int f = 0; f < nf && f < 8; ++f) mean[v * nf + f] = __fdiv_rn(sum[f], d);
  }
}

// voxels per frame = flagged points inside the frame's point range
__global__ void vox_frame_counts(const int* __restrict__ pos, long long n, VoxGeom g,
                                 const long long* __restrict__ total, long long* __restrict__ counts) {
  const int f = threadIdx.x;
  if (f >= g.n_frames) return;
  n = points_of(g, n);
  const long long o0 = frame_off(g, f), o1 = frame_off(g, f + 1);
  const long long a = o0 < n ? pos[o0] : *total;
  const long long b = o1 < n ? pos[o1] : *total;
  counts[f] = b - a;
}

struct VoxWs {
  VxSlot* table;
  long long cap;
  int *slot_of, *flag, *pos, *cntv, *off, *fill, *plist;
  long long* total;
  void* scan_ws;
  size_t scan_bytes;
  size_t bytes;
};

VoxWs carve(void* base, int64_t n) {
  VoxWs w;
  char* p = (char*)base;
  size_t o = 0;
  auto take = [&](size_t b) {
    char* r = p ? p + o : nullptr;
    o += vx_align(b);
    return r;
  };
  const int64_t nn = n > 0 ? n : 1;
  w.cap = vx_hash_capacity(nn);
  w.table = (VxSlot*)take(sizeof(VxSlot) * (size_t)w.cap);
  w.slot_of = (int*)take(4 * (size_t)nn);
  w.flag = (int*)take(4 * (size_t)nn);
  w.pos = (int*)take(4 * (size_t)nn);
  w.cntv = (int*)take(4 * (size_t)nn);
  w.off = (int*)take(4 * (size_t)nn);
  w.fill = (int*)take(4 * (size_t)nn);
  w.plist = (int*)take(4 * (size_t)nn);
  w.total = (long long*)take(64);
  w.scan_bytes = vx_scan_workspace_bytes(nn);
  w.scan_ws = take(w.scan_bytes);
  w.bytes = o;
  return w;
}
}  // namespace

extern "C" size_t vx_voxelize_workspace_bytes(int64_t n_points) {
  return carve(nullptr, n_points).bytes;
}

static int32_t voxelize_impl(const float* d_points, int64_t n, int32_t nf, const float* voxel_size,
                             const float* range, int32_t max_points, int64_t max_voxels,
                             int32_t batch_idx, int32_t n_frames, const int64_t* frame_offsets,
                             const int64_t* d_frame_offsets, int32_t* d_coords, int32_t* d_num_points, float* d_voxels,
                             float* d_mean, int64_t cap, int64_t* d_num_voxels,
                             int64_t* h_num_voxels, void* d_ws, size_t ws_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  VX_REQUIRE(n >= 0 && n < (1ll << 30), "voxelize: n_points out of range");
  VX_REQUIRE(nf >= 3 && nf <= 8, "voxelize: n_feat must be in [3, 8]");
  VX_REQUIRE(max_points >= 1 && max_voxels >= 1, "voxelize: max_points/max_voxels must be >= 1");
  VX_REQUIRE(voxel_size && range && d_coords && d_num_points && d_num_voxels && d_ws,
             "voxelize: null pointer");
  VX_REQUIRE(n == 0 || d_points, "voxelize: null points");
  int64_t need_cap = n < max_voxels ? n : max_voxels;
  if (d_frame_offsets && cap < need_cap) need_cap = cap;   // "_dn": rows past `cap` are dropped, the count says so
  if (cap < need_cap) {
    vx_set_error("voxelize: output capacity %lld < min(n_points, max_voxels) = %lld",
                 (long long)cap, (long long)need_cap);
    return VX_ERR_CAPACITY;
  }
  VoxWs w = carve(d_ws, n);
  if (ws_bytes < w.bytes) {
    vx_set_error("voxelize: workspace %zu < %zu bytes", ws_bytes, w.bytes);
    return VX_ERR_WORKSPACE;
  }
  VoxGeom g;
  g.n_frames = n_frames;
  g.batch_idx0 = batch_idx;
  g.d_off = (const long long*)d_frame_offsets;
  g.row_cap = cap;
  for (int f = 0; f <= n_frames; ++f)
    g.off[f] = d_frame_offsets ? 0 : (frame_offsets ? frame_offsets[f] : (f ? n : 0));
  long long cells = n_frames;
  for (int d = 0; d < 3; ++d) {
    g.lo[d] = range[d];
    g.vs[d] = voxel_size[d];
    // point_cloud_ops.py:171-172: round((hi - lo) / vs) in fp32 (round-half-even == nearbyintf)
    volatile float span = range[3 + d] - range[d];
    volatile float q = span / voxel_size[d];
    g.grid[d] = (int)nearbyintf(q);
    VX_REQUIRE(g.grid[d] >= 1, "voxelize: empty grid");
    cells *= g.grid[d];
  }
  VX_REQUIRE(cells < (1ll << 40), "voxelize: grid too large");
  const size_t n_counts = n_frames > 1 ? (size_t)n_frames + 1 : 1;
  if (n == 0) {
    if (int32_t rc_ = vx_fill_async(d_num_voxels, 0, sizeof(int64_t) * n_counts, stream)) return rc_;
  } else {
    const unsigned gp = (unsigned)vx_cdiv(n, kThreads);
    vox_init_table<<<(unsigned)vx_cdiv(w.cap, kThreads), kThreads, 0, stream>>>(w.table, w.cap);
    VX_LAUNCH_CHECK();
    vox_insert<<<gp, kThreads, 0, stream>>>(d_points, n, nf, g, w.table, w.cap - 1, w.slot_of);
    VX_LAUNCH_CHECK();
    vox_flag<<<gp, kThreads, 0, stream>>>(w.table, w.slot_of, n, g, w.flag);
    VX_LAUNCH_CHECK();
    int32_t rc = vx_scan_exclusive_i32(w.flag, w.pos, n, (int64_t*)w.total, w.scan_ws,
                                       w.scan_bytes, stream);
    if (rc) return rc;
    const bool lists = d_voxels || d_mean;
    if (lists) {
      if (int32_t rc_ = vx_fill_async(w.cntv, 0, 4 * (size_t)n, stream)) return rc_;
      if (int32_t rc_ = vx_fill_async(w.fill, 0, 4 * (size_t)n, stream)) return rc_;
    }
    // d_num_voxels[0] = total; [1 .. n_frames] = per-frame counts (batched call only)
    vox_assign<<<gp, kThreads, 0, stream>>>(w.table, w.slot_of, w.flag, w.pos, n, g, max_points,
                                            (long long)max_voxels, d_coords,
                                            d_num_points, w.cntv, w.total,
                                            (long long*)d_num_voxels);
    VX_LAUNCH_CHECK();
    if (n_frames > 1) {
      vox_frame_counts<<<1, kMaxFrames, 0, stream>>>(w.pos, n, g, w.total,
                                                     (long long*)d_num_voxels + 1);
      VX_LAUNCH_CHECK();
    }
    if (lists) {
      rc = vx_scan_exclusive_i32(w.cntv, w.off, need_cap, nullptr, w.scan_ws, w.scan_bytes,
                                 stream);
      if (rc) return rc;
      vox_fill_lists<<<gp, kThreads, 0, stream>>>(w.table, w.slot_of, n, g, w.off, w.fill, w.plist);
      VX_LAUNCH_CHECK();
      vox_gather<<<(unsigned)vx_cdiv(need_cap, kThreads), kThreads, 0, stream>>>(
          d_points, nf, w.off, w.cntv, w.plist, (const long long*)d_num_voxels, (long long)cap, max_points,
          d_voxels, d_mean);
      VX_LAUNCH_CHECK();
    }
  }
  if (h_num_voxels) {
    VX_HIP(hipMemcpyAsync(h_num_voxels, d_num_voxels, sizeof(int64_t) * n_counts,
                          hipMemcpyDeviceToHost, stream));
    VX_HIP(hipStreamSynchronize(stream));
  }
  return VX_OK;
}

extern "C" int32_t vx_voxelize_hard(const float* d_points, int64_t n, int32_t nf,
                                    const float* voxel_size, const float* range,
                                    int32_t max_points, int64_t max_voxels, int32_t batch_idx,
                                    int32_t* d_coords, int32_t* d_num_points, float* d_voxels,
                                    float* d_mean, int64_t cap, int64_t* d_num_voxels,
                                    int64_t* h_num_voxels, void* d_ws, size_t ws_bytes,
                                    void* stream) {
  return voxelize_impl(d_points, n, nf, voxel_size, range, max_points, max_voxels, batch_idx, 1,
                       nullptr, nullptr, d_coords, d_num_points, d_voxels, d_mean, cap, d_num_voxels,
                       h_num_voxels, d_ws, ws_bytes, stream);
}

extern "C" int32_t vx_voxelize_hard_batch(const float* d_points, int32_t n_frames,
                                          const int64_t* frame_offsets, int32_t nf,
                                          const float* voxel_size, const float* range,
                                          int32_t max_points, int64_t max_voxels,
                                          int32_t* d_coords, int32_t* d_num_points,
                                          float* d_voxels, float* d_mean, int64_t cap,
                                          int64_t* d_num_voxels, int64_t* h_num_voxels,
                                          void* d_ws, size_t ws_bytes, void* stream) {
  VX_REQUIRE(n_frames >= 1 && n_frames <= kMaxFrames && frame_offsets,
             "voxelize_batch: 1 <= n_frames <= 64 and frame_offsets required");
  VX_REQUIRE(frame_offsets[0] == 0, "voxelize_batch: frame_offsets[0] must be 0");
  for (int f = 0; f < n_frames; ++f) {
    VX_REQUIRE(frame_offsets[f + 1] >= frame_offsets[f], "voxelize_batch: offsets must be non-decreasing");
    // no frame can produce more voxels than points: with n_f <= max_voxels truncation cannot occur
    VX_REQUIRE(frame_offsets[f + 1] - frame_offsets[f] <= max_voxels,
               "voxelize_batch: a frame has more points than max_voxels; use the per-frame call");
  }
  const int64_t n = frame_offsets[n_frames];
  return voxelize_impl(d_points, n, nf, voxel_size, range, max_points, (int64_t)1 << 40, 0, n_frames,
                       frame_offsets, nullptr, d_coords, d_num_points, d_voxels, d_mean, cap, d_num_voxels,
                       h_num_voxels, d_ws, ws_bytes, stream);
}

extern "C" int32_t vx_voxelize_hard_batch_dn(const float* d_points, int32_t n_frames,
                                             const int64_t* d_frame_offsets, int64_t cap_points,
                                             int32_t nf, const float* voxel_size, const float* range,
                                             int32_t max_points, int32_t* d_coords,
                                             int32_t* d_num_points, float* d_mean, int64_t cap,
                                             int64_t* d_num_voxels, void* d_ws, size_t ws_bytes,
                                             void* stream) {
  VX_REQUIRE(n_frames >= 1 && n_frames <= kMaxFrames && d_frame_offsets && cap_points >= 1 && cap >= 1,
             "voxelize_batch_dn: 1 <= n_frames <= 64, device frame offsets and capacities >= 1 required");
  return voxelize_impl(d_points, cap_points, nf, voxel_size, range, max_points, (int64_t)1 << 40, 0, n_frames,
                       nullptr, d_frame_offsets, d_coords, d_num_points, nullptr, d_mean, cap, d_num_voxels,
                       nullptr, d_ws, ws_bytes, stream);
}

// ---------------------------------------------------------------------------------------------------------
// Allocation-free form (no counterpart in the reference; include/vx_abi.h). The workspace of voxelize_impl
// -- hash table, slot / flag / rank / count / CSR arrays, scan partials -- lives in a file-scope device array
// instead of a caller buffer. Its NAME is a host-side stand-in, not a device address: the address each GPU gave
// its own copy is read once per device with hipGetSymbolAddress and only that pointer is ever handed to kernels,
// memsets and copies.
// ---------------------------------------------------------------------------------------------------------
namespace {
constexpr int64_t kStaticMaxPoints = 262144;   // one 10-sweep nuScenes-shaped frame is ~247 k points
// carve(kStaticMaxPoints): 8 MiB table (2^19 slots x 16 B) + 7 int32 arrays of 1 MiB + 64 B total + scan partials (< 2 KiB)
constexpr size_t kStaticBytes = (size_t)16 << 20;
__device__ __align__(256) unsigned char scratch[kStaticBytes];

std::mutex g_scratch_mu;
void* g_scratch_ptr[64];                        // per device: device address of that GPU's copy of `scratch`

int32_t scratch_address(void** out) {
  int dev = -1;
  VX_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64) {
    vx_set_error("voxelize_static: device index %d out of range", dev);
    return VX_ERR_ARG;
  }
  std::lock_guard<std::mutex> lock(g_scratch_mu);
  if (!g_scratch_ptr[dev]) {
    void* p = nullptr;
    VX_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));   // with device `dev` current: ITS copy of the array
    if (!p) {
      vx_set_error("voxelize_static: hipGetSymbolAddress returned a null address");
      return VX_ERR_HIP;
    }
    g_scratch_ptr[dev] = p;
  }
  *out = g_scratch_ptr[dev];
  return VX_OK;
}
}  // namespace

extern "C" int64_t vx_voxelize_static_max_points(void) { return kStaticMaxPoints; }

extern "C" int32_t vx_voxelize_static_init(void) {
  void* p = nullptr;
  return scratch_address(&p);
}

extern "C" int32_t vx_voxelize_hard_static(const float* d_points, int64_t n, int32_t nf,
                                           const float* voxel_size, const float* range,
                                           int32_t max_points, int64_t max_voxels, int32_t batch_idx,
                                           int32_t* d_coords, int32_t* d_num_points, float* d_voxels,
                                           float* d_mean, int64_t cap, int64_t* d_num_voxels,
                                           int64_t* h_num_voxels, void* stream) {
  if (n > kStaticMaxPoints) {
    vx_set_error("voxelize_static: %lld points exceed the static scratch limit of %lld; use vx_voxelize_hard "
                 "with a caller workspace", (long long)n, (long long)kStaticMaxPoints);
    return VX_ERR_CAPACITY;
  }
  static_assert(kStaticBytes % 256 == 0, "scratch size");
  if (carve(nullptr, n).bytes > kStaticBytes) {   // cannot happen for n <= kStaticMaxPoints; kept as a guard
    vx_set_error("voxelize_static: workspace of %lld points does not fit the static scratch", (long long)n);
    return VX_ERR_CAPACITY;
  }
  void* ws = nullptr;
  if (int32_t rc = scratch_address(&ws)) return rc;
  // voxelize_impl re-initialises everything it reads (table, counters) on every call, so the contents the last
  // call left behind do not matter; it launches kernels, memsets and (only when h_num_voxels != NULL) one D2H copy.
  return voxelize_impl(d_points, n, nf, voxel_size, range, max_points, max_voxels, batch_idx, 1, nullptr, nullptr,
                       d_coords, d_num_points, d_voxels, d_mean, cap, d_num_voxels, h_num_voxels, ws,
                       kStaticBytes, stream);
}
