// SparseConvTensor.dense(): zero-fill [B][C][D][H][W] then scatter rows (HBM-bound).
// Rows arrive sorted by linear index, so for a fixed channel consecutive rows land on consecutive
// addresses: 64 rows are transposed through LDS and every wave-instruction writes one channel of
// 64 rows (coalesced wherever the rows are x-adjacent).
#include "vx_common.h"

namespace {
constexpr int kThreads = 256;
constexpr int kRows = 64;

template <typename T>  // moved as raw 4- or 2-byte words (fp32 / bf16)
__global__ __launch_bounds__(kThreads) void to_dense_kernel(const T* __restrict__ feat,
                                                            const int* __restrict__ coords,
                                                            long long n, const long long* __restrict__ d_n,
                                                            int c, int d, int hh, int ww,
                                                            T* __restrict__ dense) {
  extern __shared__ __attribute__((aligned(16))) char tile_raw[];
  T* tile = reinterpret_cast<T*>(tile_raw);  // [kRows][c + 1]
  const int cp = c + 1;
  const long long row0 = (long long)blockIdx.x * kRows;
  if (d_n) {                                 // "_dn" form: n is the capacity, the row count stays on the device
    const long long dn = *d_n;
    n = dn < n ? (dn > 0 ? dn : 0) : n;
  }
  if (row0 >= n) return;
  const int rows = (int)((n - row0) < kRows ? (n - row0) : kRows);
  for (int idx = threadIdx.x; idx < rows * c; idx += kThreads) {
    const int rr = idx / c, cc = idx - rr * c;
    tile[rr * cp + cc] = feat[(row0 + rr) * c + cc];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane >= rows) return;
  const int4 q = *reinterpret_cast<const int4*>(coords + 4 * (row0 + lane));
  const long long vol = (long long)d * hh * ww;
  const long long sp = ((long long)q.y * hh + q.z) * ww + q.w;
  T* dst = dense + (long long)q.x * c * vol + sp;
  for (int cc = wave; cc < c; cc += kThreads / 64) dst[(long long)cc * vol] = tile[lane * cp + cc];
}

// The gradient of dense(): row r of the result is dense[b][:][z][y][x] at r's coordinates. Same tile, other
// direction: every wave-instruction reads one channel of 64 rows, the tile leaves as whole rows.
template <typename T>
__global__ __launch_bounds__(kThreads) void from_dense_kernel(const T* __restrict__ dense,
                                                              const int* __restrict__ coords, long long n, int c,
                                                              int d, int hh, int ww, T* __restrict__ feat) {
  extern __shared__ __attribute__((aligned(16))) char tile_raw[];
  T* tile = reinterpret_cast<T*>(tile_raw);  // [kRows][c + 1]
  const int cp = c + 1;
  const long long row0 = (long long)blockIdx.x * kRows;
  const int rows = (int)((n - row0) < kRows ? (n - row0) : kRows);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane < rows) {
    const int4 q = *reinterpret_cast<const int4*>(coords + 4 * (row0 + lane));
    const long long vol = (long long)d * hh * ww;
    const long long sp = ((long long)q.y * hh + q.z) * ww + q.w;
    const T* src = dense + (long long)q.x * c * vol + sp;
    for (int cc = wave; cc < c; cc += kThreads / 64) tile[lane * cp + cc] = src[(long long)cc * vol];
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < rows * c; idx += kThreads) {
    const int rr = idx / c, cc = idx - rr * c;
    feat[(row0 + rr) * c + cc] = tile[rr * cp + cc];
  }
}
}  // namespace

static int32_t to_dense_impl(const void* d_feat, const int32_t* d_coords, int64_t n, const int64_t* d_n,
                             int32_t c, int32_t batch, const int32_t* shape, void* d_dense, int32_t dtype,
                             hipStream_t stream) {
  VX_REQUIRE(dtype == VX_F32 || dtype == VX_BF16, "to_dense: dtype must be VX_F32 or VX_BF16");
  const size_t esz = dtype == VX_F32 ? 4 : 2;
  VX_REQUIRE(shape && d_dense && batch >= 1 && c >= 1 && c <= 4096 && n >= 0, "to_dense: bad arguments");
  const size_t total = (size_t)batch * c * shape[0] * shape[1] * shape[2];
  if (int32_t rc_ = vx_fill_async(d_dense, 0, total * esz, stream)) return rc_;
  if (n == 0) return VX_OK;
  VX_REQUIRE(d_feat && d_coords, "to_dense: null pointer");
  const size_t lds = (size_t)kRows * (c + 1) * esz;
  const unsigned grid = (unsigned)vx_cdiv(n, kRows);
  if (dtype == VX_F32)
    to_dense_kernel<uint32_t><<<grid, kThreads, lds, stream>>>(
        (const uint32_t*)d_feat, d_coords, n, (const long long*)d_n, c, shape[0], shape[1], shape[2],
        (uint32_t*)d_dense);
  else
    to_dense_kernel<uint16_t><<<grid, kThreads, lds, stream>>>(
        (const uint16_t*)d_feat, d_coords, n, (const long long*)d_n, c, shape[0], shape[1], shape[2],
        (uint16_t*)d_dense);
  VX_LAUNCH_CHECK();
  return VX_OK;
}

extern "C" int32_t vx_to_dense(const void* d_feat, const int32_t* d_coords, int64_t n, int32_t c,
                               int32_t batch, const int32_t* shape, void* d_dense, int32_t dtype,
                               void* stream) {
  return to_dense_impl(d_feat, d_coords, n, nullptr, c, batch, shape, d_dense, dtype, (hipStream_t)stream);
}

extern "C" int32_t vx_to_dense_dn(const void* d_feat, const int32_t* d_coords, const int64_t* d_n, int64_t cap,
                                  int32_t c, int32_t batch, const int32_t* shape, void* d_dense, int32_t dtype,
                                  void* stream) {
  VX_REQUIRE(d_n && cap >= 1, "to_dense_dn: need the device row count and a capacity >= 1");
  return to_dense_impl(d_feat, d_coords, cap, d_n, c, batch, shape, d_dense, dtype, (hipStream_t)stream);
}

extern "C" int32_t vx_from_dense(const void* d_dense, const int32_t* d_coords, int64_t n, int32_t c, int32_t batch,
                                 const int32_t* shape, void* d_feat, int32_t dtype, void* stream) {
  VX_REQUIRE(dtype == VX_F32 || dtype == VX_BF16, "from_dense: dtype must be VX_F32 or VX_BF16");
  VX_REQUIRE(shape && batch >= 1 && c >= 1 && c <= 4096 && n >= 0, "from_dense: bad arguments");
  if (n == 0) return VX_OK;
  VX_REQUIRE(d_dense && d_coords && d_feat, "from_dense: null pointer");
  const size_t esz = dtype == VX_F32 ? 4 : 2;
  const size_t lds = (size_t)kRows * (c + 1) * esz;
  const unsigned grid = (unsigned)vx_cdiv(n, kRows);
  if (dtype == VX_F32)
    from_dense_kernel<uint32_t><<<grid, kThreads, lds, (hipStream_t)stream>>>(
        (const uint32_t*)d_dense, d_coords, n, c, shape[0], shape[1], shape[2], (uint32_t*)d_feat);
  else
    from_dense_kernel<uint16_t><<<grid, kThreads, lds, (hipStream_t)stream>>>(
        (const uint16_t*)d_dense, d_coords, n, c, shape[0], shape[1], shape[2], (uint16_t*)d_feat);
  VX_LAUNCH_CHECK();
  return VX_OK;
}
