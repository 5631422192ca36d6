// Train-mode BatchNorm1d over sparse features [n][c] (fp32), fused with the residual add and the ReLU that
// follow it in the backbone (det3d/models/backbones/scn.py:82-98 block wiring, :333-364 conv+BN+ReLU;
// norm layer det3d/models/utils/norm.py:62,99-103; cross-rank statistics det3d/torchie/apis/train.py:301).
//
// All five kernels are HBM-bound streams over [n][c]: a thread owns 8 consecutive channels of a row (two 16-byte
// loads), the grid stride is a multiple of c/8 so a thread keeps its channels (per-channel constants live in
// registers). Reductions are two-level and ordered (doubles throughout: per thread, then per block, then over blocks in a
// fixed order): no atomics, bitwise reproducible. The forward pass leaves one bit per element (output > 0), so the
// backward pass never re-reads the activation:
//   forward : stats (read x) . finalize . apply (read x [+ residual], write y + n*c/8 mask bytes)
//   backward: reduce (read dy, x, mask) . apply (read dy, x, mask, write dx [+ residual gradient])
#include "vx_common.h"

namespace {

constexpr int kBnThreads = 256;
constexpr int kBnMaxBlocks = 1024;
constexpr int kBnFinalThreads = 1024;

struct f8 {
  float4 a, b;
};
__device__ __forceinline__ f8 ld8(const float* p) {
  f8 v;
  v.a = *reinterpret_cast<const float4*>(p);
  v.b = *reinterpret_cast<const float4*>(p + 4);
  return v;
}
__device__ __forceinline__ void st8(float* p, const float (&v)[8]) {
  *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}
__device__ __forceinline__ void unpack(const f8& v, float (&o)[8]) {
  o[0] = v.a.x; o[1] = v.a.y; o[2] = v.a.z; o[3] = v.a.w;
  o[4] = v.b.x; o[5] = v.b.y; o[6] = v.b.z; o[7] = v.b.w;
}

// block-level ordered combine of per-thread (s[8], q[8]) -> part[blockIdx.x][2][c] (doubles)
__device__ __forceinline__ void block_combine(const double (&s)[8], const double (&q)[8], int c, int c8,
                                              double* __restrict__ part) {
  __shared__ double sh[kBnThreads][17];   // +1: threads of one channel group sit c8 rows apart
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    sh[threadIdx.x][e] = s[e];
    sh[threadIdx.x][8 + e] = q[e];
  }
  __syncthreads();
  const int per = kBnThreads / c8;        // threads per channel group
  for (int j = threadIdx.x; j < 2 * c; j += kBnThreads) {
    const int stat = j / c, ch = j % c, cg = ch >> 3, e = ch & 7;
    double acc = 0.0;
    for (int m = 0; m < per; ++m) acc += sh[cg + m * c8][stat * 8 + e];
    part[(long long)blockIdx.x * 2 * c + j] = acc;
  }
}

__global__ __launch_bounds__(kBnThreads) void bn_stats_partial(const float* __restrict__ x, long long items, int c,
                                                               int c8, double* __restrict__ part) {
  // sums in double from the first element on: E[x^2] - E[x]^2 cancels, and the stream is HBM-bound anyway
  double s[8] = {0, 0, 0, 0, 0, 0, 0, 0}, q[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const long long stride = (long long)gridDim.x * kBnThreads;
  for (long long it = (long long)blockIdx.x * kBnThreads + threadIdx.x; it < items; it += stride) {
    float v[8];
    unpack(ld8(x + it * 8), v);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const double d = (double)v[e];
      s[e] += d;
      q[e] += d * d;
    }
  }
  block_combine(s, q, c, c8, part);
}

// sums[j] = sum over blocks of part[g][j], fixed order; optional fp32 copy (parameter gradients).
// One workgroup per 32 columns: 32 x 32 threads, thread (col, split) adds rows split, split + 32, ... (256-byte
// coalesced row segments), then the 32 splits of a column are added in order.
__global__ __launch_bounds__(kBnFinalThreads) void bn_reduce_final(const double* __restrict__ part, int blocks, int c2,
                                                                   double* __restrict__ sums, float* __restrict__ sums_f32) {
  __shared__ double sh[32][33];
  const int col = threadIdx.x & 31, split = threadIdx.x >> 5;
  const int j = blockIdx.x * 32 + col;
  double acc = 0.0;
  if (j < c2)
    for (int g = split; g < blocks; g += 32) acc += part[(long long)g * c2 + j];
  sh[split][col] = acc;
  __syncthreads();
  if (split == 0 && j < c2) {
    double t = 0.0;
#pragma unroll
    for (int m = 0; m < 32; ++m) t += sh[m][col];
    sums[j] = t;
    if (sums_f32) sums_f32[j] = (float)t;
  }
}

__global__ void bn_set_count(double* __restrict__ dst, double n) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *dst = n;
}

// sums = {sum x [c], sum x^2 [c], count}: batch mean, 1/sqrt(biased var + eps), running statistics
__global__ void bn_finalize(const double* __restrict__ sums, int c, float eps, float momentum, float* __restrict__ running_mean,
                            float* __restrict__ running_var, float* __restrict__ mean, float* __restrict__ invstd) {
  const int ch = blockIdx.x * blockDim.x + threadIdx.x;
  if (ch >= c) return;
  const double n = sums[2 * c];
  if (n <= 0.0) {
    mean[ch] = 0.0f;
    invstd[ch] = 0.0f;
    return;
  }
  const double m = sums[ch] / n;
  double var = sums[c + ch] / n - m * m;
  if (var < 0.0) var = 0.0;
  mean[ch] = (float)m;
  invstd[ch] = (float)(1.0 / sqrt(var + (double)eps));
  if (running_mean) running_mean[ch] = (1.0f - momentum) * running_mean[ch] + momentum * (float)m;
  if (running_var) {
    const double unbiased = n > 1.0 ? var * n / (n - 1.0) : var;
    running_var[ch] = (1.0f - momentum) * running_var[ch] + momentum * (float)unbiased;
  }
}

template <bool RES, bool RELU>
__global__ __launch_bounds__(kBnThreads) void bn_apply(const float* __restrict__ x, long long items, int c8,
                                                       const float* __restrict__ mean, const float* __restrict__ invstd,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       const float* __restrict__ res, float* __restrict__ y,
                                                       unsigned char* __restrict__ mask) {
  const long long t0 = (long long)blockIdx.x * kBnThreads + threadIdx.x;
  const int cb = (int)(t0 % c8) * 8;
  float mu[8], sc[8], sh[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    mu[e] = mean[cb + e];
    sc[e] = invstd[cb + e] * (gamma ? gamma[cb + e] : 1.0f);
    sh[e] = beta ? beta[cb + e] : 0.0f;
  }
  const long long stride = (long long)gridDim.x * kBnThreads;
  for (long long it = t0; it < items; it += stride) {
    float v[8], r[8];
    unpack(ld8(x + it * 8), v);
    if (RES) unpack(ld8(res + it * 8), r);
    unsigned bits = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float o = (v[e] - mu[e]) * sc[e] + sh[e];
      if (RES) o += r[e];
      if (RELU) {
        bits |= (o > 0.0f ? 1u : 0u) << e;
        o = fmaxf(o, 0.0f);
      }
      v[e] = o;
    }
    st8(y + it * 8, v);
    if (RELU) mask[it] = (unsigned char)bits;
  }
}

// g = dy where the forward output was positive (all of dy without a mask): sum g, sum g * xhat per channel
__global__ __launch_bounds__(kBnThreads) void bn_bwd_partial(const float* __restrict__ dy, const float* __restrict__ x,
                                                             const unsigned char* __restrict__ mask, long long items, int c,
                                                             int c8, const float* __restrict__ mean,
                                                             const float* __restrict__ invstd, double* __restrict__ part) {
  const long long t0 = (long long)blockIdx.x * kBnThreads + threadIdx.x;
  const int cb = (int)(t0 % c8) * 8;
  float mu[8], is[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    mu[e] = mean[cb + e];
    is[e] = invstd[cb + e];
  }
  double s[8] = {0, 0, 0, 0, 0, 0, 0, 0}, q[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const long long stride = (long long)gridDim.x * kBnThreads;
  for (long long it = t0; it < items; it += stride) {
    float g[8], v[8];
    unpack(ld8(dy + it * 8), g);
    unpack(ld8(x + it * 8), v);
    const unsigned bits = mask ? mask[it] : 0xffu;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float ge = (bits >> e) & 1u ? g[e] : 0.0f;
      s[e] += (double)ge;
      q[e] += (double)ge * (double)((v[e] - mu[e]) * is[e]);
    }
  }
  block_combine(s, q, c, c8, part);
}

// dx = gamma * invstd * (g - mean(g) - xhat * mean(g * xhat)); the residual branch receives g
template <bool DRES>
__global__ __launch_bounds__(kBnThreads) void bn_bwd_apply(const float* __restrict__ dy, const float* __restrict__ x,
                                                           const unsigned char* __restrict__ mask, long long items, int c,
                                                           int c8, const float* __restrict__ mean,
                                                           const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                           const double* __restrict__ sums, const double* __restrict__ count,
                                                           float* __restrict__ dx, float* __restrict__ dres) {
  const long long t0 = (long long)blockIdx.x * kBnThreads + threadIdx.x;
  const int cb = (int)(t0 % c8) * 8;
  const double n = *count;
  float mu[8], is[8], k[8], mg[8], mgx[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    mu[e] = mean[cb + e];
    is[e] = invstd[cb + e];
    k[e] = is[e] * (gamma ? gamma[cb + e] : 1.0f);
    mg[e] = n > 0.0 ? (float)(sums[cb + e] / n) : 0.0f;
    mgx[e] = n > 0.0 ? (float)(sums[c + cb + e] / n) : 0.0f;
  }
  const long long stride = (long long)gridDim.x * kBnThreads;
  for (long long it = t0; it < items; it += stride) {
    float g[8], v[8];
    unpack(ld8(dy + it * 8), g);
    unpack(ld8(x + it * 8), v);
    const unsigned bits = mask ? mask[it] : 0xffu;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      g[e] = (bits >> e) & 1u ? g[e] : 0.0f;
      v[e] = k[e] * (g[e] - mg[e] - (v[e] - mu[e]) * is[e] * mgx[e]);
    }
    st8(dx + it * 8, v);
    if (DRES) st8(dres + it * 8, g);
  }
}

int bn_blocks(long long items) {
  long long b = vx_cdiv(items, kBnThreads);
  if (b < 1) b = 1;
  return (int)(b < kBnMaxBlocks ? b : kBnMaxBlocks);
}

// widths the thread mapping supports: c/8 a power of two that divides the block, 2c within the final block
bool bn_width_ok(int c) { return c >= 8 && c <= 512 && (c & (c - 1)) == 0; }

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" {

size_t vx_bn_workspace_bytes(int32_t c) { return c > 0 ? (size_t)kBnMaxBlocks * 2 * (size_t)c * sizeof(double) : 0; }

int32_t vx_bn_stats(const float* d_x, int64_t n, int32_t c, double* d_sums, void* d_workspace, size_t workspace_bytes,
                    void* stream) {
  VX_REQUIRE(bn_width_ok(c), "vx_bn_stats: c=%d must be a power of two in [8, 512]", c);
  VX_REQUIRE(n >= 0 && d_sums && (n == 0 || d_x), "vx_bn_stats: null argument");
  VX_REQUIRE(aligned16(d_x), "vx_bn_stats: x must be 16-byte aligned");
  if (workspace_bytes < vx_bn_workspace_bytes(c) || !d_workspace) {
    vx_set_error("vx_bn_stats: workspace of %zu bytes needed", vx_bn_workspace_bytes(c));
    return VX_ERR_WORKSPACE;
  }
  hipStream_t s = (hipStream_t)stream;
  const long long items = n * (c / 8);
  const int blocks = bn_blocks(items);
  double* part = (double*)d_workspace;
  bn_stats_partial<<<blocks, kBnThreads, 0, s>>>(d_x, items, c, c / 8, part);
  VX_LAUNCH_CHECK();
  bn_reduce_final<<<(2 * c + 31) / 32, kBnFinalThreads, 0, s>>>(part, blocks, 2 * c, d_sums, nullptr);
  VX_LAUNCH_CHECK();
  bn_set_count<<<1, 64, 0, s>>>(d_sums + 2 * c, (double)n);
  VX_LAUNCH_CHECK();
  return VX_OK;
}

int32_t vx_bn_finalize(const double* d_sums, int32_t c, float eps, float momentum, float* d_running_mean,
                       float* d_running_var, float* d_mean, float* d_invstd, void* stream) {
  VX_REQUIRE(c > 0 && d_sums && d_mean && d_invstd, "vx_bn_finalize: null argument");
  bn_finalize<<<(unsigned)vx_cdiv(c, 256), 256, 0, (hipStream_t)stream>>>(d_sums, c, eps, momentum, d_running_mean,
                                                                           d_running_var, d_mean, d_invstd);
  VX_LAUNCH_CHECK();
  return VX_OK;
}

int32_t vx_bn_apply(const float* d_x, int64_t n, int32_t c, const float* d_mean, const float* d_invstd,
                    const float* d_gamma, const float* d_beta, const float* d_residual, int32_t relu, float* d_y,
                    uint8_t* d_mask, void* stream) {
  VX_REQUIRE(bn_width_ok(c), "vx_bn_apply: c=%d must be a power of two in [8, 512]", c);
  VX_REQUIRE(n >= 0 && d_mean && d_invstd && (n == 0 || (d_x && d_y)), "vx_bn_apply: null argument");
  VX_REQUIRE(!relu || n == 0 || d_mask, "vx_bn_apply: relu needs the mask output (n*c/8 bytes)");
  VX_REQUIRE(aligned16(d_x) && aligned16(d_y) && aligned16(d_residual), "vx_bn_apply: tensors must be 16-byte aligned");
  if (n == 0) return VX_OK;
  hipStream_t s = (hipStream_t)stream;
  const long long items = n * (c / 8);
  const int blocks = (int)(vx_cdiv(items, kBnThreads) < 8192 ? vx_cdiv(items, kBnThreads) : 8192);
  const int c8 = c / 8;
#define VX_BN_APPLY(RES, RELU)                                                                                    \
  bn_apply<RES, RELU><<<blocks, kBnThreads, 0, s>>>(d_x, items, c8, d_mean, d_invstd, d_gamma, d_beta, d_residual, \
                                                    d_y, d_mask)
  if (d_residual) {
    if (relu) VX_BN_APPLY(true, true); else VX_BN_APPLY(true, false);
  } else {
    if (relu) VX_BN_APPLY(false, true); else VX_BN_APPLY(false, false);
  }
#undef VX_BN_APPLY
  VX_LAUNCH_CHECK();
  return VX_OK;
}

int32_t vx_bn_bwd_reduce(const float* d_dy, const float* d_x, const uint8_t* d_mask, int64_t n, int32_t c,
                         const float* d_mean, const float* d_invstd, double* d_sums, float* d_param_grads,
                         void* d_workspace, size_t workspace_bytes, void* stream) {
  VX_REQUIRE(bn_width_ok(c), "vx_bn_bwd_reduce: c=%d must be a power of two in [8, 512]", c);
  VX_REQUIRE(n >= 0 && d_sums && d_mean && d_invstd && (n == 0 || (d_dy && d_x)), "vx_bn_bwd_reduce: null argument");
  VX_REQUIRE(aligned16(d_dy) && aligned16(d_x), "vx_bn_bwd_reduce: tensors must be 16-byte aligned");
  if (workspace_bytes < vx_bn_workspace_bytes(c) || !d_workspace) {
    vx_set_error("vx_bn_bwd_reduce: workspace of %zu bytes needed", vx_bn_workspace_bytes(c));
    return VX_ERR_WORKSPACE;
  }
  hipStream_t s = (hipStream_t)stream;
  const long long items = n * (c / 8);
  const int blocks = bn_blocks(items);
  double* part = (double*)d_workspace;
  bn_bwd_partial<<<blocks, kBnThreads, 0, s>>>(d_dy, d_x, d_mask, items, c, c / 8, d_mean, d_invstd, part);
  VX_LAUNCH_CHECK();
  bn_reduce_final<<<(2 * c + 31) / 32, kBnFinalThreads, 0, s>>>(part, blocks, 2 * c, d_sums, d_param_grads);
  VX_LAUNCH_CHECK();
  return VX_OK;
}

int32_t vx_bn_bwd_apply(const float* d_dy, const float* d_x, const uint8_t* d_mask, int64_t n, int32_t c,
                        const float* d_mean, const float* d_invstd, const float* d_gamma, const double* d_sums,
                        const double* d_count, float* d_dx, float* d_dres, void* stream) {
  VX_REQUIRE(bn_width_ok(c), "vx_bn_bwd_apply: c=%d must be a power of two in [8, 512]", c);
  VX_REQUIRE(n >= 0 && d_sums && d_count && d_mean && d_invstd && (n == 0 || (d_dy && d_x && d_dx)),
             "vx_bn_bwd_apply: null argument");
  VX_REQUIRE(aligned16(d_dy) && aligned16(d_x) && aligned16(d_dx) && aligned16(d_dres),
             "vx_bn_bwd_apply: tensors must be 16-byte aligned");
  if (n == 0) return VX_OK;
  hipStream_t s = (hipStream_t)stream;
  const long long items = n * (c / 8);
  const int blocks = (int)(vx_cdiv(items, kBnThreads) < 8192 ? vx_cdiv(items, kBnThreads) : 8192);
  if (d_dres)
    bn_bwd_apply<true><<<blocks, kBnThreads, 0, s>>>(d_dy, d_x, d_mask, items, c, c / 8, d_mean, d_invstd, d_gamma, d_sums,
                                                     d_count, d_dx, d_dres);
  else
    bn_bwd_apply<false><<<blocks, kBnThreads, 0, s>>>(d_dy, d_x, d_mask, items, c, c / 8, d_mean, d_invstd, d_gamma, d_sums,
                                                      d_count, d_dx, d_dres);
  VX_LAUNCH_CHECK();
  return VX_OK;
}

}  // extern "C"
