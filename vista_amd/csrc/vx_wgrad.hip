// Sparse-conv weight gradient on gfx950 fp32 MFMA, atomic-free and bitwise reproducible.
//
//   dW[co][k][ci] = sum over the pairs (i, o) of kernel offset k of  dout[o][co] * in[i][ci]
//
// Pass 1: a workgroup takes the pairs of ONE offset k whose output rows fall in ONE row range
// [chunk*R, chunk*R + R) of the COMPACTED pair list (exact FLOPs: no work on missing neighbours; the
// list is ordered by output row, so the range is found by two binary searches) and accumulates a
// [Cout x Cin] partial in MFMA accumulators: v_mfma_f32_32x32x2_f32 with A[co][p] = dout[out_p][co],
// B[p][ci] = in[in_p][ci]; both operands are row-major rows read with 128 contiguous bytes per
// half-wave (no LDS: nothing is reused inside a workgroup). The partial goes to a slab
// [chunk][k][co][ci]. Workgroup ids are dealt so that the kvol workgroups of one row range run
// together on one XCD: the same dout rows and (shifted) in rows are then read 27 times out of L2 /
// Infinity Cache instead of once per offset sweep out of HBM (the intensity of this product is
// fixed at Cout/4 flop per gathered byte, so where the bytes come from is what sets its speed).
// Pass 2: per (k, co, ci) the slabs are summed in chunk order.
#include "vx_common.h"
#include "vx_split.h"

namespace {
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int kThreads = 256;
constexpr int kChunkRows = 4096;  // output rows per workgroup row range

__device__ __forceinline__ unsigned xcd_remap(unsigned b, unsigned n) {
  const unsigned q = n >> 3, r = n & 7, x = b & 7;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
}

// first p in [0, n) with v[p] >= x (v ascending)
__device__ __forceinline__ int lower_bound_i32(const int* __restrict__ v, int n, int x) {
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (v[mid] < x) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// bounds[k][c] = first pair of offset k whose output row is >= c * kChunkRows (c = 0..nchunks)
__global__ __launch_bounds__(kThreads) void wgrad_bounds(const int* __restrict__ pairs,
                                                         const int* __restrict__ pair_num,
                                                         long long n_out, int nchunks, int kvol,
                                                         int* __restrict__ bounds) {
  const int e = blockIdx.x * kThreads + threadIdx.x;
  if (e >= kvol * (nchunks + 1)) return;
  const int k = e / (nchunks + 1), c = e - k * (nchunks + 1);
  const int* pout = pairs + ((long long)k * 2 + 1) * n_out;
  const long long row = (long long)c * kChunkRows;
  bounds[e] = row >= n_out ? pair_num[k] : lower_bound_i32(pout, pair_num[k], (int)row);
}

// Vector of N consecutive channels of one row (N = 1, 2, 4): one load per lane serves N MFMA tiles.
template <int N>
struct Vec { float v[N]; };

// Branch-free: the address is always valid (row / channel clamped), absent rows and channels past the
// row end are zeroed by a select. VEC: ld % N == 0 and c0 % N == 0, so the N channels are one aligned load.
template <int N, bool VEC>
__device__ __forceinline__ Vec<N> load_channels(const float* __restrict__ base, int row, int ld, int c0) {
  Vec<N> out;
  const bool row_ok = row >= 0;
  const float* p = base + (long long)(row_ok ? row : 0) * ld;
  if constexpr (VEC) {
    const bool ok = row_ok && c0 < ld;
    p += c0 < ld ? c0 : 0;
    if constexpr (N == 4) {
      const float4 t = *reinterpret_cast<const float4*>(p);
      out.v[0] = ok ? t.x : 0.0f; out.v[1] = ok ? t.y : 0.0f; out.v[2] = ok ? t.z : 0.0f; out.v[3] = ok ? t.w : 0.0f;
    } else if constexpr (N == 2) {
      const float2 t = *reinterpret_cast<const float2*>(p);
      out.v[0] = ok ? t.x : 0.0f; out.v[1] = ok ? t.y : 0.0f;
    } else {
      const float t = *p;
      out.v[0] = ok ? t : 0.0f;
    }
  } else {
#pragma unroll
    for (int j = 0; j < N; ++j) {
      const int c = c0 + j;
      const float t = p[c < ld ? c : 0];
      out.v[j] = (row_ok && c < ld) ? t : 0.0f;
    }
  }
  return out;
}

// TM / TN = number of 32-channel MFMA tiles along Cout / Cin. Tile t holds the channels {T*r + t}: a
// lane's T consecutive channels feed its T tiles, so ONE 4/8/16-byte load per lane (a whole row per
// half-wave) supplies an operand for every tile. A wave owns all TM Cout tiles and NT = TN/WN Cin
// tiles; PS = 4/WN waves split the pair range. Operands are fetched one batch (U two-pair steps)
// ahead of the MFMAs that use them.
template <int TM, int TN, bool VEC>
__global__ __launch_bounds__(kThreads, 2) void wgrad_partial(const float* __restrict__ in, int cin,
                                                          const float* __restrict__ dout, int cout,
                                                          const int* __restrict__ pairs,
                                                          const int* __restrict__ bounds,
                                                          long long n_out, int nchunks, int kvol,
                                                          float* __restrict__ slabs) {
  constexpr int WN = (TN >= 2 && TM * TN >= 8) ? 2 : 1;
  constexpr int NT = TN / WN;
  constexpr int PS = 4 / WN;
  constexpr int U = (TM * NT >= 4) ? 4 : 8;     // two-pair steps per batch
  const unsigned wg = xcd_remap(blockIdx.x, gridDim.x);
  const int k = (int)(wg % (unsigned)kvol), chunk = (int)(wg / (unsigned)kvol);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int ps = wave / WN, wn = wave % WN;
  const int* pin = pairs + (long long)k * 2 * n_out;
  const int* pout = pin + n_out;
  const int p_lo = bounds[k * (nchunks + 1) + chunk], p_hi = bounds[k * (nchunks + 1) + chunk + 1];
  const int sub = ((p_hi - p_lo + PS - 1) / PS + 1) & ~1;   // even: a step consumes two pairs
  const int p_begin = p_lo + ps * sub;
  int p_end = p_begin + sub;
  if (p_end > p_hi) p_end = p_hi;
  const int nb = p_end > p_begin ? (p_end - p_begin + 2 * U - 1) / (2 * U) : 0;   // batches of 2U pairs

  const int a_c0 = TM * r, b_c0 = TN * r + wn * NT;

  f32x16 acc[TM][NT];
#pragma unroll
  for (int m = 0; m < TM; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[m][n][i] = 0.0f;

  // pair ids of a batch: lane l < 2U holds pair (batch start + l); -1 past the end
  auto load_ids = [&](int bt, int& li, int& lo) {
    const int p = p_begin + bt * 2 * U + lane;
    li = lo = -1;
    if (bt < nb && lane < 2 * U && p < p_end) {
      li = pin[p];
      lo = pout[p];
    }
  };
  auto fetch = [&](int li, int lo, Vec<TM>(&av)[U], Vec<NT>(&bv)[U]) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int ii = __shfl(li, 2 * u + h, 64), oo = __shfl(lo, 2 * u + h, 64);
      av[u] = load_channels<TM, VEC>(dout, oo, cout, a_c0);
      bv[u] = load_channels<NT, VEC>(in, ii, cin, b_c0);
    }
  };
  auto mma = [&](const Vec<TM>(&av)[U], const Vec<NT>(&bv)[U]) {
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
          acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u].v[m], bv[u].v[n], acc[m][n], 0, 0, 0);
  };

  Vec<TM> a0[U], a1[U];
  Vec<NT> b0[U], b1[U];
  int li0, lo0, li1, lo1;
  load_ids(0, li0, lo0);
  load_ids(1, li1, lo1);
  fetch(li0, lo0, a0, b0);
  // two batches per trip (ping-pong register sets); ids past the end are -1: zero operands, no loads
  for (int bt = 0; bt < nb; bt += 2) {
    load_ids(bt + 2, li0, lo0);
    fetch(li1, lo1, a1, b1);
    mma(a0, b0);
    load_ids(bt + 3, li1, lo1);
    fetch(li0, lo0, a0, b0);
    mma(a1, b1);
  }

  // combine the PS partials of the workgroup in a fixed order (ps = 0 adds 1, 2, ... in turn): lane-local
  // exchange through LDS, one slab per (chunk, k)
  __shared__ float part[(PS - 1) * WN * TM * NT * 16 * 64];
  static_assert(sizeof(part) <= 64 * 1024, "static LDS");
  if (ps > 0) {
    float* dst = part + ((ps - 1) * WN + wn) * (TM * NT * 16 * 64) + lane;
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int i = 0; i < 16; ++i) dst[((m * NT + n) * 16 + i) * 64] = acc[m][n][i];
  }
  __syncthreads();
  if (ps > 0) return;
#pragma unroll
  for (int q = 0; q < PS - 1; ++q) {
    const float* src = part + (q * WN + wn) * (TM * NT * 16 * 64) + lane;
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[m][n][i] += src[((m * NT + n) * 16 + i) * 64];
  }
  float* slab = slabs + ((long long)chunk * kvol + k) * (long long)cout * cin;
#pragma unroll
  for (int m = 0; m < TM; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      const int ci = b_c0 + n;
      if (ci >= cin) continue;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int co = TM * ((i & 3) + 8 * (i >> 2) + 4 * h) + m;
        if (co < cout) slab[(long long)co * cin + ci] = acc[m][n][i];
      }
    }
}

// The same pass on the bf16 matrix pipe (split arithmetic of spconv_w8s, vx_conv.hip): every fp32 operand is the exact
// sum of three bf16 planes, NP plane products per fp32 product on v_mfma_f32_32x32x16_bf16, fp32 accumulation. One
// MFMA reduces over 16 pairs: lane (r, h) holds its channel of pairs {2 i + h, i = 0..7} of a 16-pair batch (eight
// row loads, as many as the fp32 form spends on 16 pairs), splits the eight values of each of its channels and feeds
// them as one k-group. A wave owns at most 2 x 2 tiles (the split costs VALU work per operand value, so operands
// per MFMA is what to minimise: 32 values for 24 or 36 MFMAs); WM x WN waves tile Cout x Cin, PS = 4 / (WM WN) waves
// share the pair range and are combined in a fixed order.
template <int TM, int TN, bool VEC, int NP>
__global__ __launch_bounds__(kThreads, 2) void wgrad_partial_s(const float* __restrict__ in, int cin,
                                                            const float* __restrict__ dout, int cout,
                                                            const int* __restrict__ pairs,
                                                            const int* __restrict__ bounds,
                                                            long long n_out, int nchunks, int kvol,
                                                            float* __restrict__ slabs) {
  static_assert(NP == 6 || NP == 9, "six or nine plane products");
  constexpr int MW = TM >= 2 ? 2 : 1, NW = TN >= 2 ? 2 : 1;   // tiles per wave along Cout / Cin
  constexpr int WM = TM / MW, WN = TN / NW;                   // waves along Cout / Cin
  static_assert(WM * WN <= 4 && 4 % (WM * WN) == 0, "wave grid");
  constexpr int PS = 4 / (WM * WN);
  constexpr int B = 16;                                       // pairs per batch = one MFMA's reduction
  const unsigned wg = xcd_remap(blockIdx.x, gridDim.x);
  const int k = (int)(wg % (unsigned)kvol), chunk = (int)(wg / (unsigned)kvol);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int ps = wave / (WM * WN), wm = (wave % (WM * WN)) / WN, wn = wave % WN;
  const int* pin = pairs + (long long)k * 2 * n_out;
  const int* pout = pin + n_out;
  const int p_lo = bounds[k * (nchunks + 1) + chunk], p_hi = bounds[k * (nchunks + 1) + chunk + 1];
  const int sub = (p_hi - p_lo + PS - 1) / PS;
  const int p_begin = p_lo + ps * sub;
  int p_end = p_begin + sub;
  if (p_end > p_hi) p_end = p_hi;
  const int nb = p_end > p_begin ? (p_end - p_begin + B - 1) / B : 0;

  const int a_c0 = TM * r + wm * MW, b_c0 = TN * r + wn * NW;

  f32x16 acc[MW][NW];
#pragma unroll
  for (int m = 0; m < MW; ++m)
#pragma unroll
    for (int n = 0; n < NW; ++n)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[m][n][i] = 0.0f;

  auto load_ids = [&](int bt, int& li, int& lo) {     // lane l < 16 holds pair (batch start + l); -1 past the end
    const int p = p_begin + bt * B + lane;
    li = lo = -1;
    if (bt < nb && lane < B && p < p_end) {
      li = pin[p];
      lo = pout[p];
    }
  };
  auto fetch = [&](int li, int lo, Vec<MW>(&av)[8], Vec<NW>(&bv)[8]) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int ii = __shfl(li, 2 * u + h, 64), oo = __shfl(lo, 2 * u + h, 64);
      av[u] = load_channels<MW, VEC>(dout, oo, cout, a_c0);
      bv[u] = load_channels<NW, VEC>(in, ii, cin, b_c0);
    }
  };
  auto mma = [&](const Vec<MW>(&av)[8], const Vec<NW>(&bv)[8]) {
    Planes A[MW], Bp[NW];
#pragma unroll
    for (int m = 0; m < MW; ++m) {
      const float x[8] = {av[0].v[m], av[1].v[m], av[2].v[m], av[3].v[m], av[4].v[m], av[5].v[m], av[6].v[m], av[7].v[m]};
      A[m] = split8s(x);
    }
#pragma unroll
    for (int n = 0; n < NW; ++n) {
      const float x[8] = {bv[0].v[n], bv[1].v[n], bv[2].v[n], bv[3].v[n], bv[4].v[n], bv[5].v[n], bv[6].v[n], bv[7].v[n]};
      Bp[n] = split8s(x);
    }
#define VX_WMMA(PA, PB)                                                                              \
  _Pragma("unroll") for (int m = 0; m < MW; ++m) _Pragma("unroll") for (int n = 0; n < NW; ++n)      \
      acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[m].PA, Bp[n].PB, acc[m][n], 0, 0, 0)
    if constexpr (NP == 9) {
      VX_WMMA(l, l);
      VX_WMMA(m, l);
      VX_WMMA(l, m);
    }
    VX_WMMA(m, m);
    VX_WMMA(h, l);
    VX_WMMA(l, h);
    VX_WMMA(h, m);
    VX_WMMA(m, h);
    VX_WMMA(h, h);
#undef VX_WMMA
  };

  Vec<MW> a0[8], a1[8];
  Vec<NW> b0[8], b1[8];
  int li0, lo0, li1, lo1;
  load_ids(0, li0, lo0);
  load_ids(1, li1, lo1);
  fetch(li0, lo0, a0, b0);
  for (int bt = 0; bt < nb; bt += 2) {
    load_ids(bt + 2, li0, lo0);
    fetch(li1, lo1, a1, b1);
    mma(a0, b0);
    load_ids(bt + 3, li1, lo1);
    fetch(li0, lo0, a0, b0);
    mma(a1, b1);
  }

  constexpr int WT = MW * NW * 16 * 64;              // floats of one wave's accumulators
  __shared__ float part[PS > 1 ? (PS - 1) * WM * WN * WT : 1];
  static_assert(sizeof(part) <= 64 * 1024, "static LDS");
  if constexpr (PS > 1) {
    if (ps > 0) {
      float* dst = part + ((ps - 1) * WM * WN + wm * WN + wn) * WT + lane;
#pragma unroll
      for (int m = 0; m < MW; ++m)
#pragma unroll
        for (int n = 0; n < NW; ++n)
#pragma unroll
          for (int i = 0; i < 16; ++i) dst[((m * NW + n) * 16 + i) * 64] = acc[m][n][i];
    }
    __syncthreads();
    if (ps > 0) return;
#pragma unroll 1
    for (int q = 0; q < PS - 1; ++q) {
      const float* src = part + (q * WM * WN + wm * WN + wn) * WT + lane;
#pragma unroll
      for (int m = 0; m < MW; ++m)
#pragma unroll
        for (int n = 0; n < NW; ++n)
#pragma unroll
          for (int i = 0; i < 16; ++i) acc[m][n][i] += src[((m * NW + n) * 16 + i) * 64];
    }
  }
  float* slab = slabs + ((long long)chunk * kvol + k) * (long long)cout * cin;
#pragma unroll
  for (int m = 0; m < MW; ++m)
#pragma unroll
    for (int n = 0; n < NW; ++n) {
      const int ci = b_c0 + n;
      if (ci >= cin) continue;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int co = TM * ((i & 3) + 8 * (i >> 2) + 4 * h) + wm * MW + m;
        if (co < cout) slab[(long long)co * cin + ci] = acc[m][n][i];
      }
    }
}

__global__ __launch_bounds__(kThreads) void wgrad_reduce(const float* __restrict__ slabs, int nchunks,
                                                         int cout, int cin, int kvol,
                                                         float* __restrict__ dw) {
  const int k = blockIdx.y;
  const int e = blockIdx.x * kThreads + threadIdx.x;  // co * cin + ci
  if (e >= cout * cin) return;
  const long long stride = (long long)cout * cin;
  float sum = 0.0f;
  for (int c = 0; c < nchunks; ++c) sum += slabs[((long long)c * kvol + k) * stride + e];
  const int co = e / cin, ci = e - co * cin;
  dw[((long long)co * kvol + k) * cin + ci] = sum;
}

template <int TM, int TN, int NP>
void launch_partial_s(const float* in, int cin, const float* dout, int cout, const int* pairs,
                      const int* bounds, long long n_out, int nchunks, int kvol, float* slabs,
                      hipStream_t stream) {
  constexpr int MW = TM >= 2 ? 2 : 1, NW = TN >= 2 ? 2 : 1;      // as in the kernel
  const dim3 grid((unsigned)(nchunks * kvol));
  if (cout % TM == 0 && cin % TN == 0 && cout % MW == 0 && cin % NW == 0)
    wgrad_partial_s<TM, TN, true, NP><<<grid, kThreads, 0, stream>>>(in, cin, dout, cout, pairs, bounds, n_out,
                                                                     nchunks, kvol, slabs);
  else
    wgrad_partial_s<TM, TN, false, NP><<<grid, kThreads, 0, stream>>>(in, cin, dout, cout, pairs, bounds, n_out,
                                                                      nchunks, kvol, slabs);
}

template <int TM, int TN>
void launch_partial(const float* in, int cin, const float* dout, int cout, const int* pairs,
                    const int* bounds, long long n_out, int nchunks, int kvol, float* slabs,
                    hipStream_t stream, int products) {
  // split arithmetic where the channels fill the MFMA tiles (the fp32 form keeps the ragged widths)
  if (products && cout % 32 == 0 && cin % 32 == 0) {
    if (products == 6)
      return launch_partial_s<TM, TN, 6>(in, cin, dout, cout, pairs, bounds, n_out, nchunks, kvol, slabs, stream);
    return launch_partial_s<TM, TN, 9>(in, cin, dout, cout, pairs, bounds, n_out, nchunks, kvol, slabs, stream);
  }
  constexpr int NT = TN / ((TN >= 2 && TM * TN >= 8) ? 2 : 1);   // as in the kernel
  const dim3 grid((unsigned)(nchunks * kvol));
  if (cout % TM == 0 && cin % NT == 0)
    wgrad_partial<TM, TN, true><<<grid, kThreads, 0, stream>>>(in, cin, dout, cout, pairs, bounds, n_out,
                                                               nchunks, kvol, slabs);
  else
    wgrad_partial<TM, TN, false><<<grid, kThreads, 0, stream>>>(in, cin, dout, cout, pairs, bounds, n_out,
                                                                nchunks, kvol, slabs);
}
}  // namespace

int vx_resolve_f32_products(int32_t flags);   // vx_conv.hip: the call's VX_CONV_F32_* flag, else the process default

extern "C" size_t vx_spconv_wgrad_workspace_bytes(int32_t cin, int32_t cout, int32_t kvol,
                                                  int64_t n_out) {
  int tm = (cout + 31) / 32, tn = (cin + 31) / 32;
  if (tm == 3) tm = 4;
  if (tn == 3) tn = 4;
  const int64_t nchunks = vx_cdiv(n_out > 0 ? n_out : 1, kChunkRows);
  (void)tm;
  (void)tn;
  return (size_t)kvol * nchunks * cout * cin * sizeof(float) + (size_t)kvol * (nchunks + 1) * sizeof(int) + 512;
}

extern "C" int32_t vx_spconv_wgrad(const void* d_in, int64_t n_in, int32_t cin, const void* d_dout,
                                   int64_t n_out, int32_t cout, const int32_t* d_pairs,
                                   const int32_t* d_pair_num, int32_t kvol, float* d_dw,
                                   int32_t dtype, int32_t flags, void* d_ws, size_t ws_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  VX_REQUIRE(dtype == VX_F32, "spconv_wgrad: only VX_F32 is built in this revision");
  VX_REQUIRE(cin >= 1 && cin <= 128 && cout >= 1 && cout <= 128 && kvol >= 1 && kvol <= 125,
             "spconv_wgrad: need 1 <= cin, cout <= 128, 1 <= kvol <= 125");
  VX_REQUIRE(d_dw && n_in >= 0 && n_out >= 0, "spconv_wgrad: bad arguments");
  if (n_out == 0 || n_in == 0) {
    return vx_fill_async(d_dw, 0, sizeof(float) * (size_t)cout * kvol * cin, stream);
  }
  VX_REQUIRE(d_in && d_dout && d_pairs && d_pair_num && d_ws, "spconv_wgrad: null pointer");
  if (ws_bytes < vx_spconv_wgrad_workspace_bytes(cin, cout, kvol, n_out)) {
    vx_set_error("spconv_wgrad: workspace too small");
    return VX_ERR_WORKSPACE;
  }
  int tm = (cout + 31) / 32, tn = (cin + 31) / 32;
  if (tm == 3) tm = 4;
  if (tn == 3) tn = 4;
  VX_REQUIRE(n_out < (1ll << 31) && n_in < (1ll << 31), "spconv_wgrad: row ids must fit int32");
  const int products = vx_resolve_f32_products(flags);
  const int nchunks = (int)vx_cdiv(n_out, kChunkRows);
  const float* in = (const float*)d_in;
  const float* dout = (const float*)d_dout;
  float* slabs = (float*)d_ws;
  int* bounds = (int*)((char*)d_ws + (((size_t)kvol * nchunks * cout * cin * sizeof(float) + 255) & ~(size_t)255));
  wgrad_bounds<<<(unsigned)vx_cdiv((int64_t)kvol * (nchunks + 1), kThreads), kThreads, 0, stream>>>(
      d_pairs, d_pair_num, n_out, nchunks, kvol, bounds);
  VX_LAUNCH_CHECK();
#define VX_WG(TM, TN)                                                                          \
  if (tm == TM && tn == TN)                                                                    \
    launch_partial<TM, TN>(in, cin, dout, cout, d_pairs, bounds, n_out, nchunks, kvol, slabs,      \
                           stream, products);
  VX_WG(1, 1) VX_WG(1, 2) VX_WG(1, 4)
  VX_WG(2, 1) VX_WG(2, 2) VX_WG(2, 4)
  VX_WG(4, 1) VX_WG(4, 2) VX_WG(4, 4)
#undef VX_WG
  VX_LAUNCH_CHECK();
  dim3 rgrid((unsigned)vx_cdiv((int64_t)cout * cin, kThreads), (unsigned)kvol);
  wgrad_reduce<<<rgrid, kThreads, 0, stream>>>(slabs, nchunks, cout, cin, kvol, d_dw);
  VX_LAUNCH_CHECK();
  return VX_OK;
}
