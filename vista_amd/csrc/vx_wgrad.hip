// Sparse-conv weight gradient on gfx950 fp32 MFMA, atomic-free and bitwise reproducible.
//
//   dW[co][k][ci] = sum over the pairs (i, o) of kernel offset k of  dout[o][co] * in[i][ci]
//
// Pass 1: a workgroup takes one chunk of the COMPACTED pair list of one offset (exact FLOPs: no
// work on missing neighbours) and accumulates a [Cout x Cin] partial in MFMA accumulators:
// v_mfma_f32_32x32x2_f32 with A[co][p] = dout[out_p][co], B[p][ci] = in[in_p][ci]; both operands
// are row-major rows read straight from HBM/L2 with 128 contiguous bytes per half-wave (no LDS:
// nothing is reused inside a workgroup). The partial goes to a slab [k][chunk][co][ci].
// Pass 2: per (k, co, ci) the slabs are summed in chunk order.
#include "vx_common.h"

namespace {
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int kThreads = 256;
constexpr int kChunkMin = 2048;  // pairs per workgroup (see pick_chunk)

template <int TM, int TN>  // 32-row tiles along Cout / Cin
__global__ __launch_bounds__(kThreads) void wgrad_partial(const float* __restrict__ in, int cin,
                                                          const float* __restrict__ dout, int cout,
                                                          const int* __restrict__ pairs,
                                                          const int* __restrict__ pair_num,
                                                          long long n_out, int nchunks, int chunk_sz,
                                                          float* __restrict__ slabs) {
  constexpr int WM = TM < 2 ? TM : 2;
  constexpr int WN = (TN < 4 / WM) ? TN : 4 / WM;
  constexpr int PS = 4 / (WM * WN);   // waves that split the pair range
  constexpr int MT = TM / WM, NT = TN / WN;
  const int k = blockIdx.y, chunk = blockIdx.x;
  const int np = pair_num[k];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int ps = wave / (WM * WN), wq = wave % (WM * WN);
  const int wm = wq / WN, wn = wq % WN;
  const int sub = chunk_sz / PS;
  const int p_begin = chunk * chunk_sz + ps * sub;
  if ((long long)chunk * chunk_sz >= np) return;  // whole chunk empty: pass 2 never reads its slabs
  int p_end = p_begin + sub;
  if (p_end > np) p_end = np;
  const int* pin = pairs + (long long)k * 2 * n_out;
  const int* pout = pin + n_out;

  f32x16 acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[m][n][i] = 0.0f;

  bool cok[MT], nok[NT];
#pragma unroll
  for (int m = 0; m < MT; ++m) cok[m] = ((wm * MT + m) * 32 + r) < cout;
#pragma unroll
  for (int n = 0; n < NT; ++n) nok[n] = ((wn * NT + n) * 32 + r) < cin;

  for (int p0 = p_begin; p0 < p_end; p0 += 64) {
    int li = -1, lo = -1;
    if (p0 + lane < p_end) {
      li = pin[p0 + lane];
      lo = pout[p0 + lane];
    }
    const int cnt = (p_end - p0) < 64 ? (p_end - p0) : 64;
#pragma unroll 4
    for (int s = 0; s < 64; s += 2) {
      if (s >= cnt) break;
      const int ii = __shfl(li, s + h, 64), oo = __shfl(lo, s + h, 64);
      float av[MT], bv[NT];
#pragma unroll
      for (int m = 0; m < MT; ++m)
        av[m] = (oo >= 0 && cok[m]) ? dout[(long long)oo * cout + (wm * MT + m) * 32 + r] : 0.0f;
#pragma unroll
      for (int n = 0; n < NT; ++n)
        bv[n] = (ii >= 0 && nok[n]) ? in[(long long)ii * cin + (wn * NT + n) * 32 + r] : 0.0f;
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
          acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m], bv[n], acc[m][n], 0, 0, 0);
    }
  }
  float* slab = slabs + (((long long)k * nchunks + chunk) * PS + ps) * (long long)cout * cin;
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      const int ci = (wn * NT + n) * 32 + r;
      if (ci >= cin) continue;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int co = (wm * MT + m) * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
        if (co < cout) slab[(long long)co * cin + ci] = acc[m][n][i];
      }
    }
}

__global__ __launch_bounds__(kThreads) void wgrad_reduce(const float* __restrict__ slabs,
                                                         const int* __restrict__ pair_num,
                                                         int nchunks, int chunk_sz, int ps, int cout,
                                                         int cin, int kvol, float* __restrict__ dw) {
  const int k = blockIdx.y;
  const int e = blockIdx.x * kThreads + threadIdx.x;  // co * cin + ci
  if (e >= cout * cin) return;
  const int np = pair_num[k];
  const int used = (np + chunk_sz - 1) / chunk_sz;
  const long long stride = (long long)cout * cin;
  const float* s = slabs + (long long)k * nchunks * ps * stride + e;
  float sum = 0.0f;
  for (int c = 0; c < used * ps; ++c) sum += s[c * stride];
  const int co = e / cin, ci = e - co * cin;
  dw[((long long)co * kvol + k) * cin + ci] = sum;
}

// pairs per workgroup: each workgroup writes a Cout x Cin slab, so long lists use bigger chunks
// (fewer slabs to write and re-read) while keeping >= ~1024 workgroups in flight
inline int pick_chunk(int64_t n_out, int kvol) {
  // Measured on MI355X (FHD, batch 4): 8192-pair chunks cut the slab traffic 4x but leave only
  // ~800 four-wave workgroups for 256 CUs and the step got 25 % slower; parallelism wins.
  (void)n_out;
  (void)kvol;
  return kChunkMin;
}

inline int ps_of(int tm, int tn) {
  const int wm = tm < 2 ? tm : 2;
  const int wn = (tn < 4 / wm) ? tn : 4 / wm;
  return 4 / (wm * wn);
}

template <int TM, int TN>
void launch_partial(const float* in, int cin, const float* dout, int cout, const int* pairs,
                    const int* pair_num, long long n_out, int nchunks, int chunk_sz, int kvol,
                    float* slabs, hipStream_t stream) {
  dim3 grid((unsigned)nchunks, (unsigned)kvol);
  wgrad_partial<TM, TN><<<grid, kThreads, 0, stream>>>(in, cin, dout, cout, pairs, pair_num, n_out,
                                                       nchunks, chunk_sz, slabs);
}
}  // namespace

extern "C" size_t vx_spconv_wgrad_workspace_bytes(int32_t cin, int32_t cout, int32_t kvol,
                                                  int64_t n_out) {
  int tm = (cout + 31) / 32, tn = (cin + 31) / 32;
  if (tm == 3) tm = 4;
  if (tn == 3) tn = 4;
  const int64_t nchunks = vx_cdiv(n_out > 0 ? n_out : 1, pick_chunk(n_out, kvol));
  return (size_t)kvol * nchunks * ps_of(tm, tn) * cout * cin * sizeof(float) + 256;
}

extern "C" int32_t vx_spconv_wgrad(const void* d_in, int64_t n_in, int32_t cin, const void* d_dout,
                                   int64_t n_out, int32_t cout, const int32_t* d_pairs,
                                   const int32_t* d_pair_num, int32_t kvol, float* d_dw,
                                   int32_t dtype, void* d_ws, size_t ws_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  VX_REQUIRE(dtype == VX_F32, "spconv_wgrad: only VX_F32 is built in this revision");
  VX_REQUIRE(cin >= 1 && cin <= 128 && cout >= 1 && cout <= 128 && kvol >= 1 && kvol <= 125,
             "spconv_wgrad: need 1 <= cin, cout <= 128, 1 <= kvol <= 125");
  VX_REQUIRE(d_dw && n_in >= 0 && n_out >= 0, "spconv_wgrad: bad arguments");
  if (n_out == 0 || n_in == 0) {
    VX_HIP(hipMemsetAsync(d_dw, 0, sizeof(float) * (size_t)cout * kvol * cin, stream));
    return VX_OK;
  }
  VX_REQUIRE(d_in && d_dout && d_pairs && d_pair_num && d_ws, "spconv_wgrad: null pointer");
  if (ws_bytes < vx_spconv_wgrad_workspace_bytes(cin, cout, kvol, n_out)) {
    vx_set_error("spconv_wgrad: workspace too small");
    return VX_ERR_WORKSPACE;
  }
  int tm = (cout + 31) / 32, tn = (cin + 31) / 32;
  if (tm == 3) tm = 4;
  if (tn == 3) tn = 4;
  const int chunk_sz = pick_chunk(n_out, kvol);
  const int nchunks = (int)vx_cdiv(n_out, chunk_sz);
  const float* in = (const float*)d_in;
  const float* dout = (const float*)d_dout;
  float* slabs = (float*)d_ws;
#define VX_WG(TM, TN)                                                                          \
  if (tm == TM && tn == TN)                                                                    \
    launch_partial<TM, TN>(in, cin, dout, cout, d_pairs, d_pair_num, n_out, nchunks, chunk_sz, kvol, \
                           slabs, stream);
  VX_WG(1, 1) VX_WG(1, 2) VX_WG(1, 4)
  VX_WG(2, 1) VX_WG(2, 2) VX_WG(2, 4)
  VX_WG(4, 1) VX_WG(4, 2) VX_WG(4, 4)
#undef VX_WG
  VX_LAUNCH_CHECK();
  dim3 rgrid((unsigned)vx_cdiv((int64_t)cout * cin, kThreads), (unsigned)kvol);
  wgrad_reduce<<<rgrid, kThreads, 0, stream>>>(slabs, d_pair_num, nchunks, chunk_sz, ps_of(tm, tn),
                                               cout, cin, kvol, d_dw);
  VX_LAUNCH_CHECK();
  return VX_OK;
}
