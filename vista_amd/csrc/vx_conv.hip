// Sparse convolution on gfx950: output-stationary implicit GEMM on fp32 MFMA.
//
//   out[o] = epi( sum_k  W_k . in[nbr[k][o]] )         nbr[k][o] = -1 -> no contribution
//
// One 256-thread workgroup owns 128 consecutive output rows (32 per wave) and all (<=128) output
// channels; accumulators never leave registers, so there is no scatter, no atomic and the result
// is bitwise reproducible. Per kernel offset k the workgroup stages W_k ([Cout][Cin], the KRSC
// nn.Parameter layout) and each wave gathers its 32 input rows as whole rows (coalesced 16-byte
// lanes) into LDS; offsets no row of the tile uses are skipped wave- and workgroup-uniformly.
// MFMA: v_mfma_f32_32x32x2_f32, A = gathered rows, B = W_k^T. The reduction index is permuted so
// that lane half h covers channels [h*Cin/2, (h+1)*Cin/2): both fragments are then contiguous
// 16-byte LDS reads (ds_read_b128) from images padded to Cin+4 floats per row (conflict-free).
//
// The same kernel computes dgrad (roles of in/out swapped, W read transposed through strides).
#include "vx_common.h"

namespace {
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kThreads = 256;
constexpr int kTileRows = 128;  // rows per workgroup (32 per wave)

struct ConvArgs {
  const float* in;      // [n_in][cin]
  const float* w;       // element (oc, k, ic) at w[oc*w_so + k*w_sk + ic*w_si]
  long long w_so, w_sk, w_si;
  const int* nbr;       // [kvol][ld]
  long long ld;
  long long n_out;
  int cin, cout, kvol;
  int flip_k;           // table row used for weights offset k is (kvol-1-k)  (subm dgrad)
  const float* bias;    // f32[cout] or null
  const float* scale;   // f32[cout] or null
  const float* shift;   // f32[cout] or null
  const float* residual;  // [n_out][cout] or null
  int relu;
  float* out;           // [n_out][cout]
};

// bijective XCD-aware remap: consecutive tiles (which share gathered neighbours) run on one XCD
__device__ __forceinline__ unsigned xcd_remap(unsigned b, unsigned n) {
  const unsigned q = n >> 3, r = n & 7, x = b & 7;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
}

template <int CIN_PAD, int COUT_PAD>
__global__ __launch_bounds__(kThreads) void spconv_f32_kernel(ConvArgs a) {
  constexpr int CP = CIN_PAD + 4;        // padded row length (floats)
  constexpr int NT = COUT_PAD / 32;      // 32-column accumulator tiles per wave
  constexpr int C4 = CIN_PAD / 4;        // float4 per row
  constexpr int LPR = C4;                // lanes per gathered row
  constexpr int RPI = 64 / LPR;          // rows per wave-instruction
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sW = smem;                                  // [COUT_PAD][CP]
  float* sA = smem + COUT_PAD * CP + (threadIdx.x >> 6) * 32 * CP;  // this wave's [32][CP]

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const long long tile = xcd_remap(blockIdx.x, gridDim.x);
  const long long row0 = tile * kTileRows + wave * 32;
  const int col0 = blockIdx.y * COUT_PAD;
  const bool vec_in = (a.cin & 3) == 0;
  const bool vec_w = vec_in && a.w_si == 1 && (a.w_so & 3) == 0 && (a.w_sk & 3) == 0;

  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.0f;

  for (int k = 0; k < a.kvol; ++k) {
    const int krow = a.flip_k ? a.kvol - 1 - k : k;
    int myrow = -1;
    if (lane < 32 && row0 + lane < a.n_out) myrow = a.nbr[(long long)krow * a.ld + row0 + lane];
    const bool wave_any = __ballot(myrow >= 0) != 0ull;
    // also the WAR barrier protecting the LDS images of the previous offset
    if (!__syncthreads_or(wave_any ? 1 : 0)) continue;

    // ---- stage W_k : sW[oc][ic], zero padded -------------------------------------------------
    for (int idx = threadIdx.x; idx < COUT_PAD * C4; idx += kThreads) {
      const int oc = idx / C4, c4 = idx - oc * C4;
      const int goc = col0 + oc;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (goc < a.cout && c4 * 4 < a.cin) {
        const float* src = a.w + goc * a.w_so + k * a.w_sk + (long long)(c4 * 4) * a.w_si;
        if (vec_w) {
          v = *reinterpret_cast<const float4*>(src);
        } else {
          const int ic = c4 * 4;
          v.x = src[0];
          if (ic + 1 < a.cin) v.y = src[a.w_si];
          if (ic + 2 < a.cin) v.z = src[2 * a.w_si];
          if (ic + 3 < a.cin) v.w = src[3 * a.w_si];
        }
      }
      *reinterpret_cast<float4*>(sW + oc * CP + c4 * 4) = v;
    }
    // ---- gather this wave's 32 input rows (whole rows, coalesced) ------------------------------
    if (wave_any) {
#pragma unroll
      for (int r0 = 0; r0 < 32; r0 += RPI) {
        const int rr = r0 + lane / LPR, c4 = lane % LPR;
        const int src_row = __shfl(myrow, rr & 31, 64);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (rr < 32) {
          if (src_row >= 0 && c4 * 4 < a.cin) {
            const float* src = a.in + (long long)src_row * a.cin + c4 * 4;
            if (vec_in) {
              v = *reinterpret_cast<const float4*>(src);
            } else {
              const int ic = c4 * 4;
              v.x = src[0];
              if (ic + 1 < a.cin) v.y = src[1];
              if (ic + 2 < a.cin) v.z = src[2];
              if (ic + 3 < a.cin) v.w = src[3];
            }
          }
          *reinterpret_cast<float4*>(sA + rr * CP + c4 * 4) = v;
        }
      }
    }
    __syncthreads();
    // ---- MFMA over the reduction dimension ---------------------------------------------------
    if (wave_any) {
      const float* pa = sA + r * CP + h * (CIN_PAD / 2);
      const float* pb = sW + r * CP + h * (CIN_PAD / 2);
#pragma unroll 4
      for (int j = 0; j < CIN_PAD / 8; ++j) {
        const float4 av = *reinterpret_cast<const float4*>(pa + 4 * j);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const float4 bv = *reinterpret_cast<const float4*>(pb + t * 32 * CP + 4 * j);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, acc[t], 0, 0, 0);
        }
      }
    }
  }

  // ---- epilogue: bias, per-channel affine (folded BatchNorm), residual, ReLU ---------------------
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int col = col0 + t * 32 + r;
    if (col >= a.cout) continue;
    const float b = a.bias ? a.bias[col] : 0.0f;
    const float sc = a.scale ? a.scale[col] : 1.0f;
    const float sh = a.shift ? a.shift[col] : 0.0f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const long long row = row0 + (i & 3) + 8 * (i >> 2) + 4 * h;
      if (row >= a.n_out) continue;
      float v = acc[t][i] + b;
      if (a.scale || a.shift) v = v * sc + sh;
      if (a.residual) v += a.residual[row * a.cout + col];
      if (a.relu) v = fmaxf(v, 0.0f);
      a.out[row * a.cout + col] = v;
    }
  }
}

template <int CIN_PAD, int COUT_PAD>
int32_t launch(const ConvArgs& a, hipStream_t stream) {
  constexpr size_t lds = (size_t)(COUT_PAD + kTileRows) * (CIN_PAD + 4) * sizeof(float);
  static_assert(lds <= 160 * 1024, "LDS budget");
  auto kern = spconv_f32_kernel<CIN_PAD, COUT_PAD>;
  if (lds > 64 * 1024)
    VX_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)lds));
  dim3 grid((unsigned)vx_cdiv(a.n_out, kTileRows), (unsigned)vx_cdiv(a.cout, COUT_PAD));
  kern<<<grid, kThreads, lds, stream>>>(a);
  VX_LAUNCH_CHECK();
  return VX_OK;
}

template <int CIN_PAD>
int32_t dispatch_cout(const ConvArgs& a, hipStream_t s) {
  if (a.cout <= 32) return launch<CIN_PAD, 32>(a, s);
  if (a.cout <= 64) return launch<CIN_PAD, 64>(a, s);
  return launch<CIN_PAD, 128>(a, s);
}

int32_t dispatch(const ConvArgs& a, hipStream_t s) {
  if (a.cin <= 8) return dispatch_cout<8>(a, s);
  if (a.cin <= 16) return dispatch_cout<16>(a, s);
  if (a.cin <= 32) return dispatch_cout<32>(a, s);
  if (a.cin <= 64) return dispatch_cout<64>(a, s);
  return dispatch_cout<128>(a, s);
}
}  // namespace

extern "C" int32_t vx_spconv_fwd(const void* d_in, int64_t n_in, int32_t cin, const void* d_weight,
                                 int32_t kvol, int32_t cout, const int32_t* d_nbr, int64_t nbr_ld,
                                 int64_t n_out, const float* d_bias, const float* d_scale,
                                 const float* d_shift, const void* d_residual, int32_t flags,
                                 void* d_out, int32_t dtype, void* stream) {
  VX_REQUIRE(dtype == VX_F32, "spconv_fwd: only VX_F32 is built in this revision");
  VX_REQUIRE(cin >= 1 && cin <= 128 && cout >= 1 && kvol >= 1 && kvol <= 125,
             "spconv_fwd: need 1 <= cin <= 128, cout >= 1, 1 <= kvol <= 125");
  VX_REQUIRE(n_in >= 0 && n_out >= 0, "spconv_fwd: negative size");
  if (n_out == 0) return VX_OK;
  VX_REQUIRE(d_weight && d_nbr && d_out && nbr_ld >= n_out, "spconv_fwd: null pointer / stride");
  VX_REQUIRE(n_in == 0 || d_in, "spconv_fwd: null input");
  ConvArgs a;
  a.in = (const float*)d_in;
  a.w = (const float*)d_weight;
  a.w_so = (long long)kvol * cin;
  a.w_sk = cin;
  a.w_si = 1;
  a.nbr = d_nbr;
  a.ld = nbr_ld;
  a.n_out = n_out;
  a.cin = cin;
  a.cout = cout;
  a.kvol = kvol;
  a.flip_k = 0;
  a.bias = d_bias;
  a.scale = d_scale;
  a.shift = d_shift;
  a.residual = (const float*)d_residual;
  a.relu = (flags & VX_EPI_RELU) ? 1 : 0;
  a.out = (float*)d_out;
  return dispatch(a, (hipStream_t)stream);
}

extern "C" int32_t vx_spconv_dgrad(const void* d_dout, int64_t n_out, int32_t cout,
                                   const void* d_weight, int32_t kvol, int32_t cin,
                                   const int32_t* d_nbr_t, int64_t nbr_t_ld, int64_t n_in,
                                   int32_t subm, void* d_din, int32_t dtype, void* stream) {
  VX_REQUIRE(dtype == VX_F32, "spconv_dgrad: only VX_F32 is built in this revision");
  VX_REQUIRE(cout >= 1 && cout <= 128 && cin >= 1 && kvol >= 1 && kvol <= 125,
             "spconv_dgrad: need 1 <= cout <= 128, cin >= 1, 1 <= kvol <= 125");
  VX_REQUIRE(n_in >= 0 && n_out >= 0, "spconv_dgrad: negative size");
  if (n_in == 0) return VX_OK;
  VX_REQUIRE(d_weight && d_nbr_t && d_din && nbr_t_ld >= n_in, "spconv_dgrad: null pointer/stride");
  ConvArgs a;
  a.in = (const float*)d_dout;            // gathered operand: dout rows, cout channels
  a.w = (const float*)d_weight;           // W[co][k][ci] read as (oc = ci, ic = co)
  a.w_so = 1;
  a.w_sk = cin;
  a.w_si = (long long)kvol * cin;
  a.nbr = d_nbr_t;
  a.ld = nbr_t_ld;
  a.n_out = n_in;
  a.cin = cout;
  a.cout = cin;
  a.kvol = kvol;
  a.flip_k = subm ? 1 : 0;
  a.bias = a.scale = a.shift = a.residual = nullptr;
  a.relu = 0;
  a.out = (float*)d_din;
  return dispatch(a, (hipStream_t)stream);
}
