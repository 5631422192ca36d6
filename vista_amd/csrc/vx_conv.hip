// Sparse convolution on gfx950: output-stationary implicit GEMM on fp32 MFMA.
//
//   out[o] = epi( sum_k  W_k . in[nbr[k][o]] )         nbr[k][o] = -1 -> no contribution
//
// Accumulators never leave registers: no scatter, no atomics, bitwise reproducible. MFMA is
// v_mfma_f32_32x32x2_f32 (A = gathered rows, B = W_k^T); the reduction index is permuted so that
// lane half h covers a contiguous half of the staged values, which makes every fragment a
// contiguous 16-byte access. Two kernels:
//   spconv_f32_w8      the hot path (cin in {8,16,32,64,128}, KRSC weights, kvol <= 32)
//   spconv_f32_generic any cin <= 128 / strided weights (first layer cin=5): simple, LDS staged
// dgrad runs on the same kernels with the roles of in/out swapped and a transposed weight copy.
#include <atomic>
#include <mutex>
#include <set>
#include <type_traits>
#include <utility>

#include "vx_common.h"
#include "vx_split.h"

namespace {
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

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
  int dtype;            // VX_F32 | VX_BF16 (element type of in / w / residual / out)
  const int* perm;      // optional: tile row j computes output row perm[j] (table is pre-permuted)
  const void* zero;     // 512 zero bytes in device memory: what a missing neighbour reads
  int no_split;         // launch spconv_w8's plain grid (VX_CONV_NO_SPLIT_TAIL)
  // "_dn" entry points (row count resident on the device, e.g. under HIP graph capture): when non-null the kernels
  // take min(*n_out_dev, n_out) as the row count, n_out is only the capacity the grid was sized for, and surplus
  // workgroups exit at once. Tile -> row mapping, split tails and every output row's summation order are the same
  // functions of the row count as on the host, so results are bitwise those of the host-count launch.
  const long long* n_out_dev;
  int cus;              // compute units of the device (tail-splitting decisions taken on the device)
  int f32_products;     // fp32 layers with Cin >= 64 on the row-stationary kernel: 0 = spconv_w8 (fp32 MFMA),
                        // 6 / 9 = spconv_w8s (split-bf16 plane products)
};

__device__ __forceinline__ long long rows_of(const ConvArgs& a) {
  long long n = a.n_out;
  if (a.n_out_dev) {
    const long long d = *a.n_out_dev;
    n = d < n ? (d > 0 ? d : 0) : n;
  }
  return n;
}

// bijective XCD-aware remap: consecutive tiles (which share gathered neighbours) run on one XCD
__device__ __forceinline__ unsigned xcd_remap(unsigned b, unsigned n) {
  const unsigned q = n >> 3, r = n & 7, x = b & 7;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
}

template <int CIN_PAD, int COUT_PAD>
__global__ __launch_bounds__(kThreads) void spconv_f32_generic(ConvArgs a) {
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
  const long long n_out = rows_of(a);
  const long long ntiles = (n_out + kTileRows - 1) / kTileRows;
  if ((long long)blockIdx.x >= ntiles) return;
  const long long tile = xcd_remap(blockIdx.x, (unsigned)ntiles);
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
    if (lane < 32 && row0 + lane < n_out) myrow = a.nbr[(long long)krow * a.ld + row0 + lane];
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
      if (row >= n_out) continue;
      const long long orow = a.perm ? a.perm[row] : row;
      float v = acc[t][i] + b;
      if (a.scale || a.shift) v = v * sc + sh;
      if (a.residual) v += a.residual[orow * a.cout + col];
      if (a.relu) v = fmaxf(v, 0.0f);
      a.out[orow * a.cout + col] = v;
    }
  }
}


#ifndef VX_CB_SPLIT
#define VX_CB_SPLIT 1   // split-bf16 arithmetic in the 32-pair compaction kernel (fp32 layers with Cin 16 / 32 / 64)
#endif
#ifndef VX_W8S_S16
#define VX_W8S_S16 1   // MFMA shape of spconv_w8s: 1 = 16x16x32, 0 = 32x32x16
#endif
#ifndef VX_F32_PRODUCTS_DEFAULT
#define VX_F32_PRODUCTS_DEFAULT 6
#endif
constexpr int kKSlots = 28;  // index-table rows kept in LDS (kvol <= 28)

// ------------------------------------------------------------------------------------------------
// 8-wave kernel (hot path), fp32 and bf16.
// 512 threads own 256 output rows; two waves share every SIMD so one wave's loads, address
// arithmetic, LDS traffic and barrier waits hide under its partner's MFMAs.
// The reduction runs in chunks of 256 staged BYTES per row (64 fp32 / 128 bf16 values; C=128 fp32:
// half a kernel offset ... C=16 bf16: 8 offsets). Per chunk a lane holds 128 contiguous bytes of
// its row's neighbours in 8 VGPR quads, loaded straight from HBM/L2 one chunk ahead (ping-pong
// register sets, no LDS for the gathered operand); missing neighbours read a zero row, so no
// select is needed. The W chunk ([Cout][256 B], rows padded to 272 B: conflict-free ds_read_b128)
// is double buffered in LDS: one barrier per chunk. A 16-byte quad feeds 4 v_mfma_f32_32x32x2_f32
// (fp32) or 1 v_mfma_f32_32x32x16_bf16 (bf16); accumulation is fp32 in both.
// ------------------------------------------------------------------------------------------------
__device__ __attribute__((aligned(256))) float vx_zero_row[128];  // zero-initialised, 512 B


constexpr int kW8Threads = 512;
constexpr int kW8Rows = 256;
constexpr int kRowBytes = 256;          // staged bytes per row per chunk

template <typename T>
struct Elem;
template <>
struct Elem<float> {
  static constexpr int VPQ = 4;  // values per 16-byte quad
  template <int NT>
  static __device__ __forceinline__ void mma(const f32x4& av, const f32x4 (&bv)[NT], f32x16 (&acc)[NT]) {
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv[t].x, acc[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv[t].y, acc[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv[t].z, acc[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv[t].w, acc[t], 0, 0, 0);
  }
  static __device__ __forceinline__ float load(const float* p) { return *p; }
  static __device__ __forceinline__ void store(float* p, float v) { *p = v; }
};
template <>
struct Elem<__bf16> {
  static constexpr int VPQ = 8;
  template <int NT>
  static __device__ __forceinline__ void mma(const f32x4& av, const f32x4 (&bv)[NT], f32x16 (&acc)[NT]) {
    const bf16x8 a8 = __builtin_bit_cast(bf16x8, av);
#pragma unroll
    for (int t = 0; t < NT; ++t)
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, __builtin_bit_cast(bf16x8, bv[t]), acc[t], 0, 0, 0);
  }
  static __device__ __forceinline__ float load(const __bf16* p) { return (float)*p; }
  static __device__ __forceinline__ void store(__bf16* p, float v) { *p = (__bf16)v; }
};

#ifdef VX_STAMP  // diagnostic build only (tools/stamp_probe.py): where does a pipeline step spend its cycles
__device__ unsigned long long vx_dbg[8];
#define VX_STAMP_T(x)                                                                  \
  unsigned long long x;                                                                \
  __builtin_amdgcn_sched_barrier(0);                                                   \
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(x)::"memory");            \
  __builtin_amdgcn_sched_barrier(0)
#define VX_STAMP_ACC(i, v) stamp_sum[i] += (v)
#define VX_STAMP_DECL unsigned long long stamp_sum[6] = {0, 0, 0, 0, 0, 0}
#define VX_STAMP_FLUSH                                                                 \
  if (lane == 0)                                                                       \
    for (int i_ = 0; i_ < 6; ++i_) atomicAdd(&vx_dbg[i_], stamp_sum[i_])
extern "C" int32_t vx_debug_read(unsigned long long* h8) {
  if (hipMemcpyFromSymbol(h8, HIP_SYMBOL(vx_dbg), 64) != hipSuccess) return -1;
  unsigned long long z[8] = {0};
  return hipMemcpyToSymbol(HIP_SYMBOL(vx_dbg), z, 64) == hipSuccess ? 0 : -1;
}
#else
#define VX_STAMP_T(x)
#define VX_STAMP_ACC(i, v)
#define VX_STAMP_DECL
#define VX_STAMP_FLUSH
#endif

// Body of the LDS-staged row-stationary kernel for row tile `tile` and output columns [col0, col0 + COUT_PAD).
template <typename T, int CIN, int COUT_PAD>
__device__ __forceinline__ void w8_body(const ConvArgs& a, const long long n_out, const long long tile, const int col0) {
  constexpr int VPQ = Elem<T>::VPQ;
  constexpr int CK = kRowBytes / (int)sizeof(T);      // staged values per chunk (64 / 128)
  constexpr int QPR = kRowBytes / 16;                 // 16-byte pieces per staged row (16)
  constexpr int NT = COUT_PAD / 32;
  constexpr int W_LD = (COUT_PAD * QPR + kW8Threads - 1) / kW8Threads;
  constexpr int NW = kW8Threads / 64;
  static_assert(CIN % VPQ == 0, "a quad must not straddle two neighbour rows");
  // LDS: W chunk x2 | gathered tile per wave | neighbour indices per wave | masks
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  char* sA = smem_raw + 2 * COUT_PAD * kRowBytes + wave * 32 * kRowBytes;
  int* sIdx = reinterpret_cast<int*>(smem_raw + 2 * COUT_PAD * kRowBytes + NW * 32 * kRowBytes) +
              wave * kKSlots * 32;
  int* sMask = reinterpret_cast<int*>(smem_raw + 2 * COUT_PAD * kRowBytes + NW * 32 * kRowBytes) +
               NW * kKSlots * 32;
  const T* in = reinterpret_cast<const T*>(a.in);
  const T* wgt = reinterpret_cast<const T*>(a.w);
  const T* zero = reinterpret_cast<const T*>(a.zero);

  const int r = lane & 31, h = lane >> 5;
  const long long row0 = tile * kW8Rows + wave * 32;
  const int nchunks = (a.kvol * CIN + CK - 1) / CK;

  unsigned wmask = 0;
#pragma unroll
  for (int i = 0; i < kKSlots / 2; ++i) {
    // unconditional loads from clamped addresses, the select afterwards: a guarded load is waited for inside its
    // branch, i.e. 14 serial L2 round trips per tile
    const int ks = 2 * i + h;
    const bool ok = ks < a.kvol && row0 + r < n_out;
    const int ksc = ks < a.kvol ? ks : 0;
    const int krow = a.flip_k ? a.kvol - 1 - ksc : ksc;
    const long long rr = row0 + r < n_out ? row0 + r : n_out - 1;
    const int got = a.nbr[(long long)krow * a.ld + rr];
    const int idx = ok ? got : -1;
    sIdx[ks * 32 + r] = idx;
    const unsigned long long b = __ballot(idx >= 0);
    if ((unsigned)b) wmask |= 1u << (2 * i);
    if ((unsigned)(b >> 32)) wmask |= 1u << (2 * i + 1);
  }
  if (lane == 0) sMask[wave] = (int)wmask;
  __syncthreads();
  unsigned gmask = 0;
#pragma unroll
  for (int i = 0; i < NW; ++i) gmask |= (unsigned)sMask[i];

  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.0f;

  // kernel offsets touched by chunk c: [c*CK/CIN, (c*CK+CK-1)/CIN]
  auto chunk_used = [&](unsigned m, int c) {
    const int g0 = c * CK / CIN;
    constexpr int span = (CK + CIN - 1) / CIN;
    constexpr unsigned bits = (span >= 32) ? 0xffffffffu : ((1u << span) - 1u);
    return ((m >> g0) & bits) != 0u;
  };
  auto next_chunk = [&](int c) {
    ++c;
    while (c < nchunks && !chunk_used(gmask, c)) ++c;
    return c;
  };

  f32x4 ra[8], rw[W_LD];
  VX_STAMP_DECL;

  // Gather: wave-instruction t covers tile rows 4t..4t+3, 16 lanes per row reading the row's 256
  // staged bytes of this chunk as whole 128-byte lines (8 distinct lines per instruction instead
  // of 64 for fragment-shaped loads: the texture-addresser was the bottleneck of that form).
  const int arow = lane >> 4, ap = lane & 15;
  auto load_a = [&](int c) {
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int row = 4 * t + arow;
      const int P = c * CK + VPQ * ap;                 // staged value position of this piece
      const int g = P / CIN, ch = P % CIN;
      const int idx = g < kKSlots ? sIdx[g * 32 + row] : -1;   // slots >= kvol hold -1
      const T* src = idx >= 0 ? in + (long long)idx * CIN + ch : zero;
      ra[t] = *reinterpret_cast<const f32x4*>(src);
    }
  };
  // LDS images are unpadded 256-byte rows with the 16-byte piece index XOR-ed by (row & 15):
  // fragment reads (ds_read_b128, one piece per lane, 16 different rows per lane group) then hit
  // 16 different 16-byte bank groups -> conflict-free without padding.
  auto store_a = [&]() {
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int row = 4 * t + arow;
      *reinterpret_cast<f32x4*>(sA + row * kRowBytes + ((ap ^ (row & 15)) << 4)) = ra[t];
    }
  };
  auto load_w = [&](int c) {
#pragma unroll
    for (int t = 0; t < W_LD; ++t) {
      const int e = t * kW8Threads + (int)threadIdx.x;
      const int oc = e / QPR, q = e % QPR;
      const int P = c * CK + VPQ * q;
      const int g = P / CIN, ch = P % CIN;
      const int goc = col0 + oc;
      const bool ok = goc < a.cout && g < a.kvol && oc < COUT_PAD;
      const T* src = ok ? wgt + ((long long)goc * a.kvol + g) * CIN + ch : zero;
      rw[t] = *reinterpret_cast<const f32x4*>(src);
    }
  };
  auto store_w = [&](int buf) {
    char* sW = smem_raw + buf * COUT_PAD * kRowBytes;
#pragma unroll
    for (int t = 0; t < W_LD; ++t) {
      const int e = t * kW8Threads + (int)threadIdx.x;
      const int oc = e / QPR, q = e % QPR;
      if (oc < COUT_PAD) *reinterpret_cast<f32x4*>(sW + oc * kRowBytes + ((q ^ (oc & 15)) << 4)) = rw[t];
    }
  };
  auto mfma_chunk = [&](int buf) {
    const char* pb = smem_raw + buf * COUT_PAD * kRowBytes + r * kRowBytes;
    const char* pa = sA + r * kRowBytes;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int off = ((h * 8 + j) ^ (r & 15)) << 4;
      const f32x4 av = *reinterpret_cast<const f32x4*>(pa + off);
      f32x4 bv[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) bv[t] = *reinterpret_cast<const f32x4*>(pb + t * 32 * kRowBytes + off);
      Elem<T>::template mma<NT>(av, bv, acc);
    }
  };

  // The loop body is straight-line on purpose: every trip issues the same loads and stores (the last trip
  // re-fetches its own chunk, a wave that skips a chunk stages zero rows for it), so the compiler can place
  // exact counted waits. With the prefetch under `if`s it merged the two paths' wait state and drained the
  // W loads (vmcnt(0)) before issuing the gathers, exposing one L2 round trip per chunk.
  int c = next_chunk(-1);
  int buf = 0;
  if (c < nchunks) {
    load_w(c);
    load_a(c);
    store_a();
    store_w(0);
  }
  __syncthreads();
  while (c < nchunks) {
    // chunk c is staged (sA, sW[buf]); prefetch chunk cn into registers while its MFMAs run
    const int cn = next_chunk(c);
    const int cl = cn < nchunks ? cn : c;
    VX_STAMP_T(t1);
    load_w(cl);
    load_a(cl);
    VX_STAMP_T(t2);
    if (chunk_used(wmask, c)) mfma_chunk(buf);
    VX_STAMP_T(t3);
    store_a();
    store_w(buf ^ 1);
    VX_STAMP_T(t4);
    __syncthreads();
    VX_STAMP_T(t5);
    VX_STAMP_ACC(1, t2 - t1); VX_STAMP_ACC(2, t3 - t2); VX_STAMP_ACC(3, t4 - t3);
    VX_STAMP_ACC(4, t5 - t4); VX_STAMP_ACC(5, 1);
    buf ^= 1;
    c = cn;
  }

  VX_STAMP_FLUSH;
  // ---- epilogue: bias, per-channel affine (folded BatchNorm), residual, ReLU ---------------------
  T* out = reinterpret_cast<T*>(a.out);
  const T* res = reinterpret_cast<const T*>(a.residual);
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
      if (row >= n_out) continue;
      const long long orow = a.perm ? a.perm[row] : row;
      float v = acc[t][i] + b;
      if (a.scale || a.shift) v = v * sc + sh;
      if (res) v += Elem<T>::load(res + orow * a.cout + col);
      if (a.relu) v = fmaxf(v, 0.0f);
      Elem<T>::store(out + orow * a.cout + col, v);
    }
  }
}

template <typename T, int CIN, int COUT_PAD>
__global__ __launch_bounds__(kW8Threads, 2) void spconv_w8(ConvArgs a) {
  const long long n_out = rows_of(a);
  const long long ntiles = (n_out + kW8Rows - 1) / kW8Rows;
  if ((long long)blockIdx.x >= ntiles) return;
  w8_body<T, CIN, COUT_PAD>(a, n_out, xcd_remap(blockIdx.x, (unsigned)ntiles), blockIdx.y * COUT_PAD);
}

// Split-tail launch. The kernel keeps one workgroup per CU (its LDS holds two W chunks), so a grid of
// G = 256*q + rem row tiles costs q + 1 rounds however small rem is: a single FHD frame has 309 tiles at
// C = 128 and pays two rounds for 1.2 rounds of work. Here the first n_full = 256*q tiles run as usual and
// each of the last `rem` tiles is taken by two workgroups computing half of the output columns each
// (2*rem <= 256 workgroups -> one round of roughly half the duration). Only used for cout == COUT_PAD.
template <typename T, int CIN, int COUT_PAD>
__global__ __launch_bounds__(kW8Threads, 2) void spconv_w8_split_tail(ConvArgs a, int n_full) {
  const long long n_out = rows_of(a);
  int n_half = (int)gridDim.x - n_full;          // half-width workgroups (two per tail tile)
  if (a.n_out_dev) {                             // the host's split rule (launch_w8), from the device row count
    const long long tiles = (n_out + kW8Rows - 1) / kW8Rows, rem = tiles % a.cus;
    const bool split = rem > 0 && 2 * rem <= a.cus;
    n_full = (int)(split ? tiles - rem : tiles);
    n_half = split ? (int)(2 * rem) : 0;
  }
  const int b = blockIdx.x;
  if (b < n_full) {
    w8_body<T, CIN, COUT_PAD>(a, n_out, xcd_remap(b, n_full), 0);
  } else {
    const int j = b - n_full;
    if (j >= n_half) return;
    w8_body<T, CIN, COUT_PAD / 2>(a, n_out, n_full + (j >> 1), (j & 1) * (COUT_PAD / 2));
  }
}

// ------------------------------------------------------------------------------------------------
// fp32 through the bf16 matrix pipe (spconv_w8s). gfx950 issues v_mfma_f32_32x32x16_bf16 at 16x the flop rate
// of v_mfma_f32_32x32x2_f32 (2.5 PFLOP/s against 157 TFLOP/s dense). An fp32 value is the EXACT sum of three
// bf16 values: x = h + m + l with h = x's top 16 bits (sign, exponent, 7 fraction bits), m = the top 16 bits
// of x - h (exact in fp32: the remaining 16 significant bits), l = x - h - m (at most 8 significant bits left,
// so it is a bf16 without rounding; bf16 has fp32's exponent range, underflow only within 2^-110 of zero).
// Every bf16 x bf16 product is exact in fp32 (8 x 8 significant bits) and the matrix core accumulates in fp32,
// so the 9 plane products carry the fp32 x fp32 product into the sum exactly (an fp32 FMA chain rounds to 24
// bits per step): NP = 9 is fp32 arithmetic at 9/16 of the fp32 MFMA's cycles. NP = 6 drops the three smallest
// plane products (m*l, l*m, l*l: below 2^-23 of |a*b|). Inputs that are not finite turn into NaN (inf - inf in
// the split), as inf * 0 does in the plain product of a row with an absent neighbour.
//
// Same tile and pipeline as spconv_w8 (256 rows, 8 waves, accumulators in registers, next chunk prefetched into
// registers, one barrier per chunk); a chunk is 32 fp32 values (128 gathered bytes per row). The W chunk is
// split ONCE per workgroup when it is staged -- three bf16 plane images [Cout][64 B] per buffer, 16-byte pieces
// XOR-swizzled by (row >> 2) & 3 -- so the eight waves read ready-made B fragments; the gathered rows are
// staged as fp32 (pieces swizzled by (row >> 1) & 7) and split by the wave that owns them right after the
// fragment read (each gathered value is read as a fragment exactly once: a wave covers all output columns).
// ------------------------------------------------------------------------------------------------
#define VX_BF16_MMA(A, B, C) C = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, C, 0, 0, 0)

constexpr int kSChunk = 32;                    // fp32 values per chunk
constexpr int kSRowBytes = kSChunk * 4;        // gathered bytes per row per chunk
constexpr int kSPlaneRow = kSChunk * 2;        // bytes of one W row in one bf16 plane image

template <int COUT_PAD, int NW>
constexpr size_t w8s_lds_bytes() {
  return (size_t)2 * 3 * COUT_PAD * kSPlaneRow + (size_t)2 * NW * 32 * kSRowBytes + (size_t)NW * kKSlots * 32 * sizeof(int) + 64;
}

template <int COUT_PAD, int NW>
constexpr int w8s_wg_per_cu() {   // resident workgroups per CU (LDS and two waves per SIMD)
  return (int)((160 * 1024) / w8s_lds_bytes<COUT_PAD, NW>()) < 8 / NW ? (int)((160 * 1024) / w8s_lds_bytes<COUT_PAD, NW>()) : 8 / NW;
}

template <int CIN, int COUT_PAD, int NP, int NW>
__device__ __forceinline__ void w8s_body(const ConvArgs& a, const long long n_out, const long long tile, const int col0) {
  static_assert(NP == 6 || NP == 9, "six or nine plane products");
  static_assert(CIN % kSChunk == 0, "a chunk must not straddle two kernel offsets");
  constexpr int NT = COUT_PAD / 32;
  constexpr int NTHR = 64 * NW;
  constexpr int WBUF = 3 * COUT_PAD * kSPlaneRow;     // bytes of one W buffer (three planes)
  constexpr int W_LD = (COUT_PAD * 4 + NTHR - 1) / NTHR;   // 8-value W pieces per thread
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int ABUF = 32 * kSRowBytes;               // bytes of one gathered tile of a wave
  char* sA = smem_raw + 2 * WBUF + wave * 2 * ABUF;   // two buffers per wave
  int* sIdx = reinterpret_cast<int*>(smem_raw + 2 * WBUF + NW * 2 * ABUF) + wave * kKSlots * 32;
  int* sMask = reinterpret_cast<int*>(smem_raw + 2 * WBUF + NW * 2 * ABUF) + NW * kKSlots * 32;
  const float* in = a.in;
  const float* wgt = a.w;
  const float* zero = reinterpret_cast<const float*>(a.zero);

  const int r = lane & 31, h = lane >> 5;
  const long long row0 = tile * (32 * NW) + wave * 32;
  const int nchunks = a.kvol * (CIN / kSChunk);

  unsigned wmask = 0;
#pragma unroll
  for (int i = 0; i < kKSlots / 2; ++i) {
    // unconditional loads from clamped addresses (a guarded load costs a round trip each: the compiler waits for
    // it inside its branch), the select afterwards
    const int ks = 2 * i + h;
    const bool ok = ks < a.kvol && row0 + r < n_out;
    const int ksc = ks < a.kvol ? ks : 0;
    const int krow = a.flip_k ? a.kvol - 1 - ksc : ksc;
    const long long rr = row0 + r < n_out ? row0 + r : n_out - 1;
    const int got = a.nbr[(long long)krow * a.ld + rr];
    const int idx = ok ? got : -1;
    sIdx[ks * 32 + r] = idx;
    const unsigned long long b = __ballot(idx >= 0);
    if ((unsigned)b) wmask |= 1u << (2 * i);
    if ((unsigned)(b >> 32)) wmask |= 1u << (2 * i + 1);
  }
  if (lane == 0) sMask[wave] = (int)wmask;
  __syncthreads();
  unsigned gmask = 0;
#pragma unroll
  for (int i = 0; i < NW; ++i) gmask |= (unsigned)sMask[i];

  // accumulators: S16 = v_mfma_f32_16x16x32_bf16 (2 row tiles x COUT_PAD/16 column tiles of 16 x 16, one MFMA per
  // chunk and plane product), else v_mfma_f32_32x32x16_bf16 (COUT_PAD/32 tiles of 32 x 32, two MFMAs per chunk)
  constexpr bool S16 = VX_W8S_S16 != 0;
  constexpr int CT = COUT_PAD / 16;
  // XOR keys of the LDS images. ds_read_b128 is served in four NON-contiguous 16-lane groups ({0-3,12-15,20-27},
  // {4-11,16-19,28-31} and the same + 32; MI355X_MICROARCH.md, LDS): with lane = n16 + 16 kg a group mixes two k-groups,
  // so the keys of the 32 x 32 form ((row >> 1) & 7, (col >> 2) & 3) leave 2-way conflicts (15 % of the CU cycles).
  // These two make every group's 16 addresses distinct for the 16 x 16 fragments (checked exhaustively off line).
  auto key_a = [](int row) {
    const int m = (row >> 1) & 7;
    if constexpr (!S16) return m;
    return 2 * ((m + 2) & 3) + (1 - (((m + 2) >> 2) & 1));
  };
  auto key_b = [](int col) {
    const int u = (col >> 2) & 3;
    if constexpr (!S16) return u;
    return (((u & 1) ^ (u >> 1)) << 1) | (u >> 1);
  };
  f32x16 acc[S16 ? 1 : NT];
  f32x4 acc16[S16 ? 2 : 1][S16 ? CT : 1];
  if constexpr (S16) {
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) acc16[rt][ct] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  } else {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[t][i] = 0.0f;
  }

  auto chunk_used = [&](unsigned m, int c) { return ((m >> (c / (CIN / kSChunk))) & 1u) != 0u; };
  auto next_chunk = [&](int c) {
    ++c;
    while (c < nchunks && !chunk_used(gmask, c)) ++c;
    return c;
  };

  f32x4 ra[4], rw[2 * W_LD];
  // gather: a wave-instruction covers 8 tile rows, 8 lanes per row reading the row's 128 bytes of this chunk
  const int arow = lane >> 3, ap = lane & 7;
  auto load_a = [&](int c) {
    const int g = c / (CIN / kSChunk), ch = (c % (CIN / kSChunk)) * kSChunk + 4 * ap;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int row = 8 * t + arow;
      const int idx = sIdx[g * 32 + row];
      const float* src = idx >= 0 ? in + (long long)idx * CIN + ch : zero;
      ra[t] = *reinterpret_cast<const f32x4*>(src);
    }
  };
  auto store_a = [&](int buf) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int row = 8 * t + arow;
      *reinterpret_cast<f32x4*>(sA + buf * ABUF + row * kSRowBytes + ((ap ^ key_a(row)) << 4)) = ra[t];
    }
  };
  // W: piece e = thread + i * NTHR holds values [8q, 8q + 8) of row oc of the chunk (oc = e / 4, q = e % 4)
  const int wq = (int)threadIdx.x & 3;
  auto load_w = [&](int c) {
    const int g = c / (CIN / kSChunk), ch = (c % (CIN / kSChunk)) * kSChunk + 8 * wq;
#pragma unroll
    for (int i = 0; i < W_LD; ++i) {
      const int woc = ((int)threadIdx.x + i * NTHR) >> 2;
      const int goc = col0 + woc;
      const bool ok = woc < COUT_PAD && goc < a.cout;
      const float* src = ok ? wgt + ((long long)goc * a.kvol + g) * CIN + ch : zero;
      rw[2 * i] = *reinterpret_cast<const f32x4*>(src);
      rw[2 * i + 1] = *reinterpret_cast<const f32x4*>(src + 4);
    }
  };
  auto store_w = [&](int buf) {
#pragma unroll
    for (int i = 0; i < W_LD; ++i) {
      const int woc = ((int)threadIdx.x + i * NTHR) >> 2;
      if (woc >= COUT_PAD) continue;
      const Planes p = split8(rw[2 * i], rw[2 * i + 1]);
      char* dst = smem_raw + buf * WBUF + woc * kSPlaneRow + ((wq ^ key_b(woc)) << 4);
      *reinterpret_cast<bf16x8*>(dst) = p.h;
      *reinterpret_cast<bf16x8*>(dst + COUT_PAD * kSPlaneRow) = p.m;
      *reinterpret_cast<bf16x8*>(dst + 2 * COUT_PAD * kSPlaneRow) = p.l;
    }
  };
  const int n16 = lane & 15, kg = lane >> 4;
  auto mfma_chunk = [&](int buf) {
    if constexpr (S16) {
      // lane (n16, kg): values [8 kg, 8 kg + 8) of the chunk = pieces 2 kg, 2 kg + 1 of gathered row 16 rt + n16, piece kg
      // of W row 16 ct + n16 in each plane image
      Planes A[2];
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
        const int row = 16 * rt + n16, ka = key_a(row);
        const char* pa = sA + buf * ABUF + row * kSRowBytes;
        A[rt] = split8(*reinterpret_cast<const f32x4*>(pa + (((2 * kg) ^ ka) << 4)),
                       *reinterpret_cast<const f32x4*>(pa + (((2 * kg + 1) ^ ka) << 4)));
      }
      const char* pb = smem_raw + buf * WBUF + n16 * kSPlaneRow + ((kg ^ key_b(n16)) << 4);
#define VX_MMA16(PA, B_) acc16[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[rt].PA, B_, acc16[rt][ct], 0, 0, 0)
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        const bf16x8 Bh = *reinterpret_cast<const bf16x8*>(pb + ct * 16 * kSPlaneRow);
        const bf16x8 Bm = *reinterpret_cast<const bf16x8*>(pb + COUT_PAD * kSPlaneRow + ct * 16 * kSPlaneRow);
        const bf16x8 Bl = *reinterpret_cast<const bf16x8*>(pb + 2 * COUT_PAD * kSPlaneRow + ct * 16 * kSPlaneRow);
        if constexpr (NP == 9) {
#pragma unroll
          for (int rt = 0; rt < 2; ++rt) VX_MMA16(l, Bl);
#pragma unroll
          for (int rt = 0; rt < 2; ++rt) VX_MMA16(m, Bl);
#pragma unroll
          for (int rt = 0; rt < 2; ++rt) VX_MMA16(l, Bm);
        }
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) VX_MMA16(m, Bm);
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) VX_MMA16(h, Bl);
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) VX_MMA16(l, Bh);
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) VX_MMA16(h, Bm);
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) VX_MMA16(m, Bh);
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) VX_MMA16(h, Bh);
      }
#undef VX_MMA16
    } else {
      const char* pa = sA + buf * ABUF + r * kSRowBytes;
      const char* pb = smem_raw + buf * WBUF + r * kSPlaneRow;
      const int ka = key_a(r), kb = key_b(r);
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int q = 4 * s + 2 * h;
        const Planes A = split8(*reinterpret_cast<const f32x4*>(pa + ((q ^ ka) << 4)),
                                *reinterpret_cast<const f32x4*>(pa + (((q + 1) ^ ka) << 4)));
        const int ob = ((2 * s + h) ^ kb) << 4;
        bf16x8 Bh[NT], Bm[NT], Bl[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          Bh[t] = *reinterpret_cast<const bf16x8*>(pb + t * 32 * kSPlaneRow + ob);
          Bm[t] = *reinterpret_cast<const bf16x8*>(pb + COUT_PAD * kSPlaneRow + t * 32 * kSPlaneRow + ob);
          Bl[t] = *reinterpret_cast<const bf16x8*>(pb + 2 * COUT_PAD * kSPlaneRow + t * 32 * kSPlaneRow + ob);
        }
        // smallest products first, h*h last
        if constexpr (NP == 9) {
#pragma unroll
          for (int t = 0; t < NT; ++t) VX_BF16_MMA(A.l, Bl[t], acc[t]);
#pragma unroll
          for (int t = 0; t < NT; ++t) VX_BF16_MMA(A.m, Bl[t], acc[t]);
#pragma unroll
          for (int t = 0; t < NT; ++t) VX_BF16_MMA(A.l, Bm[t], acc[t]);
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) VX_BF16_MMA(A.m, Bm[t], acc[t]);
#pragma unroll
        for (int t = 0; t < NT; ++t) VX_BF16_MMA(A.h, Bl[t], acc[t]);
#pragma unroll
        for (int t = 0; t < NT; ++t) VX_BF16_MMA(A.l, Bh[t], acc[t]);
#pragma unroll
        for (int t = 0; t < NT; ++t) VX_BF16_MMA(A.h, Bm[t], acc[t]);
#pragma unroll
        for (int t = 0; t < NT; ++t) VX_BF16_MMA(A.m, Bh[t], acc[t]);
#pragma unroll
        for (int t = 0; t < NT; ++t) VX_BF16_MMA(A.h, Bh[t], acc[t]);
      }
    }
  };

  // Pipeline: registers hold chunk c+1 (loaded during the previous trip), LDS buffer `buf` holds chunk c. A trip
  // stages the registers into the OTHER buffers, issues the loads of chunk c+2, then runs chunk c's MFMAs: the
  // loads have a whole trip to land, and one barrier per trip still orders everything (the buffers written here
  // were last read by the MFMAs of the previous trip, which every wave finished before the barrier that ended it).
  // The body is straight-line as in w8_body (the last trips re-fetch a valid chunk and stage it unused).
  int c = next_chunk(-1);
  int buf = 0;
  int cn = c;
  if (c < nchunks) {
    load_w(c);
    load_a(c);
    store_a(0);
    store_w(0);
    cn = next_chunk(c);
    const int cl = cn < nchunks ? cn : c;
    load_w(cl);
    load_a(cl);
  }
  __syncthreads();
  while (c < nchunks) {
    const int cnn = cn < nchunks ? next_chunk(cn) : cn;
    const int cl = cnn < nchunks ? cnn : c;
    store_a(buf ^ 1);
    store_w(buf ^ 1);
    load_w(cl);
    load_a(cl);
    if (chunk_used(wmask, c)) mfma_chunk(buf);
    __syncthreads();
    buf ^= 1;
    c = cn;
    cn = cnn;
  }

  // ---- epilogue: bias, per-channel affine (folded BatchNorm), residual, ReLU ---------------------
  float* out = a.out;
  const float* res = a.residual;
  auto emit = [&](long long row, int col, float acc_v, float b, float sc, float sh) {
    if (row >= n_out) return;
    const long long orow = a.perm ? a.perm[row] : row;
    float v = acc_v + b;
    if (a.scale || a.shift) v = v * sc + sh;
    if (res) v += res[orow * a.cout + col];
    if (a.relu) v = fmaxf(v, 0.0f);
    out[orow * a.cout + col] = v;
  };
  if constexpr (S16) {
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      const int col = col0 + ct * 16 + n16;
      if (col >= a.cout) continue;
      const float b = a.bias ? a.bias[col] : 0.0f;
      const float sc = a.scale ? a.scale[col] : 1.0f;
      const float sh = a.shift ? a.shift[col] : 0.0f;
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int i = 0; i < 4; ++i) emit(row0 + 16 * rt + 4 * kg + i, col, acc16[rt][ct][i], b, sc, sh);
    }
  } else {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int col = col0 + t * 32 + r;
      if (col >= a.cout) continue;
      const float b = a.bias ? a.bias[col] : 0.0f;
      const float sc = a.scale ? a.scale[col] : 1.0f;
      const float sh = a.shift ? a.shift[col] : 0.0f;
#pragma unroll
      for (int i = 0; i < 16; ++i) emit(row0 + (i & 3) + 8 * (i >> 2) + 4 * h, col, acc[t][i], b, sc, sh);
    }
  }
}

template <int CIN, int COUT_PAD, int NP, int NW>
__global__ __launch_bounds__(64 * NW, 2) void spconv_w8s(ConvArgs a) {
  const long long n_out = rows_of(a);
  const long long ntiles = (n_out + 32 * NW - 1) / (32 * NW);
  if ((long long)blockIdx.x >= ntiles) return;
  w8s_body<CIN, COUT_PAD, NP, NW>(a, n_out, xcd_remap(blockIdx.x, (unsigned)ntiles), blockIdx.y * COUT_PAD);
}

// split-tail form, as spconv_w8_split_tail; a round is `slots` = cus * (workgroups per CU) tiles
template <int CIN, int COUT_PAD, int NP, int NW>
__global__ __launch_bounds__(64 * NW, 2) void spconv_w8s_split_tail(ConvArgs a, int n_full) {
  const long long n_out = rows_of(a);
  int n_half = (int)gridDim.x - n_full;
  if (a.n_out_dev) {
    const long long slots = (long long)a.cus * w8s_wg_per_cu<COUT_PAD, NW>();
    const long long tiles = (n_out + 32 * NW - 1) / (32 * NW), rem = tiles % slots;
    const bool split = rem > 0 && 2 * rem <= slots;
    n_full = (int)(split ? tiles - rem : tiles);
    n_half = split ? (int)(2 * rem) : 0;
  }
  const int b = blockIdx.x;
  if (b < n_full) {
    w8s_body<CIN, COUT_PAD, NP, NW>(a, n_out, xcd_remap(b, n_full), 0);
  } else {
    const int j = b - n_full;
    if (j >= n_half) return;
    w8s_body<CIN, COUT_PAD / 2, NP, NW>(a, n_out, n_full + (j >> 1), (j & 1) * (COUT_PAD / 2));
  }
}

// Direct-gather variant (small C): the gathered operand goes straight into VGPRs in fragment
// layout (no LDS round trip); W rows are padded to 272 bytes. Faster than the LDS-staged gather
// when a chunk holds few MFMAs (C <= 32), slower at C >= 64 where its 64-lines-per-instruction
// loads saturate the texture addresser (measured with in-kernel stamps).
constexpr int kRowPadD = kRowBytes + 16;

template <typename T, int CIN, int COUT_PAD>
__global__ __launch_bounds__(kW8Threads, 2) void spconv_w8d(ConvArgs a) {
  constexpr int VPQ = Elem<T>::VPQ;
  constexpr int CK = kRowBytes / (int)sizeof(T);      // staged values per chunk (64 / 128)
  constexpr int QPR = kRowBytes / 16;                 // quads per W row (16)
  constexpr int NT = COUT_PAD / 32;
  constexpr int W_LD = (COUT_PAD * QPR + kW8Threads - 1) / kW8Threads;
  constexpr int NW = kW8Threads / 64;
  static_assert(CIN % VPQ == 0, "a quad must not straddle two neighbour rows");
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int* sIdx = reinterpret_cast<int*>(smem_raw + 2 * COUT_PAD * kRowPadD) + wave * kKSlots * 32;
  int* sMask = reinterpret_cast<int*>(smem_raw + 2 * COUT_PAD * kRowPadD) + NW * kKSlots * 32;
  const T* in = reinterpret_cast<const T*>(a.in);
  const T* wgt = reinterpret_cast<const T*>(a.w);
  const T* zero = reinterpret_cast<const T*>(a.zero);

  const int r = lane & 31, h = lane >> 5;
  const long long n_out = rows_of(a);
  const long long ntiles = (n_out + kW8Rows - 1) / kW8Rows;
  if ((long long)blockIdx.x >= ntiles) return;
  const long long tile = xcd_remap(blockIdx.x, (unsigned)ntiles);
  const long long row0 = tile * kW8Rows + wave * 32;
  const int col0 = blockIdx.y * COUT_PAD;
  const int nchunks = (a.kvol * CIN + CK - 1) / CK;

  unsigned wmask = 0;
#pragma unroll
  for (int i = 0; i < kKSlots / 2; ++i) {
    // unconditional loads from clamped addresses, the select afterwards: a guarded load is waited for inside its
    // branch, i.e. 14 serial L2 round trips per tile
    const int ks = 2 * i + h;
    const bool ok = ks < a.kvol && row0 + r < n_out;
    const int ksc = ks < a.kvol ? ks : 0;
    const int krow = a.flip_k ? a.kvol - 1 - ksc : ksc;
    const long long rr = row0 + r < n_out ? row0 + r : n_out - 1;
    const int got = a.nbr[(long long)krow * a.ld + rr];
    const int idx = ok ? got : -1;
    sIdx[ks * 32 + r] = idx;
    const unsigned long long b = __ballot(idx >= 0);
    if ((unsigned)b) wmask |= 1u << (2 * i);
    if ((unsigned)(b >> 32)) wmask |= 1u << (2 * i + 1);
  }
  if (lane == 0) sMask[wave] = (int)wmask;
  __syncthreads();
  unsigned gmask = 0;
#pragma unroll
  for (int i = 0; i < NW; ++i) gmask |= (unsigned)sMask[i];

  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.0f;

  // kernel offsets touched by chunk c: [c*CK/CIN, (c*CK+CK-1)/CIN]
  auto chunk_used = [&](unsigned m, int c) {
    const int g0 = c * CK / CIN;
    constexpr int span = (CK + CIN - 1) / CIN;
    constexpr unsigned bits = (span >= 32) ? 0xffffffffu : ((1u << span) - 1u);
    return ((m >> g0) & bits) != 0u;
  };
  auto next_chunk = [&](int c) {
    ++c;
    while (c < nchunks && !chunk_used(gmask, c)) ++c;
    return c;
  };

  f32x4 r0[8], r1[8], rw[W_LD];

  auto load_a = [&](int c, f32x4(&dst)[8]) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int P = c * CK + h * (CK / 2) + VPQ * q;   // staged value position
      const int g = P / CIN, ch = P % CIN;
      const int idx = g < kKSlots ? sIdx[g * 32 + r] : -1;   // slots >= kvol hold -1
      const T* src = idx >= 0 ? in + (long long)idx * CIN + ch : zero;
      dst[q] = *reinterpret_cast<const f32x4*>(src);
    }
  };
  auto load_w = [&](int c) {
#pragma unroll
    for (int t = 0; t < W_LD; ++t) {
      const int e = t * kW8Threads + (int)threadIdx.x;
      const int oc = e / QPR, q = e % QPR;
      const int P = c * CK + VPQ * q;
      const int g = P / CIN, ch = P % CIN;
      const int goc = col0 + oc;
      const bool ok = goc < a.cout && g < a.kvol && oc < COUT_PAD;
      const T* src = ok ? wgt + ((long long)goc * a.kvol + g) * CIN + ch : zero;
      rw[t] = *reinterpret_cast<const f32x4*>(src);
    }
  };
  auto store_w = [&](int buf) {
    char* sW = smem_raw + buf * COUT_PAD * kRowPadD;
#pragma unroll
    for (int t = 0; t < W_LD; ++t) {
      const int e = t * kW8Threads + (int)threadIdx.x;
      const int oc = e / QPR, q = e % QPR;
      if (oc < COUT_PAD) *reinterpret_cast<f32x4*>(sW + oc * kRowPadD + q * 16) = rw[t];
    }
  };
  auto mfma_chunk = [&](int buf, const f32x4(&ra)[8]) {
    const char* pb = smem_raw + buf * COUT_PAD * kRowPadD + r * kRowPadD + h * (kRowBytes / 2);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      f32x4 bv[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) bv[t] = *reinterpret_cast<const f32x4*>(pb + t * 32 * kRowPadD + 16 * j);
      Elem<T>::template mma<NT>(ra[j], bv, acc);
    }
  };
  // one pipeline step: chunk c is in `cur` + sW[buf]; prefetch chunk cn into `nxt` + sW[buf^1].
  // Straight-line like spconv_w8's loop (unconditional loads/stores; the last step re-fetches its own chunk).
  auto step = [&](int c, int cn, int buf, const f32x4(&cur)[8], f32x4(&nxt)[8]) {
    const int cl = cn < nchunks ? cn : c;
    load_w(cl);
    load_a(cl, nxt);
    if (chunk_used(wmask, c)) mfma_chunk(buf, cur);
    store_w(buf ^ 1);
    __syncthreads();
  };

  int c = next_chunk(-1);
  if (c < nchunks) {
    load_w(c);
    load_a(c, r0);
    store_w(0);
  }
  __syncthreads();
  while (c < nchunks) {
    const int c1 = next_chunk(c);
    step(c, c1, 0, r0, r1);
    if (c1 >= nchunks) break;
    const int c2 = next_chunk(c1);
    step(c1, c2, 1, r1, r0);
    c = c2;
  }

  // ---- epilogue: bias, per-channel affine (folded BatchNorm), residual, ReLU ---------------------
  T* out = reinterpret_cast<T*>(a.out);
  const T* res = reinterpret_cast<const T*>(a.residual);
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
      if (row >= n_out) continue;
      const long long orow = a.perm ? a.perm[row] : row;
      float v = acc[t][i] + b;
      if (a.scale || a.shift) v = v * sc + sh;
      if (res) v += Elem<T>::load(res + orow * a.cout + col);
      if (a.relu) v = fmaxf(v, 0.0f);
      Elem<T>::store(out + orow * a.cout + col, v);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Per-offset compaction kernel (fp32, Cout <= 64): the output-stationary kernels above spend an MFMA row
// on every (row, offset) slot of an active 32-row group, present or not -- 42 % (C=64) to 66 % (C=16) of
// their matrix work at the FHD stages, and far more in the strided convs. Here a 4-wave workgroup owns
// 128 output rows whose accumulators live in LDS; for each kernel offset every wave compacts the tile's
// present (row, input) pairs in registers (two ballots + ds_permute, no LDS list, no extra barrier),
// wave g takes pairs [32g, 32g+32), gathers their input rows straight into MFMA fragment layout,
// multiplies with W_k (LDS, double buffered: one barrier per offset) and adds its 32 x Cout result into
// the LDS accumulator rows of those pairs. Within one offset an output row belongs to exactly one pair,
// hence one wave; offsets are separated by the barrier: plain in-order sums, bitwise reproducible.
// ------------------------------------------------------------------------------------------------
#ifndef VX_W8_SPLIT_TAIL
#define VX_W8_SPLIT_TAIL 1  // split the last partial round of spconv_w8 over output-column halves
#endif
#ifndef VX_CB_WAVES
#define VX_CB_WAVES 4      // waves (= 32-row groups) per workgroup of the compaction kernel
#endif
#ifndef VX_CB_BF16_128
#define VX_CB_BF16_128 0
#endif
#ifndef VX_CB_BF16
#define VX_CB_BF16 1
#endif
#ifndef VX_CB_NS_64
#define VX_CB_NS_64 2      // ... with two waves per pair group (32 of the 64 channels each): 16 waves on a 256-row tile
#endif
#ifndef VX_CB_WAVES_64
#define VX_CB_WAVES_64 16  // Cin = 64, Cout >= 64: W_k is 16 KB per offset, a 256-row tile halves its re-reads
#endif

// NS waves share one 32-pair group, each taking Cout/NS of its output channels (more, lighter waves per tile).
// NP = 6 / 9 (fp32 only): the split-bf16 arithmetic of spconv_w8s -- W_k split once per workgroup into three bf16 plane
// images [Cout][CIN * 2 + 16 B], the gathered half rows split by their lane, v_mfma_f32_32x32x16_bf16 (lane (r, h) feeds
// its channels [h CIN/2 + 8 s, + 8) to MFMA s as k-group h).
template <typename T, int CIN, int COUT_PAD, int NP>
constexpr size_t cb_w_bytes() {   // one W buffer
  return NP ? (size_t)3 * COUT_PAD * (CIN * 2 + 16) : (size_t)COUT_PAD * (CIN * sizeof(T) + 16);
}

template <typename T, int CIN, int COUT_PAD, int WAVES, int NS, int NP = 0>
__global__ __launch_bounds__(64 * WAVES, 512 / (64 * WAVES)) void spconv_cb(ConvArgs a) {
  constexpr int kCbRows = 32 * WAVES / NS, kCbThreads = 64 * WAVES;
  static_assert(NP == 0 || (sizeof(T) == 4 && (NP == 6 || NP == 9) && CIN % 16 == 0), "split form: fp32, Cin % 16 == 0");
  constexpr int P_ROW = CIN * 2 + 16, P_PLANE = COUT_PAD * P_ROW;
  constexpr int W_BUF = (int)cb_w_bytes<T, CIN, COUT_PAD, NP>();
  constexpr int W_P8 = COUT_PAD * CIN / 8;                                 // 8-value pieces of W_k (split form)
  constexpr int W_PL = (W_P8 + kCbThreads - 1) / kCbThreads;
  static_assert(kCbRows % 64 == 0 && (COUT_PAD / 32) % NS == 0, "tile shape");
  constexpr int NE = kCbRows / 64;                   // neighbour entries per lane
  constexpr int NT = COUT_PAD / 32 / NS;             // 32-channel tiles per wave
  constexpr int EB = (int)sizeof(T);
  constexpr int QR = CIN * EB / 16;                  // 16-byte quads per row
  constexpr int QH = QR / 2;                         // quads per lane half (half a row)
  static_assert(QH >= 1, "a lane half needs at least one 16-byte quad");
  constexpr int W_ROW = CIN * EB + 16;               // padded W row (bytes): conflict-free ds_read_b128
  constexpr int W_Q = COUT_PAD * QR;                 // quads of one offset's weights
  constexpr int W_LD = (W_Q + kCbThreads - 1) / kCbThreads;
  constexpr int ACC_LD = COUT_PAD + 4;               // +16 B per row: rows 16 apart share a bank group, not neighbours
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* sAcc = reinterpret_cast<float*>(smem_raw);                       // [(R + 1)][ACC_LD], row R = dump
  char* sWb = smem_raw + (kCbRows + 1) * ACC_LD * 4;                      // 2 x [COUT_PAD][W_ROW]

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int grp = wave / NS, t0 = (wave % NS) * NT;  // this wave's pair group and first channel tile
  const long long n_out = rows_of(a);
  const long long ntiles = (n_out + kCbRows - 1) / kCbRows;
  if ((long long)blockIdx.x >= ntiles) return;
  const long long row0 = (long long)xcd_remap(blockIdx.x, (unsigned)ntiles) * kCbRows;
  const int col0 = blockIdx.y * COUT_PAD;
  const char* in = reinterpret_cast<const char*>(a.in);
  const char* wgt = reinterpret_cast<const char*>(a.w);
  const char* zero = reinterpret_cast<const char*>(a.zero);

  for (int e = threadIdx.x; e < (kCbRows + 1) * ACC_LD; e += kCbThreads) sAcc[e] = 0.0f;

  // neighbour entries of offset ks for tile rows lane, lane + 64, ... (-1 past the end / absent)
  auto load_nbr = [&](int ks, int (&e)[NE]) {
#pragma unroll
    for (int j = 0; j < NE; ++j) e[j] = -1;
    if (ks < a.kvol) {
      const int krow = a.flip_k ? a.kvol - 1 - ks : ks;
      const int* t = a.nbr + (long long)krow * a.ld + row0;
#pragma unroll
      for (int j = 0; j < NE; ++j)
        if (row0 + lane + 64 * j < n_out) e[j] = t[lane + 64 * j];
    }
  };
  // compaction of the tile's entries; this wave keeps pairs [32*grp, 32*grp + 32):
  // lanes 0..31 receive (input row + 1, local output row + 1) of their slot, 0 = empty slot
  auto my_group = [&](const int (&e)[NE], int& idx_p1, int& row_p1) {
    int base = -32 * grp, ii = 0, qq = 0;
#pragma unroll
    for (int j = 0; j < NE; ++j) {
      const unsigned long long m = __ballot(e[j] >= 0);
      const int d = base + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
      const bool p = e[j] >= 0 && d >= 0 && d < 32;
      // ds_permute pushes to lane (addr / 4); lanes that push nothing aim a zero at lanes 32..63
      const int ad = (p ? d : 32 + (lane & 31)) << 2;
      ii |= __builtin_amdgcn_ds_permute(ad, p ? e[j] + 1 : 0);
      qq |= __builtin_amdgcn_ds_permute(ad, p ? lane + 64 * j + 1 : 0);
      base += __popcll(m);
    }
    idx_p1 = lane < 32 ? ii : 0;
    row_p1 = lane < 32 ? qq : 0;
    return base + 32 * grp;        // pairs of this offset in the tile
  };
  auto load_a = [&](int idx_p1, f32x4(&dst)[QH]) {
    const int idx = __shfl(idx_p1, r, 64) - 1;
    const char* src = idx >= 0 ? in + ((long long)idx * CIN + h * (CIN / 2)) * EB : zero;
#pragma unroll
    for (int q = 0; q < QH; ++q) dst[q] = *reinterpret_cast<const f32x4*>(src + 16 * q);
  };
  f32x4 rw[NP ? 2 * W_PL : W_LD];
  auto load_w = [&](int ks) {
    const int kk = ks < a.kvol ? ks : a.kvol - 1;
    if constexpr (NP != 0) {
#pragma unroll
      for (int t = 0; t < W_PL; ++t) {
        const int e = t * kCbThreads + (int)threadIdx.x;
        const int oc = e / (CIN / 8), p8 = e % (CIN / 8);
        const int goc = col0 + oc;
        const bool ok = e < W_P8 && goc < a.cout;
        const char* src = ok ? wgt + ((long long)goc * a.kvol + kk) * CIN * 4 + 32 * p8 : zero;
        rw[2 * t] = *reinterpret_cast<const f32x4*>(src);
        rw[2 * t + 1] = *reinterpret_cast<const f32x4*>(src + 16);
      }
    } else {
#pragma unroll
      for (int t = 0; t < W_LD; ++t) {
        const int e = t * kCbThreads + (int)threadIdx.x;
        const int oc = e / QR, q = e % QR;
        const int goc = col0 + oc;
        const bool ok = e < W_Q && goc < a.cout;
        const char* src = ok ? wgt + ((long long)goc * a.kvol + kk) * CIN * EB + 16 * q : zero;
        rw[t] = *reinterpret_cast<const f32x4*>(src);
      }
    }
  };
  auto store_w = [&](int buf) {
    char* sW = sWb + buf * W_BUF;
    if constexpr (NP != 0) {
#pragma unroll
      for (int t = 0; t < W_PL; ++t) {
        const int e = t * kCbThreads + (int)threadIdx.x;
        if (e >= W_P8) continue;
        const int oc = e / (CIN / 8), p8 = e % (CIN / 8);
        const Planes p = split8(rw[2 * t], rw[2 * t + 1]);
        char* dst = sW + oc * P_ROW + 16 * p8;
        *reinterpret_cast<bf16x8*>(dst) = p.h;
        *reinterpret_cast<bf16x8*>(dst + P_PLANE) = p.m;
        *reinterpret_cast<bf16x8*>(dst + 2 * P_PLANE) = p.l;
      }
    } else {
#pragma unroll
      for (int t = 0; t < W_LD; ++t) {
        const int e = t * kCbThreads + (int)threadIdx.x;
        const int oc = e / QR, q = e % QR;
        if (e < W_Q) *reinterpret_cast<f32x4*>(sW + oc * W_ROW + q * 16) = rw[t];
      }
    }
  };
  // 32 pairs x Cout, computed TRANSPOSED (W fragment as the A operand, gathered rows as B): lane (r, h) then
  // holds Cout values 8q + 4h + {0..3} of pair slot r, i.e. 16 contiguous bytes of that pair's accumulator row:
  // the += into LDS is 4 ds_read_b128 + 4 ds_write_b128 per 32 channels
  auto mfma_group = [&](int buf, const f32x4(&ra)[QH], int row_p1) {
    f32x16 d[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) d[t][i] = 0.0f;
    if constexpr (NP != 0) {
      static_assert(NP == 0 || QH % 2 == 0, "whole 8-value k-groups per lane");
      const char* pw = sWb + buf * W_BUF + (t0 * 32 + r) * P_ROW + h * CIN;   // h * (CIN / 2) bf16 values
#pragma unroll
      for (int s2 = 0; s2 < QH / 2; ++s2) {
        const Planes g = split8(ra[2 * s2], ra[2 * s2 + 1]);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const char* q0 = pw + t * 32 * P_ROW + 16 * s2;
          const bf16x8 wh = *reinterpret_cast<const bf16x8*>(q0);
          const bf16x8 wm = *reinterpret_cast<const bf16x8*>(q0 + P_PLANE);
          const bf16x8 wl = *reinterpret_cast<const bf16x8*>(q0 + 2 * P_PLANE);
#define VX_MMA32(A, B) d[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, d[t], 0, 0, 0)
          if constexpr (NP == 9) {
            VX_MMA32(wl, g.l);
            VX_MMA32(wm, g.l);
            VX_MMA32(wl, g.m);
          }
          VX_MMA32(wm, g.m);
          VX_MMA32(wh, g.l);
          VX_MMA32(wl, g.h);
          VX_MMA32(wh, g.m);
          VX_MMA32(wm, g.h);
          VX_MMA32(wh, g.h);
#undef VX_MMA32
        }
      }
    }
    const char* pb = sWb + buf * W_BUF + (t0 * 32 + r) * W_ROW + h * (CIN * EB / 2);
#pragma unroll
    for (int j = 0; j < (NP ? 0 : QH); ++j) {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const f32x4 bv = *reinterpret_cast<const f32x4*>(pb + t * 32 * W_ROW + 16 * j);
        if constexpr (EB == 4) {
          d[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv.x, ra[j].x, d[t], 0, 0, 0);
          d[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv.y, ra[j].y, d[t], 0, 0, 0);
          d[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv.z, ra[j].z, d[t], 0, 0, 0);
          d[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv.w, ra[j].w, d[t], 0, 0, 0);
        } else {
          d[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, bv), __builtin_bit_cast(bf16x8, ra[j]),
                                                         d[t], 0, 0, 0);
        }
      }
    }
    const int rp = __shfl(row_p1, r, 64);
    float* arow = sAcc + (rp > 0 ? rp - 1 : kCbRows) * ACC_LD + t0 * 32 + 4 * h;   // empty slots add zeros to the dump row
    // no other wave touches this row during this offset: plain read-add-write (ds_add_f32 measured 4-5x
    // slower for the whole kernel: LDS float atomics do not run at the ds_read / ds_write rate)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4* dst = reinterpret_cast<f32x4*>(arow + t * 32 + 8 * q);
        f32x4 v = *dst;
        v.x += d[t][4 * q + 0];
        v.y += d[t][4 * q + 1];
        v.z += d[t][4 * q + 2];
        v.w += d[t][4 * q + 3];
        *dst = v;
      }
  };

  int en[NE], fn[NE];                 // entries of offsets k+1 (en) and k+2 (fn), prefetched
  int idxA, rowA, idxB, rowB;
  f32x4 raA[QH], raB[QH];
  load_nbr(0, en);
  load_nbr(1, fn);
  int cntA = my_group(en, idxA, rowA);
  load_w(0);
  load_a(idxA, raA);
  store_w(0);
  __syncthreads();
  // one trip = two offsets (ping-pong register sets); straight-line bodies, clamped loads past the end
  auto step = [&](int k, int buf, int cnt_cur, const f32x4(&cur)[QH], int row_cur, int& cnt_nxt, f32x4(&nxt)[QH],
                  int& idx_nxt, int& row_nxt) {
#pragma unroll
    for (int j = 0; j < NE; ++j) en[j] = fn[j];
    load_nbr(k + 2, fn);
    cnt_nxt = my_group(en, idx_nxt, row_nxt);
    load_w(k + 1);
    load_a(idx_nxt, nxt);
    if (32 * grp < cnt_cur) mfma_group(buf, cur, row_cur);
    store_w(buf ^ 1);
    __syncthreads();
  };
  int cntB = 0;
  for (int k = 0; k < a.kvol; k += 2) {
    step(k, 0, cntA, raA, rowA, cntB, raB, idxB, rowB);
    if (k + 1 >= a.kvol) break;
    step(k + 1, 1, cntB, raB, rowB, cntA, raA, idxA, rowA);
  }
  __syncthreads();

  // ---- epilogue: bias, per-channel affine (folded BatchNorm), residual, ReLU; coalesced rows ----
  constexpr int CQ = COUT_PAD / 4;
  for (int e = threadIdx.x; e < kCbRows * CQ; e += kCbThreads) {
    const int rr = e / CQ, c = col0 + 4 * (e % CQ);
    const long long row = row0 + rr;
    if (row >= n_out) continue;
    const f32x4 v4 = *reinterpret_cast<const f32x4*>(sAcc + rr * ACC_LD + 4 * (e % CQ));
    const float vv[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int col = c + j;
      if (col >= a.cout) continue;
      float v = vv[j] + (a.bias ? a.bias[col] : 0.0f);
      if (a.scale || a.shift) v = v * (a.scale ? a.scale[col] : 1.0f) + (a.shift ? a.shift[col] : 0.0f);
      if (a.residual) v += Elem<T>::load(reinterpret_cast<const T*>(a.residual) + row * a.cout + col);
      if (a.relu) v = fmaxf(v, 0.0f);
      Elem<T>::store(reinterpret_cast<T*>(a.out) + row * a.cout + col, v);
    }
  }
}

// 16-pair tasks (fp32, Cin = 64): the same per-offset compaction with three changes, each from a timing-only ablation of
// the 32-pair NS = 2 form on the FHD 64->64 layer (2.85 ms: MFMAs 1.1 ms, LDS accumulate 0.9, compaction 0.7, floor 0.5,
// gathers 0.15 -- nearly additive, i.e. the phases ran one after the other behind the per-offset barrier):
//  * a wave takes 16 pairs and ALL Cout channels on v_mfma_f32_16x16x4_f32 (same flops per cycle as 32x32x2) instead of
//    two waves sharing a 32-pair group by output-channel halves: every input row is gathered once, four lanes share a row
//    (64 bytes each), and the pair count is rounded up to 16 instead of 32;
//  * ONE producer wave (the last: it has a task of its own only when more than 16 (WAVES - 1) pairs are present) ranks the
//    tile's entries two offsets ahead and publishes the compacted (input row, output row) list in LDS; the other waves
//    read their 16 slots (one ds_read_b64 per lane) instead of every wave ranking all 256 entries with ballots and
//    ds_permutes, and only the producer reads the neighbour table;
//  * the += of offset k's products into the LDS accumulators is deferred into offset k + 1's MFMA stream (read issued
//    before a 16-MFMA group, add + write after it), so its LDS latency hides under matrix work. Offsets are still
//    separated by one barrier: within an offset an output row belongs to one pair, hence one wave, and the deferred
//    accumulate of offset k finishes before the barrier that precedes any accumulate of offset k + 1.
// Operands: B (gathered rows): lane (n = lane & 15, kq = lane >> 4) holds channels [16 kq, 16 kq + 16) of pair n; the
// reduction index is permuted accordingly (MFMA step (q, j) sums channels 16 kq + 4 q + j over kq); A: the W fragment
// from LDS with the same permutation; D: lane (n, kq) holds output channels 16 t + 4 kq + {0..3} of pair n, 16 contiguous
// bytes of that pair's accumulator row.
#ifndef VX_CB16
#define VX_CB16 1
#endif
// NP = 6 / 9: the split-bf16 arithmetic of spconv_w8s (three exact bf16 planes per fp32 operand, NP plane products on
// v_mfma_f32_16x16x32_bf16, fp32 accumulation). W_k is split once per workgroup when it is staged (three plane images
// [Cout][CIN * 2 + 16 B]: the 16 rows of a fragment read land on 16 different bank groups), the gathered rows by the
// wave that fetched them; lane (n, kq) feeds its channels [16 kq + 8 s, 16 kq + 8 s + 8) to MFMA s as k-group kq.
template <int CIN, int COUT_PAD, int NP>
constexpr size_t cb16_w_bytes() {   // one W buffer
  return NP ? (size_t)3 * COUT_PAD * (CIN * 2 + 16) : (size_t)COUT_PAD * (CIN * 4 + 16);
}

template <int CIN, int COUT_PAD, int WAVES, int NP = 0>
__global__ __launch_bounds__(64 * WAVES) void spconv_cb16(ConvArgs a, long long row_base, long long ntiles, int part) {
  constexpr int kCbRows = 16 * WAVES, kCbThreads = 64 * WAVES;
  static_assert(NP == 0 || ((NP == 6 || NP == 9) && CIN == 64), "split form: Cin = 64");
  constexpr int P_ROW = CIN * 2 + 16;                // padded W row of one bf16 plane image (bytes)
  constexpr int P_PLANE = COUT_PAD * P_ROW;
  constexpr int W_BUF = (int)cb16_w_bytes<CIN, COUT_PAD, NP>();
  static_assert(kCbRows % 64 == 0 && COUT_PAD % 16 == 0 && CIN % 16 == 0, "tile shape");
  constexpr int NE = kCbRows / 64;                   // neighbour entries per producer lane
  constexpr int NT = COUT_PAD / 16;                  // 16-channel output tiles (a wave computes all of them)
  constexpr int QR = CIN * 4 / 16;                   // 16-byte quads per row
  constexpr int QK = QR / 4;                         // quads per lane (a quarter of a row)
  static_assert(NT == QK, "one deferred accumulator tile per MFMA group");
  constexpr int W_ROW = CIN * 4 + 16;                // padded W row (bytes)
  constexpr int W_Q = COUT_PAD * QR;
  constexpr int W_LD = (W_Q + kCbThreads - 1) / kCbThreads;
  constexpr int ACC_LD = COUT_PAD + 4;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* sAcc = reinterpret_cast<float*>(smem_raw);                       // [(R + 1)][ACC_LD], row R = dump
  char* sWb = smem_raw + (kCbRows + 1) * ACC_LD * 4;                      // 2 x [COUT_PAD][W_ROW] (or 2 x 3 planes)
  int2* sList = reinterpret_cast<int2*>(sWb + 2 * W_BUF);                 // 2 x [R] (input row + 1, output row + 1)
  int* sCnt = reinterpret_cast<int*>(sList + 2 * kCbRows);                // 2 pair counts

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n16 = lane & 15, kq = lane >> 4;
  const bool producer = wave == WAVES - 1;
  const long long n_out = rows_of(a);
  if (a.n_out_dev) {      // launch_cb16's split between full-height tiles (part 0) and a half-height tail (part 1)
    const int rf = part ? 2 * kCbRows : kCbRows;
    const long long tiles = (n_out + rf - 1) / rf, rem = tiles % a.cus;
    const bool split = !a.no_split && a.cout <= COUT_PAD && rem > 0 && 2 * rem <= a.cus && tiles > rem;
    if (part == 0) {
      row_base = 0;
      ntiles = split ? tiles - rem : tiles;
    } else {
      if (!split) return;
      row_base = (tiles - rem) * rf;
      ntiles = (n_out - row_base + kCbRows - 1) / kCbRows;
    }
  }
  if ((long long)blockIdx.x >= ntiles) return;
  const long long row0 = row_base + (long long)xcd_remap(blockIdx.x, (unsigned)ntiles) * kCbRows;
  const int col0 = blockIdx.y * COUT_PAD;
  const char* in = reinterpret_cast<const char*>(a.in);
  const char* wgt = reinterpret_cast<const char*>(a.w);
  const char* zero = reinterpret_cast<const char*>(a.zero);

  for (int e = threadIdx.x; e < (kCbRows + 1) * ACC_LD; e += kCbThreads) sAcc[e] = 0.0f;

  // ---- producer: neighbour entries of offset ks for tile rows lane, lane + 64, ... and their compaction ----
  auto load_nbr = [&](int ks, int (&e)[NE]) {
#pragma unroll
    for (int j = 0; j < NE; ++j) e[j] = -1;
    if (ks < a.kvol) {
      const int krow = a.flip_k ? a.kvol - 1 - ks : ks;
      const int* t = a.nbr + (long long)krow * a.ld + row0;
#pragma unroll
      for (int j = 0; j < NE; ++j)
        if (row0 + lane + 64 * j < n_out) e[j] = t[lane + 64 * j];
    }
  };
  auto publish = [&](const int (&e)[NE], int lb) {
    int base = 0;
#pragma unroll
    for (int j = 0; j < NE; ++j) {
      const unsigned long long m = __ballot(e[j] >= 0);
      const int d = base + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
      if (e[j] >= 0) sList[lb * kCbRows + d] = make_int2(e[j] + 1, lane + 64 * j + 1);
      base += __popcll(m);
    }
    if (lane == 0) sCnt[lb] = base;
  };
  // ---- consumers: this wave's 16 slots of a published list (slots past the count read as empty) ----
  auto fetch_pairs = [&](int lb, int& idx_p1, int& row_p1) {
    const int cnt = sCnt[lb];
    const int slot = 16 * wave + n16;
    const int2 v = sList[lb * kCbRows + slot];
    idx_p1 = slot < cnt ? v.x : 0;
    row_p1 = slot < cnt ? v.y : 0;
    return cnt;
  };
  auto load_a = [&](int idx_p1, f32x4(&dst)[QK]) {
    const char* src = idx_p1 > 0 ? in + ((long long)(idx_p1 - 1) * CIN + kq * (CIN / 4)) * 4 : zero;
#pragma unroll
    for (int q = 0; q < QK; ++q) dst[q] = *reinterpret_cast<const f32x4*>(src + 16 * q);
  };
  constexpr int W_P8 = COUT_PAD * CIN / 8;           // split form: 8-value pieces of W_k, one per thread
  static_assert(NP == 0 || W_P8 <= kCbThreads, "one W piece per thread");
  f32x4 rw[NP ? 2 : W_LD];
  auto load_w = [&](int ks) {
    const int kk = ks < a.kvol ? ks : a.kvol - 1;
    if constexpr (NP != 0) {
      const int e = (int)threadIdx.x;
      const int oc = e / (CIN / 8), p8 = e % (CIN / 8);
      const int goc = col0 + oc;
      const bool ok = e < W_P8 && goc < a.cout;
      const char* src = ok ? wgt + ((long long)goc * a.kvol + kk) * CIN * 4 + 32 * p8 : zero;
      rw[0] = *reinterpret_cast<const f32x4*>(src);
      rw[1] = *reinterpret_cast<const f32x4*>(src + 16);
    } else {
#pragma unroll
      for (int t = 0; t < W_LD; ++t) {
        const int e = t * kCbThreads + (int)threadIdx.x;
        const int oc = e / QR, q = e % QR;
        const int goc = col0 + oc;
        const bool ok = e < W_Q && goc < a.cout;
        const char* src = ok ? wgt + ((long long)goc * a.kvol + kk) * CIN * 4 + 16 * q : zero;
        rw[t] = *reinterpret_cast<const f32x4*>(src);
      }
    }
  };
  auto store_w = [&](int buf) {
    char* sW = sWb + buf * W_BUF;
    if constexpr (NP != 0) {
      const int e = (int)threadIdx.x;
      if (e < W_P8) {
        const int oc = e / (CIN / 8), p8 = e % (CIN / 8);
        const Planes p = split8(rw[0], rw[1]);
        char* dst = sW + oc * P_ROW + 16 * p8;
        *reinterpret_cast<bf16x8*>(dst) = p.h;
        *reinterpret_cast<bf16x8*>(dst + P_PLANE) = p.m;
        *reinterpret_cast<bf16x8*>(dst + 2 * P_PLANE) = p.l;
      }
    } else {
#pragma unroll
      for (int t = 0; t < W_LD; ++t) {
        const int e = t * kCbThreads + (int)threadIdx.x;
        const int oc = e / QR, q = e % QR;
        if (e < W_Q) *reinterpret_cast<f32x4*>(sW + oc * W_ROW + q * 16) = rw[t];
      }
    }
  };
  auto acc_row = [&](int row_p1) {   // empty slots add zeros to the dump row
    return sAcc + (row_p1 > 0 ? row_p1 - 1 : kCbRows) * ACC_LD + 4 * kq;
  };
  // pending products of the previous offset (this wave's 16 pairs x Cout), added to LDS during the next MFMA stream
  f32x4 dP[NT];
  int rowP = 0;
  bool haveP = false;
  auto flush_pending = [&]() {
    float* arow = acc_row(rowP);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      f32x4* dst = reinterpret_cast<f32x4*>(arow + t * 16);
      *dst = *dst + dP[t];
    }
  };
  auto mfma_group = [&](int buf, const f32x4(&ra)[QK], int row_p1) {
    f32x4 d[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) d[t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    float* arowP = acc_row(rowP);
    if constexpr (NP != 0) {
      static_assert(QK == 4, "two 8-value k-groups per lane");
      const Planes g[2] = {split8(ra[0], ra[1]), split8(ra[2], ra[3])};
      const char* pw = sWb + buf * W_BUF + n16 * P_ROW + kq * (CIN / 2);   // kq * (CIN / 4) bf16 values
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        f32x4 old = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        if (haveP) old = *reinterpret_cast<const f32x4*>(arowP + t * 16);
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const char* q0 = pw + t * 16 * P_ROW + 16 * s2;
          const bf16x8 wh = *reinterpret_cast<const bf16x8*>(q0);
          const bf16x8 wm = *reinterpret_cast<const bf16x8*>(q0 + P_PLANE);
          const bf16x8 wl = *reinterpret_cast<const bf16x8*>(q0 + 2 * P_PLANE);
#define VX_MMA16(A, B) d[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, B, d[t], 0, 0, 0)
          if constexpr (NP == 9) {
            VX_MMA16(wl, g[s2].l);
            VX_MMA16(wm, g[s2].l);
            VX_MMA16(wl, g[s2].m);
          }
          VX_MMA16(wm, g[s2].m);
          VX_MMA16(wh, g[s2].l);
          VX_MMA16(wl, g[s2].h);
          VX_MMA16(wh, g[s2].m);
          VX_MMA16(wm, g[s2].h);
          VX_MMA16(wh, g[s2].h);
#undef VX_MMA16
        }
        if (haveP) *reinterpret_cast<f32x4*>(arowP + t * 16) = old + dP[t];
      }
    } else {
    const char* pb = sWb + buf * COUT_PAD * W_ROW + n16 * W_ROW + kq * CIN;   // kq * (CIN / 4) floats
#pragma unroll
    for (int q = 0; q < QK; ++q) {
      f32x4 old = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
      if (haveP) old = *reinterpret_cast<const f32x4*>(arowP + q * 16);
      // W fragments two tiles at a time: the kernel sits at the 128-VGPR cap of a 1024-thread workgroup
#pragma unroll
      for (int t0 = 0; t0 < NT; t0 += 2) {
        f32x4 bv[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) bv[t] = *reinterpret_cast<const f32x4*>(pb + (t0 + t) * 16 * W_ROW + 16 * q);
#pragma unroll
        for (int t = 0; t < 2; ++t) d[t0 + t] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv[t].x, ra[q].x, d[t0 + t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 2; ++t) d[t0 + t] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv[t].y, ra[q].y, d[t0 + t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 2; ++t) d[t0 + t] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv[t].z, ra[q].z, d[t0 + t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 2; ++t) d[t0 + t] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv[t].w, ra[q].w, d[t0 + t], 0, 0, 0);
      }
      if (haveP) *reinterpret_cast<f32x4*>(arowP + q * 16) = old + dP[q];
    }
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) dP[t] = d[t];
    rowP = row_p1;
    haveP = true;
  };

  // ---- prologue: lists of offsets 0 and 1, entries of offset 2 in flight; first gather and W ----
  int pn[NE];
  if (producer) {
    int e0[NE], e1[NE];
    load_nbr(0, e0);
    load_nbr(1, e1);
    load_nbr(2, pn);
    publish(e0, 0);
    publish(e1, 1);
  }
  __syncthreads();
  int idxA, rowA, idxB, rowB;
  f32x4 raA[QK], raB[QK];
  int cntA = fetch_pairs(0, idxA, rowA);
  load_w(0);
  load_a(idxA, raA);
  store_w(0);
  __syncthreads();
  // one trip = two offsets (ping-pong register sets)
  auto step = [&](int k, int buf, int cnt_cur, const f32x4(&cur)[QK], int row_cur, int& cnt_nxt, f32x4(&nxt)[QK],
                  int& idx_nxt, int& row_nxt) {
    if (producer) {                       // list of offset k + 2 into the buffer offset k's list has left
      publish(pn, k & 1);
      load_nbr(k + 3, pn);
    }
    cnt_nxt = fetch_pairs((k + 1) & 1, idx_nxt, row_nxt);
    load_w(k + 1);
    load_a(idx_nxt, nxt);
    if (16 * wave < cnt_cur) {
      mfma_group(buf, cur, row_cur);
    } else if (haveP) {
      flush_pending();
      haveP = false;
    }
    store_w(buf ^ 1);
    __syncthreads();
  };
  int cntB = 0;
  for (int k = 0; k < a.kvol; k += 2) {
    step(k, 0, cntA, raA, rowA, cntB, raB, idxB, rowB);
    if (k + 1 >= a.kvol) break;
    step(k + 1, 1, cntB, raB, rowB, cntA, raA, idxA, rowA);
  }
  if (haveP) flush_pending();
  __syncthreads();

  // ---- epilogue: bias, per-channel affine (folded BatchNorm), residual, ReLU; coalesced rows ----
  constexpr int CQ = COUT_PAD / 4;
  for (int e = threadIdx.x; e < kCbRows * CQ; e += kCbThreads) {
    const int rr = e / CQ, c = col0 + 4 * (e % CQ);
    const long long row = row0 + rr;
    if (row >= n_out) continue;
    const f32x4 v4 = *reinterpret_cast<const f32x4*>(sAcc + rr * ACC_LD + 4 * (e % CQ));
    const float vv[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int col = c + j;
      if (col >= a.cout) continue;
      float v = vv[j] + (a.bias ? a.bias[col] : 0.0f);
      if (a.scale || a.shift) v = v * (a.scale ? a.scale[col] : 1.0f) + (a.shift ? a.shift[col] : 0.0f);
      if (a.residual) v += a.residual[row * a.cout + col];
      if (a.relu) v = fmaxf(v, 0.0f);
      a.out[row * a.cout + col] = v;
    }
  }
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel, device): keeps repeated launches -- and launches
// recorded under stream capture -- free of the attribute call.
int32_t set_lds_once(const void* kern, size_t lds) {
  static std::mutex mu;
  static std::set<std::pair<const void*, int>> done;
  int dev = 0;
  VX_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lock(mu);
  if (done.count({kern, dev})) return VX_OK;
  VX_HIP(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  done.insert({kern, dev});
  return VX_OK;
}

int cu_count();
std::atomic<int>& f32_products_mode();

template <int CIN, int COUT_PAD, int WAVES, int NP>
int32_t launch_cb16_rows(const ConvArgs& a, long long row_base, long long n_tiles, int part, hipStream_t stream) {
  constexpr int R = 16 * WAVES;
  constexpr size_t lds = (size_t)(R + 1) * (COUT_PAD + 4) * 4 + 2 * cb16_w_bytes<CIN, COUT_PAD, NP>() +
                         (size_t)2 * R * sizeof(int2) + 16;
  static_assert(lds <= 160 * 1024, "LDS budget");
  static_assert(((R + 1) * (COUT_PAD + 4) * 4) % 16 == 0 && (2 * cb16_w_bytes<CIN, COUT_PAD, NP>()) % 16 == 0,
                "LDS alignment");
  auto kern = spconv_cb16<CIN, COUT_PAD, WAVES, NP>;
  if (int32_t rc = set_lds_once((const void*)kern, lds)) return rc;
  dim3 grid((unsigned)n_tiles, (unsigned)vx_cdiv(a.cout, COUT_PAD));
  kern<<<grid, 64 * WAVES, lds, stream>>>(a, row_base, n_tiles, part);
  VX_LAUNCH_CHECK();
  return VX_OK;
}

// The kernel keeps one 256-row workgroup per CU, so G = cus * q + rem tiles cost q + 1 rounds however small rem is (one FHD
// frame: 824 tiles at the 64-channel stage = 3.2 rounds, paid as 4). When the last partial round would fill at most half of
// the CUs, its rows run as a second launch of the same kernel on 128-row tiles (8 waves; two workgroups per CU in the fp32
// MFMA form, one in the split form whose W planes take more LDS): about half a round instead of a whole one. An output
// row's sum runs over the offsets in the same order whatever tile it sits in: bitwise identical to the single launch
// (VX_CONV_NO_SPLIT_TAIL selects that one).
template <int CIN, int COUT_PAD, int WAVES, int NP = 0>
int32_t launch_cb16(const ConvArgs& a, hipStream_t stream) {
  if constexpr (NP == 0) {
    if (a.f32_products == 6) return launch_cb16<CIN, COUT_PAD, WAVES, 6>(a, stream);
    if (a.f32_products == 9) return launch_cb16<CIN, COUT_PAD, WAVES, 9>(a, stream);
  }
  constexpr int R = 16 * WAVES;
  const long long tiles = vx_cdiv(a.n_out, R);
  const int cus = a.cus;
  if (a.n_out_dev) {
    // row count on the device: a.n_out is the capacity. Part 0 over every tile the capacity allows, part 1 (the
    // half-height tail, at most `cus` tiles) whenever the split rule could apply; the kernels take the same
    // decisions as below from the device count and surplus workgroups exit.
    if (int32_t rc = launch_cb16_rows<CIN, COUT_PAD, WAVES, NP>(a, 0, tiles, 0, stream)) return rc;
    if (!a.no_split && a.cout <= COUT_PAD)
      return launch_cb16_rows<CIN, COUT_PAD, WAVES / 2, NP>(a, 0, cus, 1, stream);
    return VX_OK;
  }
  const long long rem = tiles % cus;
  if (!a.no_split && a.cout <= COUT_PAD && rem > 0 && 2 * rem <= cus && tiles > rem) {
    if (int32_t rc = launch_cb16_rows<CIN, COUT_PAD, WAVES, NP>(a, 0, tiles - rem, 0, stream)) return rc;
    const long long base = (tiles - rem) * R;
    return launch_cb16_rows<CIN, COUT_PAD, WAVES / 2, NP>(a, base, vx_cdiv(a.n_out - base, R / 2), 1, stream);
  }
  return launch_cb16_rows<CIN, COUT_PAD, WAVES, NP>(a, 0, tiles, 0, stream);
}

template <typename T, int CIN, int COUT_PAD, int WAVES, int NS = 1, int NP = 0>
int32_t launch_cb(const ConvArgs& a, hipStream_t stream) {
  // measured on the FHD layers (tools/conv_probe.py 8): SubM 32->32 1.04 -> 0.96 ms; the strided 16->32 / 32->64 layers lose
  // 5-8 % (few pairs per tile and offset: splitting W_k for every offset costs more than the shorter MFMAs give back) and
  // 16->16 ties, so only the 32 x 32 form takes the split arithmetic here
  if constexpr (NP == 0 && sizeof(T) == 4 && CIN == 32 && COUT_PAD == 32 && VX_CB_SPLIT) {
    if (a.f32_products == 6) return launch_cb<T, CIN, COUT_PAD, WAVES, NS, 6>(a, stream);
    if (a.f32_products == 9) return launch_cb<T, CIN, COUT_PAD, WAVES, NS, 9>(a, stream);
  }
  constexpr int R = 32 * WAVES / NS;
  constexpr size_t lds = (size_t)(R + 1) * (COUT_PAD + 4) * 4 + 2 * cb_w_bytes<T, CIN, COUT_PAD, NP>();
  static_assert(lds <= 160 * 1024, "LDS budget");
  auto kern = spconv_cb<T, CIN, COUT_PAD, WAVES, NS, NP>;
  if (int32_t rc = set_lds_once((const void*)kern, lds)) return rc;
  dim3 grid((unsigned)vx_cdiv(a.n_out, R), (unsigned)vx_cdiv(a.cout, COUT_PAD));
  kern<<<grid, 64 * WAVES, lds, stream>>>(a);
  VX_LAUNCH_CHECK();
  return VX_OK;
}

// 1 = not eligible
template <typename T>
int32_t dispatch_cb_t(const ConvArgs& a, hipStream_t s) {
  const bool wide = a.cout > 32;       // Cout > 64 is tiled over blockIdx.y in slices of 64
  switch (a.cin) {
    case 16:
      if constexpr (sizeof(T) == 4)
        return wide ? launch_cb<T, 16, 64, VX_CB_WAVES>(a, s) : launch_cb<T, 16, 32, VX_CB_WAVES>(a, s);
      else
        return 1;                      // bf16 rows of 16 channels: half a row is less than one 16-byte quad
    case 32: return wide ? launch_cb<T, 32, 64, VX_CB_WAVES>(a, s) : launch_cb<T, 32, 32, VX_CB_WAVES>(a, s);
    case 64:
      if constexpr (sizeof(T) == 4)
        return wide ? (VX_CB16 ? launch_cb16<64, 64, 16>(a, s) : launch_cb<T, 64, 64, VX_CB_WAVES_64, VX_CB_NS_64>(a, s))
                    : launch_cb<T, 64, 32, VX_CB_WAVES>(a, s);
      else
        return wide ? launch_cb<T, 64, 64, VX_CB_WAVES>(a, s) : launch_cb<T, 64, 32, VX_CB_WAVES>(a, s);
    case 128:
      if constexpr (sizeof(T) == 2 && VX_CB_BF16_128)
        return wide ? launch_cb<T, 128, 64, 8>(a, s) : launch_cb<T, 128, 32, VX_CB_WAVES>(a, s);
      else
        return 1;
    default: return 1;
  }
}

int32_t dispatch_cb(const ConvArgs& a, hipStream_t s) {
  if (a.perm || (reinterpret_cast<uintptr_t>(a.in) & 15) || (reinterpret_cast<uintptr_t>(a.w) & 15) ||
      a.w_si != 1 || a.w_sk != a.cin || a.w_so != (long long)a.kvol * a.cin || a.cout > 128)
    return 1;
  if (a.dtype == VX_F32) return dispatch_cb_t<float>(a, s);
  // bf16, measured on the FHD layers (tools/conv_probe.py 8 bf16): compaction wins at 32->64 (-12 %) and 64->64 (-16 %),
  // ties at 64->128, loses at 32->32 (+3 %) and 128->128 (+38 %: both Cout slices re-gather every row)
  if (a.dtype == VX_BF16 && VX_CB_BF16 && (a.cin == 32 || a.cin == 64) && a.cout > 32 && a.cout <= 64)
    return dispatch_cb_t<__bf16>(a, s);
  return 1;
}

// Process-wide choice of the fp32 matrix arithmetic of the row-stationary kernel (forward and dgrad):
// VX_F32_PRODUCTS=0|6|9 at load time, vx_spconv_set_f32_products() afterwards.
std::atomic<int>& f32_products_mode() {
  static std::atomic<int> mode{[] {
    const char* e = getenv("VX_F32_PRODUCTS");
    const int v = e ? atoi(e) : VX_F32_PRODUCTS_DEFAULT;
    return v == 6 || v == 9 ? v : 0;
  }()};
  return mode;
}

// compute units of the current device (MI355X: 8 XCDs x 32 = 256; fewer in a partitioned mode), cached per device
int cu_count() {
  static std::atomic<int> cache[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  int n = cache[dev].load(std::memory_order_relaxed);
  if (n == 0) {
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    cache[dev].store(n, std::memory_order_relaxed);
  }
  return n;
}

// split-bf16 form of the fp32 row-stationary kernel; grid and split-tail rule of launch_w8
#ifndef VX_W8S_WAVES
#define VX_W8S_WAVES 8
#endif
template <int CIN, int COUT_PAD, int NP, int NW = VX_W8S_WAVES>
int32_t launch_w8s(const ConvArgs& a, hipStream_t stream) {
  constexpr size_t lds = w8s_lds_bytes<COUT_PAD, NW>();
  static_assert(lds <= 160 * 1024, "LDS budget");
  constexpr int R = 32 * NW;
  auto kern = spconv_w8s<CIN, COUT_PAD, NP, NW>;
  if (int32_t rc = set_lds_once((const void*)kern, lds)) return rc;
  dim3 grid((unsigned)vx_cdiv(a.n_out, R), (unsigned)vx_cdiv(a.cout, COUT_PAD));
  if constexpr (COUT_PAD >= 64) {
    const long long tiles = grid.x;
    const long long slots = (long long)a.cus * w8s_wg_per_cu<COUT_PAD, NW>();
    const long long rem = tiles % slots;
    const bool may_split = VX_W8_SPLIT_TAIL && !a.no_split && a.cout == COUT_PAD;
    if (may_split && (a.n_out_dev || (rem > 0 && 2 * rem <= slots))) {
      auto kt = spconv_w8s_split_tail<CIN, COUT_PAD, NP, NW>;
      if (int32_t rc = set_lds_once((const void*)kt, lds)) return rc;
      const long long g = a.n_out_dev ? tiles + slots : tiles + rem;
      kt<<<dim3((unsigned)g), 64 * NW, lds, stream>>>(a, (int)(tiles - rem));
      VX_LAUNCH_CHECK();
      return VX_OK;
    }
  }
  kern<<<grid, 64 * NW, lds, stream>>>(a);
  VX_LAUNCH_CHECK();
  return VX_OK;
}

template <typename T, int CIN, int COUT_PAD>
int32_t launch_w8(const ConvArgs& a, hipStream_t stream) {
  // LDS-staged gather where a chunk carries enough MFMA work to hide its LDS round trip
  constexpr bool kLdsGather = sizeof(T) == 4 ? (CIN >= 64) : (CIN >= 128);
  if constexpr (sizeof(T) == 4 && CIN >= 64) {
    if (a.f32_products == 6) return launch_w8s<CIN, COUT_PAD, 6>(a, stream);
    if (a.f32_products == 9) return launch_w8s<CIN, COUT_PAD, 9>(a, stream);
  }
  constexpr size_t lds_l = (size_t)2 * COUT_PAD * kRowBytes + (size_t)(kW8Threads / 64) * 32 * kRowBytes +
                           (size_t)(kW8Threads / 64) * kKSlots * 32 * sizeof(int) + 64;
  constexpr size_t lds_d = (size_t)2 * COUT_PAD * kRowPadD +
                           (size_t)(kW8Threads / 64) * kKSlots * 32 * sizeof(int) + 64;
  constexpr size_t lds = kLdsGather ? lds_l : lds_d;
  static_assert(lds <= 160 * 1024, "LDS budget");
  auto kern = kLdsGather ? spconv_w8<T, CIN, COUT_PAD> : spconv_w8d<T, CIN, COUT_PAD>;
  if (int32_t rc = set_lds_once((const void*)kern, lds)) return rc;
  dim3 grid((unsigned)vx_cdiv(a.n_out, kW8Rows), (unsigned)vx_cdiv(a.cout, COUT_PAD));
  if constexpr (kLdsGather && COUT_PAD >= 64) {
    const long long tiles = grid.x;
    const int cus = a.cus;
    const long long rem = tiles % cus;
    const bool may_split = VX_W8_SPLIT_TAIL && !a.no_split && a.cout == COUT_PAD;
    // device row count: the kernel applies the split rule itself; room for the largest tail (cus / 2 tiles, two
    // half-width workgroups each) beside the capacity's tiles
    if (may_split && (a.n_out_dev || (rem > 0 && 2 * rem <= cus))) {
      auto kt = spconv_w8_split_tail<T, CIN, COUT_PAD>;
      if (int32_t rc = set_lds_once((const void*)kt, lds)) return rc;
      const long long g = a.n_out_dev ? tiles + cus : tiles + rem;
      kt<<<dim3((unsigned)g), kW8Threads, lds, stream>>>(a, (int)(tiles - rem));
      VX_LAUNCH_CHECK();
      return VX_OK;
    }
  }
  kern<<<grid, kW8Threads, lds, stream>>>(a);
  VX_LAUNCH_CHECK();
  return VX_OK;
}

template <typename T, int CIN>
int32_t dispatch_w8_cout(const ConvArgs& a, hipStream_t s) {
  if (a.cout <= 32) return launch_w8<T, CIN, 32>(a, s);
  if (a.cout <= 64) return launch_w8<T, CIN, 64>(a, s);
  return launch_w8<T, CIN, 128>(a, s);
}

template <typename T>
int32_t dispatch_w8(const ConvArgs& a, hipStream_t s) {
  switch (a.cin) {
    case 8: return dispatch_w8_cout<T, 8>(a, s);
    case 16: return dispatch_w8_cout<T, 16>(a, s);
    case 32: return dispatch_w8_cout<T, 32>(a, s);
    case 64: return dispatch_w8_cout<T, 64>(a, s);
    case 128: return dispatch_w8_cout<T, 128>(a, s);
    default: return 1;  // not eligible
  }
}


// ------------------------------------------------------------------------------------------------
// Streaming-W kernel (fp32 only): one wave = one workgroup = 32 output rows, NO barriers.
// W is read from a copy packed in MFMA-fragment order (vx_spconv_pack_weight): every B-fragment
// load is one coalesced 1 KiB wave-instruction served by L2, prefetched one j-step ahead; the
// gathered operand is prefetched one active chunk ahead as in spconv_w8. Because waves never wait
// for each other, a wave skips exactly the chunks ITS 32 rows do not use -- this is what makes the
// mask-sorted row order pay (useful MFMA rows 0.58 -> 0.76 at C=64). Only sensible where the MFMA
// is slow relative to L2 bandwidth: fp32 (a bf16 wave would need 128 B/clk/CU of W from L2).
// ------------------------------------------------------------------------------------------------
template <int CIN, int COUT_PAD>
__global__ __launch_bounds__(64) void spconv_f32_ws(ConvArgs a, const f32x4* __restrict__ wp) {
  constexpr int CK = 64;               // staged values per chunk
  constexpr int NT = COUT_PAD / 32;
  __shared__ int sIdx[kKSlots * 32];
  const int lane = threadIdx.x;
  const int r = lane & 31, h = lane >> 5;
  const long long n_out = rows_of(a);
  const long long ntiles = (n_out + 31) / 32;
  if ((long long)blockIdx.x >= ntiles) return;
  const long long row0 = (long long)xcd_remap(blockIdx.x, (unsigned)ntiles) * 32;
  const int nchunks = (a.kvol * CIN + CK - 1) / CK;
  const float* in = a.in;

  unsigned wmask = 0;
#pragma unroll
  for (int i = 0; i < kKSlots / 2; ++i) {
    // unconditional loads from clamped addresses, the select afterwards: a guarded load is waited for inside its
    // branch, i.e. 14 serial L2 round trips per tile
    const int ks = 2 * i + h;
    const bool ok = ks < a.kvol && row0 + r < n_out;
    const int ksc = ks < a.kvol ? ks : 0;
    const int krow = a.flip_k ? a.kvol - 1 - ksc : ksc;
    const long long rr = row0 + r < n_out ? row0 + r : n_out - 1;
    const int got = a.nbr[(long long)krow * a.ld + rr];
    const int idx = ok ? got : -1;
    sIdx[ks * 32 + r] = idx;
    const unsigned long long b = __ballot(idx >= 0);
    if ((unsigned)b) wmask |= 1u << (2 * i);
    if ((unsigned)(b >> 32)) wmask |= 1u << (2 * i + 1);
  }
  __syncthreads();  // single wave: orders the LDS writes above before the reads below

  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.0f;

  auto chunk_used = [&](int c) {
    const int g0 = c * CK / CIN;
    constexpr int span = (CK + CIN - 1) / CIN;
    constexpr unsigned bits = (span >= 32) ? 0xffffffffu : ((1u << span) - 1u);
    return ((wmask >> g0) & bits) != 0u;
  };
  auto next_chunk = [&](int c) {
    ++c;
    while (c < nchunks && !chunk_used(c)) ++c;
    return c;
  };
  auto load_a = [&](int c, f32x4(&dst)[8]) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int P = c * CK + h * (CK / 2) + 4 * q;
      const int g = P / CIN, ch = P % CIN;
      const int idx = g < kKSlots ? sIdx[g * 32 + r] : -1;
      const float* src = idx >= 0 ? in + (long long)idx * CIN + ch : reinterpret_cast<const float*>(a.zero);
      dst[q] = *reinterpret_cast<const f32x4*>(src);
    }
  };
  // packed layout: [chunk][t][j][lane] quads
  auto load_b = [&](int c, int j, f32x4(&dst)[NT]) {
    const f32x4* p = wp + (((long long)c * NT) * 8 + j) * 64 + lane;
#pragma unroll
    for (int t = 0; t < NT; ++t) dst[t] = p[t * 8 * 64];
  };

  f32x4 a0[8], a1[8], b0[NT], b1[NT];
  // process chunk c (A in `cur`, B(j=0) already in b0); prefetch chunk cn
  auto run_chunk = [&](int c, int cn, const f32x4(&cur)[8], f32x4(&nxt)[8]) {
    if (cn < nchunks) load_a(cn, nxt);
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
      load_b(c, j + 1, b1);
      Elem<float>::template mma<NT>(cur[j], b0, acc);
      if (j + 2 < 8) load_b(c, j + 2, b0);
      else if (cn < nchunks) load_b(cn, 0, b0);
      Elem<float>::template mma<NT>(cur[j + 1], b1, acc);
    }
  };

  int c = next_chunk(-1);
  if (c < nchunks) {
    load_a(c, a0);
    load_b(c, 0, b0);
  }
  while (c < nchunks) {
    const int c1 = next_chunk(c);
    run_chunk(c, c1, a0, a1);
    if (c1 >= nchunks) break;
    const int c2 = next_chunk(c1);
    run_chunk(c1, c2, a1, a0);
    c = c2;
  }

  const int col0 = 0;
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
      if (row >= n_out) continue;
      const long long orow = a.perm ? a.perm[row] : row;
      float v = acc[t][i] + b;
      if (a.scale || a.shift) v = v * sc + sh;
      if (a.residual) v += a.residual[orow * a.cout + col];
      if (a.relu) v = fmaxf(v, 0.0f);
      a.out[orow * a.cout + col] = v;
    }
  }
}

// W[cout][kvol][cin] (KRSC) -> packed quads [chunk][t][j][lane], zero padded
__global__ __launch_bounds__(256) void pack_weight_ws(const float* __restrict__ w, int cout, int kvol,
                                                      int cin, int nt, int nchunks,
                                                      f32x4* __restrict__ wp) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long total = (long long)nchunks * nt * 8 * 64;
  if (e >= total) return;
  const int lane = (int)(e & 63);
  const int j = (int)((e >> 6) & 7);
  const int t = (int)((e >> 9) % nt);
  const int c = (int)((e >> 9) / nt);
  const int col = t * 32 + (lane & 31);
  const long long s0 = (long long)c * 64 + (lane >> 5) * 32 + 4 * j;  // staged position (k*cin + ci)
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (col < cout) {
    const float* src = w + (long long)col * kvol * cin;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (s0 + i < (long long)kvol * cin) v[i] = src[s0 + i];
  }
  wp[e] = v;
}

template <int CIN, int COUT_PAD>
int32_t launch_ws(const ConvArgs& a, const void* wp, hipStream_t stream) {
  spconv_f32_ws<CIN, COUT_PAD><<<(unsigned)vx_cdiv(a.n_out, 32), 64, 0, stream>>>(a, (const f32x4*)wp);
  VX_LAUNCH_CHECK();
  return VX_OK;
}

template <int CIN>
int32_t dispatch_ws_cout(const ConvArgs& a, const void* wp, hipStream_t s) {
  if (a.cout <= 32) return launch_ws<CIN, 32>(a, wp, s);
  if (a.cout <= 64) return launch_ws<CIN, 64>(a, wp, s);
  return launch_ws<CIN, 128>(a, wp, s);
}

int32_t dispatch_ws(const ConvArgs& a, const void* wp, hipStream_t s) {
  switch (a.cin) {
    case 8: return dispatch_ws_cout<8>(a, wp, s);
    case 16: return dispatch_ws_cout<16>(a, wp, s);
    case 32: return dispatch_ws_cout<32>(a, wp, s);
    case 64: return dispatch_ws_cout<64>(a, wp, s);
    case 128: return dispatch_ws_cout<128>(a, wp, s);
    default: return 1;
  }
}

template <int CIN_PAD, int COUT_PAD>
int32_t launch(const ConvArgs& a, hipStream_t stream) {
  constexpr size_t lds = (size_t)(COUT_PAD + kTileRows) * (CIN_PAD + 4) * sizeof(float);
  static_assert(lds <= 160 * 1024, "LDS budget");
  auto kern = spconv_f32_generic<CIN_PAD, COUT_PAD>;
  if (lds > 64 * 1024)
    if (int32_t rc = set_lds_once((const void*)kern, lds)) return rc;
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
  const bool fast = a.w_si == 1 && a.w_sk == a.cin && a.w_so == (long long)a.kvol * a.cin &&
                    a.kvol <= kKSlots && (reinterpret_cast<uintptr_t>(a.in) & 15) == 0 &&
                    (reinterpret_cast<uintptr_t>(a.w) & 15) == 0;
  if (fast) {
    const int32_t rc = a.dtype == VX_BF16 ? dispatch_w8<__bf16>(a, s) : dispatch_w8<float>(a, s);
    if (rc <= 0) return rc;
  }
  if (a.dtype != VX_F32) {
    vx_set_error("spconv: bf16 needs cin in {8,16,32,64,128}, contiguous KRSC weights and kvol <= 32");
    return VX_ERR_ARG;
  }
  if (a.cin <= 8) return dispatch_cout<8>(a, s);
  if (a.cin <= 16) return dispatch_cout<16>(a, s);
  if (a.cin <= 32) return dispatch_cout<32>(a, s);
  if (a.cin <= 64) return dispatch_cout<64>(a, s);
  return dispatch_cout<128>(a, s);
}
// Device address of vx_zero_row on the current device (passed to the kernels as an argument: taking the
// symbol's address inside the kernel costs a GOT load per use in the main loop).
int32_t zero_row_address(const void** out) {
  static std::atomic<const void*> cache[64];     // per device; a racing first call stores the same value
  int dev = 0;
  VX_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64) {
    vx_set_error("zero_row_address: device index %d out of range", dev);
    return VX_ERR_ARG;
  }
  const void* p = cache[dev].load(std::memory_order_acquire);
  if (!p) {
    void* q = nullptr;
    VX_HIP(hipGetSymbolAddress(&q, HIP_SYMBOL(vx_zero_row)));
    cache[dev].store(q, std::memory_order_release);
    p = q;
  }
  *out = p;
  return VX_OK;
}
}  // namespace

extern "C" int32_t vx_spconv_set_f32_products(int32_t products) {
  VX_REQUIRE(products == 0 || products == 6 || products == 9,
             "spconv_set_f32_products: 0 (fp32 MFMA), 6 or 9 (split-bf16 plane products)");
  f32_products_mode().store(products, std::memory_order_relaxed);
  return VX_OK;
}

extern "C" int32_t vx_spconv_get_f32_products(void) { return f32_products_mode().load(std::memory_order_relaxed); }

// arithmetic of one call: its VX_CONV_F32_* flag, else the process default
int vx_resolve_f32_products(int32_t flags) {
  switch (flags & VX_CONV_ARITH_MASK) {
    case VX_CONV_F32_MFMA: return 0;
    case VX_CONV_F32_PRODUCTS6: return 6;
    case VX_CONV_F32_PRODUCTS9: return 9;
    default: return f32_products_mode().load(std::memory_order_relaxed);
  }
}

static int32_t spconv_fwd_impl(const void* d_in, int64_t n_in, int32_t cin, const void* d_weight,
                               int32_t kvol, int32_t cout, const int32_t* d_nbr, int64_t nbr_ld,
                               int64_t n_out, const int64_t* d_n_out, const float* d_bias,
                               const float* d_scale, const float* d_shift, const void* d_residual,
                               int32_t flags, const int32_t* d_perm, const void* d_wpack, void* d_out,
                               int32_t dtype, void* stream) {
  VX_REQUIRE(dtype == VX_F32 || dtype == VX_BF16, "spconv_fwd: dtype must be VX_F32 or VX_BF16");
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
  a.no_split = (flags & VX_CONV_NO_SPLIT_TAIL) ? 1 : 0;
  a.out = (float*)d_out;
  a.dtype = dtype;
  a.perm = d_perm;
  a.n_out_dev = (const long long*)d_n_out;
  a.cus = cu_count();
  a.f32_products = vx_resolve_f32_products(flags);
  if (int32_t rc = zero_row_address(&a.zero)) return rc;
  if (!(flags & VX_CONV_NO_COMPACT)) {   // fp32, Cin in {16,32,64}, Cout <= 128: per-offset compaction kernel
    const int32_t rc = dispatch_cb(a, (hipStream_t)stream);
    if (rc <= 0) return rc;
  }
  if (d_wpack && dtype == VX_F32 && cout <= 128 && kvol <= kKSlots &&
      (reinterpret_cast<uintptr_t>(d_in) & 15) == 0) {
    const int32_t rc = dispatch_ws(a, d_wpack, (hipStream_t)stream);
    if (rc <= 0) return rc;
  }
  return dispatch(a, (hipStream_t)stream);
}

extern "C" int32_t vx_spconv_fwd(const void* d_in, int64_t n_in, int32_t cin, const void* d_weight,
                                 int32_t kvol, int32_t cout, const int32_t* d_nbr, int64_t nbr_ld,
                                 int64_t n_out, const float* d_bias, const float* d_scale,
                                 const float* d_shift, const void* d_residual, int32_t flags,
                                 const int32_t* d_perm, const void* d_wpack, void* d_out,
                                 int32_t dtype, void* stream) {
  return spconv_fwd_impl(d_in, n_in, cin, d_weight, kvol, cout, d_nbr, nbr_ld, n_out, nullptr, d_bias, d_scale,
                         d_shift, d_residual, flags, d_perm, d_wpack, d_out, dtype, stream);
}

extern "C" int32_t vx_spconv_fwd_dn(const void* d_in, int64_t cap_in, int32_t cin, const void* d_weight,
                                    int32_t kvol, int32_t cout, const int32_t* d_nbr, int64_t nbr_ld,
                                    const int64_t* d_n_out, int64_t cap_out, const float* d_bias,
                                    const float* d_scale, const float* d_shift, const void* d_residual,
                                    int32_t flags, void* d_out, int32_t dtype, void* stream) {
  VX_REQUIRE(d_n_out && cap_out >= 1, "spconv_fwd_dn: need the device row count and a capacity >= 1");
  return spconv_fwd_impl(d_in, cap_in, cin, d_weight, kvol, cout, d_nbr, nbr_ld, cap_out, d_n_out, d_bias, d_scale,
                         d_shift, d_residual, flags, nullptr, nullptr, d_out, dtype, stream);
}

extern "C" size_t vx_spconv_pack_weight_bytes(int32_t cin, int32_t cout, int32_t kvol) {
  const size_t nchunks = ((size_t)kvol * cin + 63) / 64, nt = cout <= 32 ? 1 : (cout <= 64 ? 2 : 4);
  return nchunks * nt * 8 * 64 * 16;
}

extern "C" int32_t vx_spconv_pack_weight(const void* d_weight, int32_t cin, int32_t cout,
                                         int32_t kvol, void* d_wpack, void* stream) {
  VX_REQUIRE(d_weight && d_wpack && cin >= 1 && cout >= 1 && cout <= 128 && kvol >= 1,
             "spconv_pack_weight: need cout <= 128 and non-null pointers");
  const int nchunks = (kvol * cin + 63) / 64, nt = cout <= 32 ? 1 : (cout <= 64 ? 2 : 4);
  const long long total = (long long)nchunks * nt * 8 * 64;
  pack_weight_ws<<<(unsigned)vx_cdiv(total, 256), 256, 0, (hipStream_t)stream>>>(
      (const float*)d_weight, cout, kvol, cin, nt, nchunks, (f32x4*)d_wpack);
  VX_LAUNCH_CHECK();
  return VX_OK;
}

extern "C" int32_t vx_spconv_dgrad(const void* d_dout, int64_t n_out, int32_t cout,
                                   const void* d_weight, int32_t kvol, int32_t cin,
                                   const int32_t* d_nbr_t, int64_t nbr_t_ld, int64_t n_in,
                                   int32_t subm, int32_t flags, void* d_din, int32_t dtype, void* stream) {
  VX_REQUIRE(dtype == VX_F32 || dtype == VX_BF16, "spconv_dgrad: dtype must be VX_F32 or VX_BF16");
  VX_REQUIRE(cout >= 1 && cout <= 128 && cin >= 1 && kvol >= 1 && kvol <= 125,
             "spconv_dgrad: need 1 <= cout <= 128, cin >= 1, 1 <= kvol <= 125");
  VX_REQUIRE(n_in >= 0 && n_out >= 0, "spconv_dgrad: negative size");
  if (n_in == 0) return VX_OK;
  VX_REQUIRE(d_weight && d_nbr_t && d_din && nbr_t_ld >= n_in, "spconv_dgrad: null pointer/stride");
  ConvArgs a;
  a.in = (const float*)d_dout;            // gathered operand: dout rows, cout channels
  a.w = (const float*)d_weight;           // transposed copy Wt[ci][k][co]: (oc = ci, ic = co)
  a.w_so = (long long)kvol * cout;
  a.w_sk = cout;
  a.w_si = 1;
  a.nbr = d_nbr_t;
  a.ld = nbr_t_ld;
  a.n_out = n_in;
  a.cin = cout;
  a.cout = cin;
  a.kvol = kvol;
  a.flip_k = subm ? 1 : 0;
  a.bias = a.scale = a.shift = a.residual = nullptr;
  a.relu = 0;
  a.no_split = 0;
  a.out = (float*)d_din;
  a.dtype = dtype;
  a.perm = nullptr;
  a.n_out_dev = nullptr;
  a.cus = cu_count();
  a.f32_products = vx_resolve_f32_products(flags);
  if (int32_t rc = zero_row_address(&a.zero)) return rc;
  {
    const int32_t rc = dispatch_cb(a, (hipStream_t)stream);
    if (rc <= 0) return rc;
  }
  return dispatch(a, (hipStream_t)stream);
}
