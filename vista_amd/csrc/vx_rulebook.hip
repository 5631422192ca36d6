// Rulebook (indice pair) generation on gfx950: lock-free open-addressing hash in HBM for
// coordinate -> row lookups, an occupancy bitmap + popcount scan for the sorted, de-duplicated
// output set of strided convolutions, wavefront-ballot compaction for the pair-list view.
//
// Semantics (spconv call sites det3d/models/backbones/scn.py:333-361,
// multiview.py:33-86): conv (k, s, p, dilation 1), O = (I + 2p - k)/s + 1; input site i feeds
// output o = (i + p - k')/s for every kernel offset k' with (i + p - k') % s == 0, 0 <= o < O.
// Canonical order: outputs sorted by linear index ((b*D+z)*H+y)*W+x; offsets row-major.
// The table is output-major: nbr[k][o] = input row or -1 (all phases are HBM/latency bound
// integer work: coalesced over o, 16-byte hash probes, no GEMM shape anywhere).
#include "vx_common.h"

namespace {
constexpr int kThreads = 256;

struct ConvGeom {
  int in_shape[3];
  int out_shape[3];
  int k[3], s[3], p[3];
  int kvol;
  int batch;
};

// Row count of a launch: `n` from the host, or -- for the "_dn" entry points, whose counts stay on the device --
// min(*d_n, n) with n the capacity the grid was sized for.
__device__ __forceinline__ long long rows_of(long long n, const long long* d_n) {
  if (d_n) {
    const long long d = *d_n;
    n = d < n ? (d > 0 ? d : 0) : n;
  }
  return n;
}

// (batch, z, y, x) inside the input grid? Rows that are not would index the occupancy / rank tables out of bounds:
// they are skipped and reported through the error word (plain idempotent store of 1).
__device__ __forceinline__ bool site_ok(const int4& c, const ConvGeom& g) {
  return c.x >= 0 && c.x < g.batch && c.y >= 0 && c.y < g.in_shape[0] && c.z >= 0 && c.z < g.in_shape[1] &&
         c.w >= 0 && c.w < g.in_shape[2];
}

__device__ __forceinline__ long long lin_idx(int b, int z, int y, int x, const int* sh) {
  return (((long long)b * sh[0] + z) * sh[1] + y) * sh[2] + x;
}

__global__ __launch_bounds__(kThreads) void hash_clear(VxSlot* t, long long cap) {
  long long i = (long long)blockIdx.x * kThreads + threadIdx.x;
  if (i < cap) *reinterpret_cast<int4*>(&t[i]) = make_int4(-1, -1, -1, 0);
}

__global__ __launch_bounds__(kThreads) void hash_build(const int* __restrict__ coords, long long n,
                                                       const long long* __restrict__ d_n,
                                                       ConvGeom g, VxSlot* t, long long mask) {
  long long i = (long long)blockIdx.x * kThreads + threadIdx.x;
  if (i >= rows_of(n, d_n)) return;
  const int4 c = *reinterpret_cast<const int4*>(coords + 4 * i);
  if (!site_ok(c, g)) return;   // never found by a lookup (lookups stay inside the grid)
  long long s = vx_hash_insert(t, mask, lin_idx(c.x, c.y, c.z, c.w, g.in_shape));
  t[s].a = (int)i;  // duplicate coordinates (invalid input) keep one of their rows
}

// submanifold: one thread per (k, o), o fastest -> coalesced coordinate reads and table writes
// Submanifold offsets come in mirrored pairs: nbr[k][o] = j  <=>  nbr[K-1-k][j] = o. Only the first half of the
// offsets is looked up (blockIdx.y = 0 .. K/2); a hit also fills the mirrored entry (each entry has exactly one
// writer), the centre offset is the identity, and the caller pre-fills rows K/2+1 .. K-1 with -1.
__global__ __launch_bounds__(kThreads) void subm_nbr(const int* __restrict__ coords, long long n,
                                                     const long long* __restrict__ d_n,
                                                     ConvGeom g, const VxSlot* __restrict__ t,
                                                     long long mask, int* __restrict__ nbr,
                                                     long long ld) {
  const long long o = (long long)blockIdx.x * kThreads + threadIdx.x;
  const int k = blockIdx.y;
  if (o >= rows_of(n, d_n)) return;
  if (2 * k + 1 == g.kvol) {
    nbr[(long long)k * ld + o] = (int)o;
    return;
  }
  const int kx = k % g.k[2], ky = (k / g.k[2]) % g.k[1], kz = k / (g.k[2] * g.k[1]);
  const int4 c = *reinterpret_cast<const int4*>(coords + 4 * o);
  const int z = c.y + kz - g.k[0] / 2, y = c.z + ky - g.k[1] / 2, x = c.w + kx - g.k[2] / 2;
  int r = -1;
  if (z >= 0 && z < g.in_shape[0] && y >= 0 && y < g.in_shape[1] && x >= 0 && x < g.in_shape[2])
    r = vx_hash_find(t, mask, lin_idx(c.x, z, y, x, g.in_shape));
  nbr[(long long)k * ld + o] = r;
  if (r >= 0) nbr[(long long)(g.kvol - 1 - k) * ld + r] = (int)o;
}

// strided conv phase 1: every input marks the outputs it feeds in the occupancy bitmap
// Output-site marking: one BYTE per output cell, plain idempotent stores (every writer stores 1, so
// races are benign and the result is deterministic). Device-scope atomicOr on a bitmap measured
// 7.5 G marks/s on MI355X (1.2 ms per FHD step); bitmap_pack then folds 32 bytes into one word.
__global__ __launch_bounds__(kThreads) void conv_mark(const int* __restrict__ coords, long long n,
                                                      const long long* __restrict__ d_n,
                                                      ConvGeom g, unsigned char* __restrict__ bytemap,
                                                      long long* __restrict__ err) {
  long long i = (long long)blockIdx.x * kThreads + threadIdx.x;
  if (i >= rows_of(n, d_n)) return;
  const int4 c = *reinterpret_cast<const int4*>(coords + 4 * i);
  if (!site_ok(c, g)) {
    *err = 1;
    return;
  }
  for (int kz = 0; kz < g.k[0]; ++kz) {
    const int tz = c.y + g.p[0] - kz;
    if (tz < 0 || tz % g.s[0]) continue;
    const int oz = tz / g.s[0];
    if (oz >= g.out_shape[0]) continue;
    for (int ky = 0; ky < g.k[1]; ++ky) {
      const int ty = c.z + g.p[1] - ky;
      if (ty < 0 || ty % g.s[1]) continue;
      const int oy = ty / g.s[1];
      if (oy >= g.out_shape[1]) continue;
      for (int kx = 0; kx < g.k[2]; ++kx) {
        const int tx = c.w + g.p[2] - kx;
        if (tx < 0 || tx % g.s[2]) continue;
        const int ox = tx / g.s[2];
        if (ox >= g.out_shape[2]) continue;
        const long long key = lin_idx(c.x, oz, oy, ox, g.out_shape);
        bytemap[key] = 1;
      }
    }
  }
}

__global__ __launch_bounds__(kThreads) void bitmap_pack(const unsigned char* __restrict__ bytemap,
                                                        long long nwords, unsigned* __restrict__ bitmap,
                                                        int* __restrict__ cnt) {
  long long i = (long long)blockIdx.x * kThreads + threadIdx.x;
  if (i >= nwords) return;
  const uint4* src = reinterpret_cast<const uint4*>(bytemap + 32 * i);
  const uint4 lo = src[0], hi = src[1];
  const unsigned q[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
  unsigned bits = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const unsigned x = q[j];   // four bytes, each 0 or 1
    bits |= ((x & 1u) | ((x >> 7) & 2u) | ((x >> 14) & 4u) | ((x >> 21) & 8u)) << (4 * j);
  }
  bitmap[i] = bits;
  cnt[i] = __popc(bits);
}

// phase 2: rank of a set bit = its row; rows come out sorted by linear index
__global__ __launch_bounds__(kThreads) void conv_emit_coords(const unsigned* __restrict__ bitmap,
                                                             const int* __restrict__ base,
                                                             long long nwords, ConvGeom g,
                                                             int* __restrict__ out_coords, long long cap_out) {
  long long w = (long long)blockIdx.x * kThreads + threadIdx.x;
  if (w >= nwords) return;
  unsigned bits = bitmap[w];
  int row = base[w];
  while (bits && row < cap_out) {   // rows past the output capacity are dropped (the caller compares count and capacity)
    const int b = __ffs(bits) - 1;
    bits &= bits - 1;
    long long key = (w << 5) + b;
    const int x = (int)(key % g.out_shape[2]); key /= g.out_shape[2];
    const int y = (int)(key % g.out_shape[1]); key /= g.out_shape[1];
    const int z = (int)(key % g.out_shape[0]); key /= g.out_shape[0];
    *reinterpret_cast<int4*>(out_coords + 4ll * row) = make_int4((int)key, z, y, x);
    ++row;
  }
}

// rank table entry: 32 cells of the (sorted) active set; row(key) = base + popc(bits below key)
__device__ __forceinline__ int rank_lookup(const uint2* __restrict__ rank, long long key) {
  const uint2 e = rank[key >> 5];
  const unsigned bit = (unsigned)key & 31u;
  if (!((e.x >> bit) & 1u)) return -1;
  return (int)e.y + __popc(e.x & ((1u << bit) - 1u));
}

__global__ __launch_bounds__(kThreads) void rank_interleave(const unsigned* __restrict__ bitmap,
                                                            const int* __restrict__ base,
                                                            long long nwords, uint2* __restrict__ rank) {
  long long w = (long long)blockIdx.x * kThreads + threadIdx.x;
  if (w < nwords) rank[w] = make_uint2(bitmap[w], (unsigned)base[w]);
}

// strided conv phase 2, from the INPUT side: every input enumerates the (<= 8 of 27 for k3/s2)
// outputs it feeds, finds their rows in the rank table and writes both tables. The output table
// was pre-filled with -1; the transpose table gets every (k, i) entry written here (coalesced).
__global__ __launch_bounds__(kThreads) void conv_pairs_from_inputs(
    const int* __restrict__ coords, long long n, const long long* __restrict__ d_n, ConvGeom g,
    const uint2* __restrict__ rank, int* __restrict__ nbr, long long ld, long long cap_out,
    int* __restrict__ nbr_t, long long ld_t) {
  long long i = (long long)blockIdx.x * kThreads + threadIdx.x;
  if (i >= rows_of(n, d_n)) return;
  const int4 c = *reinterpret_cast<const int4*>(coords + 4 * i);
  if (!site_ok(c, g)) {           // reported by conv_mark; feeds nothing
    if (nbr_t)
      for (int k = 0; k < g.kvol; ++k) nbr_t[(long long)k * ld_t + i] = -1;
    return;
  }
  for (int kz = 0; kz < g.k[0]; ++kz) {
    const int tz = c.y + g.p[0] - kz;
    const bool vz = tz >= 0 && (tz % g.s[0]) == 0 && (tz / g.s[0]) < g.out_shape[0];
    for (int ky = 0; ky < g.k[1]; ++ky) {
      const int ty = c.z + g.p[1] - ky;
      const bool vy = vz && ty >= 0 && (ty % g.s[1]) == 0 && (ty / g.s[1]) < g.out_shape[1];
      for (int kx = 0; kx < g.k[2]; ++kx) {
        const int k = (kz * g.k[1] + ky) * g.k[2] + kx;
        const int tx = c.w + g.p[2] - kx;
        int row = -1;
        if (vy && tx >= 0 && (tx % g.s[2]) == 0 && (tx / g.s[2]) < g.out_shape[2]) {
          row = rank_lookup(rank, lin_idx(c.x, tz / g.s[0], ty / g.s[1], tx / g.s[2], g.out_shape));
          if (row >= cap_out) row = -1;
          if (row >= 0) nbr[(long long)k * ld + row] = (int)i;
        }
        if (nbr_t) nbr_t[(long long)k * ld_t + i] = row;
      }
    }
  }
}

// submanifold rulebook over a set that carries a rank table (outputs of a strided conv: sorted rows)
__global__ __launch_bounds__(kThreads) void subm_nbr_ranked(const int* __restrict__ coords,
                                                            long long n, const long long* __restrict__ d_n,
                                                            ConvGeom g, const uint2* __restrict__ rank,
                                                            int* __restrict__ nbr, long long ld) {
  const long long o = (long long)blockIdx.x * kThreads + threadIdx.x;
  const int k = blockIdx.y;                          // first half of the offsets + centre (see subm_nbr)
  n = rows_of(n, d_n);
  if (o >= n) return;
  if (2 * k + 1 == g.kvol) {
    nbr[(long long)k * ld + o] = (int)o;
    return;
  }
  const int kx = k % g.k[2], ky = (k / g.k[2]) % g.k[1], kz = k / (g.k[2] * g.k[1]);
  const int4 c = *reinterpret_cast<const int4*>(coords + 4 * o);
  const int z = c.y + kz - g.k[0] / 2, y = c.z + ky - g.k[1] / 2, x = c.w + kx - g.k[2] / 2;
  int r = -1;
  if (c.x >= 0 && c.x < g.batch && z >= 0 && z < g.in_shape[0] && y >= 0 && y < g.in_shape[1] && x >= 0 &&
      x < g.in_shape[2])
    r = rank_lookup(rank, lin_idx(c.x, z, y, x, g.in_shape));
  if (r >= n) r = -1;             // a rank table built for more rows than this call covers (capacity overflow)
  nbr[(long long)k * ld + o] = r;
  if (r >= 0) nbr[(long long)(g.kvol - 1 - k) * ld + r] = (int)o;
}

// ---- pair-list view: wavefront-ballot compaction ------------------------------------------------
__global__ __launch_bounds__(kThreads) void pairs_count(const int* __restrict__ nbr, long long ld,
                                                        long long n_out, long long nw,
                                                        int* __restrict__ wcnt) {
  const long long o = (long long)blockIdx.x * kThreads + threadIdx.x;
  const int k = blockIdx.y;
  const bool valid = o < n_out && nbr[(long long)k * ld + o] >= 0;
  const unsigned long long m = __ballot(valid);
  if ((threadIdx.x & 63) == 0 && (o >> 6) < nw) wcnt[(long long)k * nw + (o >> 6)] = __popcll(m);
}

__global__ __launch_bounds__(kThreads) void pairs_write(const int* __restrict__ nbr, long long ld,
                                                        long long n_out, long long nw,
                                                        const int* __restrict__ woff,
                                                        int* __restrict__ pairs) {
  const long long o = (long long)blockIdx.x * kThreads + threadIdx.x;
  const int k = blockIdx.y;
  const int i = o < n_out ? nbr[(long long)k * ld + o] : -1;
  const unsigned long long m = __ballot(i >= 0);
  if (i < 0) return;
  const int lane = threadIdx.x & 63;
  const int pos = woff[(long long)k * nw + (o >> 6)] - woff[(long long)k * nw] +
                  __popcll(m & ((1ull << lane) - 1ull));
  int* pin = pairs + (long long)k * 2 * n_out;
  pin[pos] = i;
  pin[n_out + pos] = (int)o;
}

__global__ void pairs_num(const int* __restrict__ woff, const int* __restrict__ wcnt, long long nw,
                          int kvol, int* __restrict__ pair_num) {
  const int k = threadIdx.x;
  if (k >= kvol) return;
  const long long last = (long long)(k + 1) * nw - 1;
  pair_num[k] = woff[last] + wcnt[last] - woff[(long long)k * nw];
}


// ---- mask-sorted row order --------------------------------------------------------------------
// The conv kernel skips a kernel offset for a 32-row MFMA group only when no row of the group uses
// it. Rows are therefore re-grouped: inside windows of 16384 consecutive rows (locality for the
// gathers is kept) rows are sorted by their 27-bit "offsets used" mask, so groups hold rows with
// equal or similar masks and far fewer (group, offset) slots carry absent neighbours
// (measured on FHD rulebooks: useful MFMA rows 0.58 -> 0.76 at C=64, 0.46 -> 0.79 at C=32).
// One 1024-thread workgroup per window: masks from the table, bitonic sort of 64-bit
// (mask << 32 | local row) keys in LDS (128 KB), then the table is permuted once per rulebook.
constexpr int kSortTile = 16384;
constexpr int kSortThreads = 1024;

__global__ __launch_bounds__(kSortThreads) void mask_sort_tiles(const int* __restrict__ nbr,
                                                                long long ld, int kvol,
                                                                long long n, int* __restrict__ perm) {
  extern __shared__ unsigned long long skey[];  // [kSortTile]
  const long long base = (long long)blockIdx.x * kSortTile;
  for (int j = threadIdx.x; j < kSortTile; j += kSortThreads) {
    const long long o = base + j;
    unsigned long long key = ~0ull;
    if (o < n) {
      unsigned m = 0;
      for (int k = 0; k < kvol && k < 32; ++k) m |= (nbr[(long long)k * ld + o] >= 0 ? 1u : 0u) << k;
      key = ((unsigned long long)m << 32) | (unsigned)j;
    }
    skey[j] = key;
  }
  __syncthreads();
  for (int k = 2; k <= kSortTile; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int t = threadIdx.x; t < kSortTile / 2; t += kSortThreads) {
        const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));  // index with bit j clear
        const int p = i | j;
        const unsigned long long a = skey[i], b = skey[p];
        const bool up = (i & k) == 0;
        if ((a > b) == up) {
          skey[i] = b;
          skey[p] = a;
        }
      }
      __syncthreads();
    }
  }
  for (int j = threadIdx.x; j < kSortTile; j += kSortThreads) {
    const long long o = base + j;
    if (o < n) perm[o] = (int)(base + (long long)(unsigned)(skey[j] & 0xffffffffull));
  }
}

__global__ __launch_bounds__(kThreads) void permute_table(const int* __restrict__ nbr, long long ld,
                                                          long long n, const int* __restrict__ perm,
                                                          int* __restrict__ out, long long ld_out) {
  const long long j = (long long)blockIdx.x * kThreads + threadIdx.x;
  const int k = blockIdx.y;
  if (j >= n) return;
  out[(long long)k * ld_out + j] = nbr[(long long)k * ld + perm[j]];
}

int fill_geom(ConvGeom* g, const int32_t* in_shape, const int32_t* ksize, const int32_t* stride,
              const int32_t* pad, bool subm, int batch) {
  g->kvol = 1;
  g->batch = batch;
  for (int d = 0; d < 3; ++d) {
    g->in_shape[d] = in_shape[d];
    g->k[d] = ksize[d];
    g->s[d] = subm ? 1 : stride[d];
    g->p[d] = subm ? ksize[d] / 2 : pad[d];
    if (in_shape[d] < 1 || g->k[d] < 1 || g->s[d] < 1 || g->p[d] < 0) return -1;
    g->out_shape[d] = (in_shape[d] + 2 * g->p[d] - g->k[d]) / g->s[d] + 1;
    if (g->out_shape[d] < 1) return -1;
    g->kvol *= g->k[d];
  }
  if (subm)
    for (int d = 0; d < 3; ++d)
      if (!(g->k[d] & 1)) return -1;
  return g->kvol <= 125 ? 0 : -1;
}

struct RbWs {
  VxSlot* table;
  void* rank_tmp;
  long long cap;
  unsigned* bitmap;
  unsigned char* bytemap;
  int* base;
  long long nwords;
  long long* total;
  void* scan_ws;
  size_t scan_bytes;
  size_t bytes;
};

RbWs carve(void* basep, int64_t n_in, int batch, const int32_t* out_shape) {
  RbWs w;
  char* p = (char*)basep;
  size_t o = 0;
  auto take = [&](size_t b) {
    char* r = p ? p + o : nullptr;
    o += vx_align(b);
    return r;
  };
  const int64_t nn = n_in > 0 ? n_in : 1;
  w.cap = vx_hash_capacity(nn);
  w.table = out_shape ? nullptr : (VxSlot*)take(sizeof(VxSlot) * (size_t)w.cap);  // subm only
  w.rank_tmp = nullptr;
  w.total = (long long*)take(64);
  w.nwords = 0;
  w.bitmap = nullptr;
  w.bytemap = nullptr;
  w.base = nullptr;
  w.scan_ws = nullptr;
  w.scan_bytes = 0;
  if (out_shape) {
    const long long cells = (long long)batch * out_shape[0] * out_shape[1] * out_shape[2];
    w.nwords = (cells + 31) / 32;
    w.bitmap = (unsigned*)take(4 * (size_t)w.nwords);
    w.bytemap = (unsigned char*)take(32 * (size_t)w.nwords);
    w.base = (int*)take(4 * (size_t)w.nwords);
    w.rank_tmp = take(8 * (size_t)w.nwords);
    w.scan_bytes = vx_scan_workspace_bytes(w.nwords);
    w.scan_ws = take(w.scan_bytes);
  }
  w.bytes = o;
  return w;
}
}  // namespace

extern "C" size_t vx_rulebook_workspace_bytes(int64_t n_in, int32_t batch,
                                              const int32_t* out_shape) {
  return carve(nullptr, n_in, batch, out_shape).bytes;
}

static int32_t build_table(const int32_t* d_coords, int64_t n, const long long* d_n, const ConvGeom& g,
                           const RbWs& w, hipStream_t stream) {
  hash_clear<<<(unsigned)vx_cdiv(w.cap, kThreads), kThreads, 0, stream>>>(w.table, w.cap);
  VX_LAUNCH_CHECK();
  hash_build<<<(unsigned)vx_cdiv(n, kThreads), kThreads, 0, stream>>>(d_coords, n, d_n, g, w.table,
                                                                      w.cap - 1);
  VX_LAUNCH_CHECK();
  return VX_OK;
}

// n = row count (d_n == NULL) or capacity (d_n = device row count)
static int32_t subm_impl(const int32_t* d_coords, int64_t n, const int64_t* d_n, int32_t batch,
                         const int32_t* shape, const int32_t* ksize, int32_t* d_nbr, int64_t nbr_ld,
                         void* d_ws, size_t ws_bytes, hipStream_t stream) {
  VX_REQUIRE(shape && ksize && d_ws, "rulebook_subm: null pointer");
  VX_REQUIRE(n >= 0 && n < (1ll << 31) && batch >= 1, "rulebook_subm: bad n/batch");
  ConvGeom g;
  VX_REQUIRE(fill_geom(&g, shape, ksize, nullptr, nullptr, true, batch) == 0,
             "rulebook_subm: bad geometry (odd kernel sizes <= 125 offsets required)");
  if (n == 0) return VX_OK;
  VX_REQUIRE(d_coords && d_nbr && nbr_ld >= n, "rulebook_subm: bad table pointer / stride");
  RbWs w = carve(d_ws, n, batch, nullptr);
  if (ws_bytes < w.bytes) {
    vx_set_error("rulebook_subm: workspace %zu < %zu", ws_bytes, w.bytes);
    return VX_ERR_WORKSPACE;
  }
  int32_t rc = build_table(d_coords, n, (const long long*)d_n, g, w, stream);
  if (rc) return rc;
  const int half = g.kvol / 2;                       // odd kvol: offsets 0..half-1 mirror half+1..kvol-1
  if (half > 0)
    if (int32_t rc_ = vx_fill_async(d_nbr + (size_t)(half + 1) * nbr_ld, 0xff, sizeof(int32_t) * (size_t)half * nbr_ld, stream)) return rc_;
  dim3 grid((unsigned)vx_cdiv(n, kThreads), (unsigned)(half + 1));
  subm_nbr<<<grid, kThreads, 0, stream>>>(d_coords, n, (const long long*)d_n, g, w.table, w.cap - 1, d_nbr,
                                          nbr_ld);
  VX_LAUNCH_CHECK();
  return VX_OK;
}

extern "C" int32_t vx_rulebook_subm(const int32_t* d_coords, int64_t n, int32_t batch,
                                    const int32_t* shape, const int32_t* ksize, int32_t* d_nbr,
                                    int64_t nbr_ld, void* d_ws, size_t ws_bytes, void* stream) {
  return subm_impl(d_coords, n, nullptr, batch, shape, ksize, d_nbr, nbr_ld, d_ws, ws_bytes, (hipStream_t)stream);
}

extern "C" int32_t vx_rulebook_subm_dn(const int32_t* d_coords, const int64_t* d_n, int64_t cap, int32_t batch,
                                       const int32_t* shape, const int32_t* ksize, int32_t* d_nbr,
                                       int64_t nbr_ld, void* d_ws, size_t ws_bytes, void* stream) {
  VX_REQUIRE(d_n && cap >= 1, "rulebook_subm_dn: need the device row count and a capacity >= 1");
  return subm_impl(d_coords, cap, d_n, batch, shape, ksize, d_nbr, nbr_ld, d_ws, ws_bytes, (hipStream_t)stream);
}

// phase 1 of the strided conv: occupancy of the output grid -> sorted rank {bits, base} per 32 cells, total -> d_total
static int32_t conv_count_impl(const int32_t* d_coords, int64_t n_in, const int64_t* d_n_in, const ConvGeom& g,
                               const RbWs& w, int64_t* d_total, long long* d_err, hipStream_t stream) {
  if (int32_t rc_ = vx_fill_async(w.bytemap, 0, 32 * (size_t)w.nwords, stream)) return rc_;
  conv_mark<<<(unsigned)vx_cdiv(n_in, kThreads), kThreads, 0, stream>>>(d_coords, n_in, (const long long*)d_n_in, g,
                                                                        w.bytemap, d_err);
  VX_LAUNCH_CHECK();
  bitmap_pack<<<(unsigned)vx_cdiv(w.nwords, kThreads), kThreads, 0, stream>>>(w.bytemap, w.nwords,
                                                                              w.bitmap, w.base);
  VX_LAUNCH_CHECK();
  return vx_scan_exclusive_i32(w.base, w.base, w.nwords, d_total, w.scan_ws, w.scan_bytes, stream);
}

// phase 2: output coordinates, rank table, neighbour table(s). cap_out = rows the output buffers hold.
static int32_t conv_emit_impl(const int32_t* d_coords, int64_t n_in, const int64_t* d_n_in, const ConvGeom& g,
                              const RbWs& w, int64_t cap_out, int32_t* d_out_coords, int32_t* d_nbr,
                              int64_t nbr_ld, int32_t* d_nbr_t, int64_t nbr_t_ld, void* d_rank,
                              hipStream_t stream) {
  const unsigned gw = (unsigned)vx_cdiv(w.nwords, kThreads);
  conv_emit_coords<<<gw, kThreads, 0, stream>>>(w.bitmap, w.base, w.nwords, g, d_out_coords, cap_out);
  VX_LAUNCH_CHECK();
  // rank table {bits, base}: caller's buffer when it wants to keep it, else the hash area of the workspace
  uint2* rank = d_rank ? (uint2*)d_rank : (uint2*)w.rank_tmp;
  rank_interleave<<<gw, kThreads, 0, stream>>>(w.bitmap, w.base, w.nwords, rank);
  VX_LAUNCH_CHECK();
  if (int32_t rc_ = vx_fill_async(d_nbr, 0xff, 4 * (size_t)g.kvol * (size_t)nbr_ld, stream)) return rc_;
  conv_pairs_from_inputs<<<(unsigned)vx_cdiv(n_in, kThreads), kThreads, 0, stream>>>(
      d_coords, n_in, (const long long*)d_n_in, g, rank, d_nbr, nbr_ld, cap_out, d_nbr_t, nbr_t_ld);
  VX_LAUNCH_CHECK();
  return VX_OK;
}

extern "C" int32_t vx_rulebook_conv_count(const int32_t* d_coords, int64_t n_in, int32_t batch,
                                          const int32_t* in_shape, const int32_t* ksize,
                                          const int32_t* stride, const int32_t* pad,
                                          int32_t* out_shape, int64_t* h_n_out, void* d_ws,
                                          size_t ws_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  VX_REQUIRE(in_shape && ksize && stride && pad && out_shape && h_n_out && d_ws,
             "rulebook_conv_count: null pointer");
  VX_REQUIRE(n_in >= 0 && n_in < (1ll << 31) && batch >= 1, "rulebook_conv_count: bad n/batch");
  ConvGeom g;
  VX_REQUIRE(fill_geom(&g, in_shape, ksize, stride, pad, false, batch) == 0,
             "rulebook_conv_count: bad geometry");
  for (int d = 0; d < 3; ++d) out_shape[d] = g.out_shape[d];
  *h_n_out = 0;
  if (n_in == 0) return VX_OK;
  VX_REQUIRE(d_coords, "rulebook_conv_count: null coords");
  RbWs w = carve(d_ws, n_in, batch, g.out_shape);
  if (ws_bytes < w.bytes) {
    vx_set_error("rulebook_conv_count: workspace %zu < %zu", ws_bytes, w.bytes);
    return VX_ERR_WORKSPACE;
  }
  if (int32_t rc_ = vx_fill_async(w.total, 0, 16, stream)) return rc_;          // {output rows, error word}
  int32_t rc = conv_count_impl(d_coords, n_in, nullptr, g, w, (int64_t*)w.total, w.total + 1, stream);
  if (rc) return rc;
  int64_t h[2] = {0, 0};
  VX_HIP(hipMemcpyAsync(h, w.total, sizeof(h), hipMemcpyDeviceToHost, stream));
  VX_HIP(hipStreamSynchronize(stream));
  if (h[1]) {
    vx_set_error("rulebook_conv_count: a coordinate lies outside batch %d x grid [%d,%d,%d] (batch index or site out "
                 "of range)", batch, in_shape[0], in_shape[1], in_shape[2]);
    return VX_ERR_ARG;
  }
  *h_n_out = h[0];
  return VX_OK;
}

extern "C" int32_t vx_rulebook_conv_emit(const int32_t* d_coords, int64_t n_in, int32_t batch,
                                         const int32_t* in_shape, const int32_t* ksize,
                                         const int32_t* stride, const int32_t* pad, int64_t n_out,
                                         int32_t* d_out_coords, int32_t* d_nbr, int64_t nbr_ld,
                                         int32_t* d_nbr_t, int64_t nbr_t_ld, void* d_rank,
                                         void* d_ws, size_t ws_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  VX_REQUIRE(in_shape && ksize && stride && pad && d_ws, "rulebook_conv_emit: null pointer");
  ConvGeom g;
  VX_REQUIRE(fill_geom(&g, in_shape, ksize, stride, pad, false, batch) == 0,
             "rulebook_conv_emit: bad geometry");
  if (n_in == 0 || n_out == 0) return VX_OK;
  VX_REQUIRE(d_coords && d_out_coords && d_nbr && nbr_ld >= n_out,
             "rulebook_conv_emit: bad output pointers / stride");
  VX_REQUIRE(!d_nbr_t || nbr_t_ld >= n_in, "rulebook_conv_emit: bad transpose stride");
  RbWs w = carve(d_ws, n_in, batch, g.out_shape);
  if (ws_bytes < w.bytes) {
    vx_set_error("rulebook_conv_emit: workspace %zu < %zu", ws_bytes, w.bytes);
    return VX_ERR_WORKSPACE;
  }
  return conv_emit_impl(d_coords, n_in, nullptr, g, w, n_out, d_out_coords, d_nbr, nbr_ld, d_nbr_t, nbr_t_ld,
                        d_rank, stream);
}

extern "C" int32_t vx_rulebook_conv_dn(const int32_t* d_coords, const int64_t* d_n_in, int64_t cap_in,
                                       int32_t batch, const int32_t* in_shape, const int32_t* ksize,
                                       const int32_t* stride, const int32_t* pad, int32_t* out_shape,
                                       int64_t cap_out, int32_t* d_out_coords, int64_t* d_n_out,
                                       int32_t* d_nbr, int64_t nbr_ld, void* d_rank, int64_t* d_err,
                                       void* d_ws, size_t ws_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  VX_REQUIRE(in_shape && ksize && stride && pad && out_shape && d_ws && d_n_in && d_n_out && d_err,
             "rulebook_conv_dn: null pointer");
  VX_REQUIRE(cap_in >= 1 && cap_in < (1ll << 31) && cap_out >= 1 && cap_out < (1ll << 31) && batch >= 1,
             "rulebook_conv_dn: bad capacities / batch");
  ConvGeom g;
  VX_REQUIRE(fill_geom(&g, in_shape, ksize, stride, pad, false, batch) == 0, "rulebook_conv_dn: bad geometry");
  for (int d = 0; d < 3; ++d) out_shape[d] = g.out_shape[d];
  VX_REQUIRE(d_coords && d_out_coords && d_nbr && nbr_ld >= cap_out, "rulebook_conv_dn: bad output pointers / stride");
  RbWs w = carve(d_ws, cap_in, batch, g.out_shape);
  if (ws_bytes < w.bytes) {
    vx_set_error("rulebook_conv_dn: workspace %zu < %zu", ws_bytes, w.bytes);
    return VX_ERR_WORKSPACE;
  }
  int32_t rc = conv_count_impl(d_coords, cap_in, d_n_in, g, w, d_n_out, (long long*)d_err, stream);
  if (rc) return rc;
  return conv_emit_impl(d_coords, cap_in, d_n_in, g, w, cap_out, d_out_coords, d_nbr, nbr_ld, nullptr, 0, d_rank,
                        stream);
}

extern "C" size_t vx_rank_table_bytes(int32_t batch, const int32_t* shape) {
  const long long cells = (long long)batch * shape[0] * shape[1] * shape[2];
  return (size_t)((cells + 31) / 32) * 8;
}

static int32_t subm_ranked_impl(const int32_t* d_coords, int64_t n, const int64_t* d_n, int32_t batch,
                                const int32_t* shape, const int32_t* ksize, const void* d_rank,
                                int32_t* d_nbr, int64_t nbr_ld, hipStream_t stream) {
  VX_REQUIRE(shape && ksize && d_rank, "rulebook_subm_ranked: null pointer");
  VX_REQUIRE(n >= 0 && n < (1ll << 31) && batch >= 1, "rulebook_subm_ranked: bad n/batch");
  ConvGeom g;
  VX_REQUIRE(fill_geom(&g, shape, ksize, nullptr, nullptr, true, batch) == 0,
             "rulebook_subm_ranked: bad geometry (odd kernel sizes <= 125 offsets required)");
  if (n == 0) return VX_OK;
  VX_REQUIRE(d_coords && d_nbr && nbr_ld >= n, "rulebook_subm_ranked: bad table pointer / stride");
  const int half = g.kvol / 2;
  if (half > 0)
    if (int32_t rc_ = vx_fill_async(d_nbr + (size_t)(half + 1) * nbr_ld, 0xff, sizeof(int32_t) * (size_t)half * nbr_ld, stream)) return rc_;
  dim3 grid((unsigned)vx_cdiv(n, kThreads), (unsigned)(half + 1));
  subm_nbr_ranked<<<grid, kThreads, 0, stream>>>(d_coords, n, (const long long*)d_n, g, (const uint2*)d_rank,
                                                 d_nbr, nbr_ld);
  VX_LAUNCH_CHECK();
  return VX_OK;
}

extern "C" int32_t vx_rulebook_subm_ranked(const int32_t* d_coords, int64_t n, int32_t batch,
                                           const int32_t* shape, const int32_t* ksize,
                                           const void* d_rank, int32_t* d_nbr, int64_t nbr_ld,
                                           void* stream) {
  return subm_ranked_impl(d_coords, n, nullptr, batch, shape, ksize, d_rank, d_nbr, nbr_ld, (hipStream_t)stream);
}

extern "C" int32_t vx_rulebook_subm_ranked_dn(const int32_t* d_coords, const int64_t* d_n, int64_t cap,
                                              int32_t batch, const int32_t* shape, const int32_t* ksize,
                                              const void* d_rank, int32_t* d_nbr, int64_t nbr_ld,
                                              void* stream) {
  VX_REQUIRE(d_n && cap >= 1, "rulebook_subm_ranked_dn: need the device row count and a capacity >= 1");
  return subm_ranked_impl(d_coords, cap, d_n, batch, shape, ksize, d_rank, d_nbr, nbr_ld, (hipStream_t)stream);
}

extern "C" int32_t vx_rulebook_sort_rows(const int32_t* d_nbr, int64_t nbr_ld, int32_t kvol,
                                         int64_t n_out, int32_t* d_nbr_sorted, int64_t sorted_ld,
                                         int32_t* d_perm, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  VX_REQUIRE(kvol >= 1 && kvol <= 125 && n_out >= 0, "rulebook_sort_rows: bad arguments");
  if (n_out == 0) return VX_OK;
  VX_REQUIRE(d_nbr && d_nbr_sorted && d_perm && nbr_ld >= n_out && sorted_ld >= n_out,
             "rulebook_sort_rows: null pointer / stride");
  const size_t lds = (size_t)kSortTile * sizeof(unsigned long long);
  VX_HIP(hipFuncSetAttribute((const void*)mask_sort_tiles,
                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  mask_sort_tiles<<<(unsigned)vx_cdiv(n_out, kSortTile), kSortThreads, lds, stream>>>(
      d_nbr, nbr_ld, kvol, n_out, d_perm);
  VX_LAUNCH_CHECK();
  dim3 grid((unsigned)vx_cdiv(n_out, kThreads), (unsigned)kvol);
  permute_table<<<grid, kThreads, 0, stream>>>(d_nbr, nbr_ld, n_out, d_perm, d_nbr_sorted, sorted_ld);
  VX_LAUNCH_CHECK();
  return VX_OK;
}

extern "C" size_t vx_pairs_workspace_bytes(int32_t kvol, int64_t n_out) {
  const int64_t nw = vx_cdiv(n_out > 0 ? n_out : 1, 64);
  const size_t a = vx_align(4 * (size_t)kvol * (size_t)nw);
  return 2 * a + vx_scan_workspace_bytes((int64_t)kvol * nw);
}

extern "C" int32_t vx_nbr_to_pairs(const int32_t* d_nbr, int64_t nbr_ld, int32_t kvol,
                                   int64_t n_out, int32_t* d_pairs, int32_t* d_pair_num,
                                   void* d_ws, size_t ws_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  VX_REQUIRE(kvol >= 1 && kvol <= 125 && n_out >= 0 && d_pair_num, "nbr_to_pairs: bad arguments");
  if (n_out == 0) {
    if (int32_t rc_ = vx_fill_async(d_pair_num, 0, 4 * (size_t)kvol, stream)) return rc_;
    return VX_OK;
  }
  VX_REQUIRE(d_nbr && d_pairs && d_ws && nbr_ld >= n_out, "nbr_to_pairs: null pointer / stride");
  if (ws_bytes < vx_pairs_workspace_bytes(kvol, n_out)) {
    vx_set_error("nbr_to_pairs: workspace too small");
    return VX_ERR_WORKSPACE;
  }
  const int64_t nw = vx_cdiv(n_out, 64);
  const size_t a = vx_align(4 * (size_t)kvol * (size_t)nw);
  int* wcnt = (int*)d_ws;
  int* woff = (int*)((char*)d_ws + a);
  void* sws = (char*)d_ws + 2 * a;
  if (int32_t rc_ = vx_fill_async(d_pairs, 0xff, 4 * (size_t)kvol * 2 * (size_t)n_out, stream)) return rc_;
  dim3 grid((unsigned)vx_cdiv(n_out, kThreads), (unsigned)kvol);
  pairs_count<<<grid, kThreads, 0, stream>>>(d_nbr, nbr_ld, n_out, nw, wcnt);
  VX_LAUNCH_CHECK();
  int32_t rc = vx_scan_exclusive_i32(wcnt, woff, (int64_t)kvol * nw, nullptr, sws,
                                     ws_bytes - 2 * a, stream);
  if (rc) return rc;
  pairs_write<<<grid, kThreads, 0, stream>>>(d_nbr, nbr_ld, n_out, nw, woff, d_pairs);
  VX_LAUNCH_CHECK();
  pairs_num<<<1, 128, 0, stream>>>(woff, wcnt, nw, kvol, d_pair_num);
  VX_LAUNCH_CHECK();
  return VX_OK;
}
