/* ORACLE (test infrastructure, not product code).
 *
 * Plain-C CPU restatement of the sparse-convolution arithmetic the reference
 * obtains from the third-party `spconv` package (PyPI spconv-cu111, "2.0+",
 * unpinned, NOT vendored in /root/reference: README.md:19,44-53). The algorithm
 * restated here is spconv's published native CPU one: a hash map from voxel
 * coordinate to row, rulebook (kernel offset, input row, output row) pairs, and
 * per-offset gather -> GEMM -> scatter-add. Call sites that fix the semantics:
 *   det3d/models/backbones/scn.py:28-38,333      SubMConv3d k=3 (stage convs)
 *   det3d/models/backbones/scn.py:338-361        SparseConv3d geometries
 *   det3d/models/backbones/multiview.py:68-86    (1,1,3)/(1,1,2)/(0,0,1) convs
 *   det3d/models/backbones/scn.py:375            .dense()
 * The reference holds no tests or golden vectors for this boundary, so this file
 * is "parity unpinned" by the reference; it is cross-checked against dense
 * torch.nn.functional.conv3d (tests/test_oracle_spconv.py).
 *
 * Canonical rulebook order: output sites sorted by
 * linear index ((b*D+z)*H+y)*W+x; kernel offset index (kz*KH+ky)*KW+kx; the
 * rulebook is stored output-major as nbr[k][o] = input row or -1.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * the library built from this file.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define EMPTY_KEY ((int64_t)-1)

typedef struct {
  int64_t *keys;
  int32_t *vals;
  int64_t cap; /* power of two */
} hmap;

static uint64_t mix64(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33;
  x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x;
}

static int hmap_init(hmap *h, int64_t n) {
  int64_t cap = 16;
  while (cap < 2 * n + 2) cap <<= 1;
  h->cap = cap;
  h->keys = (int64_t *)malloc(sizeof(int64_t) * cap);
  h->vals = (int32_t *)malloc(sizeof(int32_t) * cap);
  if (!h->keys || !h->vals) return -1;
  for (int64_t i = 0; i < cap; ++i) h->keys[i] = EMPTY_KEY;
  return 0;
}
static void hmap_free(hmap *h) { free(h->keys); free(h->vals); }

/* returns existing value, or inserts val and returns -1 */
static int32_t hmap_get_or_put(hmap *h, int64_t key, int32_t val) {
  int64_t s = (int64_t)(mix64((uint64_t)key) & (uint64_t)(h->cap - 1));
  for (;;) {
    if (h->keys[s] == key) return h->vals[s];
    if (h->keys[s] == EMPTY_KEY) { h->keys[s] = key; h->vals[s] = val; return -1; }
    s = (s + 1) & (h->cap - 1);
  }
}
static int32_t hmap_get(const hmap *h, int64_t key) {
  int64_t s = (int64_t)(mix64((uint64_t)key) & (uint64_t)(h->cap - 1));
  for (;;) {
    if (h->keys[s] == key) return h->vals[s];
    if (h->keys[s] == EMPTY_KEY) return -1;
    s = (s + 1) & (h->cap - 1);
  }
}

static inline int64_t lin(const int32_t *c, const int32_t *shape) {
  return (((int64_t)c[0] * shape[0] + c[1]) * shape[1] + c[2]) * shape[2] + c[3];
}

int orc_version(void) { return 1; }

/* out_shape[d] = (in + 2p - k) / s + 1 */
void orc_conv_out_shape(const int32_t *in_shape, const int32_t *ksize, const int32_t *stride,
                        const int32_t *pad, int32_t *out_shape) {
  for (int d = 0; d < 3; ++d)
    out_shape[d] = (in_shape[d] + 2 * pad[d] - ksize[d]) / stride[d] + 1;
}

/* Submanifold rulebook: outputs == inputs (same rows). nbr is int32[K][n]. */
int orc_rulebook_subm(const int32_t *coords, int64_t n, const int32_t *shape,
                      const int32_t *ksize, int32_t *nbr) {
  hmap h;
  if (hmap_init(&h, n)) return -1;
  for (int64_t i = 0; i < n; ++i) hmap_get_or_put(&h, lin(coords + 4 * i, shape), (int32_t)i);
  const int KD = ksize[0], KH = ksize[1], KW = ksize[2];
  for (int kz = 0; kz < KD; ++kz)
    for (int ky = 0; ky < KH; ++ky)
      for (int kx = 0; kx < KW; ++kx) {
        int k = (kz * KH + ky) * KW + kx;
        for (int64_t o = 0; o < n; ++o) {
          const int32_t *c = coords + 4 * o;
          int32_t q[4] = {c[0], c[1] + kz - KD / 2, c[2] + ky - KH / 2, c[3] + kx - KW / 2};
          int32_t r = -1;
          if (q[1] >= 0 && q[1] < shape[0] && q[2] >= 0 && q[2] < shape[1] && q[3] >= 0 &&
              q[3] < shape[2])
            r = hmap_get(&h, lin(q, shape));
          nbr[(int64_t)k * n + o] = r;
        }
      }
  hmap_free(&h);
  return 0;
}

static int cmp_i64(const void *a, const void *b) {
  int64_t x = *(const int64_t *)a, y = *(const int64_t *)b;
  return (x > y) - (x < y);
}

/* Strided sparse conv rulebook. Pass 1 (out_coords == NULL): returns N_out.
 * Pass 2: fills out_coords int32[N_out][4] (sorted by linear index) and
 * nbr int32[K][N_out]. Follows spconv's CPU scheme: every input row enumerates
 * the outputs it feeds; new outputs enter a hash map; afterwards outputs are put
 * in canonical (sorted) order. */
int64_t orc_rulebook_conv(const int32_t *coords, int64_t n, const int32_t *in_shape,
                          const int32_t *ksize, const int32_t *stride, const int32_t *pad,
                          int32_t *out_coords, int32_t *nbr, int64_t n_out_expected) {
  int32_t osh[3];
  orc_conv_out_shape(in_shape, ksize, stride, pad, osh);
  const int KD = ksize[0], KH = ksize[1], KW = ksize[2];
  const int K = KD * KH * KW;
  hmap h;
  if (hmap_init(&h, n * (int64_t)(K < 8 ? K : 8) + 16)) return -1;
  int64_t cap_keys = 1024, n_out = 0;
  int64_t *okeys = (int64_t *)malloc(sizeof(int64_t) * cap_keys);
  for (int64_t i = 0; i < n; ++i) {
    const int32_t *c = coords + 4 * i;
    for (int kz = 0; kz < KD; ++kz) {
      int tz = c[1] + pad[0] - kz;
      if (tz < 0 || tz % stride[0]) continue;
      int oz = tz / stride[0];
      if (oz >= osh[0]) continue;
      for (int ky = 0; ky < KH; ++ky) {
        int ty = c[2] + pad[1] - ky;
        if (ty < 0 || ty % stride[1]) continue;
        int oy = ty / stride[1];
        if (oy >= osh[1]) continue;
        for (int kx = 0; kx < KW; ++kx) {
          int tx = c[3] + pad[2] - kx;
          if (tx < 0 || tx % stride[2]) continue;
          int ox = tx / stride[2];
          if (ox >= osh[2]) continue;
          int32_t q[4] = {c[0], oz, oy, ox};
          int64_t key = lin(q, osh);
          if (hmap_get_or_put(&h, key, (int32_t)n_out) < 0) {
            if (n_out == cap_keys) {
              cap_keys *= 2;
              okeys = (int64_t *)realloc(okeys, sizeof(int64_t) * cap_keys);
            }
            okeys[n_out++] = key;
          }
        }
      }
    }
  }
  hmap_free(&h);
  if (!out_coords) { free(okeys); return n_out; }
  if (n_out != n_out_expected) { free(okeys); return -2; }
  qsort(okeys, (size_t)n_out, sizeof(int64_t), cmp_i64);
  for (int64_t o = 0; o < n_out; ++o) {
    int64_t key = okeys[o];
    int32_t *q = out_coords + 4 * o;
    q[3] = (int32_t)(key % osh[2]); key /= osh[2];
    q[2] = (int32_t)(key % osh[1]); key /= osh[1];
    q[1] = (int32_t)(key % osh[0]); key /= osh[0];
    q[0] = (int32_t)key;
  }
  free(okeys);
  /* neighbour table: probe the INPUT map from every output site */
  hmap hin;
  if (hmap_init(&hin, n)) return -1;
  for (int64_t i = 0; i < n; ++i) hmap_get_or_put(&hin, lin(coords + 4 * i, in_shape), (int32_t)i);
  for (int kz = 0; kz < KD; ++kz)
    for (int ky = 0; ky < KH; ++ky)
      for (int kx = 0; kx < KW; ++kx) {
        int k = (kz * KH + ky) * KW + kx;
        for (int64_t o = 0; o < n_out; ++o) {
          const int32_t *c = out_coords + 4 * o;
          int32_t q[4] = {c[0], c[1] * stride[0] - pad[0] + kz, c[2] * stride[1] - pad[1] + ky,
                          c[3] * stride[2] - pad[2] + kx};
          int32_t r = -1;
          if (q[1] >= 0 && q[1] < in_shape[0] && q[2] >= 0 && q[2] < in_shape[1] && q[3] >= 0 &&
              q[3] < in_shape[2])
            r = hmap_get(&hin, lin(q, in_shape));
          nbr[(int64_t)k * n_out + o] = r;
        }
      }
  hmap_free(&hin);
  return n_out;
}

/* pair-list view of a neighbour table: pairs int32[K][2][n_out] (in,out), pair_num int32[K];
 * per offset ordered by output row. */
void orc_nbr_to_pairs(const int32_t *nbr, int K, int64_t n_out, int32_t *pairs, int32_t *pair_num) {
  for (int k = 0; k < K; ++k) {
    int32_t c = 0;
    int32_t *pin = pairs + (int64_t)k * 2 * n_out, *pout = pin + n_out;
    for (int64_t o = 0; o < n_out; ++o) {
      int32_t i = nbr[(int64_t)k * n_out + o];
      if (i >= 0) { pin[c] = i; pout[c] = (int32_t)o; ++c; }
    }
    for (int64_t t = c; t < n_out; ++t) { pin[t] = -1; pout[t] = -1; }
    pair_num[k] = c;
  }
}

/* The three arithmetic routines exist in two precisions: fp32 (the reference's own precision, the oracle proper)
 * and fp64 (suffix _f64: the "truth" that tests use to tell the GPU's rounding error from the fp32 oracle's own). */
#define ORC_DEFINE_CONV(REAL, SFX)                                                                        \
  /* weights arrive KRSC [Cout][K][Cin]; repack to [K][Cin][Cout] for a unit-stride inner loop */         \
  static REAL *repack_kio##SFX(const REAL *w, int K, int cin, int cout) {                                 \
    REAL *r = (REAL *)malloc(sizeof(REAL) * (size_t)K * cin * cout);                                      \
    for (int co = 0; co < cout; ++co)                                                                     \
      for (int k = 0; k < K; ++k)                                                                         \
        for (int ci = 0; ci < cin; ++ci)                                                                  \
          r[((size_t)k * cin + ci) * cout + co] = w[((size_t)co * K + k) * cin + ci];                     \
    return r;                                                                                             \
  }                                                                                                       \
  /* forward: out[o] = bias + sum_k sum_ci in[nbr[k][o]][ci] * W[co][k][ci], k then ci order.             \
   * Equivalent to per-offset gather-GEMM-scatter; written output-stationary so threads never             \
   * share an output row (deterministic for any thread count). */                                         \
  int orc_conv_fwd##SFX(const REAL *in, int64_t n_in, int cin, const REAL *w, int K, int cout,            \
                        const int32_t *nbr, int64_t n_out, const REAL *bias, REAL *out) {                 \
    (void)n_in;                                                                                           \
    REAL *wk = repack_kio##SFX(w, K, cin, cout);                                                          \
    if (!wk) return -1;                                                                                   \
    _Pragma("omp parallel for schedule(dynamic, 64)")                                                     \
    for (int64_t o = 0; o < n_out; ++o) {                                                                 \
      REAL *dst = out + o * cout;                                                                         \
      for (int co = 0; co < cout; ++co) dst[co] = bias ? bias[co] : (REAL)0;                              \
      for (int k = 0; k < K; ++k) {                                                                       \
        int32_t i = nbr[(int64_t)k * n_out + o];                                                          \
        if (i < 0) continue;                                                                              \
        const REAL *src = in + (int64_t)i * cin;                                                          \
        const REAL *wp = wk + (size_t)k * cin * cout;                                                     \
        for (int ci = 0; ci < cin; ++ci) {                                                                \
          const REAL a = src[ci];                                                                         \
          const REAL *wr = wp + (size_t)ci * cout;                                                        \
          for (int co = 0; co < cout; ++co) dst[co] += a * wr[co];                                        \
        }                                                                                                 \
      }                                                                                                   \
    }                                                                                                     \
    free(wk);                                                                                             \
    return 0;                                                                                             \
  }                                                                                                       \
  /* dgrad: din[i] = sum over (k,o) with nbr[k][o]==i of W[:,k,:]^T dout[o]; accumulation order           \
   * k ascending (each (i,k) has at most one o). The table is transposed first so that threads own        \
   * input rows: same sums in the same order for any thread count. */                                     \
  int orc_conv_dgrad##SFX(const REAL *dout, int64_t n_out, int cout, const REAL *w, int K, int cin,       \
                          const int32_t *nbr, int64_t n_in, REAL *din) {                                  \
    int32_t *tr = (int32_t *)malloc(sizeof(int32_t) * (size_t)K * (size_t)(n_in > 0 ? n_in : 1));         \
    if (!tr) return -1;                                                                                   \
    memset(tr, 0xff, sizeof(int32_t) * (size_t)K * (size_t)(n_in > 0 ? n_in : 1));                        \
    for (int k = 0; k < K; ++k)                                                                           \
      for (int64_t o = 0; o < n_out; ++o) {                                                               \
        int32_t i = nbr[(int64_t)k * n_out + o];                                                          \
        if (i >= 0) tr[(int64_t)k * n_in + i] = (int32_t)o;                                               \
      }                                                                                                   \
    _Pragma("omp parallel for schedule(dynamic, 64)")                                                     \
    for (int64_t i = 0; i < n_in; ++i) {                                                                  \
      REAL *dst = din + i * cin;                                                                          \
      for (int ci = 0; ci < cin; ++ci) dst[ci] = (REAL)0;                                                 \
      for (int k = 0; k < K; ++k) {                                                                       \
        const int32_t o = tr[(int64_t)k * n_in + i];                                                      \
        if (o < 0) continue;                                                                              \
        const REAL *g = dout + (int64_t)o * cout;                                                         \
        for (int co = 0; co < cout; ++co) {                                                               \
          const REAL gv = g[co];                                                                          \
          const REAL *wr = w + ((size_t)co * K + k) * cin;                                                \
          for (int ci = 0; ci < cin; ++ci) dst[ci] += gv * wr[ci];                                        \
        }                                                                                                 \
      }                                                                                                   \
    }                                                                                                     \
    free(tr);                                                                                             \
    return 0;                                                                                             \
  }                                                                                                       \
  /* wgrad: dW[co][k][ci] = sum_o dout[o][co] * in[nbr[k][o]][ci] (double accumulation, then REAL);       \
   * dbias[co] = sum_o dout[o][co] if dbias != NULL. */                                                   \
  int orc_conv_wgrad##SFX(const REAL *in, int64_t n_in, int cin, const REAL *dout, int64_t n_out,         \
                          int cout, const int32_t *nbr, int K, REAL *dw, REAL *dbias) {                   \
    (void)n_in;                                                                                           \
    double *acc = (double *)calloc((size_t)cout * K * cin, sizeof(double));                               \
    if (!acc) return -1;                                                                                  \
    _Pragma("omp parallel for schedule(dynamic, 1)") /* a thread owns kernel offset k: ordered sums */    \
    for (int k = 0; k < K; ++k)                                                                           \
      for (int64_t o = 0; o < n_out; ++o) {                                                               \
        int32_t i = nbr[(int64_t)k * n_out + o];                                                          \
        if (i < 0) continue;                                                                              \
        const REAL *src = in + (int64_t)i * cin;                                                          \
        const REAL *g = dout + o * cout;                                                                  \
        for (int co = 0; co < cout; ++co) {                                                               \
          double *a = acc + ((size_t)co * K + k) * cin;                                                   \
          const double gv = g[co];                                                                        \
          for (int ci = 0; ci < cin; ++ci) a[ci] += gv * src[ci];                                         \
        }                                                                                                 \
      }                                                                                                   \
    for (size_t t = 0; t < (size_t)cout * K * cin; ++t) dw[t] = (REAL)acc[t];                             \
    free(acc);                                                                                            \
    if (dbias) {                                                                                          \
      for (int co = 0; co < cout; ++co) {                                                                 \
        double s = 0;                                                                                     \
        for (int64_t o = 0; o < n_out; ++o) s += dout[o * cout + co];                                     \
        dbias[co] = (REAL)s;                                                                              \
      }                                                                                                   \
    }                                                                                                     \
    return 0;                                                                                             \
  }

ORC_DEFINE_CONV(float, )
ORC_DEFINE_CONV(double, _f64)

/* .dense(): zeros [B][C][D][H][W], out[b][c][z][y][x] = feat[row][c] */
int orc_to_dense(const float *feat, const int32_t *coords, int64_t n, int c, int batch,
                 const int32_t *shape, float *dense) {
  const int64_t vol = (int64_t)shape[0] * shape[1] * shape[2];
  memset(dense, 0, sizeof(float) * (size_t)batch * c * vol);
  for (int64_t r = 0; r < n; ++r) {
    const int32_t *q = coords + 4 * r;
    int64_t sp = ((int64_t)q[1] * shape[1] + q[2]) * shape[2] + q[3];
    for (int ch = 0; ch < c; ++ch) dense[((int64_t)q[0] * c + ch) * vol + sp] = feat[r * c + ch];
  }
  return 0;
}

int orc_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
void orc_set_num_threads(int n) {
#ifdef _OPENMP
  omp_set_num_threads(n);
#else
  (void)n;
#endif
}
