// Rotated BEV box overlap / IoU / NMS on gfx950.
// Boxes are [x1, y1, x2, y2, ry]: the axis-aligned extent of the box BEFORE it is turned about its centre by
// ry (corner = c + (dx*cos + dy*sin, -dx*sin + dy*cos), the convention of the reference's kernel).
// Own formulation: the overlap polygon is obtained by clipping rectangle A against the four half-planes of
// rectangle B (Sutherland-Hodgman, at most 8 vertices, all in registers) and measured with the shoelace sum --
// the reference collects edge intersections and inside corners and sorts them by angle
// (det3d/ops/iou3d_nms/src/iou3d_nms_kernel.cu:109-213); the two agree to fp32 rounding on non-degenerate input.
// NMS: 64 x 64 suppression-mask blocks (one wave per block row), then ONE wave walks the boxes in score order
// on the device (no mask copy to the host, unlike iou3d_nms.cpp:98-122) and emits the kept indices.
#include "vx_common.h"

namespace {
constexpr float kEps = 1e-8f;

struct P2 { float x, y; };

__device__ __forceinline__ void box_corners(const float* b, P2 (&c)[4]) {
  const float cx = 0.5f * (b[0] + b[2]), cy = 0.5f * (b[1] + b[3]);
  const float hx = 0.5f * (b[2] - b[0]), hy = 0.5f * (b[3] - b[1]);
  const float cs = cosf(b[4]), sn = sinf(b[4]);
  const float dx[4] = {-hx, hx, hx, -hx}, dy[4] = {-hy, -hy, hy, hy};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    c[i].x = dx[i] * cs + dy[i] * sn + cx;
    c[i].y = -dx[i] * sn + dy[i] * cs + cy;
  }
}

// area of (rectangle a) ∩ (rectangle b)
__device__ float rbox_overlap(const float* a, const float* b) {
  P2 ca[4], cb[4];
  box_corners(a, ca);
  box_corners(b, cb);
  // the corner order (x1,y1),(x2,y1),(x2,y2),(x1,y2) keeps its orientation under the rotation; make "inside" mean
  // "left of the directed edge" whatever the sign of the box extents
  const float orient = (cb[1].x - cb[0].x) * (cb[2].y - cb[1].y) - (cb[1].y - cb[0].y) * (cb[2].x - cb[1].x);
  const float sgn = orient >= 0.0f ? 1.0f : -1.0f;
  P2 poly[8], next[8];
  int n = 4;
#pragma unroll
  for (int i = 0; i < 4; ++i) poly[i] = ca[i];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const P2 p0 = cb[e], p1 = cb[(e + 1) & 3];
    const float ex = p1.x - p0.x, ey = p1.y - p0.y;
    int m = 0;
    for (int i = 0; i < n; ++i) {
      const P2 s = poly[i], t = poly[i + 1 < n ? i + 1 : 0];
      const float ds = sgn * (ex * (s.y - p0.y) - ey * (s.x - p0.x));
      const float dt = sgn * (ex * (t.y - p0.y) - ey * (t.x - p0.x));
      if (ds >= 0.0f && m < 8) next[m++] = s;
      if ((ds >= 0.0f) != (dt >= 0.0f) && m < 8) {
        const float u = ds / (ds - dt);
        next[m].x = s.x + u * (t.x - s.x);
        next[m].y = s.y + u * (t.y - s.y);
        ++m;
      }
    }
    n = m;
    for (int i = 0; i < n; ++i) poly[i] = next[i];
    if (n == 0) break;
  }
  float area2 = 0.0f;
  for (int i = 1; i + 1 < n; ++i)
    area2 += (poly[i].x - poly[0].x) * (poly[i + 1].y - poly[0].y) - (poly[i].y - poly[0].y) * (poly[i + 1].x - poly[0].x);
  return 0.5f * fabsf(area2);
}

__device__ __forceinline__ float rbox_iou(const float* a, const float* b) {
  const float sa = (a[2] - a[0]) * (a[3] - a[1]), sb = (b[2] - b[0]) * (b[3] - b[1]);
  const float so = rbox_overlap(a, b);
  return so / fmaxf(sa + sb - so, kEps);
}

__device__ __forceinline__ float abox_iou(const float* a, const float* b) {
  const float w = fmaxf(fminf(a[2], b[2]) - fmaxf(a[0], b[0]), 0.0f);
  const float h = fmaxf(fminf(a[3], b[3]) - fmaxf(a[1], b[1]), 0.0f);
  const float so = w * h;
  const float sa = (a[2] - a[0]) * (a[3] - a[1]), sb = (b[2] - b[0]) * (b[3] - b[1]);
  return so / fmaxf(sa + sb - so, kEps);
}

template <bool IOU>
__global__ __launch_bounds__(256) void pairwise_kernel(const float* __restrict__ a, int na, const float* __restrict__ b,
                                                       int nb, float* __restrict__ out) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= (long long)na * nb) return;
  const int i = (int)(e / nb), j = (int)(e - (long long)i * nb);
  float ba[5], bb[5];
#pragma unroll
  for (int t = 0; t < 5; ++t) {
    ba[t] = a[i * 5 + t];
    bb[t] = b[j * 5 + t];
  }
  out[e] = IOU ? rbox_iou(ba, bb) : rbox_overlap(ba, bb);
}

// mask[i][cb] bit j: box (64*cb + j) is suppressed by box i (only j > i inside the diagonal block; blocks below
// the diagonal are never read). One wave per (row i, column block cb): lane j computes IoU(i, 64*cb + j) and the
// wave's ballot IS the mask word -- n*n/2 independent pair evaluations instead of 64 serial ones per lane.
template <bool ROTATED>
__global__ __launch_bounds__(256) void nms_mask_kernel(const float* __restrict__ boxes, int n, float thresh,
                                                      unsigned long long* __restrict__ mask) {
  const int cb = blockIdx.x, lane = threadIdx.x & 63;
  const int i = blockIdx.y * 4 + (threadIdx.x >> 6);
  const int col_blocks = (n + 63) / 64;
  if (i >= n || cb < (i >> 6)) return;               // wave-uniform
  const int j = cb * 64 + lane;
  bool hit = false;
  if (j < n && j > i) {
    float bi[5], bj[5];
#pragma unroll
    for (int t = 0; t < 5; ++t) {
      bi[t] = boxes[i * 5 + t];
      bj[t] = boxes[j * 5 + t];
    }
    hit = (ROTATED ? rbox_iou(bi, bj) : abox_iou(bi, bj)) > thresh;
  }
  const unsigned long long bits = __ballot(hit);
  if (lane == 0) mask[(long long)i * col_blocks + cb] = bits;
}

// greedy pass in box order: one wave; lane c owns the removed-bits of column blocks c, c + 64, ...
// The mask rows of 64 boxes at a time are staged in LDS with coalesced loads, so the serial walk pays one
// global round trip per 64 boxes instead of one per kept box (n = 1000: 0.86 ms -> see DESIGN.md).
__global__ __launch_bounds__(64) void nms_scan_kernel(const unsigned long long* __restrict__ mask, int n,
                                                      long long* __restrict__ keep, long long* __restrict__ num_keep) {
  extern __shared__ unsigned long long stage[];   // [64 rows][win] words of the current row block
  const int lane = threadIdx.x;
  const int col_blocks = (n + 63) / 64;
  constexpr int kMaxOwned = 64;                   // n <= 64 * 64 * kMaxOwned boxes
  unsigned long long removed[kMaxOwned];
  const int owned = (col_blocks + 63) / 64;
  for (int t = 0; t < owned; ++t) removed[t] = 0;
  long long kept = 0;
  for (int blk = 0; blk < col_blocks; ++blk) {
    const int rows = n - blk * 64 < 64 ? n - blk * 64 : 64;
    // which boxes of this row block survive depends only on the diagonal word column `blk` and on `removed`
    unsigned long long w = 0;
    for (int t = 0; t < owned; ++t)
      if (t == (blk >> 6)) w = removed[t];
    unsigned long long alive_mask = ~__shfl(w, blk & 63, 64);
    const unsigned long long diag = lane < rows ? mask[(long long)(blk * 64 + lane) * col_blocks + blk] : 0ull;
    unsigned long long kept_bits = 0;
    for (int j = 0; j < rows; ++j) {
      if ((alive_mask >> j) & 1ull) {
        kept_bits |= 1ull << j;
        alive_mask &= ~__shfl(diag, j, 64);
      }
    }
    // emit the kept boxes of this block in order
    if ((kept_bits >> lane) & 1ull) keep[kept + __popcll(kept_bits & ((1ull << lane) - 1ull))] = blk * 64 + lane;
    kept += __popcll(kept_bits);
    // OR the mask rows of the kept boxes into `removed` for the later column blocks, window by window
    for (int c0 = blk + 1; c0 < col_blocks; c0 += 256) {
      const int win = col_blocks - c0 < 256 ? col_blocks - c0 : 256;
      for (int e = lane; e < rows * win; e += 64) {
        const int rr = e / win, cc = e - rr * win;
        stage[rr * win + cc] = ((kept_bits >> rr) & 1ull) ? mask[(long long)(blk * 64 + rr) * col_blocks + c0 + cc] : 0ull;
      }
      __syncthreads();
      for (int t = 0; t < owned; ++t) {
        const int c = t * 64 + lane;
        if (c >= c0 && c < c0 + win) {
          unsigned long long acc = 0;
          for (int rr = 0; rr < rows; ++rr) acc |= stage[rr * win + (c - c0)];
          removed[t] |= acc;
        }
      }
      __syncthreads();
    }
  }
  if (lane == 0) *num_keep = kept;
}
}  // namespace

extern "C" int32_t vx_boxes_overlap_bev(const float* d_boxes_a, int32_t num_a, const float* d_boxes_b, int32_t num_b,
                                        float* d_out, int32_t iou, void* stream) {
  VX_REQUIRE(num_a >= 0 && num_b >= 0, "boxes_overlap_bev: negative box count");
  if (num_a == 0 || num_b == 0) return VX_OK;
  VX_REQUIRE(d_boxes_a && d_boxes_b && d_out, "boxes_overlap_bev: null pointer");
  const long long total = (long long)num_a * num_b;
  const unsigned grid = (unsigned)vx_cdiv(total, 256);
  if (iou)
    pairwise_kernel<true><<<grid, 256, 0, (hipStream_t)stream>>>(d_boxes_a, num_a, d_boxes_b, num_b, d_out);
  else
    pairwise_kernel<false><<<grid, 256, 0, (hipStream_t)stream>>>(d_boxes_a, num_a, d_boxes_b, num_b, d_out);
  VX_LAUNCH_CHECK();
  return VX_OK;
}

extern "C" size_t vx_nms_workspace_bytes(int32_t num_boxes) {
  const size_t n = num_boxes > 0 ? (size_t)num_boxes : 1;
  return vx_align(n * ((n + 63) / 64) * sizeof(unsigned long long)) + 256;
}

extern "C" int32_t vx_nms_bev(const float* d_boxes, int32_t num_boxes, float thresh, int32_t rotated,
                              int64_t* d_keep, int64_t* h_num_keep, void* d_ws, size_t ws_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  VX_REQUIRE(h_num_keep, "nms_bev: null count pointer");
  *h_num_keep = 0;
  VX_REQUIRE(num_boxes >= 0 && num_boxes <= 64 * 64 * 64, "nms_bev: need 0 <= num_boxes <= 262144");
  if (num_boxes == 0) return VX_OK;
  VX_REQUIRE(d_boxes && d_keep && d_ws, "nms_bev: null pointer");
  if (ws_bytes < vx_nms_workspace_bytes(num_boxes)) {
    vx_set_error("nms_bev: workspace too small");
    return VX_ERR_WORKSPACE;
  }
  const int col_blocks = (num_boxes + 63) / 64;
  unsigned long long* mask = (unsigned long long*)d_ws;
  long long* d_count = (long long*)((char*)d_ws + vx_align((size_t)num_boxes * col_blocks * sizeof(unsigned long long)));
  dim3 grid((unsigned)col_blocks, (unsigned)vx_cdiv(num_boxes, 4));
  if (rotated)
    nms_mask_kernel<true><<<grid, 256, 0, stream>>>(d_boxes, num_boxes, thresh, mask);
  else
    nms_mask_kernel<false><<<grid, 256, 0, stream>>>(d_boxes, num_boxes, thresh, mask);
  VX_LAUNCH_CHECK();
  {
    const int win = col_blocks < 256 ? col_blocks : 256;
    const size_t lds = (size_t)64 * win * sizeof(unsigned long long);
    VX_HIP(hipFuncSetAttribute((const void*)nms_scan_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    nms_scan_kernel<<<1, 64, lds, stream>>>(mask, num_boxes, (long long*)d_keep, d_count);
  }
  VX_LAUNCH_CHECK();
  long long h = 0;
  VX_HIP(hipMemcpyAsync(&h, d_count, sizeof(h), hipMemcpyDeviceToHost, stream));
  VX_HIP(hipStreamSynchronize(stream));
  *h_num_keep = h;
  return VX_OK;
}
