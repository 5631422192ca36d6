/* vx_abi.h -- C-ABI of libvx.so: the MI355X (gfx950) native kernels of the VISTA sparse voxel
 * backbone path (hard voxelization + mean VFE + SubMConv3d / SparseConv3d stack).
 *
 * Boundary rules:
 *  - flat extern "C" functions, plain pointers and sizes, no C++ / torch types;
 *  - every function returns int32 status: 0 = ok, < 0 = error (text: vx_last_error(), thread-local);
 *  - the CALLER owns all memory. Pointers named d_* / documented "device" are device pointers of
 *    the current HIP device; scratch comes in through an explicit workspace whose size the
 *    matching *_workspace_bytes() query returns. No function here allocates device memory;
 *  - every launch goes to the caller's stream (void* = hipStream_t, e.g.
 *    torch.cuda.current_stream().cuda_stream) and is asynchronous, except where a function is
 *    documented to return a host count (one small D2H copy + stream synchronize);
 *  - re-entrant and thread-safe. Process-wide state, all of it listed here: the thread-local error text; the DEFAULT
 *    fp32 matrix arithmetic (vx_spconv_set_f32_products: an atomic int that only calls WITHOUT a VX_CONV_F32_* flag
 *    read -- pass the flag and two threads / captured plans with different arithmetics cannot interfere); the
 *    per-device scratch array of vx_voxelize_hard_static (documented there: one call at a time per GPU);
 *    read-only per-device caches (CU count, address of a zero row).
 *
 * What each entry point replaces in the reference is cited as file:line under /root/reference.
 * The reference's FFI for the sparse part is the third-party `spconv` package (not vendored:
 * README.md:19,44-53); those entries cite the reference's call sites instead.
 */
#ifndef VX_ABI_H_
#define VX_ABI_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VX_OK 0
#define VX_ERR_ARG (-1)     /* bad argument (null pointer, shape/limit violated) */
#define VX_ERR_HIP (-2)     /* a HIP runtime call failed */
#define VX_ERR_WORKSPACE (-3) /* workspace too small */
#define VX_ERR_CAPACITY (-4)  /* an output capacity was exceeded */

#define VX_F32 0
#define VX_BF16 1

/* epilogue flags of vx_spconv_fwd */
#define VX_EPI_RELU 1
/* kernel choice of vx_spconv_fwd: fp32 layers with cin in {16,32,64} and cout <= 128 run the per-offset
   compaction kernel unless this flag asks for the row-stationary kernels (A/B timing, tests) */
#define VX_CONV_NO_COMPACT 0x100
/* the row-stationary kernel splits its last partial round of row tiles over output-column halves, and the
   16-pair compaction kernel (fp32 cin = 64) runs its last partial round as a second launch on half-height tiles
   (same sums, bitwise identical output); this flag launches the plain grids instead (A/B timing, tests) */
#define VX_CONV_NO_SPLIT_TAIL 0x200
/* matrix arithmetic of THIS call for fp32 operands (vx_spconv_fwd / _dn / dgrad / wgrad `flags`; see
   vx_spconv_set_f32_products below for what the three mean). 0 in these bits = the process default. */
#define VX_CONV_ARITH_MASK 0x3000
#define VX_CONV_F32_MFMA 0x1000       /* v_mfma_f32_32x32x2_f32 */
#define VX_CONV_F32_PRODUCTS6 0x2000  /* six of the nine exact bf16 plane products */
#define VX_CONV_F32_PRODUCTS9 0x3000  /* all nine */

int32_t vx_version(void);
const char* vx_last_error(void);

/* ---------------------------------------------------------------------------------------------
 * Hard voxelization + mean VFE.
 * Replaces: det3d/ops/point_cloud/point_cloud_ops.py:138-221 (points_to_voxel_new) /:224-296,
 *           det3d/core/input/voxel_generator.py:21-39 (VoxelGenerator.generate),
 *           det3d/models/readers/voxel_encoder.py:206-211 (mean VFE, fused),
 *           det3d/torchie/parallel/collate.py:135-142 (batch-index column of coordinates).
 *
 * d_points     device f32[n_points][n_feat] (xyz first)
 * voxel_size   HOST f32[3] (x,y,z), range HOST f32[6] (xmin,ymin,zmin,xmax,ymax,zmax), already fp32
 * batch_idx    value written to column 0 of d_coords
 * d_coords     device int32[cap][4]  (batch, z, y, x), voxel id = order of first point
 * d_num_points device int32[cap]     min(count, max_points)
 * d_voxels     device f32[cap][max_points][n_feat] or NULL (rows [0,M) fully written, zero padded)
 * d_mean       device f32[cap][n_feat] or NULL (sum of the kept points / num_points)
 * cap          row capacity of the outputs; M <= min(n_points, max_voxels) always holds
 * d_num_voxels device int64[1]: M
 * h_num_voxels HOST int64[1] or NULL; when non-NULL the call synchronises the stream and stores M
 * ------------------------------------------------------------------------------------------- */
size_t vx_voxelize_workspace_bytes(int64_t n_points);
int32_t vx_voxelize_hard(const float* d_points, int64_t n_points, int32_t n_feat,
                         const float* voxel_size, const float* range, int32_t max_points,
                         int64_t max_voxels, int32_t batch_idx, int32_t* d_coords,
                         int32_t* d_num_points, float* d_voxels, float* d_mean, int64_t cap,
                         int64_t* d_num_voxels, int64_t* h_num_voxels, void* d_workspace,
                         size_t workspace_bytes, void* stream);

/* Batched form: n_frames clouds concatenated frame-major in d_points, frame f = points
 * [frame_offsets[f], frame_offsets[f+1]) (HOST int64[n_frames+1], frame_offsets[0] == 0, n_frames <= 64).
 * One launch sequence for the whole batch (replaces the per-frame loop + collate_kitti,
 * det3d/torchie/parallel/collate.py:105-106,135-142): rows come out frame after frame, each frame in
 * its own first-appearance order, column 0 of d_coords = f. Every frame must have <= max_voxels
 * points (then the reference's max_voxels truncation cannot trigger); otherwise call per frame.
 * d_num_voxels / h_num_voxels: int64[n_frames + 1] = {total, M_0, ..., M_{n_frames-1}}. */
int32_t vx_voxelize_hard_batch(const float* d_points, int32_t n_frames, const int64_t* frame_offsets,
                               int32_t n_feat, const float* voxel_size, const float* range,
                               int32_t max_points, int64_t max_voxels, int32_t* d_coords,
                               int32_t* d_num_points, float* d_voxels, float* d_mean, int64_t cap,
                               int64_t* d_num_voxels, int64_t* h_num_voxels, void* d_workspace,
                               size_t workspace_bytes, void* stream);

/* Allocation-free form of vx_voxelize_hard. It has NO counterpart in the reference: it exists for callers
 * that cannot allocate device memory or carry a workspace -- code under HIP graph capture, tight serving loops,
 * harnesses that forbid hipMalloc. Same arguments and results as vx_voxelize_hard minus the workspace pair: the
 * scratch (hash table, per-point and per-voxel index arrays, scan partials) is a 16 MiB file-scope __device__
 * array of the library, one copy per GPU held until the process ends, whose device address is looked up once
 * per device (vx_voxelize_static_init, or the first call on that device).
 * Limit: n_points <= vx_voxelize_static_max_points() (262144); larger clouds return VX_ERR_CAPACITY.
 * The call allocates nothing (no hipMalloc / hipFree / allocating library routine); pass h_num_voxels = NULL to
 * keep it free of host copies and synchronisation as well. Every call on one GPU shares that GPU's array:
 * calls that overlap on one GPU (two streams, two threads) race -- serialise them. */
int64_t vx_voxelize_static_max_points(void);
int32_t vx_voxelize_static_init(void);
int32_t vx_voxelize_hard_static(const float* d_points, int64_t n_points, int32_t n_feat,
                                const float* voxel_size, const float* range, int32_t max_points,
                                int64_t max_voxels, int32_t batch_idx, int32_t* d_coords,
                                int32_t* d_num_points, float* d_voxels, float* d_mean, int64_t cap,
                                int64_t* d_num_voxels, int64_t* h_num_voxels, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Rulebooks (indice pairs). Stored output-major: nbr[k][o] = input row feeding output row o
 * through kernel offset k = (kz*KH + ky)*KW + kx, or -1. Row stride of nbr is nbr_ld (>= n_out).
 * Replaces spconv's indice-pair generation behind
 *   det3d/models/backbones/scn.py:28-38,333 (SubMConv3d, cached under indice_key) and
 *   det3d/models/backbones/scn.py:338-361, multiview.py:39-82 (SparseConv3d).
 * d_coords device int32[n][4] (batch,z,y,x); shape HOST int32[3] (D,H,W); ksize/stride/pad HOST int32[3].
 * ------------------------------------------------------------------------------------------- */
size_t vx_rulebook_workspace_bytes(int64_t n_in, int32_t batch, const int32_t* out_shape);

/* submanifold: outputs == inputs (same rows); padding is ksize/2 */
int32_t vx_rulebook_subm(const int32_t* d_coords, int64_t n, int32_t batch, const int32_t* shape,
                         const int32_t* ksize, int32_t* d_nbr, int64_t nbr_ld, void* d_workspace,
                         size_t workspace_bytes, void* stream);

/* strided conv, phase 1: find the active output set. Leaves its state in the workspace, returns
 * the number of output sites in *h_n_out (synchronises the stream). out_shape is written (HOST). */
int32_t vx_rulebook_conv_count(const int32_t* d_coords, int64_t n_in, int32_t batch,
                               const int32_t* in_shape, const int32_t* ksize, const int32_t* stride,
                               const int32_t* pad, int32_t* out_shape, int64_t* h_n_out,
                               void* d_workspace, size_t workspace_bytes, void* stream);

/* strided conv, phase 2 (same workspace, untouched since phase 1): output coordinates sorted by
 * linear index ((b*D+z)*H+y)*W+x, the neighbour table, and optionally its transpose
 * d_nbr_t[k][i] = output row fed by input row i through offset k, or -1 (row stride nbr_t_ld).
 * d_rank (optional, vx_rank_table_bytes(batch, out_shape) bytes): receives the rank table of the
 * OUTPUT set ({occupancy bits, row base} per 32 cells) for vx_rulebook_subm_ranked. */
int32_t vx_rulebook_conv_emit(const int32_t* d_coords, int64_t n_in, int32_t batch,
                              const int32_t* in_shape, const int32_t* ksize, const int32_t* stride,
                              const int32_t* pad, int64_t n_out, int32_t* d_out_coords,
                              int32_t* d_nbr, int64_t nbr_ld, int32_t* d_nbr_t, int64_t nbr_t_ld,
                              void* d_rank, void* d_workspace, size_t workspace_bytes, void* stream);

/* submanifold rulebook over a site set that carries a rank table (the outputs of a strided conv):
 * coordinate -> row is one 8-byte read + popcount instead of a hash probe; no workspace. */
size_t vx_rank_table_bytes(int32_t batch, const int32_t* shape);
int32_t vx_rulebook_subm_ranked(const int32_t* d_coords, int64_t n, int32_t batch,
                                const int32_t* shape, const int32_t* ksize, const void* d_rank,
                                int32_t* d_nbr, int64_t nbr_ld, void* stream);

/* Mask-sorted row order for the conv kernel (no counterpart in the reference: an MI355X scheduling
 * aid). Inside windows of 16384 rows, rows are sorted by their "offsets used" bit mask so the 32-row
 * MFMA groups of vx_spconv_fwd skip more (group, offset) slots. d_nbr_sorted[k][j] = d_nbr[k][d_perm[j]]. */
int32_t vx_rulebook_sort_rows(const int32_t* d_nbr, int64_t nbr_ld, int32_t kvol, int64_t n_out,
                              int32_t* d_nbr_sorted, int64_t sorted_ld, int32_t* d_perm, void* stream);

/* pair-list view (spconv's indice_pairs): d_pairs int32[K][2][n_out] ((in,out) per offset, ordered
 * by output row, -1 padded), d_pair_num int32[K]. Wavefront-ballot compaction. */
size_t vx_pairs_workspace_bytes(int32_t kvol, int64_t n_out);
int32_t vx_nbr_to_pairs(const int32_t* d_nbr, int64_t nbr_ld, int32_t kvol, int64_t n_out,
                        int32_t* d_pairs, int32_t* d_pair_num, void* d_workspace,
                        size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Sparse convolution, forward (implicit GEMM, output-stationary, MFMA):
 *   out[o] = epi( sum_k W[:,k,:] . in[nbr[k][o]] + bias )
 *   epi(x) = relu?( x * scale + shift + residual[o] )     (scale/shift/residual optional)
 * Replaces spconv's conv forward behind SubMConv3d / SparseConv3d (call sites above) and, through
 * the optional epilogue, the BatchNorm1d(eval) + ReLU + residual add of scn.py:82-98.
 * d_in   device [n_in][cin], d_out device [n_out][cout], d_residual device [n_out][cout] or NULL
 * d_weight device [cout][kvol][cin] (KRSC, the nn.Parameter layout), dtype = VX_F32 | VX_BF16 for
 * in/out/weight/residual; bias/scale/shift are always f32[cout] (or NULL).
 * d_perm (optional): d_nbr is a mask-sorted table from vx_rulebook_sort_rows and table row j
 * computes output row d_perm[j]; results are bitwise identical to the unsorted call.
 * d_wpack (optional): see vx_spconv_pack_weight below.
 * ------------------------------------------------------------------------------------------- */
int32_t vx_spconv_fwd(const void* d_in, int64_t n_in, int32_t cin, const void* d_weight,
                      int32_t kvol, int32_t cout, const int32_t* d_nbr, int64_t nbr_ld,
                      int64_t n_out, const float* d_bias, const float* d_scale,
                      const float* d_shift, const void* d_residual, int32_t flags,
                      const int32_t* d_perm, const void* d_wpack, void* d_out, int32_t dtype,
                      void* stream);

/* Optional fp32 fast path: a copy of the weight packed in MFMA-fragment order. When d_wpack is
 * given to vx_spconv_fwd (fp32, cin in {8,16,32,64,128}, cout <= 128) the barrier-free streaming-W
 * kernel runs: same results to fp32 rounding of the same sums (identical summation order). */
size_t vx_spconv_pack_weight_bytes(int32_t cin, int32_t cout, int32_t kvol);
int32_t vx_spconv_pack_weight(const void* d_weight, int32_t cin, int32_t cout, int32_t kvol,
                              void* d_wpack, void* stream);

/* dgrad: din[i] = sum_k W[:,k,:]^T . dout[nbr_t[k][i]]  (input-stationary, atomic-free, same MFMA
 * kernels as forward). d_weight_t is the TRANSPOSED weight copy [cin][kvol][cout]
 * (weight.permute(4,1,2,3,0)). For a submanifold conv pass the forward table with subm=1
 * (nbr_t[k] == nbr[kvol-1-k]); for a strided conv pass the transpose table of
 * vx_rulebook_conv_emit. Replaces spconv's input gradient reached through autograd at
 * det3d/core/utils/dist_utils.py:50-56 (loss.backward()). flags: VX_CONV_F32_* or 0. */
int32_t vx_spconv_dgrad(const void* d_dout, int64_t n_out, int32_t cout, const void* d_weight_t,
                        int32_t kvol, int32_t cin, const int32_t* d_nbr_t, int64_t nbr_t_ld,
                        int64_t n_in, int32_t subm, int32_t flags, void* d_din, int32_t dtype, void* stream);

/* wgrad: dW[co][k][ci] = sum over the pairs (i,o) of offset k of dout[o][co] * in[i][ci], from the
 * compacted pair lists of vx_nbr_to_pairs (exact FLOPs); two-pass slab reduction, atomic-free and
 * bitwise reproducible. d_dw f32[cout][kvol][cin] is OVERWRITTEN (it may be any device address, e.g. the
 * parameter's slice of a flat gradient buffer). flags: VX_CONV_F32_* or 0. Replaces spconv's weight gradient
 * reached through autograd at det3d/core/utils/dist_utils.py:50-56 (loss.backward()). */
size_t vx_spconv_wgrad_workspace_bytes(int32_t cin, int32_t cout, int32_t kvol, int64_t n_out);
int32_t vx_spconv_wgrad(const void* d_in, int64_t n_in, int32_t cin, const void* d_dout,
                        int64_t n_out, int32_t cout, const int32_t* d_pairs,
                        const int32_t* d_pair_num, int32_t kvol, float* d_dw, int32_t dtype, int32_t flags,
                        void* d_workspace, size_t workspace_bytes, void* stream);

/* DEFAULT matrix arithmetic of the fp32 convolutions with Cin >= 64 (forward, dgrad) and of the weight gradients
 * with Cin, Cout multiples of 32, used by calls whose flags carry no VX_CONV_F32_* bits:
 *   0  v_mfma_f32_32x32x2_f32 (157 TFLOP/s dense on MI355X)
 *   9  every fp32 operand split exactly into three bf16 planes (x = h + m + l), all nine plane products on
 *      v_mfma_f32_32x32x16_bf16 with fp32 accumulation: each fp32 x fp32 product enters the sum exactly; 9/16 of
 *      the fp32 MFMA's cycles
 *   6  the same without the three smallest plane products (each below 2^-23 of |a*b|); 6/16 of the cycles
 * Inputs, weights, outputs and the accumulators stay fp32 in all three. Initial value: VX_F32_PRODUCTS in the
 * environment at load time, else 6. No counterpart in the reference (spconv picks its own GEMM kernels); the
 * tolerance of the path (1e-4 relative in fp32, BASELINE.json north_star) is tested in all three modes. */
int32_t vx_spconv_set_f32_products(int32_t products);
int32_t vx_spconv_get_f32_products(void);

/* ---------------------------------------------------------------------------------------------
 * SparseConvTensor.dense(): zero-filled [batch][c][D][H][W] with out[b][:,z,y,x] = feat[row].
 * Replaces spconv's dense() at det3d/models/backbones/scn.py:375, multiview.py:97,102.
 * ------------------------------------------------------------------------------------------- */
int32_t vx_to_dense(const void* d_feat, const int32_t* d_coords, int64_t n, int32_t c,
                    int32_t batch, const int32_t* shape, void* d_dense, int32_t dtype,
                    void* stream);
/* The gradient of dense(): d_feat[row][:] = d_dense[b][:,z,y,x] at row's coordinates (rows of out-of-grid coordinates
 * are the caller's error, as for vx_to_dense). Replaces the scatter's autograd gather reached through
 * loss.backward() (det3d/core/utils/dist_utils.py:50-56) behind scn.py:375. */
int32_t vx_from_dense(const void* d_dense, const int32_t* d_coords, int64_t n, int32_t c, int32_t batch,
                      const int32_t* shape, void* d_feat, int32_t dtype, void* stream);

/* ---------------------------------------------------------------------------------------------
 * "_dn" forms: the same operations with their data-dependent ROW COUNTS RESIDENT ON THE DEVICE, so that the whole
 * per-frame pass (voxelization, every rulebook, every convolution, dense()) can be queued -- or captured in a HIP
 * graph -- without one host synchronisation. They replace the per-layer host dispatch of
 * det3d/models/backbones/scn.py:332-381 / multiview.py:88-106 (one Python call + spconv's own host-side pair
 * counting per layer) the way SURVEY.md 8f row N2 asks. Each takes
 *   d_n  / d_n_in / d_n_out   device int64: the row count, written by the producing call on the same stream
 *   cap / cap_in / cap_out    host upper bound: grids, table strides and buffer sizes are sized for it
 * Kernels use min(count, capacity) rows and workgroups beyond the count exit at once; tile mapping, tail splitting
 * and every row's summation order are the same functions of the count as in the host-count forms, so results are
 * bitwise identical to them. A count above its capacity drops the surplus rows (never an out-of-bounds access):
 * the caller reads the counts back after the pass, compares them with the capacities and re-runs with larger ones.
 *
 * vx_voxelize_hard_batch_dn   d_frame_offsets = DEVICE int64[n_frames + 1] (point ranges, [0] == 0); the point
 *                             count is d_frame_offsets[n_frames] <= cap_points; d_num_voxels int64[n_frames + 1] =
 *                             {total (NOT clamped to cap), per frame}; no max_voxels truncation (caller guarantees
 *                             every frame has <= max_voxels points, as for vx_voxelize_hard_batch). Workspace:
 *                             vx_voxelize_workspace_bytes(cap_points).
 * vx_rulebook_subm_dn         workspace vx_rulebook_workspace_bytes(cap, batch, NULL)
 * vx_rulebook_conv_dn         count + emit in one call: d_n_out receives the number of output sites (not clamped),
 *                             d_out_coords / d_nbr (stride nbr_ld >= cap_out) / d_rank as vx_rulebook_conv_emit;
 *                             *d_err (device int64, zeroed once by the caller) is set to 1 when a coordinate lies
 *                             outside batch x in_shape. Workspace vx_rulebook_workspace_bytes(cap_in, batch, out_shape).
 * vx_rulebook_subm_ranked_dn, vx_spconv_fwd_dn, vx_to_dense_dn   as their host-count forms.
 * ------------------------------------------------------------------------------------------- */
/* Stream-ordered byte fill by a kernel (d_ptr[0..bytes) = value & 0xff): what hipMemsetAsync does, as a KERNEL node.
 * hipMemsetAsync nodes captured in a HIP graph replay out of order on ROCm 7.2 (DESIGN.md "Sync-free pass"), so a
 * caller that captures the "_dn" calls clears its count / error words with this. No counterpart in the reference. */
int32_t vx_fill(void* d_ptr, int32_t value, size_t bytes, void* stream);

int32_t vx_voxelize_hard_batch_dn(const float* d_points, int32_t n_frames, const int64_t* d_frame_offsets,
                                  int64_t cap_points, int32_t n_feat, const float* voxel_size,
                                  const float* range, int32_t max_points, int32_t* d_coords,
                                  int32_t* d_num_points, float* d_mean, int64_t cap, int64_t* d_num_voxels,
                                  void* d_workspace, size_t workspace_bytes, void* stream);
int32_t vx_rulebook_subm_dn(const int32_t* d_coords, const int64_t* d_n, int64_t cap, int32_t batch,
                            const int32_t* shape, const int32_t* ksize, int32_t* d_nbr, int64_t nbr_ld,
                            void* d_workspace, size_t workspace_bytes, void* stream);
int32_t vx_rulebook_conv_dn(const int32_t* d_coords, const int64_t* d_n_in, int64_t cap_in, int32_t batch,
                            const int32_t* in_shape, const int32_t* ksize, const int32_t* stride,
                            const int32_t* pad, int32_t* out_shape, int64_t cap_out, int32_t* d_out_coords,
                            int64_t* d_n_out, int32_t* d_nbr, int64_t nbr_ld, void* d_rank, int64_t* d_err,
                            void* d_workspace, size_t workspace_bytes, void* stream);
int32_t vx_rulebook_subm_ranked_dn(const int32_t* d_coords, const int64_t* d_n, int64_t cap, int32_t batch,
                                   const int32_t* shape, const int32_t* ksize, const void* d_rank,
                                   int32_t* d_nbr, int64_t nbr_ld, void* stream);
int32_t vx_spconv_fwd_dn(const void* d_in, int64_t cap_in, int32_t cin, const void* d_weight, int32_t kvol,
                         int32_t cout, const int32_t* d_nbr, int64_t nbr_ld, const int64_t* d_n_out,
                         int64_t cap_out, const float* d_bias, const float* d_scale, const float* d_shift,
                         const void* d_residual, int32_t flags, void* d_out, int32_t dtype, void* stream);
int32_t vx_to_dense_dn(const void* d_feat, const int32_t* d_coords, const int64_t* d_n, int64_t cap, int32_t c,
                       int32_t batch, const int32_t* shape, void* d_dense, int32_t dtype, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Train-mode BatchNorm1d over sparse features [n][c] (fp32, c a power of two in [8, 512]), fused with the
 * residual add and ReLU that follow it. Replaces the torch BatchNorm1d / add / ReLU passes over `.features`
 * of det3d/models/backbones/scn.py:82-98 (block) and :333-364 (conv + BN + ReLU) in the training step
 * (norm layer: det3d/models/utils/norm.py:62,99-103; SyncBatchNorm: det3d/torchie/apis/train.py:301 -- the
 * caller all-reduces d_sums between the two halves of each pass, which is all SyncBN exchanges).
 *
 * forward : vx_bn_stats    d_sums f64[2c+1] = {sum x, sum x^2 per channel, n}   (ordered two-level reduction)
 *           [all-reduce d_sums over ranks]
 *           vx_bn_finalize d_mean, d_invstd f32[c] (biased variance + eps); running stats updated with
 *                          `momentum` and the unbiased variance when the pointers are non-NULL
 *           vx_bn_apply    y = relu?((x - mean) * invstd * gamma + beta + residual?); with relu, d_mask
 *                          (n*c/8 bytes) receives one bit per element: y > 0
 * backward: vx_bn_bwd_reduce d_sums f64[2c] = {sum g, sum g*xhat}, g = dy where the mask bit is set (all of
 *                          dy when d_mask is NULL); d_param_grads f32[2c] (optional) = the same sums in fp32
 *                          = {d beta, d gamma} of THIS rank
 *           [all-reduce d_sums over ranks]
 *           vx_bn_bwd_apply dx = gamma*invstd*(g - sum_g/N - xhat*sum_gxhat/N), N = *d_count (f64, the total
 *                          row count, e.g. d_sums[2c] of the forward pass); d_dres (optional) receives g
 * d_gamma / d_beta / d_residual / d_mask may be NULL. Workspace: vx_bn_workspace_bytes(c).
 * ------------------------------------------------------------------------------------------- */
size_t vx_bn_workspace_bytes(int32_t c);
int32_t vx_bn_stats(const float* d_x, int64_t n, int32_t c, double* d_sums, void* d_workspace,
                    size_t workspace_bytes, void* stream);
int32_t vx_bn_finalize(const double* d_sums, int32_t c, float eps, float momentum, float* d_running_mean,
                       float* d_running_var, float* d_mean, float* d_invstd, void* stream);
int32_t vx_bn_apply(const float* d_x, int64_t n, int32_t c, const float* d_mean, const float* d_invstd,
                    const float* d_gamma, const float* d_beta, const float* d_residual, int32_t relu,
                    float* d_y, uint8_t* d_mask, void* stream);
int32_t vx_bn_bwd_reduce(const float* d_dy, const float* d_x, const uint8_t* d_mask, int64_t n, int32_t c,
                         const float* d_mean, const float* d_invstd, double* d_sums, float* d_param_grads,
                         void* d_workspace, size_t workspace_bytes, void* stream);
int32_t vx_bn_bwd_apply(const float* d_dy, const float* d_x, const uint8_t* d_mask, int64_t n, int32_t c,
                        const float* d_mean, const float* d_invstd, const float* d_gamma,
                        const double* d_sums, const double* d_count, float* d_dx, float* d_dres,
                        void* stream);

/* ---------------------------------------------------------------------------------------------
 * Rotated BEV boxes [x1, y1, x2, y2, ry] (post-processing, outside the sparse hot path).
 * vx_boxes_overlap_bev: d_out f32[num_a][num_b] = overlap area (iou = 0) or IoU (iou = 1). Replaces
 *   boxes_overlap_bev_gpu / boxes_iou_bev_gpu (det3d/ops/iou3d_nms/src/iou3d_nms.cpp:37-77).
 * vx_nms_bev: greedy NMS over boxes ALREADY sorted by descending score; rotated = 1 uses the rotated IoU
 *   (nms_gpu, iou3d_nms.cpp:79-126), 0 the axis-aligned IoU of the first four fields (nms_normal_gpu, :129-176).
 *   d_keep int64[num_boxes] receives the kept indices in order, *h_num_keep their count (one host sync).
 * ------------------------------------------------------------------------------------------- */
int32_t vx_boxes_overlap_bev(const float* d_boxes_a, int32_t num_a, const float* d_boxes_b, int32_t num_b,
                             float* d_out, int32_t iou, void* stream);
size_t vx_nms_workspace_bytes(int32_t num_boxes);
int32_t vx_nms_bev(const float* d_boxes, int32_t num_boxes, float thresh, int32_t rotated, int64_t* d_keep,
                   int64_t* h_num_keep, void* d_workspace, size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VX_ABI_H_ */
