/*
 * libu3d_hip.so — C ABI of the MI355X-native Uni3DETR detection hot path.
 *
 * Every entry point: plain device pointers + explicit sizes + a hipStream_t; returns 0 or a negative
 * U3D_ERR_* code (u3d_strerror); never throws, never allocates device memory, never synchronises the
 * stream, keeps no global mutable state.  The caller (the ctypes host layer in uni3detr_amd/native.py,
 * or any other FFI) owns every buffer.  Citations `ref:` are paths under the reference repository
 * (zhenyuw16/Uni3DETR) naming the Python call site / upstream op each function replaces.
 *
 * Target: gfx950 only.
 */
#ifndef U3D_HIP_H_
#define U3D_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* u3d_stream;   /* == hipStream_t */

enum {
  U3D_OK = 0,
  U3D_ERR_ARG = -1,        /* bad argument (null pointer, size out of range)        */
  U3D_ERR_UNSUPPORTED = -2,/* shape / dtype combination not compiled in              */
  U3D_ERR_LAUNCH = -3,     /* hipGetLastError() != hipSuccess after a launch         */
  U3D_ERR_WORKSPACE = -4   /* workspace too small                                    */
};

enum { U3D_F32 = 0, U3D_BF16 = 1 };

int32_t u3d_version(void);
const char* u3d_strerror(int32_t code);

/* Measurement helpers (bench.py's roofline block): HIP events on the stream the kernels run on.  external != 0 records with
 * hipEventRecordExternal: inside a stream capture the record becomes an event-record NODE of the graph, so a pair of them brackets
 * one kernel as it runs inside every replay (elapsed time read after the replay).  Not part of the reference's interface. */
int32_t u3d_event_create(void** event);
int32_t u3d_event_record(void* event, int32_t external, u3d_stream s);
int32_t u3d_event_elapsed_ms(void* start, void* stop, float* ms);     /* both events must have completed */
int32_t u3d_event_destroy(void* event);

/* ------------------------------------------------------------------------------------------------
 * Occupancy lattice ("BitGrid").  One 64-bit word per 4x4x4 block of cells, bit = (z&3)*16+(y&3)*4+(x&3),
 * word = ((b*bz + z/4)*by + y/4)*bx + x/4 with bz=ceil(dz/4) etc.  `prefix` = exclusive popcount scan
 * (nwords+1 entries).  The rank of a set bit is the row index of that voxel at this sparse level.
 * Replaces the hash-table indice-pair builder of spconv (ref: models/pts_encoder/sparse_encoder_hd.py:9-12,
 * upstream ext.get_indice_pairs_*).
 * ---------------------------------------------------------------------------------------------- */
typedef struct u3d_bitgrid {
  void* words;      /* uint64 [nwords]   */
  void* prefix;     /* uint32 [nwords+1] */
  int32_t batch, dz, dy, dx;
  int32_t layout;   /* 0: 4x4x4-block words (rank = block-major order, used by the conv levels);
                       1: linear, one bit per cell in (b,z,y,x) order (rank = lexicographic = torch.unique(dim=0) order) */
  int32_t row_capacity; /* > 0: static-shape mode — rows are stored in buffers of this many rows; lookups (rank, neighbour
                           tables) return -1 for any row id >= row_capacity, so an overflowing level degrades to dropped
                           voxels instead of out-of-bounds gathers; 0: unlimited */
} u3d_bitgrid;

int64_t u3d_bitgrid_nwords(int32_t batch, int32_t dz, int32_t dy, int32_t dx);
int64_t u3d_bitgrid_nwords_layout(int32_t batch, int32_t dz, int32_t dy, int32_t dx, int32_t layout);

/* words must be zeroed by the caller (hipMemsetAsync) before marking. coors: int32 [n,4] (b,z,y,x);
 * rows with b < 0 are ignored. */
int32_t u3d_bitgrid_mark(const u3d_bitgrid* g, const int32_t* coors, int32_t n, u3d_stream s);

/* Mark every output site of a strided sparse conv whose window touches an active input:
 * o = (i + pad - kappa) / stride when divisible and inside `g` (ref: SURVEY.md App. A4 active-set rule;
 * sparse_encoder_hd.py:181-192).  n_dev: device int32 count of valid rows in `in_coors` (<= n_cap). */
int32_t u3d_bitgrid_mark_strided(const u3d_bitgrid* g, const int32_t* in_coors, const int32_t* n_dev,
                                 int32_t n_cap, const int32_t ksize[3], const int32_t stride[3],
                                 const int32_t pad[3], u3d_stream s);

/* prefix[] from words[]; scratch: uint32 [u3d_bitgrid_scan_scratch(nwords)] ; total count lands in
 * prefix[nwords] (device). */
int64_t u3d_bitgrid_scan_scratch(int64_t nwords);
int32_t u3d_bitgrid_scan(const u3d_bitgrid* g, void* scratch, u3d_stream s);

/* rank[i] = row index of coors[i] in g (or -1). */
int32_t u3d_bitgrid_rank(const u3d_bitgrid* g, const int32_t* coors, int32_t n, int32_t* rank, u3d_stream s);

/* Enumerate occupied cells in rank order: coors_out int32 [cap,4] (b,z,y,x); cap = rows available; rows [count, cap) = (-1,-1,-1,-1). */
int32_t u3d_bitgrid_coords(const u3d_bitgrid* g, int32_t* coors_out, int32_t cap, u3d_stream s);

/* Neighbour table ("rulebook") nbr[kappa][ld] int32, kappa = (kz*k1 + ky)*k2 + kx, -1 = no partner.
 *   mode 0 (gather/forward): partner of query q at kappa = target cell  q*stride - pad + kappa
 *       (SubMConv3d: stride 1, pad (k-1)/2, target == query grid; SparseConv3d forward: query = output sites).
 *   mode 1 (transposed/dgrad): partner = (q + pad - kappa)/stride when divisible (query = input sites,
 *       target = output grid).
 * n_dev: device count of valid query rows; rows >= *n_dev get -1.  ld >= n_cap.
 * Replaces upstream indice_pairs (ref: SURVEY.md §8a a-4; sparse_encoder_hd.py:195-199 builds them per conv). */
int32_t u3d_nbr_table(const u3d_bitgrid* target, const int32_t* q_coors, const int32_t* n_dev, int32_t n_cap,
                      const int32_t ksize[3], const int32_t stride[3], const int32_t pad[3], int32_t mode,
                      int32_t* nbr, int32_t ld, u3d_stream s);

/* Same table for a DENSE lattice (every cell of [batch, dims] is a row; row id = lexicographic (b,z,y,x) index, i.e. the
 * memory order of a channels-last volume): lets the dense SECOND3D / SECOND3DFPN convolutions run on the same
 * implicit-GEMM kernels (ref: models/backbones/second_3d.py:52-76, models/necks/second3d_fpn.py:48-104). */
int32_t u3d_dense_nbr_table(int32_t batch, const int32_t q_dims[3], const int32_t t_dims[3], const int32_t ksize[3],
                            const int32_t stride[3], const int32_t pad[3], int32_t mode, int32_t* nbr, int32_t ld,
                            u3d_stream s);

/* ------------------------------------------------------------------------------------------------
 * Hard voxelization + mean VFE (ref: models/detectors/uni3detr.py:148-149; upstream mmcv Voxelization
 * hard mode + HardSimpleVFE).  Sequential semantics reproduced exactly: voxels in
 * first-appearance point order, first `max_points` points per voxel in point order, at most `max_voxels`
 * voxels per scene.
 *   points      f32 [n_total, nfeat], scenes concatenated; scene_off int32 [B+1] (device)
 *   voxel_size/pc_range: host arrays (x,y,z) / (x0,y0,z0,x1,y1,z1)
 *   outputs (capacity B*max_voxels rows, scene-major, compacted):
 *     voxels  f32 [cap, max_points, nfeat] or NULL, coors int32 [cap,4] (b,z,y,x), num_points int32 [cap],
 *     mean    f32 [cap, nfeat] or NULL (sum of kept points / count),
 *     voxel_off int32 [B+1] device (row offsets per scene; voxel_off[B] = total)
 *   workspace: u3d_voxelize_hard_workspace(n_total, B) bytes, caller-zeroing NOT required.
 * ---------------------------------------------------------------------------------------------- */
int64_t u3d_voxelize_hard_workspace(int32_t n_total, int32_t batch, int32_t max_pts_per_scene);
int32_t u3d_voxelize_hard(const float* points, const int32_t* scene_off, int32_t batch, int32_t n_total,
                          int32_t max_pts_per_scene, int32_t nfeat, const float voxel_size[3],
                          const float pc_range[6], int32_t max_points, int32_t max_voxels,
                          float* voxels, int32_t* coors, int32_t* num_points, float* mean,
                          int32_t* voxel_off, void* workspace, int64_t workspace_bytes, u3d_stream s);

/* ------------------------------------------------------------------------------------------------
 * Dynamic voxelization + DynamicSimpleVFE (ref: models/detectors/uni3detr.py:155-171; upstream mmcv Voxelization with
 * max_num_points=-1 and DynamicScatter(average_points=True)).
 *   u3d_voxelize_dynamic: per point (b, z, y, x) int32, or (b,-1,-1,-1) when out of range (same floor formula as hard mode).
 *   u3d_scatter_mean: feats[rank[i], :] = mean of points with that rank (rank < 0 skipped); sums f32 [V,nfeat] and
 *   counts int32 [V] must be zeroed by the caller; f32 atomics (as the upstream kernel), then an in-place divide.
 * ---------------------------------------------------------------------------------------------- */
int32_t u3d_voxelize_dynamic(const float* points, const int32_t* scene_off, int32_t batch, int32_t n_total, int32_t nfeat,
                             const float voxel_size[3], const float pc_range[6], int32_t* coors, u3d_stream s);
int32_t u3d_scatter_mean(const float* points, const int32_t* rank, int32_t n_total, int32_t nfeat, float* sums,
                         int32_t* counts, int32_t n_voxels, u3d_stream s);

/* ------------------------------------------------------------------------------------------------
 * Sparse convolution as output-stationary implicit GEMM over the neighbour table
 *   out[m,:] = sum_kappa  in[nbr[kappa][m], :] @ W[kappa]        (W: [K, Cin, Cout], K = 27 or 1)
 * Covers SubMConv3d, SparseConv3d forward (nbr mode 0) and both dgrads (nbr mode 1 + transpose_w=1,
 * which reads W[kappa] as [Cout_of_fwd x Cin_of_fwd]^T).  nbr == NULL means identity (1x1x1 conv).
 * (ref: sparse_encoder_hd.py:71-104,181-199; upstream ext.indice_conv_forward/backward.)
 * dtype U3D_F32: exact-f32 MFMA (v_mfma_f32_16x16x4_f32); U3D_BF16: bf16 operands, f32 accumulate.
 * n_out_dev: device count of valid output rows (<= n_out_cap).
 * ---------------------------------------------------------------------------------------------- */
int32_t u3d_spconv_fwd(const void* in, const void* w, const int32_t* nbr, int32_t ld, void* out,
                       const int32_t* n_out_dev, int32_t n_out_cap, int32_t cin, int32_t cout, int32_t kvol,
                       int32_t transpose_w, int32_t dtype, u3d_stream s);

/* Second-generation bf16 kernels (256-row tiles, double-buffered LDS, LDS transpose reads): same contract as
 * u3d_spconv_fwd / u3d_spconv_wgrad with dtype U3D_BF16; return U3D_ERR_UNSUPPORTED for shapes (cin % 64, cout % 64)
 * that the first-generation kernels serve. */
int32_t u3d_igemm_fwd_bf16(const void* in, const void* w, const int32_t* nbr, int32_t ld, void* out,
                           const int32_t* n_out_dev, int32_t n_out_cap, int32_t cin, int32_t cout, int32_t kvol,
                           int32_t transpose_w, u3d_stream s);
/* out = conv(in) + addend in one pass (addend bf16, out's shape): the input gradient of a residual block's first conv with the
 * residual branch's gradient summed in by the epilogue (ref: SparseBasicBlock's `out += identity`, mmdet3d; autograd adds the two
 * gradients with a separate element-wise kernel).  U3D_ERR_UNSUPPORTED for shapes without that epilogue: the caller adds. */
int32_t u3d_igemm_fwd_add_bf16(const void* in, const void* w, const int32_t* nbr, int32_t ld, const void* addend, void* out,
                               const int32_t* n_out_dev, int32_t n_out_cap, int32_t cin, int32_t cout, int32_t kvol,
                               int32_t transpose_w, u3d_stream s);
/* nn.Linear on rows with the same kernel: out[M,N] = act(x[M,K] @ W[N,K]^T + bias); bf16 x/W/out, f32 bias (may be NULL),
 * relu != 0 applies max(.,0).  (ref: the Linear layers of models/utils/uni3detr_transformer.py:18-30,142-143,248-260 and
 * models/dense_heads/uni3detr_head.py:365-387.)  U3D_ERR_UNSUPPORTED unless K % 64 == 0 and N % 64 == 0. */
int32_t u3d_linear_bf16(const void* x, const void* w, const float* bias, int32_t relu, void* out, const int32_t* m_dev,
                        int32_t m_cap, int32_t k, int32_t n, u3d_stream s);
/* Split-bf16 convolution: f32-grade products on the bf16 matrix pipe, for the modules the reference keeps in fp32 (ref:
 * models/pts_encoder/sparse_encoder_hd.py:62-64 fp16_enabled=False, models/detectors/uni3detr.py:150-151; SECOND3D is never wrapped in
 * auto_fp16).  An f32 row matrix x travels as two bf16 planes (u3d_split_rows_f32: hi = bf16(x), lo = bf16(x - hi), stacked as rows
 * [hi ; lo] with a plane stride of n_cap rows) and x.w ~ hi.wh + hi.wl + lo.wh with f32 accumulation.  The caller passes the three
 * products as three sets of offsets: nbr int32 [kvol3][ld] = (nbr, nbr, nbr + n_in_cap), w bf16 [kvol3][Cout][Cin] = (wh, wl, wh),
 * kvol3 = 3 x offsets.  out is F32 [n_out_cap][Cout]; stats: NULL or f64 [ceil(n_out_cap / u3d_igemm_fwd_stats_rows(.., kvol3))][2][Cout]
 * per-tile BatchNorm sums of the f32 output; addend: NULL or f32 [n_out_cap][Cout] summed into the result in the epilogue (the
 * residual / fan-out gradient sums of the input gradients, as u3d_igemm_fwd_add_bf16).  U3D_ERR_UNSUPPORTED unless Cin % 64 == 0 and
 * Cout % 64 == 0. */
int32_t u3d_igemm_fwd_split_bf16(const void* in, const void* w, const int32_t* nbr, int32_t ld, float* out, const int32_t* n_out_dev,
                                 int32_t n_out_cap, int32_t cin, int32_t cout, int32_t kvol3, double* stats, const float* addend, u3d_stream s);
/* The same product on the NARROW 27-offset sparse levels (cin, cout in {16, 32, 64}, not 64 -> 64; the encoder's stride-1 / stride-2
 * stages, ref: sparse_encoder_hd.py:140-214): three launches of the direct-operand kernel (igemm_direct.hip) accumulating into one f32
 * output - no tripled table: in = the bf16 planes [2 * n_in_cap][cin], w3 = (wh, wl, wh) n-major [3 * 27][cout][cin], nbr / ld as
 * u3d_igemm_fwd_bf16 (ld < 0: the forward table read reversed = the SubM input gradient), out f32 [n_out_cap][cout]. */
int32_t u3d_igemm_direct_split_bf16(const void* in, const void* w3, const int32_t* nbr, int32_t ld, float* out, const int32_t* n_out_dev,
                                    int32_t n_out_cap, int32_t n_in_cap, int32_t cin, int32_t cout, u3d_stream s);
/* The weight side of the same product: dst bf16 [3][k][a][b] = (hi, lo, hi) of the f32 element src[ik * sk + ia * sa + ib * sb]
 * (element strides: the checkpoint layouts [kD,kH,kW,Cin,Cout] and [Cout,Cin,kD,kH,kW] are read in place). */
int32_t u3d_split3_weights(const float* src, int64_t sk, int64_t sa, int64_t sb, int32_t k, int32_t a, int32_t b, void* dst, u3d_stream s);
/* The same for every convolution weight of a step in ONE launch.  jobs: device array of njobs records of u3d_split3_job_bytes() bytes,
 * natural C layout { const float* src; void* dst; int64_t sk, sa, sb; int32_t k, a, b, first_block; }, first_block = sum of
 * u3d_split3_job_blocks(k, a, b) over the jobs before this one; total_blocks = that sum over all jobs. */
/* hi / lo bf16 planes (as u3d_split_rows_f32, all rows live) of up to 32 strided f32 row matrices in one launch; `jobs` is a HOST
 * structure (its contents travel in the kernel arguments; first_block is filled by the call). */
typedef struct U3dSplitRowsJobs {
  const float* src[32];
  void* dst[32];
  int32_t ld[32], rows[32], cols[32];
  int32_t first_block[33];
  int32_t njobs;
} U3dSplitRowsJobs;
int32_t u3d_split_rows_batch(const U3dSplitRowsJobs* jobs, u3d_stream s);
/* out = (a + b) + c over n f32 elements (n % 4 == 0, 16-byte aligned): the three partial weight gradients of a split-bf16 product. */
int32_t u3d_sum3_f32(const float* a, const float* b, const float* c, float* out, int64_t n, u3d_stream s);
int64_t u3d_split3_job_bytes(void);
int32_t u3d_split3_job_blocks(int32_t k, int32_t a, int32_t b);
int32_t u3d_split3_weights_batch(const void* jobs, int32_t njobs, int32_t total_blocks, u3d_stream s);
/* dst bf16 [2 * n_cap][c]: rows [0, n) = bf16(x), rows [n_cap, n_cap + n) = bf16(x - hi), n = min(*n_dev, n_cap); c % 4 == 0. */
int32_t u3d_split_rows_f32(const float* x, const int32_t* n_dev, int32_t n_cap, int32_t c, void* dst, u3d_stream s);
/* Forward with n-major weights w[K][Cout][Cin] (the layout u3d_igemm_fwd_bf16 takes with transpose_w = 1) that also emits the
 * BatchNorm statistics of its (bf16-rounded) output per row tile: stats f64 [ceil(n_out_cap / T)][2][Cout] with
 * T = u3d_igemm_fwd_stats_tile_rows(...) (0: shape not served - use u3d_igemm_fwd_bf16 + u3d_bn_stats).  Feeds
 * u3d_bn_finalize_partials; saves the separate statistics pass over the conv output (ref: conv -> BatchNorm pairs of
 * sparse_encoder_hd.py:71-104 and second_3d.py:52-76). */
int32_t u3d_igemm_fwd_stats_tile_rows(int32_t n_out_cap, int32_t cin, int32_t cout);
/* Number of partials [blocks][2][Cout] u3d_igemm_fwd_stats_bf16 writes for this shape (0: not served).  The direct-operand kernels of
 * the narrow 27-offset levels (Cin, Cout in {16, 32, 64}, not 64 -> 64) write one partial per WAVE of their persistent grid:
 * u3d_igemm_fwd_stats_tile_rows is 0 for them and u3d_bn_finalize_partials takes rows_per_block = 0 (= every partial counts). */
int32_t u3d_igemm_fwd_stats_blocks(int32_t n_out_cap, int32_t cin, int32_t cout, int32_t kvol);
/* Rows per partial (= rows_per_block of u3d_bn_finalize_partials) for a conv with a neighbour table and kvol offsets: like
 * u3d_igemm_fwd_stats_tile_rows, but aware of the kernels whose choice depends on the reduction length (256 x 128 eight-phase tiles
 * for long reductions); 0 = per-wave partials / shape not served. */
int32_t u3d_igemm_fwd_stats_rows(int32_t n_out_cap, int32_t cin, int32_t cout, int32_t kvol);
int32_t u3d_igemm_fwd_stats_bf16(const void* in, const void* w, const int32_t* nbr, int32_t ld, void* out,
                                 const int32_t* n_out_dev, int32_t n_out_cap, int32_t cin, int32_t cout, int32_t kvol,
                                 double* stats, u3d_stream s);
/* BatchNorm-BACKWARD statistics out of the input-gradient launch that writes the BatchNorm's dy (ref: the conv -> BatchNorm -> ReLU
 * chains of sparse_encoder_hd.py:71-104 / second_3d.py:52-76; torch's batch_norm_backward makes a separate reduction pass over dy and
 * x).  x: the BatchNorm's input (bf16 [n][C]); y: its output when the ReLU mask cannot be recomputed from x (residual layers), else
 * NULL; mean/invstd (+ gamma/beta when relu and y == NULL) f32 [C].  The launch leaves, per row tile, sum(g) and sum(g * xhat) with
 * g = dy under the ReLU mask: f64 [ceil(n/T)][2][C], T = u3d_igemm_fwd_stats_rows(...) (128 for the halo kernel), which
 * u3d_bn_bwd_finalize_partials reduces to what u3d_bn_bwd_stats returns. */
typedef struct {
  const void* x;
  const void* y;
  const float* mean;
  const float* invstd;
  const float* gamma;
  const float* beta;
  int32_t relu;
  int32_t reserved;
} u3d_bn_epi;
int32_t u3d_igemm_dgrad_bnstats_bf16(const void* in, const void* w, const int32_t* nbr, int32_t ld, const void* addend, void* out,
                                     const int32_t* n_out_dev, int32_t n_out_cap, int32_t cin, int32_t cout, int32_t kvol,
                                     const u3d_bn_epi* bn, double* stats, u3d_stream s);
int32_t u3d_bn_bwd_finalize_partials(const double* partial, int32_t nblocks, int32_t rows_per_block, const int32_t* n_dev,
                                     int32_t n_cap, int32_t c, double* sums, float* sums_f32, u3d_stream s);
/* Output-stationary path of the 64 -> 64 channel, 27-offset submanifold convs (the stride-4 stage's SparseBasicBlocks, ref:
 * sparse_encoder_hd.py:106-138): rows are numbered in 4x4x4-block-major order, so the 27 x 128 table entries of a 128-row tile name
 * only a few hundred DISTINCT rows.  u3d_subm_halo_build (once per level and step) leaves, per tile, the sorted distinct rows
 * tile_rows int32 [tiles][3460] (slot 0 = the all-zero row), their number tile_cnt int32 [tiles] and the table rewritten as 16-bit
 * slots loc u16 [tiles][27][128] (within a (tile, offset): entry (r % 16) * 8 + r / 16 for row r).  u3d_subm_halo_sizes gives the
 * element counts for n_cap rows (the build returns U3D_ERR_UNSUPPORTED above 869 k rows: its per-tile row bitmap lives in LDS).
 * u3d_subm_halo_conv64_bf16 then computes what u3d_igemm_fwd_bf16 (transpose_w = 1) /
 * u3d_igemm_fwd_add_bf16 / u3d_igemm_fwd_stats_bf16 compute from the table: in/out/addend bf16 [n][64]; w_packed =
 * u3d_subm_halo_wpack of the n-major bf16 weights [27][64][64] (MFMA fragment order, same size); krev != 0 reads the offsets reversed (the transposed table of a SubM layer: input gradients, pass [K][Cin][Cout]);
 * addend nullable; stats nullable f64 [tiles][2][64] (128 rows per partial, as u3d_bn_finalize_partials takes them); bn nullable:
 * with it the statistics are the BatchNorm-backward sums described at u3d_bn_epi above, instead of the output's own. */
int32_t u3d_subm_halo_sizes(int32_t n_cap, int64_t* tile_rows_elems, int64_t* loc_elems, int32_t* tiles);
int32_t u3d_subm_halo_build(const int32_t* nbr, int32_t ld, const int32_t* n_dev, int32_t n_cap, int32_t* tile_rows,
                            uint16_t* loc, int32_t* tile_cnt, int32_t kvol, u3d_stream s);
int32_t u3d_subm_halo_wpack(const void* w_nmajor, void* w_packed, u3d_stream s);
/* n weights in one launch: device arrays of n source / destination pointers (the per-step shadow refresh packs every 64 -> 64 SubM weight,
 * forward and transposed, at once). */
int32_t u3d_subm_halo_wpack_batched(const void* const* srcs_dev, void* const* dsts_dev, int32_t n, u3d_stream s);
int32_t u3d_subm_halo_conv64_bf16(const void* in, const void* w_packed, const int32_t* tile_rows, const uint16_t* loc,
                                  const int32_t* tile_cnt, const int32_t* n_dev, int32_t n_cap, int32_t krev,
                                  const void* addend, void* out, double* stats, const u3d_bn_epi* bn, int32_t max_slots,
                                  u3d_stream s);
/* The same for 128 -> 128 channels (the stride-8 stage): in/out/addend bf16 [n][128], w_packed = u3d_subm_halo_wpack128 of the n-major
 * bf16 weights [kvol][128][128]; stats f64 [tiles][2][128].  Tables from u3d_subm_halo_build (they do not depend on the channel count).
 * kvol <= 27 offsets (the build takes the same kvol; loc keeps 27 slots per tile): 27 for the sparse SubM levels, 9 for the stride-1
 * (1,3,3) convs of the dense stack (SECOND3D's 128-channel branch, ref: second_3d.py:52-76), whose tables are static - the transposed
 * table of a stride-1 "same" conv on a lattice is the forward one reversed, exactly as for SubM. */
int32_t u3d_subm_halo_wpack128(const void* w_nmajor, void* w_packed, int32_t kvol, u3d_stream s);
int32_t u3d_subm_halo_wpack128_batched(const void* const* srcs_dev, void* const* dsts_dev, int32_t n, int32_t kvol, u3d_stream s);
int32_t u3d_subm_halo_conv128_bf16(const void* in, const void* w_packed, const int32_t* tile_rows, const uint16_t* loc,
                                   const int32_t* tile_cnt, const int32_t* n_dev, int32_t n_cap, int32_t krev,
                                   const void* addend, void* out, double* stats, int32_t max_slots, int32_t kvol, u3d_stream s);
/* Weight gradient of the same 64 -> 64 SubM layers from the same tables: dw f32 [27][64][64] (spconv-1.x layout) =
 * sum over rows m of x[nbr_k(m)]^T dy[m]; x / dy bf16 [n][64].  Persistent workgroups, both MFMA operands by transpose reads out of
 * the staged distinct rows / the dy tile, offsets split over four workgroup groups, one f32 partial per workgroup summed in a fixed
 * order (deterministic).  workspace: u3d_subm_halo_wgrad64_workspace() bytes.  max_slots (both functions): 0 = the stage buffer's
 * capacity; a lower value only forces the fall-back paths for tiles with more distinct rows (test hook).  Replaces u3d_igemm_wgrad_bf16 for this shape (ref: the
 * weight half of spconv's indice_conv_backward for sparse_encoder_hd.py:106-138). */
int64_t u3d_subm_halo_wgrad64_workspace(void);
int32_t u3d_subm_halo_wgrad64_bf16(const void* x, const void* dy, const int32_t* tile_rows, const uint16_t* loc,
                                   const int32_t* tile_cnt, const int32_t* n_dev, int32_t n_cap, float* dw, void* workspace,
                                   int64_t workspace_bytes, int32_t max_slots, u3d_stream s);
int64_t u3d_igemm_wgrad_bf16_workspace(int32_t n_out_cap, int32_t cin, int32_t cout, int32_t kvol);
/* out_layout 0: dw [K][Cin][Cout] (spconv-1.x / this library's layout); 1: dw [Cout][Cin][K] (nn.Conv3d's [Cout,Cin,kD,kH,kW]). */
int32_t u3d_igemm_wgrad_bf16(const void* in, const void* dout, const int32_t* nbr, int32_t ld, float* dw,
                             const int32_t* n_out_dev, int32_t n_out_cap, int32_t cin, int32_t cout, int32_t kvol,
                             int32_t out_layout, void* workspace, int64_t workspace_bytes, u3d_stream s);

/* dW[kappa] = sum_m in[nbr[kappa][m],:]^T @ dout[m,:]   (f32 accumulate, dW f32 [K,Cin,Cout], overwritten).
 * workspace: u3d_spconv_wgrad_workspace() bytes. */
int64_t u3d_spconv_wgrad_workspace(int32_t n_out_cap, int32_t cin, int32_t cout, int32_t kvol);
int32_t u3d_spconv_wgrad(const void* in, const void* dout, const int32_t* nbr, int32_t ld, float* dw,
                         const int32_t* n_out_dev, int32_t n_out_cap, int32_t cin, int32_t cout, int32_t kvol,
                         int32_t dtype, void* workspace, int64_t workspace_bytes, u3d_stream s);

/* out[c] = sum_r x[r, c] over a dense [n, C] matrix (f32 or bf16 in, f32 out, fixed summation order): the bias gradient of every
 * nn.Linear of the decoder / head (ref: uni3detr_transformer.py:93-214, uni3detr_head.py:95-125 — autograd's sum-to-size there). */
int64_t u3d_colsum_workspace(int32_t n, int32_t c);
int32_t u3d_colsum(const void* x, int32_t n, int32_t c, int32_t dtype, float* out, void* workspace,
                   int64_t workspace_bytes, u3d_stream s);

/* Batched forms for the parameter gradients of the decoder / head linears (one shape per call, count <= 48 / 64): all weight
 * gradients dW_b = in_b^T @ dout_b (bf16 [n_rows, cin] x [n_rows, cout] -> f32 [cin, cout]) in two launches, all bias gradients
 * in two launches; pointer arrays are HOST arrays of device pointers (they travel in the kernel arguments).  CONSECUTIVE batch
 * slots that name the same output are SUMMED into it in a fixed order (a linear shared by several decoder layers: ref
 * models/utils/uni3detr_transformer.py:276-283 ref_point_head / query_scale) - no separate accumulate launches. */
int64_t u3d_wgrad_batched_workspace(int32_t count, int32_t n_rows, int32_t cin, int32_t cout);
int32_t u3d_wgrad_batched_bf16(const void* const* in, const void* const* dout, float* const* dw, int32_t count,
                               const int32_t* n_dev, int32_t n_rows, int32_t cin, int32_t cout, void* workspace,
                               int64_t workspace_bytes, u3d_stream s);
int64_t u3d_colsum_batched_workspace(int32_t count, int32_t n, int32_t c);
int32_t u3d_colsum_batched(const void* const* x, float* const* out, int32_t count, int32_t n, int32_t c, int32_t dtype,
                           void* workspace, int64_t workspace_bytes, u3d_stream s);

/* Weight gradient of a linear layer with <= 16 input or output features (bf16 dy [m, n], x [m, k]): partial f32
 * [u3d_skinny_wgrad_chunks(m)][n*k]; the column sums over the chunks are dW [n][k] (u3d_colsum / u3d_colsum_batched). */
int32_t u3d_skinny_wgrad_chunks(int32_t m);
int32_t u3d_skinny_wgrad_bf16(const void* dy, const void* x, int32_t m, int32_t n, int32_t k, float* partial, u3d_stream s);
/* `count` (<= 32) such products over the same m rows in one launch; dy / x / partial / n / k are HOST arrays of length count;
 * dy_skinny != 0: every n[i] <= 32 (thread = column of x), else every k[i] <= 32 (thread = column of dy). */
int32_t u3d_skinny_wgrad_batched(const void* const* dy, const void* const* x, float* const* partial, const int32_t* n, const int32_t* k,
                                 int32_t count, int32_t m, int32_t dy_skinny, u3d_stream s);

/* ------------------------------------------------------------------------------------------------
 * BatchNorm1d over sparse rows [n, C] (training statistics) with optional residual add and ReLU
 * (ref: sparse_encoder_hd.py:40; upstream make_sparse_convmodule / SparseBasicBlock).
 * stats: sums f64 [2*C] = (sum x, sum x^2), deterministic two-stage reduction.
 * ---------------------------------------------------------------------------------------------- */
int64_t u3d_bn_stats_workspace(int32_t n_cap, int32_t c);
int32_t u3d_bn_stats(const void* x, const int32_t* n_dev, int32_t n_cap, int32_t c, int32_t dtype,
                     double* sums, void* workspace, int64_t workspace_bytes, u3d_stream s);
/* sums -> mean, invstd = 1/sqrt(biased var + eps); optional nn.BatchNorm1d running-stat update
 * (running = (1-m)*running + m*batch, unbiased variance) and num_batches_tracked += 1. */
int32_t u3d_bn_finalize(const double* sums, const int32_t* n_dev, int32_t n_cap, int32_t c, float eps, float momentum,
                        float* running_mean, float* running_var, int64_t* num_batches, float* mean, float* invstd,
                        u3d_stream s);
/* u3d_bn_stats + u3d_bn_finalize in two launches (stage 2 of the reduction fused with the finalisation); same results. */
int32_t u3d_bn_forward_stats(const void* x, const int32_t* n_dev, int32_t n_cap, int32_t c, int32_t dtype, float eps,
                             float momentum, float* running_mean, float* running_var, int64_t* num_batches, float* mean,
                             float* invstd, void* workspace, int64_t workspace_bytes, u3d_stream s);
/* Same finalisation from statistics the producing convolution already reduced per row tile (u3d_igemm_fwd_stats_bf16):
 * partial f64 [nblocks][2][C], tile b = rows [b*rows_per_block, (b+1)*rows_per_block). */
int32_t u3d_bn_finalize_partials(const double* partial, int32_t nblocks, int32_t rows_per_block, const int32_t* n_dev,
                                 int32_t n_cap, int32_t c, float eps, float momentum, float* running_mean, float* running_var,
                                 int64_t* num_batches, float* mean, float* invstd, u3d_stream s);
/* y = relu?( (x-mean)*invstd*gamma + beta (+ residual) ); mean/invstd/gamma/beta f32 [C].
 * row_map (nullable; all three functions): y / dy are stored in a permuted row order, row r of x <-> row row_map[r] of y / dy
 * (a bijection of [0, n)).  The (1,s,s)/(1,s,s) transposed convolutions of the FPN (ref: second3d_fpn.py:60-75) produce their
 * rows tap-major; their BatchNorm writes the lattice order directly instead of a separate row gather.  Requires no residual /
 * dres and C % 8 == 0 (bf16) or C % 4 == 0 (f32).
 * post_add (nullable, u3d_bn_apply): a tensor of y's shape and row order added AFTER the activation, y = act(..) + post_add - the
 * running sum of the FPN's upsampled levels (ref: second3d_fpn.py:131-134) without a separate add pass. */
int32_t u3d_bn_apply(const void* x, const float* mean, const float* invstd, const float* gamma,
                     const float* beta, const void* residual, int32_t relu, void* y,
                     const int32_t* n_dev, int32_t n_cap, int32_t c, int32_t dtype, const int32_t* row_map, const void* post_add,
                     u3d_stream s);
/* backward: given dy (grad wrt y), y (for relu mask), x: sums f64 [2*C] = (sum g, sum g*xhat) where
 * g = dy * (y>0 if relu).  y may be NULL when relu is set and the forward had NO residual: the mask is then recomputed as
 * (x-mean)*invstd*gamma+beta > 0 (the forward's own expression; gamma/beta required) - one tensor less to stream.
 * sums_f32 (nullable, f32 [2*C]): the same sums rounded to f32 = (d beta, d gamma), the parameter gradients as torch stores them. */
int32_t u3d_bn_bwd_stats(const void* dy, const void* y, const void* x, const float* mean, const float* invstd,
                         const float* gamma, const float* beta, int32_t relu, const int32_t* n_dev, int32_t n_cap,
                         int32_t c, int32_t dtype, double* sums, void* workspace, int64_t workspace_bytes,
                         const int32_t* row_map, float* sums_f32, u3d_stream s);
/* dx = gamma*invstd*( g - sum_g/n - xhat*sum_gx/n ); dres = g (optional, may be NULL).  y NULL: as above (beta required). */
int32_t u3d_bn_bwd_apply(const void* dy, const void* y, const void* x, const float* mean, const float* invstd,
                         const float* gamma, const float* beta, const double* sums, int32_t relu, void* dx, void* dres,
                         const int32_t* n_dev, int32_t n_cap, int32_t c, int32_t dtype, const int32_t* row_map, u3d_stream s);
/* The same two passes over F32 rows that ALSO leave their output as the hi / lo bf16 planes of a split-bf16 product (planes: bf16
 * [2 * n_cap][C] in the output's row order, rows n .. n_cap of both planes zero - exactly what u3d_split_rows_f32 of the output would
 * hold): the BatchNorm of the fp32 modules (ref: sparse_encoder_hd.py:62-64, second_3d.py:52-76 - kept in fp32 by the reference)
 * hands its consumer convolution the planes without a further pass over the tensor; the backward hands the producer convolution the
 * planes of dy.  U3D_ERR_UNSUPPORTED when C does not fit the vector kernels (C % 4, (C / 4) | 256): run the plain pass + the split. */
int32_t u3d_bn_apply_planes(const float* x, const float* mean, const float* invstd, const float* gamma, const float* beta,
                            const float* residual, int32_t relu, float* y, void* planes, const int32_t* n_dev, int32_t n_cap,
                            int32_t c, const int32_t* row_map, const float* post_add, u3d_stream s);
int32_t u3d_bn_bwd_apply_planes(const float* dy, const float* y, const float* x, const float* mean, const float* invstd,
                                const float* gamma, const float* beta, const double* sums, int32_t relu, float* dx, float* dres,
                                void* planes, const int32_t* n_dev, int32_t n_cap, int32_t c, const int32_t* row_map, u3d_stream s);

/* ------------------------------------------------------------------------------------------------
 * SparseConvTensor.dense() (ref: sparse_encoder_hd.py:133): rows -> channels-last dense volume
 * dense[b, z, y, x, :] (= a torch channels_last_3d tensor of logical shape [B,C,D,H,W]); caller zeroes it.
 * ---------------------------------------------------------------------------------------------- */
int32_t u3d_to_dense(const void* feat, const int32_t* coors, const int32_t* n_dev, int32_t n_cap, int32_t c,
                     void* dense, int32_t dz, int32_t dy, int32_t dx, int32_t dtype, u3d_stream s);
int32_t u3d_from_dense(const void* dense, const int32_t* coors, const int32_t* n_dev, int32_t n_cap, int32_t c,
                       void* feat, int32_t dz, int32_t dy, int32_t dx, int32_t dtype, u3d_stream s);

/* row gather / permutation: out[i,:] = in[idx[i],:] (idx<0 -> zeros). elem_bytes*c must be a multiple of 4 */
int32_t u3d_gather_rows(const void* in, const int32_t* idx, int32_t n, int32_t row_bytes, void* out, u3d_stream s);
/* out[idx[i],:] = in[i,:]  (idx<0 skipped; idx must be injective) */
int32_t u3d_scatter_rows(const void* in, const int32_t* idx, int32_t n, int32_t row_bytes, void* out, u3d_stream s);

/* ------------------------------------------------------------------------------------------------
 * D-FPS (ref: models/detectors/uni3detr.py:138,178-187; upstream mmcv PointsSampler / furthest_point_sample,
 * SURVEY.md App. A5).  nsets independent point sets; point k of set s is the float triple base[set_off[s]+3k..+2]
 * (the packed-triple view the upstream kernel takes of whatever buffer it is handed).  idx[0]=0; squared-L2 running
 * minimum; arg-max ties resolved as the upstream 2^k-thread block reduction does: smallest (k mod T, k),
 * T = min(1024, 2^floor(log2 n)).  out_idx int32 [nsets, m].  temp: f32 [nsets, temp_stride >= max_n] workspace, only needed
 * when max_n > 20480: such sets run either on ceil(n / 20480) <= 16 resident workgroups per set that exchange their round
 * winners through the head of temp (sets go out in launches of at most max_wg workgroups; max_wg = 0: 3/4 of the device's CUs,
 * hipDeviceAttributeMultiprocessorCount), or - beyond 16 x 20480 points, or when one set needs more than max_wg workgroups
 * (max_wg = 1: always) - on one workgroup streaming the running minima through temp.  Same indices either way.
 * err: int32 [2], device; REQUIRED for the several-workgroup form, optional otherwise.  err[0] is written by the call: 0, or 1
 * when a workgroup waited longer than poll_ticks (100 MHz ticks of the constant clock; 0 = 0.5 s) for a sibling's round winner -
 * then every workgroup of the call leaves at once, unfinished rounds hold index 0 and the samples of this call MUST NOT be
 * used (the training step ORs err[0] into its collective hold flag).  err[1] += 1 per call that timed out (the caller clears it).
 * ---------------------------------------------------------------------------------------------- */
int32_t u3d_fps(const float* base, const int64_t* set_off, const int32_t* set_n, int32_t nsets, int32_t max_n,
                int32_t m, int32_t* out_idx, float* temp, int64_t temp_stride, int32_t* err, int64_t poll_ticks, int32_t max_wg,
                u3d_stream s);
/* the same over two buffers: sets [0, split) are offsets into base, sets [split, nsets) into base2 */
int32_t u3d_fps2(const float* base, const float* base2, int32_t split, const int64_t* set_off, const int32_t* set_n,
                 int32_t nsets, int32_t max_n, int32_t m, int32_t* out_idx, float* temp, int64_t temp_stride, int32_t* err,
                 int64_t poll_ticks, int32_t max_wg, u3d_stream s);
/* The detector's glue around its two FPS passes (ref: models/detectors/uni3detr.py:178-189), one launch each side.
 *   u3d_fps_prep: vox f32 [v_rows,3] = float (z,y,x) of coors int32 [v_rows,4] (b,z,y,x); set descriptors for u3d_fps2 with
 *     split = batch: sets 0..B-1 = the packed-triple view of the [N,nfeat] point buffer from row scene_off[b] (offset
 *     scene_off[b]*nfeat, n = scene_off[b+1]-scene_off[b]); sets B..2B-1 = the voxel rows voxel_off[b]..voxel_off[b+1]
 *     (offset voxel_off[b]*3 into vox).  set_off int64 [2B], set_n int32 [2B].
 *   u3d_fps_points: idx int32 [2B,m] (u3d_fps2's output) -> out f32 [B,2m,3]: rows 0..m-1 = the sampled points' (x,y,z),
 *     rows m..2m-1 = the sampled voxel coordinates as (x,y,z), each group mapped to the unit cube by its per-scene
 *     min / max over the m samples (shift_scale_points, ref :18-46: (x - lo) / (hi - lo)). */
int32_t u3d_fps_prep(const int32_t* coors, int32_t v_rows, const int32_t* scene_off, const int32_t* voxel_off, int32_t batch,
                     int32_t nfeat, float* vox, int64_t* set_off, int32_t* set_n, u3d_stream s);
int32_t u3d_fps_points(const float* pts, int32_t nfeat, const float* vox, const int32_t* idx, const int32_t* scene_off,
                       const int32_t* voxel_off, int32_t batch, int32_t m, float* out, u3d_stream s);
/* Query assembly of Uni3DETRHead.forward (ref: dense_heads/uni3detr_head.py:436-455) in one launch: `groups` query groups of nq
 * queries; group 0 = (tgt[:nq], anchor), group g >= 1 = (tgt[nq:], inverse_sigmoid(points of group g)) with the points of
 * groups 1, 2 in fps f32 [B,2nq,3] and of group 3 (eval layout) in rnd f32 [B,nq,3].  tgt f32 [2nq,c], anchor f32 [nq,3].
 * Outputs: query_embeds f32 [B,G*nq,c+3] and its two column blocks as contiguous tensors, query [B,G*nq,c], ref [B,G*nq,3];
 * ref_sig (nullable) [B,G*nq,3] = sigmoid(ref), the transformer's init_reference.
 * _bwd: d_tgt [2nq,c], d_anchor [nq,3] = sums over scenes (and sharing groups) of the output gradients (each may be null;
 *   d_ref_sig reaches the anchor through the sigmoid, which is why `anchor` is passed). */
int32_t u3d_query_embed_fwd(const float* tgt, const float* anchor, const float* fps, const float* rnd, int32_t batch, int32_t nq,
                            int32_t groups, int32_t c, float* query_embeds, float* query, float* ref, float* ref_sig, u3d_stream s);
int32_t u3d_query_embed_bwd(const float* d_query_embeds, const float* d_query, const float* d_ref, const float* d_ref_sig,
                            const float* anchor, int32_t batch, int32_t nq, int32_t groups, int32_t c, float* d_tgt, float* d_anchor,
                            u3d_stream s);

/* ------------------------------------------------------------------------------------------------
 * Matching (ref: core/bbox/assigners/hungarian_assigner_3d.py:53-151; match_costs/match_cost.py:19-30,91-97; upstream
 * FocalLossCost; scipy.optimize.linear_sum_assignment).
 *   cls f32 [L,B,Q,C] logits, box f32 [L,B,Q,code] codes, gt f32 [sumG,7] gravity-centre boxes, labels int32 [sumG],
 *   gt_off int32 [B+1] (device).  cost f32 [L*B, gmax, Q] (GT-major) = w_cls*focal + w_reg*L1(8 code dims) +
 *   w_iou*(1 - nearest-BEV IoU).  u3d_lsa solves one problem per (layer, scene, query group of nq) on one wavefront
 *   in float64 with scipy's scan order and tie rule; assigned int32 [L,B,Q]: 0 background, else 1-based GT index.
 * ---------------------------------------------------------------------------------------------- */
int32_t u3d_match_cost(const float* cls, const float* box, const float* gt, const int32_t* labels,
                       const int32_t* gt_off, int32_t nlayers, int32_t batch, int32_t nq_total, int32_t ncls,
                       int32_t code_size, int32_t gmax, float w_cls, float w_reg, float w_iou, float alpha,
                       float gamma, float* cost, u3d_stream s);
int32_t u3d_lsa(const float* cost, const int32_t* gt_off, int32_t nlayers, int32_t batch, int32_t nq_total,
                int32_t nq, int32_t gmax, int32_t* assigned, u3d_stream s);

/* Trilinear sampling of a channels-last volume at query points = UniCrossAtten's F.grid_sample (mode bilinear, zeros padding,
 * align_corners=False; ref: models/utils/uni3detr_transformer.py:342-345).  value: rows [batch*dz*dy*dx, c] (f32|bf16),
 * grid f32 [batch,nq,3] (x,y,z) in [-1,1], out [batch,nq,c].  Backward: dvalue f32 [rows,c] accumulated with atomics
 * (caller zeroes; may be NULL), dgrid f32 [batch,nq,3] (may be NULL). */
int32_t u3d_trilinear_fwd(const void* value, const float* grid, int32_t batch, int32_t nq, int32_t dz, int32_t dy, int32_t dx,
                          int32_t c, void* out, int32_t dtype, u3d_stream s);
int32_t u3d_trilinear_bwd(const void* value, const float* grid, const void* dout, int32_t batch, int32_t nq, int32_t dz,
                          int32_t dy, int32_t dx, int32_t c, float* dvalue, float* dgrid, int32_t dtype, u3d_stream s);

/* diag(bbox_overlaps_3d(a,b)) for a,b f32 [n,7] (ref: models/dense_heads/uni3detr_head.py:695). */
int32_t u3d_iou3d_rotated_aligned(const float* a, const float* b, int32_t n, float* out, u3d_stream s);

/* Class-aware rotated-BEV NMS (ref: models/dense_heads/uni3detr_head.py:849-865 -> mmcv.ops.nms3d applied per class).
 * boxes f32 [n,7] sorted by descending score, labels int32 [n]; keep uint8 [n] (1 = survives).  A box is suppressed by a
 * kept earlier box of the same label with BEV rotated IoU > thr (height ignored). */
int64_t u3d_nms3d_workspace(int32_t n);
int32_t u3d_nms3d(const float* boxes, const int32_t* labels, int32_t n, float thr, uint8_t* keep, void* workspace,
                  int64_t workspace_bytes, u3d_stream s);

/* Second half of the strided-convolution input gradient (first half: P = dout @ [W_0^T | ... | W_{K-1}^T] with u3d_linear_bf16 on
 * the weight viewed as [K*Cin, Cout]): din[i][c] = sum_kappa P[nbr[kappa][i]][kappa*C + c], nbr = transposed table (mode 1 of
 * u3d_nbr_table / u3d_dense_nbr_table), f32 accumulation.  Replaces the dgrad half of spconv's indice_conv_backward / cuDNN dgrad
 * for stride > 1 (ref: models/backbones/second_3d.py:52-76 first conv of each block). */
int32_t u3d_tap_gather_sum(const void* p, const int32_t* nbr, int32_t ld, const int32_t* n_dev, int32_t n_cap, int32_t c,
                           int32_t kvol, int32_t dtype, void* out, u3d_stream s);
/* ... + addend (nullable; out's shape and dtype): the input gradients the other branches of a shared input already summed
 * (SECOND3D with is_cascade=False, ref: second_3d.py:89-114) - autograd's separate element-wise add over the full tensor disappears. */
int32_t u3d_tap_gather_sum_add(const void* p, const int32_t* nbr, int32_t ld, const int32_t* n_dev, int32_t n_cap, int32_t c,
                               int32_t kvol, int32_t dtype, const void* addend, void* out, u3d_stream s);

/* ------------------------------------------------------------------------------------------------
 * LayerNorm over the last dimension of a row matrix [n, C] (C <= 1024) with optional fused ReLU; input and output dtypes are
 * independent (f32 / bf16).  Replaces nn.LayerNorm (+ nn.ReLU) of the decoder layers, the position encoder and the head branches
 * (ref: models/utils/uni3detr_transformer.py:232-236, models/dense_heads/uni3detr_head.py:95-125, mmcv BaseTransformerLayer).
 * fwd: y = relu?((x-mean)*rstd*gamma+beta); mean/rstd f32 [n] are saved for the backward.
 * bwd: dx (dtype of x) and partial f32 [2][u3d_layernorm_blocks(n)][C] = per-workgroup sums of (dgamma, dbeta) terms; the caller
 *      column-sums them (u3d_colsum / u3d_colsum_batched).  With relu the mask is recomputed from x.
 * ---------------------------------------------------------------------------------------------- */
int32_t u3d_layernorm_blocks(int32_t n);
int32_t u3d_layernorm_fwd(const void* x, int32_t x_dtype, int32_t n, int32_t c, const float* gamma, const float* beta,
                          float eps, int32_t relu, void* y, int32_t y_dtype, float* mean, float* rstd, u3d_stream s);
int32_t u3d_layernorm_bwd(const void* dy, int32_t y_dtype, const void* x, int32_t x_dtype, int32_t n, int32_t c,
                          const float* gamma, const float* beta, const float* mean, const float* rstd, int32_t relu,
                          void* dx, float* partial, u3d_stream s);

/* ------------------------------------------------------------------------------------------------
 * Detection losses of Uni3DETRHead for all L decoder layers at once (ref: models/dense_heads/uni3detr_head.py:579-720 loss_single /
 * loss, models/losses/rdiouloss.py:93-223, core/bbox/util.py:8-80).  m = B*Q elements per layer, tensors [L, m, *] contiguous f32:
 * cls [L,m,C] logits, box [L,m,code] codes (code 8 or 10), iou_logit [L,m], tgt [L,m,code-1] assigned GT boxes (zeros for
 * background), lab int64 [L,m] (C = background), w [L,m] (1 for matched), iou_true [L,m] rotated 3-D IoU targets, cls_avg / npos [L]
 * avg factors (already clamped to >= 1), code_w [code].  out [L][4] = (loss_cls, loss_bbox, loss_iou, loss_iou_pred) per layer.
 * bwd: gout [L][4] = incoming gradients of those scalars; dcls / dbox / diou have the shapes of cls / box / iou_logit.
 * The quality (soft) target of the focal loss is NOT detached; gamma of the focal weight is 2.
 * ---------------------------------------------------------------------------------------------- */
int64_t u3d_det_loss_workspace(int32_t L, int32_t m);
int32_t u3d_det_loss_fwd(const float* cls, const float* box, const float* iou_logit, const float* tgt, const int64_t* lab,
                         const float* w, const float* iou_true, const float* cls_avg, const float* npos, const float* code_w,
                         int32_t L, int32_t m, int32_t c, int32_t code, int32_t tdim, float alpha, float w_cls, float w_box,
                         float w_iou, float eps, float* out, void* workspace, int64_t workspace_bytes, u3d_stream s);
int32_t u3d_det_loss_bwd(const float* cls, const float* box, const float* iou_logit, const float* tgt, const int64_t* lab,
                         const float* w, const float* iou_true, const float* cls_avg, const float* npos, const float* code_w,
                         const float* gout, int32_t L, int32_t m, int32_t c, int32_t code, int32_t tdim, float alpha, float w_cls,
                         float w_box, float w_iou, float eps, float* dcls, float* dbox, float* diou, u3d_stream s);
/* Target construction behind the assignment (ref: dense_heads/uni3detr_head.py:510-570), all layers and scenes in one launch:
 * asg int32 [L,B,Q] (u3d_lsa: 0 background, else 1-based GT of the scene), gt f32 [sumG,gd], labels int32 [sumG], gt_off int32 [B+1] ->
 * asg64 int64 [L,B,Q], w f32 [L,B,Q] (1 on matched queries), tgt f32 [L,B,Q,gd] (matched GT row, zeros for background),
 * lab int64 [L,B,Q] (GT label, ncls for background), num_pos f32 [L] (matched queries per layer). */
int32_t u3d_loss_targets(const int32_t* asg, const float* gt, const int32_t* labels, const int32_t* gt_off, int32_t L, int32_t B,
                         int32_t Q, int32_t gd, int32_t ncls, int64_t* asg64, float* w, float* tgt, int64_t* lab, float* num_pos,
                         u3d_stream s);
/* codes [n, code] -> boxes [n, 7] (cx, cy, cz, dx, dy, dz, yaw) (ref: core/bbox/util.py denormalize_bbox). */
int32_t u3d_denormalize_boxes(const float* codes, int32_t n, int32_t code, float* boxes, u3d_stream s);
/* Box decode of the head (ref: dense_heads/uni3detr_head.py:475-490): out = tmp with columns 0,1,4 replaced by
 * sigmoid(tmp + inverse_sigmoid(ref)[x,y,z]) * (pc_range hi - lo) + lo.  tmp [n, code] f32 or bf16 (dtype), ref [n, 3] f32 in
 * sigmoid space (inverse_sigmoid clamps at eps, 1e-5 upstream), pc_range 6 host floats, out f32 [n, code].  The backward returns
 * dtmp in tmp's dtype and, when dref != NULL, the gradient w.r.t. ref (layer 0's reference points depend on learned anchors). */
int32_t u3d_box_decode_fwd(const void* tmp, int32_t dtype, const float* ref, int32_t n, int32_t code, const float* pc_range,
                           float eps, float* out, u3d_stream s);
int32_t u3d_box_decode_bwd(const void* tmp, int32_t dtype, const float* ref, const float* dout, int32_t n, int32_t code,
                           const float* pc_range, float eps, void* dtmp, float* dref, u3d_stream s);
/* Reference-point refinement + box decode of one decoder layer in one launch (ref: models/utils/uni3detr_transformer.py:194-202,
 * dense_heads/uni3detr_head.py:463-490): tmp f32 [n,code] (regression branch), ref_in f32 [n,3] (the layer's input reference
 * logits), ref_s f32 [n,3] (the same in sigmoid space) -> out [n,code] (= u3d_box_decode_fwd(tmp, ref_s)), ref_out [n,3] =
 * ref_in + tmp[:,(0,1,4)], ref_sig [n,3] = sigmoid(ref_out).  Backward w.r.t. tmp: u3d_box_decode_bwd(tmp, ref_s, dout). */
int32_t u3d_refine_decode_fwd(const float* tmp, const float* ref_in, const float* ref_s, int32_t n, int32_t code,
                              const float* pc_range, float eps, float* out, float* ref_out, float* ref_sig, u3d_stream s);
/* Sine position embedding of the decoder's reference points (ref: models/utils/uni3detr_transformer.py:33-65,181):
 * out[n][j*nfeat + f] = (f even ? sin : cos)(sigmoid(logits[n][j]) * 2*pi / dim_t[f]); logits f32 [n, nc], dim_t f32 [nfeat] (host
 * table T^(2*(f/2)/nfeat)), out f32 or bf16 [n, nc*nfeat].  Backward: dlogits f32 [n, nc] from dout (f32 or bf16). */
int32_t u3d_sine_embed_fwd(const float* logits, const float* dim_t, int32_t n, int32_t nc, int32_t nfeat, int32_t out_dtype,
                           void* out, u3d_stream s);
int32_t u3d_sine_embed_bwd(const float* logits, const float* dim_t, const void* dout, int32_t dout_dtype, int32_t n, int32_t nc,
                           int32_t nfeat, float* dlogits, u3d_stream s);

/* ------------------------------------------------------------------------------------------------
 * Parameter update of the training step: global-norm gradient clipping + AdamW over FLAT f32 buffers
 * (ref: projects/configs/uni3detr/uni3detr_sunrgbd.py:234-235 — AdamW(lr, weight_decay=0.01), grad_clip max_norm=10; upstream
 * torch.optim.AdamW + torch.nn.utils.clip_grad_norm_, same arithmetic: coef = min(1, max_norm/(||g||+1e-6)), decoupled decay,
 * bias-corrected moments).  state: 16 floats of device memory, zero-initialised by the caller: [0] step count (incremented here),
 * [1] clip coefficient, [2] 1-beta1^t, [3] 1-beta2^t, [4] ||g||, [5..10] lr, beta1, beta2, eps, weight_decay, max_norm (<= 0: no
 * clipping).  The kernels read the hyper-parameters from the state vector at run time: u3d_adamw_set_hyper (a one-thread launch,
 * issued between hipGraph replays) makes a captured u3d_adamw_step_state follow a learning-rate / momentum schedule (ref: the
 * `step` lr policy of uni3detr_sunrgbd.py:236-241 and the `cyclic` lr + momentum policies of the KITTI / nuScenes configs).
 * skip (optional): uint8 per 64-element chunk, 1 = parameter chunk received no gradient and is left untouched (torch.optim.AdamW
 * skips parameters whose .grad is None).  u3d_adamw_step = set_hyper + step_state with host scalars.  All pointers 16-byte aligned.
 * ---------------------------------------------------------------------------------------------- */
int64_t u3d_adamw_workspace(int64_t n);
int32_t u3d_adamw_set_hyper(float* state, float lr, float beta1, float beta2, float eps, float weight_decay, float max_norm,
                            u3d_stream s);
int32_t u3d_adamw_step_state(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float* state,
                             const uint8_t* skip, void* workspace, int64_t workspace_bytes, u3d_stream s);
/* u3d_adamw_step_state with a HOLD flag: hold (nullable) -> one device float; > 0 makes this step a no-op (parameters, moments and
 * the step count untouched; state[11] = 1, state[12] += 1 = held steps so far).  Static-shape training (hipGraph replay over
 * capacity-sized sparse levels) uses it so that a batch that overflowed a level - on ANY rank: the flag rides in the job's
 * positive-count all-reduce - never trains on truncated levels; the host reads state[12] now and then and re-captures. */
int32_t u3d_adamw_step_hold(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float* state,
                            const uint8_t* skip, const float* hold, void* workspace, int64_t workspace_bytes, u3d_stream s);
/* flag[0] = number of i < n (n <= 8) with *counts[i] > caps[i]; counts: HOST array of device pointers, caps: HOST array */
int32_t u3d_capacity_flag(const int32_t* const* counts, const int32_t* caps, int32_t n, float* flag, u3d_stream s);
int32_t u3d_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1,
                       float beta2, float eps, float weight_decay, float max_norm, float* state, void* workspace,
                       int64_t workspace_bytes, u3d_stream s);

/* bf16 shadows of the flat f32 parameter buffer, refreshed once per training step (bf16 mode; upstream has no counterpart: its
 * fp32 run reads the parameters directly).  u3d_cast_bf16: dst[i] = bf16(src[i]) (round to nearest even), 16-byte aligned pointers.
 * u3d_permute_bf16_batched: for every descriptor, dst[dst_off + (k*rows + r)*cols + c] = src[src_off + k*stride_k + r*stride_r +
 * c*stride_c] for k < n/(rows*cols): the conv-weight layouts the implicit-GEMM kernels stage row-linearly, i.e. [K,Cout,Cin] from
 * spconv's [K,Cin,Cout] and both [K,Cin,Cout] / [K,Cout,Cin] from nn.Conv3d's [Cout,Cin,K].  blocks_dev: int32 pairs (descriptor
 * index, first element) - one per u3d_permute_block_elems() output elements; dst_off multiples of 8; descriptors in device memory. */
typedef struct u3d_permute_desc {
  int64_t src_off, dst_off;
  int32_t n, rows, cols, reserved;
  int64_t stride_k, stride_r, stride_c;
} u3d_permute_desc;
int32_t u3d_cast_bf16(const float* src, void* dst, int64_t n, u3d_stream s);
int32_t u3d_permute_block_elems(void);
int32_t u3d_permute_bf16_batched(const void* src, void* dst, const u3d_permute_desc* descs_dev, const int32_t* blocks_dev,
                                 int32_t nblocks, u3d_stream s);
/* The same re-layout through a 32 x 32 x K LDS tile for descriptors whose source is contiguous along k (stride_k == 1, one of
 * stride_r / stride_c == K = n / (rows * cols) <= max_k <= 32, rows % 32 == cols % 32 == 0, even src_off): coalesced reads of
 * nn.Conv3d's [Cout][Cin][K] weights.  tiles_dev: int32 [ntiles][4] = (descriptor index, first row, first column, 0). */
int32_t u3d_permute_bf16_tiled(const void* src, void* dst, const u3d_permute_desc* descs_dev, const int32_t* tiles_dev,
                               int32_t ntiles, int32_t max_k, u3d_stream s);

/* ------------------------------------------------------------------------------------------------
 * Test-time post-processing.
 * u3d_soft_nms: class-wise Gaussian soft-NMS with the rotated 3-D IoU (ref: models/dense_heads/uni3detr_head.py:796-823, per-class loop
 *   :849-880).  boxes f32 [n,7] bottom-centre LiDAR boxes, scores f32 [n], labels int32 [n].  One workgroup per class; outputs
 *   out_idx int32 / out_score f32 [num_classes][n] = indices into the input and decayed scores in selection order, out_cnt [num_classes].
 * u3d_box_merge: the KITTI configs' `box_merging` (ref: core/bbox/bbox_merging.py:112-158 as called from uni3detr_head.py:881-891).
 *   boxes f32 [n,7] / labels sorted by DESCENDING score (the caller sorts); keep[i] = survives the greedy sweep, merged [n,7] = for
 *   kept boxes the per-coordinate median over itself and the same-class boxes it absorbed (overlap > thr), else a copy.
 * ---------------------------------------------------------------------------------------------- */
int32_t u3d_soft_nms(const float* boxes, const float* scores, const int32_t* labels, int32_t n, int32_t num_classes, float sigma,
                     float prune, int32_t* out_idx, float* out_score, int32_t* out_cnt, u3d_stream s);
int64_t u3d_box_merge_workspace(int32_t n);
int32_t u3d_box_merge(const float* boxes, const int32_t* labels, int32_t n, float thr, float* merged, uint8_t* keep, void* workspace,
                      int64_t workspace_bytes, u3d_stream s);

/* ------------------------------------------------------------------------------------------------
 * Fused decoder layer (bf16 MFMA, f32 accumulation / residual stream / LayerNorm statistics / softmax statistics).
 * One call = one Uni3DETRTransformerDecoder layer over ALL query groups of all scenes, plus everything the decoder loop and the head
 * hang on that layer's state (ref: models/utils/uni3detr_transformer.py:145-212 decoder loop, :33-65 sine embedding, :271-360
 * UniCrossAtten; mmcv BaseTransformerLayer / MultiheadAttention / FFN as configured in the "transformer" section of the shipped configs;
 * models/dense_heads/uni3detr_head.py:367-387 cls / reg / iou branches):
 *
 *   pos   = ref_point_head(sine(sigmoid(ref))) [* query_scale(x) for layers > 0]
 *   x1    = LN1(x + dropout(out_proj(MHA(q = k = x + pos, v = x))))            attention inside each group of nq queries
 *   x2    = LN2(x1 + dropout(output_proj(trilinear(value, ref) * sigmoid(attention_weights(x1 + pos)))) + position_encoder(ref))
 *   x3    = LN3(x2 + dropout(W2 dropout(relu(W1 x2))))
 *   reg   = reg_branch(x3), cls = cls_branch(x3), iou = iou_branch(x3)
 *
 * Rows are [B, G*nq] (scene-major), M = B*G*nq; embed dim 256, 8 heads, FFN 512, num_points 1 (every shipped config).
 * Launches: forward 3 (row-chain kernel | attention | row-chain kernel), backward 4; the row-chain kernels keep a 32-row block of
 * the layer state in LDS through the whole chain of GEMMs / LayerNorms and stream the bf16 weights from L2 straight into MFMA
 * fragments.  Weight / bias gradients are NOT computed here: the backward leaves every linear's dY (bf16) and every LayerNorm's
 * per-workgroup (dgamma, dbeta) partial sums in the gradient workspace; the caller batches dW = dY^T X over layers
 * (u3d_wgrad_batched_bf16 / u3d_skinny_wgrad_bf16 / u3d_colsum_batched).  Slot offsets: u3d_decoder_layer_slots().
 * Dropout: counter-based (hash of seed, layer, site, element index); the backward regenerates the masks from the same seed.
 * ---------------------------------------------------------------------------------------------- */
enum {  /* wide linears: w = [N(pad)][K] (nn.Linear's matrix) and wt = [K][N(pad)] (dgrad), both in u3d_wpack's fragment order; b f32 [N] */
  U3D_DL_RPH0, U3D_DL_RPH1, U3D_DL_RPH2,      /* ref_point_head 384->256->256->256 */
  U3D_DL_QS0, U3D_DL_QS1, U3D_DL_QS2,         /* query_scale 256->256->256->256 (layers > 0) */
  U3D_DL_INQK, U3D_DL_INV,                    /* in_proj rows [0,512) and [512,768) */
  U3D_DL_OUTP,                                /* attn.out_proj */
  U3D_DL_OPROJ,                               /* UniCrossAtten.output_proj */
  U3D_DL_PE1,                                 /* position_encoder[3] */
  U3D_DL_FFN0, U3D_DL_FFN1,                   /* 256->512, 512->256 */
  U3D_DL_REG0, U3D_DL_REG1, U3D_DL_REG2,      /* REG2 / CLS2 / IOU2: N <= 32, w padded to 64 rows, wt = ROW-MAJOR [K][32] (t_plain) */
  U3D_DL_CLS0, U3D_DL_CLS1, U3D_DL_CLS2,
  U3D_DL_IOU0, U3D_DL_IOU1, U3D_DL_IOU2,
  U3D_DL_NLIN
};
enum { U3D_DLN_1, U3D_DLN_2, U3D_DLN_3, U3D_DLN_PE0, U3D_DLN_PE1, U3D_DLN_C1, U3D_DLN_C2, U3D_DL_NLN };
typedef struct u3d_declayer_params {
  const void* w[U3D_DL_NLIN];
  const void* wt[U3D_DL_NLIN];
  const float* b[U3D_DL_NLIN];
  const float* ln_g[U3D_DL_NLN];
  const float* ln_b[U3D_DL_NLN];
  const float* attw_w; const float* attw_b;   /* attention_weights: f32 [256], [1] */
  const float* pe0_w;  const float* pe0_b;    /* position_encoder[0]: f32 [256,3], [256] */
  const float* dim_t;                         /* f32 [128]: T^(2*(f/2)/128) */
} u3d_declayer_params;
typedef struct u3d_declayer_dims {
  int32_t m;            /* rows = batch * queries_per_scene */
  int32_t nq;           /* queries per attention group */
  int32_t qps;          /* queries per scene (groups * nq) */
  int32_t batch, dz, dy, dx;   /* value volume: rows [(b*dz+z)*dy+y)*dx+x][256] bf16 */
  int32_t ncls, code;   /* widths of the cls / reg outputs (<= 32) */
  int32_t has_qs;       /* 0: first layer (pos = ref_point_head output) */
  int32_t need_dref;    /* backward: also produce the gradient w.r.t. the reference-point logits */
  int32_t layer;        /* dropout stream id */
  float p_attn, p_drop; /* dropout probabilities: attention weights | the four residual / FFN sites (0 = off) */
  float ln_eps;
  int32_t dtype;        /* U3D_BF16: bf16 activations / weights / slots, v_mfma_f32_16x16x32_bf16 (throughput mode);
                           U3D_F32: f32 everywhere, exact v_mfma_f32_16x16x4_f32 (parity mode) - the SAME kernels, instantiated on the
                           element type; every `void*` operand / slot below then holds f32 and xc may alias x */
  int32_t dvalue_bf16;  /* backward, U3D_BF16 only: 1 = dvalue is a bf16 [rows][256] accumulator (packed bf16 atomics: two channels per
                           atomic, no f32 volume to zero and to cast afterwards); 0 = f32 [rows][256] */
} u3d_declayer_dims;
/* forward-save slots (row matrices [m, cols]); u3d_decoder_layer_slots fills byte offsets (U3D_DS_COUNT + 1 entries, last = total) */
enum {
  U3D_DS_SINE, U3D_DS_RPH1, U3D_DS_RPH2, U3D_DS_RAW, U3D_DS_QS1, U3D_DS_QS2, U3D_DS_QS, U3D_DS_POS, U3D_DS_QKIN,
  U3D_DS_QK, U3D_DS_V, U3D_DS_LSE, U3D_DS_O, U3D_DS_U1, U3D_DS_MR, U3D_DS_QP, U3D_DS_SAMP, U3D_DS_GATED, U3D_DS_PEH0,
  U3D_DS_UPE1, U3D_DS_U2, U3D_DS_X2C, U3D_DS_FFH, U3D_DS_U3, U3D_DS_R1, U3D_DS_R2, U3D_DS_I1, U3D_DS_I2, U3D_DS_UC1,
  U3D_DS_C1, U3D_DS_UC2, U3D_DS_C2,
  U3D_DS_AMASK,   /* uint32 [m*8][10]: attention-dropout keep bits, row (group*8 + head)*nq + query, bit b of word w = key 32w + b; written by the
                     fused in-projection + attention launch (bf16, nq <= 320, p_attn > 0), read by the layer's backward */
  U3D_DS_COUNT
};
/* backward-workspace slots: dY of every linear (bf16), LayerNorm partials, intermediate f32 gradients */
enum {
  U3D_DG_CLSO, U3D_DG_C2U, U3D_DG_C1U, U3D_DG_IOUO, U3D_DG_I2, U3D_DG_I1, U3D_DG_REGO, U3D_DG_R2, U3D_DG_R1, U3D_DG_F,
  U3D_DG_FFH, U3D_DG_OUT, U3D_DG_UPE1, U3D_DG_P0, U3D_DG_WL, U3D_DG_O2, U3D_DG_DO, U3D_DG_DQK, U3D_DG_DV, U3D_DG_QS,
  U3D_DG_QS2, U3D_DG_QS1, U3D_DG_RAW, U3D_DG_RPH2, U3D_DG_RPH1, U3D_DG_LNP, U3D_DG_DU1, U3D_DG_DPOSA, U3D_DG_SINE,
  U3D_DG_COUNT
};
int32_t u3d_decoder_layer_slots(int32_t m, int32_t ncls, int32_t code, int64_t* save_off, int64_t* grad_off);      /* = _dt(.., U3D_BF16, ..) */
int32_t u3d_decoder_layer_slots_dt(int32_t m, int32_t ncls, int32_t code, int32_t dtype, int64_t* save_off, int64_t* grad_off);
int32_t u3d_decoder_layer_blocks(int32_t m);   /* workgroups of the row-chain kernels = rows of every U3D_DG_LNP partial matrix (bf16) */
int32_t u3d_decoder_layer_blocks_dt(int32_t m, int32_t dtype);   /* rows per workgroup: 32 (U3D_BF16) / 16 (U3D_F32) */
/* ROW PADDING: every row matrix this layer WRITES (x_out, xc_out, reg_out, cls_out, iou_out, dx, dref and all slots) must hold
 * u3d_decoder_layer_blocks_dt(m, dtype) * (32 | 16) rows; rows >= m receive values nobody reads.  Matrices it only READS (x, xc, ref, dx_out, dreg,
 * dcls, diou) have m rows.  (The row kernels contain no branch on the row index: see csrc/decoder_common.h.) */
/* x f32 [m,256] layer input, xc its bf16 copy, ref f32 [m,3] logits, value bf16 rows, rng: device uint64 seed.
 * Outputs: x_out f32 / xc_out bf16 [m,256], reg_out f32 [m,code], cls_out f32 [m,ncls], iou_out f32 [m]; save: the slot buffer. */
int32_t u3d_decoder_layer_fwd(const u3d_declayer_params* p, const u3d_declayer_dims* d, const float* x, const void* xc,
                              const float* ref, const void* value, const uint64_t* rng, float* x_out, void* xc_out,
                              float* reg_out, float* cls_out, float* iou_out, void* save, int64_t save_bytes, u3d_stream s);
/* Gradients in: dx_out f32 [m,256] (NULL = zero), dreg f32 [m,code], dcls f32 [m,ncls], diou f32 [m].
 * Out: dx f32 [m,256] (w.r.t. the layer input x), dvalue f32 rows (ACCUMULATED with atomics: caller zero-fills once per step),
 * dref f32 [m,3] (only when need_dref), and the gradient workspace `grad` (dY slots for the caller's weight-gradient pass). */
int32_t u3d_decoder_layer_bwd(const u3d_declayer_params* p, const u3d_declayer_dims* d, const float* x, const void* xc,
                              const float* ref, const void* value, const uint64_t* rng, const void* xc_out, const void* save,
                              const float* dx_out, const float* dreg, const float* dcls, const float* diou, float* dx,
                              void* dvalue, float* dref, void* grad, int64_t grad_bytes, u3d_stream s);
/* Self-attention over groups on its own (the middle launch of the layer): q,k rows of qk [m,512] (q | k), v [m,256], 8 heads x 32;
 * o bf16 [m,256], lse f32 [m,8].  Backward: dqk [m,512], dv [m,256] bf16. */
int32_t u3d_mha_fwd(const void* qk, const void* v, int32_t m, int32_t nq, float p_attn, int32_t layer, const uint64_t* rng, void* o,
                    float* lse, u3d_stream s);
int32_t u3d_mha_bwd(const void* qk, const void* v, const void* o, const void* d_o, const float* lse, int32_t m, int32_t nq,
                    float p_attn, int32_t layer, const uint64_t* rng, void* dqk, void* dv, u3d_stream s);
/* the same kernels on either element type (dtype U3D_BF16 | U3D_F32: all matrices then f32, exact-f32 MFMA) */
int32_t u3d_mha_fwd_dt(const void* qk, const void* v, int32_t m, int32_t nq, float p_attn, int32_t layer, const uint64_t* rng, void* o,
                       float* lse, int32_t dtype, u3d_stream s);
int32_t u3d_mha_bwd_dt(const void* qk, const void* v, const void* o, const void* d_o, const float* lse, int32_t m, int32_t nq,
                       float p_attn, int32_t layer, const uint64_t* rng, void* dqk, void* dv, int32_t dtype, u3d_stream s);
/* Refresh of the weight copies the fused layer reads: for each descriptor dst = bf16(src [n][k]) (rows n >= N zero up to n_pad) and
 * dst_t = its transpose [k][n_pad_t], both stored in MFMA FRAGMENT ORDER: block (16-row tile, 32-element k-step) = 64 x 16 bytes in
 * lane order (lane = kq * 16 + row, 8 consecutive k each), blocks of a tile consecutive - one wave load of a weight fragment is
 * 1 KiB contiguous.  descs in device memory; one launch for all linears of all layers. */
typedef struct u3d_wpack_desc {
  const float* src; void* dst; void* dst_t;
  int32_t n, k, n_pad, n_pad_t;   /* n_pad % 16 == 0, k % 32 == 0 (bf16) / % 16 (f32); packed transposes: k % 16 == 0, n_pad_t % 32 == 0 */
  int32_t t_plain;                /* 1: dst_t stays row-major [k][n_pad_t] (the <= 32-column final layers); 0: fragment order */
  int32_t reserved;
} u3d_wpack_desc;
int32_t u3d_wpack_bf16(const u3d_wpack_desc* descs_dev, int32_t count, int32_t max_elems, u3d_stream s);
/* dtype U3D_F32: the same copies in f32 (zero-padded rows / transposes of the f32 masters) for the parity-mode instantiation */
int32_t u3d_wpack(const u3d_wpack_desc* descs_dev, int32_t count, int32_t max_elems, int32_t dtype, u3d_stream s);
/* keep-mask of the layer's dropout sites as bytes (testing aid): site 0..3 = out_proj, output_proj, FFN hidden, FFN out over
 * [m, 256 | 512], linear element index (cols = 0); site 4 = attention weights over [m*8, nq] (row = (group*8 + head)*nq + query):
 * pass cols = nq (rows are padded to an even length in the generator's index space).  p is honoured to 2^-16. */
int32_t u3d_dropout_mask(const uint64_t* rng, int32_t layer, int32_t site, int64_t n, float p, int32_t cols, uint8_t* keep, u3d_stream s);

/* ------------------------------------------------------------------------------------------------
 * On-device training data path.  Replaces, for a packed batch that already sits in HBM, the DataLoader-worker
 * transforms of the shipped train pipelines (ref: projects/configs/uni3detr/uni3detr_sunrgbd.py:150-174: RandomFlip3D ->
 * GlobalRotScaleTrans -> PointsRangeFilter -> PointSample; the plugin's own UnifiedRandomFlip3D / UnifiedRotScaleTrans,
 * projects/mmdet3d_plugin/datasets/pipelines/transform_3d.py:325-589, apply the same geometry).  The random DRAWS stay with the
 * caller (host RNG, as in the reference): per-scene parameters arrive as f32 [batch][U3D_AUG_NPARAM] =
 * (flip_horizontal, flip_vertical, sin(angle), cos(angle), angle, scale, tx, ty, tz): flip -> rotate -> scale -> translate.  coord: 0 = Depth boxes/points (SUN RGB-D, ScanNet),
 * 1 = LiDAR (KITTI, nuScenes) - it selects which axis a horizontal / vertical flip mirrors.
 * ------------------------------------------------------------------------------------------------ */
#define U3D_AUG_NPARAM 9
/* in place: points [n_total, feat] f32 (x, y, z, ...), scene b = rows scene_off[b] .. scene_off[b+1]); height_dim >= 3 scales that
 * attribute with the scene (shift_height=True), -1 = none */
int32_t u3d_points_augment(float* points, const int32_t* scene_off, int32_t batch, int32_t n_total, int32_t feat, const float* params,
                           int32_t coord, int32_t height_dim, u3d_stream s);
/* in place: boxes [n, box_dim] f32 (x, y, z, dx, dy, dz, yaw [, vx, vy]), box_dim 7 or 9, scene b = rows gt_off[b] .. gt_off[b+1]);
 * velocities flip, rotate and scale with the frame */
int32_t u3d_boxes_augment(float* boxes, const int32_t* gt_off, int32_t batch, int32_t n, int32_t box_dim, const float* params, int32_t coord,
                          u3d_stream s);
/* PointsRangeFilter: the points of scene b with lo < (x, y, z) < hi (strict), in their original order, compacted to the front of the
 * scene's own segment of `out` (same offsets as the input; out may alias points); count[b] = survivors.  range6 is a HOST array
 * (x0, y0, z0, x1, y1, z1); feat <= 8. */
int32_t u3d_points_range_filter(const float* points, const int32_t* scene_off, int32_t batch, int32_t feat, const float* range6, float* out,
                                int32_t* count, u3d_stream s);
/* PointSample: out [batch * num_points, feat], scene b = rows b*num_points ...: num_points rows of the first count[b] rows of the
 * scene's segment (count NULL = the whole segment) - distinct rows (a keyed pseudo-random permutation) when count[b] >= num_points,
 * uniform draws with replacement otherwise, zeros for an empty scene.  seed: device u64 (advance it per step); idx_out (nullable)
 * int32 [batch * num_points] = the chosen row within the scene (-1 for an empty scene). */
int32_t u3d_point_sample(const float* points, const int32_t* scene_off, const int32_t* count, int32_t batch, int32_t feat, int32_t num_points,
                         const uint64_t* seed, float* out, int32_t* idx_out, u3d_stream s);

/* ObjectRangeFilter (KITTI / nuScenes train pipelines; ref: projects/configs/uni3detr/uni3detr_kitti_3classes.py train_pipeline):
 * in place, per scene, order kept: boxes whose BEV centre lies strictly inside bev_range4 = (x0, y0, x1, y1) (HOST array) move to the
 * front of the scene's segment (labels int32, nullable, move with them), yaw wrapped into [-pi, pi); count[b] = survivors. */
int32_t u3d_boxes_range_filter(float* boxes, int32_t* labels, const int32_t* gt_off, int32_t batch, int32_t box_dim,
                               const float* bev_range4, int32_t* count, u3d_stream s);

#ifdef __cplusplus
}
#endif
#endif /* U3D_HIP_H_ */
