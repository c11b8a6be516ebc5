/*
 * ddp_hip.h - C ABI of the MI355X-native DiffDock-Pocket score-model hot path (libddp_hip.so).
 *
 * The reference (plainerman/DiffDock-Pocket) is 100 % Python and has NO native boundary of its own: its
 * "plugin API" for this path is the Python class TensorProductScoreModel
 * (models/all_atom_score_model.py:21-436).  This header is the C-ABI layer underneath our drop-in for that
 * class; each entry point names the reference code it replaces.  All pointers are DEVICE pointers owned by
 * the caller (fp32 / int32), all launches are stream-ordered on `stream` (a hipStream_t passed as void*),
 * no entry point allocates, frees or synchronises.  Return value: 0 on success, otherwise a hipError_t
 * (>0) or a negative DDP_E* code; ddp_last_error() returns a static description.
 *
 * INTEGRATION.md shows the ctypes binding a reference maintainer would add.
 */
#ifndef DDP_HIP_H
#define DDP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DDP_ABI_VERSION 17
#define DDP_EINVAL (-1)   /* bad argument (shape not supported, null pointer, ...) */
#define DDP_ELIMIT (-2)   /* exceeds a compiled-in limit (see DDP_MAX_*) */

#define DDP_MAX_TASKS 9   /* convs fused in one launch = the 9 convs of one layer (all_atom_score_model.py:118) */
#define DDP_MAX_BLOCKS 4  /* weight blocks 0e,1o,1e,0o of FasterTensorProduct (models/layers.py:26-31) */
#define DDP_MAX_SEGS 3
#define DDP_MAX_NS 64     /* scalar multiplicity supported by the fixed-size register tiles */
#define DDP_EDGE_TILE 64  /* edges per workgroup */

/* How one group of tensor-product basis features is built from the gathered node irreps a[.] and the edge's
 * spherical harmonics sh = [s0 | s1(3)]  (models/layers.py:40-53). */
enum ddp_feat_kind {
  DDP_F_SCALAR_S0 = 0, /* a[u] * s0                       scalar in -> scalar feature (:41, :53) */
  DDP_F_DOT = 1,       /* dot(a3[u], s1) / sqrt(3)        vector in -> scalar feature (:44, :50) */
  DDP_F_SCALAR_S1 = 2, /* a[u] * s1[c]                    scalar in -> vector feature (:42, :52) */
  DDP_F_VEC_S0 = 3,    /* a3[u][c] * s0                   vector in -> vector feature (:45, :49) */
  DDP_F_CROSS = 4      /* cross(a3[u], s1)[c] / sqrt(2)   vector in -> vector feature (:46, :48) */
};

typedef struct {
  int32_t kind;   /* ddp_feat_kind */
  int32_t in_off; /* first column of the input irreps block inside a node row */
  int32_t count;  /* multiplicity (number of features this segment contributes) */
} ddp_seg_t;

/* One weight block: out[e, n(,c)] = sum_u F[e, u(,c)] * w[e, u, n]  (models/layers.py:55-80). */
typedef struct {
  int32_t U;       /* number of basis features (rows of the per-edge weight block) */
  int32_t n;       /* output multiplicity */
  int32_t C;       /* 1 = scalar block (0e/0o), 3 = vector block (1o/1e) */
  int32_t out_off; /* first output column of this block in a message row */
  int32_t tile0;   /* first 32-column tile of this block in the packed fc2 weight */
  int32_t ntiles;  /* number of tiles */
  int32_t nsub;    /* n > 32: tiles per feature (n split in 32-column pieces); else 1 */
  int32_t ups;     /* n <= 32: features packed per tile = 32 / n; else 1 */
  int32_t nseg;
  ddp_seg_t seg[DDP_MAX_SEGS];
  /* Source-node factorisation of the block's scalar-input features (a0e*s0, a0e*s1[c], a0o*s1[c], a0o*s0):
   *   out[e, n(,c)] += (C == 1 ? s0 : s1[c]) * ( sum_k h[e,k] * G[src(e)][k][g_col0 + n] + Gb[src(e)][g_col0 + n] )
   * with G = task.g[g_slot], a per-SOURCE-NODE tensor produced by one plain GEMM per conv (host side).  U / seg[] then
   * describe only the remaining (vector-input) features, which stay on the per-edge MFMA path.  g_slot = -1: none. */
  int32_t g_slot;
  int32_t g_col0;
} ddp_block_t;

/* Work split of the 32-edge (factorised) kernel: wave w of a workgroup runs the tiles tile0 + j * tstride, j < count, of
 * block `block` for each of its nrole[w] segments; segments of one block whose tile sets cover the same output columns are
 * summed in LDS in `round` order (packing.conv32_roles). */
#define DDP_CONV32_WAVES 4
#define DDP_MAX_ROLE_SEGS 2
typedef struct {
  int32_t block, tile0, tstride, count, round;
} ddp_role_seg_t;

/* Static shape of a TensorProductConvLayer (models/score_model.py:84-107); shared by all tasks of a launch. */
typedef struct {
  int32_t f_in;  /* n_edge_features = width of edge_attr_ (3*ns, final_conv: 2*ns) */
  int32_t hid;   /* hidden_features of fc (= n_edge_features) */
  int32_t kp1;   /* f_in rounded up to 8 */
  int32_t hp;    /* hid rounded up to 8 */
  int32_t hs;    /* LDS row stride (floats) of the staged edge_attr_/h tiles */
  int32_t nct1;  /* 32-column tiles of fc layer 1 = ceil(hid/32) */
  int32_t d_out; /* message width */
  int32_t nblocks;
  int32_t fbuf_floats; /* LDS floats reserved for the per-block feature / reduction buffer */
  int32_t g_cols[2];   /* columns per (node, k) row of the two G arrays (0 = unused) */
  ddp_block_t blk[DDP_MAX_BLOCKS];
  /* factorised shapes only (g_cols != 0): */
  int32_t nrounds;     /* LDS summation rounds of the role segments (0 = no block has tiles) */
  int32_t nrole[DDP_CONV32_WAVES];
  ddp_role_seg_t role[DDP_CONV32_WAVES][DDP_MAX_ROLE_SEGS];
} ddp_conv_shape_t;

/* One conv = one edge set + one set of weights.  Edge arrays are in CSR order of the RECEIVING node
 * (edge_index[0], models/score_model.py:113,117). */
typedef struct {
  const float* x_src;   /* node features gathered by edge_index[1]          [n_src, ldx_src] */
  int32_t ldx_src;
  int32_t n_edges;
  const int32_t* src;   /* edge_index[1] per CSR position                    [E] */
  const int32_t* eid;   /* row of edge_sh / seg[0] for this CSR position     [E] */
  const float* sh;      /* edge spherical harmonics, 4 floats per edge        [E_canonical, 4] */
  /* edge_attr_ = cat(seg0[idx0], seg1[idx1], seg2[idx2]) (all_atom_score_model.py:273-312) */
  const float* seg_ptr[DDP_MAX_SEGS];
  const int32_t* seg_idx[DDP_MAX_SEGS];
  int32_t seg_ld[DDP_MAX_SEGS];
  int32_t seg_n[DDP_MAX_SEGS]; /* 0 = segment unused */
  const float* w1p;     /* fc.0 weight, packed by packing.py (tile-major, K-interleaved) */
  const float* b1p;
  const float* w2p;     /* fc.3 weight, packed, 1/sqrt(U) folded in */
  const float* b2p;
  float* msg;           /* per-edge messages, CSR order                       [E, d_out] */
  /* factorised convs only (shape.g_cols != 0): edges are then listed in SOURCE-node order (so that a workgroup streams
   * each G[j] once) and `pos` gives the message row (= position in the receiver-CSR order) of every listed edge */
  const float* g[2];    /* G[s]: [n_src, DDP_G_LD(hid, g_cols[s])] floats per node: first [hg/4, g_cols[s], 4] (hg = hid rounded
                           up to 4): G[j][k/4][c][k%4]; then the g_cols[s] values Gb[j][c] (the fc.3 bias part); then
                           padding to a 128-byte multiple.  One ddp_stage_a product writes the whole row.  16-byte aligned */
  const int32_t* pos;   /* [E] message row per listed edge; NULL = identity */
  const int32_t* n_edges_dev; /* optional, device memory: the actual edge count (see "Device-side counts"); n_edges is then the
                           capacity of the edge arrays and sizes the grid */
  /* Optional (ABI 12): fc.0 / fc.3 weights as fp16 hi/lo operand planes (packing.pack_tiles_h2; 16-byte aligned): when EVERY
   * task of a launch carries both and f_in = hid = 3 ns with ns in {60, 32, 24, 16}, the fc products run on
   * v_mfma_f32_32x32x16_f16 with both operands split as v = hi + lo / 2048 (three products per 16 k, fp32 accumulation: 22-bit
   * operands, the fp32 MFMA chain's error class, csrc/ddp_conv.hip; the 22 bits hold while both halves are normal fp16 numbers: below
   * |v| ~ 6.1e-5 the split resolves v to an ABSOLUTE 2^-36, i.e. |err| <= 2^-20 sum|x w| + K 2^-35 max|w| for small x -
   * tests/test_gpu_parity.py::test_stage_a_h2_error_floor_for_small_operands).  Layout per 32-column tile: 2 * NS fragments of 1 KiB (NS = k16 steps,
   * K zero-padded), fragment q = 2 * ks + plane (0 = hi, 1 = lo): [hh = 0..1][column j = 0..31][8 halves k = 16 ks + 8 hh + i].
   * b1p / b2p stay fp32.  NULL: the exact fp32 MFMA form (w1p / w2p). */
  const void* w1h;
  const void* w2h;
  /* Optional, host-visible (pinned) or device memory: set to 1 by the h2 form when a value it has to split (an edge_attr_ entry, an
   * h = relu(fc1) value) lies outside the fp16 range (|v| > 65504, or NaN): the result of that launch is then not to be trusted and
   * the caller should rerun with the fp32 form (w1h = w2h = NULL).  NULL: not reported. */
  int32_t* h2_range_flag;
  /* Optional (ABI 13; plane form of ABI 14), for ddp_conv_rows (128-edge, row-stationary workgroups; csrc/ddp_conv_rows.hip).
   * Its operands are UNIFIED fp16 hi/lo planes: V = v * S = hi + lo with hi = fp16(V), lo = fp16(V - hi) - both halves at the SAME
   * power-of-two scale S (DDP_ROWS_S* below), so that the three split products of a k-step accumulate into ONE register tile (the form
   * v = hi + lo / 2048 of w1h / w2h needs two and a multiply-add per element to join them).  22 significant bits while lo is a normal
   * fp16 number (|V| >= 0.125), an absolute 2^-25 / S below; |V| must stay <= 65504 (h2_range_flag; the host-side packers refuse).
   *  wsh   the SAME fc.0 / fc.3 weights as w1h / w2h, scaled by DDP_ROWS_SW, as ONE stream of 32-column tiles in the order the kernel
   *        walks them: the nct1 tiles of fc.0, then for every block and every 32-column part of its n output columns the part's tiles
   *        in feature order (packing.rows_stream).  Tile layout as w1h / w2h (2 NS fragments of 1 KiB), but the K index of the fc.3
   *        tiles is PERMUTED (DDP_ROWS_KPERM below): the kernel computes h = relu(fc1) as the transposed product, whose accumulator
   *        registers then ARE the A-operand fragments of the fc.3 products in that k order (no transpose through LDS).
   *  bsp   the tiles' bias words, fp32 [stream tiles][32], AT THE SCALE OF THE TILE'S ACCUMULATOR: fc.0 tiles DDP_ROWS_SW DDP_ROWS_SX b,
   *        fc.3 tiles DDP_ROWS_SH DDP_ROWS_SW b
   *  gh    DDP_ROWS_SG G[s] of a factorised conv as unified planes, per source node DDP_GH_LD(hid, gcp) floats: the
 *        32-column parts of the slot's blocks (in block order, a block's parts in order) one after the other, part p a CONTIGUOUS tile
 *        [k8 < ceil(hid/8)][c < wp][plane][8 halves] with wp = the part's width rounded up to 4 (k8 group g holds the permuted k's
 *        DDP_ROWS_KPERM(g >> 1, g & 1, i), i < 8; padding columns are zero; hi and lo words of a column side by side, so that stage A
 *        fills whole 128-byte lines), then DDP_ROWS_SH DDP_ROWS_SG Gb per padded column as fp32 [gcp = sum of the wp],
 *        then padding to 128 bytes; written by ddp_stage_a_gh from right-hand sides that carry the scales (packing.factor_weights_gh).
 * NULL: the task can only run through ddp_conv_messages. */
  const void* wsh;
  const float* bsp;
  const void* gh[2];
  /* Plane form of gh (ABI 16, form 1 redefined in ABI 17; all tasks of a launch carry the same):
   *  0  as above - hi and lo words of a column side by side, 32 bytes per 8 values, DDP_GH_LD floats per node;
   *  1  "fp16 + a continuation byte" (ABI 17): the unit (k8, c) of a part's tile is 24 bytes - 8 fp16 hi words, then 8 bytes - with
   *     V = hi + sign(hi) 2^E(hi) u8 / 2^18: hi = V rounded to 19 significant bits and TRUNCATED to fp16, u8 = the next 8 mantissa bits
   *     (19 significant bits, |err| <= 2^-19 |V| for |V| >= 2^-14; below, hi alone: absolute 2^-24); the units of a part [k8][c < wp] one
   *     after the other, the parts one after the other (unit g of the row at byte 24 g), then Gb per padded column c of the slot as fp32 at
   *     byte 24 (units + c / 6) + 4 (c % 6), padding to whole 384-byte pieces: DDP_GH3_LD floats per node.  Written by ddp_stage_a_gh3,
   *     read by both forms of the rows kernel.  (ABI 16 had an e4m3 byte there: 15 - 16 bits, 2.4e-4 on the scores - outside the path's 1e-4,
   *     profiles/r06_g3byte_parity.txt; the 19-bit form: 3.5e-6, profiles/r06_g19bit_precision.txt.) */
  int32_t gh_fmt;
  /* Operand images of wsh / gh / the kernel's h (ABI 16; all tasks of a launch carry the same):
   *  0  v_mfma_f32_32x32x16_f16 (csrc/ddp_conv_rows.hip): wsh fragments [ks][plane] of [hh][column 32][8 halves], k of the fc.3 tiles and of gh
   *     permuted by DDP_ROWS_KPERM;
   *  1  v_mfma_f32_16x16x32_f16 (csrc/ddp_conv_rows16.hip): a 32-column tile = 2 NS fragments of 1 KiB ordered [k32 step s][column tile ct]
   *     [plane], each [k group g < 4][column n < 16][8 halves] = plane(W)[column 16 ct + n][k = 32 s + 8 g + i]; the k order of the fc.3 tiles and
   *     of gh is the NATURAL one (gh: the same bytes per node as form 0, k8 group = k / 8), and fc.0's OUTPUT columns are placed inside every
   *     32-column stream tile t so that the transposed fc1 product leaves h in that k order: h column 32 t + 8 g + i sits at position
   *     16 (i / 4) + 4 g + i % 4 of the tile (DDP_ROWS16_POS; bias words in position order).  gh_fmt 0 or 1 in either form. */
  int32_t rows_form;
  /* ABI 17, rows_form 1 only.  1: the bias words of the fc.3 stream tiles ride in the tiles themselves - k row `hid` of every fc.3 tile holds the
   * tile's bias, fc.0's output column `hid` is the constant 1 (zero weights, bias word 1 at the tile's scale: h[hid] = relu(1)), bsp rows of the
   * fc.3 tiles are zero and not read; needs hid % 16 != 0 (a padding k).  The kernel then keeps only fc.0's bias words in LDS: shapes with
   * hundreds of stream tiles (the DIRECT convs: every feature is a stream tile) fit two workgroups per CU.  packing.rows_stream(bias_in_k). */
  int32_t rows_bias_k;
  /* ABI 17, rows_form 1 only.  A task may cover a RANGE of the shape's output segments (block, 32-column part; in the order the kernel walks
   * them): segments [rows_seg0, rows_seg1) - 0, 0 = all.  wsh / bsp then hold fc.0's tiles followed by the tiles of THOSE segments only,
   * rows_nts tiles in all (0 = the shape's count).  Several tasks with the same edges and msg and disjoint ranges = one conv spread over
   * several workgroups per 128 edges (the direct convs of small batches: 340 tiles per workgroup otherwise); every range recomputes fc1. */
  int32_t rows_seg0, rows_seg1, rows_nts;
} ddp_conv_task_t;
/* plane scales of ddp_conv_rows' operands: edge_attr_ (split in the kernel), fc.0 / fc.3 weights (task.wsh), h = relu(fc1) (split in
 * the kernel), G (task.gh).  Ranges |edge_attr_|, |h| < 4094, |w| < 255, |G| < 2047; absolute floors 2^-29, 2^-33, 2^-30. */
#define DDP_ROWS_SX 16
#define DDP_ROWS_SW 256
#define DDP_ROWS_SH 16
#define DDP_ROWS_SG 32
/* k index held by element i of the 8-k group (ks, hh) of an h / fc.3 / G operand fragment in ddp_conv_rows */
#define DDP_ROWS_KPERM(ks, hh, i) (32 * ((ks) >> 1) + ((8 * ((ks) & 1) + (i)) & 3) + 8 * ((8 * ((ks) & 1) + (i)) >> 2) + 4 * (hh))
/* position inside a 32-column fc.0 stream tile of the h column with in-tile index j = 8 g + i (rows_form 1) */
#define DDP_ROWS16_POS(j) (16 * (((j) & 7) >> 2) + 4 * ((j) >> 3) + ((j) & 3))
#define DDP_GH3_LD(hid, gcp) ((((((hid) + 7) / 8) * (gcp) + ((gcp) + 5) / 6 + 15) / 16) * 96)   /* floats per node, plane form 1: 6 per 8-column group, 16 groups per piece */
#define DDP_GH_LD(hid, gcp) ((((((hid) + 7) / 8) * 8 + 1) * (gcp) + 31) / 32 * 32)   /* floats per node of a G array in plane form, gcp = padded columns */
/* Fused fc -> tensor product -> per-edge message for up to 9 convs that share one shape.
 * All tasks of one call share `shape` (factorised and plain convs therefore go in separate calls).
 * Replaces: TensorProductConvLayer.forward up to (not including) the scatter
 * (models/score_model.py:108-114) + FasterTensorProduct.forward (models/layers.py:34-85) + the edge_attr_
 * concatenations (models/all_atom_score_model.py:273-312).  The [E, weight_numel] tensor never exists. */
int ddp_conv_messages(const ddp_conv_shape_t* shape, const ddp_conv_task_t* tasks, int ntasks, void* stream);

/* The same contract as ddp_conv_messages for FACTORISED shapes of the size classes ns = 60 / 32 (f_in = hid = 3 ns), through
 * 128-edge workgroups (4 waves, two workgroups per CU): every wave keeps h of ITS 32 edges as A-operand fragments in registers, the
 * weight tiles (task.wsh) are staged ONCE per workgroup through an LDS ring and read from there by all four waves, the factorised
 * features are G tiles (task.gh: one pass of the same tile product per run of edges with one source node), and a wave accumulates
 * all contributions to its output columns in registers and stores message rows directly: the fc.3 weights leave L2 once per 128
 * edges instead of once per 32 (the 32-edge kernel was bound by that traffic: 15 TB/s of L2 -> CU reads, L2 90 % busy).
 * Every task needs wsh, bsp and - for shapes with g_cols != 0 - gh (unified fp16 hi/lo planes, DDP_ROWS_S* above).  A value beyond
 * the planes' range (|edge_attr_| or |h| > 4094: h2_range_flag; |w| > 255: refused by packing.rows_stream; |G| > 2047: ddp_stage_a_gh's
 * range flag) is reported, never saturated.  Results: within fp32 rounding of ddp_conv_messages (another summation order),
 * deterministic. */
int ddp_conv_rows(const ddp_conv_shape_t* shape, const ddp_conv_task_t* tasks, int ntasks, void* stream);

/* Segmented mean over CSR rows + e3nn BatchNorm (eval) + residual accumulate:
 *   x[n, :d_out] (+)= sum_k ( mean_{p in rowptr_k[n]..rowptr_k[n+1]} msg_k[p, :] * bn_scale_k + bn_shift_k )
 * for up to 3 incoming convs k whose n_edges > 0 (an empty conv contributes exactly 0, models/score_model.py:109-111).
 * Replaces torch_scatter.scatter(reduce='mean') (score_model.py:117), e3nn BatchNorm (score_model.py:123-124)
 * and the pad+add residual (all_atom_score_model.py:315-324).  accumulate=0 overwrites x instead. */
typedef struct {
  const float* msg;
  const int32_t* rowptr; /* [n_nodes + 1] */
  const float* bn_scale; /* [d_out]  weight / sqrt(running_var + eps), broadcast over vector components */
  const float* bn_shift; /* [d_out]  bias - running_mean * scale on 0e channels, 0 elsewhere */
  int32_t n_edges;
  const int32_t* rowmap; /* optional [n_edges]: row of `msg` that holds the message of CSR position p (NULL: row p).  Lets
                            identical messages be stored once (score_model: layer-1 atom<-atom messages between atoms that no
                            ligand message has reached are the same in every sample of a sampling batch) */
  const int32_t* n_edges_dev; /* optional, device memory: the actual edge count (0 -> the conv contributes exactly 0) */
} ddp_reduce_src_t;
/* n_rep > 1: the update of node n - computed from 0 exactly as with accumulate = 0 - is ADDED to the rows n + g * rep_stride,
 * g < n_rep, of x (the layer-0 update of a receptor shared by all samples of a batch, added to every sample's copy). */
int ddp_segment_reduce(float* x, int ldx, int n_nodes, int d_out, const ddp_reduce_src_t* srcs, int nsrc,
                       int accumulate, int n_rep, int rep_stride, void* stream);

/* Edge featurisation: edge vector -> length -> Gaussian RBF -> 2-layer MLP, and spherical harmonics (lmax=1).
 *   vec = pos_b[ib[e]] - pos_a[ia[e]];  d = |vec|;  rbf_k = exp(coeff * (d - offset[k])^2)
 *   hidden = relu(pre[pre_idx[e]] + W1d^T rbf);  out[e] = W2^T hidden + b2;  sh[e] = [1, sqrt(3) vec/|vec|]
 * `pre` holds the part of the first Linear that does not depend on the distance (sigma embedding, bias): rows of ld_pre floats
 * (ddp_node_linear writes them per node); `pre2` (optional) is a second table whose row e is added for the edges e < n_pre2
 * (the bond-type columns for the bonds, which come first in the ligand edge list, all_atom_score_model.py:462-468).
 * n_edges_dev (optional, device memory): the actual edge count, read by the kernel; n_edges is then the capacity of the edge
 * arrays (see "Device-side counts" below).
 * Replaces GaussianSmearing (models/score_model.py:661-671), o3.spherical_harmonics and the *_edge_embedding MLPs
 * (models/all_atom_score_model.py:71-81,164-169,187-192,212-217 and the builders :444-636).
 * w1d: [k_rbf, 64] (zero padded columns), w2: [64, 64] (zero padded), b2: [64]; out: [E, ns]; sh: [E, 4]; vec_out
 * (optional, may be null): [E, 4] = (vec, d). */
int ddp_edge_featurize(const float* pos_a, const int32_t* ia, const float* pos_b, const int32_t* ib, int n_edges,
                       const int32_t* n_edges_dev, const float* offset, int k_rbf, float coeff, const float* pre,
                       const int32_t* pre_idx, int ld_pre, const float* pre2, int n_pre2, int ld_pre2, const float* w1d,
                       const float* w2, const float* b2, int ns, float* out, float* sh, void* stream);

/* The same for several edge sets in ONE launch (a forward has six: ligand, receptor, atom, ligand-receptor, ligand-atom,
 * atom-receptor; for small batches each is a handful of workgroups and a launch of its own ~5 - 14 us of latency).  Fields as
 * the arguments above; k_rbf a multiple of 8 in [8, 64] (the matrix-core form). */
#define DDP_MAX_FEATURIZE_JOBS 8
typedef struct {
  const float* pos_a;
  const int32_t* ia;
  const float* pos_b;
  const int32_t* ib;
  int32_t n_edges;
  const int32_t* n_edges_dev;
  const float* offset;
  int32_t k_rbf;
  float coeff;
  const float* pre;
  const int32_t* pre_idx;
  int32_t ld_pre;
  const float* pre2;
  int32_t n_pre2, ld_pre2;
  const float* w1d;
  const float* w2;
  const float* b2;
  int32_t ns;
  float* out;
  float* sh;
} ddp_featurize_job_t;
int ddp_edge_featurize_jobs(const ddp_featurize_job_t* jobs, int njobs, void* stream);

/* Torsion-head edge harmonics: the 1o block of FullTensorProduct(sh(edge), Y2(bond)) in closed form,
 *   t[e] = sqrt(3/2) * (3 (n.v) v - n),  n = unit(sh edge vector), v = unit(bond vector of bond ib[e])
 * written as [0, t] so the conv kernel reads it like an edge_sh row.
 * Replaces all_atom_score_model.py:394-395,418-419 (o3.spherical_harmonics("2e") + o3.FullTensorProduct). */
/* The same launch also writes the bonds' node attributes (optional, bond_attr == NULL skips it):
 *   bond_attr[b, 0..ns) = x[b0[b], 0..ns) + x[b1[b], 0..ns)       (all_atom_score_model.py:399,423: the scalar features of
 *   the bond's two atoms, gathered per edge by the conv kernel). */
int ddp_torsion_sh(const float* sh_edge, const float* bond_vec, const int32_t* bond_of_edge, int n_edges,
                   const int32_t* n_edges_dev, float* out, const float* x, int ldx, int ns, const int32_t* b0,
                   const int32_t* b1, int n_bonds, float* bond_attr, void* stream);

/* ---- Per-graph / per-bond scalar work of a forward (csrc/ddp_heads.hip).  In PyTorch these are 15 - 30 launches of a few
 * hundred bytes each; a dependent launch of a replayed hipGraph costs ~5 us whatever it does, so for small batches they were
 * the step.  All sums run in a fixed order (bitwise repeatable).
 *
 * ddp_step_prologue: everything that depends on the times and the poses only, ONE launch at the top of a forward.
 *   t[k], t_stride[k]     diffusion time of graph g for component k (tr, rot, tor, sc_tor): t[k][g * t_stride[k]] (stride 0: one time)
 *   sigma[k][g]           = sig_min[k]^(1-t) * sig_max[k]^t     (utils/diffusion_utils.py:22-34 t_to_sigma; NULL: skipped;
 *                         sig_max[k] <= 0 or t[k] == NULL: sigma[k] is an INPUT, made by the caller's own t_to_sigma)
 *   cut[g]                = sigma[0][g] * cut_mul + cut_add     (dynamic cross cutoff 3 sigma_tr + 20, all_atom_score_model.py:548-550)
 *   graph_emb[g, 0..sd)   sinusoidal embedding of t[0][g] (utils/diffusion_utils.py:73-84: [sin | cos] of emb_scale * t * freq[s]);
 *                         the read-out MLPs' input (:371)
 *   center[g]             mean of lig_pos over graph_ptr[g] .. graph_ptr[g+1] (:571-576)
 *   bonds[h]              h = 0 ligand torsion head, 1 side-chain head: mid[i] = (pos[b0[i]] + pos[b1[i]]) / 2 (:589-592,613-616),
 *                         vec[i] = pos[b1[i]] - pos[b0[i]] (:392,416)
 *   copy[h]               dst[0..n) = src[0..n) (int32): the bond rows in front of the ligand edge list (:462-468) */
typedef struct {
  const float* t[4];
  int32_t t_stride[4];
  float sig_min[4], sig_max[4];
  float* sigma[4];
  int32_t n_graphs;
  float* cut;
  float cut_mul, cut_add;
  float* graph_emb;
  int32_t sd;
  float emb_scale;
  const float* freq;
  const float* lig_pos;
  const int32_t* graph_ptr;
  float* center;
  struct {
    const float* pos;
    const int32_t* b0;
    const int32_t* b1;
    int32_t n;
    float* mid;
    float* vec;
  } bonds[2];
  struct {
    const int32_t* src;
    int32_t* dst;
    int32_t n;
  } copy[2];
} ddp_prologue_args_t;
int ddp_step_prologue(const ddp_prologue_args_t* args, void* stream);

/* ddp_trrot_head: translation / rotation read-out (all_atom_score_model.py:362-384) from the reduced final conv gp[B, >= 12]
 * = [tr 1o | rot 1o | tr 1e | rot 1e]:  v = 1o + 1e halves,  out = v / |v| * MLP([|v|, graph_emb]),  MLP = Linear(1 + sd, ns) ->
 * ReLU -> Linear(ns, 1) (index 0: tr_final_layer, 1: rot_final_layer; w1 [ns, 1 + sd] and w2 [ns] as nn.Linear stores them).
 * sigma[0] != NULL: tr / sigma_tr;  sigma[1] != NULL: rot * so3_table[clamp(round((log10(sigma_rot) - so3_lo) / so3_span * so3_n),
 * 0, so3_n - 1)] (utils/so3.py:85-89 score_norm). */
typedef struct {
  const float* gp;
  int32_t ld_gp, n_graphs, ns, sd;
  const float* graph_emb;
  const float* w1[2];
  const float* b1[2];
  const float* w2[2];
  const float* b2[2];
  const float* sigma[2];
  const float* so3_table;
  int32_t so3_n;
  float so3_lo, so3_span;
  float* out[2];
} ddp_trrot_args_t;
int ddp_trrot_head(const ddp_trrot_args_t* args, void* stream);

/* ddp_tor_head: torsion read-out (all_atom_score_model.py:400-410,424-434) per rotatable bond from the reduced bond conv
 * h[T, >= 2 ns]:  out[b] = Linear(ns, 1, no bias)(tanh(Linear(2 ns, ns, no bias)(h[b]))), and with sigma != NULL
 * * sqrt(torus_table[round(clamp((ln(sigma[graph_of_bond[b]] / pi) - torus_lo) / torus_span * torus_n, 0, torus_n))])
 * (utils/torus.py:78-82 score_norm; the table has torus_n + 1 entries). */
typedef struct {
  const float* h;
  int32_t ld_h, n_bonds, ns;
  const float* w1;
  const float* w2;
  const float* sigma;
  const int32_t* graph_of_bond;
  const float* torus_table;
  int32_t torus_n;
  float torus_lo, torus_span;
  float* out;
} ddp_tor_args_t;
int ddp_tor_head(const ddp_tor_args_t* args, void* stream);

/* Stage A of the source-node factorisation (ddp_block_t::g_slot): the per-source-node tensors consumed through
 * ddp_conv_task_t::g, for all (conv, G slot) pairs that read one node-feature array x:
 *   out[b][j * ldo + n] = sum_{u < k} x[j * ldx + offs[b] + u] * w[b][u, n],     b < nbatch, j < nrows, n < ncols
 * (offs[b] = first scalar channel of the slot, host array; w = the fc.3 weight rows of the slot's scalar-input features
 * regrouped per (k, column) by the host - models/score_model.py:100-105 and models/layers.py:41,42,52,53 contracted
 * over the input channel u per NODE instead of per edge).  w is [nbatch][k][ncols] contiguous, out [nbatch][nrows][ldo]
 * (ldo >= ncols; the columns [ncols, ldo) are not written); k even, <= 64; out 8-byte aligned.
 * For G use ldo = DDP_G_LD(hid, g_cols): rows that start on a 128-byte boundary are written (and later fetched) at twice
 * the rate of unaligned ones. */
#define DDP_MAX_GEMM_BATCH 16
#define DDP_G_LD(hid, gcols) ((((((hid) + 3) / 4) * 4 + 1) * (gcols) + 31) / 32 * 32)   /* floats per node of a G array */
/* rows (optional, device int32 [nrows]): only the listed node rows are computed - x row rows[i] -> row rows[i] of out[b], which
 * has out_rows rows per batch slice; the other rows of out are left as they are - and nrows_dev (optional, device) holds the
 * length of the list (nrows = its capacity).  rows == NULL: out_rows is ignored (= nrows).
 * w_bf16x3 (optional): the same weights split into three bfloat16 terms w = hi + mid + lo (hi = bf16(w), mid = bf16(w - hi),
 * lo = bf16(w - hi - mid)), laid out [nbatch][3 planes][kp / 16][2][ncols][8] with element (b, p, s, h, c, j) = plane p of
 * w[b][16 s + 8 h + j, c] (zero for k >= the product's k; kp = k rounded up to 16), 16-byte aligned.  When given (and k is 60 or
 * 32, ncols >= 1024, out 16-byte aligned with ldo % 4 == 0) the product runs on v_mfma_f32_32x32x16_bf16 with x split the
 * same way in the kernel and the six largest cross terms accumulated in fp32: error <= ~2^-22 sum_u |x w|, i.e. fp32 class,
 * not bitwise the fp32-MFMA form, at 1/2.7 of its matrix time (csrc/ddp_gemm.hip). */
int ddp_stage_a(const float* x, int ldx, int nrows, const int32_t* rows, const int32_t* nrows_dev, int out_rows, const int32_t* offs,
                int nbatch, const float* w, const void* w_bf16x3, int k, int ncols, float* out, int ldo, void* stream);

/* ddp_stage_a with the weights ALSO given as fp16 hi/lo operand planes, w_h2 = [nbatch][plane 0 = hi, 1 = lo][ceil(k / 16)][2][ncols][8]
 * halves (packing.split_h2: v = hi + lo / 2048, K zero-padded to a multiple of 16; 16-byte aligned).  For wide products (ncols % 4
 * == 0, ncols >= 1024, out 16-byte aligned with ldo % 4 == 0) and k in {60, 32, 24, 16} the product runs on v_mfma_f32_32x32x16_f16
 * with x split the same way in the kernel, three products per 16 k, fp32 accumulation (the conv kernels' h2 form: 22-bit operands,
 * error <= 2^-20 sum|x w|); other shapes run the exact fp32 forms on `w`.  ABI 12. */
int ddp_stage_a_h2(const float* x, int ldx, int nrows, const int32_t* rows, const int32_t* nrows_dev, int out_rows, const int32_t* offs,
                   int nbatch, const float* w, const void* w_h2, int k, int ncols, float* out, int ldo, int32_t* range_flag, void* stream);
/* ... with every output row written in the plane form of ddp_conv_task_t::gh.  The host orders the product's columns [part][k8][c][8]
 * (packing.factor_weights_gh) and passes, per batch entry and per group of 8 columns, the float offsets inside the row of the group's two
 * 16-byte pieces: dest[b][g][0..1], device int32 [nbatch][ncols / 8][2]; bit 0 of dest[b][g][0] marks a plane group: the 8 k's of one
 * k8 group of one G column, whose values V leave as UNIFIED planes (ddp_conv_task_t, ABI 14): 8 fp16 hi words fp16(V) at
 * dest[..][0] & ~3 and 8 fp16 lo words fp16(V - hi) at dest[..][1] - the same bytes as the fp32 form, the same number of stores; the
 * plane scale DDP_ROWS_SG rides in the right-hand side's G columns (V = DDP_ROWS_SG G), |V| > 65504 raises range_flag.  The other groups
 * are 8 fp32 columns stored as they are (Gb, padding).
 * Only on the h2 path (w_h2 given, k in {60, 32, 24, 16}, ncols % 32 == 0).  ABI 13. */
int ddp_stage_a_gh(const float* x, int ldx, int nrows, const int32_t* rows, const int32_t* nrows_dev, int out_rows, const int32_t* offs,
                   int nbatch, const float* w, const void* w_h2, int k, int ncols, float* out, int ldo, int32_t* range_flag,
                   const int32_t* dest, void* stream);

/* ... in plane form 1 of ddp_conv_task_t::gh (gh_fmt = 1: fp16 hi + a continuation byte, 24 bytes per 8-column group).  The product's
 * columns are ordered as for ddp_stage_a_gh; group g of the product (8 columns) leaves at byte 24 g of the row: a plane group as 8 hi words
 * + 8 bytes, an fp32 group (Gb, padding) as SIX values - product columns 0, 1, 4, 5, 2, 6 of the group in that order, columns 3 and 7 are
 * not stored (packing.factor_weights_gh(fmt = 1) places Gb accordingly).  Of dest only bit 0 of [b][g][0] is read (a plane group);
 * ncols % 128 == 0, ldo = 6 ncols / 8 = DDP_GH3_LD floats per row, a multiple of 32.  |V| > 65504 raises range_flag.  ABI 17. */
int ddp_stage_a_gh3(const float* x, int ldx, int nrows, const int32_t* rows, const int32_t* nrows_dev, int out_rows, const int32_t* offs,
                    int nbatch, const float* w, const void* w_h2, int k, int ncols, float* out, int ldo, int32_t* range_flag,
                    const int32_t* dest, void* stream);

/* ddp_stage_a_gh computes on unified planes too: x is split at DDP_GH_SX inside the kernel, w_h2 must hold the unified planes of
 * w / DDP_GH_SX (packing.split_h2(w, unified_scale = 1 / DDP_GH_SX)): the block's one accumulator is then the value that leaves as planes. */
/* (DDP_GH_SX = 2 since ABI 16: the right-hand side's planes are then those of DDP_ROWS_SG / 2 = 16 w - fc.3 x block-scale weights of
 * 1e-2 .. 5e-2 keep a NORMAL lo half (|V| >= 0.125; with 16 they were planes of 2 w: subnormal lo halves, an absolute 2^-26 on the
 * weight, ~19 bits) - and node features |x| >= 0.0625 keep theirs; ranges |x| < 32752, |16 w| < 65504; absolute floors 2^-26 on x,
 * 2^-29 on w.) */
#define DDP_GH_SX 2

/* Occupancy shaping (ABI 16).  Launches enqueued after this call: ddp_conv_rows asks for at least rows_min_lds_bytes of dynamic LDS - more
 * than 80 KiB leaves ONE of its 4-wave workgroups per CU (one 256-register wave per SIMD) instead of two, so that a kernel launched
 * beside it on another stream finds a free wave slot and LDS on every CU; the fp16 forms of ddp_stage_a_gh get stage_a_lds_pad_bytes of
 * dynamic LDS on top of their static 55 KiB (more than 25 KiB: one workgroup per CU).  (0, 0) restores the kernels' own occupancy.
 * Process-wide, read when a launch is enqueued (captured launches keep what they were captured with); results do not depend on it. */
int ddp_set_occupancy_shaping(int rows_min_lds_bytes, int stage_a_lds_pad_bytes);

/* The pose update between two score-model calls, for all samples of a batch in one launch:
 * modify_conformer(pos, tr_update, rot_update, torsion_updates) of utils/diffusion_utils.py:37-60 = rigid move about the
 * centre (rot_update is an axis-angle vector, utils/geometry.py:72-86), torsions in bond order (utils/torsion.py:68-94:
 * atoms with mask_rotate[j][i] != 0 turn by tor[s][j] about pos[bonds[j][0]] - pos[bonds[j][1]] through pos[bonds[j][1]]),
 * Kabsch re-alignment onto the rigid conformer (utils/geometry.py:209-243).  pos_in / pos_out [n_samples][n_atoms][3]
 * (may alias), tr / rot [n_samples][3], tor [n_samples][n_tor] or NULL (rigid move only). */
int ddp_pose_update(const float* pos_in, int n_samples, int n_atoms, const float* tr, const float* rot, const float* tor,
                    int n_tor, const int32_t* bonds, const uint8_t* mask_rotate, float* pos_out, void* stream);

/* Side-chain torsion update of all samples in one launch: modify_sidechains (utils/diffusion_utils.py:63-70,
 * utils/torsion.py:251-278): bond j (edge_idx[j] = (u, v)) turns the atoms subcomponents[mapping[j][0] .. mapping[j][1]) by
 * angles[s][j] about pos[u] - pos[v] through pos[v]; bonds are applied in list order.  pos_in / pos_out
 * [n_samples][n_atoms][3] (may alias). */
int ddp_sidechain_update(const float* pos_in, int n_samples, int n_atoms, const float* angles, int n_bonds,
                         const int32_t* edge_idx, const int32_t* subcomponents, const int32_t* mapping, float* pos_out,
                         void* stream);

/* Scores -> pose updates of one denoising step (utils/sampling.py:146-193, the SDE with low-temperature sampling or the ODE):
 *   out[k][i] = coef[2k] * score[k][i] + coef[2k+1] * z[k][i]      k = tr, rot, tor, sc_tor;  i < n[k];  z[k] == NULL: no noise
 * coef (8 floats) lives in DEVICE memory: the host writes the step's coefficients next to the noise, so a captured step (hipGraph)
 * has no kernel argument that changes from step to step. */
typedef struct {
  const float* score[4];
  const float* z[4];
  float* out[4];
  int32_t n[4];
} ddp_sde_args_t;
int ddp_sde_update(const float* coef, const ddp_sde_args_t* args, void* stream);

/* Neighbour search of the forward (torch_cluster radius / radius_graph / knn_graph, models/all_atom_score_model.py:457,
 * 524,545-564,607,627).  Graphs are contiguous node ranges x_ptr[g] .. x_ptr[g+1]; y_batch[q] is the graph of query q.
 *  radius: every x of the query's graph with |x - y|^2 < r^2 (strict).  More than max_neighbors matches: by default the
 *          FIRST max_neighbors in ascending x index are kept - what torch_cluster's CUDA kernel, the one the reference runs
 *          on a GPU, does (it scans x in index order and stops at the cap); with DDP_RADIUS_NEAREST in `flags` the nearest
 *          max_neighbors (and ties at that distance) instead.  DDP_RADIUS_DROP_SELF removes the pair (q, x == q) AFTER
 *          the cap (radius_graph: call with max_neighbors + 1); two passes: counts[q], then (after an exclusive prefix sum
 *          into offsets) the pairs (out_query[e], out_x[e]) query-major with ascending x.
 *  knn:    out_neighbors[q][0..k) = the k nearest other nodes of q's graph, nearest first, -1 where the graph is smaller. */
#define DDP_RADIUS_DROP_SELF 1
#define DDP_RADIUS_NEAREST 2
int ddp_radius_count(const float* x, const int32_t* x_ptr, const float* y, const int32_t* y_batch, int ny, float r,
                     int max_neighbors, int flags, int32_t* counts, void* stream);
int ddp_radius_fill(const float* x, const int32_t* x_ptr, const float* y, const int32_t* y_batch, int ny, float r,
                    int max_neighbors, int flags, const int32_t* offsets, int32_t* out_query, int32_t* out_x, void* stream);
int ddp_knn(const float* x, const int32_t* x_ptr, const int32_t* batch, int n, int k, int32_t* out_neighbors, void* stream);

/* ---- CSR / source-order views of an edge list (csrc/ddp_views.hip).  Replaces, per view, the torch.sort(stable) + index_add +
 * cumsum + gathers of the host-side graph code that stands for the reference's per-conv `edge_index` handling
 * (models/score_model.py:115-117: messages are gathered by edge_index[1] and mean-reduced over edge_index[0]).
 * Stable grouping of n_items items by key[i] in [0, n_keys):
 *   rowptr[n_keys + 1]   items of key k occupy positions rowptr[k] .. rowptr[k+1] of the outputs
 *   perm[n_items]        original index of the item at each position (ascending inside a key = stable sort)
 *   out_key / out0..2    key / payload pay0..2 of the item at each position (each optional: NULL skips it)
 *   scratch              n_keys + n_items int32 of workspace
 * perm == NULL: only rowptr is produced (the items are already grouped).
 * Results do not depend on the order in which the atomics inside land (bitwise deterministic). */
int ddp_group_by_key(const int32_t* key, int n_items, int n_keys, const int32_t* pay0, const int32_t* pay1, const int32_t* pay2,
                     int32_t* rowptr, int32_t* perm, int32_t* out_key, int32_t* out0, int32_t* out1, int32_t* out2,
                     int32_t* scratch, void* stream);

/* ---- Device-side counts.  The sizes of the pose-dependent lists of a denoising step (radius graphs, their views, the sub-lists
 * of the exact work eliminations) stay in device memory: where an entry point or a job takes `n` together with `n_dev`, a
 * non-NULL n_dev points to the actual count, the kernels read it (clamped to n), and n is the CAPACITY that sizes the launch
 * grid.  No entry point reads device memory on the host, so a whole step can be queued - or captured in a hipGraph - without
 * a host synchronisation.  The list primitives (csrc/ddp_lists.hip, ddp_graph.hip, ddp_views.hip) are batched: one call serves up
 * to DDP_MAX_LIST_JOBS independent jobs with a fixed number of launches. */
#define DDP_MAX_LIST_JOBS 12

/* Exclusive prefix sum over i < n of  value_i = (flag ? flag[i] != 0 : 1) * (val ? val[i] : rowptr ? rowptr[i+1] - rowptr[i] : 1):
 *   excl[i] = excl2[i] = base + sum_{i' < i} value_i',  excl[n] = *total = base + sum of all (each output optional);
 *   list[excl[i] - base ... ] : with 0 / 1 values (val == rowptr == NULL) list[excl[i]] = i for the flagged i, ascending. */
typedef struct {
  int32_t n;
  const int32_t* n_dev;
  const int32_t* flag;
  const int32_t* val;
  const int32_t* rowptr;
  int32_t base;
  int32_t* excl;
  int32_t* excl2;
  int32_t* list;
  int32_t* total;
} ddp_scan_job_t;
int ddp_scan_jobs(const ddp_scan_job_t* jobs, int njobs, void* stream);

/* mask[idx[i]] = 1 for i < n (idx == NULL: mask[i] = 1). */
typedef struct {
  const int32_t* idx;
  int32_t n;
  const int32_t* n_dev;
  int32_t* mask;
} ddp_mark_job_t;
int ddp_mark_jobs(const ddp_mark_job_t* jobs, int njobs, void* stream);

/* Compaction of whole CSR rows: the entries old_rowptr[r] .. old_rowptr[r+1] of in[k] move to new_rowptr[r] .. of out[k] for
 * every row with keep[r] != 0 (new_rowptr = ddp_scan_jobs over (flag = keep, rowptr = old_rowptr)). */
typedef struct {
  int32_t n_rows;
  const int32_t* keep;
  const int32_t* old_rowptr;
  const int32_t* new_rowptr;
  const int32_t* in[3];
  int32_t* out[3];
} ddp_rowcopy_job_t;
int ddp_rowcopy_jobs(const ddp_rowcopy_job_t* jobs, int njobs, void* stream);

/* Stable compaction of the items i < n with  mask_a[idx_a ? idx_a[i] : i] | mask_b[idx_b ? idx_b[i] : i]  (a NULL mask counts as
 * 0): out_idx[j] = i, out[k][j] = pay[k][i] + pay_add[k] in ascending i, *total = number kept.  block_count / block_off:
 * (n + 2047) / 2048 (+ 1 for block_off) int32 of workspace each. */
typedef struct {
  int32_t n;
  const int32_t* n_dev;
  const int32_t* mask_a;
  const int32_t* idx_a;
  const int32_t* mask_b;
  const int32_t* idx_b;
  int32_t* out_idx;
  const int32_t* pay[4];
  int32_t pay_add[4];
  int32_t* out[4];
  int32_t* total;
  int32_t* block_count;
  int32_t* block_off;
} ddp_select_job_t;
int ddp_select_jobs(const ddp_select_job_t* jobs, int njobs, void* stream);

/* out[i, :ncols] = x[idx[i], :ncols] for i < n. */
int ddp_gather_rows(const float* x, int ldx, const int32_t* idx, int n, const int32_t* n_dev, float* out, int ldo, int ncols,
                    void* stream);

/* The two index maps of the layer-1 clean-pair sharing (see csrc/ddp_lists.hip); n_edges = e0 * n_graphs. */
int ddp_clean_pair_maps(const int32_t* touched, const int32_t* recv, const int32_t* src, const int32_t* rowptr, int n_edges, int e0,
                        int n_graphs, int n0, int32_t* rowmap, int32_t* rows_v, void* stream);

/* Partial sharing across the poses of ONE complex whose side chains move (csrc/ddp_lists.hip): node (s, i) = copy i of sample s,
 * the samples' edge lists stored sample after sample with e0 edges each.  An atom is "off" in sample s when its position there
 * differs bitwise from its position in sample 0 (flag == NULL) or when flag[(s, i)] != 0.
 * ddp_flex_mark over an edge list (a[e], b[e]): mark[a[e]] = 1 if b[e] is off (a_too: or a[e] is); ref_list: mark[(s, a0)] = 1
 *   for every edge (a0, b0) of SAMPLE 0's list whose b0 is off in sample s; mark[(0, i)] = 1 for all i.  (rowptr in
 *   ddp_clean_pair_maps, optional: the receiver CSR's row pointers, for lists whose rows do not sit at the same positions in
 *   every sample.)
 * ddp_fallback_rowmap: rowmap[p] = new_rowptr[t] + (p - old_rowptr[recv[p]]), t = mark[recv[p]] ? recv[p] : recv[p] % n_recv_per_graph:
 *   the message row of position p of the full list when only the marked receivers' rows were kept (ddp_rowcopy_jobs). */
int ddp_flex_mark(const float* pos, const int32_t* flag, int n_b_per_graph, const int32_t* a, const int32_t* b, int n_edges, int e0,
                  int n_a_per_graph, int a_too, int ref_list, int32_t* mark, void* stream);
int ddp_fallback_rowmap(const int32_t* mark, const int32_t* recv, const int32_t* old_rowptr, const int32_t* new_rowptr, int n_edges,
                        int n_recv_per_graph, int32_t* rowmap, void* stream);

/* One neighbour search of ddp_radius_search_jobs = ddp_radius_count + prefix sum + ddp_radius_fill without the host in between:
 * counts[ny] and offsets[ny + 1] are outputs (offsets[q] = base + pairs of the queries before q), *total = base + all pairs,
 * pairs are written from index offsets[q] on and never at or behind `capacity`.  out_x == NULL: a count-only job.
 * graph_div (optional, [n_graphs]): both point sets are divided by graph_div[graph] before the distance is formed (the
 * reference's dynamic cross cutoff, models/all_atom_score_model.py:548-556). */
typedef struct {
  const float* x;
  const int32_t* x_ptr;
  const float* y;
  const int32_t* y_batch;
  int32_t ny;
  float r;
  int32_t max_neighbors;
  int32_t flags;
  const float* graph_div;
  int32_t* counts;
  int32_t* offsets;
  int32_t base;
  int32_t* total;
  int32_t* out_query;
  int32_t* out_x;
  int32_t capacity;
  int32_t* overflow;   /* optional: set to 1 when a pair had to be dropped because it lay at or behind `capacity` (may point to
                          pinned host memory: the host can then notice it without synchronising) */
} ddp_radius_job_t;
int ddp_radius_search_jobs(const ddp_radius_job_t* jobs, int njobs, void* stream);

/* One grouping job of ddp_group_by_key_jobs (arguments as ddp_group_by_key; n_items_dev: device-side item count; key_map
 * (optional): out_key[p] = key_map[key] instead of the key). */
typedef struct {
  const int32_t* key;
  int32_t n_items;
  const int32_t* n_items_dev;
  int32_t n_keys;
  const int32_t* pay[3];
  int32_t* rowptr;
  int32_t* perm;
  int32_t* out_key;
  int32_t* out[3];
  const int32_t* key_map;
  int32_t* scratch;
} ddp_group_job_t;
int ddp_group_by_key_jobs(const ddp_group_job_t* jobs, int njobs, void* stream);

/* ---- node encoders + sigma-dependent columns of the edge-embedding MLPs (csrc/ddp_node.hip), one launch for all jobs.
 * A job is a gathered-row Linear over n_rows nodes:
 *   out[n, :ncols] = (emb_mode == 2 ? emb_sum(n) : 0) + bias + [ emb_sum(n) | dense[0](n) | dense[1](n) | sigma_emb(n) ] @ w
 *   emb_sum(n)   = sum_f table[feat_off[f] + cat[n, f]]  in feature order, emb_dim wide   (models/score_model.py:75-76)
 *                  emb_mode 0: absent; 1: a K segment of the Linear (AtomEncoder, :78-80); 2: added to the result instead
 *                  (OldAtomEncoder: x_embedding += linear(scalars), :46-47; emb_dim >= ncols)
 *   dense[d](n)  = dense[d][n * ld_dense[d] .. + n_dense[d])   (the ESM block of the receptor, :80 / :50)
 *   sigma_emb(n) = sig_emb[n * ld_sig .. + sd) if sig_emb is given, otherwise the sinusoidal timestep embedding of
 *                  utils/diffusion_utils.py:73-84,106 evaluated in the kernel: a = scale * t[n * t_stride];
 *                  [sin(a * freq[k]) | cos(a * freq[k]) | 0 if sd is odd], k < sd / 2, freq = exp(-k ln(10000) / (sd/2 - 1))
 *                  handed over as a device array (t_stride = 0: one time for all rows)
 *   w            = [K, ncols] row-major (the nn.Linear weight transposed), K = the widths of the present segments in the
 *                  order above; bias [ncols] or NULL.  Columns [ncols, zero_to) of out are set to 0 (zero_to <= 0: none):
 *                  the node-feature rows are allocated at their final irreps width (all_atom_score_model.py:315-324 pads).
 * Also used for OldAtomEncoder's second stage (lm_embedding_layer over [x_embedding | ESM], :48-50: two dense parts). */
#define DDP_MAX_NODE_JOBS 8
#define DDP_MAX_NODE_CAT 16
typedef struct {
  int32_t n_rows;
  const int32_t* cat;      /* [n_rows, ld_cat] categorical features (n_cat used columns) */
  int32_t ld_cat, n_cat;
  const float* table;      /* all embedding tables of the encoder stacked row-wise, [sum dims, emb_dim] */
  int32_t feat_off[DDP_MAX_NODE_CAT]; /* first table row of feature f */
  int32_t emb_dim, emb_mode;
  const float* dense[2];
  int32_t ld_dense[2], n_dense[2];
  const float* t;          /* diffusion time per row (node_t['tr'], all_atom_score_model.py:453,495,520) */
  int32_t t_stride;
  float scale;             /* embedding_scale */
  const float* freq;       /* [sd / 2] */
  const float* sig_emb;    /* precomputed embedding instead of (t, scale, freq), or NULL */
  int32_t ld_sig, sd;
  float* sig_out;          /* optional: the sigma embedding rows are also written here (data[node type].node_sigma_emb) */
  int32_t ld_sig_out;
  const float* w;
  const float* bias;
  float* out;
  int32_t ld_out, ncols, zero_to;
  const float* add;     /* optional [n_rows, ld_add]: added to the result, out = add + (bias + [..] @ w) - the part of a Linear that */
  int32_t ld_add;       /* does not change between denoising steps (embedding tables, ESM block), computed once by another job   */
} ddp_node_job_t;
int ddp_node_linear(const ddp_node_job_t* jobs, int njobs, void* stream);

int ddp_abi_version(void);
const char* ddp_last_error(void);
/* 16 hex digits of the SHA-256 over the sources (every csrc .hip file, csrc/ddp_internal.h, include/ddp_hip.h) the library was built from */
const char* ddp_source_hash(void);

#ifdef __cplusplus
}
#endif
#endif /* DDP_HIP_H */
