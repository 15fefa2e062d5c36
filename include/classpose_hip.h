/*
 * classpose_hip.h -- C ABI of libclasspose_hip.so (MI355X / gfx950 only).
 *
 * Drop-in boundary for the `classpose-predict-wsi` tile path of
 * sohmandal/classpose.  The reference is pure Python; every entry point below
 * replaces one library call the reference makes on that path (cited per
 * function as /root/reference/<file>:<line>, or as the cellpose==4.0.8 /
 * segment-anything==1.0 symbol that call bottoms out in).  INTEGRATION.md shows
 * the ctypes binding a maintainer of the reference would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in `_host`;
 *   - `stream` is a hipStream_t passed as void* (0 = null stream);
 *   - return 0 on success, a negative CPX_E* code otherwise; nothing throws,
 *     nothing allocates: the caller supplies workspaces sized by the
 *     *_workspace_bytes() queries (graph-capture safe);
 *   - tiles are batched: leading dimension nT, then the reference's layout;
 *   - thread-safe for distinct streams + distinct workspaces: the library keeps no mutable
 *     process-global compute state (the element type travels in cpx_net_weights.dtype / the
 *     `dtype` arguments, the optional timing handle in cpx_net_weights.prof); a bf16, an fp16
 *     and an fp32 engine may run from different host threads.  The A/B and ablation switches
 *     of include/classpose_hip_debug.h are process-global test hooks and are NOT covered.
 */
#ifndef CLASSPOSE_HIP_H
#define CLASSPOSE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CPX_OK 0
#define CPX_EINVAL (-22)
#define CPX_ENOMEM (-12)
#define CPX_EHIP (-5)

/* element types of the network path (cpx_net_weights.dtype and the `dtype` arguments) */
#define CPX_DT_BF16 0
#define CPX_DT_F16 1
#define CPX_DT_F32 2   /* --precision fp32: exact-f32 MFMA (v_mfma_f32_32x32x2_f32), f32 activations */

/* Library / device identification. Returns the ABI version (this header: 3).
 * Version history: 2 = no process-global compute state (round 2); round 4 ADDED cpx_build_id, cpx_compute_masks_records and
 * cpx_prof_collect_launches without a bump (additive); 3 (round 5) = cpx_prof_collect fills CPX_PROF_N_KINDS = 7 entries (was 5) and
 * the post-processing workspace of cpx_postproc_workspace_bytes holds six table sets per tile -- callers compiled against version 2
 * must be rebuilt, and classpose_amd/_lib.py refuses a library whose version differs.                                          */
int cpx_abi_version(void);
/* Last HIP error string recorded by a failing call on this thread (host ptr). */
const char *cpx_last_error(void);
/* which sources this binary was built from: first 16 hex digits of the sha256 over classpose_amd/csrc/{*.hip,*.cpp,*.h,
 * Makefile} (sorted by name) + include/classpose_hip.h + include/classpose_hip_debug.h ("+debug" appended by the
 * -DCPX_DEBUG library).  classpose_amd._lib.source_build_id() recomputes it from the files on disk. */
const char *cpx_build_id(void);

/* ------------------------------------------------------------------------
 * a6  image normalisation
 * replaces cellpose.transforms.normalize_img as called from
 * ClassposeModel.eval, /root/reference/src/classpose/models.py:642-666
 * ---------------------------------------------------------------------- */
/* Per tile and channel: 1st / 99th percentile of the uint8 pixels with numpy's
 * float32 'linear' quantile arithmetic (prev-index + gamma supplied by the host
 * because they only depend on H*W), then x01 / (x99-x01) / mode.
 * tiles_u8 [nT][H][W][3];  stats [nT][3][4] = {x01, x99-x01, mode, unused},
 * mode 0: ptp==0 -> channel left untouched, 1: (x-x01)/(x99-x01), 2: zeros.
 * hist_ws: nT*3*256 uint32.                                               */
int cpx_normalize_stats_u8(const uint8_t *tiles_u8, int nT, int H, int W,
                           int lo_prev, float lo_gamma, int hi_prev, float hi_gamma,
                           float *stats, uint32_t *hist_ws, void *stream);
/* Applies the stats: out_f32 [nT][H][W][3] float32 (what normalize_img returns). */
int cpx_normalize_apply_u8(const uint8_t *tiles_u8, const float *stats, int nT, int H, int W,
                           float *out_f32, void *stream);

/* ------------------------------------------------------------------------
 * a3  residual rescale of a tile read from the pyramid level to the model mpp
 * replaces resize_tile_to_target_mpp -> cv2.resize(tile, (dw, dh), INTER_LINEAR),
 * /root/reference/src/classpose/entrypoints/predict_wsi.py:102-123 (and 462-483)
 * ---------------------------------------------------------------------- */
/* src_u8 [nT][sh][sw][3] -> dst_u8 [nT][dh][dw][3], OpenCV's 8-bit fixed-point
 * bilinear (11-bit weights from the float32 source coordinate, exact 2x2
 * decimation dispatched to the area average like cv::resize does).  Equal sizes
 * copy.  The caller computes dw = max(1, round(sw * resize_factor)) as the
 * reference does.                                                          */
int cpx_resize_linear_u8(const uint8_t *src_u8, int nT, int sh, int sw,
                         uint8_t *dst_u8, int dh, int dw, void *stream);

/* ------------------------------------------------------------------------
 * a7  sub-tiling and taper blending
 * replaces transforms.get_pad_yx + np.pad + transforms.make_tiles in
 * core.run_net, /root/reference/src/classpose/core.py:129-178, and
 * unaugment_tiles / unaugment_class_tiles / average_tiles + crop,
 * core.py:197-231 and transforms/transforms.py:4-21
 * ---------------------------------------------------------------------- */
typedef struct cpx_tiling {
    int H, W;            /* WSI tile size (pixels)                              */
    int ypad1, xpad1;    /* leading zero pad (get_pad_yx)                       */
    int Ly, Lx;          /* padded size                                         */
    int ny, nx;          /* sub-tile grid                                       */
    int bsize;           /* sub-tile size, 256                                  */
    int augment;         /* TTA flips by (j,i) parity                           */
    int ystart[16];      /* np.linspace(0, Ly-bsize, ny).astype(int)            */
    int xstart[16];
} cpx_tiling;

/* Normalised pixels (from stats) -> zero pad -> sub-tile -> flip -> 8x8 patch
 * rows (im2col of the patch-embed conv) in bf16.
 * patches [nT*ny*nx][1024][192], k = c*64 + i*8 + j.                       */
int cpx_make_subtiles(const uint8_t *tiles_u8, const float *stats, int nT,
                      const cpx_tiling *tiling_host, void *patches_bf16, void *stream);
/* Same with the element type of the patch rows chosen by `dtype` (CPX_DT_*): what
 * core._forward's X.to(dtype=net dtype) produces, core.py:61-63.             */
int cpx_make_patches(const uint8_t *tiles_u8, const float *stats, int nT,
                     const cpx_tiling *tiling_host, int dtype, void *patches, void *stream);
/* Same but float32 NCHW sub-tiles [nT*ny*nx][3][bsize][bsize] (what
 * make_tiles returns; used by parity tests and the fp32 debug path).        */
int cpx_make_subtiles_f32(const uint8_t *tiles_u8, const float *stats, int nT,
                          const cpx_tiling *tiling_host, float *subtiles, void *stream);
/* Head outputs of all sub-tiles -> per WSI tile dP/cellprob/logits.
 * head [nT*ny*nx][1024][ld_head] float32 token-major; column c*64+i*8+j of the
 * flow block (cols 0..191: dY,dX,cellprob) and of the class block (cols
 * 192..192+ncls*64) is pixel (8*ph+i, 8*pw+j) of channel c (pixel shuffle W2/W3,
 * vit_sam.py:181-188).  taper1d_host: 1-D cellpose taper (float64, bsize).
 * Outputs float32: dP [nT][2][H][W], cellprob [nT][H][W], logits [nT][ncls][H][W]. */
int cpx_blend_subtiles(const float *head, int ld_head, int ncls, int nT,
                       const cpx_tiling *tiling_host, const double *taper1d,
                       float *dP, float *cellprob, float *logits, void *stream);
/* Same from NCHW sub-tile outputs y [nS][3][b][b], y_class [nS][ncls][b][b]
 * (the arrays core.run_net holds before average_tiles).                      */
int cpx_blend_subtiles_nchw(const float *y, const float *y_class, int ncls, int nT,
                            const cpx_tiling *tiling_host, const double *taper1d,
                            float *dP, float *cellprob, float *logits, void *stream);

/* ------------------------------------------------------------------------
 * a9/a10  network forward (ClassTransformer.forward,
 * /root/reference/src/classpose/vit_sam.py:148-197 with flash_forward :15-65)
 * ---------------------------------------------------------------------- */
typedef struct cpx_block_weights {
    const float *ln1_w, *ln1_b;          /* [1024]                            */
    const void *qkv_w;  const float *qkv_b;   /* bf16 [3072][1024], [3072]    */
    const void *proj_w; const float *proj_b;  /* bf16 [1024][1024]            */
    const void *rel_h, *rel_w;           /* bf16 [64][64]: rows 0..62 = table
                                            interpolated to 63 rows, /scale   */
    const float *ln2_w, *ln2_b;
    const void *fc1_w;  const float *fc1_b;   /* bf16 [4096][1024]            */
    const void *fc2_w;  const float *fc2_b;   /* bf16 [1024][4096]            */
    /* LayerNorm folded into the consuming GEMM (cpx_net_weights.fuse_ln = 1): qkv_w / fc1_w
     * hold W * diag(gamma), qkv_b / fc1_b hold b + W.beta, and these are the row sums of the
     * folded (half-rounded) weights; ln*_w / ln*_b are then unused.                        */
    const float *qkv_colsum, *fc1_colsum;     /* [3072], [4096]               */
} cpx_block_weights;

/* Optional UNet semantic head (classpose/unet.py:121-196, chosen when the checkpoint has
 * out_class.encoder_blocks.* keys, predict_wsi.py:1393-1405).  The host flattens it into a list
 * of convolutions over token-major [sub-tile*h*w][C] half tensors; each runs as (im2col |
 * space-to-depth | nothing) + the MFMA GEMM (+ depth-to-space for the transposed convs).    */
typedef struct cpx_conv_op {
    int kind;              /* 0: conv3x3 pad 1, 1: conv2x2 stride 2, 2: convT2x2 stride 2      */
    int src_a, src_b;      /* tensor ids (src_b = -1: none; channel concat a|b); 0 = neck output */
    int dst;               /* tensor id produced (> 0)                                         */
    int cin_a, cin_b, cout;
    int h, w;              /* INPUT spatial size per sub-tile                                  */
    int relu;
    const void *weight;    /* half [Npad][Kpad]; k = tap*(cin_a+cin_b)+c (kinds 0,1), n = tap*cout+co (kind 2) */
    const float *bias;     /* [Npad]                                                           */
} cpx_conv_op;

typedef struct cpx_net_weights {
    int depth;              /* 24 for vit_l                                    */
    int ncls;               /* n_cell_classes (W3.shape[1])                    */
    int n_head_cols;        /* 192 + ncls*64                                   */
    int ld_head;            /* n_head_cols rounded up to 128                   */
    int dtype;              /* CPX_DT_BF16 / CPX_DT_F16 / CPX_DT_F32: element type of every "void *"
                               GEMM operand below (vectors are always float32)            */
    int fuse_ln;            /* 1: norm1 / norm2 are folded into qkv / mlp.lin1 (half types only) */
    const void *pe_w;       /* [1024][192]                                     */
    const float *pe_b;      /* [1024]                                          */
    const float *pos;       /* [1024 tokens][1024] float32                     */
    const cpx_block_weights *blocks;   /* HOST array of depth entries          */
    const void *neck0_w;    /* [256][1024]                                     */
    const float *neck_ln1_w, *neck_ln1_b;
    const void *neck2_w;    /* [256][9*256], k = (ky*3+kx)*256 + c             */
    const float *neck_ln2_w, *neck_ln2_b;
    const void *head_w;     /* [ld_head][256] rows: out (192) then out_class   */
    const float *head_b;    /* [ld_head]                                       */
    int n_unet_ops;         /* 0: 1x1-conv class head inside head_w; > 0: UNet head below      */
    const cpx_conv_op *unet_ops;   /* HOST array; the last op writes the ncls*64 class columns  */
    void *prof;             /* NULL, or a handle from cpx_prof_create: per-launch HIP-event timing   */
} cpx_net_weights;

/* ------------------------------------------------------------------------
 * a5  checkpoint -> kernel operands on the device (round 6, additive: no ABI bump)
 * replaces the host-side `net.to(torch.bfloat16 | torch.float16)` of ClassposeModel.__init__ as driven by
 * /root/reference/src/classpose/entrypoints/predict_wsi.py:659-727 (`resolve_precision`, models.py:37-69): the float32
 * checkpoint is uploaded as stored and rounded here.
 *   cpx_round_weights   dst[i] = round-to-nearest-even(src[i]) in `dtype`; keep_f32 = 0: stored as 2-byte elements (GEMM operands),
 *                       keep_f32 = 1: widened back to float32 (epilogue vectors).  dtype = CPX_DT_F32: a copy (keep_f32 must be 1).
 *                       src and dst 16-byte aligned; dst may not alias src unless keep_f32.
 *   cpx_fold_layernorm  LayerNorm folded into the Linear that consumes it (cpx_net_weights.fuse_ln): with every input first rounded to
 *                       `dtype`, w_folded[n][k] = round(w[n][k] * gamma[k]), b_folded[n] = b[n] + sum_k w[n][k] beta[k],
 *                       colsum[n] = sum_k w_folded[n][k]; both sums in float64 (fixed order), rounded once to float32.
 *                       w [N][K] float32, w_folded [N][K] 2-byte elements.                                                        */
int cpx_round_weights(const float *src, void *dst, long long n, int dtype, int keep_f32, void *stream);
int cpx_fold_layernorm(const float *w, const float *b, const float *gamma, const float *beta, int N, int K, int dtype,
                       void *w_folded, float *b_folded, float *colsum, void *stream);
/* The whole checkpoint in one call: job i streams its float32 HOST source(s) into the device staging area (stage_base + stage_off[k],
 * 16-byte aligned offsets, disjoint per job) with hipMemcpyAsync on `stream` and queues the kernel above behind the copy.
 *   CPX_WJ_ROUND_HALF / CPX_WJ_ROUND_F32  src_host[0] [n] -> dst[0] (cpx_round_weights, keep_f32 = 0 / 1)
 *   CPX_WJ_COPY_F32                       src_host[0] [n] -> dst[0] float32 as stored (no staging)
 *   CPX_WJ_FOLD_LN                        src_host = {w [n][K], b [n], gamma [K], beta [K]} -> dst = {w_folded, b_folded, colsum}
 * The sources may be pageable (the pages of a memory-mapped checkpoint); they must stay valid until the stream has run.       */
#define CPX_WJ_ROUND_HALF 0
#define CPX_WJ_ROUND_F32 1
#define CPX_WJ_COPY_F32 2
#define CPX_WJ_FOLD_LN 3
typedef struct cpx_weight_job {
    int op;                     /* CPX_WJ_*                                             */
    int dtype;                  /* CPX_DT_BF16 / CPX_DT_F16 (CPX_WJ_COPY_F32: ignored)  */
    long long n;                /* elements; CPX_WJ_FOLD_LN: rows N                     */
    int K;                      /* CPX_WJ_FOLD_LN: columns                              */
    int reserved;
    const void *src_host[4];
    size_t stage_off[4];
    void *dst[3];
} cpx_weight_job;
int cpx_weights_build(const cpx_weight_job *jobs_host, int n_jobs, void *stage_base, size_t stage_bytes, void *stream);

size_t cpx_net_workspace_bytes(int n_subtiles, int dtype);
/* extra bytes (appended to the network workspace) when w->n_unet_ops > 0 */
size_t cpx_unet_workspace_bytes(const cpx_conv_op *ops_host, int n_ops, int n_subtiles, int dtype);
int cpx_unet_head_forward(const cpx_conv_op *ops_host, int n_ops, const void *feat, int n_subtiles,
                          float *head, int ld_head, int col0, int dtype, void *workspace,
                          size_t workspace_bytes, void *stream);
/* patches [nS*1024][192] of w_host->dtype -> head [nS*1024][ld_head] float32
 * (the .float() of core._forward, core.py:67).                               */
/* row parts the MLP of every layer is run in by cpx_net_forward (1, or floor(n_subtiles / 16) for >= 32 sub-tiles: parts of 16 384 token rows, the last
 * one with the remainder, whose hidden activations stay in the Infinity Cache between mlp.lin1 and mlp.lin2) -- what a profiled mlp.lin1 / mlp.lin2
 * launch covers is n_subtiles / parts sub-tiles (exactly, when 16 divides n_subtiles)                                                           */
int cpx_net_mlp_parts(int n_subtiles, int dtype);
int cpx_net_forward(const cpx_net_weights *w_host, const void *patches, int n_subtiles,
                    float *head, void *workspace, size_t workspace_bytes, void *stream);

/* Per-launch timing of the dominant kernels of cpx_net_forward (bench.py's roofline lines): HIP
 * events recorded on the launch stream around every `stride`-th layer's kernels (the sampled layers
 * rotate by one per forward, so every layer is covered equally over `stride` forwards) of the kinds in
 * kinds_mask (bit 0 mlp.lin1, 1 attention, 2 qkv, 3 attn.proj, 4 mlp.lin2; once per forward, whatever the
 * stride: 5 the patch embedding, 6 ONE span over neck + head = the launches behind the last block).  The handle
 * is carried in cpx_net_weights.prof and belongs to one engine / host thread.  cpx_prof_collect (after a stream
 * sync) fills ms_sum[CPX_PROF_N_KINDS] / count[CPX_PROF_N_KINDS] per kind and resets the handle.  */
#define CPX_PROF_N_KINDS 7
int cpx_prof_create(int max_launches, int stride, unsigned kinds_mask, void **prof_out);
int cpx_prof_collect(void *prof, double *ms_sum, int *count);
/* the same launches one by one (launch order; ms[i], kind[i] for i < min(n, cap); *n_out = n; no reset: call it before
 * cpx_prof_collect) -- bench.py reports min / median / max per stage from these                                      */
int cpx_prof_collect_launches(void *prof, float *ms, int *kind, int cap, int *n_out);
void cpx_prof_destroy(void *prof);

/* ------------------------------------------------------------------------
 * a20  GrandQC tissue / artefact networks: UNet++ decoder on an EfficientNet-B0 encoder
 * replaces model.predict(x_tensor) + np.argmax in detect_tissue_wsi / detect_artefacts_wsi,
 * /root/reference/src/classpose/grandqc/wsi_tissue_detection.py:153-158 and
 * wsi_artefact_detection.py:190-195 (smp.UnetPlusPlus("timm-efficientnet-b0"), float32)
 * ---------------------------------------------------------------------- */
/* The host flattens the network into float32 NHWC operations over one workspace (BatchNorm
 * folded into weights/bias; byte offsets, (size_t)-1 = none).
 *  kind 0  dense k x k convolution as an implicit GEMM on v_mfma_f32_32x32x2_f32.  Input
 *          channels = source A (c_a channels, row stride ld_a floats, read through a nearest
 *          x2 upsample if up_a, scaled by gate[n][c] if gate) followed by source B (c_b, ld_b;
 *          the UNet++ concat buffer slice).  w [c_out rounded up to 32|64][k*k][pad16(c_a)+pad16(c_b)],
 *          bias [same rounding]; out = act(acc + bias) (+ res) -> dst with row stride ld_dst.
 *  kind 1  depthwise k x k convolution over source A; w [k*k][c_a], bias [c_a]; dst stride c_a.
 *  kind 2  squeeze-excite gate of source A: dst [nB][c_a] = sigmoid(w2 silu(w mean + bias) + bias2),
 *          w [c_red][c_a], w2 [c_a][c_red]; res = scratch of 16*nB*c_a floats.
 * h_in/w_in are the LOGICAL input grid (after the upsample).  act: 0 none, 1 ReLU, 2 SiLU.   */
typedef struct cpx_qc_op {
    int kind, k, stride, pad, act;
    int h_in, w_in, h_out, w_out;
    size_t src_a, src_b, gate, res, dst;
    int c_a, ld_a, up_a, c_b, ld_b, ld_res, c_out, ld_dst, c_red;
    const float *w, *bias, *w2, *bias2;
} cpx_qc_op;
/* patches_u8 [nB][H][W][3] (H, W multiples of 32) -> ImageNet preprocessing into
 * workspace+input_off ([nB][H][W][4] float, 4th channel 0) -> ops -> argmax of the first
 * n_classes floats of each ld_logits-wide row at workspace+logits_off -> class_map int8
 * [nB][H][W] (first maximum wins, like np.argmax); logits_out (optional) [nB][H][W][n_classes]. */
int cpx_qc_forward(const cpx_qc_op *ops_host, int n_ops, const uint8_t *patches_u8, int nB, int H, int W,
                   size_t input_off, size_t logits_off, int n_classes, int ld_logits,
                   int8_t *class_map, float *logits_out, void *workspace, size_t workspace_bytes,
                   void *stream);

/* Building blocks (exposed for parity tests / rooflines).
 * C[M][N] = A[M][K] * W[N][K]^T (+epilogue).  M%128==0, N%128==0, K%64==0.    */
#define CPX_EPI_BF16 0            /* out bf16 = acc (+bias if bias)            */
#define CPX_EPI_GELU_BF16 1       /* out bf16 = gelu_erf(acc + bias)           */
#define CPX_EPI_RESID_BF16 2      /* out bf16 = resid + acc + bias             */
#define CPX_EPI_F32 3             /* out f32  = acc (+bias)                    */
#define CPX_EPI_POS_BF16 4        /* out bf16 = acc + bias + pos[row%1024]     */
#define CPX_EPI_RELU_BF16 5
#define CPX_EPI_QKV_BF16 6        /* out bf16 [M][3072] = acc + bias; the V third (cols >= 2048) is
                                     ALSO/INSTEAD written transposed to aux = vT [M/1024][16][64][1024] */
int cpx_gemm_bf16(const void *A, const void *Wt, int M, int N, int K, int epilogue,
                  const float *bias, const void *resid_or_pos, void *out, int ld_out,
                  void *stream);
/* Same for any element type.  CPX_DT_F32: A, Wt, resid and out are float32 (out always f32, the
 * QKV epilogue is the plain one: the f32 attention reads V from the qkv rows), K % 16 == 0.  */
int cpx_gemm(int dtype, const void *A, const void *Wt, int M, int N, int K, int epilogue,
             const float *bias, const void *resid_or_pos, void *out, int ld_out, void *stream);

/* 3x3 convolution, padding 1, over 32 x 32-token images as an IMPLICIT GEMM (the neck's Conv2d(256, 256, 3, padding=1,
 * bias=False) of the SAM image encoder; cellpose vit_sam.Transformer): x [M = S*1024][C] token-major
 * (token = 32 y + x), Wt [N][9*C] with k = (3 ky + kx) * C + c, out [M][ld_out] bf16 / fp16.  No im2col buffer: the
 * LDS-DMA of every K tile reads the shifted token's chunk, or a zero chunk outside the image; the accumulation order is
 * that of cpx_gemm on the materialised [M][9*C] operand (bitwise equal).  M % 1024 == 0, N % 128 == 0, C % 64 == 0,
 * epilogue CPX_EPI_BF16 or CPX_EPI_RELU_BF16, dtype CPX_DT_BF16 / CPX_DT_F16.                                      */
int cpx_conv3x3(int dtype, const void *x, const void *Wt, int M, int N, int C, int epilogue, const float *bias,
                void *out, int ld_out, void *stream);
int cpx_gemm_uses_big_tile(int M, int N, int K, int epilogue);
/* Same with a LayerNorm over the K = 1024 input row folded in (consumer) and/or partial row
 * statistics of the output emitted (producer, RESID epilogue, N = 1024, 256^2-tile shapes):
 *   out = rstd[m] * (acc - mean[m] * ln_colsum[n]) + bias[n]     (then the epilogue)
 * ln_stats / stats_out: [M][4][2] float partial (sum, sum of squares); cpx_row_stats fills
 * slot 0 from a [rows][1024] half matrix.  vit_sam.py:175-176 (SAM Block: norm1/norm2). */
int cpx_gemm_ln(const void *A, const void *Wt, int M, int N, int K, int epilogue,
                const float *bias, const void *resid_or_pos, void *out, int ld_out,
                const float *ln_stats, const float *ln_colsum, float *stats_out, void *stream);
int cpx_row_stats(const void *x, int rows, float *stats, void *stream);
int cpx_layernorm_bf16(const void *x, const float *w, const float *b, int rows, int C,
                       float eps, void *out, void *stream);
/* qkv [nS*1024][3072] bf16 (q|k|v, head-major inside) -> attn out [nS*1024][1024]. */
int cpx_attention_relpos(const void *qkv, const void *rel_h, const void *rel_w, int n_subtiles,
                         void *vT_ws, void *out, void *stream);
/* any element type (CPX_DT_F32: float32 qkv / tables [64][64] / out; vT_ws unused)           */
int cpx_layernorm(int dtype, const void *x, const float *w, const float *b, int rows, int C,
                  float eps, void *out, void *stream);
int cpx_attention(int dtype, const void *qkv, const void *rel_h, const void *rel_w, int n_subtiles,
                  void *vT_ws, void *out, void *stream);

/* ------------------------------------------------------------------------
 * a11-a16  flows -> instance ids -> classes
 * ---------------------------------------------------------------------- */
size_t cpx_postproc_workspace_bytes(int nT, int H, int W);
int cpx_postproc_max_labels(int H, int W);
/* Diagnostic, read-only: number of kernel launches this THREAD has issued from the a11-a17 entry points since it first
 * called into the library (difference of two reads = launches of the calls in between).  bench.py reports it next to
 * the device / host times of the dispatch-bound post-processing stage.  Replaces nothing in the reference. */
unsigned long long cpx_postproc_launch_count(void);

/* cellpose.dynamics.follow_flows (steps_interp) as driven by
 * dynamics.compute_masks; call site models.py:149-159.
 * dP [nT][2][H][W] (dY,dX) raw network flows, cellprob [nT][H][W].
 * p_final [nT][H*W] int32: (y<<16)|x of the truncated end point, -1 where
 * cellprob <= threshold.  p_float (nullable) [nT][2][H*W] float end points. */
int cpx_follow_flows(const float *dP, const float *cellprob, int nT, int H, int W,
                     float cellprob_threshold, int niter, int32_t *p_final, float *p_float,
                     void *workspace, void *stream);
/* cellpose.dynamics.get_masks_torch: histogram, 5x5 NMS seeds (>10), 11x11
 * seeded growth (h>2, 5 iters), label gather, big-mask removal, renumber.
 * masks [nT][H*W] int32, nlabels [nT].                                      */
int cpx_get_masks(const int32_t *p_final, int nT, int H, int W, double max_size_fraction,
                  int32_t *masks, int32_t *nlabels, void *workspace, void *stream);
/* cellpose.dynamics.remove_bad_flow_masks (metrics.flow_error +
 * masks_to_flows_gpu, fp64 diffusion).  flow_errors (nullable) [nT][max_labels]. */
int cpx_remove_bad_flow_masks(int32_t *masks, const float *dP, int nT, int H, int W,
                              double threshold, double *flow_errors, void *workspace,
                              void *stream);
/* cellpose.utils.fill_holes_and_remove_small_masks.                          */
int cpx_fill_holes_and_remove_small_masks(int32_t *masks, int nT, int H, int W, int min_size,
                                          int32_t *nlabels, void *workspace, void *stream);
/* classpose.models.compute_class_masks, models.py:191-230.
 * logits [nT][ncls][H][W]; class_masks [nT][H*W] uint8.                      */
int cpx_compute_class_masks(const int32_t *masks, const float *logits, int nT, int ncls, int H,
                            int W, uint8_t *class_masks, void *workspace, void *stream);
/* classpose.metrics.pq.remove_border_instances, metrics/pq.py:65-92
 * (class_masks nullable = the (H,W) form; else the (H,W,2) form).            */
int cpx_remove_border_instances(int32_t *masks, uint8_t *class_masks, int nT, int H, int W,
                                void *workspace, void *stream);
/* The whole of dynamics.resize_and_compute_masks + compute_class_masks
 * (models.py:750-768): masks_u16 [nT][H*W] uint16, class_masks [nT][H*W] uint8. */
int cpx_compute_masks(const float *dP, const float *cellprob, const float *logits, int nT,
                      int ncls, int H, int W, float cellprob_threshold, double flow_threshold,
                      int niter, int min_size, double max_size_fraction, uint16_t *masks_u16,
                      uint8_t *class_masks, int32_t *nlabels, void *workspace, void *stream);


/* Compact per-instance records: what leaves the device in place of the pickled
 * (masks, class_masks) arrays of predict_wsi.py:757-763 / :595-652.          */
typedef struct cpx_record {
    int32_t tile;        /* index in the batch                                 */
    int32_t label;       /* instance id inside the tile                        */
    int32_t cls;         /* class of its first raster pixel (predict_wsi.py:634) */
    int32_t area;        /* pixel count                                        */
    int32_t y0, x0, y1, x1;   /* bbox, end exclusive                           */
    int64_t sum_y, sum_x;     /* pixel-coordinate sums (centroid = sum/area)   */
} cpx_record;
int cpx_instance_records(const uint16_t *masks_u16, const uint8_t *class_masks, int nT, int H,
                         int W, int max_records_per_tile, cpx_record *records,
                         int32_t *counts, void *workspace, void *stream);
/* The same chain with the per-cell records of cpx_instance_records produced by its last pass (records / rec_counts / max_rec
 * as there; both NULL: exactly cpx_compute_masks).  One launch sequence of 22 kernels per batch of tiles (the stage-wise
 * entry points above, called one after the other, need 38): one initialisation for all stages, relabelling rides on the next
 * stage's pixel pass, removals go through the rank table, the records leave with the final pass.  Bit-identical outputs.
 * Replaces: dynamics.resize_and_compute_masks + compute_class_masks (models.py:750-768) and the (masks, class_masks) ->
 * per-cell table step of PostProcessor.__call__ (predict_wsi.py:595-652) for one batch of tiles.                        */
int cpx_compute_masks_records(const float *dP, const float *cellprob, const float *logits, int nT,
                              int ncls, int H, int W, float cellprob_threshold,
                              double flow_threshold, int niter, int min_size,
                              double max_size_fraction, uint16_t *masks_u16,
                              uint8_t *class_masks, int32_t *nlabels, int max_rec, cpx_record *records,
                              int32_t *rec_counts, void *workspace, void *stream);

/* ------------------------------------------------------------------------
 * a17  polygonisation (HOST function: all pointers are host pointers)
 * replaces, per instance, cv2.findContours(cell_mask, RETR_EXTERNAL,
 * CHAIN_APPROX_SIMPLE)[0] + shapely.Polygon(...).is_valid/.centroid/.area/.length
 * of PostProcessor.__call__, /root/reference/src/classpose/entrypoints/predict_wsi.py:601-652.
 * ---------------------------------------------------------------------- */
typedef struct cpx_cell {
    double area, perimeter, cx, cy;  /* in level-0 slide pixels (after scale + offset)   */
    int32_t n_pts, offset;           /* vertices xy_pool[offset .. offset+n_pts) (ring not closed) */
    int32_t valid;                   /* >= 4 vertices and a simple (OGC-valid) ring       */
    int32_t cls;
} cpx_cell;
/* masks_host [H][W] uint16 of ONE tile; recs_host[n] its records; vertex i of a cell is
 * (x_px * scale + off_x, y_px * scale + off_y).  Returns #vertices written (>= 0) or < 0. */
int cpx_polygonize_host(const uint16_t *masks_host, int H, int W, const cpx_record *recs_host, int n,
                        double scale, double off_x, double off_y, double *xy_pool, int max_pts,
                        cpx_cell *cells);

/* f1: the same polygonisation on the DEVICE (one thread per instance, post-processing stream), so
 * that only vertex lists and cpx_cell rows leave the GPU instead of the 2 B/pixel id maps.
 * masks_u16 [nT][H][W]; records [nT][max_rec] + rec_counts [nT] as cpx_instance_records wrote them;
 * origins [nT][2] double (level-0 x, y of each tile); cells [nT][max_rec] (rows >= rec_counts[t]
 * untouched); vertex i of a cell = (x_px * scale + origin_x, y_px * scale + origin_y) at
 * xy_pool[2 * (offset + i)], offsets are an exclusive scan in (tile, record) order; n_pts_total [1]
 * receives the pool use (cells whose vertices would pass max_pts come back with n_pts = 0).  A contour visits a
 * pixel at most twice, so max_pts = nT * 2 * H * W can never overflow (what classpose_amd.engine allocates: 16 B per
 * vertex, 16.8 MB for 8 tiles of 256 px); only n_pts_total vertices need to leave the device.
 * Precondition: hole-free instances (what cpx_fill_holes_and_remove_small_masks produces); then the
 * outputs are bit-identical to cpx_polygonize_host.                                            */
size_t cpx_polygonize_workspace_bytes(int nT, int H, int W, int max_rec);
int cpx_polygonize_device(const uint16_t *masks_u16, const cpx_record *records, const int32_t *rec_counts,
                          int nT, int H, int W, int max_rec, double scale, const double *origins,
                          double *xy_pool, int max_pts, cpx_cell *cells, int32_t *n_pts_total,
                          void *workspace, void *stream);

/* ------------------------------------------------------------------------
 * f2  cross-tile de-duplication: the fixed-radius pair search
 * replaces KDTree(centers).query_pairs(max_dist) of deduplicate,
 * /root/reference/src/classpose/entrypoints/predict_wsi.py:923-927 (scipy.spatial.KDTree, p = 2)
 * ---------------------------------------------------------------------- */
/* centers_xy [n][2] double (x, y) = the rounded centroids of every cell of the slide.  The pair set
 * {(i, j), i < j : (xi-xj)^2 + (yi-yj)^2 <= max_dist^2} (double arithmetic, products and sum unfused: exactly
 * scipy's test) via a uniform grid of `cell` >= max_dist covering [x0, x0 + grid_w*cell) x [y0, y0 + grid_h*cell).
 * Two calls on the same workspace: pairs == NULL buckets + counts (-> *n_pairs, device int64); then with
 * pairs [max_pairs][2] int32 writes them sorted by i (deterministic).  The greedy grouping of :929-960 follows
 * the iteration order of a Python set and stays on the host (classpose_amd.geojson.dedup_from_pairs).   */
size_t cpx_dedup_pairs_workspace_bytes(int n_points, int grid_w, int grid_h);
int cpx_dedup_pairs(const double *centers_xy, int n, double x0, double y0, double cell, int grid_w, int grid_h,
                    double max_dist, int32_t *pairs, long long max_pairs, long long *n_pairs,
                    void *workspace, size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------
 * f3 / a19  the two GeoJSON FeatureCollections of a slide, streamed (host code, no HIP calls)
 * replaces json.dump({"type": "FeatureCollection", "features": [...]}) over the features of
 * to_geojson_polygon / apply_bounds_offset_to_feature / polygons_to_centroids,
 * /root/reference/src/classpose/entrypoints/predict_wsi.py:813-893,1336-1374,1772-1785.
 * Byte-identical to CPython's json encoder (", " and ": " separators, float.__repr__ numbers) except for the
 * random uuid4 ids.  cells [n_cells] rows of {double area, perimeter, cx, cy; int64 n_pts, cls} (48 B);
 * centroids_xy [n_cells][2] the centroids rounded to 2 decimals; ring i = xy_pool[2*offsets[i] .. 2*offsets[i+1])
 * (not closed; the writer repeats the first vertex); keep [n_keep] indices of the cells to write, in order;
 * class_json [n_class_json] the serialised {"name": ..., "color": [...]} object per value of cls;
 * bounds are subtracted from every coordinate when non-zero; n_threads <= 0 = one per core (max 16). */
int cpx_write_geojson(const char *contours_path, const char *centroids_path, const void *cells,
                      int64_t n_cells, const double *centroids_xy, const double *xy_pool,
                      const int64_t *offsets, const int64_t *keep, int64_t n_keep,
                      const char *const *class_json, int n_class_json, double bounds_x, double bounds_y,
                      int n_threads);

/* cv2.findContours(mask, RETR_CCOMP, CHAIN_APPROX_SIMPLE) of the GrandQC class maps
 * (/root/reference/src/classpose/grandqc/wsi_tissue_detection.py:209-213,
 * wsi_artefact_detection.py:262-265), host code.  mask [H][W] uint8, non-zero = foreground.
 * Contour c = xy_pool[2*offsets[c] .. 2*(offsets[c]+n_pts[c])) as (x, y) int32 pairs;
 * parent[c] = -1 for outer borders, else the index of the enclosing component's outer border
 * (hierarchy[0, c, 3] of RETR_CCOMP).  Discovery (raster) order.  Returns #contours or < 0. */
int cpx_find_contours_ccomp_host(const uint8_t *mask_host, int H, int W, int32_t *xy_pool, int max_pts,
                                 int32_t *offsets, int32_t *n_pts, int32_t *parent, int max_contours);


#ifdef __cplusplus
}
#endif
#endif
