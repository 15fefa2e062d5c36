/*
 * classpose_hip_debug.h -- A/B, ablation and diagnostic switches.
 *
 * NOT part of the product ABI (include/classpose_hip.h) and NOT in the product library: these symbols -- and the
 * non-production kernel variants they select -- exist only in libclasspose_hip_debug.so, the same sources built with
 * -DCPX_DEBUG (csrc/Makefile).  libclasspose_hip.so exports none of them; there each switch is a compile-time constant
 * at its production value.  They are process-global test hooks used by tools/*.py and a few tests to compare kernel
 * variants in one process (classpose_amd._lib.use_debug_library() / CLASSPOSE_HIP_DEBUG=1); not thread-safe.
 */
#ifndef CLASSPOSE_HIP_DEBUG_H
#define CLASSPOSE_HIP_DEBUG_H
#ifdef __cplusplus
extern "C" {
#endif
void cpx_gemm_set_variant(int glds);        /* 128^2 GEMM: 1 LDS-DMA staging (default), 0 register staging */
void cpx_gemm_set_big(int on);              /* 1 (default): 256^2 kernel when the shape allows          */
void cpx_gemm_set_persistent(int on);       /* 1 (default): persistent 256^2 kernel with next-tile prefetch under the epilogue */
void cpx_gemm_set_persistent_qkv(int on);   /* qkv projection on the persistent kernel with a balanced q|k / V^T tile list */
void cpx_gemm_set_l2_block(int on);         /* 1 (default): 8 x 4 super-tile order per XCD, N-sweep; 2: M-sweep; 0: row-major */
void cpx_gemm_set_epi4(int on);             /* 0 (default): 1 = quarter-tile epilogue of the persistent 256^2 kernel (experiment) */
void cpx_gemm_set_pingpong(int on);         /* 0 (default): 1 = 256 x 128 "ping-pong" kernel, two 4-wave workgroups per CU (bias / GELU / ReLU epilogues) */
void cpx_gemm_set_pingpong_opts(int persistent, int delay);   /* (1, 2) default: persistent grid; start offset of a CU's 2nd workgroup x ~8k cycles */
int cpx_gemm_pingpong_stamps(unsigned long long *host_out, size_t n_words);   /* cycle stamps written under cpx_gemm_set_dbg(32) */
int cpx_gemm_pingpong_occupancy(void);      /* workgroups of the ping-pong kernel admitted per CU (design point: 2) */
void cpx_gemm_set_reverse(int on);          /* 0 (default): mlp.lin2 walks M backwards when 1            */
void cpx_gemm_set_direct(int mode);         /* 1 (default): direct-store epilogue of the persistent 256^2 kernel (accumulators -> v_permlane16_swap -> 16-byte buffer stores, no LDS staging) for the GELU epilogue; 2: for every non-residual epilogue; 0: staged rows */
void cpx_gemm_set_balanced(int on);         /* 1 (default): balanced fragment-read schedule of the persistent 256^2 main loop for the bf16 residual + row-statistics epilogue (proj, mlp.lin2); 0: plain schedule */
int cpx_gemm4w(const void *A, const void *W, int M, int N, int K, const float *bias, void *out, int ld_out, void *stream);   /* the 256^2 GEMM tile with ONE wave per SIMD (4 waves x 128 x 128, AGPR accumulators), persistent, bias epilogue, bf16 (csrc/cpx_gemm4w.hip) */
void cpx_gemm_set_4w(int mask);             /* 1 (default): mlp.lin1 (bf16, folded LayerNorm + GELU) on the one-wave-per-SIMD kernel; +2: attn.proj / mlp.lin2 (residual + statistics) too (slower, not shipped); 0: everything on the 8-wave persistent kernel (same bits) */
void cpx_gemm4w_set_variant(int v);         /* experiment switches of cpx_gemm4w (0 default) */
void cpx_gemm_set_split(int on);            /* 0 (default): 1 = counted LDS waits inside the main-loop phases of the persistent 256^2 kernel (experiment) */
void cpx_gemm_set_dbg(int mask);            /* timing-only ablations of the 256^2 epilogue (0 default)   */
void cpx_net_set_mlp_parts(int on);         /* 1 (default): the MLP of a layer in row parts of 16 384 tokens (hidden activations stay in the Infinity Cache); 0: one launch pair */
void cpx_attention_set_xcd_order(int on);   /* 1 (default): (sub-tile, head) pairs pinned to one XCD     */
void cpx_attention_set_variant(int v);      /* 2: 4-wave, LDS-DMA ring + pipelined S; 0: 4-wave register ring; 1: 8-wave ping-pong; 3: one wave per SIMD, two query rows per wave (LDS ring); 4: the same with fragments straight from global memory; 5: 4 without the per-half-step overflow vote (bf16); 6: fragments two tiles ahead in AGPR sets */
void cpx_attention_set_lsum(int on);        /* 0 (default): 1 = softmax denominators by an all-ones MFMA (experiment)           */
void cpx_attention_set_trv(int on);         /* 0 (default): V through ds_read_b64_tr_b16                  */
void cpx_postproc_set_fused(int on);        /* 1 (default): the 23-launch fused chain of cpx_compute_masks / cpx_compute_masks_records; 0: the stage-wise sequence (39 launches) */
void cpx_gemm_set_nt(int on);               /* bits 0 / 1 / 2: non-temporal stores for the q / k / V^T thirds of the qkv projection's output (7 = default); 0: ordinary stores */
void cpx_follow_set_early_exit(int on);     /* 1 (default): Euler loop leaves when its orbit closes      */
void cpx_follow_set_lds_window(int on);     /* 1 (default): 32 x 32-cell foreground segments, the Euler loop's taps from an LDS copy of the segment's neighbourhood; 0: round 4 */
/* per-wave cycle stamps of the attention loop segments: dbg [n_subtiles*16*8][4][9] */
int cpx_attention_debug(const void *qkv, const void *rel_h, const void *rel_w, int n_subtiles,
                        void *vT_ws, void *out, unsigned *dbg, void *stream);
/* the same for the production kernel (variant 2, k_attention4p): dbg [n_subtiles*16*8][4][6] */
int cpx_attention4_debug(const void *qkv, const void *rel_h, const void *rel_w, int n_subtiles,
                         void *vT_ws, void *out, unsigned *dbg, void *stream);
/* the same for the 8-wave ping-pong kernel (variant 1): dbg [n_subtiles*16*4][8][9] */
/* round-4 kernel (variant 3, k_attention2q: one wave per SIMD, two query rows per wave): dbg [n_subtiles*64][4][8]; vT = V^T */
int cpx_attention2q_debug(const void *qkv, const void *vT, const void *rel_h, const void *rel_w, int n_subtiles,
                          void *out, unsigned *dbg, void *stream);
void cpx_attention2q_set_ablation(int v);    /* stamped kernel, timing only: 1 = key loop without its MFMAs, 2 = without its vector stream, 3 = without both, 4 = the GL variant */
int cpx_attention8_debug(const void *qkv, const void *rel_h, const void *rel_w, int n_subtiles,
                         void *vT_ws, void *out, unsigned *dbg, void *stream);
#ifdef __cplusplus
}
#endif
#endif
