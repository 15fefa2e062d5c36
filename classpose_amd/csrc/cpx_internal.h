// Internal (non-ABI) interfaces between the translation units of libclasspose_hip.
// Every function takes the element type explicitly: there is no process-global compute dtype,
// so engines of different precision can run from different host threads (include/classpose_hip.h,
// "thread-safe for distinct streams + distinct workspaces").
#pragma once
#include "cpx_common.h"
#include <atomic>

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel instantiation, device), race-free
struct CpxOncePerDevice {
    std::atomic<unsigned long long> done{0};
    template <class F>
    void operator()(F &&f) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        const unsigned long long bit = 1ull << (dev & 63);
        if (!(done.load(std::memory_order_acquire) & bit)) {
            f();                                            // idempotent: two racing threads set the same value
            done.fetch_or(bit, std::memory_order_release);
        }
    }
};

// optional per-launch timing (bench.py's roofline lines): HIP events on the launch stream
// around the kernels of the selected kinds.  Created by cpx_prof_create, carried in
// cpx_net_weights.prof, owned by the caller; one handle per engine / host thread.
enum { CPX_PROF_FC1 = 0, CPX_PROF_ATTN = 1, CPX_PROF_QKV = 2, CPX_PROF_PROJ = 3, CPX_PROF_FC2 = 4,
       CPX_PROF_PE = 5,        // the patch embedding (one launch per forward)
       CPX_PROF_TAIL = 6,      // neck (1x1 conv, LayerNorm2d, 3x3 conv, LayerNorm2d) + head GEMM [+ UNet head]: ONE span per forward
       CPX_PROF_KINDS = 7 };
struct CpxProf {
    hipEvent_t *ev = nullptr;      // 2 per timed launch
    int *kind = nullptr;
    int cap = 0, n = 0, stride = 1;
    int phase = 0;                 // advanced once per forward: layer l is timed when (l + phase) % stride == 0, so a
                                   // stride > 1 rotates through every layer instead of always sampling the same ones
    unsigned kinds_mask = 1;       // bit k: time kernels of kind k
};
// returns true and records the start event when this launch is to be timed
static inline bool cpx_prof_begin(CpxProf *p, int kind, int layer, hipStream_t s) {
    if (!p || !p->ev || p->n >= p->cap || !((p->kinds_mask >> kind) & 1)) return false;
    if (kind < CPX_PROF_PE && ((layer + p->phase) % p->stride) != 0) return false;      // (the once-per-forward kinds ignore the layer stride)
    p->kind[p->n] = kind;
    return hipEventRecord(p->ev[2 * p->n], s) == hipSuccess;
}
static inline void cpx_prof_end(CpxProf *p, hipStream_t s) {
    (void)hipEventRecord(p->ev[2 * p->n + 1], s);
    ++p->n;
}

// half-precision (bf16 / fp16) kernels: cpx_gemm.hip, cpx_net.hip
int cpx_gemm_half(int dtype, const void *A, const void *Wt, int M, int N, int K, int epilogue, const float *bias,
                  const void *aux, void *out, int ld_out, const float *ln_stats, const float *ln_colsum,
                  float *stats_out, void *stream);
int cpx_conv3_half(int dtype, const void *x, const void *Wt, int M, int N, int C, int epilogue, const float *bias,
                   void *out, int ld_out, void *stream);
int cpx_row_stats_half(int dtype, const void *x, int rows, float *stats, void *stream);
int cpx_gemm_half_uses_big_tile(int M, int N, int K, int epilogue);
// one-wave-per-SIMD 256^2 kernel (cpx_gemm4w.hip), bf16 / fp16: gelu(folded-LayerNorm(A) W^T + bias); 1 = launched, 0 = not this kernel's shape
int cpx_gemm4w_gelu_ln(int f16, const void *A, const void *W, int M, int N, int K, const float *bias, const float *ln_stats, const float *ln_colsum,
                       void *out, int ld_out, hipStream_t s);
// ... and bf16(bf16(A W^T + bias) + resid) with the partial row statistics of the output (N = 1024); resid may be out
int cpx_gemm4w_resid_stats(const void *A, const void *W, int M, int N, int K, const float *bias, const void *resid, void *out, int ld_out,
                           float *stats_out, hipStream_t s);
int cpx_layernorm_half(int dtype, const void *x, const float *w, const float *b, int rows, int C, float eps,
                       void *out, void *stream);
int cpx_attention_half(int dtype, const void *qkv, const void *rel_h, const void *rel_w, int n_subtiles, void *vT_ws,
                       void *out, void *stream, bool transpose_v);
int cpx_attention_trv_enabled(void);
// round-4 attention kernel (cpx_attn2q.hip): one wave per SIMD, two query rows per wave; vT holds V^T already
int cpx_attention2q_launch(int dtype, const void *qkv, const void *vT, const void *rel_h, const void *rel_w,
                           int n_subtiles, void *out, int gl, hipStream_t s);

// float32 kernels (exact-f32 MFMA): cpx_net_f32.hip
int cpx_gemm_f32(const float *A, const float *Wt, int M, int N, int K, int epilogue, const float *bias,
                 const float *aux, float *out, int ld_out, void *stream);
int cpx_layernorm_f32(const float *x, const float *w, const float *b, int rows, int C, float eps, float *out,
                      void *stream);
int cpx_attention_f32(const float *qkv, const float *rel_h, const float *rel_w, int n_subtiles, float *out,
                      void *stream);
size_t cpx_net_f32_workspace_bytes(int n_subtiles);
int cpx_net_forward_f32(const cpx_net_weights *w, const void *patches, int nS, float *head, void *workspace,
                        size_t workspace_bytes, void *stream);

// UNet semantic head over token-major tensors of any element type (cpx_net.hip)
int cpx_unet_head_run(int dtype, const cpx_conv_op *ops, int n_ops, const void *feat, int nS, float *head, int ld_head,
                      int col0, void *workspace, size_t ws_bytes, void *stream);
size_t cpx_unet_ws_bytes(int dtype, const cpx_conv_op *ops, int n_ops, int nS);
