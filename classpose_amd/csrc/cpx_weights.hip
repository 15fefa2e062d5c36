// Checkpoint -> kernel operands ON THE DEVICE (a5: the `net.load_model(...)` + `net.to(dtype)` of
// /root/reference/src/classpose/entrypoints/predict_wsi.py:659-727 and models.py:37-69 `resolve_precision`).
//
// The reference's `.to(torch.bfloat16 | float16)` rounds 304 M float32 parameters on the host; until round 5 so did this engine (and folded
// LayerNorm into the two consuming GEMMs there too): 0.6 s at one rank, 3.5 - 4.7 s per rank with eight ranks on a 16-core quota.  Here the
// float32 checkpoint is uploaded as it is and these two kernels do the rounding and the fold -- HBM-bound byte work:
//   k_round_weights   4 B read + 2 (or 4) B written per parameter; 16-byte loads, 8 / 16-byte stores
//   k_fold_layernorm  one workgroup per output row: W[n][:] * gamma -> half, b[n] + W[n][:] . beta, the row sum of the folded, rounded row.
// Both sums are accumulated in float64 in a FIXED order (lane-strided partials, then a fixed tree): the products of two half-rounded values
// and the rounded weights themselves carry <= 22 mantissa bits, so a float64 sum of K <= 2^20 of them is exact short of a 2^30 dynamic
// range -- the result does not depend on the summation order and equals the numpy float64 restatement bit for bit (tests/test_gpu_net.py).
#include "cpx_internal.h"

template <int DT> struct HalfOf;
template <> struct HalfOf<CPX_DT_BF16> { typedef __bf16 T; };
template <> struct HalfOf<CPX_DT_F16> { typedef _Float16 T; };

template <int DT> __device__ __forceinline__ float round_through(float x) { return (float)(typename HalfOf<DT>::T)x; }
template <int DT> __device__ __forceinline__ unsigned short half_bits(float x) {
    typename HalfOf<DT>::T h = (typename HalfOf<DT>::T)x;
    return *reinterpret_cast<unsigned short *>(&h);
}

// n4 = number of 4-element groups; the tail (< 4 elements) is done by the last thread
template <int DT, bool KEEP_F32>
__global__ __launch_bounds__(256) void k_round_weights(const float *__restrict__ src, void *__restrict__ dst, long long n) {
    const long long n4 = n >> 2;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const float4 v = reinterpret_cast<const float4 *>(src)[i];
        const float a = v.x, b = v.y, c = v.z, d = v.w;
        if (KEEP_F32) {
            reinterpret_cast<float4 *>(dst)[i] = make_float4(round_through<DT>(a), round_through<DT>(b), round_through<DT>(c), round_through<DT>(d));
        } else {
            uint2 o;
            o.x = (unsigned)half_bits<DT>(a) | ((unsigned)half_bits<DT>(b) << 16);
            o.y = (unsigned)half_bits<DT>(c) | ((unsigned)half_bits<DT>(d) << 16);
            reinterpret_cast<uint2 *>(dst)[i] = o;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (long long i = n4 << 2; i < n; ++i) {
            const float a = src[i];
            if (KEEP_F32) reinterpret_cast<float *>(dst)[i] = round_through<DT>(a);
            else reinterpret_cast<unsigned short *>(dst)[i] = half_bits<DT>(a);
        }
}

extern "C" int cpx_round_weights(const float *src, void *dst, long long n, int dtype, int keep_f32, void *stream) {
    CPX_REQUIRE(src && dst && n >= 0 && (dtype == CPX_DT_BF16 || dtype == CPX_DT_F16 || dtype == CPX_DT_F32));
    CPX_REQUIRE(((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 15) == 0);
    if (n == 0) return CPX_OK;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == CPX_DT_F32) {                       // --precision fp32: parameters stay what they are
        CPX_REQUIRE(keep_f32);
        if ((const void *)src != dst) CPX_HIP(hipMemcpyAsync(dst, src, (size_t)n * 4, hipMemcpyDeviceToDevice, s));
        return CPX_OK;
    }
    const int grid = (int)((n / 4 + 255) / 256 < 2048 ? ((n / 4 + 255) / 256 > 0 ? (n / 4 + 255) / 256 : 1) : 2048);
    if (dtype == CPX_DT_BF16) {
        if (keep_f32) k_round_weights<CPX_DT_BF16, true><<<grid, 256, 0, s>>>(src, dst, n);
        else k_round_weights<CPX_DT_BF16, false><<<grid, 256, 0, s>>>(src, dst, n);
    } else {
        if (keep_f32) k_round_weights<CPX_DT_F16, true><<<grid, 256, 0, s>>>(src, dst, n);
        else k_round_weights<CPX_DT_F16, false><<<grid, 256, 0, s>>>(src, dst, n);
    }
    CPX_CHECK_LAUNCH();
    return CPX_OK;
}

// fixed-order float64 sum over the 256 threads of a workgroup: within a wave by xor-shuffles (16, 8, 4, 2, 1 ... 32), then the four wave sums in order
__device__ __forceinline__ double block_sum_f64(double v, double *lds) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    const int wave = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) lds[wave] = v;
    __syncthreads();
    return ((lds[0] + lds[1]) + lds[2]) + lds[3];
}

template <int DT>
__global__ __launch_bounds__(256) void k_fold_layernorm(const float *__restrict__ w, const float *__restrict__ b, const float *__restrict__ gamma,
                                                        const float *__restrict__ beta, int K, unsigned short *__restrict__ wf,
                                                        float *__restrict__ bf, float *__restrict__ colsum) {
    __shared__ double lds[4];
    const int n = blockIdx.x;
    const float *row = w + (size_t)n * K;
    unsigned short *orow = wf + (size_t)n * K;
    double dot = 0.0, sum = 0.0;
    for (int k = threadIdx.x; k < K; k += 256) {
        const float wq = round_through<DT>(row[k]);
        const float g = round_through<DT>(gamma[k]), bt = round_through<DT>(beta[k]);
        const float f = wq * g;                                         // float32 product (exact: two <= 11-bit significands), rounded once
        const unsigned short h = half_bits<DT>(f);
        orow[k] = h;
        typename HalfOf<DT>::T hv = *reinterpret_cast<const typename HalfOf<DT>::T *>(&h);
        sum += (double)(float)hv;
        dot += (double)wq * (double)bt;
    }
    dot = block_sum_f64(dot, lds);
    sum = block_sum_f64(sum, lds);
    if (threadIdx.x == 0) {
        bf[n] = round_through<DT>(b[n]) + (float)dot;
        colsum[n] = (float)sum;
    }
}

extern "C" int cpx_fold_layernorm(const float *w, const float *b, const float *gamma, const float *beta, int N, int K, int dtype,
                                  void *w_folded, float *b_folded, float *colsum, void *stream) {
    CPX_REQUIRE(w && b && gamma && beta && w_folded && b_folded && colsum && N > 0 && K > 0);
    CPX_REQUIRE(dtype == CPX_DT_BF16 || dtype == CPX_DT_F16);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == CPX_DT_BF16)
        k_fold_layernorm<CPX_DT_BF16><<<N, 256, 0, s>>>(w, b, gamma, beta, K, (unsigned short *)w_folded, b_folded, colsum);
    else
        k_fold_layernorm<CPX_DT_F16><<<N, 256, 0, s>>>(w, b, gamma, beta, K, (unsigned short *)w_folded, b_folded, colsum);
    CPX_CHECK_LAUNCH();
    return CPX_OK;
}

// The whole checkpoint in ONE call: the host side (engine.NetWeights.from_state_dict) lists what every parameter becomes, this function
// streams the float32 sources host -> staging (hipMemcpyAsync from the checkpoint's mapped pages) and queues the kernel behind each copy.
// One foreign call instead of ~600: the converting thread shares its interpreter with the slide readers, and re-acquiring the
// interpreter lock after every copy / launch cost it 2.5 s per rank with eight ranks on a 16-core quota (0.3 s of actual work).
extern "C" int cpx_weights_build(const cpx_weight_job *jobs, int n_jobs, void *stage_base, size_t stage_bytes, void *stream) {
    CPX_REQUIRE(jobs && n_jobs >= 0 && (stage_base || stage_bytes == 0));
    hipStream_t s = (hipStream_t)stream;
    char *stage = (char *)stage_base;
    for (int i = 0; i < n_jobs; ++i) {
        const cpx_weight_job &j = jobs[i];
        CPX_REQUIRE(j.op >= CPX_WJ_ROUND_HALF && j.op <= CPX_WJ_FOLD_LN && j.n >= 0);
        if (j.n == 0) continue;
        if (j.op == CPX_WJ_COPY_F32) {
            CPX_REQUIRE(j.src_host[0] && j.dst[0]);
            CPX_HIP(hipMemcpyAsync(j.dst[0], j.src_host[0], (size_t)j.n * 4, hipMemcpyHostToDevice, s));
            continue;
        }
        const int n_src = j.op == CPX_WJ_FOLD_LN ? 4 : 1;
        const size_t elems[4] = {j.op == CPX_WJ_FOLD_LN ? (size_t)j.n * (size_t)j.K : (size_t)j.n, (size_t)j.n, (size_t)j.K, (size_t)j.K};
        const float *dev_src[4] = {nullptr, nullptr, nullptr, nullptr};
        for (int k = 0; k < n_src; ++k) {
            CPX_REQUIRE(j.src_host[k] && (j.stage_off[k] & 15) == 0 && j.stage_off[k] + elems[k] * 4 <= stage_bytes);
            dev_src[k] = (const float *)(stage + j.stage_off[k]);
            CPX_HIP(hipMemcpyAsync((void *)dev_src[k], j.src_host[k], elems[k] * 4, hipMemcpyHostToDevice, s));
        }
        int rc;
        if (j.op == CPX_WJ_FOLD_LN) {
            CPX_REQUIRE(j.K > 0);
            rc = cpx_fold_layernorm(dev_src[0], dev_src[1], dev_src[2], dev_src[3], (int)j.n, j.K, j.dtype, j.dst[0], (float *)j.dst[1],
                                    (float *)j.dst[2], stream);
        } else {
            rc = cpx_round_weights(dev_src[0], j.dst[0], j.n, j.dtype, j.op == CPX_WJ_ROUND_F32, stream);
        }
        if (rc != CPX_OK) return rc;
    }
    return CPX_OK;
}
