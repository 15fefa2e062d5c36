// Sustained bf16 matrix throughput of the two MFMA shapes on the whole chip, accumulators in AGPRs, operands in VGPRs, nothing else running:
// v_mfma_f32_16x16x32_bf16 (the GEMMs of csrc/cpx_gemm*.hip) against v_mfma_f32_32x32x16_bf16 -- the same flops per operand fragment read
// from LDS, but HALF the operand-register reads and half the instructions per flop.  The part holds ~1.85 - 2.0 GHz under the MFMA stream
// of the GEMMs, i.e. it is power-limited there: does the shape change what it sustains?
// One wave per SIMD with 256 accumulator registers (the 4w kernel's shape: 128 x 128 per wave) and random normal operands (toggle rate of real data);
// 16 distinct A and 16 distinct B fragments per wave so that consecutive instructions read different registers, as a K step of the GEMM does.
// ~2 ms per launch (past the power controller's time constant), HIP events around it.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_shape mfma_shape.hip && ./mfma_shape
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

// SHAPE 0: 16x16x32, a 128 x 128 wave tile = 8 x 8 accumulator tiles of 4 registers; one K = 32 step = 64 MFMAs on 8 A and 8 B fragments
// SHAPE 1: 32x32x16, a 128 x 128 wave tile = 4 x 4 accumulator tiles of 16 registers; one K = 32 step = 2 x 16 MFMAs on 2 x (4 A + 4 B) fragments
template <int SHAPE>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) k(const u32x4 *__restrict__ frag, int steps, float *sink) {
    u32x4 a[8], b[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = frag[(i * 256 + threadIdx.x)]; b[i] = frag[((8 + i) * 256 + threadIdx.x)]; }
    if constexpr (SHAPE == 0) {
        f32x4 acc[64];
#pragma unroll
        for (int i = 0; i < 64; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
        for (int s = 0; s < steps; ++s) {
#pragma unroll
            for (int m = 0; m < 8; ++m)
#pragma unroll
                for (int n = 0; n < 8; ++n)
                    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[m * 8 + n]) : "v"(a[m]), "v"(b[n]));
        }
        float r = 0.f;
#pragma unroll
        for (int i = 0; i < 64; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
        sink[blockIdx.x * 256 + threadIdx.x] = r;
    } else {
        f32x16 acc[16];
#pragma unroll
        for (int i = 0; i < 16; ++i)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
#pragma unroll 1
        for (int s = 0; s < steps; ++s) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int n = 0; n < 4; ++n)
                        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[m * 4 + n]) : "v"(a[kk * 4 + m]), "v"(b[kk * 4 + n]));
        }
        float r = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i)
#pragma unroll
            for (int j = 0; j < 16; ++j) r += acc[i][j];
        sink[blockIdx.x * 256 + threadIdx.x] = r;
    }
}

int main() {
    std::mt19937 rng(1);
    std::normal_distribution<float> nd(0.f, 1.f);
    std::vector<unsigned> h(16 * 256 * 4);
    auto bf = [](float x) { unsigned u; memcpy(&u, &x, 4); return (u + 0x7FFF + ((u >> 16) & 1)) >> 16; };
    for (auto &w : h) w = bf(nd(rng)) | (bf(nd(rng)) << 16);
    u32x4 *frag; float *sink;
    hipMalloc(&frag, h.size() * 4); hipMemcpy(frag, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMalloc(&sink, 1024 * 256 * 4);
    const int steps = 20000;                                  // x 128 x 128 x 32 x 2 flops per wave
    const double flops = 2.0 * 128 * 128 * 32 * (double)steps * 4 * 256;
    for (int rep = 0; rep < 3; ++rep)
        for (int shape = 0; shape < 2; ++shape) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0, 0);
            if (shape == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 0, 0, frag, steps, sink);
            else hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 0, 0, frag, steps, sink);
            hipEventRecord(e1, 0); hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("%s  %8.3f ms  %7.1f TFLOP/s = %.3f of 2.5 PF   (%.0f MHz-equivalent of the 1024 flops/clk/SIMD rate)\n",
                   shape == 0 ? "v_mfma_f32_16x16x32_bf16" : "v_mfma_f32_32x32x16_bf16", ms, flops / ms / 1e9, flops / ms / 1e9 / 2500.0,
                   flops / (ms * 1e-3) / (1024.0 * 1024) / 1e6);
        }
    return 0;
}
