// Micro-benchmark: does VALU work of one wave overlap with fp32 MFMA (v_mfma_f32_32x32x2_f32) of another wave on the same
// SIMD?  512-thread workgroups = 2 waves per SIMD, one workgroup per CU.  Waves 0-3 run `mf` MFMA iterations, waves 4-7 run
// `va` VALU iterations (mode 0: v_fma_f32, mode 1: 32-bit integer mul/add address-style math, mode 2: ds_write+ds_read LDS traffic).
// Prints cycles for MFMA alone, VALU alone and both together.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(512) k(int mf, int va, int mode, float* out, long long* cyc) {
    __shared__ float lds[8192];
    const int wave = threadIdx.x >> 6;
    long long t0 = __builtin_readcyclecounter();
    if (wave < 4) {
        f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
        float x = threadIdx.x * 1e-3f, y = 1.0f;
        for (int i = 0; i < mf; ++i) {
            a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a3, 0, 0, 0);
        }
        out[blockIdx.x * 512 + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
    } else {
        if (mode == 0) {
            float a = threadIdx.x, b = 1.0001f, c = 0.5f, d = 0.25f;
            for (int i = 0; i < va; ++i) {
#pragma unroll
                for (int u = 0; u < 8; ++u) { a = fmaf(a, b, c); d = fmaf(d, b, a); c = fmaf(c, b, d); b = fmaf(b, 0.999f, 1e-6f); }
            }
            out[blockIdx.x * 512 + threadIdx.x] = a + d + c + b;
        } else if (mode == 1) {
            unsigned a = threadIdx.x, b = 12345u, c = 7u, d = 99u;
            for (int i = 0; i < va; ++i) {
#pragma unroll
                for (int u = 0; u < 8; ++u) { a = a * b + c; d = (d + a) ^ (a >> 3); c = c + d; b = b + (c & 15u); }
            }
            out[blockIdx.x * 512 + threadIdx.x] = (float)(a + d + c + b);
        } else {
            f32x4 v = {1.f, 2.f, 3.f, 4.f};
            float* p = lds + (threadIdx.x & 255) * 4 * 4;
            for (int i = 0; i < va; ++i) {
#pragma unroll
                for (int u = 0; u < 4; ++u) { *reinterpret_cast<f32x4*>(p + 4 * u) = v; }
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int u = 0; u < 4; ++u) { v += *reinterpret_cast<f32x4*>(p + 4 * ((u + 1) & 3)); }
            }
            out[blockIdx.x * 512 + threadIdx.x] = v[0] + v[1] + v[2] + v[3];
        }
    }
    long long t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 8);
    long long h[256 * 8];
    const char* names[3] = {"v_fma_f32", "int mul/add", "LDS b128 write+read"};
    for (int mode = 0; mode < 3; ++mode) {
        int mf = 2000, va = mode == 2 ? 2000 : 1000;
        for (int cfg = 0; cfg < 3; ++cfg) {
            int m = cfg == 1 ? 0 : mf, v = cfg == 0 ? 0 : va;
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            k<<<256, 512>>>(m, v, mode, out, cyc); hipDeviceSynchronize();
            hipEventRecord(e0); k<<<256, 512>>>(m, v, mode, out, cyc); hipEventRecord(e1); hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
            printf("%-20s %-10s  %.3f ms   wave0 (mfma) %lld cyc   wave4 (other) %lld cyc\n", names[mode],
                   cfg == 0 ? "mfma only" : cfg == 1 ? "other only" : "both", ms, h[0], h[4]);
        }
    }
    return 0;
}
