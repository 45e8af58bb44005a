// Micro-benchmark (gfx950): does VALU / LDS work of one wave overlap with the bf16 MFMA (v_mfma_f32_32x32x16_bf16) of ANOTHER wave on the
// same SIMD, and of the SAME wave?  512-thread workgroups (2 waves per SIMD), one workgroup per CU.
//   pairing 0: waves 0-3 MFMA, waves 4-7 other     pairing 1: even waves MFMA, odd waves other     pairing 2: every wave does both, interleaved
//   other = mode 0: v_fma_f32   mode 1: the exact 3-term bf16 split (v_cvt_pk_bf16_f32 + subtracts)   mode 2: LDS b128 reads
// build + run: hipcc --offload-arch=gfx950 -O2 tools/ubench/bf16mfma_overlap.hip -o /tmp/ovl && /tmp/ovl
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void split2(float x0, float x1, unsigned& w1, unsigned& w2, unsigned& w3) {
    f32x2 v; v[0] = x0; v[1] = x1;
    w1 = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
    f32x2 r; r[0] = v[0] - __uint_as_float(w1 << 16); r[1] = v[1] - __uint_as_float(w1 & 0xffff0000u);
    w2 = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2));
    f32x2 q; q[0] = r[0] - __uint_as_float(w2 << 16); q[1] = r[1] - __uint_as_float(w2 & 0xffff0000u);
    w3 = __builtin_bit_cast(unsigned, __builtin_convertvector(q, bf16x2));
}

template <int MODE>
__device__ __forceinline__ void other_step(float& a, float& b, float& c, float& d, unsigned& acc, const float* lds, f32x4& lv) {
    if (MODE == 0) {
#pragma unroll
        for (int u = 0; u < 4; ++u) { a = fmaf(a, b, c); d = fmaf(d, b, a); c = fmaf(c, b, d); b = fmaf(b, 0.999f, 1e-6f); }
    } else if (MODE == 1) {
        unsigned w1, w2, w3;
        split2(a, b, w1, w2, w3); acc ^= w1 + w2 + w3;
        split2(c, d, w1, w2, w3); acc ^= w1 ^ w2 ^ w3;
        a += 1.25f; b += 0.75f; c -= 0.5f; d += 0.125f;
    } else {
#pragma unroll
        for (int u = 0; u < 4; ++u) lv += *reinterpret_cast<const f32x4*>(lds + ((threadIdx.x * 4 + u * 2048 + (int)lv[0]) & 8188));
    }
}

template <int MODE>
__global__ void __launch_bounds__(512) k(int mf, int va, int pairing, float* out, long long* cyc) {
    __shared__ float lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 512) lds[i] = 0.f;
    __syncthreads();
    const int wave = threadIdx.x >> 6;
    const bool do_m = pairing == 2 ? true : (pairing == 0 ? wave < 4 : (wave & 1) == 0);
    const bool do_o = pairing == 2 ? true : !do_m;
    f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
    bf16x8 x, y;
    for (int e = 0; e < 8; ++e) { x[e] = (__bf16)(threadIdx.x * 1e-3f + e); y[e] = (__bf16)1.0f; }
    float a = threadIdx.x, b = 1.0001f, c = 0.5f, d = 0.25f;
    unsigned acc = 0;
    f32x4 lv = {0.f, 0.f, 0.f, 0.f};
    long long t0 = __builtin_readcyclecounter();
    if (pairing == 2) {
        // same wave: 4 MFMAs then one other_step, mf iterations (va scales the other work: va other_steps per 4 MFMAs)
        for (int i = 0; i < mf; ++i) {
            a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a3, 0, 0, 0);
            for (int j = 0; j < va; ++j) other_step<MODE>(a, b, c, d, acc, lds, lv);
        }
    } else {
        if (do_m)
            for (int i = 0; i < mf; ++i) {
                a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a2, 0, 0, 0);
                a3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a3, 0, 0, 0);
            }
        if (do_o)
            for (int i = 0; i < va; ++i) other_step<MODE>(a, b, c, d, acc, lds, lv);
    }
    long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 512 + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3] + a + b + c + d + (float)acc + lv[0] + lv[1];
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int MODE>
void run(const char* name, float* out, long long* cyc) {
    long long h[8];
    for (int pairing = 0; pairing < 3; ++pairing) {
        const int mf = 2000, va = pairing == 2 ? 1 : 8000;
        for (int cfg = 0; cfg < 3; ++cfg) {
            int m = cfg == 1 ? 0 : mf, v = cfg == 0 ? 0 : va;
            if (pairing == 2 && cfg == 1) m = mf;  // (same-wave form needs the loop: "other only" is not separable; skip)
            if (pairing == 2 && cfg == 1) continue;
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            k<MODE><<<256, 512>>>(m, v, pairing, out, cyc); hipDeviceSynchronize();
            hipEventRecord(e0); k<MODE><<<256, 512>>>(m, v, pairing, out, cyc); hipEventRecord(e1); hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
            printf("%-28s pairing %d %-10s  %.3f ms   wave0 %lld  wave1 %lld  wave4 %lld cyc\n", name, pairing,
                   cfg == 0 ? "mfma only" : cfg == 1 ? "other only" : "both", ms, h[0], h[1], h[4]);
        }
    }
}

int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 8);
    run<0>("v_fma_f32", out, cyc);
    run<1>("3-term bf16 split", out, cyc);
    run<2>("LDS b128 reads", out, cyc);
    return 0;
}
