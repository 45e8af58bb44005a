// dense_bf3.h -- the large dense contractions of the policy (visual fc 25088 -> 512 of SimpleCNN: forward, data gradient, weight
// gradient; `nn.Linear` in rl/models/simple_cnn.py:91-93 and its autograd) as ONE plain GEMM kernel on the split-bf16 matrix path.
//
// igemm_bf3.h serves these shapes through its problem functors (gather keys, a single LDS buffer, 128 x 128 tiles on four waves,
// two barriers per k-tile) at 0.22-0.27 of the 416.7 TFLOP/s-equivalent ceiling (profiles/r05_c2_bench.json).  A dense matrix needs
// none of that generality:
//   * 256 x 128 output tile per workgroup, 8 waves as 4 (M) x 2 (N), each wave 64 x 64 = 2 x 2 MFMA tiles: per k-step of 16 a wave
//     reads 6 + 6 operand-plane fragments for 24 v_mfma_f32_32x32x16_bf16 (the 3-term split shares fragments between the six partial
//     products) -- half of the LDS bandwidth at the full MFMA rate;
//   * DOUBLE-buffered LDS, one barrier per k-tile of 32: tile kt+1 is split and written while tile kt feeds the matrix pipe, the
//     global loads of tile kt+2 are in flight across the whole iteration;
//   * an operand is either k-contiguous (activations / weights in forward form: 16-byte loads along k, planes [row][32 k] with 64-byte
//     rows and an XOR swizzle of the 16-byte chunks, fragments by ds_read_b128) or k-strided (the frame-indexed operands of the weight
//     gradient, W in the data gradient: 16-byte loads along the row index, planes [32-row block][k][32 rows] written as they arrive,
//     fragments by the LDS transpose read ds_read_b64_tr_b16 -- no register transposes, no strided gathers);
//   * the accumulators leave through LDS (the operand buffers are dead by then): whole 512-byte output rows per 32 lanes instead of
//     the MFMA layout's 16-byte pieces; the weight gradient's NHWC-flatten -> NCHW-flatten column permutation (hw*C + c -> c*HW + hw)
//     is applied while the tile is read back, 16-byte stores of 4 consecutive hw;
//   * work items (tile, K slice) are dealt so that items sharing operand bytes sit on one XCD (private L2s): all tiles of a K slice
//     for split-K launches, the row tiles of a column block otherwise.
// Arithmetic is igemm_bf3.h's: exact 3-term split, six partial products smallest first, fp32 accumulate, alternating sign schedule.
#pragma once
#include "igemm_bf3.h"

namespace hab {

struct DenseArgs {
    int M, N, K;               // C[M][N] = A[M][K] * B[N][K]^T; K % 32 == 0
    const float* a; long long lda;  // A_IC == 0: a[m * lda + k]   A_IC == 1: a[k * lda + m]
    const float* b; long long ldb;  // B_IC == 0: b[n * ldb + k]   B_IC == 1: b[k * ldb + n]
    float* c; long long ldc;        // c[m * ldc + col(n)]
    const float* bias;              // [N] or null
    int relu, accumulate;
    int perm_c, perm_hw;            // perm_c > 0: col(n = hw * perm_c + cc) = cc * perm_hw + hw   (perm_c == 32, N % 128 == 0)
    long long a_bytes, b_bytes;     // extents of a / b in bytes (buffer descriptors; < 2^31)
    float* partial;                 // nsplit > 1: slabs [kz][M][N], summed (+ epilogue) by the caller's reduction pass
    int nsplit;
    int sign_schedule;
    int ablate;                     // development (HAB_DENSE_ABLATE): 1 no global loads, 2 no MFMAs, 4 no split / LDS writes
    int skew;                       // development: 0 = every wave stages before its MFMAs, 1 = waves 0-3 before / 4-7 after, 2 = even / odd
};

constexpr int DN_BM = 256, DN_BN = 128, DN_BK = 32, DN_NT = 512;
constexpr int DN_IC_BLOCK = 32 * 64 + 64;             // bytes of one [32 k][32 rows] block + 64 (consecutive blocks land on distinct bank windows)
constexpr int DN_A_PLANE = (DN_BM / 32) * DN_IC_BLOCK;  // 16896 B (>= 256 rows x 64 B of the k-contiguous form)
constexpr int DN_B_PLANE = (DN_BN / 32) * DN_IC_BLOCK;  // 8448 B
constexpr int DN_BUF = 3 * (DN_A_PLANE + DN_B_PLANE);   // 76032 B
constexpr int DN_EPI_LD = DN_BN + 4;                    // floats per row of the output tile in LDS
constexpr int DN_LDS_BYTES = (2 * DN_BUF > DN_BM * DN_EPI_LD * 4) ? 2 * DN_BUF : DN_BM * DN_EPI_LD * 4;  // 152064 B

// development (HAB_DENSE_ABLATE bit 4): shader-clock stamps of workgroup 0's waves at the phase boundaries of k-tiles 8..15
// [wave][k-tile - 8][5]: iteration start, after the early stage + fetch, after the MFMAs, after the late stage + fetch, after the barrier
__device__ long long dn_trace[8][8][5];
// The ablation masks and the stamps are compiled in only with -DHAB_DENSE_DEV (make EXTRA=-DHAB_DENSE_DEV; tools/dense_trace.py,
// tools/bench_dense.py with HAB_DENSE_ABLATE): the production kernel carries none of their branches.
#ifdef HAB_DENSE_DEV
constexpr bool DN_DEV = true;
#else
constexpr bool DN_DEV = false;
#endif

typedef short dn_v4s __attribute__((ext_vector_type(4)));
__device__ __forceinline__ dn_v4s dn_tr_read(const unsigned char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) dn_v4s*)p);
}

template <int A_IC, int B_IC>
__global__ void __launch_bounds__(DN_NT) dense_bf3_kernel(const DenseArgs g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dn_smem[];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, hi = lane >> 5;

    const int nt_m = cdiv(g.M, DN_BM), nt_n = cdiv(g.N, DN_BN), ntiles = nt_m * nt_n;
    int tile, kz;
    {
        const int b = blockIdx.x, xcd = b & 7, slot = b >> 3;
        if (g.nsplit > 1) {  // the tiles of one K slice share both operands' slice: same XCD, consecutive slots
            tile = slot % ntiles;
            kz = (slot / ntiles) * 8 + xcd;
            if (kz >= g.nsplit) return;
        } else {             // XCD-contiguous runs of tiles
            const int q = ntiles >> 3, r = ntiles & 7;
            const int len = q + (xcd < r ? 1 : 0);
            if (slot >= len) return;
            tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
            kz = 0;
        }
    }
    // the operand with more tiles is the large one: its tile index runs slowest, so the tiles that share one of its blocks are neighbours
    const bool m_fastest = nt_m <= nt_n;
    const int tile_m = m_fastest ? tile % nt_m : tile / nt_n, tile_n = m_fastest ? tile / nt_m : tile % nt_n;
    const int m0 = tile_m * DN_BM, n0 = tile_n * DN_BN;
    const int ktiles = g.K / DN_BK;
    const int kt0 = (int)((long long)kz * ktiles / g.nsplit), kt1 = (int)((long long)(kz + 1) * ktiles / g.nsplit);
    const int ntk = kt1 - kt0;
    const bool flip = g.sign_schedule && (((tile_m + tile_n + kz) & 1) != 0);
    const unsigned sgn = flip ? 0x80000000u : 0u;
    // Flatten permutation written by this kernel (no split-K): the tile's 128 columns are 8 consecutive hw x 16 channels (local column
    // c = hw_l * 16 + cc_l  <->  n = (8 (tile_n / 2) + hw_l) * 32 + 16 (tile_n % 2) + cc_l) instead of 4 hw x 32 channels, so that the 8 hw of a
    // channel leave as ONE full 32-byte sector of the output row (with 4 hw per tile every sector reached HBM twice, half-filled: 112 MB
    // written for the 51 MB of SimpleCNN's fc gradient, profiles/r06_c2_hbm_traffic.txt).  The operand is still read 64 contiguous bytes
    // per hw; the other half of each 128-byte line belongs to the neighbouring tile of the same XCD.
    const bool ptile = B_IC && g.perm_c > 0 && g.nsplit <= 1 && (g.perm_hw & 7) == 0;

    // ---- staging maps (everything per-thread is computed ONCE: on this chip VALU instructions and bf16 MFMAs of the two waves of a SIMD do
    // not overlap -- tools/ubench/bf16mfma_overlap.hip, profiles/r06_bf16mfma_overlap.txt: a wave's VALU stream waits out the other
    // wave's MFMA stream -- so every VALU instruction in the k-loop is matrix time lost) ----
    // k-contiguous operand: unit = (row, 4 consecutive k): kq = t & 7, row = (t >> 3) + 64 j
    // k-strided operand (A: 256 rows): unit = (k, 4 consecutive rows): rq = t & 63, k = (t >> 6) + 8 j;   (B: 128 rows): rq = t & 31, k = (t >> 5) + 16 j
    // Loads go through buffer descriptors: per-thread byte offset fixed, the k-tile advance is a scalar offset, rows beyond M / N point
    // past the descriptor's extent and read as zero.
    constexpr int AU = 4, BU = 2;
    constexpr unsigned OOB = 0x80000000u;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.a), 0, (int)g.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.b), 0, (int)g.b_bytes, 0x00020000);
    u32x4 araw[AU], braw[BU];
    unsigned avoff[AU], bvoff[BU];
    int aoff[AU], boff[BU];   // LDS byte offsets inside a buffer's A / B plane 0
#pragma unroll
    for (int j = 0; j < AU; ++j) {
        if constexpr (A_IC) {
            const int rq = t & 63, k = (t >> 6) + 8 * j, m = m0 + 4 * rq;
            avoff[j] = m < g.M ? (unsigned)(((size_t)(kt0 * DN_BK + k) * g.lda + m) * 4) : OOB;
            aoff[j] = (rq >> 3) * DN_IC_BLOCK + k * 64 + (rq & 7) * 8;
        } else {
            const int kq = t & 7, row = (t >> 3) + 64 * j, m = m0 + row;
            avoff[j] = m < g.M ? (unsigned)(((size_t)m * g.lda + kt0 * DN_BK + 4 * kq) * 4) : OOB;
            aoff[j] = row * 64 + ((((kq >> 1) ^ (row >> 2)) & 3) << 4) + (kq & 1) * 8;
        }
    }
#pragma unroll
    for (int j = 0; j < BU; ++j) {
        if constexpr (B_IC) {
            const int rq = t & 31, k = (t >> 5) + 16 * j;
            const int n = ptile ? (8 * (tile_n >> 1) + (rq >> 2)) * 32 + (tile_n & 1) * 16 + 4 * (rq & 3) : n0 + 4 * rq;
            bvoff[j] = n < g.N ? (unsigned)(((size_t)(kt0 * DN_BK + k) * g.ldb + n) * 4) : OOB;
            boff[j] = (rq >> 3) * DN_IC_BLOCK + k * 64 + (rq & 7) * 8;
        } else {
            const int kq = t & 7, row = (t >> 3) + 64 * j, n = n0 + row;
            bvoff[j] = n < g.N ? (unsigned)(((size_t)n * g.ldb + kt0 * DN_BK + 4 * kq) * 4) : OOB;
            boff[j] = row * 64 + ((((kq >> 1) ^ (row >> 2)) & 3) << 4) + (kq & 1) * 8;
        }
    }
    const unsigned a_step = (A_IC ? (unsigned)(DN_BK * g.lda) : (unsigned)DN_BK) * 4u, b_step = (B_IC ? (unsigned)(DN_BK * g.ldb) : (unsigned)DN_BK) * 4u;
    auto fetch = [&](int kt) {
        if (DN_DEV && (g.ablate & 1)) return;
#pragma unroll
        for (int j = 0; j < AU; ++j) araw[j] = __builtin_amdgcn_raw_buffer_load_b128(ra, (int)avoff[j], (int)((unsigned)kt * a_step), 0);
#pragma unroll
        for (int j = 0; j < BU; ++j) braw[j] = __builtin_amdgcn_raw_buffer_load_b128(rb, (int)bvoff[j], (int)((unsigned)kt * b_step), 0);
    };
    auto put = [&](unsigned char* dst, int plane_bytes, const u32x4 raw, const unsigned flipbit) {  // 4 values -> 8 bytes in each plane
        unsigned p1, p2, p3, q1, q2, q3;
        bf3_split2(__uint_as_float(raw[0] ^ flipbit), __uint_as_float(raw[1] ^ flipbit), p1, p2, p3);
        bf3_split2(__uint_as_float(raw[2] ^ flipbit), __uint_as_float(raw[3] ^ flipbit), q1, q2, q3);
        u32x2 w1, w2, w3;
        w1[0] = p1; w1[1] = q1; w2[0] = p2; w2[1] = q2; w3[0] = p3; w3[1] = q3;
        *reinterpret_cast<u32x2*>(dst) = w1;
        *reinterpret_cast<u32x2*>(dst + plane_bytes) = w2;
        *reinterpret_cast<u32x2*>(dst + 2 * plane_bytes) = w3;
    };
    auto stage = [&](int buf) {
        if (DN_DEV && (g.ablate & 4)) return;
        unsigned char* As = dn_smem + buf * DN_BUF;
        unsigned char* Bs = As + 3 * DN_A_PLANE;
#pragma unroll
        for (int j = 0; j < AU; ++j) put(As + aoff[j], DN_A_PLANE, araw[j], 0u);
        if (flip) {  // (workgroup-uniform: the sign flips of B cost VALU only where the schedule asks for them)
#pragma unroll
            for (int j = 0; j < BU; ++j) put(Bs + boff[j], DN_B_PLANE, braw[j], 0x80000000u);
        } else {
#pragma unroll
            for (int j = 0; j < BU; ++j) put(Bs + boff[j], DN_B_PLANE, braw[j], 0u);
        }
    };
    // fragment of the 32 rows [rb, rb + 32) of an operand tile at k-step c (16 k): lane l <- row l & 31, k = 8 (l >> 5) .. + 7.
    // Byte offsets inside plane 0 of a buffer, computed once; the plane and the buffer are immediate / scalar offsets.
    int fa[2][2], fb[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            {
                const int rbase = wm * 64 + i * 32;
                if constexpr (A_IC) {
                    const int grp = lane >> 4, ii = lane & 15;
                    fa[i][c] = (rbase >> 5) * DN_IC_BLOCK + (c * 16 + 8 * (grp >> 1) + (ii >> 2)) * 64 + (grp & 1) * 32 + (ii & 3) * 8;
                } else {
                    const int row = rbase + li;
                    fa[i][c] = row * 64 + ((((c * 2 + hi) ^ (row >> 2)) & 3) << 4);
                }
            }
            {
                const int rbase = wn * 64 + i * 32;
                if constexpr (B_IC) {
                    const int grp = lane >> 4, ii = lane & 15;
                    fb[i][c] = (rbase >> 5) * DN_IC_BLOCK + (c * 16 + 8 * (grp >> 1) + (ii >> 2)) * 64 + (grp & 1) * 32 + (ii & 3) * 8;
                } else {
                    const int row = rbase + li;
                    fb[i][c] = row * 64 + ((((c * 2 + hi) ^ (row >> 2)) & 3) << 4);
                }
            }
        }
    auto frag = [&](const unsigned char* p, bool ic) -> bf16x8 {
        if (ic) {
            const dn_v4s lo = dn_tr_read(p), hi4 = dn_tr_read(p + 4 * 64);
            return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi4, 0, 1, 2, 3, 4, 5, 6, 7));
        }
        return *reinterpret_cast<const bf16x8*>(p);
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;

    if (ntk > 0) {
        fetch(0);
        stage(0);
        if (ntk > 1) fetch(1);
        __syncthreads();
    }
    // Two waves share a SIMD (waves w and w + 4; HAB_DENSE_SKEW picks the pairing): one of them splits + writes tile kt + 1 BEFORE its MFMAs
    // of tile kt, the other AFTER -- so that one wave's VALU / LDS-write phase runs beside the other's matrix phase instead of all eight
    // waves staging together and then all eight queueing for the matrix pipe.  Both orders touch different buffers; the barrier at the
    // end of the iteration is the only ordering either needs.
    const bool early = g.skew == 0 ? true : (g.skew == 1 ? wave < 4 : (wave & 1) == 0);
    for (int kt = 0; kt < ntk; ++kt) {
        const unsigned char* As = dn_smem + (kt & 1) * DN_BUF;
        const unsigned char* Bs = As + 3 * DN_A_PLANE;
        // (stage / fetch run unconditionally: past the end they rewrite the dead buffer from the last tile's registers / re-read the last tile)
        const int ktf = kt + 2 < ntk ? kt + 2 : ntk - 1;
        const bool tr = DN_DEV && (g.ablate & 16) && blockIdx.x == 0 && kt >= 8 && kt < 16;
        if (tr && lane == 0) dn_trace[wave][kt - 8][0] = __builtin_readcyclecounter();
        if (early) {
            stage((kt + 1) & 1);   // (its buffer was last read in iteration kt - 1, behind that iteration's barrier)
            fetch(ktf);
        }
        if (tr && lane == 0) dn_trace[wave][kt - 8][1] = __builtin_readcyclecounter();
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            bf16x8 af[2][3], bf[2][3];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) af[i][pl] = frag(As + pl * DN_A_PLANE + fa[i][c], A_IC);
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) bf[j][pl] = frag(Bs + pl * DN_B_PLANE + fb[j][c], B_IC);
            if (DN_DEV && (g.ablate & 2)) {  // keep the fragments alive without the matrix pipe
                if (af[0][0][0] == (__bf16)123.f && bf[1][2][7] == (__bf16)77.f) acc[0][0][0] += 1.f;
                continue;
            }
            constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};  // six partial products, smallest weight first
#pragma unroll
            for (int q = 0; q < 6; ++q)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[j][PB[q]], af[i][PA[q]], acc[i][j], 0, 0, 0);  // transposed accumulator
        }
        if (tr && lane == 0) dn_trace[wave][kt - 8][2] = __builtin_readcyclecounter();
        if (!early) {
            stage((kt + 1) & 1);
            fetch(ktf);
        }
        if (tr && lane == 0) dn_trace[wave][kt - 8][3] = __builtin_readcyclecounter();
        __syncthreads();
        if (tr && lane == 0) dn_trace[wave][kt - 8][4] = __builtin_readcyclecounter();
    }

    // ---- epilogue through LDS: T[256][DN_EPI_LD] ----
    float* T = reinterpret_cast<float*>(dn_smem);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = flip ? -acc[i][j][4 * q + e] : acc[i][j][4 * q + e];
                *reinterpret_cast<f32x4*>(T + (wm * 64 + i * 32 + li) * DN_EPI_LD + wn * 64 + j * 32 + 8 * q + 4 * hi) = v;
            }
    __syncthreads();
    const bool split = g.nsplit > 1;
    float* cbase = split ? g.partial + (size_t)kz * g.M * g.N : g.c;
    const long long ldc = split ? g.N : g.ldc;
    if (ptile) {
        const int ccl = t & 15, half = (t >> 4) & 1, hw0 = 8 * (tile_n >> 1) + 4 * half, cc = (tile_n & 1) * 16 + ccl;
        for (int row = t >> 5; row < DN_BM; row += DN_NT / 32) {
            const int m = m0 + row;
            if (m >= g.M) break;
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = T[row * DN_EPI_LD + (half * 4 + e) * 16 + ccl];
            float* dst = cbase + (size_t)m * ldc + (size_t)cc * g.perm_hw + hw0;
            if (g.accumulate) { const f32x4 o = *reinterpret_cast<const f32x4*>(dst); v += o; }
            *reinterpret_cast<f32x4*>(dst) = v;
        }
        return;
    }
    if (!split && g.perm_c > 0) {
        // n = n0 + hw_l * 32 + cc (perm_c == 32, n0 % 128 == 0): 4 consecutive hw of channel cc are 16 contiguous bytes of the output row
        const int cc = t & 31, hw0 = n0 >> 5;
        for (int row = t >> 5; row < DN_BM; row += DN_NT / 32) {
            const int m = m0 + row;
            if (m >= g.M) break;
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = T[row * DN_EPI_LD + e * 32 + cc];
            float* dst = cbase + (size_t)m * ldc + (size_t)cc * g.perm_hw + hw0;
            if (g.accumulate) { const f32x4 o = *reinterpret_cast<const f32x4*>(dst); v += o; }
            *reinterpret_cast<f32x4*>(dst) = v;
        }
        return;
    }
    const int nq = t & 31, n = n0 + 4 * nq;
    const bool vec = ((ldc & 3) == 0) && ((reinterpret_cast<uintptr_t>(cbase) & 15) == 0) && (n + 3 < g.N);
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (!split && g.bias) {
#pragma unroll
        for (int e = 0; e < 4; ++e) bv[e] = (n + e < g.N) ? g.bias[n + e] : 0.f;
    }
    for (int row = t >> 5; row < DN_BM; row += DN_NT / 32) {
        const int m = m0 + row;
        if (m >= g.M) break;
        f32x4 v = *reinterpret_cast<const f32x4*>(T + row * DN_EPI_LD + 4 * nq);
        float* dst = cbase + (size_t)m * ldc + n;
        if (!split) {
            v += bv;
            if (g.accumulate) {
                if (vec) v += *reinterpret_cast<const f32x4*>(dst);
                else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (n + e < g.N) v[e] += dst[e];
                }
            }
            if (g.relu) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
            }
        }
        if (vec) *reinterpret_cast<f32x4*>(dst) = v;
        else {
#pragma unroll
            for (int e = 0; e < 4; ++e) if (n + e < g.N) dst[e] = v[e];
        }
    }
}

// Applicability of the dense kernel to C[M][N] = A * B^T with the given operand forms (the callers fall back to igemm otherwise).
inline bool dense_bf3_ok(const DenseArgs& g, bool a_ic, bool b_ic) {
    if (g.M < 256 || g.N < 128 || g.K < 256 || (g.K % DN_BK)) return false;
    if ((g.lda & 3) || (g.ldb & 3) || ((reinterpret_cast<uintptr_t>(g.a) | reinterpret_cast<uintptr_t>(g.b)) & 15)) return false;
    if (g.a_bytes <= 0 || g.b_bytes <= 0 || g.a_bytes >= (1LL << 31) || g.b_bytes >= (1LL << 31)) return false;
    if (a_ic && (g.M & 3)) return false;   // 16-byte loads along the row index
    if (b_ic && (g.N & 3)) return false;
    if (g.perm_c > 0 && (g.perm_c != 32 || (g.N % DN_BN) || g.N != g.perm_c * g.perm_hw || (g.perm_hw & 3) || (g.ldc & 3) ||
                         (reinterpret_cast<uintptr_t>(g.c) & 15))) return false;
    return true;
}

// Split-K plan: forward-form launches with fewer tiles than CUs cut K into equal runs of k-tiles until ~256 work items exist.
inline int dense_bf3_splits(const DenseArgs& g, size_t ws_floats) {
    const int ntiles = cdiv(g.M, DN_BM) * cdiv(g.N, DN_BN), ktiles = g.K / DN_BK;
    if (ntiles >= 128) return 1;
    int s = 256 / ntiles;
    if (s > ktiles / 4) s = ktiles / 4;   // at least 4 k-tiles per item
    while (s > 1 && (size_t)s * g.M * g.N > ws_floats) --s;
    return s < 1 ? 1 : s;
}

template <int A_IC, int B_IC>
inline int dense_bf3_launch(DenseArgs g, hipStream_t stream) {
    auto kern = dense_bf3_kernel<A_IC, B_IC>;
    static const hipError_t attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, DN_LDS_BYTES);
    if (attr_err != hipSuccess) return (int)attr_err;
    static const int sign_schedule = !hab_env_flag("HAB_BF3_NOSIGN");
    g.sign_schedule = sign_schedule;
    static const int skew = hab_env_int("HAB_DENSE_SKEW", 1);
    g.skew = skew;
    static const int ablate = hab_env_int("HAB_DENSE_ABLATE", 0);
    g.ablate = ablate;
    const int ntiles = cdiv(g.M, DN_BM) * cdiv(g.N, DN_BN);
    const int grid = g.nsplit > 1 ? ntiles * ((g.nsplit + 7) / 8 * 8) : ((ntiles + 7) / 8) * 8;
    kern<<<grid, DN_NT, DN_LDS_BYTES, stream>>>(g);
    HAB_LAUNCH_CHECK();
    return HAB_OK;
}

}  // namespace hab
