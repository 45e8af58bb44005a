// obs_ops.hip -- observation transformers on the device (N4).
//
// Reference: habitat_baselines/common/obs_transformers.py:70-231 (ResizeShortestEdge, CenterCropper) through
// habitat_baselines/utils/common.py:481-557 (image_resize_shortest_edge = F.interpolate(mode "area" | "nearest") on the NCHW
// float view and a cast back to the sensor dtype; center_crop = a slice).  Real ObjectNav sensors are 640x480 and the policy
// wants 256x256: the reference runs permute -> float -> adaptive_avg_pool2d -> cast -> permute -> slice per sensor, six passes
// over HBM and four temporaries.  Here ONE launch reads each source pixel once and writes the cropped, resized NHWC frame
// straight into its destination (a rollout-storage row): only the resized pixels that survive the crop are computed.
//
// Arithmetic is ATen's, so the result is bit-identical to the reference on CPU:
//   area    : window [floor(i*H/h), ceil((i+1)*H/h)) x [floor(j*W/w), ceil((j+1)*W/w)) (integer index math of
//             adaptive_avg_pool2d), fp32 sum in row-major window order, then sum / kh / kw (two divisions, as ATen), then the cast
//             (uint8: truncation);
//   nearest : src = min(int(floorf(dst * float(in) / out)), in - 1)  (upsample_nearest's float scale), plain copy.
// HBM-bound: bytes per frame = source window bytes read once + output bytes written once.
#include <type_traits>

#include "ops.h"
#include "../../include/habitat_amd.h"

namespace hab {

struct ResizeCropArgs {
    const void* src; void* dst;
    int N, H, W, C;        // source NHWC
    int rh, rw;            // extent of the (virtual) resized image
    int y0, x0, oh, ow;    // crop window inside the resized image = output extent
    int mode;              // 0 area, 1 nearest
};

template <class T>
struct PixCast;
template <> struct PixCast<uint8_t> { static __device__ uint8_t from(float v) { return (uint8_t)v; } };
template <> struct PixCast<float>   { static __device__ float from(float v) { return v; } };
template <> struct PixCast<int32_t> { static __device__ int32_t from(float v) { return (int32_t)v; } };

__device__ inline int nearest_src(int dst, int in, int out) {
    if (in == out) return dst;
    if (out == 2 * in) return dst >> 1;
    const float scale = (float)in / (float)out;
    const int s = (int)floorf((float)dst * scale);
    return s < in - 1 ? s : in - 1;
}

#pragma clang fp contract(off)
template <class T, int CMAX>
__global__ void __launch_bounds__(256) obs_resize_crop_kernel(const ResizeCropArgs a) {
    const long long total = (long long)a.N * a.oh * a.ow;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const int ox = (int)(e % a.ow);
        long long r = e / a.ow;
        const int oy = (int)(r % a.oh);
        const int n = (int)(r / a.oh);
        const int ry = oy + a.y0, rx = ox + a.x0;
        const T* img = static_cast<const T*>(a.src) + (size_t)n * a.H * a.W * a.C;
        T* out = static_cast<T*>(a.dst) + ((size_t)e) * a.C;
        if (a.mode == 1) {
            const T* p = img + ((size_t)nearest_src(ry, a.H, a.rh) * a.W + nearest_src(rx, a.W, a.rw)) * a.C;
            for (int c = 0; c < a.C; ++c) out[c] = p[c];
            continue;
        }
        // adaptive_avg_pool2d index math: start = floor(i * in / out), end = ceil((i + 1) * in / out)
        const int ys = (int)(((long long)ry * a.H) / a.rh), ye = (int)((((long long)ry + 1) * a.H + a.rh - 1) / a.rh);
        const int xs = (int)(((long long)rx * a.W) / a.rw), xe = (int)((((long long)rx + 1) * a.W + a.rw - 1) / a.rw);
        float sum[CMAX];
#pragma unroll
        for (int c = 0; c < CMAX; ++c) sum[c] = 0.f;
        for (int y = ys; y < ye; ++y) {
            const T* row = img + ((size_t)y * a.W + xs) * a.C;
            for (int x = xs; x < xe; ++x, row += a.C) {
#pragma unroll
                for (int c = 0; c < CMAX; ++c)
                    if (c < a.C) sum[c] = sum[c] + (float)row[c];
            }
        }
        const float kh = (float)(ye - ys), kw = (float)(xe - xs);
#pragma unroll
        for (int c = 0; c < CMAX; ++c)
            if (c < a.C) out[c] = PixCast<T>::from(__fdiv_rn(__fdiv_rn(sum[c], kh), kw));  // ATen: sum / kh / kw, two IEEE divisions
    }
}

// Area mode, LDS-staged: a workgroup owns an 8 x 32 tile of output pixels of one frame.  The source footprint of the tile (about
// 16 rows x 62 pixels at the 640x480 -> 256 geometry) is copied to LDS with coalesced 32-bit loads (uint8 rgb: 3 loads per thread
// instead of 27 byte loads per output pixel), then every thread sums its window from LDS in ATen's order.
constexpr int RC_TH = 8, RC_TW = 32;
template <class T, int C>
__global__ void __launch_bounds__(256) obs_resize_crop_tile_kernel(const ResizeCropArgs a, int words_max, long long frame_bytes) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const int t = threadIdx.x, tx = t % RC_TW, ty = t / RC_TW;
    const int n = blockIdx.z;
    const int oy0 = blockIdx.y * RC_TH, ox0 = blockIdx.x * RC_TW;
    const int oy1 = min(oy0 + RC_TH, a.oh) - 1, ox1 = min(ox0 + RC_TW, a.ow) - 1;  // last output row / col of the tile
    const int ys0 = (int)(((long long)(oy0 + a.y0) * a.H) / a.rh), ye1 = (int)((((long long)(oy1 + a.y0) + 1) * a.H + a.rh - 1) / a.rh);
    const int xs0 = (int)(((long long)(ox0 + a.x0) * a.W) / a.rw), xe1 = (int)((((long long)(ox1 + a.x0) + 1) * a.W + a.rw - 1) / a.rw);
    const int pitch = a.W * C * (int)sizeof(T);                 // bytes per source row (multiple of 4, checked by the launcher)
    const int b0 = (xs0 * C * (int)sizeof(T)) & ~3;             // staged byte range of a row: [b0, b0 + 4*words)
    const int words = (xe1 * C * (int)sizeof(T) - b0 + 3) >> 2;
    const int nrows = ye1 - ys0;
    const unsigned char* img = static_cast<const unsigned char*>(a.src) + (size_t)n * frame_bytes;
    const long long limit = (long long)(a.N - n) * frame_bytes;  // bytes from `img` to the end of the source buffer
    for (int i = t; i < nrows * words; i += 256) {
        const int r = i / words, wd = i - r * words;
        const long long off = (long long)(ys0 + r) * pitch + b0 + 4 * wd;
        uint32_t v;
        if (off + 4 <= limit) {
            v = *reinterpret_cast<const uint32_t*>(img + off);
        } else {  // the last word of the last row of the last frame may stick out of the buffer
            v = 0;
            for (int b = 0; b < 4; ++b)
                if (off + b < limit) v |= (uint32_t)img[off + b] << (8 * b);
        }
        lds[r * words_max + wd] = v;
    }
    __syncthreads();
    const int oy = oy0 + ty, ox = ox0 + tx;
    if (oy >= a.oh || ox >= a.ow) return;
    const int ry = oy + a.y0, rx = ox + a.x0;
    const int ys = (int)(((long long)ry * a.H) / a.rh), ye = (int)((((long long)ry + 1) * a.H + a.rh - 1) / a.rh);
    const int xs = (int)(((long long)rx * a.W) / a.rw), xe = (int)((((long long)rx + 1) * a.W + a.rw - 1) / a.rw);
    float sum[C];
#pragma unroll
    for (int c = 0; c < C; ++c) sum[c] = 0.f;
    const unsigned char* lb = reinterpret_cast<const unsigned char*>(lds);
    for (int y = ys; y < ye; ++y) {
        const T* row = reinterpret_cast<const T*>(lb + (size_t)(y - ys0) * words_max * 4 + (size_t)xs * C * sizeof(T) - b0);
        for (int x = xs; x < xe; ++x, row += C) {
#pragma unroll
            for (int c = 0; c < C; ++c) sum[c] = sum[c] + (float)row[c];
        }
    }
    const float kh = (float)(ye - ys), kw = (float)(xe - xs);
    T* out = static_cast<T*>(a.dst) + (((size_t)n * a.oh + oy) * a.ow + ox) * C;
#pragma unroll
    for (int c = 0; c < C; ++c) out[c] = PixCast<T>::from(__fdiv_rn(__fdiv_rn(sum[c], kh), kw));
}

template <class T, int C>
static int try_tile_launch(const ResizeCropArgs& a, hipStream_t stream) {
    const long long pitch = (long long)a.W * C * sizeof(T), frame_bytes = pitch * a.H;
    if (pitch % 4 || (reinterpret_cast<uintptr_t>(a.src) & 3) || a.N > 65535) return -1;
    // upper bounds of the tile's source footprint
    const int rows_max = (int)(((long long)RC_TH * a.H + a.rh - 1) / a.rh) + 2;
    const int px_max = (int)(((long long)RC_TW * a.W + a.rw - 1) / a.rw) + 2;
    const int words_max = (int)((px_max * C * sizeof(T) + 3) / 4) + 2;
    const size_t lds_bytes = (size_t)rows_max * words_max * 4;
    if (lds_bytes > 48 * 1024) return -1;
    dim3 grid(cdiv(a.ow, RC_TW), cdiv(a.oh, RC_TH), a.N);
    if (grid.y > 65535) return -1;
    obs_resize_crop_tile_kernel<T, C><<<grid, 256, lds_bytes, stream>>>(a, words_max, frame_bytes);
    HAB_LAUNCH_CHECK();
    return HAB_OK;
}

// uint8 RGB, area mode, windows at most 4 pixels wide (down-scaling by < 3x: the 640x480 -> 256 case has 2-3 pixel windows).
// Integer sums of <= 16 bytes are exact in any order, so the ATen summation order does not matter here: a window row is fetched as
// four aligned dwords, byte-aligned with v_alignbyte, and the three channel sums of the row are three v_dot4_u32_u8 each with
// window-width dependent 0/1 byte weights (byte j*3 + c of the 12-byte row segment belongs to channel c of window pixel j).
// ~45 instructions per output pixel and no LDS / barrier, against ~110 with byte-wise LDS reads.
__global__ void __launch_bounds__(256) obs_resize_crop_rgb8_kernel(const ResizeCropArgs a, long long last_dword) {
    const long long total = (long long)a.N * a.oh * a.ow;
    const uint32_t* src = static_cast<const uint32_t*>(a.src);
    uint8_t* dst = static_cast<uint8_t*>(a.dst);
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const int ox = (int)(e % a.ow);
        long long r_ = e / a.ow;
        const int oy = (int)(r_ % a.oh);
        const int n = (int)(r_ / a.oh);
        const int ry = oy + a.y0, rx = ox + a.x0;
        const int ys = (int)(((long long)ry * a.H) / a.rh), ye = (int)((((long long)ry + 1) * a.H + a.rh - 1) / a.rh);
        const int xs = (int)(((long long)rx * a.W) / a.rw), xe = (int)((((long long)rx + 1) * a.W + a.rw - 1) / a.rw);
        const int kw = xe - xs;
        const uint32_t k1 = kw > 1, k2 = kw > 2, k3 = kw > 3;
        const uint32_t rw0 = 1u | (k1 << 24), rw1 = k2 << 16, rw2 = k3 << 8;
        const uint32_t gw0 = 1u << 8, gw1 = k1 | (k2 << 24), gw2 = k3 << 16;
        const uint32_t bw0 = 1u << 16, bw1 = k1 << 8, bw2 = k2 | (k3 << 24);
        uint32_t sr = 0, sg = 0, sb = 0;
        for (int y = ys; y < ye; ++y) {
            const long long addr = (((long long)n * a.H + y) * a.W + xs) * 3;
            const long long i0 = addr >> 2;
            const uint32_t off = (uint32_t)(addr & 3);
            const uint32_t d0 = src[i0];
            const uint32_t d1 = src[i0 + 1 < last_dword ? i0 + 1 : last_dword];
            const uint32_t d2 = src[i0 + 2 < last_dword ? i0 + 2 : last_dword];
            const uint32_t d3 = src[i0 + 3 < last_dword ? i0 + 3 : last_dword];
            const uint32_t w0 = __builtin_amdgcn_alignbyte(d1, d0, off);
            const uint32_t w1 = __builtin_amdgcn_alignbyte(d2, d1, off);
            const uint32_t w2 = __builtin_amdgcn_alignbyte(d3, d2, off);
            sr = __builtin_amdgcn_udot4(w0, rw0, sr, false); sr = __builtin_amdgcn_udot4(w1, rw1, sr, false); sr = __builtin_amdgcn_udot4(w2, rw2, sr, false);
            sg = __builtin_amdgcn_udot4(w0, gw0, sg, false); sg = __builtin_amdgcn_udot4(w1, gw1, sg, false); sg = __builtin_amdgcn_udot4(w2, gw2, sg, false);
            sb = __builtin_amdgcn_udot4(w0, bw0, sb, false); sb = __builtin_amdgcn_udot4(w1, bw1, sb, false); sb = __builtin_amdgcn_udot4(w2, bw2, sb, false);
        }
        const float kh = (float)(ye - ys), kwf = (float)kw;
        uint8_t* o = dst + (size_t)e * 3;
        o[0] = (uint8_t)__fdiv_rn(__fdiv_rn((float)sr, kh), kwf);
        o[1] = (uint8_t)__fdiv_rn(__fdiv_rn((float)sg, kh), kwf);
        o[2] = (uint8_t)__fdiv_rn(__fdiv_rn((float)sb, kh), kwf);
    }
}

template <class T>
static int launch_resize_crop(const ResizeCropArgs& a, hipStream_t stream) {
    static const bool no_tile = hab_env_flag("HAB_OBS_NO_TILE");
    if constexpr (std::is_same_v<T, uint8_t>) {
        static const bool no_rgb8 = hab_env_flag("HAB_OBS_NO_RGB8");
        const long long bytes = (long long)a.N * a.H * a.W * 3;
        // widest window = ceil(W / rw) + 1 pixels
        if (a.mode == HAB_RESIZE_AREA && a.C == 3 && !no_rgb8 && (a.W + a.rw - 1) / a.rw + 1 <= 4 && bytes % 4 == 0 &&
            (reinterpret_cast<uintptr_t>(a.src) & 3) == 0) {
            const long long total = (long long)a.N * a.oh * a.ow;
            int blocks = (int)cdivl(total, 256);
            if (blocks > 65536) blocks = 65536;
            obs_resize_crop_rgb8_kernel<<<blocks, 256, 0, stream>>>(a, bytes / 4 - 1);
            HAB_LAUNCH_CHECK();
            return HAB_OK;
        }
    }
    if (a.mode == HAB_RESIZE_AREA && !no_tile) {
        int rc = -1;
        if (a.C == 1) rc = try_tile_launch<T, 1>(a, stream);
        else if (a.C == 3) rc = try_tile_launch<T, 3>(a, stream);
        else if (a.C == 4) rc = try_tile_launch<T, 4>(a, stream);
        if (rc >= 0) return rc;
    }
    const long long total = (long long)a.N * a.oh * a.ow;
    int blocks = (int)cdivl(total, 256);
    if (blocks > 16384) blocks = 16384;
    if (a.C <= 4)
        obs_resize_crop_kernel<T, 4><<<blocks, 256, 0, stream>>>(a);
    else
        return HAB_ERR_UNSUPPORTED;
    HAB_LAUNCH_CHECK();
    return HAB_OK;
}

}  // namespace hab

extern "C" int hab_obs_resize_crop(const void* src, void* dst, int dtype, int N, int H, int W, int C, int resized_h, int resized_w,
                                   int crop_y0, int crop_x0, int out_h, int out_w, int mode, hipStream_t stream) {
    using namespace hab;
    if (!src || !dst || N <= 0 || H <= 0 || W <= 0 || C <= 0 || resized_h <= 0 || resized_w <= 0 || out_h <= 0 || out_w <= 0)
        return HAB_ERR_ARG;
    if (crop_y0 < 0 || crop_x0 < 0 || crop_y0 + out_h > resized_h || crop_x0 + out_w > resized_w) return HAB_ERR_ARG;
    if (mode != HAB_RESIZE_AREA && mode != HAB_RESIZE_NEAREST) return HAB_ERR_ARG;
    ResizeCropArgs a{src, dst, N, H, W, C, resized_h, resized_w, crop_y0, crop_x0, out_h, out_w, mode};
    switch (dtype) {
    case HAB_DTYPE_U8: return launch_resize_crop<uint8_t>(a, stream);
    case HAB_DTYPE_F32: return launch_resize_crop<float>(a, stream);
    case HAB_DTYPE_I32: return launch_resize_crop<int32_t>(a, stream);
    default: return HAB_ERR_UNSUPPORTED;
    }
}
