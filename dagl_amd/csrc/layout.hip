// Layout kernels: NCHW -> zero-bordered NHWC maps, fc weight repack, value-row materialisation.
//
// The reference extracts patches with ZeroPad2d + Unfold (same_padding / extract_image_patches,
// DN_Gray/model/dagl.py:123-169) and so writes two 784-float rows per pixel.  Here the maps are
// re-laid once as channels-last with a 3-pixel zero border; every 7x7x16 patch (stride-1 keys/values
// AND the stride-4 SAME-padded queries, whose top/left pad is <= 3) is then 7 contiguous 448-byte
// row segments of that map, addressed on the fly.
#include "dagl_common.h"

namespace dagl {

__global__ void pad_nhwc_kernel(int H, int W, const float* __restrict__ src, float* __restrict__ dst) {
    const int Hp = H + 2 * PADPIX, Wp = W + 2 * PADPIX;
    const int b = blockIdx.z;
    const int yp = blockIdx.y;
    const int xp = blockIdx.x * blockDim.x + threadIdx.x;
    if (xp >= Wp) return;
    const int y = yp - PADPIX, x = xp - PADPIX;
    const bool in = (y >= 0) & (y < H) & (x >= 0) & (x < W);
    const int yc = y < 0 ? 0 : (y >= H ? H - 1 : y), xc = x < 0 ? 0 : (x >= W ? W - 1 : x);
    const float* s = src + (size_t)b * CH * H * W + (size_t)yc * W + xc;
    float v[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) v[c] = s[(size_t)c * H * W];         // unconditional loads (all in flight), zeroed below
#pragma unroll
    for (int c = 0; c < CH; ++c) v[c] = in ? v[c] : 0.0f;
    float4* d = reinterpret_cast<float4*>(dst + (((size_t)b * Hp + yp) * Wp + xp) * CH);
#pragma unroll
    for (int q = 0; q < CH / 4; ++q) d[q] = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
}

int launch_pad_nhwc(hipStream_t s, int B, int H, int W, const float* src, float* dst) {
    const int Wp = W + 2 * PADPIX, Hp = H + 2 * PADPIX;
    dim3 block(128), grid((Wp + 127) / 128, Hp, B);
    hipLaunchKernelGGL(pad_nhwc_kernel, grid, block, 0, s, H, W, src, dst);
    DAGL_LAUNCH_CHECK("pad_nhwc_kernel");
    return DAGL_OK;
}

// rows[b][n][(kh,kw,c)] = b2p[b][y+kh][x+kw][c]   (n = y*W + x): one float4 per thread.
__global__ void unfold_values_kernel(int H, int W, const float* __restrict__ mp, float* __restrict__ rows) {
    const int Wp = W + 2 * PADPIX, Hp = H + 2 * PADPIX;
    const int b = blockIdx.y;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // float4 index within the image
    const size_t per_img = (size_t)H * W * (P / 4);
    if (i >= per_img) return;
    const int n = (int)(i / (P / 4));
    const int r = (int)(i % (P / 4));            // float4 within the row: (kh, kw, c4)
    const int kh = r / (KS * CH / 4), rem = r % (KS * CH / 4);   // 28 float4 per kh
    const int y = n / W, x = n % W;
    const float4* src = reinterpret_cast<const float4*>(mp + (((size_t)b * Hp + y + kh) * Wp + x) * CH) + rem;
    reinterpret_cast<float4*>(rows)[(size_t)b * per_img + i] = *src;
}

int launch_unfold_values(hipStream_t s, int B, const Grid& g, const float* b2p, float* rows) {
    const size_t per_img = (size_t)g.N * (P / 4);
    dim3 grid((unsigned)((per_img + 255) / 256), B);
    hipLaunchKernelGGL(unfold_values_kernel, grid, dim3(256), 0, s, g.H, g.W, b2p, rows);
    DAGL_LAUNCH_CHECK("unfold_values_kernel");
    return DAGL_OK;
}

// ---- zero fills -------------------------------------------------------------------------------------------
// One launch for all the small regions a call has to clear (tail rows of the feature matrices, column sums,
// flags, counters): a hipMemsetAsync each costs ~4.5 us of launch + gap.
__global__ void zero_regions_kernel(ZeroList z) {
    const int r = blockIdx.y;
    if (r >= z.n) return;
    const size_t n16 = z.bytes[r] / 16;
    for (int rep = blockIdx.z; rep < z.reps[r]; rep += gridDim.z) {
        uint4* p = reinterpret_cast<uint4*>(static_cast<char*>(z.ptr[r]) + (size_t)rep * z.stride[r]);
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x)
            p[i] = make_uint4(0u, 0u, 0u, 0u);
    }
}

int launch_zero_regions(hipStream_t s, const ZeroList& z) {
    if (z.n == 0) return DAGL_OK;
    int max_reps = 1;
    for (int i = 0; i < z.n; ++i) {
        if ((reinterpret_cast<uintptr_t>(z.ptr[i]) % 16) != 0 || (z.bytes[i] % 16) != 0 || (z.stride[i] % 16) != 0) {
            set_error("zero_regions: region %d not 16-byte granular", i);
            return DAGL_ERR_INVALID;
        }
        if (z.reps[i] > max_reps) max_reps = z.reps[i];
    }
    if (max_reps > 64) max_reps = 64;
    hipLaunchKernelGGL(zero_regions_kernel, dim3(max_reps > 1 ? 8 : 64, z.n, max_reps), dim3(256), 0, s, z);
    DAGL_LAUNCH_CHECK("zero_regions_kernel");
    return DAGL_OK;
}

// zero the 3-pixel border of two padded NHWC maps [B,Hp,Wp,16]
__global__ void zero_borders_kernel(int H, int W, float* __restrict__ m1, float* __restrict__ m2) {
    const int Hp = H + 2 * PADPIX, Wp = W + 2 * PADPIX;
    const int b = blockIdx.y;
    const int nb = 2 * PADPIX * Wp + 2 * PADPIX * H;                  // border pixels per image
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nb * 4) return;                                          // 4 float4 per pixel
    const int pix = t >> 2, q = t & 3;
    int yp, xp;
    if (pix < 2 * PADPIX * Wp) {                                      // top / bottom bands
        const int r = pix / Wp; xp = pix - r * Wp;
        yp = (r < PADPIX) ? r : (Hp - 2 * PADPIX + r);
    } else {                                                          // left / right bands of the interior rows
        const int e = pix - 2 * PADPIX * Wp;
        const int r = e / (2 * PADPIX), c = e - r * (2 * PADPIX);
        yp = PADPIX + r; xp = (c < PADPIX) ? c : (Wp - 2 * PADPIX + c);
    }
    const size_t o = ((((size_t)b * Hp + yp) * Wp + xp) * CH) / 4 + q;
    reinterpret_cast<float4*>(m1)[o] = make_float4(0.f, 0.f, 0.f, 0.f);
    reinterpret_cast<float4*>(m2)[o] = make_float4(0.f, 0.f, 0.f, 0.f);
}

int launch_zero_borders(hipStream_t s, int B, int H, int W, float* m1, float* m2) {
    const int Wp = W + 2 * PADPIX;
    const int nb = (2 * PADPIX * Wp + 2 * PADPIX * H) * 4;
    hipLaunchKernelGGL(zero_borders_kernel, dim3((nb + 255) / 256, B), dim3(256), 0, s, H, W, m1, m2);
    DAGL_LAUNCH_CHECK("zero_borders_kernel");
    return DAGL_OK;
}

// same for two fp16 maps [B,Hp,Wp,16] (32 bytes per pixel = two 16-byte stores)
__global__ void zero_borders16_kernel(int H, int W, unsigned short* __restrict__ m1, unsigned short* __restrict__ m2) {
    const int Hp = H + 2 * PADPIX, Wp = W + 2 * PADPIX;
    const int b = blockIdx.y;
    const int nb = 2 * PADPIX * Wp + 2 * PADPIX * H;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nb * 2) return;
    const int pix = t >> 1, q = t & 1;
    int yp, xp;
    if (pix < 2 * PADPIX * Wp) {
        const int r = pix / Wp; xp = pix - r * Wp;
        yp = (r < PADPIX) ? r : (Hp - 2 * PADPIX + r);
    } else {
        const int e = pix - 2 * PADPIX * Wp;
        const int r = e / (2 * PADPIX), c = e - r * (2 * PADPIX);
        yp = PADPIX + r; xp = (c < PADPIX) ? c : (Wp - 2 * PADPIX + c);
    }
    const size_t o = ((((size_t)b * Hp + yp) * Wp + xp) * CH) / 8 + q;
    reinterpret_cast<uint4*>(m1)[o] = make_uint4(0u, 0u, 0u, 0u);
    reinterpret_cast<uint4*>(m2)[o] = make_uint4(0u, 0u, 0u, 0u);
}

int launch_zero_borders16(hipStream_t s, int B, int H, int W, uint16_t* m1, uint16_t* m2) {
    const int Wp = W + 2 * PADPIX;
    const int nb = (2 * PADPIX * Wp + 2 * PADPIX * H) * 2;
    hipLaunchKernelGGL(zero_borders16_kernel, dim3((nb + 255) / 256, B), dim3(256), 0, s, H, W, m1, m2);
    DAGL_LAUNCH_CHECK("zero_borders16_kernel");
    return DAGL_OK;
}

}  // namespace dagl
