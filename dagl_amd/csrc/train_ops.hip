// Stages of the differentiable path that are not matrix products: patch unfold / fold on zero-bordered NHWC maps, layout
// copies, ReLU backward, column sums.  Together with gemm32.hip they make up the prologue convolutions and the two patch
// projections of CE.forward under autograd (DN_Gray/model/dagl.py:208-249):
//     conv / Linear over patches   =  unfold (im2col, element order (kh,kw,c))  ->  rows x weight^T  (+ bias, ReLU)
//     d weight = d Z^T rows,   d rows = d Z weight,   d map = fold(d rows)   (gather form: no atomics, fixed order)
// The inference path never materialises patches (project16.hip, prologue.hip); under autograd the unfolded rows are
// what the weight gradient contracts with, so they are built here (recomputed in the backward, not kept).
#include "dagl_common.h"

namespace dagl {

// rows[b, (py,px), (kh,kw,c)] = map[b, oy + py*stride + kh, ox + px*stride + kw, c];  thread = one float4 of a row
__global__ __launch_bounds__(256) void unfold_patches_kernel(int Hp, int Wp, int C, int k, int stride, int oy, int ox, int oh,
                                                             int ow, const float* __restrict__ map, float* __restrict__ rows) {
    const int b = blockIdx.y;
    const int c4n = C / 4, per_row = k * k * c4n;
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)oh * ow * per_row) return;
    const int patch = (int)(t / per_row), e = (int)(t - (size_t)patch * per_row);
    const int tap = e / c4n, c4 = e - tap * c4n;
    const int kh = tap / k, kw = tap - kh * k;
    const int py = patch / ow, px = patch - py * ow;
    const float4 v = *reinterpret_cast<const float4*>(
        map + (((size_t)b * Hp + oy + py * stride + kh) * Wp + ox + px * stride + kw) * C + 4 * c4);
    reinterpret_cast<float4*>(rows + ((size_t)b * oh * ow + patch) * (size_t)(k * k * C))[e] = v;
}

// dmap[b, y, x, c] = sum over the patches (py,px) and taps (kh,kw) with oy + py*stride + kh = y, ox + px*stride + kw = x of
// drows[b, (py,px), (kh,kw,c)];  thread = one float4 of a map pixel, taps visited in (kh, kw) order
__global__ __launch_bounds__(256) void fold_patches_kernel(int Hp, int Wp, int C, int k, int stride, int oy, int ox, int oh,
                                                           int ow, const float* __restrict__ drows, float* __restrict__ dmap) {
    const int b = blockIdx.y;
    const int c4n = C / 4;
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)Hp * Wp * c4n) return;
    const int pix = (int)(t / c4n), c4 = (int)(t - (size_t)pix * c4n);
    const int y = pix / Wp, x = pix - y * Wp;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const size_t row_len = (size_t)k * k * C;
    for (int kh = 0; kh < k; ++kh) {
        const int ty = y - oy - kh;
        if (ty < 0 || (ty % stride) != 0 || ty / stride >= oh) continue;
        for (int kw = 0; kw < k; ++kw) {
            const int tx = x - ox - kw;
            if (tx < 0 || (tx % stride) != 0 || tx / stride >= ow) continue;
            const float4 v = *reinterpret_cast<const float4*>(
                drows + ((size_t)b * oh * ow + (size_t)(ty / stride) * ow + tx / stride) * row_len + (size_t)(kh * k + kw) * C + 4 * c4);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
    }
    reinterpret_cast<float4*>(dmap + ((size_t)b * Hp * Wp + pix) * C)[c4] = acc;
}

// generic strided 4-D copy dst[i0,i1,i2,i3] = src[i0,i1,i2,i3] (element strides); the fastest index is i3
__global__ void copy4_kernel(int n0, int n1, int n2, int n3, const float* __restrict__ src, long long s0, long long s1,
                             long long s2, long long s3, float* __restrict__ dst, long long d0, long long d1, long long d2,
                             long long d3) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t n = (size_t)n0 * n1 * n2 * n3;
    if (t >= n) return;
    const int i3 = (int)(t % n3); size_t r = t / n3;
    const int i2 = (int)(r % n2); r /= n2;
    const int i1 = (int)(r % n1); const int i0 = (int)(r / n1);
    dst[i0 * d0 + i1 * d1 + i2 * d2 + i3 * d3] = src[i0 * s0 + i1 * s1 + i2 * s2 + i3 * s3];
}

// ---- single-parameter PReLU of the trunk's ResBlocks (common.py:59-79, act = nn.PReLU()) ---------------------------------------
// torch's own backward (unrolled_elementwise_kernel_for_multi_outputs + a reduce) ran at 0.44 TB/s: 24 calls x 307 us + 2.6 ms of
// reductions = 10 ms of the 62 ms training step (profiles/r04_kernel_stats_train.csv, first pass).  Forward y = x > 0 ? x : a x;
// backward dx = dy (x > 0 ? 1 : a), da = sum dy x [x <= 0]: 16-byte accesses, per-block fp64 partials of da, a fixed-order final
// sum (deterministic: the same bits every run).
constexpr int PRELU_BLOCK = 256, PRELU_VEC_PER_THREAD = 8;
__global__ __launch_bounds__(PRELU_BLOCK) void prelu_forward_kernel(size_t n4, const float4* __restrict__ x, const float* __restrict__ a,
                                                                    float4* __restrict__ y) {
    const float s = a[0];
    const size_t base = (size_t)blockIdx.x * (PRELU_BLOCK * PRELU_VEC_PER_THREAD) + threadIdx.x;
#pragma unroll
    for (int u = 0; u < PRELU_VEC_PER_THREAD; ++u) {
        const size_t i = base + (size_t)u * PRELU_BLOCK;
        if (i < n4) {
            const float4 v = x[i];
            y[i] = make_float4(v.x > 0.f ? v.x : s * v.x, v.y > 0.f ? v.y : s * v.y, v.z > 0.f ? v.z : s * v.z, v.w > 0.f ? v.w : s * v.w);
        }
    }
}
__global__ __launch_bounds__(PRELU_BLOCK) void prelu_backward_kernel(size_t n4, const float4* __restrict__ x, const float4* __restrict__ dy,
                                                                     const float* __restrict__ a, float4* __restrict__ dx,
                                                                     double* __restrict__ part) {
    __shared__ double sh[PRELU_BLOCK / 64];
    const float s = a[0];
    const size_t base = (size_t)blockIdx.x * (PRELU_BLOCK * PRELU_VEC_PER_THREAD) + threadIdx.x;
    float acc = 0.f;                                                    // 32 products per thread in fp32, then fp64
#pragma unroll
    for (int u = 0; u < PRELU_VEC_PER_THREAD; ++u) {
        const size_t i = base + (size_t)u * PRELU_BLOCK;
        if (i < n4) {
            const float4 v = x[i], g = dy[i];
            dx[i] = make_float4(v.x > 0.f ? g.x : s * g.x, v.y > 0.f ? g.y : s * g.y, v.z > 0.f ? g.z : s * g.z, v.w > 0.f ? g.w : s * g.w);
            acc += (v.x > 0.f ? 0.f : g.x * v.x) + (v.y > 0.f ? 0.f : g.y * v.y) + (v.z > 0.f ? 0.f : g.z * v.z) + (v.w > 0.f ? 0.f : g.w * v.w);
        }
    }
    const double w = wave_sum_f64((double)acc);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = w;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}
__global__ __launch_bounds__(256) void prelu_final_kernel(int n_part, const double* __restrict__ part, float* __restrict__ da) {
    __shared__ double sh[4];
    double t = 0.0;
    for (int i = threadIdx.x; i < n_part; i += 256) t += part[i];       // fixed assignment of partials to threads
    const double w = wave_sum_f64(t);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = w;
    __syncthreads();
    if (threadIdx.x == 0) da[0] = (float)((sh[0] + sh[1]) + (sh[2] + sh[3]));
}

__global__ void relu_backward_kernel(size_t n, const float* __restrict__ y, const float* __restrict__ dy, float* __restrict__ dz) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dz[i] = y[i] > 0.f ? dy[i] : 0.f;
}

// out[c] = sum_r src[r, c] (bias gradients).  Two fixed-order stages: a block sums CS_ROWS consecutive rows -- thread t owns
// column t % cols and every (256 / cols)-th row of the chunk, so a wave reads consecutive addresses -- into part[block, c];
// the second launch adds the blocks' partials in a fixed order (fp64, one wave per column).  cols <= 256.
constexpr int CS_ROWS = 256;
__global__ __launch_bounds__(256) void col_sum_partial_kernel(size_t rows, int cols, const float* __restrict__ src,
                                                               double* __restrict__ part) {
    __shared__ double sh[256];
    const int groups = 256 / cols;                                     // row phases handled in parallel
    const int c = threadIdx.x % cols, gph = threadIdx.x / cols;
    const size_t r0 = (size_t)blockIdx.x * CS_ROWS;
    const size_t r1 = (r0 + CS_ROWS < rows) ? r0 + CS_ROWS : rows;
    double t = 0.0;
    if (gph < groups)
        for (size_t r = r0 + gph; r < r1; r += groups) t += (double)src[r * cols + c];
    sh[threadIdx.x] = t;
    __syncthreads();
    if (threadIdx.x < cols) {
        double a = 0.0;
        for (int g2 = 0; g2 < groups; ++g2) a += sh[g2 * cols + threadIdx.x];
        part[(size_t)blockIdx.x * cols + threadIdx.x] = a;
    }
}
// one wave per column: lane l adds partials l, l + 64, ... in that order, then a fixed butterfly (a single thread walking
// 512 partials with dependent loads took 93 us per call, 4.5 ms of a training step)
__global__ __launch_bounds__(256) void col_sum_final_kernel(int n_part, int cols, const double* __restrict__ part, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= cols) return;
    double a = 0.0;
    for (int p = lane; p < n_part; p += 64) a += part[(size_t)p * cols + c];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
    if (lane == 0) out[c] = (float)a;
}

int launch_col_sum_final(hipStream_t s, int n_part, int cols, const double* part, float* out) {
    hipLaunchKernelGGL(col_sum_final_kernel, dim3((cols + 3) / 4), dim3(256), 0, s, n_part > 0 ? n_part : 1, cols, part, out);
    DAGL_LAUNCH_CHECK("col_sum_final_kernel");
    return DAGL_OK;
}

int launch_unfold_patches(hipStream_t s, int B, int Hp, int Wp, int C, int k, int stride, int oy, int ox, int oh, int ow,
                          const float* map, float* rows) {
    const size_t n = (size_t)oh * ow * k * k * (C / 4);
    hipLaunchKernelGGL(unfold_patches_kernel, dim3((unsigned)((n + 255) / 256), B), dim3(256), 0, s, Hp, Wp, C, k, stride, oy, ox,
                       oh, ow, map, rows);
    DAGL_LAUNCH_CHECK("unfold_patches_kernel");
    return DAGL_OK;
}

int launch_fold_patches(hipStream_t s, int B, int Hp, int Wp, int C, int k, int stride, int oy, int ox, int oh, int ow,
                        const float* drows, float* dmap) {
    const size_t n = (size_t)Hp * Wp * (C / 4);
    hipLaunchKernelGGL(fold_patches_kernel, dim3((unsigned)((n + 255) / 256), B), dim3(256), 0, s, Hp, Wp, C, k, stride, oy, ox,
                       oh, ow, drows, dmap);
    DAGL_LAUNCH_CHECK("fold_patches_kernel");
    return DAGL_OK;
}

}  // namespace dagl

using namespace dagl;

extern "C" {

static int patch_geometry_ok(int B, int Hp, int Wp, int C, int k, int stride, int oy, int ox, int oh, int ow) {
    return B >= 1 && Hp >= 1 && Wp >= 1 && C >= 4 && (C % 4) == 0 && k >= 1 && stride >= 1 && oy >= 0 && ox >= 0 && oh >= 1 &&
           ow >= 1 && oy + (oh - 1) * stride + k <= Hp && ox + (ow - 1) * stride + k <= Wp;
}

int dagl_unfold_patches(void* stream, int B, int Hp, int Wp, int C, int k, int stride, int oy, int ox, int oh, int ow,
                        const float* map, float* rows) {
    DAGL_REQUIRE(patch_geometry_ok(B, Hp, Wp, C, k, stride, oy, ox, oh, ow) && map && rows, "dagl_unfold_patches: bad argument");
    return launch_unfold_patches((hipStream_t)stream, B, Hp, Wp, C, k, stride, oy, ox, oh, ow, map, rows);
}

int dagl_fold_patches(void* stream, int B, int Hp, int Wp, int C, int k, int stride, int oy, int ox, int oh, int ow,
                      const float* drows, float* dmap) {
    DAGL_REQUIRE(patch_geometry_ok(B, Hp, Wp, C, k, stride, oy, ox, oh, ow) && drows && dmap, "dagl_fold_patches: bad argument");
    return launch_fold_patches((hipStream_t)stream, B, Hp, Wp, C, k, stride, oy, ox, oh, ow, drows, dmap);
}

int dagl_copy4(void* stream, int n0, int n1, int n2, int n3, const float* src, long long s0, long long s1, long long s2,
               long long s3, float* dst, long long d0, long long d1, long long d2, long long d3) {
    DAGL_REQUIRE(n0 >= 1 && n1 >= 1 && n2 >= 1 && n3 >= 1 && src && dst, "dagl_copy4: bad argument");
    const size_t n = (size_t)n0 * n1 * n2 * n3;
    hipLaunchKernelGGL(copy4_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n0, n1, n2, n3, src, s0,
                       s1, s2, s3, dst, d0, d1, d2, d3);
    DAGL_LAUNCH_CHECK("copy4_kernel");
    return DAGL_OK;
}

int dagl_relu_backward(void* stream, size_t n, const float* y, const float* dy, float* dz) {
    DAGL_REQUIRE(y && dy && dz, "dagl_relu_backward: null pointer");
    if (n == 0) return DAGL_OK;
    hipLaunchKernelGGL(relu_backward_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n, y, dy, dz);
    DAGL_LAUNCH_CHECK("relu_backward_kernel");
    return DAGL_OK;
}

size_t dagl_col_sum_scratch_bytes(size_t rows, int cols) {
    return ((rows + CS_ROWS - 1) / CS_ROWS) * (size_t)cols * sizeof(double);
}

int dagl_col_sum(void* stream, size_t rows, int cols, const float* src, float* out, void* scratch) {
    DAGL_REQUIRE(cols >= 1 && cols <= 256 && src && out && scratch, "dagl_col_sum: bad argument (1 <= cols <= 256, scratch required)");
    const int n_part = (int)((rows + CS_ROWS - 1) / CS_ROWS);
    hipLaunchKernelGGL(col_sum_partial_kernel, dim3(n_part > 0 ? n_part : 1), dim3(256), 0, (hipStream_t)stream, rows, cols, src,
                       static_cast<double*>(scratch));
    DAGL_LAUNCH_CHECK("col_sum_partial_kernel");
    hipLaunchKernelGGL(col_sum_final_kernel, dim3((cols + 3) / 4), dim3(256), 0, (hipStream_t)stream, n_part > 0 ? n_part : 1, cols,
                       static_cast<const double*>(scratch), out);
    DAGL_LAUNCH_CHECK("col_sum_final_kernel");
    return DAGL_OK;
}

size_t dagl_prelu_scratch_bytes(size_t n) {
    const size_t n4 = (n + 3) / 4, per = (size_t)PRELU_BLOCK * PRELU_VEC_PER_THREAD;
    return ((n4 + per - 1) / per) * sizeof(double);
}

int dagl_prelu_forward(void* stream, size_t n, const float* x, const float* a, float* y) {
    DAGL_REQUIRE(x && a && y && (n % 4) == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)y % 16) == 0,
                 "dagl_prelu_forward: null pointer, n not a multiple of 4 or unaligned tensors");
    if (n == 0) return DAGL_OK;
    const size_t n4 = n / 4, per = (size_t)PRELU_BLOCK * PRELU_VEC_PER_THREAD;
    hipLaunchKernelGGL(prelu_forward_kernel, dim3((unsigned)((n4 + per - 1) / per)), dim3(PRELU_BLOCK), 0, (hipStream_t)stream, n4,
                       reinterpret_cast<const float4*>(x), a, reinterpret_cast<float4*>(y));
    DAGL_LAUNCH_CHECK("prelu_forward_kernel");
    return DAGL_OK;
}

int dagl_prelu_backward(void* stream, size_t n, const float* x, const float* dy, const float* a, float* dx, float* da, void* scratch) {
    DAGL_REQUIRE(x && dy && a && dx && da && scratch && (n % 4) == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)dy % 16) == 0 &&
                 ((uintptr_t)dx % 16) == 0, "dagl_prelu_backward: null pointer, n not a multiple of 4 or unaligned tensors");
    const size_t n4 = n / 4, per = (size_t)PRELU_BLOCK * PRELU_VEC_PER_THREAD;
    const int n_part = (int)((n4 + per - 1) / per);
    if (n_part > 0) {
        hipLaunchKernelGGL(prelu_backward_kernel, dim3(n_part), dim3(PRELU_BLOCK), 0, (hipStream_t)stream, n4, reinterpret_cast<const float4*>(x),
                           reinterpret_cast<const float4*>(dy), a, reinterpret_cast<float4*>(dx), static_cast<double*>(scratch));
        DAGL_LAUNCH_CHECK("prelu_backward_kernel");
    }
    hipLaunchKernelGGL(prelu_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, n_part, static_cast<const double*>(scratch), da);
    DAGL_LAUNCH_CHECK("prelu_final_kernel");
    return DAGL_OK;
}

}  // extern "C"
