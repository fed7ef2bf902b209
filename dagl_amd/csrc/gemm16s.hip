// Gradient products of the two patch projections (fc1 / fc2, DN_Gray/model/dagl.py:196-203,248-249) on the fp16 matrix cores
// with split operands.  Under autograd the projections are  Z = rows W^T  (rows = unfolded 7x7x16 patches, [n,784];
// W [196,784]); their backward needs
//     d W    = d Z^T rows        [196 x n] x [n x 784]      (n = 131 072 key patches for a batch of eight 128 x 128 crops)
//     d rows = d Z  W            [n x 196] x [196 x 784]
// 40 GFLOP each and per head -- on the fp32 matrix cores (gemm32.hip, 157 TF peak) 0.75 + 0.5 ms, a fifth of the whole
// training step.  The forward already runs the same contraction with split-fp16 operands at 16x that rate (project16.hip);
// this file does it for the gradients:
//   * every operand is pre-scaled by a power of two and split into two fp16 numbers, s x = hi + lo (>= 21 significant bits; the
//     matrix cores keep fp16 denormals), a product keeps hi*hi + hi*lo + lo*hi in ONE fp32 accumulator (small terms first);
//     activations 16 x and weights 1024 w as in the forward, the gradient d Z by a per-call power of two taken from its
//     largest magnitude (fcg_absmax_kernel -> device word; gradients span many orders of magnitude between steps);
//   * both operands of a product are laid out K-CONTIGUOUS ([rows][K] halfs, K a multiple of 32), so a fragment is one
//     16-byte LDS read and no transposing loader is needed: the producers write what the GEMM wants --
//       d Z  -> [n][224] (K = outputs, for d rows) and, through an LDS transpose, [256][n] (K = patches, for d W),
//       rows -> [896][n] straight from the zero-bordered map (an unfold that transposes: the [n,784] fp32 rows, 411 MB
//               at this size, are never built for the weight gradient),
//       W    -> [896][224] (K = outputs);
//   * gemm16s_kernel: 128 x 128 x 32 (or 256 x 128 x 32) tiles, waves of 64 x 64, operands by LDS-DMA into a two-stage ring (64-byte rows, the
//     four 16-byte slots of a row stored at slot ^ ((row >> 2) & 3): conflict-free ds_read_b128 of 32 consecutive rows, the
//     projection's weight layout), v_mfma_f32_32x32x16_f16, the operands swapped so that a lane ends up with four
//     consecutive columns of one output row (16-byte stores); split-K with a fixed-order reduce for the weight gradient.
// Everything is summed in a fixed order: bit-reproducible.  Range: |16 x| (the forward's own limit) and |1024 w| < 65504.
#include <stdlib.h>

#include "dagl_common.h"

namespace dagl {

typedef _Float16 g16h8 __attribute__((ext_vector_type(8)));

constexpr int G16_BK = 32;                             // K step: rows of 64 bytes in the LDS
constexpr int FCG_O = 196, FCG_OP = 224, FCG_OM = 256; // outputs, padded to the K step / to the M tile
constexpr int FCG_P = 784, FCG_PP = 896;               // patch length, padded to the N tile (7 x 128)
constexpr float FCG_XS = 16.0f, FCG_WS = 1024.0f;      // activation / weight pre-scaling (project16.hip)
#ifdef DAGL_FCG_NO_IMPLICIT                            // (A/B builds: the transposed patch rows as up to round 5)
constexpr bool FCG_NO_IMPLICIT = true;
#else
constexpr bool FCG_NO_IMPLICIT = false;
#endif

// largest |x| of a tensor -> *word (bits of a non-negative float: integer max = float max, order-independent)
__global__ __launch_bounds__(256) void fcg_absmax_kernel(size_t n4, const float4* __restrict__ x, unsigned* __restrict__ word) {
    float m = 0.f;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n4; i += 4 * stride) {                           // four loads in flight per thread
        const float4 v0 = x[i], v1 = x[i + stride], v2 = x[i + 2 * stride], v3 = x[i + 3 * stride];
        const float a = fmaxf(fmaxf(fabsf(v0.x), fabsf(v0.y)), fmaxf(fabsf(v0.z), fabsf(v0.w)));
        const float b = fmaxf(fmaxf(fabsf(v1.x), fabsf(v1.y)), fmaxf(fabsf(v1.z), fabsf(v1.w)));
        const float c = fmaxf(fmaxf(fabsf(v2.x), fabsf(v2.y)), fmaxf(fabsf(v2.z), fabsf(v2.w)));
        const float d = fmaxf(fmaxf(fabsf(v3.x), fabsf(v3.y)), fmaxf(fabsf(v3.z), fabsf(v3.w)));
        m = fmaxf(m, fmaxf(fmaxf(a, b), fmaxf(c, d)));
    }
    for (; i < n4; i += stride) {
        const float4 v = x[i];
        m = fmaxf(m, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    // ONE atomic per block (round 6: one per wave -- 8192 atomics on one address for a 53 MB tensor -- serialised in the L2: ~80 us)
    __shared__ float wm[4];
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]));
        if (m > 0.f) atomicMax(word, __float_as_uint(m));
    }
}

int launch_absmax(hipStream_t s, size_t n, const float* x, unsigned* word) {
    const size_t n4 = n / 4;                      // (n a multiple of 4, x 16-byte aligned: every caller's tensors are rows of 4 k floats)
    if (n4 == 0) return DAGL_OK;
    size_t blocks = (n4 + 256 * 8 - 1) / (256 * 8);
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(fcg_absmax_kernel, dim3((unsigned)blocks), dim3(256), 0, s, n4, reinterpret_cast<const float4*>(x), word);
    DAGL_LAUNCH_CHECK("fcg_absmax_kernel");
    return DAGL_OK;
}

// d Z = d Y (Y > 0) and, in the same pass over it, its largest magnitude (the split's scale) and its column sums (the bias
// gradient): four separate passes over 103 MB per projection before (ReLU backward, maximum, column sums at 1.7 TB/s, split).
// Block = FCG_SROWS rows; thread = (row phase 0..4, float4 column 0..48); per-block sums in fp64, phases added in a fixed
// order, blocks added in a fixed order by col_sum_final_kernel.
constexpr int FCG_SROWS = 160;                        // (640 rows = 205 blocks for 131 072 rows: fewer blocks than CUs, 2.2 TB/s; 160: 4.3)
__global__ __launch_bounds__(256) void fcg_relu_stats_kernel(size_t n, const float4* __restrict__ y, const float4* __restrict__ dy,
                                                             float4* __restrict__ dz, double* __restrict__ part,
                                                             unsigned* __restrict__ max_word) {
    __shared__ double sh[5][FCG_O];
    const int t = threadIdx.x, ph = t / 49, c4 = t - ph * 49;
    const size_t r0 = (size_t)blockIdx.x * FCG_SROWS;
    const size_t r1 = r0 + FCG_SROWS < n ? r0 + FCG_SROWS : n;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    float m = 0.f;
    if (ph < 5) {
#pragma unroll 4
        for (size_t r = r0 + ph; r < r1; r += 5) {
            float4 v = dy[r * 49 + c4];
            if (y != nullptr) {
                const float4 a = y[r * 49 + c4];
                v.x = a.x > 0.f ? v.x : 0.f; v.y = a.y > 0.f ? v.y : 0.f; v.z = a.z > 0.f ? v.z : 0.f; v.w = a.w > 0.f ? v.w : 0.f;
                dz[r * 49 + c4] = v;
            }
            s0 += (double)v.x; s1 += (double)v.y; s2 += (double)v.z; s3 += (double)v.w;
            m = fmaxf(m, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
        }
        sh[ph][4 * c4] = s0; sh[ph][4 * c4 + 1] = s1; sh[ph][4 * c4 + 2] = s2; sh[ph][4 * c4 + 3] = s3;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((t & 63) == 0 && m > 0.f) atomicMax(max_word, __float_as_uint(m));
    __syncthreads();
    if (t < FCG_O) part[(size_t)blockIdx.x * FCG_O + t] = (((sh[0][t] + sh[1][t]) + sh[2][t]) + sh[3][t]) + sh[4][t];
}

// d Z [n, 196] fp32 -> hi / lo [n_pad][224] halfs (pad columns and pad rows zero), scaled by the call's power of two
__global__ __launch_bounds__(256) void fcg_split_rows_kernel(size_t n, size_t n_pad, const float* __restrict__ dz,
                                                             const unsigned* __restrict__ max_word, unsigned short* __restrict__ hi,
                                                             unsigned short* __restrict__ lo) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;          // (row, octet of columns)
    if (t >= n_pad * (FCG_OP / 8)) return;
    const size_t row = t / (FCG_OP / 8); const int c8 = (int)(t - row * (FCG_OP / 8));
    const float s = fcg_scale_of(*max_word);
    unsigned short vh[8], vl[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int c = 8 * c8 + u;
        const float v = (row < n && c < FCG_O) ? dz[row * FCG_O + c] * s : 0.f;
        g16_split(v, vh[u], vl[u]);
    }
    reinterpret_cast<uint4*>(hi)[t] = *reinterpret_cast<const uint4*>(vh);
    reinterpret_cast<uint4*>(lo)[t] = *reinterpret_cast<const uint4*>(vl);
}

// d Z [n, 196] fp32 -> transposed hi / lo [256][n_pad] halfs (rows 196.. and columns n.. zero).  Block = 128 rows x 32
// columns of d Z: coalesced 128-byte row reads, transposed in LDS, 256-byte runs written.
// (round 6) khi / klo non-null: the row-major copy [n_pad][224] of fcg_split_rows_kernel from the same read of d Z (both products of a
// call want it: one pass over the 103 MB instead of two)
// (zt_ld: row stride of the transposed copy, n_pad + 64 halfs -- with n_pad a power of two (131 072 at [8, 128, 128]) the 16 rows of every
// LDS-DMA piece of the weight gradient's first operand sat on one memory channel: dense_train.hip's lesson of round 4)
__global__ __launch_bounds__(256) void fcg_split_transpose_kernel(size_t n, size_t n_pad, size_t zt_ld, const float* __restrict__ dz,
                                                                  const unsigned* __restrict__ max_word,
                                                                  unsigned short* __restrict__ hi, unsigned short* __restrict__ lo,
                                                                  unsigned short* __restrict__ khi, unsigned short* __restrict__ klo) {
    __shared__ __attribute__((aligned(16))) unsigned short th[32][128 + 8];
    __shared__ __attribute__((aligned(16))) unsigned short tl[32][128 + 8];
    __shared__ __attribute__((aligned(16))) unsigned short rh[128][32 + 8];
    __shared__ __attribute__((aligned(16))) unsigned short rl[128][32 + 8];
    const size_t r0 = (size_t)blockIdx.x * 128;
    const int c0 = blockIdx.y * 32;
    const float s = fcg_scale_of(*max_word);
    float v[16];                                                             // (all 16 loads of a thread in flight, then the splits)
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const int e = threadIdx.x + 256 * u, r = e >> 5, c = e & 31;
        v[u] = (r0 + r < n && c0 + c < FCG_O) ? dz[(r0 + r) * FCG_O + c0 + c] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const int e = threadIdx.x + 256 * u, r = e >> 5, c = e & 31;
        g16_split(v[u] * s, th[c][r], tl[c][r]);
        if (khi != nullptr) { rh[r][c] = th[c][r]; rl[r][c] = tl[c][r]; }
    }
    __syncthreads();
    if (khi != nullptr && c0 < FCG_OP) {
        for (int e = threadIdx.x; e < 128 * 4; e += 256) {                    // (row, octet of columns)
            const int r = e >> 2, c8 = e & 3;
            const size_t o = ((r0 + r) * FCG_OP + c0) / 8 + c8;
            reinterpret_cast<uint4*>(khi)[o] = *reinterpret_cast<const uint4*>(&rh[r][8 * c8]);
            reinterpret_cast<uint4*>(klo)[o] = *reinterpret_cast<const uint4*>(&rl[r][8 * c8]);
        }
    }
    for (int e = threadIdx.x; e < 32 * 16; e += 256) {                        // (column, octet of rows)
        const int c = e >> 4, r8 = e & 15;
        const size_t o = ((size_t)(c0 + c) * zt_ld + r0) / 8 + r8;
        reinterpret_cast<uint4*>(hi)[o] = *reinterpret_cast<const uint4*>(&th[c][8 * r8]);
        reinterpret_cast<uint4*>(lo)[o] = *reinterpret_cast<const uint4*>(&tl[c][8 * r8]);
    }
}

// patches of the zero-bordered map [B,Hp,Wp,16], transposed and split: out[(tap, c)][patch] = 16 map[b, oy + py s + kh, ox + px s + kw, c]
// (patch = (b, py, px) row-major; rows 784..895 are zeroed by the caller, columns past the last patch written as zero).
// Block = 128 consecutive patches x one tap: 64-byte pixel reads, transposed in LDS, 256-byte runs written.
__global__ __launch_bounds__(256) void fcg_unfold_transpose_kernel(int Hp, int Wp, int stride, int oy, int ox, int oh, int ow,
                                                                   size_t n, size_t n_pad, const float* __restrict__ map,
                                                                   unsigned short* __restrict__ hi, unsigned short* __restrict__ lo) {
    __shared__ __attribute__((aligned(16))) unsigned short th[16][128 + 8];
    __shared__ __attribute__((aligned(16))) unsigned short tl[16][128 + 8];
    const size_t p0 = (size_t)blockIdx.x * 128;
    const int tap = blockIdx.y, kh = tap / KS, kw = tap - kh * KS;
    for (int e = threadIdx.x; e < 128 * 4; e += 256) {
        const int r = e >> 2, c4 = e & 3;
        const size_t p = p0 + r;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p < n) {
            const size_t b = p / ((size_t)oh * ow); const int q = (int)(p - b * (size_t)oh * ow);
            const int py = q / ow, px = q - py * ow;
            v = *reinterpret_cast<const float4*>(map + ((b * Hp + oy + py * stride + kh) * (size_t)Wp + ox + px * stride + kw) * CH + 4 * c4);
        }
        g16_split(v.x * FCG_XS, th[4 * c4 + 0][r], tl[4 * c4 + 0][r]);
        g16_split(v.y * FCG_XS, th[4 * c4 + 1][r], tl[4 * c4 + 1][r]);
        g16_split(v.z * FCG_XS, th[4 * c4 + 2][r], tl[4 * c4 + 2][r]);
        g16_split(v.w * FCG_XS, th[4 * c4 + 3][r], tl[4 * c4 + 3][r]);
    }
    __syncthreads();
    {
        const int c = threadIdx.x >> 4, r8 = threadIdx.x & 15;               // 16 channels x 16 octets of patches
        const size_t o = ((size_t)(tap * CH + c) * n_pad + p0) / 8 + r8;
        reinterpret_cast<uint4*>(hi)[o] = *reinterpret_cast<const uint4*>(&th[c][8 * r8]);
        reinterpret_cast<uint4*>(lo)[o] = *reinterpret_cast<const uint4*>(&tl[c][8 * r8]);
    }
}

// W [196][784] -> transposed hi / lo [896][224] halfs, 1024 w = hi + lo, pads zero
// (fold_layout: row kk = kh * 128 + kw * 16 + c, kw = 7 a zero pad -- every 128-row tile one kernel row: Gemm16s::fold_part)
__global__ void fcg_weight_transpose_kernel(const float* __restrict__ w, unsigned short* __restrict__ hi, unsigned short* __restrict__ lo,
                                            int fold_layout) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= FCG_PP * FCG_OP) return;
    const int kk = t / FCG_OP, o = t - kk * FCG_OP;
    int src = kk;
    if (fold_layout) {
        const int kh = kk >> 7, kw = (kk >> 4) & 7, c = kk & 15;
        src = (kw < KS) ? (kh * KS + kw) * CH + c : FCG_P;
    }
    const float v = (src < FCG_P && o < FCG_O) ? w[(size_t)o * FCG_P + src] * FCG_WS : 0.f;
    g16_split(v, hi[t], lo[t]);
}

// d map[b, y, x, :] = sum over kh (then the row segments that cover x) of the partial rows gemm16s_kernel's fold_tile wrote; thread = one
// float4 of a map pixel.  Stride-1 patches: patch (py, px) covers map rows oy + py + kh, columns ox + px + kw.
__global__ __launch_bounds__(256) void fcg_fold_kh_kernel(int Hp, int Wp, int oy, int ox, int oh, int ow, int seg, int rows_per_tile,
                                                          const float4* __restrict__ part, float4* __restrict__ dmap) {
    const int b = blockIdx.y;
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)Hp * Wp * 4) return;
    const int pix = (int)(t >> 2), c4 = (int)(t & 3);
    const int y = pix / Wp, x = pix - y * Wp;
    const int tx = x - ox;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const int segw = seg + KS - 1;
    if (tx >= 0 && tx < ow + KS - 1) {
        for (int kh = 0; kh < KS; ++kh) {
            const int ty = y - oy - kh;
            if (ty < 0 || ty >= oh) continue;
            const size_t p0 = ((size_t)b * oh + ty) * ow;                   // first patch of that image row
            for (int s0 = 0; s0 < ow; s0 += seg) {
                const int xs = tx - s0;
                if (xs < 0 || xs >= segw) continue;
                const size_t tile = (p0 + s0) >> 7;
                const int ry = (int)(((p0 + s0) & 127) / seg);
                const float4 v = part[((((tile * KS + kh) * rows_per_tile + ry) * segw + xs) << 2) + c4];
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
        }
    }
    dmap[((size_t)b * Hp * Wp + pix) * 4 + c4] = acc;
}

// planes[kw][c][b Hp + y][x] = 16 map[b, y, x + kw, c] as fp16 hi / lo, x < W (Gemm16s::b_implicit).  Block = one row (b, y) of the map in
// the LDS (Wp pixels x 16 channels); thread = (plane (kw, c), octet of x): 16-byte stores, runs of 2 W bytes per plane
__global__ __launch_bounds__(256) void fcg_shift_planes_kernel(int Hp, int Wp, int W, long long plane, const float* __restrict__ map,
                                                               unsigned short* __restrict__ hi, unsigned short* __restrict__ lo) {
    extern __shared__ float row_s[];                                             // [Wp][16 + 1]
    const int by = blockIdx.x;                                                   // b Hp + y
    const float4* src = reinterpret_cast<const float4*>(map + (size_t)by * Wp * CH);
    for (int e = threadIdx.x; e < Wp * 4; e += 256) {
        const float4 v = src[e];
        float* d = row_s + (e >> 2) * 17 + 4 * (e & 3);
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
    __syncthreads();
    const int octs = W / 8;
    for (int e = threadIdx.x; e < KS * CH * octs; e += 256) {
        const int pl = e / octs, o8 = e - pl * octs;                            // plane = kw * 16 + c
        const int kw = pl >> 4, c = pl & 15;
        unsigned short vh[8], vl[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) g16_split(row_s[(8 * o8 + u + kw) * 17 + c] * FCG_XS, vh[u], vl[u]);
        const size_t o = ((size_t)pl * plane + (size_t)by * W) / 8 + o8;
        reinterpret_cast<uint4*>(hi)[o] = *reinterpret_cast<const uint4*>(vh);
        reinterpret_cast<uint4*>(lo)[o] = *reinterpret_cast<const uint4*>(vl);
    }
}

// C[m][n] = sum_k A[m][k] B[n][k]; block = (64 WM) x (64 WN) outputs by WM x WN waves of 64 x 64, grid.z = K slices.
// The operand stream through LDS-DMA is what bounds this kernel (~23 GB/s per CU measured, hi + lo double the bytes of an
// fp16 GEMM): 256 x 128 tiles fetch 3/4 of the bytes per product of 128 x 128 ones.
// NS = stages of the operand ring: 2 = the tile of step t + 1 is requested at the start of step t and awaited at its end (a step is
// ~0.7 us of multiplies per SIMD: less than the request's latency); 3 = two steps ahead, awaited with a counted vmcnt.
template <int WM, int WN, int NS = 2, bool IMPL = false>
__global__ __launch_bounds__(64 * WM * WN, (WM * WN <= 4 && NS == 2) ? 2 : 1) void gemm16s_kernel(Gemm16s g) {
    constexpr int BM = 64 * WM, BN = 64 * WN, NW = WM * WN;
    constexpr int PA = BM * 64, PB = BN * 64;                                     // bytes of one part (hi or lo) of an operand tile
    constexpr int STAGE = 2 * PA + 2 * PB;
    constexpr int PIECES = STAGE / 1024;
    static_assert(PIECES % NW == 0, "pieces per wave");
    constexpr int PPW = PIECES / NW;                                              // requests per wave and stage
    __shared__ __attribute__((aligned(16))) unsigned char smem[NS * STAGE];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, i = lane & 31, h = lane >> 5;
    // block -> (column tile, row tile, slice): the column tiles of one (row tile, slice) share its A tile -- in dispatch order they sat on
    // 7 different XCDs and each fetched it for itself (round 5 counters on the fc2 backward: 1.53 GB / 1.45 GB of HBM traffic per launch
    // for 0.53 GB of operands + results, profiles/r05_pmc_gemm16.json); remapped so that they run side by side on ONE XCD
    const int lin_id = (int)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z));
    const int logical = xcd_remap2(lin_id, (int)(gridDim.x * gridDim.y * gridDim.z));
    const int bx = logical % (int)gridDim.x, byz = logical / (int)gridDim.x;
    const int by = byz % (int)gridDim.y, bzz = byz / (int)gridDim.y;
    const int nl = g.n_loop > 1 ? g.n_loop : 1;
    const int m0 = by * BM, n0 = bx * nl * BN;
    const int bz = (g.batch > 1) ? bzz / g.slices : 0;                          // batch index; slice = bzz % slices
    const long long k0 = (long long)((g.batch > 1) ? bzz - bz * g.slices : bzz) * g.K;
    const long long boff_a = (long long)bz * g.sA, boff_b = (long long)bz * g.sB;       // (offsets, not pointers: a table of four local
                                                                                        // pointers indexed by the piece number lands in scratch)
    const int wm = wave / WN, wn = wave % WN;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr_of(smem));

    // (IMPL) first patch of the slice -> its row in the planes: b Hp + py  (one scalar division per block)
    int im_row0 = 0;
    if (IMPL) {
        const long long r0 = k0 >> g.im_logw;                                      // image row index b H + py of the slice's first patch
        const int ib = (int)(r0 / g.im_H);
        im_row0 = ib * g.im_Hp + (int)(r0 - (long long)ib * g.im_H);
    }
    // LDS-DMA: a piece = 16 rows x 64 B; lane l -> row l >> 2, physical slot l & 3 holds logical slot (l & 3) ^ ((row >> 2) & 3)
    const int prow = lane >> 2, pslot = (lane & 3) ^ ((prow >> 2) & 3);
    auto piece_src = [&](int p, int nt, long long k) -> const unsigned short* {
            // A hi | A lo | B hi | B lo, 16-row groups
            const bool is_a = p < 2 * (BM / 16);
            const int q = is_a ? p : p - 2 * (BM / 16);
            const int per = is_a ? BM / 16 : BN / 16;
            const int part = q / per, grp = q - part * per;
            const unsigned short* base = is_a ? (part ? g.a_lo : g.a_hi) : (part ? g.b_lo : g.b_hi);
            const long long boff = is_a ? boff_a : boff_b;
            const long long ld = is_a ? g.lda : g.ldb;
            int row = (is_a ? m0 : n0 + nt * BN) + grp * 16 + prow;
            const int lim = (is_a ? g.a_rows : g.b_rows) - 1;
            if (row > lim) row = lim;
            long long kc = k + 8 * pslot;                                         // (k_valid: the piece past the row's end reads the row's zero columns)
            if (g.k_valid > 0 && kc - k0 >= g.k_valid) kc = k0 + g.k_valid - 8;
            const unsigned short* src = base + boff + (long long)row * ld + kc;
            if (IMPL && !is_a) {
                // B row n = (kh, kw, c), contraction index = patch (b, py, px): the shifted plane (kw, c), map row b Hp + py + kh, column px
                const int tap = row >> 4, c = row & 15;
                const int kh = (tap * 37) >> 8, kw = tap - KS * kh;                // (tap / 7 for tap < 49 .. 55)
                const int kl = (int)(kc - k0);                                     // inside the slice: whole image rows of image im_b
                const int px = kl & ((1 << g.im_logw) - 1), dy = kl >> g.im_logw;
                src = base + (long long)(kw * CH + c) * g.im_plane + ((long long)(im_row0 + dy + kh) << g.im_logw) + px;
            }
            return src;
    };
    auto stage = [&](int buf, int nt, long long k) {        // operand tiles of column tile nt of this block, contraction offset k
        const unsigned dst = lds0 + (unsigned)buf * STAGE;
        const int p0 = wave * PPW;                            // a wave's pieces are consecutive: 1 KiB apart in the LDS
#if defined(DAGL_G16_M0_PER_PIECE)
#pragma unroll
        for (int j = 0; j < PPW; ++j)
            glds16_asm(reinterpret_cast<const float*>(piece_src(p0 + j, nt, k)), __builtin_amdgcn_readfirstlane(dst + (unsigned)(p0 + j) * 1024));
#else
        // four requests per M0 value (glds16x4_any_asm): with one M0 value per piece the requests were what a step waited for
#pragma unroll
        for (int j = 0; j + 4 <= PPW; j += 4)
            glds16x4_any_asm(piece_src(p0 + j, nt, k), piece_src(p0 + j + 1, nt, k), piece_src(p0 + j + 2, nt, k), piece_src(p0 + j + 3, nt, k),
                             __builtin_amdgcn_readfirstlane(dst + (unsigned)(p0 + j) * 1024));
        if (PPW % 4 >= 2) {
            constexpr int j = PPW / 4 * 4;
            glds16x2_any_asm(piece_src(p0 + j, nt, k), piece_src(p0 + j + 1, nt, k), __builtin_amdgcn_readfirstlane(dst + (unsigned)(p0 + j) * 1024));
        }
        if (PPW % 2 == 1) {
            constexpr int j = PPW - 1;
            glds16_asm(reinterpret_cast<const float*>(piece_src(p0 + j, nt, k)), __builtin_amdgcn_readfirstlane(dst + (unsigned)(p0 + j) * 1024));
        }
#endif
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int nk = g.K / G16_BK;                                                  // steps of one column tile
    const float alpha = (g.slices > 1) ? 1.0f : g.alpha0 / (fcg_scale_of(g.scale_word ? *g.scale_word : 0u) *
                                                               fcg_scale_of(g.scale_word_b ? *g.scale_word_b : 0u));
    // split-K partials: [slice][batch][M][N] (the batches' outputs must then be dense: sC = M N, ldc = N)
    float* out = (g.slices > 1) ? g.part + ((size_t)(bzz - bz * g.slices) * (g.batch > 1 ? g.batch : 1) + bz) * g.M * g.N
                                : g.C + (long long)bz * g.sC;
    const long long ldo = (g.slices > 1) ? g.N : g.ldc;
    // Results: a lane holds four consecutive columns of ONE row per register quad -- stored from the registers a wave instruction
    // touched 32 rows with 32 bytes each (round 5, -DDAGL_G16_NOSTORE experiment: a third of the d rows product was its 411 MB of
    // results leaving that way).  The wave's 64 x 64 tile goes through the LDS instead (the operand ring is dead behind the last
    // step's barrier: two passes of 32 rows, 8.5 KiB per wave) and leaves as whole 256-byte row segments, four rows per instruction.
    constexpr int SPITCH = 68;                                                    // floats per staged row (64 + 4: rows 8 apart share a bank)
    static_assert(NW * 32 * SPITCH * 4 <= NS * STAGE, "the result staging must fit the operand ring");
    static_assert(BM != 128 || BN != 128 || 128 * 116 * 4 <= NS * STAGE, "the fold's staged tile must fit the operand ring");
    auto store_tile = [&](int nt) {
        float* stg = reinterpret_cast<float*>(smem) + wave * (32 * SPITCH);       // 32 rows x 64 columns at a time (the wave's upper / lower half)
        const int nw = n0 + nt * BN + wn * 64;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<float4*>(stg + i * SPITCH + b * 32 + 8 * q + 4 * h) =
                        make_float4(acc[a][b][4 * q] * alpha, acc[a][b][4 * q + 1] * alpha, acc[a][b][4 * q + 2] * alpha,
                                    acc[a][b][4 * q + 3] * alpha);
            // (the wave reads back only what it wrote itself: LDS operations of a wave execute in order, no barrier)
            const int mw = m0 + wm * 64 + a * 32;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int e = lane + 64 * j;
                const int row = e >> 4, c4 = e & 15;
                const int m = mw + row, n = nw + 4 * c4;
#ifdef DAGL_G16_NOSTORE                       // (timing experiment only: results wrong)
                if (m < g.M && n < g.N && stg[0] == 12345.678f)
#else
                if (m < g.M && n < g.N)                                           // (N is a multiple of 4: whole quads)
#endif
                    {
                        // results larger than the caches (the top-k modes' score rows beyond 64 neighbours: 537 MB per 2048 queries; d rows of the
                        // projections' backward: 411 MB) leave as streaming stores: they are read back from HBM whatever happens, and as ordinary
                        // stores they push the operands' tiles out of the L2 on their way (k = 100: 1.41 -> 1.29 ms, profiles/r05_topk_wide.log)
                        typedef float g16f4 __attribute__((ext_vector_type(4)));
                        const g16f4 v = *reinterpret_cast<const g16f4*>(stg + row * SPITCH + 4 * c4);
                        g16f4* dstp = reinterpret_cast<g16f4*>(out + (long long)m * ldo + n);
                        // (inline assembly: as two IR stores the compiler folds the branches into ONE ordinary store and the hint is gone)
                        if (g.nt_store) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(dstp), "v"(v) : "memory");
                        else *dstp = v;
                    }
            }
        }
    };
    // (Gemm16s::fold_part) column tile nt = kernel row kh of the block's 128 patches: the tile through the LDS (112 real columns at a pitch of
    // 116 floats: 58 KiB of the dead operand ring), then every thread adds its outputs over kw in a fixed order
    constexpr int FPITCH = 116;
    auto fold_tile = [&](int nt) {
        float* T = reinterpret_cast<float*>(smem);
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int col = wn * 64 + b * 32 + 8 * q + 4 * h;
                    if (col < KS * CH)
                        *reinterpret_cast<float4*>(T + (wm * 64 + a * 32 + i) * FPITCH + col) =
                            make_float4(acc[a][b][4 * q] * alpha, acc[a][b][4 * q + 1] * alpha, acc[a][b][4 * q + 2] * alpha,
                                        acc[a][b][4 * q + 3] * alpha);
                }
        __syncthreads();
        const int segw = g.fold_seg + KS - 1;
        const int items = g.fold_rows * segw * 4;
        float4* dst = reinterpret_cast<float4*>(g.fold_part) + ((size_t)by * KS + (bx * nl + nt)) * (size_t)items;     // (column tile = kernel row kh)
        for (int e = tid; e < items; e += 64 * NW) {
            const int c4 = e & 3, rx = e >> 2;
            const int ry = rx / segw, xs = rx - ry * segw;
            float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int kw = 0; kw < KS; ++kw) {
                const int px = xs - kw;
                if (px >= 0 && px < g.fold_seg) {
                    const float4 v = *reinterpret_cast<const float4*>(T + (ry * g.fold_seg + px) * FPITCH + kw * CH + 4 * c4);
                    sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
                }
            }
            dst[e] = sum;
        }
    };
    const int swz = (i >> 2) & 3;
    for (int ntile = 0; ntile < nl; ++ntile) {
        if (ntile > 0) {
            __syncthreads();                                                      // every wave has read its staged results back
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
        }
#pragma unroll
        for (int st = 0; st < NS - 1; ++st)
            if (st < nk) stage(st, ntile, k0 + (long long)st * G16_BK);
        if (NS > 2 && nk > 1) dma_wait_le<(NS - 2) * PPW>(); else dma_wait_all();
        __syncthreads();
        int cur = 0, nxt = NS - 1;                                                // buffer of step t, buffer of step t + NS - 1
        for (int t = 0; t < nk; ++t) {
            if (t + NS - 1 < nk) stage(nxt, ntile, k0 + (long long)(t + NS - 1) * G16_BK);   // (the buffer step t - 1 read: behind that step's barrier)
            const unsigned char* sb = smem + cur * STAGE;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                g16h8 a_hi[2], a_lo[2], b_hi[2], b_lo[2];
                const int off = (((2 * kb + h) ^ swz) << 4);
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int ra = (wm * 64 + u * 32 + i) * 64 + off, rb = (wn * 64 + u * 32 + i) * 64 + off;
                    a_hi[u] = *reinterpret_cast<const g16h8*>(sb + ra);
                    a_lo[u] = *reinterpret_cast<const g16h8*>(sb + PA + ra);
                    b_hi[u] = *reinterpret_cast<const g16h8*>(sb + 2 * PA + rb);
                    b_lo[u] = *reinterpret_cast<const g16h8*>(sb + 2 * PA + PB + rb);
                }
                // D[row = n (first operand's row)][col = m]: a lane holds four consecutive n of one m per register quad
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b_hi[b], a_lo[a], acc[a][b], 0, 0, 0);
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b_lo[b], a_hi[a], acc[a][b], 0, 0, 0);
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b_hi[b], a_hi[a], acc[a][b], 0, 0, 0);
            }
            // step t + 1's tile must have landed; the requests of the steps after it may stay in flight (a wave's requests complete in order)
            if (NS > 2 && t + NS - 1 < nk) dma_wait_le<(NS - 2) * PPW>(); else dma_wait_all();
            __syncthreads();
            cur = (cur + 1 == NS) ? 0 : cur + 1;
            nxt = (nxt + 1 == NS) ? 0 : nxt + 1;
        }
        if (BM == 128 && BN == 128 && g.fold_part != nullptr) fold_tile(ntile);
        else store_tile(ntile);
    }
}

// C[i] = alpha * (part[0][i] + part[1][i] + ...) in slice order
__global__ void gemm16s_reduce_kernel(size_t n4, int slices, const float4* __restrict__ part, float4* __restrict__ C,
                                      const unsigned* __restrict__ scale_word, const unsigned* __restrict__ scale_word_b, float alpha0) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const float alpha = alpha0 / (fcg_scale_of(scale_word ? *scale_word : 0u) * fcg_scale_of(scale_word_b ? *scale_word_b : 0u));
    float4 s = part[i];
    for (int k = 1; k < slices; ++k) {
        const float4 v = part[(size_t)k * n4 + i];
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    C[i] = make_float4(s.x * alpha, s.y * alpha, s.z * alpha, s.w * alpha);
}

int launch_gemm16s(hipStream_t s, const Gemm16s& g) {
    const int nb = g.batch > 1 ? g.batch : 1;
    // (k_valid: rows shorter than the contraction -- the pieces past a row's end re-read its LAST eight columns, which the caller keeps zero)
    DAGL_REQUIRE(g.k_valid == 0 || (g.k_valid >= 8 && g.k_valid % 8 == 0 && g.k_valid < g.K),
                 "gemm16s: k_valid = %d must be a multiple of 8 in [8, K = %d)", g.k_valid, g.K);
    // tile shape by measurement (tools/time_fc_grad.py, n = 131 072): the weight gradient (one 196-row M tile, long K) is
    // faster on 256 x 128 tiles / 8 waves / one block per CU (0.55 against 0.59 ms with its producers), d rows (K = 224:
    // seven steps per block) on 128 x 128 / 4 waves / two blocks per CU (0.42 against 0.49 ms)
    // 256 x 128 tiles / 8 waves / one block per CU with the three-stage ring: the weight gradient (one M tile, split K) and the dense
    // backward's batched products (K >= 512, M >= 1024): 114.5 -> 113.0 ms per adaptive-mode step, 47.6 -> 47.1 top-k; the 128 x 128
    // kernel with three stages (one block per CU instead of two): 123.6.  d rows (K = 224: seven steps per block): 128 x 128.
    const bool small_m = g.M <= 256 && g.slices > 1;
    Gemm16s gl = g;
    gl.nt_store = (size_t)nb * (size_t)(g.slices > 1 ? g.slices : 1) * (size_t)g.M * (size_t)g.N * sizeof(float) >= ((size_t)64 << 20);
    if (!g.fold_part && (small_m || (g.K >= 512 && g.M >= 1024))) {
        dim3 grid((g.N + 127) / 128, (g.M + 255) / 256, g.slices * nb);
        if (g.b_implicit) hipLaunchKernelGGL((gemm16s_kernel<4, 2, 3, true>), grid, dim3(512), 0, s, gl);
        else hipLaunchKernelGGL((gemm16s_kernel<4, 2, 3>), grid, dim3(512), 0, s, gl);
    } else {
        // short contraction, many column tiles (d rows of the patch projections: K = 224, N = 784): a block walks ALL column tiles of
        // its row tile -- one pipeline of 49 steps instead of seven blocks of 7 (each: request latency, 7 steps, 64 KiB of stores)
        const int nt = (g.N + 127) / 128;
        gl.n_loop = (g.slices == 1 && g.K <= 256 && nt > 1 && nt <= 8 && (long long)((g.M + 127) / 128) * nb >= 512) ? nt : 1;
#ifdef DAGL_G16_NLOOP_256
        if (gl.n_loop > 1) {                  // the same walk on 256 x 128 tiles / 8 waves / three stages: 2/3 of the operand bytes per product
            dim3 grid2((nt + gl.n_loop - 1) / gl.n_loop, (g.M + 255) / 256, g.slices * nb);
            hipLaunchKernelGGL((gemm16s_kernel<4, 2, 3>), grid2, dim3(512), 0, s, gl);
        } else
#endif
        {
        dim3 grid((nt + gl.n_loop - 1) / gl.n_loop, (g.M + 127) / 128, g.slices * nb);
        hipLaunchKernelGGL((gemm16s_kernel<2, 2>), grid, dim3(256), 0, s, gl);
        }
    }
    DAGL_LAUNCH_CHECK("gemm16s_kernel");
    if (g.slices > 1) {
        const size_t n4 = (size_t)nb * g.M * g.N / 4;
        hipLaunchKernelGGL(gemm16s_reduce_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, n4, g.slices,
                           reinterpret_cast<const float4*>(g.part), reinterpret_cast<float4*>(g.C), g.scale_word, g.scale_word_b, g.alpha0);
        DAGL_LAUNCH_CHECK("gemm16s_reduce_kernel");
    }
    return DAGL_OK;
}

// ---- the two gradient products of a patch projection -------------------------------------------------------------------
struct FcgPlan {
    size_t n, n_pad;                  // patches of the batch, rounded up to the K granule of the weight gradient
    int slices; size_t k_slice;       // split-K of d W
    size_t o_word, o_zk_hi, o_zk_lo, o_zt_hi, o_zt_lo, o_rt_hi, o_rt_lo, o_wt_hi, o_wt_lo, o_part, o_dz, o_csum, o_fold, o_end;
    int fold_seg, fold_rows;          // > 0: d rows can be folded on the way out (fcg_fold_ok)
    int im_rps;                       // > 0: the weight gradient reads shifted planes of the map (Gemm16s::b_implicit): image rows per K slice
    int stat_blocks;
};

static size_t fcg_carve(size_t& off, size_t bytes) { const size_t o = off; off = align_up(off + bytes, 256); return o; }

// d rows folded inside the product (Gemm16s::fold_part): stride-1 patches whose row tiles of 128 hold whole image rows or one 128-patch
// segment of a row
static bool fcg_fold_ok(int stride, int ow) { return stride == 1 && ow >= 16 && ((ow % 128) == 0 || (128 % ow) == 0); }

// The weight gradient from shifted planes (Gemm16s::b_implicit): stride-1 patches from the map's corner, rows a power of two >= 32 patches
// wide, and a divisor of the image height as the rows of a K slice (a slice = whole rows of one image)
static int fcg_implicit_rows(int B, int stride, int oy, int ox, int oh, int ow, int Hp, int Wp) {
    if (stride != 1 || oy != 0 || ox != 0 || ow < 32 || ow > 512 || (ow & (ow - 1)) != 0 || Hp != oh + KS - 1 || Wp != ow + KS - 1) return 0;   // (<= 512: fcg_shift_planes_kernel keeps a map row in 64 KiB of LDS)
    if (((size_t)B * oh * ow) % 128 != 0) return 0;
    const double want = (double)B * oh / 36.0;                 // ~36 slices x 7 column tiles fill the chip with one block per CU
    int best = 0;
    for (int r = 1; r <= oh; ++r)
        if (oh % r == 0 && (long long)r * ow >= 256 && (best == 0 || (r <= want * 1.5 && r > best))) best = r;
#ifdef DAGL_ABLATION                   // (debug builds: DAGL_FCG_RPS = image rows per K slice of the weight gradient)
    { static const int e = [] { const char* v = getenv("DAGL_FCG_RPS"); return v ? atoi(v) : 0; }(); if (e > 0 && oh % e == 0 && (long long)e * ow >= 256) best = e; }
#endif
    return best;
}

static FcgPlan fcg_plan(size_t n, int ow_fold = 0, int im_rps = 0, int im_ow = 0) {
    FcgPlan p;
    p.n = n;
    p.im_rps = im_rps;
    p.fold_seg = ow_fold > 0 ? (ow_fold < 128 ? ow_fold : 128) : 0;
    p.fold_rows = p.fold_seg > 0 ? 128 / p.fold_seg : 0;
    // d W has 1 x 7 output tiles of 256 x 128: K slices fill the chip (one block per CU: 7 x 36 = 252 blocks), each a multiple
    // of the 32-deep K step
    int slices = 36;
    size_t ks = (n + (size_t)slices - 1) / slices;
    ks = (ks + G16_BK - 1) / G16_BK * G16_BK;
    if (ks < 256) { ks = 256; }
    slices = (int)((n + ks - 1) / ks);
    if (slices < 1) slices = 1;
    if (im_rps > 0) { ks = (size_t)im_rps * im_ow; slices = (int)(n / ks); }      // (whole image rows: n = slices * ks exactly)
    p.slices = slices; p.k_slice = ks; p.n_pad = (size_t)slices * ks;
    if (p.n_pad % 128) p.n_pad = (p.n_pad + 127) / 128 * 128;          // (the transposing producers work in tiles of 128 patches)
    size_t off = 0;
    p.o_word = fcg_carve(off, 256);
    p.o_zk_hi = fcg_carve(off, p.n_pad * FCG_OP * 2); p.o_zk_lo = fcg_carve(off, p.n_pad * FCG_OP * 2);
    p.o_zt_hi = fcg_carve(off, (size_t)FCG_OM * (p.n_pad + 64) * 2); p.o_zt_lo = fcg_carve(off, (size_t)FCG_OM * (p.n_pad + 64) * 2);
    p.o_rt_hi = fcg_carve(off, (size_t)FCG_P * p.n_pad * 2); p.o_rt_lo = fcg_carve(off, (size_t)FCG_P * p.n_pad * 2);
    p.o_wt_hi = fcg_carve(off, (size_t)FCG_PP * FCG_OP * 2); p.o_wt_lo = fcg_carve(off, (size_t)FCG_PP * FCG_OP * 2);
    p.o_part = fcg_carve(off, (size_t)slices * FCG_O * FCG_P * sizeof(float));
    p.o_dz = fcg_carve(off, n * FCG_O * sizeof(float));
    p.stat_blocks = (int)((n + FCG_SROWS - 1) / FCG_SROWS);
    p.o_csum = fcg_carve(off, (size_t)p.stat_blocks * FCG_O * sizeof(double));
    p.o_fold = off;
    if (p.fold_seg > 0)       // [row tiles][7 kh][rows of a tile][seg + 6][16] partial rows
        p.o_fold = fcg_carve(off, (p.n_pad / 128) * KS * (size_t)p.fold_rows * (p.fold_seg + KS - 1) * CH * sizeof(float));
    p.o_end = off;
    return p;
}

}  // namespace dagl

using namespace dagl;

extern "C" {

size_t dagl_fc_grad16_scratch_bytes(int B, int oh, int ow) {
    if (B < 1 || oh < 1 || ow < 1) return 0;
    // (the larger of the two layouts the call may choose: K slices of whole image rows -- shifted planes -- or of ~n / 36 patches)
    const size_t a = fcg_plan((size_t)B * oh * ow).o_end;
    const int rps = fcg_implicit_rows(B, 1, 0, 0, oh, ow, oh + KS - 1, ow + KS - 1);
    const size_t b = rps ? fcg_plan((size_t)B * oh * ow, 0, rps, ow).o_end : 0;
    return a > b ? a : b;
}

// (ABI 405) the same call with the gradient of the MAP as its output instead of the patch rows' (dagl_fc_grad16 + dagl_fold_patches in one):
// stride-1 projections whose rows are 16 / 32 / 64 or a multiple of 128 patches wide fold the rows inside the product
// (Gemm16s::fold_part: [n, 784] rows -- 411 MB at n = 131 072 -- are never written); dagl_fc_grad16_dmap_ok tells whether a geometry qualifies
int dagl_fc_grad16_dmap_ok(int stride, int ow) { return fcg_fold_ok(stride, ow) ? 1 : 0; }
size_t dagl_fc_grad16_dmap_scratch_bytes(int B, int oh, int ow) {
    if (B < 1 || oh < 1 || ow < 1 || !fcg_fold_ok(1, ow)) return 0;
    const size_t a = fcg_plan((size_t)B * oh * ow, ow).o_end;
    const int rps = fcg_implicit_rows(B, 1, 0, 0, oh, ow, oh + KS - 1, ow + KS - 1);
    const size_t b = rps ? fcg_plan((size_t)B * oh * ow, ow, rps, ow).o_end : 0;
    return a > b ? a : b;
}

static int fc_grad16_impl(void* stream, int B, int Hp, int Wp, int stride, int oy, int ox, int oh, int ow, const float* map_nhwc,
                          const float* w_rows, const float* y, const float* dy, float* d_w, float* d_b, float* d_rows, float* d_map,
                          void* scratch, size_t scratch_bytes) {
    DAGL_REQUIRE(B >= 1 && Hp >= 1 && Wp >= 1 && stride >= 1 && oy >= 0 && ox >= 0 && oh >= 1 && ow >= 1 &&
                 oy + (oh - 1) * stride + KS <= Hp && ox + (ow - 1) * stride + KS <= Wp && dy && scratch && (d_w || d_rows || d_b || d_map),
                 "dagl_fc_grad16: bad argument");
    DAGL_REQUIRE((!d_w || map_nhwc) && ((!d_rows && !d_map) || w_rows), "dagl_fc_grad16: d_w needs the map, d_rows / d_map the weight");
    DAGL_REQUIRE(!(d_rows && d_map), "dagl_fc_grad16: d_rows or d_map, not both");
    DAGL_REQUIRE(!d_map || fcg_fold_ok(stride, ow), "dagl_fc_grad16_dmap: stride %d, %d patches per row: not a geometry the product can fold", stride, ow);
    DAGL_REQUIRE(((uintptr_t)scratch % 256) == 0, "dagl_fc_grad16: scratch must be 256-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    const size_t n = (size_t)B * oh * ow;
    const int im_rps = (d_w && !FCG_NO_IMPLICIT) ? fcg_implicit_rows(B, stride, oy, ox, oh, ow, Hp, Wp) : 0;
    const FcgPlan p = fcg_plan(n, d_map ? ow : 0, im_rps, ow);
    DAGL_REQUIRE(scratch_bytes >= p.o_end, "dagl_fc_grad16: scratch %zu B, need %zu B", scratch_bytes, p.o_end);
    char* ws = static_cast<char*>(scratch);
    unsigned* word = reinterpret_cast<unsigned*>(ws + p.o_word);
    auto H = [&](size_t o) { return reinterpret_cast<unsigned short*>(ws + o); };
    DAGL_HIP_TRY(hipMemsetAsync(word, 0, 256, s));
    // one pass over the gradient: ReLU backward (y given), largest magnitude, column sums
    const float* dz = y ? reinterpret_cast<const float*>(ws + p.o_dz) : dy;
    {
        double* csum = reinterpret_cast<double*>(ws + p.o_csum);
        hipLaunchKernelGGL(fcg_relu_stats_kernel, dim3(p.stat_blocks), dim3(256), 0, s, n, reinterpret_cast<const float4*>(y),
                           reinterpret_cast<const float4*>(dy), reinterpret_cast<float4*>(ws + p.o_dz), csum, word);
        DAGL_LAUNCH_CHECK("fcg_relu_stats_kernel");
        if (d_b) { const int rc = launch_col_sum_final(s, p.stat_blocks, FCG_O, csum, d_b); if (rc) return rc; }
    }
    const bool merged_split = (d_rows || d_map) && d_w;          // one pass over d Z writes both of its split copies
    if (merged_split) {
        hipLaunchKernelGGL(fcg_split_transpose_kernel, dim3((unsigned)(p.n_pad / 128), FCG_OM / 32), dim3(256), 0, s, n, p.n_pad, p.n_pad + 64, dz,
                           word, H(p.o_zt_hi), H(p.o_zt_lo), H(p.o_zk_hi), H(p.o_zk_lo));
        DAGL_LAUNCH_CHECK("fcg_split_transpose_kernel");
    }
    if (d_rows || d_map) {
        const size_t items = p.n_pad * (FCG_OP / 8);
        if (!merged_split) {
        hipLaunchKernelGGL(fcg_split_rows_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, s, n, p.n_pad, dz, word,
                           H(p.o_zk_hi), H(p.o_zk_lo));
        DAGL_LAUNCH_CHECK("fcg_split_rows_kernel");
        }
        hipLaunchKernelGGL(fcg_weight_transpose_kernel, dim3((FCG_PP * FCG_OP + 255) / 256), dim3(256), 0, s, w_rows, H(p.o_wt_hi),
                           H(p.o_wt_lo), d_map ? 1 : 0);
        DAGL_LAUNCH_CHECK("fcg_weight_transpose_kernel");
        Gemm16s g;
        g.M = (int)n; g.N = d_map ? FCG_PP : FCG_P; g.K = FCG_OP;
        g.a_hi = H(p.o_zk_hi); g.a_lo = H(p.o_zk_lo); g.lda = FCG_OP; g.a_rows = (int)p.n_pad;
        g.b_hi = H(p.o_wt_hi); g.b_lo = H(p.o_wt_lo); g.ldb = FCG_OP; g.b_rows = FCG_PP;
        g.C = d_map ? reinterpret_cast<float*>(ws + p.o_fold) : d_rows;
        g.ldc = FCG_P; g.part = nullptr; g.slices = 1; g.scale_word = word; g.alpha0 = 1.0f / FCG_WS;
        if (d_map) { g.fold_part = reinterpret_cast<float*>(ws + p.o_fold); g.fold_seg = p.fold_seg; g.fold_rows = p.fold_rows; }
        const int rc = launch_gemm16s(s, g);
        if (rc) return rc;
        if (d_map) {
            const size_t nt = (size_t)Hp * Wp * 4;
            hipLaunchKernelGGL(fcg_fold_kh_kernel, dim3((unsigned)((nt + 255) / 256), B), dim3(256), 0, s, Hp, Wp, oy, ox, oh, ow, p.fold_seg,
                               p.fold_rows, reinterpret_cast<const float4*>(ws + p.o_fold), reinterpret_cast<float4*>(d_map));
            DAGL_LAUNCH_CHECK("fcg_fold_kh_kernel");
        }
    }
    if (d_w) {
        if (!merged_split) {
        hipLaunchKernelGGL(fcg_split_transpose_kernel, dim3((unsigned)(p.n_pad / 128), FCG_OM / 32), dim3(256), 0, s, n, p.n_pad, p.n_pad + 64, dz,
                           word, H(p.o_zt_hi), H(p.o_zt_lo), (unsigned short*)nullptr, (unsigned short*)nullptr);
        DAGL_LAUNCH_CHECK("fcg_split_transpose_kernel");
        }
        int logw = 0;
        while ((1 << logw) < ow) ++logw;
        const long long plane = (long long)B * Hp * ow;
        if (p.im_rps > 0) {
            // (round 6) no transposed patch rows (470 MB at [8, 128, 128]): seven shifted copies of the map's planes (61 MB), which the
            // product addresses by (kw, c, row + kh, px)
            hipLaunchKernelGGL(fcg_shift_planes_kernel, dim3((unsigned)(B * Hp)), dim3(256), (size_t)Wp * 17 * sizeof(float), s, Hp, Wp, ow, plane,
                               map_nhwc, H(p.o_rt_hi), H(p.o_rt_lo));
            DAGL_LAUNCH_CHECK("fcg_shift_planes_kernel");
        } else {
        hipLaunchKernelGGL(fcg_unfold_transpose_kernel, dim3((unsigned)(p.n_pad / 128), KS * KS), dim3(256), 0, s, Hp, Wp, stride, oy,
                           ox, oh, ow, n, p.n_pad, map_nhwc, H(p.o_rt_hi), H(p.o_rt_lo));
        DAGL_LAUNCH_CHECK("fcg_unfold_transpose_kernel");
        }
        // (rows 784..895 of the last N tile do not exist: its loads are clamped to row 783, their products are never stored)
        Gemm16s g;
        g.M = FCG_O; g.N = FCG_P; g.K = (int)p.k_slice;
        g.a_hi = H(p.o_zt_hi); g.a_lo = H(p.o_zt_lo); g.lda = (long long)p.n_pad + 64; g.a_rows = FCG_OM;
        g.b_hi = H(p.o_rt_hi); g.b_lo = H(p.o_rt_lo); g.ldb = (long long)p.n_pad; g.b_rows = FCG_P;
        g.C = d_w; g.ldc = FCG_P; g.part = reinterpret_cast<float*>(ws + p.o_part); g.slices = p.slices; g.scale_word = word;
        g.alpha0 = 1.0f / FCG_XS;
        if (p.im_rps > 0) { g.b_implicit = 1; g.im_logw = logw; g.im_H = oh; g.im_Hp = Hp; g.im_plane = plane; }
        const int rc = launch_gemm16s(s, g);
        if (rc) return rc;
    }
    return DAGL_OK;
}

int dagl_fc_grad16(void* stream, int B, int Hp, int Wp, int stride, int oy, int ox, int oh, int ow, const float* map_nhwc,
                   const float* w_rows, const float* y, const float* dy, float* d_w, float* d_b, float* d_rows, void* scratch,
                   size_t scratch_bytes) {
    return fc_grad16_impl(stream, B, Hp, Wp, stride, oy, ox, oh, ow, map_nhwc, w_rows, y, dy, d_w, d_b, d_rows, nullptr, scratch, scratch_bytes);
}

int dagl_fc_grad16_dmap(void* stream, int B, int Hp, int Wp, int stride, int oy, int ox, int oh, int ow, const float* map_nhwc,
                        const float* w_rows, const float* y, const float* dy, float* d_w, float* d_b, float* d_map, void* scratch,
                        size_t scratch_bytes) {
    return fc_grad16_impl(stream, B, Hp, Wp, stride, oy, ox, oh, ow, map_nhwc, w_rows, y, dy, d_w, d_b, nullptr, d_map, scratch, scratch_bytes);
}

}  // extern "C"
