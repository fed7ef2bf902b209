// Backward of the block's first two convolutions (DN_Gray/model/dagl.py:208-209 under loss.backward(), DN_Gray/trainer.py:48-57),
//     b1 = g(b)      3x3, 64 -> 16, pad 1
//     b2 = theta(b)  1x1, 64 -> 16
// straight on the maps: no patch rows.  The unfold / GEMM / fold route (train_ops.hip + gemm32.hip) writes 300 MB of 3x3 patch
// rows per head at [8,64,128,128], contracts them in a GEMM whose 32 output channels fill a quarter of its tile (225 us),
// writes 300 MB of row gradients (107 us) and folds them (65 us): ~0.6 ms per head with the layout copies around it, twelve
// heads per step.  Here both gradients are tap-wise products on the fp32 matrix cores (v_mfma_f32_16x16x4_f32: exact fp32
// products, fp32 accumulation like the stock convolution's backward):
//
//   weights:  dWg[o][c][ky][kx] = sum_pix x[c][pix + (ky-1, kx-1)] * dB1[pix][o],   dWth[o][c] = sum_pix x[c][pix] * dB2[pix][o]
//             D[m = c][n = o], K = pixels: one block = (image, strip of rows, 16 input channels); its eight waves split the
//             K steps (4 pixels) of a row, each keeps ten accumulators (nine taps + theta); three input rows and one row of
//             output gradients live in LDS rings, the next row is fetched into registers under the multiplies.  Per-block
//             partial sums go to scratch and a second kernel adds them in a fixed order (deterministic; no atomics), together
//             with the bias gradients (column sums of dB1 / dB2, collected on the way by the blocks of channel group 0).
//   input:    dX[c][pix] = sum_{tap, o} Wg[o][c][tap] * dB1[pix - tap][o] + sum_o Wth[o][c] * dB2[pix][o]
//             D[m = c][n = pix], K = o: a wave holds the 40 weight fragments of its 16 channels in registers for its lifetime
//             and walks 16-pixel tiles of a row; three rows of dB1 (one-pixel halo) and one of dB2 in LDS rings; NCHW stores
//             of 16 consecutive pixels per channel.
// Inputs as the autograd graph has them: x NCHW (saved by the forward), dB1 / dB2 as gradients of the zero-bordered NHWC maps
// [B, H+6, W+6, 16] (only the interior is read: the border is a constant of the forward).
#include "dagl_common.h"

namespace dagl {

typedef float f32x4g __attribute__((ext_vector_type(4)));

constexpr int CG_C = 64;                       // input channels
constexpr int CG_O = 16;                       // output channels of each convolution
constexpr int CG_CB = 16;                      // channels per block (weight gradient) / per wave (input gradient)
constexpr int CG_WAVES = 8;
constexpr int CG_THREADS = CG_WAVES * 64;
constexpr int CG_MAX_W = 256;
constexpr int CG_XSLOTS = 4, CG_DSLOTS = 2;
constexpr int CG_ACC = 10;                     // nine taps of g + theta
constexpr int CG_PART = CG_ACC * CG_CB * CG_O; // 2560 partial sums per block
constexpr int CG_RED_FLOATS = 4 * CG_PART;     // cross-wave reduction buffer (40 KiB)

__host__ __device__ inline int cg_xstride(int W) {            // floats per staged channel row: >= W + 2, = 4 mod 64 (the A fragment's
    int s = ((W + 2 + 63) / 64) * 64 + 4;                     // 16 channels x 4 pixels land in 64 different banks)
    if (s - 64 >= W + 2) s -= 64;
    return s;
}

struct ConvGradArgs {
    int B, H, W, rows_per_block, n_strips;
    const float* x;                             // [B, 64, H, W]
    const float* d1; const float* d2;           // [B, H+6, W+6, 16]
    const float* g_w; const float* th_w;        // [16, 64, 3, 3], [16, 64]
    float* part;                                // [B * n_strips][4][CG_PART] (+ bias partials [B * n_strips][32] behind them)
    float* d_x;                                 // [B, 64, H, W]
};

// ---- weight gradient -----------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(CG_THREADS, 2) void conv_pair_wgrad_kernel(ConvGradArgs a) {
    extern __shared__ __attribute__((aligned(16))) float cg_smem[];
    const int W = a.W, H = a.H, Hp = H + 2 * PADPIX, Wp = W + 2 * PADPIX;
    const int S = cg_xstride(W);
    float* const xs = cg_smem;                                          // [CG_XSLOTS][16][S]
    float* const dg = xs + CG_XSLOTS * CG_CB * S;                       // [CG_DSLOTS][W][16]
    float* const dt = dg + CG_DSLOTS * W * CG_O;                        // [CG_DSLOTS][W][16]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int strip = blockIdx.x, cb = blockIdx.y, b = blockIdx.z;
    const int y0 = strip * a.rows_per_block;
    const int y1 = (y0 + a.rows_per_block < H) ? y0 + a.rows_per_block : H;
    const float* xb = a.x + ((size_t)b * CG_C + (size_t)cb * CG_CB) * H * W;
    const float* d1b = a.d1 + ((size_t)b * Hp * Wp + (size_t)PADPIX * Wp + PADPIX) * CG_O;
    const float* d2b = a.d2 + ((size_t)b * Hp * Wp + (size_t)PADPIX * Wp + PADPIX) * CG_O;

    const int xq = W / 4;                          // float4 per channel row
    const int n_x4 = CG_CB * xq;                   // <= 1024: two per thread
    const int n_d4 = W * 4;                        // float4 per gradient row and map: <= 1024, two per thread and map

    // zero the halo columns of every slot once (rows are written into columns 1..W only)
    for (int i = tid; i < CG_XSLOTS * CG_CB; i += CG_THREADS) { xs[i * S] = 0.f; xs[i * S + W + 1] = 0.f; }

    float4 rx[2], rg[2], rt[2];
    auto fetch_x = [&](int y) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int i = tid + j * CG_THREADS;
            rx[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < n_x4 && y >= 0 && y < H) {
                const int c = i / xq, q = i - c * xq;
                rx[j] = *reinterpret_cast<const float4*>(xb + ((size_t)c * H + y) * W + 4 * q);
            }
        }
    };
    auto store_x = [&](int y) {
        float* s = xs + ((y + 1) & (CG_XSLOTS - 1)) * CG_CB * S;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int i = tid + j * CG_THREADS;
            if (i < n_x4) {
                const int c = i / xq, q = i - c * xq;
                float* p = s + c * S + 1 + 4 * q;
                p[0] = rx[j].x; p[1] = rx[j].y; p[2] = rx[j].z; p[3] = rx[j].w;
            }
        }
    };
    auto fetch_d = [&](int y) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int i = tid + j * CG_THREADS;
            rg[j] = make_float4(0.f, 0.f, 0.f, 0.f); rt[j] = rg[j];
            if (i < n_d4 && y < H) {
                rg[j] = *reinterpret_cast<const float4*>(d1b + (size_t)y * Wp * CG_O + 4 * i);
                rt[j] = *reinterpret_cast<const float4*>(d2b + (size_t)y * Wp * CG_O + 4 * i);
            }
        }
    };
    auto store_d = [&](int y) {
        const int s = (y & (CG_DSLOTS - 1)) * W * CG_O;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int i = tid + j * CG_THREADS;
            if (i < n_d4) {
                *reinterpret_cast<float4*>(dg + s + 4 * i) = rg[j];
                *reinterpret_cast<float4*>(dt + s + 4 * i) = rt[j];
            }
        }
    };

    // prologue: input rows y0-1, y0, y0+1 and gradient row y0
    for (int y = y0 - 1; y <= y0 + 1; ++y) { fetch_x(y); store_x(y); }
    fetch_d(y0); store_d(y0);
    __syncthreads();

    f32x4g acc[CG_ACC];
#pragma unroll
    for (int t = 0; t < CG_ACC; ++t) acc[t] = (f32x4g){0.f, 0.f, 0.f, 0.f};
    float bsum_g = 0.f, bsum_t = 0.f;
    const int m = lane & 15, kq = lane >> 4;       // A: channel m, pixel kq of the step; B: output channel m, pixel kq
    const int n_steps = W / 4;

    for (int y = y0; y < y1; ++y) {
        const bool more = y + 1 < y1;
        if (more) { fetch_x(y + 2); fetch_d(y + 1); }
        const float* x0 = xs + ((y + 0) & (CG_XSLOTS - 1)) * CG_CB * S + m * S + kq;      // row y-1 (slot of y-1 is (y-1+1) & 3)
        const float* x1 = xs + ((y + 1) & (CG_XSLOTS - 1)) * CG_CB * S + m * S + kq;
        const float* x2 = xs + ((y + 2) & (CG_XSLOTS - 1)) * CG_CB * S + m * S + kq;
        const float* gp = dg + (y & (CG_DSLOTS - 1)) * W * CG_O + kq * CG_O + m;
        const float* tp = dt + (y & (CG_DSLOTS - 1)) * W * CG_O + kq * CG_O + m;
        for (int st = wave; st < n_steps; st += CG_WAVES) {
            const int p0 = 4 * st;
            const float bg = gp[p0 * CG_O], bt = tp[p0 * CG_O];
            bsum_g += bg; bsum_t += bt;
            // staged column j holds pixel j - 1: tap kx reads pixel p + kx - 1 = column p + kx
            const float a00 = x0[p0], a01 = x0[p0 + 1], a02 = x0[p0 + 2];
            const float a10 = x1[p0], a11 = x1[p0 + 1], a12 = x1[p0 + 2];
            const float a20 = x2[p0], a21 = x2[p0 + 1], a22 = x2[p0 + 2];
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a00, bg, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a01, bg, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a02, bg, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a10, bg, acc[3], 0, 0, 0);
            acc[4] = __builtin_amdgcn_mfma_f32_16x16x4f32(a11, bg, acc[4], 0, 0, 0);
            acc[5] = __builtin_amdgcn_mfma_f32_16x16x4f32(a12, bg, acc[5], 0, 0, 0);
            acc[6] = __builtin_amdgcn_mfma_f32_16x16x4f32(a20, bg, acc[6], 0, 0, 0);
            acc[7] = __builtin_amdgcn_mfma_f32_16x16x4f32(a21, bg, acc[7], 0, 0, 0);
            acc[8] = __builtin_amdgcn_mfma_f32_16x16x4f32(a22, bg, acc[8], 0, 0, 0);
            acc[9] = __builtin_amdgcn_mfma_f32_16x16x4f32(a11, bt, acc[9], 0, 0, 0);
        }
        if (more) { store_x(y + 2); store_d(y + 1); }       // slots of rows y-2 / y-1: nobody reads them in this step
        __syncthreads();
    }

    // cross-wave sums (fixed order): waves 4..7 -> 0..3, 2..3 -> 0..1, 1 -> 0; the rings are dead, the buffer overlays them
    float* red = cg_smem;
    float bias2[2] = {bsum_g, bsum_t};
#pragma unroll
    for (int half = 4; half >= 1; half >>= 1) {
        if (wave >= half && wave < 2 * half) {
            float* r = red + (wave - half) * CG_PART;
#pragma unroll
            for (int t = 0; t < CG_ACC; ++t)
#pragma unroll
                for (int e = 0; e < 4; ++e) r[(t * 4 + e) * 64 + lane] = acc[t][e];
            float* rb = red + CG_RED_FLOATS + (wave - half) * 128;
            rb[lane] = bias2[0]; rb[64 + lane] = bias2[1];
        }
        __syncthreads();
        if (wave < half) {
            const float* r = red + wave * CG_PART;
#pragma unroll
            for (int t = 0; t < CG_ACC; ++t)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[t][e] += r[(t * 4 + e) * 64 + lane];
            const float* rb = red + CG_RED_FLOATS + wave * 128;
            bias2[0] += rb[lane]; bias2[1] += rb[64 + lane];
        }
        __syncthreads();
    }
    if (wave == 0) {
        // D[row = 4 (lane / 16) + e][col = lane % 16]: row = channel within the group, col = output channel
        float* out = a.part + ((size_t)(b * a.n_strips + strip) * 4 + cb) * CG_PART;
#pragma unroll
        for (int t = 0; t < CG_ACC; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) out[(t * CG_CB + 4 * kq + e) * CG_O + m] = acc[t][e];
        if (cb == 0) {
            // bias: the lane's sum covers output channel m over its pixel residue kq: add the four residues
            float sg = bias2[0], st = bias2[1];
            sg += __shfl_xor(sg, 16); sg += __shfl_xor(sg, 32);
            st += __shfl_xor(st, 16); st += __shfl_xor(st, 32);
            float* bo = a.part + (size_t)a.B * a.n_strips * 4 * CG_PART + (size_t)(b * a.n_strips + strip) * 32;
            if (lane < 16) { bo[lane] = sg; bo[16 + lane] = st; }
        }
    }
}

// partial sums -> the parameters' gradients in PyTorch's layouts.  Block = 64 consecutive sums x 4 interleaved quarters of the blocks'
// partials (p = q, q + 4, ..: 256-byte loads), the quarters added in order: the same bits on every call
__global__ __launch_bounds__(256) void conv_pair_wgrad_reduce_kernel(int n_part, const float* __restrict__ part,
                                                                     float* __restrict__ d_g_w, float* __restrict__ d_g_b,
                                                                     float* __restrict__ d_th_w, float* __restrict__ d_th_b) {
    __shared__ float sq[4][64];
    constexpr int NW = 4 * CG_PART;                       // 10240 weight sums, then 32 bias sums
    const int j = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + j;
    float s = 0.f;
    if (i < NW) {
        for (int p = q; p < n_part; p += 4) s += part[(size_t)p * NW + i];
    } else if (i < NW + 32) {
        const float* bp = part + (size_t)n_part * NW + (i - NW);
        for (int p = q; p < n_part; p += 4) s += bp[(size_t)p * 32];
    }
    sq[q][j] = s;
    __syncthreads();
    if (q != 0) return;
    s = ((sq[0][j] + sq[1][j]) + sq[2][j]) + sq[3][j];
    if (i < NW) {
        // i = ((cb * 10 + t) * 16 + cl) * 16 + o
        const int o = i & 15, cl = (i >> 4) & 15, t = (i >> 8) % CG_ACC, cb = (i >> 8) / CG_ACC;
        const int c = cb * CG_CB + cl;
        if (t < 9) d_g_w[(o * CG_C + c) * 9 + t] = s;
        else d_th_w[o * CG_C + c] = s;
    } else if (i < NW + 32) {
        const int b = i - NW;
        if (b < 16) d_g_b[b] = s; else d_th_b[b - 16] = s;
    }
}

// ---- input gradient ------------------------------------------------------------------------------------------------------------
constexpr int CG_DST = 20;                     // staged floats per pixel (16 + 4: a B fragment's 16 pixels x 4 channels hit 64 banks)

__global__ __launch_bounds__(CG_THREADS, 2) void conv_pair_dgrad_kernel(ConvGradArgs a) {
    extern __shared__ __attribute__((aligned(16))) float cg_smem[];
    const int W = a.W, H = a.H, Hp = H + 2 * PADPIX, Wp = W + 2 * PADPIX;
    const int WT = (W + 15) / 16 * 16;                                  // pixel tiles cover WT columns
    const int gs = (WT + 2) * CG_DST;                                   // one staged row of dB1: columns -1 .. WT
    float* const dg = cg_smem;                                          // [4][WT + 2][20]
    float* const dt = dg + 4 * gs;                                      // [2][WT][20]
    const int ts = WT * CG_DST;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int strip = blockIdx.x, b = blockIdx.z;
    const int y0 = strip * a.rows_per_block;
    const int y1 = (y0 + a.rows_per_block < H) ? y0 + a.rows_per_block : H;
    const float* d1b = a.d1 + ((size_t)b * Hp * Wp + (size_t)PADPIX * Wp + PADPIX) * CG_O;
    const float* d2b = a.d2 + ((size_t)b * Hp * Wp + (size_t)PADPIX * Wp + PADPIX) * CG_O;
    const int cbw = wave & 3, half = wave >> 2;                         // the wave's channel group, its half of the pixel tiles
    const int m = lane & 15, kq = lane >> 4;

    // weight fragments: A[m = c][k = o] = W[o][c][tap], o = 4 chunk + kq
    float wg[9][4], wt[4];
    {
        const int c = cbw * CG_CB + m;
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) {
            const int o = 4 * ch + kq;
#pragma unroll
            for (int t = 0; t < 9; ++t) wg[t][ch] = a.g_w[(o * CG_C + c) * 9 + t];
            wt[ch] = a.th_w[o * CG_C + c];
        }
    }

    // zero everything once: halo columns, the columns W..WT-1 and the pad floats stay zero
    for (int i = tid; i < 4 * gs + 2 * ts; i += CG_THREADS) cg_smem[i] = 0.f;
    __syncthreads();

    const int n_d4 = W * 4;                        // float4 per gradient row and map (<= 1024)
    float4 rg[2], rt[2];
    auto fetch_g = [&](int y) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int i = tid + j * CG_THREADS;
            rg[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < n_d4 && y >= 0 && y < H) rg[j] = *reinterpret_cast<const float4*>(d1b + (size_t)y * Wp * CG_O + 4 * i);
        }
    };
    auto store_g = [&](int y) {
        float* s = dg + ((y + 1) & 3) * gs + CG_DST;               // column of pixel 0
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int i = tid + j * CG_THREADS;
            if (i < n_d4) *reinterpret_cast<float4*>(s + (i >> 2) * CG_DST + 4 * (i & 3)) = rg[j];
        }
    };
    auto fetch_t = [&](int y) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int i = tid + j * CG_THREADS;
            rt[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < n_d4 && y < H) rt[j] = *reinterpret_cast<const float4*>(d2b + (size_t)y * Wp * CG_O + 4 * i);
        }
    };
    auto store_t = [&](int y) {
        float* s = dt + (y & 1) * ts;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int i = tid + j * CG_THREADS;
            if (i < n_d4) *reinterpret_cast<float4*>(s + (i >> 2) * CG_DST + 4 * (i & 3)) = rt[j];
        }
    };

    for (int y = y0 - 1; y <= y0 + 1; ++y) { fetch_g(y); store_g(y); }
    fetch_t(y0); store_t(y0);
    __syncthreads();

    const int n_tiles = WT / 16;
    float* const dxb = a.d_x + ((size_t)b * CG_C + (size_t)cbw * CG_CB) * H * W;
    for (int y = y0; y < y1; ++y) {
        const bool more = y + 1 < y1;
        if (more) { fetch_g(y + 2); fetch_t(y + 1); }
        // dX[y] takes dB1 rows y+1 (ky = 0), y (ky = 1), y-1 (ky = 2); row r sits in slot (r + 1) & 3
        const float* r0 = dg + ((y + 2) & 3) * gs;
        const float* r1 = dg + ((y + 1) & 3) * gs;
        const float* r2 = dg + ((y + 0) & 3) * gs;
        const float* rth = dt + (y & 1) * ts;
        for (int tile = half; tile < n_tiles; tile += 2) {
            const int p0 = 16 * tile;
            f32x4g acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};       // two chains: a multiply never waits for its predecessor
            // B[k = o][n = pixel]: staged column (pixel + 1); tap kx reads pixel p - kx + 1 = column p - kx + 2
            const int col = (p0 + m) * CG_DST + kq;
#pragma unroll
            for (int ch = 0; ch < 4; ++ch) {
                const int o4 = 4 * ch;
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int cx = col + (2 - kx) * CG_DST + o4;
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wg[0 * 3 + kx][ch], r0[cx], acc, 0, 0, 0);
                    acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(wg[1 * 3 + kx][ch], r1[cx], acc2, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wg[2 * 3 + kx][ch], r2[cx], acc, 0, 0, 0);
                }
                acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(wt[ch], rth[col + o4], acc2, 0, 0, 0);
            }
            acc += acc2;
            // D[row = 4 kq + e][col = m]: channel 4 kq + e of the group, pixel p0 + m
            if (p0 + m < W) {
#pragma unroll
                for (int e = 0; e < 4; ++e) dxb[((size_t)(4 * kq + e) * H + y) * W + p0 + m] = acc[e];
            }
        }
        if (more) { store_g(y + 2); store_t(y + 1); }
        __syncthreads();
    }
}

static int cg_rows_per_block(int B, int H, int groups) {
    // about two resident blocks per CU over the launch (weights), one (input); strips of 2..32 / 4..32 rows
    const int target = (groups > 1) ? 512 : 256;          // the input gradient's blocks are not split by channel group: one per CU
    int r = (int)(((long long)B * H * groups + target - 1) / target);
    if (r < (groups > 1 ? 2 : 4)) r = (groups > 1) ? 2 : 4;
    if (r > 32) r = 32;
    return r;
}

static size_t cg_wgrad_lds(int W) {
    size_t ring = ((size_t)CG_XSLOTS * CG_CB * cg_xstride(W) + (size_t)2 * CG_DSLOTS * W * CG_O) * sizeof(float);
    const size_t red = ((size_t)CG_RED_FLOATS + 4 * 128) * sizeof(float);
    return ring > red ? ring : red;
}

static size_t cg_dgrad_lds(int W) {
    const int WT = (W + 15) / 16 * 16;
    return ((size_t)4 * (WT + 2) * CG_DST + (size_t)2 * WT * CG_DST) * sizeof(float);
}

}  // namespace dagl

using namespace dagl;

extern "C" {

int dagl_conv_pair_backward_supported(int B, int H, int W) {
    return (B >= 1 && H >= 1 && W >= 4 && (W % 4) == 0 && W <= CG_MAX_W) ? 1 : 0;
}

size_t dagl_conv_pair_backward_scratch_bytes(int B, int H, int W) {
    if (!dagl_conv_pair_backward_supported(B, H, W)) return 0;
    const int rpb = cg_rows_per_block(B, H, 4);
    const size_t n_part = (size_t)B * ((H + rpb - 1) / rpb);
    return n_part * (4 * CG_PART + 32) * sizeof(float);
}

int dagl_conv_pair_backward(void* stream, int B, int H, int W, const float* x, const float* d_b1p, const float* d_b2p,
                            const float* g_w, const float* th_w, float* d_x, float* d_g_w, float* d_g_b, float* d_th_w,
                            float* d_th_b, void* scratch) {
    DAGL_REQUIRE(dagl_conv_pair_backward_supported(B, H, W), "dagl_conv_pair_backward: W must be a multiple of 4 and <= 256");
    DAGL_REQUIRE(x && d_b1p && d_b2p && ((uintptr_t)x % 16) == 0 && ((uintptr_t)d_b1p % 16) == 0 && ((uintptr_t)d_b2p % 16) == 0,
                 "dagl_conv_pair_backward: null or unaligned input");
    const bool want_w = d_g_w != nullptr;
    DAGL_REQUIRE(!want_w || (d_g_b && d_th_w && d_th_b && scratch), "dagl_conv_pair_backward: the four parameter gradients and scratch go together");
    DAGL_REQUIRE(!d_x || (g_w && th_w), "dagl_conv_pair_backward: the input gradient needs the weights");
    hipStream_t s = (hipStream_t)stream;
    ConvGradArgs a = {};
    a.B = B; a.H = H; a.W = W; a.x = x; a.d1 = d_b1p; a.d2 = d_b2p; a.g_w = g_w; a.th_w = th_w; a.d_x = d_x;
    if (want_w) {
        a.rows_per_block = cg_rows_per_block(B, H, 4);
        a.n_strips = (H + a.rows_per_block - 1) / a.rows_per_block;
        a.part = static_cast<float*>(scratch);
        const size_t lds = cg_wgrad_lds(W);
        DAGL_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_pair_wgrad_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(conv_pair_wgrad_kernel, dim3(a.n_strips, 4, B), dim3(CG_THREADS), lds, s, a);
        DAGL_LAUNCH_CHECK("conv_pair_wgrad_kernel");
        const int n_part = B * a.n_strips;
        hipLaunchKernelGGL(conv_pair_wgrad_reduce_kernel, dim3((4 * CG_PART + 32 + 63) / 64), dim3(256), 0, s, n_part,
                           static_cast<const float*>(scratch), d_g_w, d_g_b, d_th_w, d_th_b);
        DAGL_LAUNCH_CHECK("conv_pair_wgrad_reduce_kernel");
    }
    if (d_x) {
        a.rows_per_block = cg_rows_per_block(B, H, 1);
        a.n_strips = (H + a.rows_per_block - 1) / a.rows_per_block;
        const size_t lds = cg_dgrad_lds(W);
        DAGL_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_pair_dgrad_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(conv_pair_dgrad_kernel, dim3(a.n_strips, 1, B), dim3(CG_THREADS), lds, s, a);
        DAGL_LAUNCH_CHECK("conv_pair_dgrad_kernel");
    }
    return DAGL_OK;
}

}  // extern "C"
