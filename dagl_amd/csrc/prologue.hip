// Fused prologue of the block: the four convolutions at the top of CE.forward
// (DN_Gray/model/dagl.py:208-215) straight from the 64-channel input,
//     b1 = g(b)      3x3, 64->16, pad 1      (keys + queries)          dagl.py:208
//     b2 = theta(b)  1x1, 64->16             (values)                  dagl.py:209
//     thr  = thr_conv (same_pad(b))  7x7 stride 4, 64->1              dagl.py:212-214
//     bias = bias_conv(same_pad(b))  7x7 stride 4, 64->1              dagl.py:215
// written directly in the layout the rest of the path consumes: b1/b2 as zero-bordered NHWC maps (so the
// pad/transposes of layout.hip disappear), thr/bias as [B,L].  The input is read once.
//
// conv_pair_kernel: one wave = 16 consecutive pixels of a row; v_mfma_f32_16x16x4_f32 with
//   A[pixel][channel]   from a 4-row LDS ring of the input strip (rows y-1..y+1 live, y+2 in flight)
//   B[channel][out]     the 3x3 + 1x1 weights, held in registers for the block's lifetime (160 VGPRs)
// three accumulators (one per kernel row) + one for theta; exact fp32 fma chains like the stock conv.
#include "dagl_common.h"

namespace dagl {

constexpr int PC = 64;                      // input channels
constexpr int PRO_TW = 64;                  // pixels per block row strip (4 waves x 16)
constexpr int PRO_LW = PRO_TW + 2;          // staged pixels per row (1-pixel halo each side)
constexpr int PRO_LS = 68;                  // LDS pixel stride (floats)
constexpr int PRO_ROWS = 4;                 // ring slots

__global__ __launch_bounds__(256) void conv_pair_kernel(int H, int W, int rows_per_block,
                                                        const float* __restrict__ x,
                                                        const float* __restrict__ g_w, const float* __restrict__ g_b,
                                                        const float* __restrict__ th_w, const float* __restrict__ th_b,
                                                        float* __restrict__ b1p, float* __restrict__ b2p,
                                                        unsigned short* __restrict__ b1hi,
                                                        unsigned short* __restrict__ b1lo) {
    __shared__ __attribute__((aligned(16))) float ring[PRO_ROWS][PC][PRO_LS];                // 69.6 KiB
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int i = lane & 15, g = lane >> 4;
    const int b = blockIdx.z;
    const int x0 = blockIdx.x * PRO_TW;
    const int y0 = blockIdx.y * rows_per_block;
    int y1 = y0 + rows_per_block; if (y1 > H) y1 = H;
    const int Hp = H + 2 * PADPIX, Wp = W + 2 * PADPIX;
    const float* xb = x + (size_t)b * PC * H * W;

    // weights -> registers, through LDS: a coalesced block copy of g_w / theta_w into the (not yet used) ring, then
    // every lane picks the 160 values of its B fragments -- lane (n = i, k = g) of MFMA (tap, T) holds
    // w[o = i][c = 4T + g][tap] -- with LDS reads instead of 160 scattered 4-byte global loads.
    float wg[9][16], wt[16];
    {
        float* wl = &ring[0][0][0];                               // 16*64*9 + 16*64 = 10240 floats < ring size
        for (int e = tid; e < 16 * PC * 9 / 4; e += 256)
            reinterpret_cast<float4*>(wl)[e] = reinterpret_cast<const float4*>(g_w)[e];
        for (int e = tid; e < 16 * PC / 4; e += 256)
            reinterpret_cast<float4*>(wl + 16 * PC * 9)[e] = reinterpret_cast<const float4*>(th_w)[e];
        __syncthreads();
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int T = 0; T < 16; ++T) wg[tap][T] = wl[(i * PC + 4 * T + g) * 9 + tap];
#pragma unroll
        for (int T = 0; T < 16; ++T) wt[T] = wl[16 * PC * 9 + i * PC + 4 * T + g];
        __syncthreads();
    }
    const float bias1 = g_b[i], bias2 = th_b[i];

    // staging: thread handles elements idx = tid + 256*j of a [64 ch][66 px] row
    constexpr int NST = (PC * PRO_LW + 255) / 256;              // 17
    float st[NST];
    auto stage_load = [&](int yy) {
#pragma unroll
        for (int j = 0; j < NST; ++j) {
            const int idx = tid + 256 * j;
            const int ch = idx / PRO_LW, p = idx - ch * PRO_LW;
            const int xx = x0 - 1 + p;
            const bool ok = (idx < PC * PRO_LW) && (yy >= 0) && (yy < H) && (xx >= 0) && (xx < W);
            st[j] = ok ? xb[((size_t)ch * H + yy) * W + xx] : 0.f;
        }
    };
    auto stage_store = [&](int yy) {
        const int slot = (yy + 1) & (PRO_ROWS - 1);
#pragma unroll
        for (int j = 0; j < NST; ++j) {
            const int idx = tid + 256 * j;
            const int ch = idx / PRO_LW, p = idx - ch * PRO_LW;
            if (idx < PC * PRO_LW) ring[slot][ch][p] = st[j];
        }
    };

    stage_load(y0 - 1); stage_store(y0 - 1);
    stage_load(y0);     stage_store(y0);
    stage_load(y0 + 1); stage_store(y0 + 1);
    __syncthreads();

    const int px0 = 16 * wave;                                   // first pixel of this wave inside the strip
    for (int y = y0; y < y1; ++y) {
        if (y + 1 < y1) stage_load(y + 2);                       // in flight during the MFMAs below
        f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, at = a0;
#pragma unroll
        for (int T = 0; T < 16; ++T) {
            const int ch = 4 * T + g;
            const float* r0 = &ring[(y + 0) & 3][ch][px0 + i];   // row y-1 lives in slot (y-1+1)&3
            const float* r1 = &ring[(y + 1) & 3][ch][px0 + i];
            const float* r2 = &ring[(y + 2) & 3][ch][px0 + i];
            const float c1 = r1[1];
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(r0[0], wg[0][T], a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(r1[0], wg[3][T], a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(r2[0], wg[6][T], a2, 0, 0, 0);
            at = __builtin_amdgcn_mfma_f32_16x16x4f32(c1, wt[T], at, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(r0[1], wg[1][T], a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(c1, wg[4][T], a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(r2[1], wg[7][T], a2, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(r0[2], wg[2][T], a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(r1[2], wg[5][T], a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(r2[2], wg[8][T], a2, 0, 0, 0);
        }
        // D[row = pixel 4g + r][col = out channel i]
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int xx = x0 + px0 + 4 * g + r;
            if (xx < W) {
                const size_t o = (((size_t)b * Hp + y + PADPIX) * Wp + xx + PADPIX) * CH + i;
                const float v1 = ((a0[r] + a1[r]) + a2[r]) + bias1;
                if (b1p != nullptr) b1p[o] = v1;
                if (b1hi != nullptr) {                           // 16 a = hi + lo, both fp16 (project16.hip)
                    const float vs = v1 * 16.0f;                 // P16_A_SCALE (project16.hip)
                    const _Float16 hh = (_Float16)vs;
                    const _Float16 ll = (_Float16)(vs - (float)hh);
                    b1hi[o] = __builtin_bit_cast(unsigned short, hh);
                    b1lo[o] = __builtin_bit_cast(unsigned short, ll);
                }
                b2p[o] = at[r] + bias2;
            }
        }
        if (y + 1 < y1) stage_store(y + 2);                      // overwrites row y-2's slot: not read any more
        __syncthreads();
    }
}

// thr / bias heads: one wave per query, lane = one of the 49 taps, loop over the 64 channels.
__global__ __launch_bounds__(256) void thr_bias_kernel(Grid gr, const float* __restrict__ x,
                                                       const float* __restrict__ thr_w, const float* __restrict__ thr_b,
                                                       const float* __restrict__ bias_w, const float* __restrict__ bias_b,
                                                       float* __restrict__ thr, float* __restrict__ bias) {
    const int lane = threadIdx.x & 63;
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int b = blockIdx.y;
    if (q >= gr.L) return;
    const int qr = q / gr.Lw, qc = q - qr * gr.Lw;
    const int kh = lane / KS, kw = lane - kh * KS;
    const int yy = QS * qr - gr.pt + kh, xx = QS * qc - gr.pl + kw;
    const bool ok = (lane < KS * KS) && yy >= 0 && yy < gr.H && xx >= 0 && xx < gr.W;   // SAME zero padding
    const float* xp = x + (size_t)b * PC * gr.N + (size_t)(ok ? yy : 0) * gr.W + (ok ? xx : 0);
    const int tap = (lane < KS * KS) ? lane : 0;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll 8
    for (int ch = 0; ch < PC; ++ch) {
        const float v = ok ? xp[(size_t)ch * gr.N] : 0.f;
        s1 = fmaf(v, thr_w[ch * (KS * KS) + tap], s1);
        s2 = fmaf(v, bias_w[ch * (KS * KS) + tap], s2);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
    if (lane == 0) {
        thr[(size_t)b * gr.L + q] = s1 + thr_b[0];
        bias[(size_t)b * gr.L + q] = s2 + bias_b[0];
    }
}

int launch_prologue(hipStream_t s, int B, const Grid& g, const float* x, const float* g_w, const float* g_b,
                    const float* th_w, const float* th_b, const float* thr_w, const float* thr_b,
                    const float* bias_w, const float* bias_b, float* b1p, float* b2p, float* thr, float* bias,
                    uint16_t* b1_hi, uint16_t* b1_lo) {
    int rcz = launch_zero_borders(s, B, g.H, g.W, b1p ? b1p : b2p, b2p);
    if (rcz) return rcz;
    if (b1_hi != nullptr && (rcz = launch_zero_borders16(s, B, g.H, g.W, b1_hi, b1_lo))) return rcz;
    const int strips = (g.W + PRO_TW - 1) / PRO_TW;
    // one block per CU over the whole launch (1 block/CU resident: 332 registers), at least 2 rows per block
    int chunks = (256 + strips * B - 1) / (strips * B);
    if (chunks > (g.H + 1) / 2) chunks = (g.H + 1) / 2;
    if (chunks < 1) chunks = 1;
    const int rows_per_block = (g.H + chunks - 1) / chunks;
    chunks = (g.H + rows_per_block - 1) / rows_per_block;
    hipLaunchKernelGGL(conv_pair_kernel, dim3(strips, chunks, B), dim3(256), 0, s, g.H, g.W, rows_per_block, x, g_w,
                       g_b, th_w, th_b, b1p, b2p, b1_hi, b1_lo);
    DAGL_LAUNCH_CHECK("conv_pair_kernel");
    if (thr != nullptr) {
        hipLaunchKernelGGL(thr_bias_kernel, dim3((g.L + 3) / 4, B), dim3(256), 0, s, g, x, thr_w, thr_b, bias_w,
                           bias_b, thr, bias);
        DAGL_LAUNCH_CHECK("thr_bias_kernel");
    }
    return DAGL_OK;
}

}  // namespace dagl
