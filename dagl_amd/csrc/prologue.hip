// Fused prologue of the block: the four convolutions at the top of CE.forward
// (DN_Gray/model/dagl.py:208-215) straight from the 64-channel input,
//     b1 = g(b)      3x3, 64->16, pad 1      (keys + queries)          dagl.py:208
//     b2 = theta(b)  1x1, 64->16             (values)                  dagl.py:209
//     thr  = thr_conv (same_pad(b))  7x7 stride 4, 64->1              dagl.py:212-214
//     bias = bias_conv(same_pad(b))  7x7 stride 4, 64->1              dagl.py:215
// written directly in the layout the rest of the path consumes: b1/b2 as zero-bordered NHWC maps (so the
// pad/transposes of layout.hip disappear), thr/bias as [B,L].
//
// conv_pair16_kernel (default path, further down): split-fp16 operands on the 16x16x32 matrix instruction.
// conv_pair_kernel (exact scan): one wave = 16 consecutive pixels of a row; v_mfma_f32_16x16x4_f32 with
//   A[pixel][channel]   from a 4-row LDS ring of the input strip (rows y-1..y+1 live, y+2 in flight)
//   B[channel][out]     the 3x3 + 1x1 weights, held in registers for the block's lifetime (160 VGPRs)
// three accumulators (one per kernel row) + one for theta; exact fp32 fma chains like the stock conv.
#include <stdlib.h>

#include "dagl_common.h"
#include "thr_bias4.h"

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
            const int yc = yy < 0 ? 0 : (yy >= H ? H - 1 : yy), xc = xx < 0 ? 0 : (xx >= W ? W - 1 : xx);
            const int cc = ch < PC ? ch : PC - 1;
            const float raw = xb[((size_t)cc * H + yc) * W + xc];     // unconditional (clamped) load, zeroed afterwards
            st[j] = ok ? raw : 0.f;
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

// ---- the same two convolutions on the fp16 matrix cores with split operands (default path) ---------------------------
// 16 x = hi + lo and 256 w = hi + lo (fp16 pairs, power-of-two pre-scaling keeps the lo parts normal; see project16.hip);
// a product keeps hi*hi + hi*lo + lo*hi in ONE fp32 accumulator.  v_mfma_f32_16x16x32_f16 with A = weights (M = 16 output
// channels) and B = pixels (N = 16 consecutive pixels of a row), K = 32 input channels of one tap: 3 x 3 x 2 K-blocks for
// g, 2 for theta, three MFMAs each: 960 matrix-core cycles per 16 pixels against 5120 of the fp32 kernel above, and no
// weights in registers, so the block starts sooner.  D[row = channel][col = pixel]: a lane ends up with 4 consecutive
// output channels of one pixel -- 8-byte (fp16 maps) and 16-byte (fp32 value map) NHWC stores.
// Input rows are staged transposed, [pixel][hi 64 ch | lo 64 ch] fp16 with a 272-byte pixel stride (17 slots of 16 B:
// the B fragments of 16 consecutive pixels are conflict-free); wave w converts channels 16w..16w+15, one pixel per lane.
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
constexpr int C16_PXB = 272;                             // staged bytes per pixel
constexpr int C16_ROWB = PRO_LW * C16_PXB;               // 17952 B per ring row
constexpr int C16_WOFF = PRO_ROWS * C16_ROWB;            // 71808: weights after the ring
constexpr int C16_WB = (18 + 2) * 16 * 128;              // 40960 B: [K-block][out channel][hi 32 | lo 32]
constexpr float C16_XS = 16.0f, C16_WS = 256.0f;

__device__ __forceinline__ void c16_split(float v, _Float16& hi, _Float16& lo) {
    hi = (_Float16)v;
    lo = (_Float16)(v - (float)hi);
}

// weights of g (3x3) and theta (1x1) as the conv kernel wants them in LDS: [K-block: 18 of g (tap, channel half) + 2 of theta]
// [out channel][hi 32 | lo 32] fp16 with the eight 16-byte slots of a 128-byte row stored at slot ^ ((row >> 1) & 7);
// built once per weight set (launch_pack_conv_weight16) -- every block used to convert all 10240 weights itself.  The last
// word flags weights outside the split-fp16 range.
__global__ __launch_bounds__(256) void pack_conv_weight16_kernel(const float* __restrict__ g_w, const float* __restrict__ th_w,
                                                                 unsigned char* __restrict__ img) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= 16 * PC * 9 + 16 * PC) return;
    float v; int o, c, kb;
    if (e < 16 * PC * 9) {                                   // g_w[o][c][tap]
        o = e / (PC * 9); const int rem = e - o * (PC * 9);
        c = rem / 9; const int tap = rem - c * 9;
        v = g_w[e]; kb = tap * 2 + (c >> 5);
    } else {                                                 // theta_w[o][c]
        const int t = e - 16 * PC * 9;
        o = t / PC; c = t - o * PC;
        v = th_w[t]; kb = 18 + (c >> 5);
    }
    _Float16 h, l;
    c16_split(v * C16_WS, h, l);
    unsigned char* d = img + (kb * 16 + o) * 128 + (c & 7) * 2;
    const int ks = (c & 31) >> 3, gs = (o >> 1) & 7;        // 16-byte slot of the row, swizzled (see the reads)
    *reinterpret_cast<_Float16*>(d + ((ks ^ gs) << 4)) = h;
    *reinterpret_cast<_Float16*>(d + (((4 + ks) ^ gs) << 4)) = l;
    if (!(fabsf(v) * C16_WS < RANGE_LIMIT)) *reinterpret_cast<int32_t*>(img + C16_WB) = 1;
}

int launch_pack_conv_weight16(hipStream_t s, const float* g_w, const float* th_w, unsigned char* img) {
    DAGL_HIP_TRY(hipMemsetAsync(img + C16_WB, 0, 16, s));
    hipLaunchKernelGGL(pack_conv_weight16_kernel, dim3((16 * PC * 9 + 16 * PC + 255) / 256), dim3(256), 0, s, g_w, th_w, img);
    DAGL_LAUNCH_CHECK("pack_conv_weight16_kernel");
    return DAGL_OK;
}

// (hs: per-head input, packed weights and biases; batch index b = head * hs.imgs + image -- a CES stage's four heads are
// one launch: 256 blocks of 16 rows instead of four launches of 256 blocks of 4 rows, each of which loads the 40 KiB of weights)
// (b1p: optional fp32 copy of the key / query map -- the differentiable path's forward, round 6: the map as autograd's tensor; b1hi / b1lo
// may then be null)
__global__ __launch_bounds__(256) void conv_pair16_kernel(int H, int W, int rows_per_block, ConvHeadSet hs,
                                                          float* __restrict__ b2p, float* __restrict__ b1p, unsigned short* __restrict__ b1hi,
                                                          unsigned short* __restrict__ b1lo, uint32_t* __restrict__ clear_a,
                                                          int clear_a_words, uint32_t* __restrict__ clear_b, int clear_b_words,
                                                          RangeTag range, unsigned long long* times) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[C16_WOFF + C16_WB + 16];      // 110 KiB (+ the out-of-range flag)
    int* const oor = reinterpret_cast<int*>(smem + C16_WOFF + C16_WB);      // "some input value of the strip left the scale's range"
    const int tid = threadIdx.x;
    const int blin = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;       // (phase stamps of ablation builds)
    dbg_stamp(times, blin, 0);
    float amax = 0.f;                                      // largest |pre-scaled operand| this thread splits (range guard)
    if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0) {      // per-call counters / flags of the later stages
        for (int t = tid; t < clear_a_words; t += 256) clear_a[t] = 0u;
        for (int t = tid; t < clear_b_words; t += 256) clear_b[t] = 0u;
    }
    if (tid == 0) *oor = 0;                               // (read after the strip's barriers)
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int i = lane & 15, kg = lane >> 4;
    const int b = blockIdx.z;
    const int head = b / hs.imgs;
    const float* __restrict__ x = hs.x[head];
    const unsigned char* __restrict__ wimg = hs.w[head];
    const float* __restrict__ g_b = hs.gb[head];
    const float* __restrict__ th_b = hs.tb[head];
    const int x0 = blockIdx.x * PRO_TW;
    const int y0 = blockIdx.y * rows_per_block;
    int y1 = y0 + rows_per_block; if (y1 > H) y1 = H;
    const int Hp = H + 2 * PADPIX, Wp = W + 2 * PADPIX;
    const float* xb = x + (size_t)(b - head * hs.imgs) * PC * H * W;

    // staging: wave w owns channels 16w..16w+15 of staged pixels 0..63 (lane = pixel: coalesced row loads, one 16-byte
    // LDS store per 8 channels); the two halo pixels 64, 65 x 64 channels are one value each for threads 0..127
    const int hpx = 64 + (tid >> 6), hch = tid & 63;           // halo duty of threads 0..127
    auto load_row = [&](int yy, float* st, float& sh) {
        // unconditional loads from clamped addresses, zeroed afterwards: a predicated load makes the compiler wait for
        // each one before issuing the next
        const bool rok = (yy >= 0) && (yy < H);
        const int xa = x0 - 1 + lane, xc = x0 - 1 + hpx;
        const bool oka = rok && xa >= 0 && xa < W;
        const bool okc = rok && xc < W;
        const int yc = yy < 0 ? 0 : (yy >= H ? H - 1 : yy);
        const int xac = xa < 0 ? 0 : (xa >= W ? W - 1 : xa);
        const float* rp = xb + ((size_t)(16 * wave) * H + yc) * W + xac;
        float raw[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) raw[c] = rp[(size_t)c * H * W];
        const float rh = xb[((size_t)hch * H + yc) * W + (xc >= W ? W - 1 : xc)];
#pragma unroll
        for (int c = 0; c < 16; ++c) st[c] = oka ? raw[c] : 0.f;
        sh = okc ? rh : 0.f;
    };
    float xs = C16_XS;                                     // the block's input scale (see B1Tiers, dagl_common.h): 16, or 2^e on the second attempt
    float amax_x = 0.f;                                    // largest |x| this thread staged (unscaled; inf stays, a NaN is skipped here and
                                                           // shows up in the outputs it poisons: b1_bits below)
    auto store_row = [&](int yy, const float* st, float sh) {
        unsigned char* row = smem + ((yy + 1) & (PRO_ROWS - 1)) * C16_ROWB;
        unsigned char* px = row + lane * C16_PXB;
        h16x8 hi0, hi1, lo0, lo1;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            _Float16 h, l;
            c16_split(st[c] * xs, h, l); hi0[c] = h; lo0[c] = l;
            c16_split(st[c + 8] * xs, h, l); hi1[c] = h; lo1[c] = l;
            amax_x = fmaxf(amax_x, fmaxf(fabsf(st[c]), fabsf(st[c + 8])));
        }
        *reinterpret_cast<h16x8*>(px + 32 * wave) = hi0;
        *reinterpret_cast<h16x8*>(px + 32 * wave + 16) = hi1;
        *reinterpret_cast<h16x8*>(px + 128 + 32 * wave) = lo0;
        *reinterpret_cast<h16x8*>(px + 128 + 32 * wave + 16) = lo1;
        if (tid < 128) {
            _Float16 h, l;
            c16_split(sh * xs, h, l);
            amax_x = fmaxf(amax_x, fabsf(sh));
            *reinterpret_cast<_Float16*>(row + hpx * C16_PXB + 2 * hch) = h;
            *reinterpret_cast<_Float16*>(row + hpx * C16_PXB + 128 + 2 * hch) = l;
        }
    };

    // prologue: the three first rows and the weights are requested together (one memory round trip), then converted
    float st0[16], st1[16], st2[16], sh0, sh1, sh2;
    unsigned b1_bits = 0u;                                 // largest |b1| this thread wrote, as bits (B1Tiers::amax): inf and NaN rank above
                                                           // every finite value, so one integer maximum carries all three
    const bool tiers = hs.tiers.amax != nullptr;           // (else: the fixed scales and the range word, as up to round 5)
    unsigned short* __restrict__ b1hi2 = hs.tiers.hi2;
    unsigned short* __restrict__ b1lo2 = hs.tiers.lo2;
    bool first = true;
    auto run_strip = [&]() {
    load_row(y0 - 1, st0, sh0);
    load_row(y0, st1, sh1);
    load_row(y0 + 1, st2, sh2);
    if (first) {
        // the packed weight image: 40 KiB = 10 x 16 B per thread, straight into LDS
        uint4 wv[10];
#pragma unroll
        for (int j = 0; j < 10; ++j) wv[j] = reinterpret_cast<const uint4*>(wimg)[tid + 256 * j];
        if (tid == 0 && *reinterpret_cast<const int32_t*>(wimg + C16_WB) != 0) amax = __builtin_inff();   // weights out of range
#pragma unroll
        for (int j = 0; j < 10; ++j) reinterpret_cast<uint4*>(smem + C16_WOFF)[tid + 256 * j] = wv[j];
    }
    store_row(y0 - 1, st0, sh0);
    store_row(y0, st1, sh1);
    store_row(y0 + 1, st2, sh2);
    __syncthreads();

    float bg[4], bt[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { bg[r] = g_b[4 * kg + r]; bt[r] = th_b[4 * kg + r]; }
    // A: row = output channel i, k = 8 kg..; the eight 16-byte slots of a 128-byte weight row are stored at
    // slot ^ ((row >> 1) & 7), which spreads the 16 lanes of an LDS read group over all banks (unswizzled: 4-way conflicts)
    const unsigned char* wbase = smem + C16_WOFF + i * 128;
    const int whi = (kg ^ ((i >> 1) & 7)) << 4, wlo = ((4 + kg) ^ ((i >> 1) & 7)) << 4;
    const int pxo = (16 * wave + i) * C16_PXB + kg * 16;                        // B: col = pixel 16 wave + i (+ dx)
    const float inv = 1.0f / (xs * C16_WS);                 // (powers of two: exact)
    // the A fragments (weights) of all 20 K-blocks stay in registers (160 VGPRs; one wave per SIMD anyway): a row then
    // needs only its 36 pixel fragments from LDS, all requested up front
    h16x8 wh[20], wl2[20];
#pragma unroll
    for (int kb = 0; kb < 20; ++kb) {
        wh[kb] = *reinterpret_cast<const h16x8*>(wbase + kb * 2048 + whi);
        wl2[kb] = *reinterpret_cast<const h16x8*>(wbase + kb * 2048 + wlo);
    }
    b1_bits = 0u;

    dbg_stamp(times, blin, 1);
    // rows y+2 and y+3 are both in flight: a row's loads are issued two iterations before its conversion (the matrix
    // part of an iteration is ~0.5 us, a memory round trip under load 2 us), register sets A / B alternate
    if (y0 + 2 < y1 + 1) load_row(y0 + 2, st0, sh0);
    auto do_row = [&](int y, float* stc, float& shc, float* stn, float& shn) {
        // stc/shc: row y+2 (requested one iteration ago) -- converted at the end; stn/shn: row y+3, requested now
        if (y + 2 < y1) load_row(y + 3, stn, shn);
        // independent accumulators per (kernel row, channel half, main / cross term): no MFMA waits on its predecessor
        f32x4 am[3][2], ac[3][2], tm[2], tc[2];
        const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 3; ++u) { am[u][0] = zero4; am[u][1] = zero4; ac[u][0] = zero4; ac[u][1] = zero4; }
        tm[0] = zero4; tm[1] = zero4; tc[0] = zero4; tc[1] = zero4;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                const unsigned char* rowp = smem + ((y + dy) & (PRO_ROWS - 1)) * C16_ROWB + pxo;   // input row y-1+dy
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) {
                    const int kb = (dy * 3 + dx) * 2 + cb;
                    const h16x8 w_hi = wh[kb], w_lo = wl2[kb];
                    const h16x8 p_hi = *reinterpret_cast<const h16x8*>(rowp + dx * C16_PXB + cb * 64);
                    const h16x8 p_lo = *reinterpret_cast<const h16x8*>(rowp + dx * C16_PXB + 128 + cb * 64);
                    ac[dy][cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w_hi, p_lo, ac[dy][cb], 0, 0, 0);
                    am[dy][cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w_hi, p_hi, am[dy][cb], 0, 0, 0);
                    ac[dy][cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w_lo, p_hi, ac[dy][cb], 0, 0, 0);
                    if (dy == 1 && dx == 1) {                    // theta is the 1x1 conv on the centre pixel
                        const h16x8 t_hi = wh[18 + cb], t_lo = wl2[18 + cb];
                        tc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(t_hi, p_lo, tc[cb], 0, 0, 0);
                        tm[cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(t_hi, p_hi, tm[cb], 0, 0, 0);
                        tc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(t_lo, p_hi, tc[cb], 0, 0, 0);
                    }
                }
            }
        }
        f32x4 ag, at;
#pragma unroll
        for (int r = 0; r < 4; ++r) {                            // fixed summation order: cross terms first
            const float cr = ((ac[0][0][r] + ac[0][1][r]) + (ac[1][0][r] + ac[1][1][r])) + (ac[2][0][r] + ac[2][1][r]);
            const float mn = ((am[0][0][r] + am[0][1][r]) + (am[1][0][r] + am[1][1][r])) + (am[2][0][r] + am[2][1][r]);
            ag[r] = mn + cr;
            at[r] = (tm[0][r] + tm[1][r]) + (tc[0][r] + tc[1][r]);
        }
        // D[row = channel 4 kg + r][col = pixel i]
        const int xx = x0 + 16 * wave + i;
        if (xx < W) {
            const size_t o = (((size_t)b * Hp + y + PADPIX) * Wp + xx + PADPIX) * CH + 4 * kg;
            h16x4 vh, vl, vh2, vl2;
            float4 v2, v1q;
            float* v2p = &v2.x;
            float* v1p = &v1q.x;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v1 = ag[r] * inv + bg[r];
                v1p[r] = v1;
                _Float16 h, l;
                c16_split(v1 * B1_FINE_SCALE, h, l);             // P16_A_SCALE (project16.hip)
                vh[r] = h; vl[r] = l;
                c16_split(v1 * B1_COARSE_SCALE, h, l);           // the coarse tier (B1Tiers)
                vh2[r] = h; vl2[r] = l;
                b1_bits = max(b1_bits, __float_as_uint(v1) & 0x7fffffffu);
                v2p[r] = at[r] * inv + bt[r];
            }
            if (b1hi != nullptr) {
                *reinterpret_cast<h16x4*>(b1hi + o) = vh;
                *reinterpret_cast<h16x4*>(b1lo + o) = vl;
            }
            if (b1p != nullptr) *reinterpret_cast<float4*>(b1p + o) = v1q;
            if (tiers && b1hi2 != nullptr) {
                *reinterpret_cast<h16x4*>(b1hi2 + o) = vh2;
                *reinterpret_cast<h16x4*>(b1lo2 + o) = vl2;
            }
            *reinterpret_cast<float4*>(b2p + o) = v2;
        }
        if (y + 1 < y1) store_row(y + 2, stc, shc);              // overwrites row y-2's slot: not read any more
        if (y + 1 >= y1 && !(amax_x * xs < RANGE_LIMIT)) *oor = 1;   // (the strip's last row: amax_x is final, its barrier publishes the flag)
        __syncthreads();
    };
    for (int y = y0; y < y1; y += 2) {
        do_row(y, st0, sh0, st1, sh1);
        if (y + 1 < y1) do_row(y + 1, st1, sh1, st0, sh0);
    }
    };
    run_strip();
    // Did every input value of the strip fit |16 x| < 60000?  (block-uniform: the flag was written before the strip's last barrier)
    if (tiers && *oor != 0) {
        // second attempt (a second copy of the strip's code, not a loop around it: with a back edge the optimizer keeps every address of
        // the strip live across the body -- 512 registers, 60 spilled): the strip again with a scale of its own.  Block maximum of |x|
        // through the LDS (the pixel ring is dead: every wave is past its last row's barrier), 2^e = the largest power of two with
        // 2^e max|x| < 32768; inf in the input: the scale does not matter, the outputs are NaN / inf, which the slots and the range
        // word below report
        float m = amax_x;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        float* red = reinterpret_cast<float*>(smem);
        if (lane == 0) red[wave] = m;
        __syncthreads();
        m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        __syncthreads();                                       // (the ring is about to be restaged)
        int e = 0;
        if (m <= 3.0e38f) { (void)frexpf(m, &e); e = 15 - e; }  // m = f 2^e', f in [0.5, 1)  ->  2^(15 - e') m < 32768
        e = e < -120 ? -120 : (e > 4 ? 4 : e);
        xs = ldexpf(1.0f, e);
        first = false;
        run_strip();
    }
    dbg_stamp(times, blin, 2);
    if (tiers) {
        // the wave's largest |b1| -> its slot (project16_kernel picks the tier from its head's slots); inf / NaN: +inf
        unsigned mb = b1_bits;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mb = max(mb, (unsigned)__shfl_xor((int)mb, o));
        if (lane == 0) {
            const float m = mb >= 0x7f800000u ? __builtin_inff() : __uint_as_float(mb);
            const int img = b - head * hs.imgs;
            hs.tiers.amax[(size_t)head * hs.tiers.slots + (((size_t)img * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 4 + wave] = m;
            // beyond the coarse tier, non-finite values (or weights packed from out-of-range values): the call's range word, as before
            if (range.word != nullptr && (!(m * B1_COARSE_SCALE < RANGE_LIMIT) || !(amax < RANGE_LIMIT))) *range.word = range.tag;
        }
    } else {
        // (a NaN / inf input compares false / true here and is flagged as well: !(amax < limit))
        if (range.word != nullptr && (!(amax < RANGE_LIMIT) || !(amax_x * xs < RANGE_LIMIT) || !((float)(b1_bits >= 0x7f800000u ? __builtin_inff() : __uint_as_float(b1_bits)) * B1_FINE_SCALE < RANGE_LIMIT)))
            *range.word = range.tag;
    }
    dbg_stamp(times, blin, 3);
}

int conv16_blocks_per_head(const Grid& g, int heads, int imgs) {
    const int B = heads * imgs;
    const int strips = (g.W + PRO_TW - 1) / PRO_TW;
    int chunks = (256 + strips * B - 1) / (strips * B);
    if (chunks > (g.H + 1) / 2) chunks = (g.H + 1) / 2;
    if (chunks < 1) chunks = 1;
    const int rows_per_block = (g.H + chunks - 1) / chunks;
    chunks = (g.H + rows_per_block - 1) / rows_per_block;
    return strips * chunks * imgs * 4;                      // (a slot per wave of every block)
}

// thr / bias heads (two 7x7 stride-4 convolutions 64 -> 1 over the SAME-padded input, dagl.py:212-215).
// One block per (image, query row, 32 queries, 16-channel group): the 7 input rows the queries need are staged in LDS
// with coalesced row loads (a per-query gather touches seven 28-byte pieces per channel: 22 % of every cache line,
// 27 us at 256^2); all loads are unconditional (clamped addresses, zeroed afterwards) and in flight together.
// lane = (query, half of the 7 rows) reads its 7 taps per row as two conflict-free ds_read_b128, the weights as broadcast
// float4; wave w owns channels 4j + w of the group.  The four channel groups write partial sums which
// thr_bias_reduce_kernel adds in a fixed order.
constexpr int TB_XW = 4 * TB_Q + 4;         // input columns a block needs (4 (Q-1) + 7, rounded up to float4)
constexpr int TB_N = TB_CG * KS * TB_XW;    // staged floats: 14784
constexpr int TB_PER = (TB_N + 255) / 256;  // 58 per thread
__global__ __launch_bounds__(256) void thr_bias_kernel(Grid gr, ThrHeadSet hs,
                                                       float* __restrict__ part_out /* per head [4][imgs][L][2] */) {
    __shared__ __attribute__((aligned(16))) float tile[TB_N + 256];                  // 58.8 KiB
    __shared__ __attribute__((aligned(16))) float wl[2][TB_CG][KS][8];               // 7 KiB: both heads' weights, rows of 7 (+1)
    __shared__ float part[4][TB_Q][2];
    const int tid = threadIdx.x;
    const int lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qi = lane & 31, hh = lane >> 5;
    const int chunks = (gr.Lw + TB_Q - 1) / TB_Q;
    const int qr = blockIdx.x / chunks, q0 = (blockIdx.x - qr * chunks) * TB_Q;
    const int b = blockIdx.y, grp = blockIdx.z;
    const int head = b / hs.imgs, img = b - head * hs.imgs;
    const float* __restrict__ thr_w = hs.thr_w[head];
    const float* __restrict__ bias_w = hs.bias_w[head];
    const int c0 = grp * TB_CG;
    const int y0 = QS * qr - gr.pt, x0 = QS * q0 - gr.pl;
    const float* xc0 = hs.x[head] + ((size_t)img * PC + c0) * gr.N;
    float v[TB_PER];
#pragma unroll
    for (int j = 0; j < TB_PER; ++j) {
        const int idx = tid + 256 * j;
        const int c = idx / (KS * TB_XW), rem = idx - c * (KS * TB_XW);
        const int r = rem / TB_XW, col = rem - r * TB_XW;
        const int yy = y0 + r, xx = x0 + col;
        const int yc = yy < 0 ? 0 : (yy >= gr.H ? gr.H - 1 : yy), xc = xx < 0 ? 0 : (xx >= gr.W ? gr.W - 1 : xx);
        const int cc = c < TB_CG ? c : TB_CG - 1;
        v[j] = xc0[(size_t)cc * gr.N + yc * gr.W + xc];
    }
    for (int e = tid; e < TB_CG * KS * KS; e += 256) {
        const int ch = e / (KS * KS), t = e - ch * (KS * KS);
        const int kh = t / KS, kw = t - kh * KS;
        wl[0][ch][kh][kw] = thr_w[c0 * (KS * KS) + e];
        wl[1][ch][kh][kw] = bias_w[c0 * (KS * KS) + e];
    }
#pragma unroll
    for (int j = 0; j < TB_PER; ++j) {
        const int idx = tid + 256 * j;
        const int c = idx / (KS * TB_XW), rem = idx - c * (KS * TB_XW);
        const int r = rem / TB_XW, col = rem - r * TB_XW;
        const int yy = y0 + r, xx = x0 + col;
        const bool ok = yy >= 0 && yy < gr.H && xx >= 0 && xx < gr.W;                 // SAME zero padding
        tile[idx] = ok ? v[j] : 0.f;
    }
    __syncthreads();
    const int kh0 = hh ? 4 : 0, kh1 = hh ? KS : 4;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < TB_CG / 4; ++j) {
        const int cl = 4 * j + w;
        for (int kh = kh0; kh < kh1; ++kh) {
            const float* rp = tile + (cl * KS + kh) * TB_XW + 4 * qi;
            const float4 v0 = *reinterpret_cast<const float4*>(rp);
            const float4 v1 = *reinterpret_cast<const float4*>(rp + 4);
            const float4 a0 = *reinterpret_cast<const float4*>(&wl[0][cl][kh][0]);
            const float4 a1 = *reinterpret_cast<const float4*>(&wl[0][cl][kh][4]);
            const float4 b0 = *reinterpret_cast<const float4*>(&wl[1][cl][kh][0]);
            const float4 b1 = *reinterpret_cast<const float4*>(&wl[1][cl][kh][4]);
            s1 = fmaf(v0.x, a0.x, s1); s1 = fmaf(v0.y, a0.y, s1); s1 = fmaf(v0.z, a0.z, s1); s1 = fmaf(v0.w, a0.w, s1);
            s1 = fmaf(v1.x, a1.x, s1); s1 = fmaf(v1.y, a1.y, s1); s1 = fmaf(v1.z, a1.z, s1);
            s2 = fmaf(v0.x, b0.x, s2); s2 = fmaf(v0.y, b0.y, s2); s2 = fmaf(v0.z, b0.z, s2); s2 = fmaf(v0.w, b0.w, s2);
            s2 = fmaf(v1.x, b1.x, s2); s2 = fmaf(v1.y, b1.y, s2); s2 = fmaf(v1.z, b1.z, s2);
        }
    }
    s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
    if (hh == 0) { part[w][qi][0] = s1; part[w][qi][1] = s2; }
    __syncthreads();
    if (tid < TB_Q && q0 + tid < gr.Lw) {
        const size_t o = (size_t)head * 8 * hs.imgs * gr.L + (((size_t)grp * hs.imgs + img) * gr.L + (size_t)qr * gr.Lw + q0 + tid) * 2;
        part_out[o] = (part[0][tid][0] + part[1][tid][0]) + (part[2][tid][0] + part[3][tid][0]);
        part_out[o + 1] = (part[0][tid][1] + part[1][tid][1]) + (part[2][tid][1] + part[3][tid][1]);
    }
}

// (W % 4 == 0: the float4 form; body in thr_bias4.h, shared with project16_kernel)
__global__ __launch_bounds__(256) void thr_bias4_kernel(Grid gr, ThrHeadSet hs,
                                                        float* __restrict__ part_out /* per head [4][imgs][L][2] */) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[TB4_LDS_BYTES];
    thr_bias4_block(gr, hs, part_out, blockIdx.x, blockIdx.y, blockIdx.z, smem);
}

__global__ void thr_bias_reduce_kernel(size_t n /* B*L */, const float* __restrict__ part, const float* __restrict__ thr_b,
                                       const float* __restrict__ bias_b, float* __restrict__ thr, float* __restrict__ bias) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float2 p0 = reinterpret_cast<const float2*>(part)[i], p1 = reinterpret_cast<const float2*>(part)[n + i];
    const float2 p2 = reinterpret_cast<const float2*>(part)[2 * n + i], p3 = reinterpret_cast<const float2*>(part)[3 * n + i];
    thr[i] = ((p0.x + p1.x) + (p2.x + p3.x)) + thr_b[0];
    bias[i] = ((p0.y + p1.y) + (p2.y + p3.y)) + bias_b[0];
}

int launch_conv_pair16_heads(hipStream_t s, int heads, int imgs, const Grid& g, const ConvHeadSet& hs, float* b2p,
                             uint16_t* b1_hi, uint16_t* b1_lo, uint32_t* clear_a, int clear_a_words, uint32_t* clear_b,
                             int clear_b_words, RangeTag range) {
    const int B = heads * imgs;
    const int strips = (g.W + PRO_TW - 1) / PRO_TW;
    constexpr int target = 256;                       // one block per CU over the whole launch, at least 2 rows per block
    int chunks = (target + strips * B - 1) / (strips * B);
    if (chunks > (g.H + 1) / 2) chunks = (g.H + 1) / 2;
    if (chunks < 1) chunks = 1;
    const int rows_per_block = (g.H + chunks - 1) / chunks;
    chunks = (g.H + rows_per_block - 1) / rows_per_block;
    unsigned long long* times = nullptr;
#ifdef DAGL_ABLATION
    if (getenv("DAGL_TIMES_FILE")) times = dbg_times_buffer((size_t)strips * chunks * B);
#endif
    hipLaunchKernelGGL(conv_pair16_kernel, dim3(strips, chunks, B), dim3(256), 0, s, g.H, g.W, rows_per_block, hs,
                       b2p, (float*)nullptr, b1_hi, b1_lo, clear_a, clear_a_words, clear_b, clear_b_words, range, times);
    DAGL_LAUNCH_CHECK("conv_pair16_kernel");
#ifdef DAGL_ABLATION
    if (times) dbg_times_dump(s, "conv_pair16_kernel", times, (size_t)strips * chunks * B);
#endif
    return DAGL_OK;
}

int launch_thr_bias_heads(hipStream_t s, int heads, int imgs, const Grid& g, const ThrHeadSet& hs, float* thr_part) {
    const dim3 tb_grid(g.Lh * ((g.Lw + TB_Q - 1) / TB_Q), heads * imgs, TB_GROUPS);
    bool aligned = true;
    for (int h = 0; h < heads; ++h) aligned = aligned && (reinterpret_cast<uintptr_t>(hs.x[h]) & 15u) == 0;
    if (g.W % 4 == 0 && g.W >= 4 && g.pl == 1 && aligned)
        hipLaunchKernelGGL(thr_bias4_kernel, tb_grid, dim3(256), 0, s, g, hs, thr_part);
    else
        hipLaunchKernelGGL(thr_bias_kernel, tb_grid, dim3(256), 0, s, g, hs, thr_part);
    DAGL_LAUNCH_CHECK("thr_bias_kernel");
    return DAGL_OK;
}

int launch_prologue(hipStream_t s, int B, const Grid& g, const float* x, const float* g_w, const float* g_b,
                    const float* th_w, const float* th_b, const float* thr_w, const float* thr_b,
                    const float* bias_w, const float* bias_b, float* b1p, float* b2p, float* thr, float* bias,
                    uint16_t* b1_hi, uint16_t* b1_lo, float* thr_part, bool borders_zero, bool defer_thr_reduce, uint32_t* clear_a,
                    int clear_a_words, uint32_t* clear_b, int clear_b_words, RangeTag range, const unsigned char* conv_w16,
                    bool skip_conv, const B1Tiers* tiers) {
    int rcz;
    if (!borders_zero) {
        if ((rcz = launch_zero_borders(s, B, g.H, g.W, b1p ? b1p : b2p, b2p))) return rcz;
        if (b1_hi != nullptr && (rcz = launch_zero_borders16(s, B, g.H, g.W, b1_hi, b1_lo))) return rcz;
        if (b1_hi != nullptr && tiers != nullptr && tiers->hi2 != nullptr && (rcz = launch_zero_borders16(s, B, g.H, g.W, tiers->hi2, tiers->lo2))) return rcz;
    }
    const int strips = (g.W + PRO_TW - 1) / PRO_TW;
    // one block per CU over the whole launch (1 block/CU resident: 332 registers), at least 2 rows per block
    constexpr int target = 256;
    int chunks = (target + strips * B - 1) / (strips * B);
    if (chunks > (g.H + 1) / 2) chunks = (g.H + 1) / 2;
    if (chunks < 1) chunks = 1;
    const int rows_per_block = (g.H + chunks - 1) / chunks;
    chunks = (g.H + rows_per_block - 1) / rows_per_block;
    if (skip_conv) {
        // (the caller ran launch_conv_pair16_heads for all its heads)
    } else if ((b1p == nullptr && b1_hi != nullptr) || (b1p != nullptr && conv_w16 != nullptr)) {
        // (split-fp16 convolutions; with b1p: the differentiable path's forward -- an fp32 map out, the fp16 pairs optional)
        if (conv_w16 == nullptr) { set_error("launch_prologue: the split-fp16 convolutions need their packed weights"); return DAGL_ERR_INVALID; }
        ConvHeadSet hs = {};
        hs.x[0] = x; hs.w[0] = conv_w16; hs.gb[0] = g_b; hs.tb[0] = th_b; hs.imgs = B;
        if (tiers != nullptr) hs.tiers = *tiers;
        unsigned long long* times = nullptr;
#ifdef DAGL_ABLATION
        if (getenv("DAGL_TIMES_FILE")) times = dbg_times_buffer((size_t)strips * chunks * B);
#endif
        hipLaunchKernelGGL(conv_pair16_kernel, dim3(strips, chunks, B), dim3(256), 0, s, g.H, g.W, rows_per_block, hs,
                           b2p, b1p, b1_hi, b1_lo, clear_a, clear_a_words, clear_b, clear_b_words, range, times);
        DAGL_LAUNCH_CHECK("conv_pair16_kernel");
#ifdef DAGL_ABLATION
        if (times) dbg_times_dump(s, "conv_pair16_kernel", times, (size_t)strips * chunks * B);
#endif
    } else {
        hipLaunchKernelGGL(conv_pair_kernel, dim3(strips, chunks, B), dim3(256), 0, s, g.H, g.W, rows_per_block, x, g_w,
                           g_b, th_w, th_b, b1p, b2p, b1_hi, b1_lo);
        DAGL_LAUNCH_CHECK("conv_pair_kernel");
    }
    if (thr != nullptr) {
        // thr_part: [4][B][L][2] floats of scratch for the channel groups' partial sums
        if (!skip_conv) {                                          // (else: one launch_thr_bias_heads for all the caller's heads)
            ThrHeadSet hs = {};
            hs.x[0] = x; hs.thr_w[0] = thr_w; hs.bias_w[0] = bias_w; hs.imgs = B;
            const int rct = launch_thr_bias_heads(s, 1, B, g, hs, thr_part);
            if (rct) return rct;
        }
        const size_t n = (size_t)B * g.L;
        if (!defer_thr_reduce)
        hipLaunchKernelGGL(thr_bias_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, n, thr_part, thr_b,
                           bias_b, thr, bias);
        DAGL_LAUNCH_CHECK("thr_bias_reduce_kernel");
    }
    return DAGL_OK;
}

}  // namespace dagl
