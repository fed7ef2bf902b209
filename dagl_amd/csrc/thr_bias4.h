// thr / bias heads of dagl.py:211-215 (7x7 stride-4 SAME convolutions, 64 -> 1) for W % 4 == 0: the body of thr_bias4_kernel
// (prologue.hip) as a device function over caller-provided LDS, so that project16_kernel can run these blocks in the shadow of its own
// launch (round 5: the heads' partial sums are first needed by query_thresholds_kernel, BEHIND the projection -- as a launch of their
// own between the convolutions and the projection they were 13 us of every adaptive-mode call).
#pragma once
#include "dagl_common.h"

namespace dagl {

constexpr int TB_PC = 64;                      // input channels
constexpr int TB_Q = 32;                       // queries per block
constexpr int TB_CG = 16;                      // channels per block
constexpr int TB_GROUPS = TB_PC / TB_CG;       // 4
// The same for W % 4 == 0 (then pl = 1 and every input row is 16-byte aligned): the tile starts four columns left of the
// first query's window and is staged with float4 loads -- 15 per thread instead of 58 scalar ones, a quarter of the address
// arithmetic; a float4 is entirely inside or entirely outside the image.  A query's 7 taps of a row are elements 3..9 of the
// three float4 at its position.
constexpr int TB4_XW = 4 * TB_Q + 8;                       // 136 staged columns = 34 float4
constexpr int TB4_F4 = TB_CG * KS * (TB4_XW / 4);          // 3808 float4 per block
constexpr int TB4_PER = (TB4_F4 + 255) / 256;              // 15 per thread
constexpr int TB4_LDS_BYTES = (TB4_F4 + 32) * 16 + 2 * TB_CG * KS * 8 * 4 + 4 * TB_Q * 2 * 4;      // 69 888
__host__ __device__ inline int thr_bias4_grid_x(const Grid& g) { return g.Lh * ((g.Lw + TB_Q - 1) / TB_Q); }
inline bool thr_bias4_ok(const Grid& g, const ThrHeadSet& hs, int heads) {
    bool aligned = true;
    for (int h = 0; h < heads; ++h) aligned = aligned && (reinterpret_cast<uintptr_t>(hs.x[h]) & 15u) == 0;
    return g.W % 4 == 0 && g.W >= 4 && g.pl == 1 && aligned;
}

// block (bx, by, bz) of the grid (thr_bias4_grid_x, heads x imgs, TB_GROUPS); 256 threads; smem: TB4_LDS_BYTES, 16-byte aligned
__device__ __forceinline__ void thr_bias4_block(const Grid& gr, const ThrHeadSet& hs, float* __restrict__ part_out, int bx, int by, int bz,
                                                unsigned char* smem) {
    constexpr int PC = TB_PC;
    float4* tile4 = reinterpret_cast<float4*>(smem);
    float (*wl)[TB_CG][KS][8] = reinterpret_cast<float (*)[TB_CG][KS][8]>(smem + (TB4_F4 + 32) * 16);
    float (*part)[TB_Q][2] = reinterpret_cast<float (*)[TB_Q][2]>(smem + (TB4_F4 + 32) * 16 + 2 * TB_CG * KS * 8 * 4);

    const int tid = threadIdx.x;
    const int lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qi = lane & 31, hh = lane >> 5;
    const int chunks = (gr.Lw + TB_Q - 1) / TB_Q;
    const int qr = bx / chunks, q0 = (bx - qr * chunks) * TB_Q;
    const int b = by, grp = bz;
    const int head = b / hs.imgs, img = b - head * hs.imgs;
    const float* __restrict__ thr_w = hs.thr_w[head];
    const float* __restrict__ bias_w = hs.bias_w[head];
    const int c0 = grp * TB_CG;
    const int y0 = QS * qr - gr.pt, xa = QS * q0 - 4;                                 // (pl = 1: window of query q0 starts at xa + 3)
    const float* xc0 = hs.x[head] + ((size_t)img * PC + c0) * gr.N;
    constexpr int RW4 = TB4_XW / 4;                                                   // 34
    float4 v[TB4_PER];
#pragma unroll
    for (int j = 0; j < TB4_PER; ++j) {
        const int idx = tid + 256 * j;
        const int c = idx / (KS * RW4), rem = idx - c * (KS * RW4);
        const int r = rem / RW4, col = 4 * (rem - r * RW4);
        const int yy = y0 + r, xx = xa + col;
        const int yc = yy < 0 ? 0 : (yy >= gr.H ? gr.H - 1 : yy), xc = xx < 0 ? 0 : (xx >= gr.W ? gr.W - 4 : xx);
        const int cc = c < TB_CG ? c : TB_CG - 1;
        v[j] = *reinterpret_cast<const float4*>(xc0 + (size_t)cc * gr.N + yc * gr.W + xc);     // clamped, zeroed below
    }
    for (int e = tid; e < TB_CG * KS * KS; e += 256) {
        const int ch = e / (KS * KS), t = e - ch * (KS * KS);
        const int kh = t / KS, kw = t - kh * KS;
        wl[0][ch][kh][kw] = thr_w[c0 * (KS * KS) + e];
        wl[1][ch][kh][kw] = bias_w[c0 * (KS * KS) + e];
    }
#pragma unroll
    for (int j = 0; j < TB4_PER; ++j) {
        const int idx = tid + 256 * j;
        const int c = idx / (KS * RW4), rem = idx - c * (KS * RW4);
        const int r = rem / RW4, col = 4 * (rem - r * RW4);
        const int yy = y0 + r, xx = xa + col;
        const bool ok = yy >= 0 && yy < gr.H && xx >= 0 && xx < gr.W;                 // SAME zero padding
        if (idx < TB4_F4 + 32) tile4[idx] = ok ? v[j] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    const int kh0 = hh ? 4 : 0, kh1 = hh ? KS : 4;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < TB_CG / 4; ++j) {
        const int cl = 4 * j + w;
        for (int kh = kh0; kh < kh1; ++kh) {
            const float4* rp = tile4 + (cl * KS + kh) * RW4 + qi;
            const float4 u0 = rp[0], u1 = rp[1], u2 = rp[2];
            const float4 a0 = *reinterpret_cast<const float4*>(&wl[0][cl][kh][0]);
            const float4 a1 = *reinterpret_cast<const float4*>(&wl[0][cl][kh][4]);
            const float4 b0 = *reinterpret_cast<const float4*>(&wl[1][cl][kh][0]);
            const float4 b1 = *reinterpret_cast<const float4*>(&wl[1][cl][kh][4]);
            // the same fma order as thr_bias_kernel: taps 0..6 = u0.w u1.x u1.y u1.z u1.w u2.x u2.y
            s1 = fmaf(u0.w, a0.x, s1); s1 = fmaf(u1.x, a0.y, s1); s1 = fmaf(u1.y, a0.z, s1); s1 = fmaf(u1.z, a0.w, s1);
            s1 = fmaf(u1.w, a1.x, s1); s1 = fmaf(u2.x, a1.y, s1); s1 = fmaf(u2.y, a1.z, s1);
            s2 = fmaf(u0.w, b0.x, s2); s2 = fmaf(u1.x, b0.y, s2); s2 = fmaf(u1.y, b0.z, s2); s2 = fmaf(u1.z, b0.w, s2);
            s2 = fmaf(u1.w, b1.x, s2); s2 = fmaf(u2.x, b1.y, s2); s2 = fmaf(u2.y, b1.z, s2);
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

}  // namespace dagl
