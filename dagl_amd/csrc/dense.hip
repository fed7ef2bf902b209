// Dense neighbourhoods of the adaptive mask (DN_Gray/model/dagl.py:250-264 when most keys pass, e.g. with
// default-initialised thr/bias heads ~95 % of them): per-query neighbour lists stop making sense, so this path is the
// reference's dense formulation, streamed -- S = Wq X^T, mask, softmax over ALL keys, A V -- in one pass over the keys,
// nothing of size L x N ever stored.
//
//   block = 64 queries = 2 query tiles x 4 waves; a wave = 32 queries x a share of the 784 output columns (6 / 6 / 7 / 6 column
//   tiles of 32 = two patch taps x 16 channels; the first two waves of a query tile also form the scores).  Eight waves: two
//   per SIMD, so that one wave's operand assembly and softmax arithmetic run under the other's matrix instructions (with four
//   waves -- 13 / 12 tiles each, 208 accumulator registers -- every SIMD had ONE wave and nothing overlapped: 3.96 ms at 256^2)
//   per 32-key tile (keys = an 8 x 4 pixel block of the map):
//     S      v_mfma_f32_32x32x16_f16 on split features (64 x = hi + lo, made once per call by feat_split_kernel; three
//            products per 16 features as in the projection), keys x queries, the 13 K-blocks split between the first two waves
//            of a query tile and handed to all four through LDS.  The key rows enter the MFMA in a permuted order so that a lane ends
//            up with scores of 16 keys of ONE query that are two runs of 8 CONSECUTIVE keys -- exactly the K-layout of
//            the next MFMA's operand;
//     l, p   the reference's fp32 expression order for m and l = (S m) 10; masked keys keep l = 0 and count in the
//            denominator (no renormalisation), keys outside the image row count nowhere; p = e^(l - M') with M' an UPPER
//            bound of the row maximum known before the pass (from the bf16 screen's row maxima, dense_rowmax_kernel): the
//            softmax is shift invariant, M' is within ~1 % of the true maximum, so nothing is ever rescaled and the
//            accumulators live in the matrix cores' registers untouched;
//     A V    v_mfma_f32_32x32x16_f16 with split operands (2^14 p = hi + lo, 16 v = hi + lo, three products, one fp32
//            accumulator: the projection's recipe): out^T[col][q] += V[key][col] p[q][key].  B = the lane's own weights,
//            straight from its registers.  A needs 8 consecutive KEYS of one column, i.e. 8 consecutive pixels of one
//            channel: the value-map region (10 rows x 14 pixels x 16 channels) is staged PLANAR in LDS as fp16 hi / lo, and a
//            patch tap's kw shift becomes a 2-byte-granular offset: five dwords are read and funnel-shifted (v_alignbyte).
//   key range split over `splits` blocks per 64 queries; dense_combine_kernel merges the partial sums and rows.
//
// Matrix time per 32 x 32 (key, query) tile: 39 + 150 fp16 MFMAs (32 clk) instead of 500 fp32 ones (64 clk).
#include <stdlib.h>

#include "dagl_common.h"

namespace dagl {

typedef _Float16 dnh8 __attribute__((ext_vector_type(8)));
typedef unsigned dnu4 __attribute__((ext_vector_type(4)));

constexpr int DN_XPART = 14 * 512;                    // halfs of one part (hi or lo) of a staged key-feature tile: 32 rows x 216, 14 KiB pieces
constexpr int DN_XT = 2 * DN_XPART;                   // halfs per tile buffer: hi | lo
constexpr int DN_KB = 13;                             // K blocks of 16 features (208 >= 196)
constexpr float DN_FS = 64.0f;                        // pre-scaling of the split features: 64 x = hi + lo
constexpr int DN_TW = 8, DN_TH = 4;                   // a key tile = 8 x 4 pixels (32 keys): narrow maps waste little of it
constexpr int DN_RH = DN_TH + KS - 1, DN_RW = DN_TW + KS - 1;   // value-map region of a tile: 10 rows x 14 pixels
constexpr int DN_XW = 16;                             // staged pixels per plane row (14 used; dword reads run to 15)
constexpr int DN_CSTR = DN_RH * DN_XW + 2;            // halfs per channel plane: 10 rows x 16 + 2 (81 dwords: odd, the 16 channels
                                                      // of a column tile hit different banks)
constexpr int DN_PLANE_H = CH * DN_CSTR;              // halfs per part (hi or lo): [channel][kernel row][pixel]
constexpr int DN_CT = 25;                             // column tiles of 32 (two taps x 16 channels; the 50th tap is a dummy)
// A V: every wave takes a share of the 25 column tiles for BOTH query tiles of the block (3 tiles, the last wave 4): a value
// fragment assembled from the LDS planes (five dword reads + four funnel shifts per 16 bytes -- what this kernel's LDS time
// consists of) then feeds six multiplies instead of three (round 2: a wave = one query tile x 6-7 column tiles, 2.08 ms)
constexpr int DN_CTMAX = 4;
__host__ __device__ constexpr int dn_ct_start(int w8) { return 3 * w8; }
__host__ __device__ constexpr int dn_ct_count(int w8) { return w8 == 7 ? 4 : 3; }
constexpr float DN_PS = 16384.0f, DN_VS = 16.0f;      // power-of-two pre-scaling of the split operands

__device__ __forceinline__ float dn_logit(float s, float mtq, float bsq, bool& pass) {
    const float m = (s - mtq) + bsq;                  // same expression order as dagl.py:256
    pass = m > 0.f;
    return pass ? __fmul_rn(__fmul_rn(s, m), SOFTMAX_SCALE) : 0.f;
}

// column tile ct = taps (2 ct, 2 ct + 1) x 16 channels; lanes i >= 16 ("second") take the odd tap (tap 49 does not exist:
// those lanes of tile 24 recompute tap 48 and their columns are never stored).
// A operand of kblock kb: halfs e = 0..7 = V[key 16 kb + 8 h + e][tap][c]; key = pixel (row 2 kb + h, column e) of the tile,
// so the value is plane[c][2 kb + h + kh][kw + e]
// One code path for all four column shares (ct0, cnt are wave-uniform): a four-way dispatch on the share made the register
// allocator keep all four instantiations' fragments alive (128 spills at 256 registers).
__device__ __forceinline__ void dn_pv(f32x16 (&acc)[2][DN_CTMAX], const unsigned char* planes, int c, int h, bool second,
                                      const dnh8 (&p_hi)[2][2], const dnh8 (&p_lo)[2][2], int ct0, int cnt, int variant = 0) {
#pragma unroll
    for (int t = 0; t < DN_CTMAX; ++t) {
        if (t >= cnt) continue;                                              // wave-uniform
        const int ct = ct0 + t;
        const int tapa = 2 * ct, tapb = (2 * ct + 1 < KS * KS) ? 2 * ct + 1 : 2 * ct;
        const int kh = second ? tapb / KS : tapa / KS, kw = second ? tapb % KS : tapa % KS;
        // byte offset of (c, row h + kh, pixel kw) inside a part; dword-aligned base + byte shift 0 / 2
        const int boff = (c * DN_CSTR + (h + kh) * DN_XW + kw) * 2;
        const unsigned char* base = planes + (boff & ~3);
        const unsigned shift = (unsigned)(boff & 3);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            dnh8 v_hi, v_lo;
            if (variant & 8) { v_hi = p_hi[0][kb]; v_lo = p_lo[0][kb]; }       // (ablation: no value-fragment assembly)
            else
#pragma unroll
            for (int part = 0; part < 2; ++part) {
                const unsigned* dp = reinterpret_cast<const unsigned*>(base + part * (DN_PLANE_H * 2) + kb * (2 * DN_XW * 2));
                const unsigned d0 = dp[0], d1 = dp[1], d2 = dp[2], d3 = dp[3], d4 = dp[4];
                dnu4 w;
                w[0] = __builtin_amdgcn_alignbyte(d1, d0, shift);
                w[1] = __builtin_amdgcn_alignbyte(d2, d1, shift);
                w[2] = __builtin_amdgcn_alignbyte(d3, d2, shift);
                w[3] = __builtin_amdgcn_alignbyte(d4, d3, shift);
                if (part == 0) v_hi = __builtin_bit_cast(dnh8, w); else v_lo = __builtin_bit_cast(dnh8, w);
            }
#pragma unroll
            for (int qq = 0; qq < 2; ++qq) acc[qq][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(v_hi, p_lo[qq][kb], acc[qq][t], 0, 0, 0);
#pragma unroll
            for (int qq = 0; qq < 2; ++qq) acc[qq][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(v_lo, p_hi[qq][kb], acc[qq][t], 0, 0, 0);
#pragma unroll
            for (int qq = 0; qq < 2; ++qq) acc[qq][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(v_hi, p_hi[qq][kb], acc[qq][t], 0, 0, 0);
        }
    }
}

__device__ __forceinline__ void dn_store(const f32x16 (&acc)[DN_CTMAX], float* po, int h, int ct0, int cnt) {
    constexpr float inv = 1.0f / (DN_PS * DN_VS);
#pragma unroll
    for (int t = 0; t < DN_CTMAX; ++t) {
        if (t >= cnt) continue;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const int col = 32 * (ct0 + t) + 8 * gq + 4 * h;                 // acc[t][4 gq + u] = out[q][col + u]
            if (col < P)
                *reinterpret_cast<float4*>(po + col) = make_float4(acc[t][4 * gq] * inv, acc[t][4 * gq + 1] * inv,
                                                                   acc[t][4 * gq + 2] * inv, acc[t][4 * gq + 3] * inv);
        }
    }
}

constexpr int DN_THREADS = 512;
__global__ __launch_bounds__(DN_THREADS) void dense_attend_kernel(DenseArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned short sm[2][DN_XT];           // 56 KiB: key-feature tiles hi | lo (LDS-DMA)
    __shared__ __attribute__((aligned(16))) unsigned short spl[2 * DN_PLANE_H + 64]; // 21 KiB: value planes hi | lo of ONE tile
    __shared__ __attribute__((aligned(16))) unsigned short sq[2][64 * DSH];        // 54 KiB: the block's 64 query rows hi | lo
    __shared__ float sx[2 * 2 * 16 * 64];                                          // 16 KiB: partial scores exchanged per tile
    __shared__ __attribute__((aligned(16))) uint4 spq[2][64][5];                   // 10 KiB: a tile's weights as halfs (80 B per lane: odd slot count)
    __shared__ double szz[2][4][32][2];                                            // 4 KiB: the waves' shares of a query's sums (end of the block)
    __shared__ int sdg[2][4][32];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int i = lane & 31, h = lane >> 5;
    const int b = blockIdx.y;
    const Grid& g = a.g;
    const int n_qblocks = (g.L + 63) / 64;
    const int qb = blockIdx.x % n_qblocks, split = blockIdx.x / n_qblocks;
    const int qt = wave >> 2, part = wave & 3;
    const int ct0 = dn_ct_start(wave), ctn = dn_ct_count(wave);                  // this wave's column tiles (wave-uniform), both query tiles
    const int tile0 = split * a.tiles_per_split;
    int tile1 = tile0 + a.tiles_per_split;
    if (tile1 > a.n_tiles) tile1 = a.n_tiles;

    const int q = qb * 64 + qt * 32 + i;
    const bool qvalid = q < g.L;
    const int qc = qvalid ? q : g.L - 1;
    const size_t qlin = (size_t)b * g.L + qc;

    // the 64 query rows (split fp16, 432-byte rows: whole 16-byte pieces; rows past L are zero guard rows) -> LDS
    for (int pt = 0; pt < 2; ++pt) {
        const uint4* src = reinterpret_cast<const uint4*>((pt ? a.wq_lo : a.wq_hi) + ((size_t)b * a.rows_qh + (size_t)qb * 64) * DSH);
        for (int e = tid; e < 64 * DSH / 8; e += DN_THREADS) reinterpret_cast<uint4*>(&sq[pt][0])[e] = src[e];
    }
    const unsigned short* qrow = &sq[0][(qt * 32 + i) * DSH + 8 * h];       // B operand of the score MFMAs: Wq[q][16 kb + 8 h ..]
    const float mtq = a.mt[qlin], bsq = a.bs[qlin];

    f32x16 acc[2][DN_CTMAX];
#pragma unroll
    for (int qq = 0; qq < 2; ++qq)
#pragma unroll
        for (int t = 0; t < DN_CTMAX; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[qq][t][r] = 0.f;
    // upper bound of the row's largest logit: S <= S~max / (1 - DELTA) for the bf16 screen's row maximum S~max, and l grows with S
    float m_run;
    {
        const float sub = a.smax[qlin] * (1.0f / (1.0f - SCREEN_DELTA)) * (1.0f + 1e-6f);
        bool ps;
        const float lub = dn_logit(sub, mtq, bsq, ps);
        m_run = fmaxf(lub, 0.f);                       // masked keys have l = 0
    }
    double z_run = 0.0, zp_run = 0.0;                  // sum over all keys / over passing keys of e^(l - m_run)
    int deg = 0;

    const float* vb = a.b2p + (size_t)b * g.Hp * g.Wp * CH;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr_of(&sm[0][0]));
    auto stage_x = [&](int tile, int buf) {            // key features hi | lo: per part 4 pixel rows x 8 keys x 432 B by LDS-DMA
        const int ty = tile / a.tiles_per_row, jy0 = ty * DN_TH, jx0 = (tile - ty * a.tiles_per_row) * DN_TW;
        for (int p = wave; p < 32; p += DN_THREADS / 64) {                   // (part, row dy, piece): 3456 B = 3 x 1 KiB + 384 B
            const int xpart = p >> 4, dy = (p >> 2) & 3, pc = p & 3;
            int jy = jy0 + dy; if (jy > g.H - 1) jy = g.H - 1;               // ragged bottom: a valid row, keys masked below
            const unsigned short* xs = (xpart ? a.x_lo : a.x_hi) + ((size_t)b * a.rows_xh + (size_t)jy * g.W + jx0) * DSH;
            if (pc < 3 || lane < 24)
                glds16_asm(reinterpret_cast<const float*>(xs + pc * 512 + lane * 8),
                           __builtin_amdgcn_readfirstlane(lds0 + (unsigned)((buf * DN_XT + xpart * DN_XPART + dy * 8 * DSH) * 2 + pc * 1024)));
        }
    };
    // value-map region of a tile: 10 rows x 14 pixels x 16 channels fp32 NHWC -> registers -> planar fp16 hi | lo in LDS.
    // Work item = (region row, pixel PAIR, channel quad): two float4 loads, then per channel the two pixels' halfs go out
    // as one 4-byte store.  The index arithmetic does not depend on the tile and is done once.
    constexpr int DN_ITEMS = DN_RH * (DN_RW / 2) * 4;                        // 280
    constexpr int DN_NIT = (DN_ITEMS + DN_THREADS - 1) / DN_THREADS;         // 1 per thread
    float4 rv[DN_NIT][2];
    int it_row[DN_NIT], it_px[DN_NIT], it_c4[DN_NIT], it_lds[DN_NIT];
#pragma unroll
    for (int j = 0; j < DN_NIT; ++j) {
        int idx = tid + DN_THREADS * j;
        const bool on = idx < DN_ITEMS;
        if (!on) idx = DN_ITEMS - 1;
        const int row = idx / (DN_RW / 2 * 4), rem = idx - row * (DN_RW / 2 * 4);
        const int pp = rem >> 2, c4 = rem & 3;
        it_row[j] = row; it_px[j] = 2 * pp; it_c4[j] = 4 * c4;
        it_lds[j] = on ? (4 * c4) * DN_CSTR + row * DN_XW + 2 * pp : -1;
    }
    auto load_region = [&](int tile) {
        const int ty = tile / a.tiles_per_row, jy0 = ty * DN_TH, jx0 = (tile - ty * a.tiles_per_row) * DN_TW;
        const int limx = g.Wp - 1 - jx0, limy = g.Hp - 1 - jy0;              // stay inside the padded map
#pragma unroll
        for (int j = 0; j < DN_NIT; ++j) {
            const int r = it_row[j] > limy ? limy : it_row[j];
            const int p0 = it_px[j] > limx ? limx : it_px[j], p1 = it_px[j] + 1 > limx ? limx : it_px[j] + 1;
            const float* rb = vb + ((size_t)(jy0 + r) * g.Wp + jx0) * CH + it_c4[j];
            rv[j][0] = *reinterpret_cast<const float4*>(rb + p0 * CH);
            rv[j][1] = *reinterpret_cast<const float4*>(rb + p1 * CH);
        }
    };
    auto store_region = [&]() {
#pragma unroll
        for (int j = 0; j < DN_NIT; ++j) {
            if (it_lds[j] < 0) continue;
            const float v0[4] = {rv[j][0].x, rv[j][0].y, rv[j][0].z, rv[j][0].w};
            const float v1[4] = {rv[j][1].x, rv[j][1].y, rv[j][1].z, rv[j][1].w};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float a0 = v0[u] * DN_VS, a1 = v1[u] * DN_VS;
                const _Float16 h0 = (_Float16)a0, h1 = (_Float16)a1;
                const _Float16 l0 = (_Float16)(a0 - (float)h0), l1 = (_Float16)(a1 - (float)h1);
                const unsigned hw = (unsigned)__builtin_bit_cast(unsigned short, h0) | ((unsigned)__builtin_bit_cast(unsigned short, h1) << 16);
                const unsigned lw = (unsigned)__builtin_bit_cast(unsigned short, l0) | ((unsigned)__builtin_bit_cast(unsigned short, l1) << 16);
                const int o = it_lds[j] + u * DN_CSTR;                       // even half index: 4-byte aligned
                *reinterpret_cast<unsigned*>(&spl[o]) = hw;
                *reinterpret_cast<unsigned*>(&spl[DN_PLANE_H + o]) = lw;
            }
        }
    };

    const bool second = i >= 16;
    // key row fed to MFMA row rho: bits (a b c dd) -> (a c b dd), so that a lane's registers are two runs of 8 consecutive keys
    const int prow = (i & 0x13) | ((i & 8) >> 1) | ((i & 4) << 1);

    if (tile0 < tile1) { stage_x(tile0, 0); load_region(tile0); }
    for (int e = tid; e < (2 * DN_PLANE_H + 64) / 2; e += DN_THREADS) reinterpret_cast<unsigned*>(spl)[e] = 0u;   // pad pixels stay zero
    __syncthreads();
    if (tile0 < tile1) store_region();
    dma_wait_all();
    __syncthreads();

    for (int tile = tile0; tile < tile1; ++tile) {
        const int cur = (tile - tile0) & 1;
        if (tile + 1 < tile1) { stage_x(tile + 1, cur ^ 1); load_region(tile + 1); }
        const int ty = tile / a.tiles_per_row, jy0 = ty * DN_TH, jx0 = (tile - ty * a.tiles_per_row) * DN_TW;

        // ---- scores of 32 keys x this lane's query ----------------------------------------------------------------
        // the first two waves of a query tile split the 13 K-blocks of the 196-term sum and publish their partial sums; all four
        // waves add them in the same order (identical scores in every wave, no redundant matrix work)
        f32x16 mine, cross;
#pragma unroll
        for (int r = 0; r < 16; ++r) { mine[r] = 0.f; cross[r] = 0.f; }
        const unsigned short* kp = &sm[cur][prow * DSH + 8 * h];
        if (!(a.variant & 2) && part < 2) {
#pragma unroll
            for (int kb = 0; kb < DN_KB; ++kb) {
                if ((kb < 7) != (part == 0)) continue;                       // wave-uniform: K blocks 0-6 / 7-12
                const dnh8 k_hi = __builtin_bit_cast(dnh8, *reinterpret_cast<const uint4*>(kp + 16 * kb));
                const dnh8 k_lo = __builtin_bit_cast(dnh8, *reinterpret_cast<const uint4*>(kp + DN_XPART + 16 * kb));
                const dnh8 q_hi = __builtin_bit_cast(dnh8, *reinterpret_cast<const uint4*>(qrow + 16 * kb));
                const dnh8 q_lo = __builtin_bit_cast(dnh8, *reinterpret_cast<const uint4*>(qrow + 64 * DSH + 16 * kb));
                cross = __builtin_amdgcn_mfma_f32_32x32x16_f16(k_hi, q_lo, cross, 0, 0, 0);
                mine = __builtin_amdgcn_mfma_f32_32x32x16_f16(k_hi, q_hi, mine, 0, 0, 0);
                cross = __builtin_amdgcn_mfma_f32_32x32x16_f16(k_lo, q_hi, cross, 0, 0, 0);
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) mine[r] += cross[r];
        if (part < 2) {
            float* ex = sx + ((qt * 2 + part) * 16) * 64 + lane;              // [query tile][K half][register][lane]
#pragma unroll
            for (int r = 0; r < 16; ++r) ex[r * 64] = mine[r];
        }
        __syncthreads();
        const float* e0 = sx + ((qt * 2 + 0) * 16) * 64 + lane;
        const float* e1 = sx + ((qt * 2 + 1) * 16) * 64 + lane;
        // ---- logits, weights: register r holds key 16 (r >> 3) + 8 h + (r & 7) of the tile = pixel (2 (r >> 3) + h, r & 7).
        // All four waves of a query tile need all 16 weights of a lane; each forms four of them (logit, exponential, fp16
        // split: ~25 VALU operations per weight) and they are exchanged through LDS as packed halfs -----------------------
        float zt = 0.f, zpt = 0.f;                         // this tile's sums in fp32, one fp64 add per tile
        unsigned passmask = 0;
        _Float16 hq[4], lq[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int r = 4 * part + u;                                       // wave-uniform
            const float sc = (e0[r * 64] + e1[r * 64]) * (1.0f / (DN_FS * DN_FS));
            const bool valid = (jx0 + (r & 7) < g.W) && (jy0 + 2 * (r >> 3) + h < g.H);   // pixel (row 2 (r>>3) + h, column r & 7) of the tile
            bool pass;
            const float l = dn_logit(sc, mtq, bsq, pass);
            const float p = (a.variant & 16) ? 0.5f : (valid ? __expf(fminf(l - m_run, 0.f)) : 0.f);      // (the bound holds; the clamp is a seat belt)
            zt += p;
            pass = pass && valid;
            const float pp = pass ? p : 0.f;
            zpt += pp;
            passmask |= (pass ? 1u : 0u) << u;
            const float ps = pp * DN_PS;
            hq[u] = (_Float16)ps;
            lq[u] = (_Float16)(ps - (float)hq[u]);
        }
        z_run += (double)zt; zp_run += (double)zpt; deg += __popc(passmask);
        unsigned char* pq = reinterpret_cast<unsigned char*>(&spq[qt][lane][0]);   // 80 B per lane: hi[16] | lo[16] | pad
        {
            typedef _Float16 dnh4 __attribute__((ext_vector_type(4)));
            const dnh4 hv = {hq[0], hq[1], hq[2], hq[3]}, lv = {lq[0], lq[1], lq[2], lq[3]};
            *reinterpret_cast<dnh4*>(pq + 8 * part) = hv;
            *reinterpret_cast<dnh4*>(pq + 32 + 8 * part) = lv;
        }
        __syncthreads();
        dnh8 p_hi[2][2], p_lo[2][2];                       // the weights of BOTH query tiles (lane = query i, key half h)
#pragma unroll
        for (int qq = 0; qq < 2; ++qq) {
            const unsigned char* pv = reinterpret_cast<const unsigned char*>(&spq[qq][lane][0]);
            p_hi[qq][0] = *reinterpret_cast<const dnh8*>(pv);      p_hi[qq][1] = *reinterpret_cast<const dnh8*>(pv + 16);
            p_lo[qq][0] = *reinterpret_cast<const dnh8*>(pv + 32); p_lo[qq][1] = *reinterpret_cast<const dnh8*>(pv + 48);
        }
        // ---- out^T[col][q] += V[key][col] * p[q][key] ------------------------------------------------------------------
        const unsigned char* planes = reinterpret_cast<const unsigned char*>(spl);
        if (!(a.variant & 1)) {
            dn_pv(acc, planes, i & 15, h, second, p_hi, p_lo, ct0, ctn, a.variant);
        }
        dma_wait_all();
        __syncthreads();                                   // everyone is done with this tile's planes and features
        if (tile + 1 < tile1 && !(a.variant & 4)) store_region();   // published by the next tile's exchange barrier
    }

    // ---- partial results of this key range ------------------------------------------------------------------------------
    const size_t orow = ((size_t)split * a.B + b) * g.L + qc;
    {
        // every wave summed its quarter of the weights: halves h first, then the four waves of the query tile in wave order
        const double z2 = z_run + __shfl_xor(z_run, 32), zp2 = zp_run + __shfl_xor(zp_run, 32);
        const int d2 = deg + __shfl_xor(deg, 32);
        if (h == 0) { szz[qt][part][i][0] = z2; szz[qt][part][i][1] = zp2; sdg[qt][part][i] = d2; }
    }
    __syncthreads();
    if (qvalid) {
        if (part == 0 && h == 0) {
            a.part_m[orow] = m_run;
            a.part_z[2 * orow] = (szz[qt][0][i][0] + szz[qt][1][i][0]) + (szz[qt][2][i][0] + szz[qt][3][i][0]);
            a.part_z[2 * orow + 1] = (szz[qt][0][i][1] + szz[qt][1][i][1]) + (szz[qt][2][i][1] + szz[qt][3][i][1]);
            a.part_deg[orow] = (sdg[qt][0][i] + sdg[qt][1][i]) + (sdg[qt][2][i] + sdg[qt][3][i]);
        }
    }
#pragma unroll
    for (int qq = 0; qq < 2; ++qq) {                       // this wave's column tiles of both query tiles
        const int q2 = qb * 64 + qq * 32 + i;
        if (q2 < g.L) dn_store(acc[qq], a.part_acc + (((size_t)split * a.B + b) * g.L + q2) * P, h, ct0, ctn);
    }
}

// merge the key-range partials of a query: M = max M_s, Z = sum Z_s e^(M_s - M), agg = sum acc_s e^(M_s - M) / Z
// (all partials of a query share one shift today, so the weights are 1; kept general).  One wave per query, float4 columns.
__global__ __launch_bounds__(256) void dense_combine_kernel(DenseArgs a, float* __restrict__ agg, int32_t* __restrict__ deg_out,
                                                            float* __restrict__ rowsum_out, float* __restrict__ lse_out) {
    const int lane = threadIdx.x & 63;
    const size_t nq = (size_t)a.B * a.g.L;
    const size_t ql = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);           // (b, q) flattened
    if (ql >= nq) return;
    // lane s < splits holds partial s's scalars
    const bool has = lane < a.splits;
    const float ms = has ? a.part_m[lane * nq + ql] : -__builtin_inff();
    float M = ms;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) M = fmaxf(M, __shfl_xor(M, o));
    const float wl = (has && ms != -__builtin_inff()) ? __expf(ms - M) : 0.f;
    double z = has ? a.part_z[2 * (lane * nq + ql)] * (double)wl : 0.0;
    double zp = has ? a.part_z[2 * (lane * nq + ql) + 1] * (double)wl : 0.0;
    int deg = has ? a.part_deg[lane * nq + ql] : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { z += __shfl_xor(z, o); zp += __shfl_xor(zp, o); deg += __shfl_xor(deg, o); }
    const float invz = (float)(1.0 / z);
    float4 acc[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) acc[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int s = 0; s < a.splits; ++s) {                                      // fixed order
        const float ws = __shfl(wl, s);
        const float4* row = reinterpret_cast<const float4*>(a.part_acc + (s * nq + ql) * P);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c4 = lane + 64 * u;
            if (c4 < P / 4) {
                const float4 v = row[c4];
                acc[u].x += v.x * ws; acc[u].y += v.y * ws; acc[u].z += v.z * ws; acc[u].w += v.w * ws;
            }
        }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int c4 = lane + 64 * u;
        if (c4 < P / 4)
            reinterpret_cast<float4*>(agg + ql * P)[c4] = make_float4(acc[u].x * invz, acc[u].y * invz, acc[u].z * invz, acc[u].w * invz);
    }
    if (lane == 0) {
        if (deg_out) deg_out[ql] = deg;
        if (rowsum_out) rowsum_out[ql] = (float)(zp / z);
        if (lse_out) { lse_out[2 * ql] = M; lse_out[2 * ql + 1] = (float)z; }        // A = e^(l - M) / Z for the backward
        a.part_deg[ql] = deg;        // slot of split 0 (read above): final degree, summed up by degree_stats_kernel afterwards
    }
}

// fp32 feature rows [B, rows_in, 204] -> split fp16 rows [B, rows_out, 216] hi and lo, 64 x = hi + lo (pad columns zero)
__global__ void feat_split_kernel(int rows, int rows_in, int rows_out, const float* __restrict__ src,
                                  unsigned short* __restrict__ hi, unsigned short* __restrict__ lo, RangeTag range) {
    const int b = blockIdx.y;
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)rows_out * (DSH / 8)) return;
    const int r = (int)(t / (DSH / 8)), c8 = (int)(t % (DSH / 8));
    unsigned short vh[8], vl[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int c = 8 * c8 + u;
        const float v = (r < rows && c < D) ? src[((size_t)b * rows_in + r) * DS + c] * DN_FS : 0.f;
        if (range.word != nullptr && !(fabsf(v) < RANGE_LIMIT)) *range.word = range.tag;
        const _Float16 h = (_Float16)v;
        vh[u] = __builtin_bit_cast(unsigned short, h);
        vl[u] = __builtin_bit_cast(unsigned short, (_Float16)(v - (float)h));
    }
    const size_t o = (((size_t)b * rows_out + r) * DSH + 8 * c8) / 8;
    reinterpret_cast<uint4*>(hi)[o] = *reinterpret_cast<const uint4*>(vh);
    reinterpret_cast<uint4*>(lo)[o] = *reinterpret_cast<const uint4*>(vl);
}

// row maximum of the screened scores: gmax holds, per query, the 4 largest values of each of its G/4 scan segments
__global__ void dense_rowmax_kernel(size_t n_rows, int G, const float* __restrict__ gmax, float* __restrict__ smax) {
    const int lane = threadIdx.x & 63;
    const size_t row = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n_rows) return;
    float m = 0.f;
    for (int t = lane; t < G; t += 64) m = fmaxf(m, gmax[row * G + t]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if (lane == 0) smax[row] = m;
}

int launch_dense_rowmax(hipStream_t s, size_t n_rows, int G, const float* gmax, float* smax) {
    hipLaunchKernelGGL(dense_rowmax_kernel, dim3((unsigned)((n_rows + 3) / 4)), dim3(256), 0, s, n_rows, G, gmax, smax);
    DAGL_LAUNCH_CHECK("dense_rowmax_kernel");
    return DAGL_OK;
}

int dense_splits(int B, const Grid& g) {
    // One block per CU is resident (LDS), so the grid runs in rounds of 256 blocks.  Few query groups: as many key ranges
    // as still fit one round.  More query groups than CUs: the split count that wastes least of the last round
    // (384 groups: 1 split = 2 rounds of whole ranges, 2 splits = 3 rounds of half ranges).
    const int n = ((g.L + 63) / 64) * B;
    const int n_tiles = ((g.H + 3) / 4) * ((g.W + 7) / 8);
    int s;
    if (n <= 256) {
        s = 256 / n;
    } else {
        s = 1;
        double best = (double)((n + 255) / 256);
        for (int c = 2; c <= 4; ++c) {
            const double cost = (double)((n * c + 255) / 256) / c;
            if (cost < best - 1e-9) { best = cost; s = c; }
        }
    }
    if (s > n_tiles) s = n_tiles;
    if (s > 16) s = 16;
    if (s < 1) s = 1;
    return s;
}

static size_t dn_feat16_bytes(int B, int rows) { return align_up((size_t)B * feat_rows_h(rows) * DSH * sizeof(uint16_t), 256); }

size_t dense_workspace_bytes(int B, const Grid& g) {
    const size_t rows = (size_t)dense_splits(B, g) * B * g.L;
    return align_up(rows * P * sizeof(float), 256) + align_up(rows * sizeof(float), 256) +
           align_up(rows * 2 * sizeof(double), 256) + align_up(rows * sizeof(int32_t), 256) +
           2 * dn_feat16_bytes(B, g.N) + 2 * dn_feat16_bytes(B, g.L);
}

int launch_dense_attend(hipStream_t s, int B, const Grid& g, const float* wq, const float* x, const float* mt,
                        const float* bs, const float* smax, const float* b2p, void* ws, float* agg, int32_t* deg_out,
                        float* rowsum_out, int64_t* stats, RangeTag range, float* lse_out) {
    DenseArgs a;
    a.smax = smax;
    a.variant = 0;
#ifdef DAGL_ABLATION      // debug builds only: the variants give wrong results by construction
    { static const int var = getenv("DAGL_DENSE_VARIANT") ? atoi(getenv("DAGL_DENSE_VARIANT")) : 0; a.variant = var; }
#endif
    a.B = B; a.g = g; a.wq = wq; a.x = x; a.rows_q = feat_rows(g.L); a.rows_x = feat_rows(g.N);
    a.mt = mt; a.bs = bs; a.b2p = b2p;
    a.splits = dense_splits(B, g);
    a.tiles_per_row = (g.W + DN_TW - 1) / DN_TW;
    a.n_tiles = ((g.H + DN_TH - 1) / DN_TH) * a.tiles_per_row;
    a.tiles_per_split = (a.n_tiles + a.splits - 1) / a.splits;
    a.splits = (a.n_tiles + a.tiles_per_split - 1) / a.tiles_per_split;
    const size_t rows = (size_t)dense_splits(B, g) * B * g.L;                // carve with the planned (upper) split count
    char* p = static_cast<char*>(ws);
    a.part_acc = reinterpret_cast<float*>(p); p += align_up(rows * P * sizeof(float), 256);
    a.part_m = reinterpret_cast<float*>(p); p += align_up(rows * sizeof(float), 256);
    a.part_z = reinterpret_cast<double*>(p); p += align_up(rows * 2 * sizeof(double), 256);
    a.part_deg = reinterpret_cast<int32_t*>(p); p += align_up(rows * sizeof(int32_t), 256);
    uint16_t* xh = reinterpret_cast<uint16_t*>(p); p += dn_feat16_bytes(B, g.N);
    uint16_t* xl = reinterpret_cast<uint16_t*>(p); p += dn_feat16_bytes(B, g.N);
    uint16_t* qh = reinterpret_cast<uint16_t*>(p); p += dn_feat16_bytes(B, g.L);
    uint16_t* ql = reinterpret_cast<uint16_t*>(p);
    a.x_hi = xh; a.x_lo = xl; a.wq_hi = qh; a.wq_lo = ql;
    a.rows_xh = feat_rows_h(g.N); a.rows_qh = feat_rows_h(g.L);
    {
        const size_t nx = (size_t)a.rows_xh * (DSH / 8), nq8 = (size_t)a.rows_qh * (DSH / 8);
        hipLaunchKernelGGL(feat_split_kernel, dim3((unsigned)((nx + 255) / 256), B), dim3(256), 0, s, g.N, a.rows_x, a.rows_xh, x, xh, xl, range);
        DAGL_LAUNCH_CHECK("feat_split_kernel");
        hipLaunchKernelGGL(feat_split_kernel, dim3((unsigned)((nq8 + 255) / 256), B), dim3(256), 0, s, g.L, a.rows_q, a.rows_qh, wq, qh, ql, range);
        DAGL_LAUNCH_CHECK("feat_split_kernel");
    }
    const int n_qblocks = (g.L + 63) / 64;
    hipLaunchKernelGGL(dense_attend_kernel, dim3(n_qblocks * a.splits, B), dim3(DN_THREADS), 0, s, a);
    DAGL_LAUNCH_CHECK("dense_attend_kernel");
    hipLaunchKernelGGL(dense_combine_kernel, dim3((unsigned)(((size_t)B * g.L + 3) / 4)), dim3(256), 0, s, a, agg, deg_out, rowsum_out, lse_out);
    DAGL_LAUNCH_CHECK("dense_combine_kernel");
    // total edges, max degree, queries beyond the neighbour lists' width (no per-query atomics)
    return launch_degree_stats(s, (size_t)B * g.L, a.part_deg, stats, nullptr, DAGL_LIST_CAP);
}

}  // namespace dagl
