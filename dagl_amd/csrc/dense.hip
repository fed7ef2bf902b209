// Dense neighbourhoods of the adaptive mask (DN_Gray/model/dagl.py:250-264 when most keys pass, e.g. with
// default-initialised thr/bias heads ~95 % of them; 0.88-1.0 with every trained weight set this repository has seen): per-query
// neighbour lists stop making sense, so this path is the reference's dense formulation, streamed -- S = Wq X^T, mask, softmax
// over ALL keys, A V -- in one pass over the keys, nothing of size L x N ever stored.
//
// Round 4 shape (round 3: three barrier-separated phases per key tile, S exchanged through the LDS, value planes re-laid out by
// the VALU, 2.07 ms per head at 256^2):
//   block = 64 queries x 12 waves (three per SIMD), key tiles of 32 keys = an 8 x 4 pixel block, ONE barrier per tile, two kinds of
//   wave that work on different tiles between two barriers:
//   producers (waves 0-3, one per SIMD, s_setprio 2): query group qg = wave (16 queries, fragments in registers)
//     S(t+3)  v_mfma_f32_16x16x32_f16 on split features (64 x = hi + lo; three products): the COMPLETE scores of 16 queries x
//             32 keys (42 multiplies), nothing about S is exchanged between waves.  A lane ends up with 2 x 4 consecutive keys of
//             one query;
//     w(t+2)  logits in the reference's fp32 expression order, p = e^(l - M') with M' the row's largest logit, known
//             before the pass (rowmax_exact_kernel: a top-1 screen + exact rescoring; the softmax is shift invariant, nothing is ever rescaled),
//             split 2^15 p = hi + lo and handed to the consumers through the LDS in the K-layout of the next MFMA (the only
//             exchange of the tile).  The VALU operations of w(t+2) (~13 a key since round 5, packed fp32) are issued BETWEEN the
//             multiplies of S(t+3) (fenced slots: fragment reads two slots ahead, three multiplies, a share of the weights): a
//             short multiply waits ~50 cycles for the pipe behind the consumers' long ones, and scores-then-weights was a
//             5 000-cycle chain per tile;
//     the key tiles' LDS-DMA requests (7 pieces per producer under two M0 set-ups), four tiles ahead, per-lane offsets carried
//     from tile to tile.
//   consumers (waves 4-11, two per SIMD):
//     A V(t)  v_mfma_f32_32x32x16_f16, out^T[col][q] += V[key][col] p[q][key], three split products, every consumer three of the 24
//             32-column tiles (two taps x 16 channels) for BOTH query tiles.  The value operand: 8 consecutive keys of one column =
//             8 consecutive pixels of one channel.  The tile's value-map region (10 rows x 14 pixels) is staged as it lies in
//             memory -- NHWC fp16 hi / lo, 32 B per pixel, by LDS-DMA from maps split once per call (the consumers' 1.5 pieces
//             each) -- and ds_read_b64_tr_b16 transposes [4 pixels][16 channels] blocks on the way into the registers: a patch
//             tap's kw shift is a whole-pixel (32-byte) address offset, no funnel shifts.
//     The 49th tap (16 columns) goes through v_mfma_f32_16x16x32_f16 (16 channels x 16 queries, K = the tile's 32 keys), one
//     query group of 16 per consumer 0-3.
//   Two role PATHS with the same number of barriers (round 5): from the role test to the block's last barrier the roles share no
//   code and everything a role keeps per lane is formed inside its path -- 165 registers, no spill (with shared segments between
//   two role loops: 168 with 26 spilled, the producers' query fragments held across the multiplying waves' loop).  Weights and
//   value regions are three stages deep: the producers run two tiles ahead, a multiplying wave fetches the next tile's first
//   weights and value fragment under its last multiplies (across the barrier).
//   The first round-4 shape had both kinds of work in every wave (8 waves, the two of a SIMD in opposite order): the pipes were
//   55 % busy -- every wave spent half its time in code that does not multiply.  This one: 63 % (trained features, 1.38 -> 1.28 ms),
//   synthetic map 0.71 -> 0.62 ms, leaf-tile batch 0.69 -> 0.65 ms (profiles/r04_pmc_dense_*.json, r04_ab_dense_roles.log).
//   Exactly-zero weights: logits reach hundreds, and 2^15 p rounds to hi = lo = 0 below l < M' - 27.7; a (64 queries x 16 keys)
//   granule whose weights are ALL zero contributes exactly nothing to A V, so its multiplies are skipped (a wave-uniform test of
//   the weight fragments: bit-identical results).  On synthetic N(0,1) maps at default init ~60-80 % of the granules are zero
//   (profiles/r04_dense_zero_granules.log); with the trained checkpoint's features (logits of 5-70) none are -- that regime runs
//   every multiply.
//   Staging: key features hi | lo (3 x 28 KiB), value region (3 x 12 KiB), weights (3 x 10 KiB); key rows at a 28-slot pitch with
//   the low slot bits XORed by a function of the row so that the S fragments' ds_read_b128 are conflict-free for the 16x16x32
//   operand pattern.
//   key range split over `splits` blocks per 64 queries; dense_combine_kernel merges the partial sums and rows.
//
// Matrix time per (64 query, 32 key) tile: 8 x 21 + 12 multiplies of 16 cycles + 288 of 32 cycles.
#include <stdlib.h>
#include <type_traits>

#include "dagl_common.h"

namespace dagl {

typedef _Float16 dnh8 __attribute__((ext_vector_type(8)));
typedef _Float16 dnh4 __attribute__((ext_vector_type(4)));
typedef short dns4 __attribute__((__vector_size__(4 * sizeof(short))));
typedef short dns8 __attribute__((ext_vector_type(8)));
typedef float dnf2 __attribute__((ext_vector_type(2)));

constexpr int DN_KS = 7;                              // k-steps of 32 features (224 >= 196; a feature row holds 216 halfs)
constexpr int DN_KPITCH = 4 * DN_KS;                  // 16-byte slots per staged key row
constexpr int DN_KPART_B = 32 * DN_KPITCH * 16;       // bytes of one part (hi or lo) of a staged key tile: 14 pieces of 1 KiB
constexpr int DN_KTILE_B = 2 * DN_KPART_B;            // hi | lo
constexpr int DN_TW = 8, DN_TH = 4;                   // a key tile = 8 x 4 pixels (32 keys): narrow maps waste little of it
constexpr int DN_RH = DN_TH + KS - 1, DN_RW = DN_TW + KS - 1;   // value-map region of a tile: 10 rows x 14 pixels
constexpr int DN_VPX = 18;                            // staged pixels per region row: 576-byte pitch = 64 B mod 256, so that the two taps
                                                      // of a column tile that sit on different rows (kw = 6 | 0) use disjoint banks
constexpr int DN_VPART_B = 6 * 1024;                  // one part of a staged region: 10 x 18 x 32 B = 5760 in 6 pieces
constexpr int DN_VTILE_B = 2 * DN_VPART_B;
constexpr int DN_PQ_ENTRY = 80;                       // a lane's weights of a tile: hi[16] | lo[16] halfs + 16 B (odd slot count)
constexpr int DN_PQ_B = 2 * 64 * DN_PQ_ENTRY;         // both query tiles
// A V: every wave takes 3 of the 24 column tiles for BOTH query tiles of the block (a value fragment then feeds six multiplies
// instead of three); the 49th tap's 16 columns go through v_mfma_f32_16x16x32_f16 (16 channels x 16 queries, K = the tile's 32
// keys: three multiplies of 16 cycles per query group of 16, waves 0-3 one group each) -- as a 25th 32-column tile whose second
// half was a dummy tap it cost wave 7 twelve 32-cycle multiplies per tile and EVERY wave 32 accumulator registers (the array is
// sized for the largest share)
constexpr int DN_CTMAX = 3;
__host__ __device__ constexpr int dn_ct_start(int w8) { return 3 * w8; }
constexpr float DN_PS = 32768.0f, DN_VS = 16.0f;      // power-of-two pre-scaling of the split operands (weights are <= 1: 2^15 is the
                                                      // largest power of two fp16 holds)
#ifndef DAGL_DN_SLACK
#define DAGL_DN_SLACK 18.5f
#endif
// largest tolerated (shift M' - largest logit of a row) of the first pass: a weight is stored to 2^-24 / 2^15 = 2^-39 absolute (hi + lo,
// the lo part denormal), i.e. to 2^-40 e^slack relative to the row's largest one: 1e-4 at 18.5 for a row that ONE key dominates
// (rows with many comparable keys average it down)
constexpr float DN_SHIFT_SLACK = DAGL_DN_SLACK;

__device__ __forceinline__ float dn_logit(float s, float mtq, float bsq, bool& pass) {
    const float m = (s - mtq) + bsq;                  // same expression order as dagl.py:256
    pass = m > 0.f;
    return pass ? __fmul_rn(__fmul_rn(s, m), SOFTMAX_SCALE) : 0.f;
}

// the softmax shift of a row.  First pass: the logit of smax[q] (rowmax_exact_kernel, screen.hip): the row's largest score -- exact
// (fp64-accumulated from the fp32 features) where the bf16 bounds of the screen lie more than DN_BOUND_OK logit units apart, their
// upper bound otherwise -- inflated by 6e-6, more than the split-fp16 products of this kernel can differ from the exact score (three
// products of 22-bit operands accumulated in fp32: <= 2e-6), so that no weight exceeds 1; l grows with S and masked keys have l = 0.
// Second pass: the largest logit itself, as the first pass formed it.
__device__ __forceinline__ float dn_shift(const DenseArgs& a, size_t ql) {
    if (a.pass == 1) return a.m_exact[ql];
    const float sub = a.smax[ql] * (1.0f + 6e-6f);
    bool ps;
    return fmaxf(dn_logit(sub, a.mt[ql], a.bs[ql], ps), 0.f);
}

__device__ __forceinline__ dns4 dn_tr16(unsigned lds_byte_addr) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) dns4*)(uintptr_t)lds_byte_addr);
}
typedef unsigned dnu4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ dnu4 dn_lds128(unsigned lds_byte_addr) {
    return *(const __attribute__((address_space(3))) dnu4*)(uintptr_t)lds_byte_addr;
}

// N LDS-DMA pieces (1 KiB each, consecutive in the LDS from lds_byte_addr) under one M0 set-up: piece j takes the instruction's
// immediate offset 1024 j, which advances the LDS AND the global address -- the caller's per-lane byte offsets are relative to
// src_uniform - 4096 and carry + 4096 - 1024 j (unsigned 32-bit offsets: the bias keeps them from wrapping)
template <int N>
__device__ __forceinline__ void dn_glds(const void* src_biased, unsigned lds_byte_addr, unsigned o0, unsigned o1, unsigned o2, unsigned o3) {
    unsigned keep;
    if (N == 1)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %3, %1\n\t"
                     "s_mov_b32 m0, %0" : "=&s"(keep) : "s"(src_biased), "s"(lds_byte_addr), "v"(o0) : "memory");
    else if (N == 2)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %3, %1\n\tglobal_load_lds_dwordx4 %4, %1 offset:1024\n\t"
                     "s_mov_b32 m0, %0" : "=&s"(keep) : "s"(src_biased), "s"(lds_byte_addr), "v"(o0), "v"(o1) : "memory");
    else if (N == 3)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %3, %1\n\tglobal_load_lds_dwordx4 %4, %1 offset:1024\n\t"
                     "global_load_lds_dwordx4 %5, %1 offset:2048\n\t"
                     "s_mov_b32 m0, %0" : "=&s"(keep) : "s"(src_biased), "s"(lds_byte_addr), "v"(o0), "v"(o1), "v"(o2) : "memory");
    else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %3, %1\n\tglobal_load_lds_dwordx4 %4, %1 offset:1024\n\t"
                     "global_load_lds_dwordx4 %5, %1 offset:2048\n\tglobal_load_lds_dwordx4 %6, %1 offset:3072\n\t"
                     "s_mov_b32 m0, %0" : "=&s"(keep) : "s"(src_biased), "s"(lds_byte_addr), "v"(o0), "v"(o1), "v"(o2), "v"(o3) : "memory");
}

__device__ __forceinline__ void dn_store(const f32x16 (&acc)[DN_CTMAX], float* po, int h, int ct0) {
    constexpr float inv = 1.0f / (DN_PS * DN_VS);
#pragma unroll
    for (int t = 0; t < DN_CTMAX; ++t) {
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const int col = 32 * (ct0 + t) + 8 * gq + 4 * h;                 // acc[t][4 gq + u] = out[q][col + u]
            *reinterpret_cast<float4*>(po + col) = make_float4(acc[t][4 * gq] * inv, acc[t][4 * gq + 1] * inv,
                                                                   acc[t][4 * gq + 2] * inv, acc[t][4 * gq + 3] * inv);
        }
    }
}

struct DnFrag { dns4 h0, h1, l0, l1; };
// region pixel (row kh + 2 kb + h, column kw + 4 j + (lane & 15) / 4): keys 8 h + 4 j .. of a k-block, hi and lo
__device__ __forceinline__ DnFrag dn_vfrag(unsigned va) {
    DnFrag f;
    f.h0 = dn_tr16(va); f.h1 = dn_tr16(va + 128); f.l0 = dn_tr16(va + DN_VPART_B); f.l1 = dn_tr16(va + DN_VPART_B + 128);
    return f;
}
// A V of one k-block (16 keys) for this wave's three column tiles and BOTH query tiles: per column tile one value fragment, six
// multiplies, the two accumulators' chains interleaved.  Straight-line code: the zero-granule skip is decided per k-block by the
// caller (64 queries x 16 keys).  Per-query-tile skipping inside here -- uniform branches around groups of three multiplies --
// skips 5-10 % more granules on synthetic maps and is 2-3 % slower where nothing is zero; compile-time instantiations per
// (query tile 0 live, query tile 1 live) made the register allocator spill ~290 registers (profiles/r04_ab_dense_variants.log).
__device__ __forceinline__ DnFrag dn_pv(f32x16 (&acc)[2][DN_CTMAX], unsigned va_kb, const unsigned (&vt_off)[DN_CTMAX],
                                        const dnh8 (&p_hi)[2], const dnh8 (&p_lo)[2], DnFrag f /* = dn_vfrag(va_kb + vt_off[0]) */,
                                        unsigned va_next) {
    // The hi half of column tile t + 1's fragment is requested before tile t's multiplies, its lo half after the two multiplies that
    // use tile t's lo half (into the registers those free: 12 fragment registers live instead of 16 -- the loop sits at the register
    // cap); before the last tile's multiplies: the first fragment of the NEXT k-block (va_next: this tile's second k-block, or the
    // next tile's first one), returned to the caller.  The weights' lo halves are used last: they are requested at the top of the
    // k-block, the hi halves one k-block ahead.
#pragma unroll
    for (int t = 0; t < DN_CTMAX; ++t) {
        const unsigned an = t + 1 < DN_CTMAX ? va_kb + vt_off[t + 1] : va_next;
        DnFrag fn;
        fn.h0 = dn_tr16(an); fn.h1 = dn_tr16(an + 128);
#ifdef DAGL_DN_FNFULL
        fn.l0 = dn_tr16(an + DN_VPART_B); fn.l1 = dn_tr16(an + DN_VPART_B + 128);
#endif
        __builtin_amdgcn_sched_barrier(0);
        const dns8 vh = {f.h0[0], f.h0[1], f.h0[2], f.h0[3], f.h1[0], f.h1[1], f.h1[2], f.h1[3]};
        const dns8 vl = {f.l0[0], f.l0[1], f.l0[2], f.l0[3], f.l1[0], f.l1[1], f.l1[2], f.l1[3]};
        const dnh8 v_hi = __builtin_bit_cast(dnh8, vh), v_lo = __builtin_bit_cast(dnh8, vl);
        acc[0][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(v_lo, p_hi[0], acc[0][t], 0, 0, 0);
        acc[1][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(v_lo, p_hi[1], acc[1][t], 0, 0, 0);
#ifndef DAGL_DN_FNFULL
        __builtin_amdgcn_sched_barrier(0);
        fn.l0 = dn_tr16(an + DN_VPART_B); fn.l1 = dn_tr16(an + DN_VPART_B + 128);
        __builtin_amdgcn_sched_barrier(0);
#endif
        acc[0][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(v_hi, p_hi[0], acc[0][t], 0, 0, 0);
        acc[1][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(v_hi, p_hi[1], acc[1][t], 0, 0, 0);
        acc[0][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(v_hi, p_lo[0], acc[0][t], 0, 0, 0);
        acc[1][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(v_hi, p_lo[1], acc[1][t], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        f = fn;
    }
    return f;
}

#ifdef DAGL_ABLATION
// phase clocks (DAGL_DENSE_VARIANT & 64): every wave sums shader-clock deltas per phase (s_memtime also waits for the wave's
// outstanding LDS operations: ~10 % perturbation)
#define DN_PH(i) do { if (clocks) { const unsigned long long n_ = __builtin_readcyclecounter(); phc[i] += (unsigned)(n_ - t_last); t_last = n_; } } while (0)
#else
#define DN_PH(i) do { } while (0)
#endif

constexpr int DN_THREADS = 768;                    // 12 waves: 4 form scores / weights (one per SIMD), 8 multiply (two per SIMD)
__global__ __launch_bounds__(DN_THREADS) void dense_attend_kernel(DenseArgs a) {
    __shared__ __attribute__((aligned(1024))) unsigned char sm[3 * DN_KTILE_B];     // 84 KiB: key-feature tiles hi | lo, three stages
    __shared__ __attribute__((aligned(1024))) unsigned char sv[3 * DN_VTILE_B];     // 36 KiB: value regions hi | lo, three stages
    __shared__ __attribute__((aligned(16))) unsigned char spq[3 * DN_PQ_B];         // 30 KiB: the tiles' weights, three stages
    __shared__ double szz[4][64][2];                                                // 4 KiB: the lanes' shares of a query's sums (end of the block)
    __shared__ int sdg[4][64];
    __shared__ float slt[4][64];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.y;
    const Grid& g = a.g;
    const int n_qblocks = (g.L + 63) / 64;
    const int qb = blockIdx.x % n_qblocks, split = blockIdx.x / n_qblocks;
#ifdef DAGL_ABLATION
    unsigned phc[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
    const bool clocks = (a.variant & 64) && a.phase_out != nullptr;
    unsigned long long t_last = clocks ? __builtin_readcyclecounter() : 0ull;
#endif
    const int tile0 = split * a.tiles_per_split;
    int tile1 = tile0 + a.tiles_per_split;
    if (tile1 > a.n_tiles) tile1 = a.n_tiles;
    const int n_t = tile1 - tile0;
    // first pass: the row's shift from rowmax_exact_kernel's score (dn_shift);
    // second pass (see dense_combine_kernel): only the blocks of 64 queries flagged by the first one, shifted by their rows' EXACT largest logit
    if (a.pass == 1 && a.redo_blk[b * n_qblocks + qb] == 0) return;
    if (a.pass == 0 && split == 0 && tid == 0) {
        a.redo_blk[b * n_qblocks + qb] = 0;                                              // (set by the first combine, behind this launch)
        if (blockIdx.x == 0 && b == 0) *a.redo_count = 0;                                // blocks the first combine flags (info->dense_rerun_blocks)
    }

    // ---- roles ------------------------------------------------------------------------------------------------------------------
    // waves 0-3 ("producers", one per SIMD): scores and weights of query group qg = wave (16 queries) against the tile's 32 keys
    // (two key groups of 16, one after the other), and the key tiles' LDS-DMA requests.  lane = (query c16, key quad gk): keys
    // 16 kg + 4 gk + r.
    // waves 4-11 ("consumers", two per SIMD): A V.  lane = (column / query i, key half h) of the 32x32x16 multiply; consumer cw =
    // column tiles [3 cw, 3 cw + 3) of both query tiles; consumers 0-3 also the 49th tap of query group cw; they request the value
    // regions.
    // The two roles are two code paths from here to the block's last barrier (the same number of barriers on both): everything a
    // role keeps per lane is formed inside its path, so that nothing of one role holds registers on the other's (both loops run at
    // the 168-register cap of three waves per SIMD; with shared segments in between, the query fragments of the producers were
    // kept -- and spilled -- across the multiplying waves' loop).
    const bool producer = wave < 4;
    const unsigned lds_sm = __builtin_amdgcn_readfirstlane(lds_addr_of(sm));
    const unsigned lds_sv = __builtin_amdgcn_readfirstlane(lds_addr_of(sv));
    const unsigned lds_pq = __builtin_amdgcn_readfirstlane(lds_addr_of(spq));
    auto lane_now = [&]() { int l = (int)(threadIdx.x & 63u); asm volatile("" : "+v"(l)); return l; };
    const int xwrap = a.tiles_per_row * DN_TW;
    auto next_tile = [&](int& y, int& x) { x += DN_TW; if (x >= xwrap) { x = 0; y += DN_TH; } };
    const int ty0 = (tile0 / a.tiles_per_row) * DN_TH, tx0 = (tile0 % a.tiles_per_row) * DN_TW;

    if (producer) {
        // ================================================= scores and weights =================================================
        const int lane = lane_now();
        const int qg = wave;
        const int c16 = lane & 15, gk = lane >> 4;
        const int qs = qb * 64 + 16 * qg + c16;                                    // the query of this lane's scores
        const int qsc = qs < g.L ? qs : g.L - 1;
        const size_t qlin = (size_t)b * g.L + qsc;
        const float mtq = a.mt[qlin], bsq = a.bs[qlin];
        const float m_run = dn_shift(a, qlin);
        // ---- the key tiles' LDS-DMA: part = wave >> 1 (hi | lo), the part's 14 pieces of 1 KiB over its two waves, 7 each, under two M0
        // set-ups (piece jj of a group carries the immediate offset 1024 jj, which advances the LDS AND the global address -- its lane
        // offset is taken back by as much; one piece per set-up cost the wave ~200 cycles a piece).  Position p of a piece: 16-byte slot
        // p of the part's LDS image = key row p / 28 of the tile (pixel row row >> 3, pixel row & 7), physical slot p % 28.  The pieces'
        // per-lane byte offsets are CARRIED from tile to tile (+ 8 pixels per tile) and formed anew only where the tile row wraps
        // (seven 64-bit multiply-adds, clamps and quarter-rate multiplies per tile otherwise) -- from the lane number again, so that
        // the plan's constants hold no registers across the loop.
        const int spart = wave >> 1, kp0 = 7 * (wave & 1);
        constexpr int kpn = 7;
        const unsigned char* xsrc = reinterpret_cast<const unsigned char*>((spart ? a.x_lo : a.x_hi) + (size_t)b * a.rows_xh * DSH) - 4096;
        unsigned k_run[kpn];
        auto key_run_set = [&](int jy0, int jx0) {
            const int ln = lane_now();
#pragma unroll
            for (int j = 0; j < kpn; ++j) {
                const int p = (kp0 + j) * 64 + ln;
                const int row = p / DN_KPITCH, phys = p - row * DN_KPITCH;
                const int logical = (phys & ~3) | ((phys & 3) ^ ((0x1320 >> (4 * ((row >> 2) & 3))) & 3));
                int jy = jy0 + (row >> 3); if (jy > g.H - 1) jy = g.H - 1;       // ragged bottom: a valid row, keys masked below
                k_run[j] = (unsigned)(jy * g.W + jx0) * (unsigned)(DSH * 2) + (unsigned)((row & 7) * (DSH * 2) + logical * 16 + 4096) -
                           1024u * (unsigned)(j < 4 ? j : j - 4);
            }
        };
        auto stage_keys_run = [&](int buf) {
            const unsigned dst = lds_sm + (unsigned)(buf * DN_KTILE_B + spart * DN_KPART_B + kp0 * 1024);
            dn_glds<4>(xsrc, dst, k_run[0], k_run[1], k_run[2], k_run[3]);
            dn_glds<3>(xsrc, dst + 4096u, k_run[4], k_run[5], k_run[6], 0);
        };
        int yk = ty0, xk = tx0;                            // the tile of the next key request
        int yw = ty0, xw = tx0;                            // the tile whose weights are formed next
        for (int t = 0; t < 3; ++t) {                      // tiles 0-2 -> stages 0-2
            if (t < n_t) { key_run_set(yk, xk); stage_keys_run(t); }
            next_tile(yk, xk);
        }
        key_run_set(yk, xk);                               // tile 3: requested in interval -1

        // ---- the query fragments of S: lane (c16, gk) holds features 32 ks + 8 gk .. + 7 of query qs, hi and lo (rows past L are zero
        // guard rows; halfs 216.. of a row do not exist: zero, and the staged key rows' slots 26 / 27 meet only these zeros) ----------
        dnh8 qf_hi[DN_KS], qf_lo[DN_KS];
        {
            const uint4* rh = reinterpret_cast<const uint4*>(a.wq_hi + ((size_t)b * a.rows_qh + (size_t)qb * 64 + 16 * qg + c16) * DSH);
            const uint4* rl = reinterpret_cast<const uint4*>(a.wq_lo + ((size_t)b * a.rows_qh + (size_t)qb * 64 + 16 * qg + c16) * DSH);
#pragma unroll
            for (int ks = 0; ks < DN_KS; ++ks) {
                const bool in = 4 * ks + gk < DSH / 8;
                const int slot = in ? 4 * ks + gk : DSH / 8 - 1;
                uint4 vh = rh[slot], vl = rl[slot];
                if (!in) { vh = make_uint4(0u, 0u, 0u, 0u); vl = vh; }
                qf_hi[ks] = __builtin_bit_cast(dnh8, vh);
                qf_lo[ks] = __builtin_bit_cast(dnh8, vl);
            }
        }
        // Per lane: the sum of 2^15 e^(l - m_run) over its PASSING keys, their number, the number of positions outside the map it met, and
        // its largest score.  Everything else the block reports follows from these at the end: a masked key weighs e^(0 - m_run) whatever its
        // score, so the sum over ALL keys is the passing keys' sum + (valid keys - degree) e^(-m_run); l = 10 S m(S) grows with S on the
        // passing side of a row (m = S - c > 0; below S = 0 it is negative and never the maximum), so the row's largest logit is the
        // logit of its largest score.  (Round 4-5 carried both sums, the degree and the running maximum of the logits per key: ~24 VALU
        // operations a key, 190 a tile, on waves that share their SIMD's issue port with two multiplying waves -- taken out, a call on
        // trained features went from 1.59 to 1.34 ms: profiles/r05_dense_ablation_weights.log.  Now ~13 a key, most of them packed.)
        double zp_run = 0.0;
        float s_top = -1e30f;
        int deg = 0, n_inv = 0;
        // S operand A: key row 16 kg + c16 of the tile, slot (4 ks + gk) with the low bits swizzled by the row (kg = 1: + 16 rows)
        const unsigned ka_off = (unsigned)(c16 * (DN_KPITCH * 16) + ((gk ^ ((0x1320 >> (4 * (c16 >> 2))) & 3)) * 16));
        // where this lane's four weights of key group kg go: entry (query tile, key half, query) of the A V lane that multiplies them
        // (kg = 1: + 16 bytes)
        const unsigned pw_off = (unsigned)((((qg >> 1) * 64 + (gk >> 1) * 32 + 16 * (qg & 1) + c16) * DN_PQ_ENTRY) + (gk & 1) * 8);

        // raw scores of the 32 keys staged in sm[kbuf] against this wave's 16 queries: three split products per key group, 42 multiplies
        struct SAcc { f32x4 hh[2], hl[2], lh[2]; };
        auto scores = [&](int kbuf, SAcc& sa) {
#pragma unroll
            for (int kg = 0; kg < 2; ++kg) { sa.hh[kg] = f32x4{0.f, 0.f, 0.f, 0.f}; sa.hl[kg] = sa.hh[kg]; sa.lh[kg] = sa.hh[kg]; }
            const unsigned kb_addr = lds_sm + (unsigned)(kbuf * DN_KTILE_B) + ka_off;
#pragma unroll
            for (int ks = 0; ks < DN_KS; ++ks)
#pragma unroll
                for (int kg = 0; kg < 2; ++kg) {
                    const unsigned ad = kb_addr + (unsigned)(kg * 16 * (DN_KPITCH * 16) + 64 * ks);
                    const dnh8 k_hi = __builtin_bit_cast(dnh8, dn_lds128(ad));
                    const dnh8 k_lo = __builtin_bit_cast(dnh8, dn_lds128(ad + DN_KPART_B));
                    sa.hl[kg] = __builtin_amdgcn_mfma_f32_16x16x32_f16(k_hi, qf_lo[ks], sa.hl[kg], 0, 0, 0);
                    sa.hh[kg] = __builtin_amdgcn_mfma_f32_16x16x32_f16(k_hi, qf_hi[ks], sa.hh[kg], 0, 0, 0);
                    sa.lh[kg] = __builtin_amdgcn_mfma_f32_16x16x32_f16(k_lo, qf_hi[ks], sa.lh[kg], 0, 0, 0);
                }
        };
        dnf2 sc_next[2][2];                                // scores of the tile whose weights the next interval forms: [key group][pair]
        auto final_scores = [&](const SAcc& sa) {
#pragma unroll
            for (int kg = 0; kg < 2; ++kg)
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) {
                    const dnf2 hh = {sa.hh[kg][2 * pr], sa.hh[kg][2 * pr + 1]}, hl = {sa.hl[kg][2 * pr], sa.hl[kg][2 * pr + 1]},
                               lh = {sa.lh[kg][2 * pr], sa.lh[kg][2 * pr + 1]};
                    sc_next[kg][pr] = (hh + (hl + lh)) * (1.0f / (DN_FS * DN_FS));
                }
        };
        // Weights of two neighbouring keys (registers 2 pr, 2 pr + 1 of key group kg = keys 16 kg + 4 gk + 2 pr .. of the tile = pixel
        // (row 2 kg + (gk >> 1), columns 4 (gk & 1) + 2 pr ..)): logits in the reference's fp32 expression order (dagl.py:256-259),
        // 2^15 e^(l - m_run) through one v_exp_f32 (the scale is an addend of its argument), split hi + lo.  A masked key's weight is
        // exactly zero (its logit is replaced by -1e30 in front of the exponential); positions outside the map (ragged last tiles,
        // wave-uniform test) get a score of -1e30, which masks them, and are counted.
        constexpr float DN_LOG2E = 1.4426950408889634f;
        // (RG: a std::bool_constant -- the tile has positions outside the map.  Two instantiations of the tile's code instead of a
        // test inside it: a branch in the fenced block splits it into basic blocks, the fences stop ordering anything and the compiler
        // sinks all of the weights' arithmetic into one burst in front of the store.)
        auto weight_pair_a = [&](auto RG, int kg, int pr, int jy0, int jx0, dnf2& zpv, int& dt, dnf2& ps) {
            dnf2 sc = sc_next[kg][pr];
            if constexpr (decltype(RG)::value) {
                const bool rowv = jy0 + 2 * kg + (gk >> 1) < g.H;
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const bool valid = rowv && (jx0 + 4 * (gk & 1) + 2 * pr + u < g.W);
                    sc[u] = valid ? sc[u] : -1e30f;
                    n_inv += valid ? 0 : 1;
                }
            }
            const dnf2 m = (sc - mtq) + bsq;                   // same expression order as dagl.py:256
            const bool p0 = m[0] > 0.f, p1 = m[1] > 0.f;
            const dnf2 l = (sc * m) * SOFTMAX_SCALE;
            dnf2 d;
            d[0] = fminf((p0 ? l[0] : -1e30f) - m_run, 0.f);   // (the bound holds; the clamp is a seat belt)
            d[1] = fminf((p1 ? l[1] : -1e30f) - m_run, 0.f);
            const dnf2 x = d * DN_LOG2E + 15.0f;
#ifdef DAGL_ABLATION
            if (a.variant & 128) { ps = x * 1e-3f; } else      // (ablation: no transcendental)
#endif
            { ps[0] = __builtin_amdgcn_exp2f(x[0]); ps[1] = __builtin_amdgcn_exp2f(x[1]); }
            dt += (p0 ? 1 : 0) + (p1 ? 1 : 0);
            s_top = fmaxf(s_top, fmaxf(sc[0], sc[1]));
        };
        auto weight_pair_b = [&](int pr, const dnf2& ps, dnf2& zpv, dnh4& hq, dnh4& lq) {
            zpv += ps;
#ifdef DAGL_ABLATION
            if (a.variant & 256) { hq[2 * pr] = (_Float16)0.004f; hq[2 * pr + 1] = (_Float16)0.003f; lq[2 * pr] = (_Float16)1e-5f; lq[2 * pr + 1] = (_Float16)1e-5f; return; }   // (ablation: no split)
#endif
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const _Float16 hp = (_Float16)ps[u];
                hq[2 * pr + u] = hp;
                lq[2 * pr + u] = (_Float16)(ps[u] - (float)hp);
            }
        };
        auto weight_store = [&](int kg, int pbuf, const dnf2& zpv, int dt, const dnh4& hq, const dnh4& lq) {
            zp_run += (double)(zpv[0] + zpv[1]); deg += dt;    // (fp32 per tile and key group, one fp64 add each)
            unsigned char* pq = spq + pbuf * DN_PQ_B + pw_off + kg * 16;
            *reinterpret_cast<dnh4*>(pq) = hq;
            *reinterpret_cast<dnh4*>(pq + 32) = lq;
        };
        auto tile_ragged = [&](int jy0, int jx0) { return jy0 + DN_TH > g.H || jx0 + DN_TW > g.W; };
        // (prologue) the weights of a tile from sc_next, all at once
        auto weights = [&](int jy0, int jx0, int pbuf) {
#pragma unroll
            for (int kg = 0; kg < 2; ++kg) {
                dnf2 zpv = {0.f, 0.f};
                int dt = 0;
                dnh4 hq, lq;
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) {
                    dnf2 ps;
                    weight_pair_a(std::true_type{}, kg, pr, jy0, jx0, zpv, dt, ps);      // (prologue: always with the test)
                    weight_pair_b(pr, ps, zpv, hq, lq);
                }
                weight_store(kg, pbuf, zpv, dt, hq, lq);
            }
        };

        dma_wait_all();
        __syncthreads();                                   // barrier 1: tiles 0-2 staged
        {
            SAcc s0;
            if (n_t > 0) { scores(0, s0); final_scores(s0); weights(yw, xw, 0); }
            next_tile(yw, xw);
            if (n_t > 1) { scores(1, s0); final_scores(s0); }
        }
        __syncthreads();                                   // barrier 2: weights of tile 0 written, stage 0 of the keys read

        // One barrier per tile; between two barriers the two kinds of wave work on different tiles.  Interval r (r = tile - tile0; the
        // producers start one interval early, r = -1): the producers request the key features of tile r + 4 (-> sm[(r + 1) % 3], read
        // last in interval r - 2), turn the raw scores of tile r + 2 (in their registers since the last interval) into weights
        // (-> spq[(r + 2) % 3]) and form the raw scores of tile r + 3 (sm[r % 3]); the multiplying waves request the value region of
        // tile r + 2 (-> sv[(r + 2) % 3]), multiply tile r (spq[r % 3], sv[r % 3]) and fetch the first weights / value fragment of tile
        // r + 1 under its last multiplies (both complete since the barrier that ended interval r - 1).
        // The producer's multiplies wait for the matrix pipe behind the consumers' (a 32-cycle multiply is not pre-empted: ~50 cycles
        // per short multiply, 42 of them): its ~250 VALU operations of the weights are issued BETWEEN them -- scores of one tile,
        // weights of the previous one, no dependence -- instead of after them (scores, then weights: the chain was 5 000 cycles per tile
        // for 3 000 of matrix work per SIMD, profiles/r04_dense_pc_phases.log).
        __builtin_amdgcn_s_setprio(2);
        int c3 = 2;                                        // (r + 3) % 3
        for (int r = -1; r < n_t; ++r) {
            const int c4 = c3 == 2 ? 0 : c3 + 1, c2 = c4 == 2 ? 0 : c4 + 1;     // (r + 4) % 3, (r + 2) % 3
            DN_PH(0);
            if (r + 4 < n_t && !(a.variant & 4)) stage_keys_run(c4);
            DN_PH(1);
            auto tile_block = [&](auto RG) {
                // One basic block, 14 slots = (k-step, key group): per slot the fragment pair of slot + 2 is requested, the slot's
                // three multiplies are issued (tile r + 3; past the last tile: on whatever the stage holds, never used) and one
                // fourteenth of the weights' VALU work is done (tile r + 2, from the scores formed one interval ago).  Fences keep
                // the order: left to the scheduler, every multiply sat behind its own fragment read and a full LDS round trip.
                constexpr int PF = 2, NSL = 2 * DN_KS;
                const unsigned kb_addr = lds_sm + (unsigned)(c3 * DN_KTILE_B) + ka_off;
                dnh8 fh[PF + 1], fl[PF + 1];
                auto kfrag = [&](int sl) {
                    const unsigned ad = kb_addr + (unsigned)((sl & 1) * 16 * (DN_KPITCH * 16) + 64 * (sl >> 1));
                    fh[sl % (PF + 1)] = __builtin_bit_cast(dnh8, dn_lds128(ad));
                    fl[sl % (PF + 1)] = __builtin_bit_cast(dnh8, dn_lds128(ad + DN_KPART_B));
                };
                SAcc s_new;
#pragma unroll
                for (int kg = 0; kg < 2; ++kg) { s_new.hh[kg] = f32x4{0.f, 0.f, 0.f, 0.f}; s_new.hl[kg] = s_new.hh[kg]; s_new.lh[kg] = s_new.hh[kg]; }
                dnh4 hq[2], lq[2];
                dnf2 zpv[2] = {{0.f, 0.f}, {0.f, 0.f}};
                dnf2 ps_t = {0.f, 0.f};
                int dt[2] = {0, 0};
                const int jy0 = yw, jx0 = xw, pbuf = c2;
#pragma unroll
                for (int sl = 0; sl < PF; ++sl) kfrag(sl);
#pragma unroll
                for (int sl = 0; sl < NSL; ++sl) {
                    if (sl + PF < NSL) kfrag(sl + PF);
                    __builtin_amdgcn_sched_barrier(0);
                    {
                        const int kg = sl & 1, ks = sl >> 1;
                        const dnh8 k_hi = fh[sl % (PF + 1)], k_lo = fl[sl % (PF + 1)];
                        s_new.hl[kg] = __builtin_amdgcn_mfma_f32_16x16x32_f16(k_hi, qf_lo[ks], s_new.hl[kg], 0, 0, 0);
                        s_new.hh[kg] = __builtin_amdgcn_mfma_f32_16x16x32_f16(k_hi, qf_hi[ks], s_new.hh[kg], 0, 0, 0);
                        s_new.lh[kg] = __builtin_amdgcn_mfma_f32_16x16x32_f16(k_lo, qf_hi[ks], s_new.lh[kg], 0, 0, 0);
                    }
#ifdef DAGL_ABLATION
                    if ((a.variant & 16) && sl < 8) {          // (ablation: no weights' arithmetic -- what its VALU operations cost the SIMD)
                        const int kg = sl >> 2, rr = sl & 3;
                        hq[kg][rr] = (_Float16)(0.001f * (float)(sl + 1)); lq[kg][rr] = (_Float16)1e-5f;
                        zpv[kg] = dnf2{1.f, 1.f}; dt[kg] = 1; s_top = 1e30f;
                    } else
#endif
                    if (sl < 8) {                              // slots 2 j, 2 j + 1: one pair of keys (logits + exponentials | sums + split)
                        const int kg = sl >> 2, pr = (sl >> 1) & 1;
                        if (!(sl & 1)) weight_pair_a(RG, kg, pr, jy0, jx0, zpv[kg], dt[kg], ps_t);
                        else weight_pair_b(pr, ps_t, zpv[kg], hq[kg], lq[kg]);
                    } else if (sl < 10) {
                        const int kg = sl - 8;
#ifdef DAGL_ABLATION
                        if (a.variant & 8) {                   // (ablation: all of the arithmetic, but constant weights handed over -- the multiplies' data, not the VALU work)
#pragma unroll
                            for (int rr = 0; rr < 4; ++rr) {
                                zpv[kg][0] += (float)hq[kg][rr] + (float)lq[kg][rr];
                                hq[kg][rr] = (_Float16)(0.001f * (float)(4 * kg + rr + 1)); lq[kg][rr] = (_Float16)1e-5f;
                            }
                            s_top = 1e30f;
                        }
#endif
                        weight_store(kg, pbuf, zpv[kg], dt[kg], hq[kg], lq[kg]);
                    }
#ifdef DAGL_DN_PSLEEP
                    if (sl >= DAGL_DN_PSLEEP_FROM) __builtin_amdgcn_s_sleep(DAGL_DN_PSLEEP);
#endif
                    __builtin_amdgcn_sched_barrier(0);
                }
                final_scores(s_new);
            };
            if (r + 2 < n_t) {
                if (!tile_ragged(yw, xw)) {
                    tile_block(std::false_type{});
                } else {                                   // (rare: a tile with positions outside the map -- the two phases one after the other)
                    weights(yw, xw, c2);
                    SAcc s_new;
                    scores(c3, s_new);
                    final_scores(s_new);
                }
            }
            next_tile(yw, xw);
            next_tile(yk, xk);                             // the next interval's key request
            if (xk != 0) {                                 // (wave-uniform)
#pragma unroll
                for (int j = 0; j < kpn; ++j) k_run[j] += (unsigned)(DN_TW * DSH * 2);
            } else {
                key_run_set(yk, xk);
            }
            c3 = c4;
            DN_PH(6);
            dma_wait_all();
            DN_PH(7);
            __syncthreads();
            DN_PH(5);
        }
        // the lanes' shares of their queries' sums (read back below, behind the block's last barrier)
        {
            constexpr double inv_ps = 1.0 / (double)DN_PS;
            const float e0 = __builtin_amdgcn_exp2f(fminf(0.f - m_run, 0.f) * DN_LOG2E + 15.0f);     // 2^15 e^(0 - m_run): a masked key's weight
            const double z_run = zp_run + (double)(8 * n_t - n_inv - deg) * (double)e0;
            bool ps_;
            const float lt = dn_logit(s_top, mtq, bsq, ps_);
            szz[wave][lane][0] = z_run * inv_ps; szz[wave][lane][1] = zp_run * inv_ps; sdg[wave][lane] = deg;
            slt[wave][lane] = ps_ ? fmaxf(lt, 0.f) : 0.f;
        }
    } else {
        // ======================================================== A V ========================================================
        const int lane = lane_now();
        const int cw = wave - 4;
        const int c16 = lane & 15, gk = lane >> 4;
        const int h = lane >> 5;
        const int ct0 = dn_ct_start(cw);
        // ---- the value regions' LDS-DMA: part = cw >> 2, the part's 6 pieces over its four waves as 1 / 1 / 2 / 2; per-lane byte
        // offsets carried like the producers' (formed anew at a tile-row wrap and at a ragged last tile, whose region is clamped to
        // the padded map)
        const int spart = cw >> 2, wl = cw & 3;
        const int vp0 = wl < 2 ? wl : 2 + 2 * (wl - 2), vpn = wl < 2 ? 1 : 2;
        const unsigned char* vsrc = reinterpret_cast<const unsigned char*>((spart ? a.v_lo : a.v_hi) + (size_t)b * g.Hp * g.Wp * CH) - 4096;
        unsigned v_run[2];
        auto value_run_set = [&](int jy0, int jx0) {
            const int ln = lane_now();
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int p = (vp0 + (j < vpn ? j : 0)) * 64 + ln;
                int row = p / (2 * DN_VPX); const int c36 = p - row * (2 * DN_VPX);
                if (row > DN_RH - 1) row = DN_RH - 1;                            // (slack positions of the last piece: any valid address)
                int px = c36 >> 1; if (px > DN_RW - 1) px = DN_RW - 1;
                int y = jy0 + row; if (y > g.Hp - 1) y = g.Hp - 1;               // stay inside the padded map
                int x = jx0 + px; if (x > g.Wp - 1) x = g.Wp - 1;
                v_run[j] = (unsigned)(y * g.Wp + x) * 32u + (unsigned)(16 * (c36 & 1) + 4096);
            }
        };
        auto stage_values_run = [&](int buf) {
            const unsigned dst = lds_sv + (unsigned)(buf * DN_VTILE_B + spart * DN_VPART_B + vp0 * 1024);
            // (one piece per statement here: with several offsets live at once the allocator spilled the accumulators)
            dn_glds<1>(vsrc, dst, v_run[0], 0, 0, 0);
            if (vpn > 1) dn_glds<1>(vsrc, dst + 1024u, v_run[1], 0, 0, 0);       // wave-uniform
        };
        int yv = ty0, xv = tx0;                            // the tile of the next value request
        for (int t = 0; t < 2; ++t) {                      // tiles 0, 1 -> stages 0, 1
            if (t < n_t) { value_run_set(yv, xv); stage_values_run(t); }
            next_tile(yv, xv);
        }
        value_run_set(yv, xv);                             // tile 2: requested in interval 0

        f32x16 acc[2][DN_CTMAX];
        f32x4 acc48;                                       // tap 48 (consumers 0-3): out^T[channel 4 gk + r][query 16 cw + c16]
        const unsigned pr_off = (unsigned)(lane * DN_PQ_ENTRY);
        // A V operand A: lane t = lane & 15 of a 16-lane group supplies the 8 bytes (pixel t >> 2, channels 4 (t & 3) ..) of the group's
        // [4 pixels][16 channels] block; the group = (tap parity, key half h)
        const bool second = (lane & 16) != 0;
        const unsigned va_lane = (unsigned)(((c16 >> 2) * 32) + (c16 & 3) * 8 + h * (DN_VPX * 32));
        // column tile ct = taps (2 ct, 2 ct + 1) x 16 channels; lanes 16-31 / 48-63 ("second") take the odd tap
        unsigned vt_off[DN_CTMAX];
#pragma unroll
        for (int t = 0; t < DN_CTMAX; ++t) {
            const int tap = 2 * (ct0 + t) + (second ? 1 : 0);
            const int kh = tap / KS, kw = tap - kh * KS;
            vt_off[t] = (unsigned)((kh * DN_VPX + kw) * 32) + va_lane;
        }
        // the weights of BOTH query tiles for a k-block (lane = query i, keys 16 kb + 8 h ..), one half (hi | lo) at a time.  A
        // granule (32 queries x 16 keys) whose weights are all exactly zero adds exactly nothing and is skipped.  (hi = 0 implies
        // lo = 0: the split of a number below half the smallest denormal; -0 cannot occur, p >= 0.)
        struct PHalf { dnu4 q[2]; };
        auto load_p = [&](int buf, int kb, int lo) {
            PHalf f;
            const unsigned pa = lds_pq + (unsigned)(buf * DN_PQ_B + 16 * kb + 32 * lo) + pr_off;
#pragma unroll
            for (int qq = 0; qq < 2; ++qq) f.q[qq] = dn_lds128(pa + (unsigned)(qq * 64 * DN_PQ_ENTRY));
            return f;
        };
        // The hi halves of the weights and the first value fragment of the NEXT k-block to multiply are fetched one k-block ahead --
        // across the tile's barrier too: the producers run two tiles ahead of the multiplies and the value regions are requested two
        // tiles ahead, so that what tile r + 1 needs was complete at the barrier BEFORE tile r.  (Rounds 4-5: one tile ahead; after
        // every barrier the multiplying waves first sat through an LDS round trip for their weights while the producers issued their
        // DMA: ~400 of a tile's ~4300 cycles with no multiply in flight on the SIMD.)  The lo halves (the zero test and a column tile's
        // first four multiplies need only the hi ones) are requested at the top of their k-block.
        PHalf ph;
        DnFrag fv;
        auto attend = [&](int buf, int nbuf) {
            if (a.variant & 1) return;
            const unsigned vbase = lds_sv + (unsigned)(buf * DN_VTILE_B);
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                const unsigned va_kb = vbase + (unsigned)(2 * kb * DN_VPX * 32);
                dnh8 p_hi[2], p_lo[2];
                bool nz[2];
                const PHalf pl = load_p(buf, kb, 1);
#pragma unroll
                for (int qq = 0; qq < 2; ++qq) {
                    p_hi[qq] = __builtin_bit_cast(dnh8, ph.q[qq]); p_lo[qq] = __builtin_bit_cast(dnh8, pl.q[qq]);
                    nz[qq] = (a.variant & 32) || __builtin_amdgcn_ballot_w64(((ph.q[qq].x | ph.q[qq].y) | (ph.q[qq].z | ph.q[qq].w)) != 0u) != 0ull;
                }
                const unsigned va_next = (kb == 0 ? va_kb + (unsigned)(2 * DN_VPX * 32) : lds_sv + (unsigned)(nbuf * DN_VTILE_B)) + vt_off[0];
                ph = kb == 0 ? load_p(buf, 1, 0) : load_p(nbuf, 0, 0);   // arrive under this k-block's multiplies
                DN_PH(3);
                if (nz[0] || nz[1])                        // (wave-uniform)
                    fv = dn_pv(acc, va_kb, vt_off, p_hi, p_lo, fv, va_next);
                else
                    fv = dn_vfrag(va_next);
                DN_PH(4);
            }
            if (cw < 4) {
                // tap 48 = patch position (6, 6): 16 channels x the 16 queries 16 cw .. of the block, K = the tile's 32 keys.  Lane
                // (c16, gk): operand B = the weights of query c16 for keys 8 gk .. (k-block gk >> 1, key half gk & 1 of the exchange
                // layout), operand A = channel c16 of the region pixels (row 6 + gk, columns 6 .. 13) through the transposing read
                const unsigned pe = lds_pq + (unsigned)(buf * DN_PQ_B + (((cw >> 1) * 64 + (gk & 1) * 32 + 16 * (cw & 1) + c16) * DN_PQ_ENTRY) + 16 * (gk >> 1));
                const dnu4 wh = dn_lds128(pe), wl_ = dn_lds128(pe + 32);
                if ((a.variant & 32) || __builtin_amdgcn_ballot_w64(((wh.x | wh.y) | (wh.z | wh.w)) != 0u) != 0ull) {
                    const DnFrag f = dn_vfrag(vbase + (unsigned)(((6 + gk) * DN_VPX + 6 + (c16 >> 2)) * 32 + (c16 & 3) * 8));
                    const dns8 vh = {f.h0[0], f.h0[1], f.h0[2], f.h0[3], f.h1[0], f.h1[1], f.h1[2], f.h1[3]};
                    const dns8 vl = {f.l0[0], f.l0[1], f.l0[2], f.l0[3], f.l1[0], f.l1[1], f.l1[2], f.l1[3]};
                    const dnh8 v_hi = __builtin_bit_cast(dnh8, vh), v_lo = __builtin_bit_cast(dnh8, vl);
                    const dnh8 q_hi = __builtin_bit_cast(dnh8, wh), q_lo = __builtin_bit_cast(dnh8, wl_);
                    acc48 = __builtin_amdgcn_mfma_f32_16x16x32_f16(v_hi, q_lo, acc48, 0, 0, 0);
                    acc48 = __builtin_amdgcn_mfma_f32_16x16x32_f16(v_lo, q_hi, acc48, 0, 0, 0);
                    acc48 = __builtin_amdgcn_mfma_f32_16x16x32_f16(v_hi, q_hi, acc48, 0, 0, 0);
                }
            }
        };

        dma_wait_all();
        __syncthreads();                                   // barrier 1
#pragma unroll
        for (int qq = 0; qq < 2; ++qq)
#pragma unroll
            for (int t = 0; t < DN_CTMAX; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[qq][t][r] = 0.f;
        acc48 = f32x4{0.f, 0.f, 0.f, 0.f};
        __syncthreads();                                   // barrier 2: the weights of tile 0 are written
        ph = load_p(0, 0, 0);
        fv = dn_vfrag(lds_sv + vt_off[0]);
        __syncthreads();                                   // (the producers' interval -1)
        int c0 = 0;                                        // r % 3
        for (int r = 0; r < n_t; ++r) {
            const int c1 = c0 == 2 ? 0 : c0 + 1, c2 = c1 == 2 ? 0 : c1 + 1;
            DN_PH(0);
            if (r + 2 < n_t && !(a.variant & 4)) stage_values_run(c2);
            attend(c0, c1);
            next_tile(yv, xv);
            if (xv != 0 && xv + DN_TW <= g.W) {            // (wave-uniform) same tile row, region inside the padded map: 8 pixels on
                v_run[0] += (unsigned)(DN_TW * 32); v_run[1] += (unsigned)(DN_TW * 32);
            } else {
                value_run_set(yv, xv);
            }
            c0 = c1;
            DN_PH(6);
            dma_wait_all();
            DN_PH(7);
            __syncthreads();
            DN_PH(5);
        }
        // ---- this wave's column tiles of both query tiles ----
        {
            const int ln = lane_now();
            const int i_ = ln & 31, h_ = ln >> 5, c16_ = ln & 15, gk_ = ln >> 4;
#pragma unroll
            for (int qq = 0; qq < 2; ++qq) {
                const int q2 = qb * 64 + qq * 32 + i_;
                if (q2 < g.L) dn_store(acc[qq], a.part_acc + (((size_t)split * a.B + b) * g.L + q2) * P, h_, ct0);
            }
            const int qs_ = qb * 64 + 16 * cw + c16_;
            if (cw < 4 && qs_ < g.L) {                     // tap 48: columns 768 + 4 gk .. of query qb 64 + 16 cw + c16
                constexpr float inv = 1.0f / (DN_PS * DN_VS);
                *reinterpret_cast<float4*>(a.part_acc + (((size_t)split * a.B + b) * g.L + qs_) * P + 768 + 4 * gk_) =
                    make_float4(acc48[0] * inv, acc48[1] * inv, acc48[2] * inv, acc48[3] * inv);
            }
        }
    }
#ifdef DAGL_ABLATION
    if (clocks && (threadIdx.x & 63u) == 0) {
        unsigned* po = a.phase_out + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 12 + wave) * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) po[e] = phc[e];
    }
#endif

    // ---- partial results of this key range ------------------------------------------------------------------------------------------
    __syncthreads();
    if (tid < 64) {
        // query tid of the block: its scores lived in producer wave tid / 16, lanes c16 + 16 gk; summed in a fixed order
        const int q = qb * 64 + tid;
        if (q < g.L) {
            const int wq_ = tid >> 4, cq = tid & 15;
            double z = 0.0, zp = 0.0; int d = 0;
            float lt = 0.f;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) { z += szz[wq_][cq + 16 * g4][0]; zp += szz[wq_][cq + 16 * g4][1]; d += sdg[wq_][cq + 16 * g4]; lt = fmaxf(lt, slt[wq_][cq + 16 * g4]); }
            const size_t orow = ((size_t)split * a.B + b) * g.L + q;
            const size_t ql = (size_t)b * g.L + q;
            a.part_m[orow] = dn_shift(a, ql);
            a.part_z[3 * orow] = z; a.part_z[3 * orow + 1] = zp; a.part_z[3 * orow + 2] = (double)lt; a.part_deg[orow] = d;
        }
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
    double z = has ? a.part_z[3 * (lane * nq + ql)] * (double)wl : 0.0;
    double zp = has ? a.part_z[3 * (lane * nq + ql) + 1] * (double)wl : 0.0;
    float ltop = has ? (float)a.part_z[3 * (lane * nq + ql) + 2] : 0.f;
    int deg = has ? a.part_deg[lane * nq + ql] : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { z += __shfl_xor(z, o); zp += __shfl_xor(zp, o); deg += __shfl_xor(deg, o); ltop = fmaxf(ltop, __shfl_xor(ltop, o)); }
    // The weights went through the matrix cores as 2^15 e^(l - M') = hi + lo in fp16.  With the exact row maximum as the shift (round 5)
    // the largest weight of a row is 1 to rounding.  A shift far ABOVE the maximum pushes the row's weights towards the fp16 denormals
    // (2^15 e^-16 = 2^-8 still has 16 significant bits in hi + lo, e^-20 has 10, below e^-27 every weight is exactly zero and the row
    // would come out as zeros; rounds 3-4 used an upper bound from a bf16 scan, ~3 % above a logit in the thousands: 6e-3 off at logits
    // of 1 700, rows of zeros at 3 300, tools/dense_large_logits.py) -- that only happens to rows with large logits whose candidates
    // rowmax_exact_kernel could not hold (flat maps).  The guard of round 4 stays as their path: the first pass records every row's largest logit (the producers form
    // it anyway) and flags the blocks of 64 queries in which some row's slack exceeds DN_SHIFT_SLACK; a second, gated launch runs
    // exactly those blocks again with the recorded maxima as shifts -- two launches that exit at once when nothing is flagged.
    const int n_qblocks = (a.g.L + 63) / 64;
    const int bq = (int)(ql / a.g.L), qin = (int)(ql - (size_t)bq * a.g.L);
    if (a.pass == 1 && a.redo_blk[bq * n_qblocks + qin / 64] == 0) return;      // (second combine: the rows of re-run blocks only)
    if (a.pass == 0 && lane == 0) {
        a.m_exact[ql] = ltop;                                                      // (>= 0: masked keys have l = 0, and so has an empty row)
        // (an empty row too: its masked keys' e^(0 - M) must not vanish either)
        // ... and a shift BELOW the kernel's own largest logit (the row kernel's score further from the split-fp16 one than its 6e-6
        // inflation at logits of 1e4-1e5, or the true arg-max missing from its candidates): the producers clamp l - M' at 0, the top
        // weights would silently read 1 instead of e^(ltop - M) -- the same second pass, whose shifts are the recorded maxima, repairs it
        if ((M - ltop > DN_SHIFT_SLACK || ltop - M > 2e-5f) && atomicExch(&a.redo_blk[bq * n_qblocks + qin / 64], 1) == 0) atomicAdd(a.redo_count, 1);
    }
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
// (scale_word: the bits of the tensor's largest magnitude -- launch_absmax -- instead of the fixed DN_FS: s = fcg_scale_of(*word) brings it
// into [2^13, 2^14), the consumer divides by s; top-k beyond 64, round 6: no fixed range of the features there)
__global__ void feat_split_kernel(int rows, int rows_in, int rows_out, const float* __restrict__ src,
                                  unsigned short* __restrict__ hi, unsigned short* __restrict__ lo, RangeTag range,
                                  const unsigned* __restrict__ scale_word) {
    const int b = blockIdx.y;
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)rows_out * (DSH / 8)) return;
    const int r = (int)(t / (DSH / 8)), c8 = (int)(t % (DSH / 8));
    unsigned short vh[8], vl[8];
    const float fs = scale_word != nullptr ? fcg_scale_of(*scale_word) : DN_FS;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int c = 8 * c8 + u;
        const float v = (r < rows && c < D) ? src[((size_t)b * rows_in + r) * DS + c] * fs : 0.f;
        if (range.word != nullptr && !(fabsf(v) < RANGE_LIMIT)) *range.word = range.tag;
        const _Float16 h = (_Float16)v;
        vh[u] = __builtin_bit_cast(unsigned short, h);
        vl[u] = __builtin_bit_cast(unsigned short, (_Float16)(v - (float)h));
    }
    const size_t o = (((size_t)b * rows_out + r) * DSH + 8 * c8) / 8;
    reinterpret_cast<uint4*>(hi)[o] = *reinterpret_cast<const uint4*>(vh);
    reinterpret_cast<uint4*>(lo)[o] = *reinterpret_cast<const uint4*>(vl);
}

int launch_feat_split(hipStream_t s, int B, int rows, int rows_in, int rows_out, const float* src, uint16_t* hi, uint16_t* lo, RangeTag range,
                      const unsigned* scale_word) {
    const size_t n8 = (size_t)rows_out * (DSH / 8);
    hipLaunchKernelGGL(feat_split_kernel, dim3((unsigned)((n8 + 255) / 256), B), dim3(256), 0, s, rows, rows_in, rows_out, src, hi, lo, range, scale_word);
    DAGL_LAUNCH_CHECK("feat_split_kernel");
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
// split value maps: whole maps + one staged row of slack (the last piece of a region may start past the map's last pixel)
static size_t dn_map16_bytes(int B, const Grid& g) { return align_up(((size_t)B * g.Hp * g.Wp * CH + 1024) * sizeof(uint16_t), 256); }

size_t dense_workspace_bytes(int B, const Grid& g) {
    const size_t rows = (size_t)dense_splits(B, g) * B * g.L;
    return align_up(rows * P * sizeof(float), 256) + align_up(rows * sizeof(float), 256) +
           align_up(rows * 3 * sizeof(double), 256) + align_up(rows * sizeof(int32_t), 256) +
           2 * dn_feat16_bytes(B, g.N) + 2 * dn_feat16_bytes(B, g.L) + 2 * dn_map16_bytes(B, g) +
           align_up((size_t)B * g.L * sizeof(float), 256) + align_up((size_t)B * ((g.L + 63) / 64) * sizeof(int32_t), 256);
}

// carve of the dense workspace (shared by launch_dense_attend and dense_split_buffers)
struct DnCarve { float* part_acc; float* part_m; double* part_z; int32_t* part_deg; uint16_t *xh, *xl, *qh, *ql, *vh, *vl; float* m_exact; int32_t* redo_blk; };
static DnCarve dn_carve(void* ws, int B, const Grid& g) {
    const size_t rows = (size_t)dense_splits(B, g) * B * g.L;                // carve with the planned (upper) split count
    DnCarve c;
    char* p = static_cast<char*>(ws);
    c.part_acc = reinterpret_cast<float*>(p); p += align_up(rows * P * sizeof(float), 256);
    c.part_m = reinterpret_cast<float*>(p); p += align_up(rows * sizeof(float), 256);
    c.part_z = reinterpret_cast<double*>(p); p += align_up(rows * 3 * sizeof(double), 256);
    c.part_deg = reinterpret_cast<int32_t*>(p); p += align_up(rows * sizeof(int32_t), 256);
    c.xh = reinterpret_cast<uint16_t*>(p); p += dn_feat16_bytes(B, g.N);
    c.xl = reinterpret_cast<uint16_t*>(p); p += dn_feat16_bytes(B, g.N);
    c.qh = reinterpret_cast<uint16_t*>(p); p += dn_feat16_bytes(B, g.L);
    c.ql = reinterpret_cast<uint16_t*>(p); p += dn_feat16_bytes(B, g.L);
    c.vh = reinterpret_cast<uint16_t*>(p); p += dn_map16_bytes(B, g);
    c.vl = reinterpret_cast<uint16_t*>(p); p += dn_map16_bytes(B, g);
    c.m_exact = reinterpret_cast<float*>(p); p += align_up((size_t)B * g.L * sizeof(float), 256);
    c.redo_blk = reinterpret_cast<int32_t*>(p);
    return c;
}

Split16Out dense_split_buffers(void* dense_ws, int B, const Grid& g) {
    const DnCarve c = dn_carve(dense_ws, B, g);
    Split16Out so;
    so.hi[0] = c.xh; so.lo[0] = c.xl; so.hi[1] = c.qh; so.lo[1] = c.ql;
    so.rows_alloc[0] = feat_rows_h(g.N); so.rows_alloc[1] = feat_rows_h(g.L);
    return so;
}

// rows past the last key / query of the split features: the kernel's LDS-DMA and its query fragments run into them
void dense_guard_rows(ZeroList& zl, int B, const Grid& g, const Split16Out& so) {
    const int rn[2] = {g.N, g.L};
    for (int w = 0; w < 2; ++w) {
        const size_t row_b = (size_t)DSH * sizeof(uint16_t);
        const size_t img = (size_t)so.rows_alloc[w] * row_b, tail = (size_t)(so.rows_alloc[w] - rn[w]) * row_b;
        zl.add(reinterpret_cast<char*>(so.hi[w]) + (size_t)rn[w] * row_b, tail, B, img);
        zl.add(reinterpret_cast<char*>(so.lo[w]) + (size_t)rn[w] * row_b, tail, B, img);
    }
}

int launch_dense_attend(hipStream_t s, int B, const Grid& g, const float* wq, const float* x, const float* mt,
                        const float* bs, const float* smax, const float* b2p, void* ws, float* agg, int32_t* deg_out,
                        float* rowsum_out, int64_t* stats, RangeTag range, float* lse_out, bool features_split, bool want_stats) {
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
    const DnCarve c = dn_carve(ws, B, g);
    a.part_acc = c.part_acc; a.part_m = c.part_m; a.part_z = c.part_z; a.part_deg = c.part_deg;
    a.m_exact = c.m_exact; a.redo_blk = c.redo_blk; a.pass = 0;
    a.redo_count = reinterpret_cast<int32_t*>(stats + DENSE_RERUN_STAT);
    uint16_t *xh = c.xh, *xl = c.xl, *qh = c.qh, *ql = c.ql, *vh = c.vh, *vl = c.vl;
    a.v_hi = vh; a.v_lo = vl;
    a.x_hi = xh; a.x_lo = xl; a.wq_hi = qh; a.wq_lo = ql;
    a.rows_xh = feat_rows_h(g.N); a.rows_qh = feat_rows_h(g.L);
    {
        const size_t nx = (size_t)a.rows_xh * (DSH / 8), nq8 = (size_t)a.rows_qh * (DSH / 8);
        if (!features_split) {
        hipLaunchKernelGGL(feat_split_kernel, dim3((unsigned)((nx + 255) / 256), B), dim3(256), 0, s, g.N, a.rows_x, a.rows_xh, x, xh, xl, range, (const unsigned*)nullptr);
        DAGL_LAUNCH_CHECK("feat_split_kernel");
        hipLaunchKernelGGL(feat_split_kernel, dim3((unsigned)((nq8 + 255) / 256), B), dim3(256), 0, s, g.L, a.rows_q, a.rows_qh, wq, qh, ql, range, (const unsigned*)nullptr);
        DAGL_LAUNCH_CHECK("feat_split_kernel");
        }
        int rc = launch_split_map(s, (size_t)B * g.Hp * g.Wp * CH, b2p, vh, vl, range);     // 16 v = hi + lo, borders stay zero
        if (rc) return rc;
    }
    const int n_qblocks = (g.L + 63) / 64;
    a.phase_out = nullptr;
#ifdef DAGL_ABLATION
    const size_t n_blocks = (size_t)n_qblocks * a.splits * B;
    if ((a.variant & 64) && getenv("DAGL_TIMES_FILE")) a.phase_out = reinterpret_cast<unsigned*>(dbg_times_buffer(n_blocks * 12));
#endif
    hipLaunchKernelGGL(dense_attend_kernel, dim3(n_qblocks * a.splits, B), dim3(DN_THREADS), 0, s, a);
    DAGL_LAUNCH_CHECK("dense_attend_kernel");
#ifdef DAGL_ABLATION
    if (a.phase_out) {                                    // mean clocks per phase and wave group, appended to DAGL_TIMES_FILE
        static int budget = 3, skip = 8;
        if (skip > 0) --skip;
        else if (budget-- > 0 && hipStreamSynchronize(s) == hipSuccess) {
            unsigned* h = static_cast<unsigned*>(malloc(n_blocks * 96 * sizeof(unsigned)));
            if (h && hipMemcpy(h, a.phase_out, n_blocks * 96 * sizeof(unsigned), hipMemcpyDeviceToHost) == hipSuccess) {
                if (FILE* f = fopen(getenv("DAGL_TIMES_FILE"), "a")) {
                    // producers (waves 0-3): 1 = key DMA issue, 6 = scores + weights; multiplying waves: 3 = value DMA issue + weight fragments,
                    // 4 = A V, 6 = tap 48 + tile coordinates; all: 7 = wait for the wave's own DMA pieces, 5 = barrier
                    static const char* nm[8] = {"top", "key-dma", "-", "v-dma+p-load", "AV", "barrier", "S+w | tap48", "dma-wait"};
                    for (int grp = 0; grp < 3; ++grp) {               // waves 0-3: producers, 4-11: consumers
                        double sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tot = 0;
                        for (size_t bl = 0; bl < n_blocks; ++bl)
                            for (int w = 4 * grp; w < 4 * grp + 4; ++w)
                                for (int e = 0; e < 8; ++e) sum[e] += h[(bl * 12 + w) * 8 + e];
                        for (int e = 0; e < 8; ++e) { sum[e] /= (double)(n_blocks * 4); tot += sum[e]; }
                        fprintf(f, "dense_attend phases, waves %d-%d (clocks per wave, %d tiles per block): total %.0f |", 4 * grp, 4 * grp + 3, a.tiles_per_split, tot);
                        for (int e = 0; e < 8; ++e) fprintf(f, " %s %.0f (%.1f %%)", nm[e], sum[e], 100.0 * sum[e] / tot);
                        fprintf(f, "\n");
                    }
                    fclose(f);
                }
            }
            free(h);
        }
    }
#endif
    hipLaunchKernelGGL(dense_combine_kernel, dim3((unsigned)(((size_t)B * g.L + 3) / 4)), dim3(256), 0, s, a, agg, deg_out, rowsum_out, lse_out);
    DAGL_LAUNCH_CHECK("dense_combine_kernel");
    // second pass: the blocks whose rows' weights came too close to the fp16 denormals, shifted by their exact row maxima (both
    // launches exit at once when the first combine flagged nothing)
    a.pass = 1; a.phase_out = nullptr;
    hipLaunchKernelGGL(dense_attend_kernel, dim3(n_qblocks * a.splits, B), dim3(DN_THREADS), 0, s, a);
    DAGL_LAUNCH_CHECK("dense_attend_kernel");
    hipLaunchKernelGGL(dense_combine_kernel, dim3((unsigned)(((size_t)B * g.L + 3) / 4)), dim3(256), 0, s, a, agg, deg_out, rowsum_out, lse_out);
    DAGL_LAUNCH_CHECK("dense_combine_kernel");
    // total edges, max degree, queries beyond the neighbour lists' width (no per-query atomics) -- only for a call that reads them back
    return want_stats ? launch_degree_stats(s, (size_t)B * g.L, a.part_deg, stats, DAGL_LIST_CAP) : DAGL_OK;
}

}  // namespace dagl
