// bf16 screening of the similarity search + exact refinement of the survivors.
//
// The score matrix S = Wq . X^T (DN_Gray/model/dagl.py:250) is only ever *used* at the few keys a query
// keeps (dagl.py:256-261); everything else is compared against a threshold and dropped.  So the O(L*N*D)
// search runs on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16, 16x the fp32 MFMA rate) and only decides
// which keys MIGHT be kept; the kept scores themselves are then recomputed from the fp32 features with
// fp64 accumulation (more accurate than an fp32 GEMM) by the refine kernels.  The screen is conservative, not
// approximate:
//   all features are >= 0 (post-ReLU), so with q~ = bf16(q), x~ = bf16(x) (round to nearest even: 8 significant bits,
//   relative error <= 2^-8 each, attained just above a power of two) every product and therefore the whole sum satisfies
//        S~ = sum q~ x~  in  [S (1-d), S (1+d)],   d = (1 + 2^-8)^2 - 1 + accumulation slack = 0.00786  <  DELTA = 0.0079
//   (tests/test_screen_band.py, tests/test_gpu_adversarial.py: rows built so that a true neighbour rounds down on every
//   feature while its competitors round up)
//   - adaptive mask:  a key with relu(S - mean*thr + bias) != 0 has S~ (1+DELTA) - mean*thr + bias > 0;
//   - top-k: split the keys into >= k disjoint groups; the k-th largest group maximum of S~, theta, is
//     attained by k distinct keys whose true S >= theta/(1+DELTA), hence tau_k(S) >= theta/(1+DELTA) and
//     every true top-k key has S~ >= theta (1-DELTA)/(1+DELTA).
// Keys passing these tests are the candidates; a (query, chunk, half) segment that receives more than its
// slot count raises a flag and that query group is redone by the exact fp32 kernel (select.hip).
#include "dagl_common.h"

namespace dagl {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

constexpr int SK = 64;                           // keys per streaming step (two 32-row MFMA tiles)
constexpr int STEP_ELEMS = SK * DSH;             // 13824 bf16 = 27648 B = 27 DMA pieces of 1 KiB exactly
constexpr int STEP_PIECES = 27;
constexpr int KB = 13;                           // MFMAs (K=16 each) per 32x32 tile: 208 = 196 + 12 zeros
constexpr float DELTA = SCREEN_DELTA;
constexpr int GKEEP = 4;                         // group maxima kept per (query, chunk, half) segment

// top-1 screen: candidate threshold from the largest SAMPLED screened score (one key reaches it, so the row's best key has
// S~ >= this; 0 = nothing sampled: pass everything) -- what screen_theta_kernel computes for k = 1
__device__ __forceinline__ float theta_of_sampled_max(float m) { return m > 0.f ? m * ((1.0f - SCREEN_DELTA) / (1.0f + SCREEN_DELTA)) : 0.f; }

// a full segment's further candidates go to the query's shared spill area (ScreenArgs::spill); past its end they are dropped and the
// count tells refine so (-> the exact redo pass, as before)
__device__ __forceinline__ void screen_spill(const ScreenArgs& a, size_t qlin, int key, float sv) {
    const unsigned pos = atomicAdd(a.spill_cnt + qlin, 1u);
    if (pos < (unsigned)SCREEN_SPILL) a.spill[qlin * SCREEN_SPILL + pos] = make_int2(key, __float_as_int(sv));
}

// A query's 13 fragments of 8 bf16: row-major copy (features 16t + 8h .. of its row: 32 rows x 32 B per wave load) or the fragment
// order project16 writes (ScreenArgs::q_tiled: [queries 0..15 | 16..31][t][h][query][8] per 32 queries, two runs of 512 B per
// wave load).  q0 = the wave tile's first query; tiles past L read the last valid one (their thresholds are +inf).
__device__ __forceinline__ void load_query_fragments(const ScreenArgs& a, int b, int q0, int qc, int lane, bf16x8 (&qf)[KB]) {
    const int h = lane >> 5;
    int t0 = q0; if (t0 > ((a.L - 1) & ~31)) t0 = (a.L - 1) & ~31;
    const int i = lane & 31;
    const unsigned short* qp = a.q_tiled ? a.wqh + ((size_t)b * a.rows_qh + t0) * DSH + (i >> 4) * 3328 + h * 128 + (i & 15) * 8
                                         : a.wqh + ((size_t)b * a.rows_qh + qc) * DSH + 8 * h;
    const int qs = a.q_tiled ? 256 : 16;
#pragma unroll
    for (int t = 0; t < KB; ++t) qf[t] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const s16x8*>(qp + qs * t));
}

// PASS 0: group maxima over every `sample`-th step (top-k threshold estimation)
// PASS 1: candidate filter over every step: key is a candidate of query q iff S~ >= theta[q]
//         (theta carries the whole conservative test of either mode; +inf for padding queries).
// Keys past N are zero rows (S~ = 0): they can only pass a degenerate theta <= 0 and are dropped by refine.
//
// A block covers SCR_QUERIES queries (512: 16 waves, one block per CU, four key tiles in flight; 256: 8 waves, two blocks
// per CU, two tiles), a wave QW query tiles of 32 (QW = 1 in the release build; QW = 2 -- half the LDS reads per MFMA at
// half the occupancy -- measures the same and is kept for the ablation build).  Every key fragment is one ds_read_b128 per
// lane; LDS (256 B/clk for b128 reads, conflict-free by the odd row stride) is ~25 % busy, the matrix pipes 50-54 %:
// DESIGN.md section 7 lists what was tried to close that gap.
// VAR (ablation builds only, results wrong by construction): 1 no tile DMA, 2 no MFMA, 4 no test, 8 theta = inf, 16 key
// fragments from registers, 32 no barrier, 64 accumulators carried across steps.
// SMAX (filter pass only): the segment's header also carries the largest screened score among its candidates (ScreenArgs::seg_max:
// the dense formulation's row maxima, rowmax_exact_kernel) -- a template parameter so that the top-k / adaptive filters keep their
// register budget untouched
template <int PASS, int QW, int VAR, int SCR_QUERIES = 256, bool SMAX = false>
__global__ __launch_bounds__(SCR_QUERIES / QW * 2, (SCR_QUERIES == 256) ? QW : 1) void screen_kernel(ScreenArgs a, int n_qgroups) {
    if (a.gate != nullptr && *a.gate == 0) return;                 // (a re-run launch of a cold workspace that is not needed)
    if (a.policy != nullptr && *a.policy != 0) { a.capseg = a.capseg_tight; a.sample = a.sample_tight; }   // device-side threshold policy
    constexpr int WAVES = SCR_QUERIES / 32 / QW;
    // key tiles in flight: a tile is requested NBUF-1 steps before it is multiplied.  LDS-DMA lands a tile ~2 us after its
    // request under load, a step's matrix work is 1.45 us: with two buffers (request one step ahead) every step ends waiting
    // for the next tile.  512-query blocks are alone on their CU and can afford four buffers (108 KiB).
    constexpr int NBUF = (SCR_QUERIES >= 384) ? 4 : 2;
    constexpr int PPW = STEP_PIECES / WAVES;                                  // DMA pieces per wave and step (+1 for the first waves)
    __shared__ __attribute__((aligned(16))) unsigned short sK[NBUF][STEP_ELEMS];     // 54 / 108 KiB

    const int tid = threadIdx.x;
    dbg_stamp(a.times, blockIdx.y * gridDim.x + blockIdx.x, 0);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int i = lane & 31, h = lane >> 5;
    const int b = blockIdx.y;

    const int logical = xcd_remap2(blockIdx.x, gridDim.x);
    const int split = logical / n_qgroups;
    const int qg = logical % n_qgroups;
    const int step0 = split * a.steps_per_split;
    int step1 = step0 + a.steps_per_split;
    if (step1 > a.n_steps) step1 = a.n_steps;
    const int stride = (PASS == 0) ? a.sample : 1;

    // query fragments: QW x 13 x 8 bf16: Wq~[q][16t + 8h .. +7]
    bf16x8 qf[QW][KB];
    bool qvalid[QW];
    size_t qlin[QW];
    float thq[QW], thlo[QW];          // thlo = predecessor of thq:  S~ >= thq  <=>  S~ > thlo  <=>  sign(thlo - S~) set
    int n_loc[QW];
    int2* cseg[QW];
    size_t seg[QW];
#pragma unroll
    for (int w = 0; w < QW; ++w) {
        const int q = ((qg * WAVES + wave) * QW + w) * QT + i;
        qvalid[w] = q < a.L;
        const int qc = qvalid[w] ? q : a.L - 1;
        qlin[w] = (size_t)b * a.L + qc;
        load_query_fragments(a, b, q - i, qc, lane, qf[w]);
        thq[w] = 0.f;
        if (PASS == 1) thq[w] = (qvalid[w] && !(VAR & 8)) ? a.theta[qlin[w]] : __builtin_inff();
        if (PASS == 1 && SMAX && qvalid[w]) thq[w] = theta_of_sampled_max(thq[w]);      // (a.theta holds the raw sampled maximum)
        {
            const unsigned tb = __float_as_uint(thq[w]);
            thlo[w] = __uint_as_float(thq[w] > 0.f ? tb - 1u : (thq[w] == 0.f ? 0x80000001u : tb + 1u));
        }
        n_loc[w] = 0;
        seg[w] = (qlin[w] * a.splits + split) * 2 + h;
        cseg[w] = a.cand + (qlin[w] * a.capseg + 1) * (size_t)(a.splits * 2) + split * 2 + h;   // slot 1 (slot 0 = header); slots are splits * 2 apart
    }
#pragma unroll
    for (int w = 0; w < QW; ++w)
#pragma unroll
        for (int t = 0; t < KB; ++t) asm volatile("" : "+v"(qf[w][t]));

    float gm[QW][16];
#pragma unroll
    for (int w = 0; w < QW; ++w)
#pragma unroll
        for (int r = 0; r < 16; ++r) gm[w][r] = -1.0f;
    float smx[QW];                                    // SMAX: largest screened score among this lane's (= this segment's) candidates
#pragma unroll
    for (int w = 0; w < QW; ++w) smx[w] = -1.0f;

    const unsigned short* xb = a.xh + (size_t)b * a.rows_xh * DSH;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr_of(&sK[0][0]));
    const int n_it = (step1 > step0) ? (step1 - step0 + stride - 1) / stride : 0;
    const bool more = wave < (STEP_PIECES % WAVES);                          // this wave carries PPW + 1 pieces per step
    auto request = [&](int it) {                                             // key tile of iteration it -> buffer it % NBUF
        const unsigned dst = lds0 + (unsigned)(it % NBUF) * (STEP_ELEMS * 2);
        const unsigned short* src = xb + (size_t)(step0 + it * stride) * STEP_ELEMS;
        for (int p = wave; p < STEP_PIECES; p += WAVES)
            glds16_asm(reinterpret_cast<const float*>(src + p * 512 + lane * 8), __builtin_amdgcn_readfirstlane(dst + p * 1024));
    };
#pragma unroll
    for (int j = 0; j < NBUF - 1; ++j)
        if (j < n_it) request(j);
    dma_wait_all();
    __syncthreads();
    dbg_stamp(a.times, blockIdx.y * gridDim.x + blockIdx.x, 1);

    int cur = 0;
    for (int it = 0; it < n_it; ++it) {
        const int step = step0 + it * stride;
        const bool ahead = it + NBUF - 1 < n_it;
        if (ahead && !(VAR & 1)) request(it + NBUF - 1);                    // into the buffer the previous step has left
        // two 32-key row tiles x QW query tiles, MFMA chains interleaved so that no instruction waits on its predecessor
        f32x16 acc[QW][2];
        if (VAR & 64) {                                  // ablation: accumulators carried across the steps, no epilogue
#pragma unroll
            for (int w = 0; w < QW; ++w)
#pragma unroll
                for (int r = 0; r < 16; ++r) { acc[w][0][r] = gm[w][r]; acc[w][1][r] = gm[w][(r + 1) & 15]; }
        } else {
#pragma unroll
        for (int w = 0; w < QW; ++w)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[w][0][r] = 0.f; acc[w][1][r] = 0.f; }
        }
        const unsigned short* kp0 = &sK[cur][i * DSH + 8 * h];
        const unsigned short* kp1 = kp0 + 32 * DSH;
#pragma unroll
        for (int t = 0; t < KB; ++t) {
            bf16x8 k0, k1;
            if (VAR & 16) { k0 = qf[0][t]; k1 = qf[0][(t + 1) % KB]; }
            else {
                k0 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const s16x8*>(kp0 + 16 * t));
                k1 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const s16x8*>(kp1 + 16 * t));
            }
            if (VAR & 2) {
#pragma unroll
                for (int w = 0; w < QW; ++w) { acc[w][0][t] += (float)k0[0] * (float)qf[w][t][0]; acc[w][1][t] += (float)k1[0] * (float)qf[w][t][1]; }
                continue;
            }
#pragma unroll
            for (int w = 0; w < QW; ++w) {
                acc[w][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k0, qf[w][t], acc[w][0], 0, 0, 0);
                acc[w][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k1, qf[w][t], acc[w][1], 0, 0, 0);
            }
        }
#pragma unroll
        for (int w = 0; w < QW; ++w) {
            if (VAR & 64) {
#pragma unroll
                for (int r = 0; r < 16; ++r) gm[w][r] = acc[w][0][r] + acc[w][1][r];
                continue;
            }
            if (VAR & 4) { gm[w][0] += acc[w][0][0] + acc[w][1][0]; continue; }
            if (PASS == 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) gm[w][r] = fmaxf(fmaxf(gm[w][r], acc[w][0][r]), acc[w][1][r]);
            } else {
                // common case (no candidate in the whole wave): 16 v_max3 + one compare + one branch
                float mxt[2];
#pragma unroll
                for (int tl = 0; tl < 2; ++tl) {
                    float m = acc[w][tl][0];
#pragma unroll
                    for (int r = 1; r < 16; ++r) m = fmaxf(m, acc[w][tl][r]);
                    mxt[tl] = m;
                }
                if (__any(fmaxf(mxt[0], mxt[1]) >= thq[w])) {
                    // ~3 of 4 wave-steps hold a candidate somewhere in their 2048 scores, so this path matters: one tile at a
                    // time, two VALU per score (sign of thlo - S~, shifted into the lane's mask by v_alignbit)
                    if (SMAX) smx[w] = fmaxf(smx[w], fmaxf(mxt[0], mxt[1]));       // (a tile maximum below theta is below every candidate)
#pragma unroll
                    for (int tl = 0; tl < 2; ++tl) {
                        if (!__any(mxt[tl] >= thq[w])) continue;
                        unsigned mask = 0;
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            mask = __builtin_amdgcn_alignbit(mask, __float_as_uint(thlo[w] - acc[w][tl][r]), 31);
                        const int kbase = step * SK + 4 * h + tl * 32;
                        // the screened score travels with the key: exact when the lane has one candidate in this tile (it is
                        // the tile maximum), otherwise the maximum with the sign bit set = "upper bound only"
                        const float sv = (__popc(mask) == 1) ? mxt[tl] : -mxt[tl];
                        while (mask) {                                   // score r sits at bit 15 - r: ascending r
                            const int bit = 31 - __clz((int)mask);
                            mask &= ~(1u << bit);
                            const int r = 15 - bit;
                            const int key = kbase + (r & 3) + 8 * (r >> 2);
                            if (n_loc[w] < a.capseg - 1) cseg[w][(size_t)n_loc[w] * (a.splits * 2)] = make_int2(key, __float_as_int(sv));
                            else if (a.spill != nullptr) screen_spill(a, (size_t)b * a.L + (size_t)((qg * (SCR_QUERIES / (QW * QT)) + wave) * QW + w) * QT + i, key, sv);
                            ++n_loc[w];
                        }
                    }
                }
            }
        }
        // the next tile must have landed; the NBUF-2 tiles requested after it may still be in flight
        if (ahead && NBUF > 2) { if (more) dma_wait_le<(NBUF - 2) * (PPW + 1)>(); else dma_wait_le<(NBUF - 2) * PPW>(); }
        else dma_wait_all();
        if (!(VAR & 32)) __syncthreads();
        cur = (cur + 1 == NBUF) ? 0 : cur + 1;
    }
    dbg_stamp(a.times, blockIdx.y * gridDim.x + blockIdx.x, 2);

#pragma unroll
    for (int w = 0; w < QW; ++w) {
        if (PASS == 0) {
            if (a.theta_max != nullptr) {
                // top-1 use (the dense formulation's row maxima): the threshold is the query's LARGEST sampled score -- an integer
                // maximum over the lanes' values (scores are >= 0: their bit patterns order like the values; the word was zeroed by
                // query_thresholds_kernel), no separate threshold kernel
                float m = gm[w][0];
#pragma unroll
                for (int r = 1; r < 16; ++r) m = fmaxf(m, gm[w][r]);
                if (qvalid[w] && m > 0.f) atomicMax(a.theta_max + (seg[w] >> 1) / a.splits, __float_as_int(m));
                continue;
            }
            if (a.gkeep == 16) {
                // few segments per query (small maps, large batches): ALL sixteen group maxima of the lane go to the threshold kernel --
                // with four of them per segment a query had 8 * chunks values to take its k-th largest from: fewer than k = 50 on a
                // batch of 64 leaf tiles (theta = 0, every key a candidate, every query group on the fp32 redo pass: 12 ms a call)
                if (qvalid[w]) {
                    float4* o = reinterpret_cast<float4*>(a.gmax + seg[w] * 16);
#pragma unroll
                    for (int u = 0; u < 4; ++u) o[u] = make_float4(gm[w][4 * u], gm[w][4 * u + 1], gm[w][4 * u + 2], gm[w][4 * u + 3]);
                }
                continue;
            }
            // keep the lane's GKEEP largest group maxima: GKEEP distinct keys, so still a valid pool for the
            // k-th-largest lower bound, at a quarter of the traffic into the theta kernel
            float top[GKEEP];
#pragma unroll
            for (int u = 0; u < GKEEP; ++u) {
                float m = gm[w][0];
#pragma unroll
                for (int r = 1; r < 16; ++r) m = fmaxf(m, gm[w][r]);
                top[u] = m;
                bool taken = false;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const bool hit = !taken && (gm[w][r] == m);
                    gm[w][r] = hit ? -1.0f : gm[w][r];
                    taken = taken || hit;
                }
            }
            if (qvalid[w]) *reinterpret_cast<float4*>(a.gmax + seg[w] * GKEEP) = make_float4(top[0], top[1], top[2], top[3]);
        } else {
            if (VAR & 64) { float t = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) t += gm[w][r];
                if (t == 12345.f) n_loc[w] = 1; }
            if (qvalid[w]) cseg[w][-(a.splits * 2)] = make_int2(n_loc[w], SMAX ? __float_as_int(smx[w]) : 0);
        }
    }
    dbg_stamp(a.times, blockIdx.y * gridDim.x + blockIdx.x, 3);
}

// ---- ring form of the same kernel: no block barrier in the step loop ---------------------------------------------------
// screen_kernel's step is "26 multiplies per wave, then the wave's tail (threshold test, candidate extraction, its share of the
// next tile request), then a barrier of all 16 waves".  The four waves of a SIMD share one matrix pipe (oldest first), so
// they leave the multiplies one after the other -- and the LAST one's tail runs with the pipe idle before the barrier lets
// anybody start the next step: step = 4 x 832 cycles of multiplies + ~1400 of tail, the matrix pipes 63 % busy in the loop
// (DESIGN.md section 7).  Here the waves are not re-aligned every step: key tiles live in a five-deep LDS ring guarded by two
// words per slot instead of the barrier,
//   ready[slot] = 1 + the tile that has landed there   (written by the wave that requested it, after its vmcnt wait)
//   done[slot]  = number of wave-steps that have finished reading the slot (one LDS add per wave and step)
// a wave starts step t as soon as ready[t % 5] says tile t is there -- while slower waves are still in their tails --, and tile
// t + 3 is requested (all 27 LDS-DMA pieces) by ONE wave, (t + 3) % 16, at the start of ITS step t, once done[] says everybody
// has left the slot's previous tile, t - 2.  Waves drift by up to a step against each other; a tail runs under the other waves'
// multiplies.  LDS operations of a wave execute in issue order and LDS-DMA data is in the LDS when the requesting wave's vmcnt
// has counted it (MI355X_MICROARCH.md, two-waves-per-SIMD item 7), so flag-after-wait / read-after-flag needs no barrier.
constexpr int RING_NBUF = 5;                       // 5 x 27 KiB = 135 KiB of the CU's 160
constexpr int RING_AHEAD = 3;                      // tile t + 3 is requested during step t, published during step t + 1
constexpr int RING_FLAG_BYTES = 64;

// VAR (ablation builds only, results wrong by construction): 1 no tile requests, 2 theta = inf (no extraction), 4 key fragments
// from registers, 8 no ready / done words, 16 no threshold test at all
// QW = query tiles of 32 per wave: 1 = 16 waves x 32 queries (four waves per SIMD, 128 registers), 2 = 8 waves x 64 queries (two per
// SIMD, 256 registers: every key fragment read from the LDS feeds two multiplies -- half the LDS traffic, requests and flag
// words per multiply)
template <int PASS, int WAVES, int VAR = 0, int QW = 1, bool SMAX = false>
__global__ __launch_bounds__(WAVES * 64, 1) void screen_ring_kernel(ScreenArgs a, int n_qgroups) {
    if (a.gate != nullptr && *a.gate == 0) return;                 // (a re-run launch of a cold workspace that is not needed)
    if (a.policy != nullptr && *a.policy != 0) { a.capseg = a.capseg_tight; a.sample = a.sample_tight; }   // device-side threshold policy
    // ONE shared object (a second one makes hipcc drain vmcnt(0) in front of every ds_read of the loop)
    __shared__ __attribute__((aligned(16))) unsigned short smem[RING_NBUF * STEP_ELEMS + RING_FLAG_BYTES / 2];
    unsigned* const flags = reinterpret_cast<unsigned*>(smem + RING_NBUF * STEP_ELEMS);     // ready[0..4] at +0, done[0..4] at +32 bytes

    const int tid = threadIdx.x;
    if (!(VAR & 64)) dbg_stamp(a.times, blockIdx.y * gridDim.x + blockIdx.x, 0);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int i = lane & 31, h = lane >> 5;
    const int b = blockIdx.y;

    const int logical = xcd_remap2(blockIdx.x, gridDim.x);
    const int split = logical / n_qgroups;
    const int qg = logical % n_qgroups;
    const int step0 = split * a.steps_per_split;
    int step1 = step0 + a.steps_per_split;
    if (step1 > a.n_steps) step1 = a.n_steps;
    const int stride = (PASS == 0) ? a.sample : 1;

    // query fragments: QW x 13 x 8 bf16: Wq~[q][16t + 8h .. +7]
    bf16x8 qf[QW][KB];
    bool qvalid[QW];
    float thq[QW], thlo[QW];          // thlo = predecessor of thq:  S~ >= thq  <=>  S~ > thlo  <=>  sign(thlo - S~) set
    int n_loc[QW];
    int2* cseg[QW];
    size_t seg[QW];
#pragma unroll
    for (int w = 0; w < QW; ++w) {
        const int q = ((qg * WAVES + wave) * QW + w) * QT + i;
        qvalid[w] = q < a.L;
        const int qc = qvalid[w] ? q : a.L - 1;
        const size_t qlin = (size_t)b * a.L + qc;
        load_query_fragments(a, b, q - i, qc, lane, qf[w]);
        thq[w] = 0.f;
        if (PASS == 1) thq[w] = (qvalid[w] && !(VAR & 2)) ? a.theta[qlin] : __builtin_inff();
        if (PASS == 1 && SMAX && qvalid[w]) thq[w] = theta_of_sampled_max(thq[w]);      // (a.theta holds the raw sampled maximum)
        const unsigned tb = __float_as_uint(thq[w]);
        thlo[w] = __uint_as_float(thq[w] > 0.f ? tb - 1u : (thq[w] == 0.f ? 0x80000001u : tb + 1u));
        n_loc[w] = 0;
        seg[w] = (qlin * a.splits + split) * 2 + h;
        cseg[w] = a.cand + (qlin * a.capseg + 1) * (size_t)(a.splits * 2) + split * 2 + h;   // slot 1 (slot 0 = header); slots are splits * 2 apart
    }
#pragma unroll
    for (int w = 0; w < QW; ++w)
#pragma unroll
        for (int t = 0; t < KB; ++t) asm volatile("" : "+v"(qf[w][t]));

    float gm[QW][16];
#pragma unroll
    for (int w = 0; w < QW; ++w)
#pragma unroll
        for (int r = 0; r < 16; ++r) gm[w][r] = -1.0f;
    float smx[QW];                                    // SMAX: largest screened score among this lane's (= this segment's) candidates
#pragma unroll
    for (int w = 0; w < QW; ++w) smx[w] = -1.0f;

    const unsigned short* xb = a.xh + (size_t)b * a.rows_xh * DSH;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr_of(&smem[0]));
    const int n_it = (step1 > step0) ? (step1 - step0 + stride - 1) / stride : 0;

    // prologue: the first RING_AHEAD tiles are requested by all waves together (one barrier, outside the loop)
    const unsigned ready0 = lds0 + RING_NBUF * STEP_ELEMS * 2, done0 = ready0 + 32;        // LDS byte addresses of the flag words
    if (tid < RING_NBUF) { flags[tid] = (tid < RING_AHEAD && tid < n_it) ? (unsigned)(tid + 1) : 0u; flags[8 + tid] = 0u; }
#pragma unroll
    for (int j = 0; j < RING_AHEAD; ++j)
        if (j < n_it) {
            const unsigned dst = lds0 + (unsigned)j * (STEP_ELEMS * 2);
            const unsigned short* src = xb + (size_t)(step0 + j * stride) * STEP_ELEMS;
#if !defined(DAGL_RING_M0_PER_PIECE)
            // a wave's pieces of the 27 are adjacent and share one M0 value (glds16x2_asm: a wave that changes M0 waits for its previous
            // request to leave the issue stage): waves 0 .. 12 two pieces, wave 13 the last one
            if (WAVES >= 14) {
                const int p = 2 * wave;
                if (p + 1 < STEP_PIECES) glds16x2_asm(reinterpret_cast<const float*>(src + p * 512 + lane * 8), __builtin_amdgcn_readfirstlane(dst + p * 1024));
                else if (p < STEP_PIECES) glds16_asm(reinterpret_cast<const float*>(src + p * 512 + lane * 8), __builtin_amdgcn_readfirstlane(dst + p * 1024));
            } else
#endif
            for (int p = wave; p < STEP_PIECES; p += WAVES)
                glds16_asm(reinterpret_cast<const float*>(src + p * 512 + lane * 8), __builtin_amdgcn_readfirstlane(dst + p * 1024));
        }
    dma_wait_all();
    __syncthreads();
    if (!(VAR & 64)) dbg_stamp(a.times, blockIdx.y * gridDim.x + blockIdx.x, 1);

    int pend_tile = -1;                              // tile this wave has requested and not yet published (wave-uniform)
    int buf = 0;                                     // it % RING_NBUF
    // VAR & 64 (ablation): per-wave sums of shader-clock deltas: 0 tile request (+ wait for the slot), 1 reads + multiplies,
    // 2 test + extraction, 3 wait for the tile's ready word, 4 publishing (vmcnt wait)
    unsigned long long ph[5] = {0, 0, 0, 0, 0};
    unsigned long long ph_t = (VAR & 64) ? __builtin_amdgcn_s_memtime() : 0ull;
    const unsigned long long clk0 = (VAR & 128) ? __builtin_amdgcn_s_memtime() : 0ull;      // VAR & 128: shader clocks of the loop -> stamp 3
#define RING_PH(k) do { if (VAR & 64) { const unsigned long long t1_ = __builtin_amdgcn_s_memtime(); ph[k] += t1_ - ph_t; ph_t = t1_; } } while (0)
    for (int it = 0; it < n_it; ++it) {
        const int step = step0 + it * stride;
        // --- requester duty: tile it + AHEAD belongs to wave (it + AHEAD) % WAVES ---------------------------------------
        {
            const int T = it + RING_AHEAD;
            if (T < n_it && (T % WAVES) == wave && !(VAR & 1)) {
                const int tb = T % RING_NBUF;
                const unsigned need = (unsigned)(WAVES * (T / RING_NBUF));        // wave-steps that have used the slot before
                if (need && !(VAR & 8)) {
                    while (lds_flag_load(done0 + 4 * tb) < need) __builtin_amdgcn_s_sleep(1);
                }
                const unsigned dst = lds0 + (unsigned)tb * (STEP_ELEMS * 2);
                {
                    const unsigned long long sa = (unsigned long long)(uintptr_t)(xb + (size_t)(step0 + T * stride) * STEP_ELEMS);
                    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)sa);
                    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(sa >> 32));
#if !defined(DAGL_TILE27_X4)
                    glds_tile27_x8_asm(reinterpret_cast<const void*>((uintptr_t)(((unsigned long long)hi << 32) | lo)), (unsigned)lane * 16u,
                                       __builtin_amdgcn_readfirstlane(dst));
#else
                    glds_tile27_asm(reinterpret_cast<const void*>((uintptr_t)(((unsigned long long)hi << 32) | lo)), (unsigned)lane * 16u,
                                    __builtin_amdgcn_readfirstlane(dst));
#endif
                }
                pend_tile = T;
            }
        }
        RING_PH(0);
        // --- tile `it` must have been published -----------------------------------------------------------------------------
        if (!(VAR & 1) && !(VAR & 8))
            while (lds_flag_load(ready0 + 4 * buf) < (unsigned)(it + 1)) __builtin_amdgcn_s_sleep(1);
        RING_PH(3);

        f32x16 acc[QW][2];
#pragma unroll
        for (int w = 0; w < QW; ++w)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[w][0][r] = 0.f; acc[w][1][r] = 0.f; }
        const unsigned short* kp0 = &smem[buf * STEP_ELEMS + i * DSH + 8 * h];
        const unsigned short* kp1 = kp0 + 32 * DSH;
#pragma unroll
        for (int t = 0; t < KB; ++t) {
            bf16x8 k0, k1;
            if (VAR & 4) { k0 = qf[0][t]; k1 = qf[0][(t + 1) % KB]; }
            else {
                k0 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const s16x8*>(kp0 + 16 * t));
                k1 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const s16x8*>(kp1 + 16 * t));
            }
#pragma unroll
            for (int w = 0; w < QW; ++w) {
                acc[w][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k0, qf[w][t], acc[w][0], 0, 0, 0);
                acc[w][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k1, qf[w][t], acc[w][1], 0, 0, 0);
            }
        }
        // issue order of the block above: key fragments are read RING_PF taps ahead of their multiplies (left alone, hipcc waits
        // for a fragment two instructions after asking for it: one exposed LDS latency per tap and wave)
        if (!(VAR & 4)) {
#ifdef DAGL_RING_PF
            constexpr int PF = DAGL_RING_PF;
#else
            constexpr int PF = (QW == 2) ? 3 : 2;
#endif
            __builtin_amdgcn_sched_group_barrier(0x100, 2 * PF, 0);
#pragma unroll
            for (int t = 0; t < KB; ++t) {
                __builtin_amdgcn_sched_group_barrier(0x008, 2 * QW, 0);
                if (t + PF < KB) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            }
        }
        // every read of the slot has been issued (LDS executes a wave's operations in order): one more wave-step is through
        if (lane == 0 && !(VAR & 8)) lds_flag_add(done0 + 4 * buf, 1u);
        if (VAR & 64) { asm volatile("s_nop 0" :: "v"(acc[0][0][0]), "v"(acc[0][1][15])); }       // (the multiplies have completed)
        RING_PH(1);
        // --- publisher duty: the tile this wave requested a step ago has had ~1.7 steps to land ---------------------------
        if (pend_tile >= 0 && it > pend_tile - RING_AHEAD) {
            dma_wait_all();
            if (lane == 0) lds_flag_store(ready0 + 4 * (pend_tile % RING_NBUF), (unsigned)(pend_tile + 1));
            pend_tile = -1;
        }
        RING_PH(4);
        // --- tail: threshold test, candidate extraction ---------------------------------------------------------------------
#pragma unroll
        for (int w = 0; w < QW; ++w) {
            if (VAR & 16) {
                gm[w][0] += acc[w][0][0] + acc[w][1][0];
            } else if (PASS == 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) gm[w][r] = fmaxf(fmaxf(gm[w][r], acc[w][0][r]), acc[w][1][r]);
            } else {
                // common case (no candidate in the whole wave): 16 v_max3 + one compare + one branch
                float mxt[2];
#pragma unroll
                for (int tl = 0; tl < 2; ++tl) {
                    float m = acc[w][tl][0];
#pragma unroll
                    for (int r = 1; r < 16; ++r) m = fmaxf(m, acc[w][tl][r]);
                    mxt[tl] = m;
                }
                if (__any(fmaxf(mxt[0], mxt[1]) >= thq[w])) {
                    if (SMAX) smx[w] = fmaxf(smx[w], fmaxf(mxt[0], mxt[1]));       // (a tile maximum below theta is below every candidate)
#pragma unroll
                    for (int tl = 0; tl < 2; ++tl) {
                        if (!__any(mxt[tl] >= thq[w])) continue;
                        unsigned mask = 0;
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            mask = __builtin_amdgcn_alignbit(mask, __float_as_uint(thlo[w] - acc[w][tl][r]), 31);
                        const int kbase = step * SK + 4 * h + tl * 32;
                        // the screened score travels with the key: exact when the lane has one candidate in this tile (it is
                        // the tile maximum), otherwise the maximum with the sign bit set = "upper bound only"
                        const float sv = (__popc(mask) == 1) ? mxt[tl] : -mxt[tl];
                        while (mask) {                                   // score r sits at bit 15 - r: ascending r
                            const int bit = 31 - __clz((int)mask);
                            mask &= ~(1u << bit);
                            const int r = 15 - bit;
                            const int key = kbase + (r & 3) + 8 * (r >> 2);
                            if (n_loc[w] < a.capseg - 1) cseg[w][(size_t)n_loc[w] * (a.splits * 2)] = make_int2(key, __float_as_int(sv));
                            else if (a.spill != nullptr) screen_spill(a, (size_t)b * a.L + (size_t)((qg * WAVES + wave) * QW + w) * QT + i, key, sv);
                            ++n_loc[w];
                        }
                    }
                }
            }
        }
        buf = (buf + 1 == RING_NBUF) ? 0 : buf + 1;
        RING_PH(2);
    }
#undef RING_PH
    const unsigned long long clk1 = (VAR & 128) ? __builtin_amdgcn_s_memtime() : 0ull;
    if (!(VAR & 64)) dbg_stamp(a.times, blockIdx.y * gridDim.x + blockIdx.x, 2);
    if ((VAR & 64) && a.times != nullptr && lane == 0) {
        unsigned long long* o = a.times + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * WAVES + wave) * 5;
        for (int k = 0; k < 5; ++k) o[k] = ph[k];
    }

#pragma unroll
    for (int w = 0; w < QW; ++w) {
        if (PASS == 0) {
            if (a.theta_max != nullptr) {
                // top-1 use (the dense formulation's row maxima): the threshold is the query's LARGEST sampled score -- an integer
                // maximum over the lanes' values (scores are >= 0: their bit patterns order like the values; the word was zeroed by
                // query_thresholds_kernel), no separate threshold kernel
                float m = gm[w][0];
#pragma unroll
                for (int r = 1; r < 16; ++r) m = fmaxf(m, gm[w][r]);
                if (qvalid[w] && m > 0.f) atomicMax(a.theta_max + (seg[w] >> 1) / a.splits, __float_as_int(m));
                continue;
            }
            if (a.gkeep == 16) {
                // few segments per query (small maps, large batches): ALL sixteen group maxima of the lane go to the threshold kernel --
                // with four of them per segment a query had 8 * chunks values to take its k-th largest from: fewer than k = 50 on a
                // batch of 64 leaf tiles (theta = 0, every key a candidate, every query group on the fp32 redo pass: 12 ms a call)
                if (qvalid[w]) {
                    float4* o = reinterpret_cast<float4*>(a.gmax + seg[w] * 16);
#pragma unroll
                    for (int u = 0; u < 4; ++u) o[u] = make_float4(gm[w][4 * u], gm[w][4 * u + 1], gm[w][4 * u + 2], gm[w][4 * u + 3]);
                }
                continue;
            }
            // keep the lane's GKEEP largest group maxima: GKEEP distinct keys, so still a valid pool for the
            // k-th-largest lower bound, at a quarter of the traffic into the theta kernel
            float top[GKEEP];
#pragma unroll
            for (int u = 0; u < GKEEP; ++u) {
                float m = gm[w][0];
#pragma unroll
                for (int r = 1; r < 16; ++r) m = fmaxf(m, gm[w][r]);
                top[u] = m;
                bool taken = false;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const bool hit = !taken && (gm[w][r] == m);
                    gm[w][r] = hit ? -1.0f : gm[w][r];
                    taken = taken || hit;
                }
            }
            if (qvalid[w]) *reinterpret_cast<float4*>(a.gmax + seg[w] * GKEEP) = make_float4(top[0], top[1], top[2], top[3]);
        } else {
            if ((VAR & 16) && gm[w][0] == 12345.f) n_loc[w] = 1;      // (keeps the ablated loop's accumulators alive)
            if (qvalid[w]) cseg[w][-(a.splits * 2)] = make_int2(n_loc[w], SMAX ? __float_as_int(smx[w]) : 0);
        }
    }
    if (!(VAR & 64)) dbg_stamp(a.times, blockIdx.y * gridDim.x + blockIdx.x, 3);
#ifdef DAGL_ABLATION
    if ((VAR & 128) && a.times != nullptr && tid == 0) a.times[(size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 4 + 3] = clk1 - clk0;
#endif
}

#ifdef DAGL_ABLATION
// ---- pipelined form of the ring kernel (ablation builds only, DAGL_SCREEN_RING=3; results identical, 2-7 us SLOWER: kept as the
// record of that measurement, profiles/r03_ab_screen_pipe.log): a wave's threshold test runs under its OWN multiplies --------------
// In screen_ring_kernel a wave alternates "26 multiplies" and "tail" (~130 vector instructions when the tile holds candidates,
// which at 3 candidates per 32 x 32 tile is nearly always); the matrix pipe of a SIMD is busy only while at least one of its
// four waves is in the first phase, and the waves' waits for tiles and slots line their phases up often enough to leave it
// idle 40 % of the loop.  Here the step is cut into its two 32-key halves and software-pipelined inside the wave: while the 13
// multiplies of one half run into one accumulator, the wave tests the OTHER accumulator (the previous half) -- mask of the
// 16 scores above the threshold, tile maximum, first candidate -- in the issue slots between the multiplies (a 32x32x16 MFMA
// occupies the pipe for 32 cycles and the issue port for 4: seven vector instructions fit behind each).  The half is ONE basic
// block: the candidate store is predicated inside an asm statement (exec from a lane flag) instead of a branch; only a lane
// with a second candidate in the same 16 scores sends the wave through the scalar loop afterwards.  Same ring, same counters,
// same records in the same order as screen_ring_kernel.
typedef int i32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void store8_if(const void* p, i32x2 v, int ok) {
    unsigned long long keep;
    asm volatile("s_mov_b64 %0, exec\n\tv_cmpx_ne_u32_e32 0, %1\n\tglobal_store_dwordx2 %2, %3, off\n\ts_mov_b64 exec, %0"
                 : "=&s"(keep)
                 : "v"(ok), "v"(p), "v"(v)
                 : "vcc");
}

// plain (not interleaved) test of one finished accumulator, for the last half of a block (32 keys x 32 queries; a lane holds 16
// scores of one query).  PASS 0: group maxima.  PASS 1: the first candidate is stored here; returns the mask of the lane's
// FURTHER candidates (score r at bit 15 - r).  screen_pipe_kernel's `half` spells the same test out in slices.
template <int PASS>
__device__ __forceinline__ unsigned screen_test(const f32x16& acc, float thq, float thlo, int kbase, int2* cseg, int s2, int& n_loc, int cap,
                                                float (&gm)[16], float& sv_out) {
    if constexpr (PASS == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) gm[r] = fmaxf(gm[r], acc[r]);
        return 0u;
    } else {
        unsigned mask = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) mask = __builtin_amdgcn_alignbit(mask, __float_as_uint(thlo - acc[r]), 31);
        float m = acc[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) m = fmaxf(m, acc[r]);
        const float sv = (__popc(mask) == 1) ? m : -m;
        const int has = mask != 0u ? 1 : 0;
        const int r = __clz((int)mask) - 16;                       // first candidate (lowest r); 16 when there is none
        const int key = kbase + (r & 3) + 8 * (r >> 2);
        i32x2 rec; rec[0] = key; rec[1] = __float_as_int(sv);
        store8_if(cseg + (size_t)n_loc * s2, rec, (has && n_loc < cap) ? 1 : 0);
        n_loc += has;
        sv_out = sv;
        return has ? (mask & ~(0x8000u >> r)) : 0u;
    }
}
__device__ __forceinline__ void screen_rest(unsigned rest, float sv, int kbase, int2* cseg, int s2, int& n_loc, int cap) {
    while (rest) {
        const int bit = 31 - __clz((int)rest);
        rest &= ~(1u << bit);
        const int r = 15 - bit;
        const int key = kbase + (r & 3) + 8 * (r >> 2);
        if (n_loc < cap) cseg[(size_t)n_loc * s2] = make_int2(key, __float_as_int(sv));
        ++n_loc;
    }
}

// QW = query tiles of 32 per wave.  2 (8 waves x 64 queries, two per SIMD, 256 registers): every key fragment read from the LDS
// feeds two multiplies -- half the LDS reads (a co-bottleneck at QW = 1: 26 KiB per wave and step, the LDS pipe busy half the
// time the matrix pipe is), ring words and tile requests per multiply; the wave hides its own tails, so two waves per SIMD
// are enough to keep the pipe fed.
template <int PASS, int WAVES, int VAR = 0, int QW = 1>
__global__ __launch_bounds__(WAVES * 64, 1) void screen_pipe_kernel(ScreenArgs a, int n_qgroups) {
    if (a.gate != nullptr && *a.gate == 0) return;                 // (a re-run launch of a cold workspace that is not needed)
    if (a.policy != nullptr && *a.policy != 0) { a.capseg = a.capseg_tight; a.sample = a.sample_tight; }   // device-side threshold policy
    // ONE shared object (a second one makes hipcc drain vmcnt(0) in front of every ds_read of the loop)
    __shared__ __attribute__((aligned(16))) unsigned short smem[RING_NBUF * STEP_ELEMS + RING_FLAG_BYTES / 2];
    unsigned* const flags = reinterpret_cast<unsigned*>(smem + RING_NBUF * STEP_ELEMS);     // ready[0..4] at +0, done[0..4] at +32 bytes

    const int tid = threadIdx.x;
    dbg_stamp(a.times, blockIdx.y * gridDim.x + blockIdx.x, 0);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int i = lane & 31, h = lane >> 5;
    const int b = blockIdx.y;

    const int logical = xcd_remap2(blockIdx.x, gridDim.x);
    const int split = logical / n_qgroups;
    const int qg = logical % n_qgroups;
    const int step0 = split * a.steps_per_split;
    int step1 = step0 + a.steps_per_split;
    if (step1 > a.n_steps) step1 = a.n_steps;
    const int stride = (PASS == 0) ? a.sample : 1;

    // query fragments: QW x 13 x 8 bf16: Wq~[q][16t + 8h .. +7]
    bf16x8 qf[QW][KB];
    bool qvalid[QW];
    float thq[QW], thlo[QW];          // thlo = predecessor of thq:  S~ >= thq  <=>  S~ > thlo  <=>  sign(thlo - S~) set
    int n_loc[QW];
    int2* cseg[QW];
    size_t seg[QW];
#pragma unroll
    for (int w = 0; w < QW; ++w) {
        const int q = ((qg * WAVES + wave) * QW + w) * QT + i;
        qvalid[w] = q < a.L;
        const int qc = qvalid[w] ? q : a.L - 1;
        const size_t qlin = (size_t)b * a.L + qc;
        load_query_fragments(a, b, q - i, qc, lane, qf[w]);
        thq[w] = 0.f;
        if (PASS == 1) thq[w] = (qvalid[w] && !(VAR & 2)) ? a.theta[qlin] : __builtin_inff();
        const unsigned tb = __float_as_uint(thq[w]);
        thlo[w] = __uint_as_float(thq[w] > 0.f ? tb - 1u : (thq[w] == 0.f ? 0x80000001u : tb + 1u));
        n_loc[w] = 0;
        seg[w] = (qlin * a.splits + split) * 2 + h;
        cseg[w] = a.cand + (qlin * a.capseg + 1) * (size_t)(a.splits * 2) + split * 2 + h;   // slot 1 (slot 0 = header); slots are splits * 2 apart
    }
    const int cap = a.capseg - 1;
#pragma unroll
    for (int w = 0; w < QW; ++w)
#pragma unroll
        for (int t = 0; t < KB; ++t) asm volatile("" : "+v"(qf[w][t]));

    float gm[QW][16];
#pragma unroll
    for (int w = 0; w < QW; ++w)
#pragma unroll
        for (int r = 0; r < 16; ++r) gm[w][r] = -1.0f;

    const unsigned short* xb = a.xh + (size_t)b * a.rows_xh * DSH;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr_of(&smem[0]));
    const int n_it = (step1 > step0) ? (step1 - step0 + stride - 1) / stride : 0;

    // prologue: the first RING_AHEAD tiles are requested by all waves together (one barrier, outside the loop)
    const unsigned ready0 = lds0 + RING_NBUF * STEP_ELEMS * 2, done0 = ready0 + 32;        // LDS byte addresses of the flag words
    if (tid < RING_NBUF) { flags[tid] = (tid < RING_AHEAD && tid < n_it) ? (unsigned)(tid + 1) : 0u; flags[8 + tid] = 0u; }
#pragma unroll
    for (int j = 0; j < RING_AHEAD; ++j)
        if (j < n_it) {
            const unsigned dst = lds0 + (unsigned)j * (STEP_ELEMS * 2);
            const unsigned short* src = xb + (size_t)(step0 + j * stride) * STEP_ELEMS;
            for (int p = wave; p < STEP_PIECES; p += WAVES)
                glds16_asm(reinterpret_cast<const float*>(src + p * 512 + lane * 8), __builtin_amdgcn_readfirstlane(dst + p * 1024));
        }
    dma_wait_all();
    __syncthreads();
    dbg_stamp(a.times, blockIdx.y * gridDim.x + blockIdx.x, 1);

    int pend_tile = -1;                              // tile this wave has requested and not yet published (wave-uniform)
    int buf = 0;                                     // it % RING_NBUF
    f32x16 accA[QW], accB[QW];                       // keys 0..31 / 32..63 of a step
#pragma unroll
    for (int w = 0; w < QW; ++w)
#pragma unroll
        for (int r = 0; r < 16; ++r) { accA[w][r] = 0.f; accB[w][r] = -__builtin_inff(); }     // (the first half tests an empty predecessor)
    int kb_prev = 0;                                 // first key of the half whose accumulator awaits its test (+ 4h)

    // one half: 13 (x QW) multiplies of 32 keys into `fresh`, the test of `old` in their shadow.  The interleave is spelled out:
    // slice t = fragment t + 2 asked for, multiply t, five-odd instructions of the test per query tile, and a scheduling fence
    // (hipcc's own order puts the test behind the last multiply).  The test starts at slice 2: `old` was completed by the
    // multiply issued just before.
    auto half = [&](const unsigned short* kp, f32x16 (&fresh)[QW], const f32x16 (&old)[QW], int kb_old) {
        f32x16 c[QW];
#pragma unroll
        for (int w = 0; w < QW; ++w)
#pragma unroll
            for (int r = 0; r < 16; ++r) c[w][r] = 0.f;
        bf16x8 kf[KB];
        kf[0] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const s16x8*>(kp));
        kf[1] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const s16x8*>(kp + 16));
        unsigned mask[QW], rest[QW];
        float m[QW], sv[QW];
        int has[QW], r1[QW], key[QW];
#pragma unroll
        for (int w = 0; w < QW; ++w) { mask[w] = 0u; rest[w] = 0u; m[w] = 0.f; sv[w] = 0.f; has[w] = 0; r1[w] = 0; key[w] = 0; }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < KB; ++t) {
            if (t + 2 < KB) kf[t + 2] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const s16x8*>(kp + 16 * (t + 2)));
#pragma unroll
            for (int w = 0; w < QW; ++w) c[w] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[t], qf[w][t], c[w], 0, 0, 0);
#pragma unroll
            for (int w = 0; w < QW; ++w) {
                if (t >= 2 && t < 10) {
                    const int r = 2 * (t - 2);
                    if constexpr (PASS == 0) {
                        gm[w][r] = fmaxf(gm[w][r], old[w][r]); gm[w][r + 1] = fmaxf(gm[w][r + 1], old[w][r + 1]);
                    } else {
                        mask[w] = __builtin_amdgcn_alignbit(mask[w], __float_as_uint(thlo[w] - old[w][r]), 31);
                        mask[w] = __builtin_amdgcn_alignbit(mask[w], __float_as_uint(thlo[w] - old[w][r + 1]), 31);
                        m[w] = (t == 2) ? fmaxf(old[w][0], old[w][1]) : fmaxf(fmaxf(m[w], old[w][r]), old[w][r + 1]);
                    }
                } else if (PASS == 1 && t == 10) {
                    // the screened score travels with the key: exact when the lane has one candidate among these 16 (it is their
                    // maximum), otherwise the maximum with the sign bit set = "upper bound only"
                    sv[w] = (__popc(mask[w]) == 1) ? m[w] : -m[w];
                    has[w] = mask[w] != 0u ? 1 : 0;
                    r1[w] = __clz((int)mask[w]) - 16;                // first candidate (lowest r); 16 when there is none
                    key[w] = kb_old + (r1[w] & 3) + 8 * (r1[w] >> 2);
                } else if (PASS == 1 && t == 11) {
                    i32x2 rec; rec[0] = key[w]; rec[1] = __float_as_int(sv[w]);
                    store8_if(cseg[w] + (size_t)n_loc[w] * (a.splits * 2), rec, (has[w] && n_loc[w] < cap) ? 1 : 0);
                    n_loc[w] += has[w];
                    rest[w] = has[w] ? (mask[w] & ~(0x8000u >> r1[w])) : 0u;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int w = 0; w < QW; ++w) fresh[w] = c[w];
        if constexpr (PASS == 1) {
            unsigned any_rest = rest[0];
#pragma unroll
            for (int w = 1; w < QW; ++w) any_rest |= rest[w];
            if (__any(any_rest != 0u)) {
#pragma unroll
                for (int w = 0; w < QW; ++w) screen_rest(rest[w], sv[w], kb_old, cseg[w], a.splits * 2, n_loc[w], cap);
            }
        }
    };

    for (int it = 0; it < n_it; ++it) {
        const int step = step0 + it * stride;
        // --- requester duty: tile it + AHEAD belongs to wave (it + AHEAD) % WAVES ---------------------------------------
        {
            const int T = it + RING_AHEAD;
            if (T < n_it && (T % WAVES) == wave && !(VAR & 1)) {
                const int tb = T % RING_NBUF;
                const unsigned need = (unsigned)(WAVES * (T / RING_NBUF));        // wave-steps that have used the slot before
                if (need) {
                    while (lds_flag_load(done0 + 4 * tb) < need) __builtin_amdgcn_s_sleep(1);
                }
                const unsigned dst = lds0 + (unsigned)tb * (STEP_ELEMS * 2);
                const unsigned long long sa = (unsigned long long)(uintptr_t)(xb + (size_t)(step0 + T * stride) * STEP_ELEMS);
                const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)sa);
                const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(sa >> 32));
                glds_tile27_asm(reinterpret_cast<const void*>((uintptr_t)(((unsigned long long)hi << 32) | lo)), (unsigned)lane * 16u,
                                __builtin_amdgcn_readfirstlane(dst));
                pend_tile = T;
            }
        }
        // --- tile `it` must have been published -----------------------------------------------------------------------------
        if (!(VAR & 1))
            while (lds_flag_load(ready0 + 4 * buf) < (unsigned)(it + 1)) __builtin_amdgcn_s_sleep(1);

        const unsigned short* kp0 = &smem[buf * STEP_ELEMS + i * DSH + 8 * h];
        const int kb0 = step * SK + 4 * h;
        half(kp0, accA, accB, kb_prev);                          // keys 0..31 of this step  | test of keys 32..63 of the previous one
        half(kp0 + 32 * DSH, accB, accA, kb0);                   // keys 32..63              | test of keys 0..31
        kb_prev = kb0 + 32;
        // every read of the slot has been issued (LDS executes a wave's operations in order): one more wave-step is through
        if (lane == 0) lds_flag_add(done0 + 4 * buf, 1u);
        // --- publisher duty: the tile this wave requested a step ago has had ~1.7 steps to land ---------------------------
        if (pend_tile >= 0 && it > pend_tile - RING_AHEAD) {
            dma_wait_all();
            if (lane == 0) lds_flag_store(ready0 + 4 * (pend_tile % RING_NBUF), (unsigned)(pend_tile + 1));
            pend_tile = -1;
        }
        buf = (buf + 1 == RING_NBUF) ? 0 : buf + 1;
    }
    dbg_stamp(a.times, blockIdx.y * gridDim.x + blockIdx.x, 2);
    if (n_it > 0) {                                              // the last half's accumulator
#pragma unroll
        for (int w = 0; w < QW; ++w) {
            float sv = 0.f;
            const unsigned rest = screen_test<PASS>(accB[w], thq[w], thlo[w], kb_prev, cseg[w], a.splits * 2, n_loc[w], cap, gm[w], sv);
            if (PASS == 1) screen_rest(rest, sv, kb_prev, cseg[w], a.splits * 2, n_loc[w], cap);
        }
    }

#pragma unroll
    for (int w = 0; w < QW; ++w) {
        if (PASS == 0) {
            if (a.gkeep == 16) {
                // few segments per query (small maps, large batches): ALL sixteen group maxima of the lane go to the threshold kernel --
                // with four of them per segment a query had 8 * chunks values to take its k-th largest from: fewer than k = 50 on a
                // batch of 64 leaf tiles (theta = 0, every key a candidate, every query group on the fp32 redo pass: 12 ms a call)
                if (qvalid[w]) {
                    float4* o = reinterpret_cast<float4*>(a.gmax + seg[w] * 16);
#pragma unroll
                    for (int u = 0; u < 4; ++u) o[u] = make_float4(gm[w][4 * u], gm[w][4 * u + 1], gm[w][4 * u + 2], gm[w][4 * u + 3]);
                }
                continue;
            }
            // keep the lane's GKEEP largest group maxima: GKEEP distinct keys, so still a valid pool for the
            // k-th-largest lower bound, at a quarter of the traffic into the theta kernel
            float top[GKEEP];
#pragma unroll
            for (int u = 0; u < GKEEP; ++u) {
                float mx = gm[w][0];
#pragma unroll
                for (int r = 1; r < 16; ++r) mx = fmaxf(mx, gm[w][r]);
                top[u] = mx;
                bool taken = false;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const bool hit = !taken && (gm[w][r] == mx);
                    gm[w][r] = hit ? -1.0f : gm[w][r];
                    taken = taken || hit;
                }
            }
            if (qvalid[w]) *reinterpret_cast<float4*>(a.gmax + seg[w] * GKEEP) = make_float4(top[0], top[1], top[2], top[3]);
        } else {
            if (qvalid[w]) cseg[w][-(a.splits * 2)] = make_int2(n_loc[w], 0);
        }
    }
    dbg_stamp(a.times, blockIdx.y * gridDim.x + blockIdx.x, 3);
}
#endif  // DAGL_ABLATION

// theta of the adaptive modes: S~ >= theta  <=  (S~ (1+DELTA) - mean*thr) + bias > 0, with slack for the fp32
// rounding of either side (dagl.py:256 evaluates (S - mean*thr) + bias in fp32).
// `both` (adaptive AND top-k mask): theta already holds the top-k threshold; a key must pass both tests, so the larger wins.
__global__ void adaptive_theta_kernel(size_t n, const float* __restrict__ mt, const float* __restrict__ bs,
                                      float* __restrict__ theta, int both) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float ta = adaptive_theta_of(mt[i], bs[i]);
    theta[i] = both ? fmaxf(ta, theta[i]) : ta;
}

int launch_adaptive_theta(hipStream_t s, size_t n, const float* mt, const float* bs, float* theta, bool both) {
    hipLaunchKernelGGL(adaptive_theta_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, n, mt, bs, theta, both ? 1 : 0);
    DAGL_LAUNCH_CHECK("adaptive_theta_kernel");
    return DAGL_OK;
}

int launch_screen(hipStream_t s, const ScreenArgs& a, int pass) {
    const int qblock = (a.qblock == 512 || a.qblock == 384) ? a.qblock : 256;
    const int n_qgroups = (a.L + qblock - 1) / qblock;
    dim3 grid(n_qgroups * a.splits, a.B);
#ifdef DAGL_ABLATION      // debug builds only: the variants give wrong results by construction
#define SCR_LAUNCH(P_, Q_, V_) hipLaunchKernelGGL((screen_kernel<P_, Q_, V_>), grid, dim3(256 / Q_ * 2), 0, s, a, n_qgroups)
#define SCR_VARIANTS(P_, Q_)                                              \
    switch (a.variant) {                                                 \
        case 0: SCR_LAUNCH(P_, Q_, 0); break;                            \
        case 1: SCR_LAUNCH(P_, Q_, 1); break;                            \
        case 2: SCR_LAUNCH(P_, Q_, 2); break;                            \
        case 4: SCR_LAUNCH(P_, Q_, 4); break;                            \
        case 8: SCR_LAUNCH(P_, Q_, 8); break;                            \
        case 16: SCR_LAUNCH(P_, Q_, 16); break;                          \
        case 40: SCR_LAUNCH(P_, Q_, 40); break;                          \
        case 9: SCR_LAUNCH(P_, Q_, 9); break;                            \
        case 41: SCR_LAUNCH(P_, Q_, 41); break;                          \
        case 32: SCR_LAUNCH(P_, Q_, 32); break;                          \
        case 48: SCR_LAUNCH(P_, Q_, 48); break;                          \
        case 24: SCR_LAUNCH(P_, Q_, 24); break;                          \
        case 25: SCR_LAUNCH(P_, Q_, 25); break;                          \
        case 57: SCR_LAUNCH(P_, Q_, 57); break;                          \
        case 23: SCR_LAUNCH(P_, Q_, 23); break;                          \
        case 39: SCR_LAUNCH(P_, Q_, 39); break;                          \
        case 55: SCR_LAUNCH(P_, Q_, 55); break;                          \
        case 15: SCR_LAUNCH(P_, Q_, 15); break;                          \
        case 63: SCR_LAUNCH(P_, Q_, 63); break;                          \
        case 121: SCR_LAUNCH(P_, Q_, 121); break;                        \
        case 89: SCR_LAUNCH(P_, Q_, 89); break;                          \
        case 73: SCR_LAUNCH(P_, Q_, 73); break;                          \
        case 72: SCR_LAUNCH(P_, Q_, 72); break;                          \
        default: SCR_LAUNCH(P_, Q_, 7); break;                           \
    }
    if (pass == 1 && a.seg_max) {
        if (qblock == 512) hipLaunchKernelGGL((screen_ring_kernel<1, 16, 0, 1, true>), grid, dim3(1024), 0, s, a, n_qgroups);
        else if (qblock == 384) hipLaunchKernelGGL((screen_ring_kernel<1, 12, 0, 1, true>), grid, dim3(768), 0, s, a, n_qgroups);
        else hipLaunchKernelGGL((screen_kernel<1, 1, 0, 256, true>), grid, dim3(512), 0, s, a, n_qgroups);
        DAGL_LAUNCH_CHECK("screen_kernel");
        return DAGL_OK;
    }
    static const int qw = [] { const char* e = getenv("DAGL_SCREEN_QW"); return e ? atoi(e) : 1; }();
    const size_t n_blk = (size_t)grid.x * grid.y;
    const bool ring_phases = a.variant == 64 && qblock == 512;             // per-wave phase clocks: 16 waves x 5 sums per block
    ScreenArgs at = a; at.times = (getenv("DAGL_TIMES_FILE") && pass == 1) ? dbg_times_buffer(ring_phases ? n_blk * 20 : n_blk) : nullptr;
#define a at
    static const int ring_env = [] { const char* e = getenv("DAGL_SCREEN_RING"); return e ? atoi(e) : 1; }();
    static const int pipe_qw = [] { const char* e = getenv("DAGL_PIPE_QW"); return e ? atoi(e) : 2; }();
    if (qblock == 512 && ring_env == 3) {
        if (pass == 0) hipLaunchKernelGGL((screen_ring_kernel<0, 16>), grid, dim3(1024), 0, s, a, n_qgroups);
        else if (a.variant == 2) hipLaunchKernelGGL((screen_pipe_kernel<1, 8, 2, 2>), grid, dim3(512), 0, s, a, n_qgroups);
        else if (pipe_qw == 1) hipLaunchKernelGGL((screen_pipe_kernel<1, 16>), grid, dim3(1024), 0, s, a, n_qgroups);
        else hipLaunchKernelGGL((screen_pipe_kernel<1, 8, 0, 2>), grid, dim3(512), 0, s, a, n_qgroups);
    } else
    if (qblock == 512 && ring_env) {
#define SCR_R5(P_, V_) do { if (ring_env == 2) hipLaunchKernelGGL((screen_ring_kernel<P_, 8, V_, 2>), grid, dim3(512), 0, s, a, n_qgroups); \
                            else hipLaunchKernelGGL((screen_ring_kernel<P_, 16, V_>), grid, dim3(1024), 0, s, a, n_qgroups); } while (0)
#define SCR_RV(P_) switch (a.variant) { case 0: SCR_R5(P_, 0); break; case 1: SCR_R5(P_, 1); break; case 2: SCR_R5(P_, 2); break; case 3: SCR_R5(P_, 3); break; \
                                         case 7: SCR_R5(P_, 7); break; case 15: SCR_R5(P_, 15); break; case 31: SCR_R5(P_, 31); break; case 6: SCR_R5(P_, 6); break; \
                                         case 18: SCR_R5(P_, 18); break; case 8: SCR_R5(P_, 8); break; case 64: SCR_R5(P_, 64); break; \
                                         case 4: SCR_R5(P_, 4); break; case 16: SCR_R5(P_, 16); break; case 5: SCR_R5(P_, 5); break; case 20: SCR_R5(P_, 20); break; \
                                         case 17: SCR_R5(P_, 17); break; case 21: SCR_R5(P_, 21); break; \
                                         case 128: SCR_R5(P_, 128); break; case 159: SCR_R5(P_, 159); break; case 144: SCR_R5(P_, 144); break; case 130: SCR_R5(P_, 130); break; \
                                         case 145: SCR_R5(P_, 145); break; case 149: SCR_R5(P_, 149); break; default: SCR_R5(P_, 0); break; }
        if (pass == 0) { SCR_R5(0, 0); } else { SCR_RV(1) }
    } else
    if (qblock == 512) {                                                  // 512-query blocks: a few variants only
#define SCR_L5(P_, V_) hipLaunchKernelGGL((screen_kernel<P_, 1, V_, 512>), grid, dim3(1024), 0, s, a, n_qgroups)
#define SCR_V5(P_) switch (a.variant) { case 0: SCR_L5(P_, 0); break; case 8: SCR_L5(P_, 8); break; case 24: SCR_L5(P_, 24); break; \
                                         case 25: SCR_L5(P_, 25); break; case 9: SCR_L5(P_, 9); break; default: SCR_L5(P_, 57); break; }
        if (pass == 0) { SCR_V5(0) } else { SCR_V5(1) }
    } else if (qblock == 384) {                                           // (no variants)
        if (pass == 0) hipLaunchKernelGGL((screen_kernel<0, 1, 0, 384>), grid, dim3(768), 0, s, a, n_qgroups);
        else hipLaunchKernelGGL((screen_kernel<1, 1, 0, 384>), grid, dim3(768), 0, s, a, n_qgroups);
    } else
    if (qw == 2) { if (pass == 0) { SCR_VARIANTS(0, 2) } else { SCR_VARIANTS(1, 2) } }
    else { if (pass == 0) { SCR_VARIANTS(0, 1) } else { SCR_VARIANTS(1, 1) } }
#undef a
    if (at.times) dbg_times_dump(s, ring_phases ? "screen_ring_kernel<1>_phases" : (qblock == 512 && ring_env ? (ring_env == 3 ? "screen_pipe_kernel<1>" : "screen_ring_kernel<1>") : "screen_kernel<1>"), at.times, ring_phases ? n_blk * 20 : n_blk);
#else
    // blocks that are alone on their CU (512 / 384 queries): the ring form; 256-query blocks (two per CU, small images) keep
    // the barrier form -- two five-deep rings do not fit one CU's LDS
    if (pass == 1 && a.seg_max) {                    // (the dense formulation's top-1 screen: segment headers carry their maxima)
        if (qblock == 512) hipLaunchKernelGGL((screen_ring_kernel<1, 16, 0, 1, true>), grid, dim3(1024), 0, s, a, n_qgroups);
        else if (qblock == 384) hipLaunchKernelGGL((screen_ring_kernel<1, 12, 0, 1, true>), grid, dim3(768), 0, s, a, n_qgroups);
        else hipLaunchKernelGGL((screen_kernel<1, 1, 0, 256, true>), grid, dim3(512), 0, s, a, n_qgroups);
    } else
    if (qblock == 512) {
        if (pass == 0) hipLaunchKernelGGL((screen_ring_kernel<0, 16>), grid, dim3(1024), 0, s, a, n_qgroups);
        else hipLaunchKernelGGL((screen_ring_kernel<1, 16>), grid, dim3(1024), 0, s, a, n_qgroups);
    } else if (qblock == 384) {
        if (pass == 0) hipLaunchKernelGGL((screen_ring_kernel<0, 12>), grid, dim3(768), 0, s, a, n_qgroups);
        else hipLaunchKernelGGL((screen_ring_kernel<1, 12>), grid, dim3(768), 0, s, a, n_qgroups);
    } else {
        if (pass == 0) hipLaunchKernelGGL((screen_kernel<0, 1, 0>), grid, dim3(512), 0, s, a, n_qgroups);
        else hipLaunchKernelGGL((screen_kernel<1, 1, 0>), grid, dim3(512), 0, s, a, n_qgroups);
    }
#endif
    DAGL_LAUNCH_CHECK("screen_kernel");
    return DAGL_OK;
}

// ---- theta: k-th largest group maximum per query (one wave per query, values in registers) -------------------
constexpr int THETA_PER_LANE = 8;                // G <= 512
__global__ __launch_bounds__(256) void screen_theta_kernel(int n_rows, int G, int k, const float* __restrict__ gmax,
                                                           float* __restrict__ theta, const int32_t* gate, unsigned* __restrict__ spill_cnt) {
    if (gate != nullptr && *gate == 0) return;
    const int lane = threadIdx.x & 63;
    const size_t row = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (size_t)n_rows) return;
    if (spill_cnt != nullptr && lane == 0) spill_cnt[row] = 0u;          // (the filter pass behind this launch counts from zero)
    float v[THETA_PER_LANE];
#pragma unroll
    for (int u = 0; u < THETA_PER_LANE; ++u) {
        const int t = lane + 64 * u;
        v[u] = (t < G) ? gmax[row * G + t] : -1.0f;
    }
    float kth = -1.0f;
    for (int r = 0; r < k; ++r) {
        float lm = v[0];
#pragma unroll
        for (int u = 1; u < THETA_PER_LANE; ++u) lm = fmaxf(lm, v[u]);
        const float wm = wave_max_f32(lm);
        kth = wm;
        if (wm < 0.f) break;                                   // fewer than k non-empty groups (wave-uniform)
        // the first lane holding the maximum drops one copy of it
        const unsigned long long bal = __ballot(lm == wm);
        const int owner = __ffsll((long long)bal) - 1;
        bool taken = false;
#pragma unroll
        for (int u = 0; u < THETA_PER_LANE; ++u) {
            const bool hit = (lane == owner) && !taken && (v[u] == wm);
            v[u] = hit ? -1.0f : v[u];
            taken = taken || hit;
        }
    }
    // fewer than k non-empty groups -> no usable bound: pass everything (theta = 0)
    if (lane == 0) theta[row] = (kth > 0.f) ? kth * ((1.0f - DELTA) / (1.0f + DELTA)) : 0.0f;
}

int launch_screen_theta(hipStream_t s, int n_rows, int G, int k, const float* gmax, float* theta, const int32_t* gate, unsigned* spill_cnt) {
    if (G > 64 * THETA_PER_LANE) { set_error("screen_theta: %d groups exceed %d", G, 64 * THETA_PER_LANE); return DAGL_ERR_INVALID; }
    hipLaunchKernelGGL(screen_theta_kernel, dim3((n_rows + 3) / 4), dim3(256), 0, s, n_rows, G, k, gmax, theta, gate, spill_cnt);
    DAGL_LAUNCH_CHECK("screen_theta_kernel");
    return DAGL_OK;
}

// ---- refine: exact scores of the candidates, final selection, edge softmax ---------------------------------------
// One wave per query.  Candidates are gathered segment by segment (deterministic order: chunk, half, ascending
// key), their scores recomputed from the fp32 feature rows with fp64 accumulation, then
//   top-k modes : the k best (score desc, key asc)             (GReccR2b_3mh_1-checkpoint.py:242-246)
//   adaptive    : those with (S - mean*thr) + bias > 0          (dagl.py:256-257)
// and the softmax weights of dagl.py:259-261 are formed exactly as in aggregate.hip.
constexpr int RF_MAX_CAND = 1024;
constexpr int RF_ROUNDS_MAX_K = 16;                 // up to this k the candidate threshold is tightened by k rounds of wave maxima, beyond by a bit-wise search

__device__ __forceinline__ float rf_logit(float s, float mtq, float bsq, bool adaptive) {
    float m = 1.0f;
    if (adaptive) m = (s - mtq) + bsq;
    return __fmul_rn(__fmul_rn(s, m), SOFTMAX_SCALE);
}

// (value desc, key asc) ordering used everywhere a "k best" is taken
__device__ __forceinline__ bool rf_before(float va, int ka, float vb, int kb) {
    return (va > vb) || (va == vb && ka < kb);
}

// One query.  HEAVY = false: a wave on its own (no block-level sync).  HEAVY = true: the RF_HEAVY_WAVES waves of a block on
// ONE query of the adaptive mode with many candidates -- wave 0 collects the candidates and does everything after the exact
// scores, the gathers of the feature rows (what a long candidate list costs: 512 candidates are 32 dependent rounds for a
// single wave) are shared out.  A candidate's exact score does not depend on the wave that forms it: same results.
constexpr int RF_HEAVY_WAVES = 8;
constexpr int RF_HEAVY_MIN = 64;                           // candidates from which a query of the adaptive mode is handed over
constexpr int RF_HEAVY_CAP = 1024;                         // listed queries (a full list: the wave keeps its query)
template <bool HEAVY>
__device__ __forceinline__ void refine_query(const RefineArgs& a, const size_t ql, int* __restrict__ ci, float* __restrict__ cv,
                                             int* sh_total) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const bool lead = !HEAVY || w == 0;
    const int b = (int)(ql / a.L);
    const int S2 = a.splits * 2;
    bool overflow = false;
    // this query's features are needed only in step 2, but nothing stops the loads from flying during step 1
    const int grp = lane >> 3, gl = lane & 7;
    const float* qrow = a.wq + ((size_t)b * a.rows_q + (ql - (size_t)b * a.L)) * DS;
    float4 qv[7];
#pragma unroll
    for (int u = 0; u < 7; ++u) {
        const int c4 = gl + 8 * u;                         // unconditional (clamped) loads, zeroed afterwards: a predicated
        const float4 raw = *reinterpret_cast<const float4*>(qrow + 4 * (c4 < D / 4 ? c4 : D / 4 - 1));   // load serialises
        qv[u] = (c4 < D / 4) ? raw : make_float4(0.f, 0.f, 0.f, 0.f);
    }

    // per-query scalars of the later steps, requested up front as well
    const bool adaptive = (a.mode != DAGL_MODE_TOPK);
    const float thq = (a.mode != DAGL_MODE_ADAPTIVE) ? a.theta[ql] : 0.f;
    const float mtq = adaptive ? a.mt[ql] : 0.f, bsq = adaptive ? a.bs[ql] : 0.f;

    // 1. gather candidate indices: lane <-> segment; the first three candidates of a segment are fetched together with
    //    its count (one memory round trip), longer segments are rare and finished in a loop
    int total = 0;
    if (lead)
    for (int s0 = 0; s0 < S2; s0 += 64) {
        const int sgi = s0 + lane;
        const bool sv = sgi < S2;
        // records are slot-major inside a query -- [slot][segment] -- so that the lanes' reads of one slot are one contiguous run
        // (8 bytes per lane); the count and the first three candidates are asked for together (one memory round trip)
        const int2* rec = a.cand + ql * (size_t)a.capseg * S2 + (sv ? sgi : 0);      // slot 0 of this lane's segment
        const int2 r0 = rec[0], c0 = rec[S2], c1 = rec[2 * (size_t)S2], c2 = rec[3 * (size_t)S2];
        int cnt = sv ? r0.x : 0;
        if (cnt > a.capseg - 1) { overflow = true; cnt = a.capseg - 1; }
        const int incl = wave_scan_incl_i32(cnt);
        const int off = total + incl - cnt;
        if (off + cnt > RF_MAX_CAND) { overflow = true; cnt = max(0, RF_MAX_CAND - off); }
        if (cnt > 0) { ci[off] = c0.x; cv[off] = __int_as_float(c0.y); }
        if (cnt > 1) { ci[off + 1] = c1.x; cv[off + 1] = __int_as_float(c1.y); }
        if (cnt > 2) { ci[off + 2] = c2.x; cv[off + 2] = __int_as_float(c2.y); }
        for (int e = 3; e < cnt; ++e) {
            const int2 c = rec[(size_t)(1 + e) * S2];
            ci[off + e] = c.x; cv[off + e] = __int_as_float(c.y);
        }
        total += __builtin_amdgcn_readlane(incl, 63);
    }
    overflow = __any(overflow);
    if (lead && a.spill != nullptr && overflow) {                             // (records are spilled by full segments only)
        // the records that did not fit their segments (ScreenArgs::spill): as long as the query's shared area and the candidate
        // arrays hold them all, nothing was dropped -- no redo.  Their order is whatever the atomics made it; everything behind
        // this point works on the SET (selections by (score, key), sums in rank order).
        const unsigned sc_n = a.spill_cnt[ql];
        if (sc_n > 0u) {
            if (sc_n <= (unsigned)SCREEN_SPILL && total + (int)sc_n <= RF_MAX_CAND) {
                const bool seg_only = !(total >= RF_MAX_CAND);               // (the only overflow so far was full segments)
                for (unsigned e = lane; e < sc_n; e += 64) {
                    const int2 c = a.spill[ql * SCREEN_SPILL + e];
                    ci[total + e] = c.x; cv[total + e] = __int_as_float(c.y);
                }
                total += (int)sc_n;
                if (seg_only) overflow = false;
            } else overflow = true;
        }
    }
    if (total > RF_MAX_CAND) total = RF_MAX_CAND;
    // adaptive lists with the per-query redo behind them: twice as many candidates as the list is wide will not fit it
    // (nearly every candidate of this mode passes the exact test) -- the row is redone on its own anyway (overflow.hip),
    // so its candidates are not rescored here (a thousand candidates = 128 rounds of row gathers by one wave)
    if (a.mode == DAGL_MODE_ADAPTIVE && a.ovf_list != nullptr && total > 2 * a.width) { overflow = true; total = 0; }
    __threadfence_block();
    if (HEAVY) {
        if (w == 0 && lane == 0) *sh_total = total;
        __syncthreads();
        total = *sh_total;
    } else if (a.mode == DAGL_MODE_ADAPTIVE && a.heavy_list != nullptr && total >= RF_HEAVY_MIN) {
        int pos = RF_HEAVY_CAP;
        if (lane == 0) { pos = atomicAdd(a.heavy_count, 1); if (pos < RF_HEAVY_CAP) a.heavy_list[pos] = (int32_t)ql; }
        if (__shfl(pos, 0) < RF_HEAVY_CAP) return;          // refine_heavy_kernel takes it from here
    }

    // 1b. top-k modes: most candidates owe their place to the loose threshold of the sampling pass.  With the screened
    //     scores at hand the wave tightens it before any feature row is fetched (that gather is what this kernel costs):
    //     k candidates have S~ >= t (t = k-th largest lower bound), so the k-th largest true score is >= t/(1+DELTA), and a
    //     candidate whose upper bound is below t (1-DELTA)/(1+DELTA) cannot be among the k best.
    //     t by k rounds of "take the largest lower bound away" (a wave maximum each); a lane owns candidates lane + 64 u.
    //     Up to 256 candidates (the usual case) the lane's four slots stay in registers, beyond that they are re-read from LDS.
    constexpr int TU = 4;
    if (a.mode != DAGL_MODE_ADAPTIVE && total > a.k && total <= 64 * TU) {
        int key[TU]; float ub[TU], work[TU];
#pragma unroll
        for (int u = 0; u < TU; ++u) {
            const int c = lane + 64 * u;
            const bool have = c < total;
            key[u] = have ? ci[c] : -1;
            const float sv = have ? cv[c] : 0.f;
            ub[u] = fabsf(sv);
            work[u] = have ? ((sv >= 0.f) ? sv : thq) : -1.0f;               // exact screened score, or only "passed theta"
        }
        float tk = -1.0f;
        if (!HEAVY && a.k > RF_ROUNDS_MAX_K) {
            // larger k: the k-th largest by 31 steps of a bit-wise search over the values' patterns (lower bounds are >= 0: their bits
            // order like the values; "not a candidate" = 0) instead of k rounds of "take the largest away" -- the same number, a third
            // of the instructions at k = 50 (the refine pass is instruction-bound there: 193 us of a 0.55 ms call on natural features)
            unsigned uw[TU];
#pragma unroll
            for (int u = 0; u < TU; ++u) uw[u] = work[u] > 0.f ? __float_as_uint(work[u]) : 0u;
            unsigned prefix = 0u;
            for (int bit = 30; bit >= 0; --bit) {
                const unsigned t = prefix | (1u << bit);
                int cnt = 0;
#pragma unroll
                for (int u = 0; u < TU; ++u) cnt += uw[u] >= t ? 1 : 0;
                if (wave_sum_i32(cnt) >= a.k) prefix = t;
            }
            tk = __uint_as_float(prefix);
        } else
        for (int r = 0; r < a.k; ++r) {
            float lm = work[0];
#pragma unroll
            for (int u = 1; u < TU; ++u) lm = fmaxf(lm, work[u]);
            const float wm = wave_max_f32(lm);
            tk = wm;
            const unsigned long long bal = __ballot(lm == wm);              // the first lane holding it drops one copy
            const int owner = __ffsll((long long)bal) - 1;
            bool taken = false;
#pragma unroll
            for (int u = 0; u < TU; ++u) {
                const bool hit = (lane == owner) && !taken && (work[u] == wm);
                work[u] = hit ? -2.0f : work[u];
                taken = taken || hit;
            }
        }
        const float cut = tk * ((1.0f - DELTA) / (1.0f + DELTA)) * (1.0f - 1e-6f);
        int base = 0;
#pragma unroll
        for (int u = 0; u < TU; ++u) {                                       // (every slot was read above)
            const bool keep = key[u] >= 0 && ub[u] >= cut;
            const unsigned long long bal = __ballot(keep);
            const int pos = base + __popcll(bal & ((1ull << lane) - 1ull));
            if (keep) ci[pos] = key[u];
            base += __popcll(bal);
        }
        total = base;
        __threadfence_block();
    } else if (a.mode != DAGL_MODE_ADAPTIVE && total > a.k) {
        const int nu = (total + 63) >> 6;                                    // <= RF_MAX_CAND / 64 = 16
        unsigned taken = 0u;                                                 // this lane's slots already taken away
        float tk = -1.0f;
        if (!HEAVY && a.k > RF_ROUNDS_MAX_K) {                               // (as above; the lane's up to 16 lower bounds in registers)
            unsigned uw[RF_MAX_CAND / 64];
#pragma unroll
            for (int u = 0; u < RF_MAX_CAND / 64; ++u) {
                const int c = lane + 64 * u;
                float lbv = 0.f;
                if (u < nu && c < total) { const float sv = cv[c]; lbv = (sv >= 0.f) ? sv : thq; }
                uw[u] = lbv > 0.f ? __float_as_uint(lbv) : 0u;
            }
            unsigned prefix = 0u;
            for (int bit = 30; bit >= 0; --bit) {
                const unsigned t = prefix | (1u << bit);
                int cnt = 0;
#pragma unroll
                for (int u = 0; u < RF_MAX_CAND / 64; ++u) cnt += uw[u] >= t ? 1 : 0;
                if (wave_sum_i32(cnt) >= a.k) prefix = t;
            }
            tk = __uint_as_float(prefix);
        } else
        for (int r = 0; r < a.k; ++r) {
            float lm = -1.0f; int lu = -1;
            for (int u = 0; u < nu; ++u) {
                const int c = lane + 64 * u;
                if (c < total && !((taken >> u) & 1u)) {
                    const float sv = cv[c];
                    const float lbv = (sv >= 0.f) ? sv : thq;
                    if (lbv > lm) { lm = lbv; lu = u; }
                }
            }
            const float wm = wave_max_f32(lm);
            tk = wm;
            const unsigned long long bal = __ballot(lu >= 0 && lm == wm);
            if (bal != 0ull && lane == __ffsll((long long)bal) - 1) taken |= 1u << lu;
        }
        const float cut = tk * ((1.0f - DELTA) / (1.0f + DELTA)) * (1.0f - 1e-6f);
        int base = 0;
        for (int u = 0; u < nu; ++u) {
            const int c = lane + 64 * u;
            const bool have = c < total;
            const int key = have ? ci[c] : -1;
            const float ub = have ? fabsf(cv[c]) : 0.f;
            const bool keep = key >= 0 && ub >= cut;
            const unsigned long long bal = __ballot(keep);
            const int pos = base + __popcll(bal & ((1ull << lane) - 1ull));
            if (keep) ci[pos] = key;                                   // pos <= c: only slots that have been read
            base += __popcll(bal);
        }
        total = base;
        __threadfence_block();
    }

    if (!HEAVY) dbg_stamp(a.times, blockIdx.x, 1);
    // 2. exact scores: 8 groups of 8 lanes, one candidate per group per round, fp64 accumulation
    const float* xb = a.x + (size_t)b * a.rows_x * DS;
#pragma unroll 2
    for (int c0 = HEAVY ? 8 * w : 0; c0 < total; c0 += HEAVY ? 8 * RF_HEAVY_WAVES : 8) {
        const int c = c0 + grp;
        const bool okc = c < total;
        int key = okc ? ci[c] : 0;
        const bool inb = key < a.N;                              // zero rows past N may pass a degenerate theta
        if (!inb) key = 0;
        const float* xrow = xb + (size_t)key * DS;
        float4 xv[7];
#pragma unroll
        for (int u = 0; u < 7; ++u) {
            const int c4 = gl + 8 * u;                       // c4 <= 55: inside the 51-float4 row stride + next row
            xv[u] = *reinterpret_cast<const float4*>(xrow + 4 * (c4 < D / 4 ? c4 : 0));
        }
        double acc = 0.0;
#pragma unroll
        for (int u = 0; u < 7; ++u) {
            acc += (double)qv[u].x * (double)xv[u].x + (double)qv[u].y * (double)xv[u].y +
                   (double)qv[u].z * (double)xv[u].z + (double)qv[u].w * (double)xv[u].w;
        }
        acc += dpp_zero_d<0xB1>(acc); acc += dpp_zero_d<0x4E>(acc); acc += dpp_zero_d<0x141>(acc);     // pairs, quads, the group's other quad
        if (okc && gl == 0) {
            cv[c] = inb ? (float)acc : -4.0f;
            if (!inb) ci[c] = -1;                          // never selected
        }
    }
    __threadfence_block();
    if (HEAVY) {
        __syncthreads();
        if (!lead) return;
    }
    if (!HEAVY) dbg_stamp(a.times, blockIdx.x, 2);

    int n = 0;
    constexpr int RU = DAGL_LIST_CAP / 64;                 // list entries per lane: entry e lives in lane e % 64, slot e / 64
    float my_s[RU]; int my_key[RU];
    int pos0 = lane;                                       // list position of slot 0 (-1: none); differs from the lane only
                                                           // after the rank-counting selection below
#pragma unroll
    for (int u = 0; u < RU; ++u) { my_s[u] = 0.f; my_key[u] = -1; }

    if (a.mode == DAGL_MODE_ADAPTIVE) {
        // keep order, drop candidates failing the exact test; survivor j goes to list entry j (<= width <= DAGL_LIST_CAP)
        for (int c0 = 0; c0 < total; c0 += 64) {
            const int c = c0 + lane;
            float s = 0.f; int key = -1; bool pass = false;
            if (c < total) {
                s = cv[c]; key = ci[c];
                pass = (key >= 0) && (((s - mtq) + bsq) > 0.f);
            }
            const unsigned long long bal = __ballot(pass);
            const int pos = n + __popcll(bal & ((1ull << lane) - 1ull));
            if (pass) {
                if (pos < a.width) { cv[pos] = s; ci[pos] = key; }   // pos <= c: in-place compaction is safe
                else overflow = true;
            }
            n += __popcll(bal);
            __threadfence_block();
        }
        overflow = __any(overflow);
        if (n > a.width) n = a.width;
        // a query that goes to the per-query redo (overflow.hip) gets its row, degree and softmax mass from there: an empty
        // list here, or the gather walks 256 neighbours whose sum is thrown away
        if (overflow && a.ovf_list != nullptr) n = 0;
#pragma unroll
        for (int u = 0; u < RU; ++u) {
            const int e = lane + 64 * u;
            if (e < n) { my_s[u] = cv[e]; my_key[u] = ci[e]; }
        }
    } else if (total <= 64) {
        // top-k of <= 64 candidates by rank counting: (value desc, key asc) is a strict order on the valid candidates, so the
        // number of candidates ahead of a lane's own is its list position.  `total` uniform lane reads, no shuffle chain.
        float v = -3.0f; int key = 0x7fffffff;
        if (lane < total) {
            v = cv[lane]; key = ci[lane];
            if (key < 0 || (a.mode == DAGL_MODE_ADAPTIVE_TOPK && !(((v - mtq) + bsq) > 0.f))) { v = -3.0f; key = 0x7fffffff; }
        }
        int rank = 0;
        for (int j = 0; j < total; ++j) {
            const float vj = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), j)); const int kj = __builtin_amdgcn_readlane(key, j);
            rank += rf_before(vj, kj, v, key) ? 1 : 0;
        }
        const unsigned long long valid_mask = __ballot(v > -2.0f);
        n = min(a.k, (int)__popcll(valid_mask));
        if (v > -2.0f && rank < n) { my_s[0] = v; my_key[0] = key; pos0 = rank; }
        else pos0 = -1;
    } else {
        if (a.mode == DAGL_MODE_ADAPTIVE_TOPK) {
            for (int c = lane; c < total; c += 64)
                if (!(((cv[c] - mtq) + bsq) > 0.f)) ci[c] = -1;
            __threadfence_block();
        }
        for (int r = 0; r < a.k; ++r) {
            float bv = -2.f; int bi = 0x7fffffff, bpos = -1;
            for (int t = lane; t < total; t += 64) {
                const float v = cv[t]; const int id = ci[t];
                if (id >= 0 && (v > bv || (v == bv && id < bi))) { bv = v; bi = id; bpos = t; }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const float ov = __shfl_xor(bv, o); const int oi = __shfl_xor(bi, o); const int op = __shfl_xor(bpos, o);
                if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; bpos = op; }
            }
            if (bpos < 0) break;
            if (lane == r) { my_s[0] = bv; my_key[0] = bi; }
            if (lane == 0) ci[bpos] = -1;
            __threadfence_block();
            ++n;
        }
    }

    // 3. edge softmax over the kept neighbours (non-neighbours contribute exp(0) each to the denominator)
    float lg[RU];
    double M = -1e300;
#pragma unroll
    for (int u = 0; u < RU; ++u) {
        const int e = (u == 0) ? pos0 : lane + 64 * u;
        const bool valid = e >= 0 && e < n;
        lg[u] = valid ? rf_logit(my_s[u], mtq, bsq, adaptive) : 0.f;
        if (valid) M = fmax(M, (double)lg[u]);
    }
    M = wave_max_f64(M);
    if (n < a.N) M = fmax(M, 0.0);
    double ev[RU], sum = 0.0;
#pragma unroll
    for (int u = 0; u < RU; ++u) {
        const int e = (u == 0) ? pos0 : lane + 64 * u;
        ev[u] = (e >= 0 && e < n) ? exp((double)lg[u] - M) : 0.0;        // (slots past the first are empty outside the long-tail case)
        sum += ev[u];
    }
    sum = wave_sum_f64(sum);
    sum += (double)(a.N - n) * exp(-M);
#pragma unroll
    for (int u = 0; u < RU; ++u) {
        const int e = (u == 0) ? pos0 : lane + 64 * u;
        if (e >= 0 && e < n) {
            a.nb_idx[ql * a.width + e] = my_key[u];
            a.nb_wgt[ql * a.width + e] = (float)(ev[u] / sum);
            if (a.nb_s != nullptr) a.nb_s[ql * a.width + e] = my_s[u];
        }
    }
    int ovf_pos = -1;
    if (lane == 0) {
        // (a row that goes to the per-query redo: count -1 = "not from the list": the gather leaves its aggregated row alone)
        a.nb_cnt[ql] = (overflow && a.ovf_list != nullptr) ? -1 : n;
        if (overflow) {
            const size_t qg = (size_t)b * a.n_qgroups_exact + (size_t)(ql - (size_t)b * a.L) / 128;
            a.redo_flags[qg] = 1;
            atomicAdd(reinterpret_cast<unsigned long long*>(&a.stats[2]), 1ull);
            if (a.ovf_list != nullptr) {                  // adaptive: this query is redone on its own (overflow.hip)
                ovf_pos = atomicAdd(a.ovf_count, 1);
                if (ovf_pos < a.ovf_cap) a.ovf_list[ovf_pos] = (int32_t)ql;
            }
        }
    }
    if (overflow && a.ovf_list != nullptr && a.ovf_qrows != nullptr) {      // wave-uniform: its feature row, compacted for the redo's product
        ovf_pos = __shfl(ovf_pos, 0);
        if (ovf_pos < a.ovf_cap)
            for (int c = lane; c < DS; c += 64) a.ovf_qrows[(size_t)ovf_pos * DS + c] = qrow[c];
    }
}

__global__ __launch_bounds__(256) void refine_kernel(RefineArgs a) {
    __shared__ int c_idx[4][RF_MAX_CAND];
    __shared__ float c_val[4][RF_MAX_CAND];
    if (a.gate != nullptr && *a.gate == 0) return;
    if (a.policy != nullptr && *a.policy != 0) a.capseg = a.capseg_tight;
    const int w = threadIdx.x >> 6;
    dbg_stamp(a.times, blockIdx.x, 0);
    const size_t ql = (size_t)blockIdx.x * 4 + w;
    if (ql >= (size_t)a.B * a.L) return;                   // no block-level sync below
    refine_query<false>(a, ql, c_idx[w], c_val[w], nullptr);
    dbg_stamp(a.times, blockIdx.x, 3);
}

__global__ __launch_bounds__(64 * RF_HEAVY_WAVES) void refine_heavy_kernel(RefineArgs a) {
    __shared__ int c_idx[RF_MAX_CAND];
    __shared__ float c_val[RF_MAX_CAND];
    __shared__ int sh_total;
    const int n = min(*a.heavy_count, RF_HEAVY_CAP);
    for (int slot = blockIdx.x; slot < n; slot += gridDim.x) {
        refine_query<true>(a, (size_t)a.heavy_list[slot], c_idx, c_val, &sh_total);
        __syncthreads();                                     // wave 0 is done with the candidate arrays
    }
}

// ---- dense formulation: the softmax shift of every row (dense_attend_kernel), tight -------------------------------------------
// The streamed dense formulation needs each row's shift before its pass over the keys, and its weights e^(l - shift) travel as fp16
// pairs: the shift may exceed the row's largest logit by ~18 units at most (dense.hip DN_SHIFT_SLACK).  Up to round 4 it was an
// upper bound from a full bf16 scan -- within 2 x 0.8 % of the largest SCORE, i.e. ~3 % of a logit that is quadratic in S: fine for
// logits of tens (trained features), beyond ~580 the rows' weights sank towards the fp16 denormals and whole 64-query blocks were
// run a second time (every block of bench.py's own default map: 1.52 ms for the advertised 0.81).
// Now: a top-1 screen -- sampled pass -> theta = the largest sampled S~ less the band (screen_theta_kernel, k = 1) -> filter pass
// with ScreenArgs::seg_max -- gives, per row, the largest screened score of ALL keys (every key above theta is seen by the filter,
// recorded or not, and the row maximum is above theta) and the candidates near it.  One wave per query:
//   * bounds  S~max / (1 + DELTA) <= max S <= S~max / (1 - DELTA)  ->  logit bounds; when they lie within DN_BOUND_OK units the upper
//     bound IS the shift (nothing to rescore: natural-image rows have thousands of keys inside the band and logits of tens);
//   * otherwise the candidates whose upper bound reaches the best lower bound x (1 - DELTA) / (1 + DELTA) (the true arg-max is among
//     them: DESIGN.md "The bf16 screen is conservative") are rescored from the fp32 features with fp64 accumulation: the exact maximum;
//   * a row with large logits whose candidates did not fit (flat maps) keeps the upper bound -- the first pass records its exact
//     maximum and dense_combine_kernel flags its block for the second pass: the round-4 mechanism, now the fallback of the fallback.
// Output: smax[q] = a score whose logit (dense.hip dn_shift) is the row's shift.
constexpr float DN_BOUND_OK = 12.0f;
__device__ __forceinline__ float rm_logit(float sc, float mtq, float bsq) {
    const float m = (sc - mtq) + bsq;                     // dagl.py:256, the expression order of dense.hip dn_logit
    return m > 0.f ? __fmul_rn(__fmul_rn(sc, m), SOFTMAX_SCALE) : 0.f;
}
__global__ __launch_bounds__(256) void rowmax_exact_kernel(RefineArgs a, float* __restrict__ smax) {
    __shared__ int c_idx[4][RF_MAX_CAND];
    __shared__ float c_val[4][RF_MAX_CAND];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const size_t ql = (size_t)blockIdx.x * 4 + w;
    if (ql >= (size_t)a.B * a.L) return;                   // no block-level sync below
    int* ci = c_idx[w]; float* cv = c_val[w];
    const int b = (int)(ql / a.L);
    const int S2 = a.splits * 2;
    const float thq = theta_of_sampled_max(a.theta[ql]);      // (a.theta: the raw sampled maximum, as the filter pass read it)
    // 1. segment headers: candidate counts and the segments' largest screened scores
    float sm = 0.f;
    bool overflow = false;
    int total = 0;
    for (int s0 = 0; s0 < S2; s0 += 64) {
        const int sgi = s0 + lane;
        const int2 hdr = (sgi < S2) ? a.cand[ql * (size_t)a.capseg * S2 + sgi] : make_int2(0, 0);
        if (hdr.x > 0) sm = fmaxf(sm, __int_as_float(hdr.y));
        if (hdr.x > a.capseg - 1) overflow = true;          // (the filter ran without a spill area: further records are lost)
        total += wave_sum_i32(min(hdr.x, a.capseg - 1));
    }
    sm = wave_max_f32(sm);
    overflow = __any(overflow) || total > RF_MAX_CAND;
    const float s_hi = sm * (1.0f / (1.0f - DELTA)) * (1.0f + 1e-6f), s_lo = sm * (1.0f / (1.0f + DELTA));
    const float mtq = a.mt[ql], bsq = a.bs[ql];
    if (overflow || rm_logit(s_hi, mtq, bsq) - rm_logit(s_lo, mtq, bsq) <= DN_BOUND_OK) {      // (wave-uniform)
        if (lane == 0) smax[ql] = s_hi;
        return;
    }
    // 2. the candidates, segment by segment (lane <-> segment)
    const int grp = lane >> 3, gl = lane & 7;
    const float* qrow = a.wq + ((size_t)b * a.rows_q + (ql - (size_t)b * a.L)) * DS;
    float4 qv[7];
#pragma unroll
    for (int u = 0; u < 7; ++u) {
        const int c4 = gl + 8 * u;
        const float4 raw = *reinterpret_cast<const float4*>(qrow + 4 * (c4 < D / 4 ? c4 : D / 4 - 1));
        qv[u] = (c4 < D / 4) ? raw : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    total = 0;
    for (int s0 = 0; s0 < S2; s0 += 64) {
        const int sgi = s0 + lane;
        const bool sv = sgi < S2;
        const int2* rec = a.cand + ql * (size_t)a.capseg * S2 + (sv ? sgi : 0);
        const int cnt = sv ? rec[0].x : 0;
        const int incl = wave_scan_incl_i32(cnt);
        const int off = total + incl - cnt;
        for (int e = 0; e < cnt; ++e) {
            const int2 c = rec[(size_t)(1 + e) * S2];
            ci[off + e] = c.x; cv[off + e] = __int_as_float(c.y);
        }
        total += __builtin_amdgcn_readlane(incl, 63);
    }
    __threadfence_block();
    // 3. the best lower bound among the candidates; those whose upper bound cannot reach it are dropped before any row is fetched
    float lb = -1.0f;
    for (int c = lane; c < total; c += 64) { const float sv = cv[c]; lb = fmaxf(lb, (sv >= 0.f) ? sv : thq); }
    lb = wave_max_f32(lb);
    const float cut = lb * ((1.0f - DELTA) / (1.0f + DELTA)) * (1.0f - 1e-6f);
    int base = 0;
    for (int c0 = 0; c0 < total; c0 += 64) {
        const int c = c0 + lane;
        const bool have = c < total;
        const int key = have ? ci[c] : -1;
        const bool keep = have && fabsf(cv[c]) >= cut && key < a.N;          // (zero rows past N may pass a degenerate theta)
        const unsigned long long bal = __ballot(keep);
        const int pos = base + __popcll(bal & ((1ull << lane) - 1ull));
        if (keep) ci[pos] = key;                                             // pos <= c: only slots that have been read
        base += __popcll(bal);
    }
    total = base;
    __threadfence_block();
    // 4. exact scores: 8 groups of 8 lanes, one candidate per group per round, fp64 accumulation; the largest
    const float* xb = a.x + (size_t)b * a.rows_x * DS;
    double best = 0.0;                                     // (scores are >= 0: post-ReLU features)
#pragma unroll 2
    for (int c0 = 0; c0 < total; c0 += 8) {
        const int c = c0 + grp;
        const bool okc = c < total;
        const float* xrow = xb + (size_t)(okc ? ci[c] : 0) * DS;
        float4 xv[7];
#pragma unroll
        for (int u = 0; u < 7; ++u) {
            const int c4 = gl + 8 * u;
            xv[u] = *reinterpret_cast<const float4*>(xrow + 4 * (c4 < D / 4 ? c4 : 0));
        }
        double acc = 0.0;
#pragma unroll
        for (int u = 0; u < 7; ++u)
            acc += (double)qv[u].x * (double)xv[u].x + (double)qv[u].y * (double)xv[u].y +
                   (double)qv[u].z * (double)xv[u].z + (double)qv[u].w * (double)xv[u].w;
        acc += dpp_zero_d<0xB1>(acc); acc += dpp_zero_d<0x4E>(acc); acc += dpp_zero_d<0x141>(acc);
        if (okc && gl == 0) best = fmax(best, acc);
    }
    best = wave_max_f64(best);
    if (lane == 0) smax[ql] = (float)best;
}

int launch_rowmax_exact(hipStream_t s, const RefineArgs& a, float* smax) {
    if (a.splits * 2 > RF_MAX_CAND) { set_error("rowmax_exact: too many segments"); return DAGL_ERR_INVALID; }
    const size_t nq = (size_t)a.B * a.L;
    hipLaunchKernelGGL(rowmax_exact_kernel, dim3((unsigned)((nq + 3) / 4)), dim3(256), 0, s, a, smax);
    DAGL_LAUNCH_CHECK("rowmax_exact_kernel");
    return DAGL_OK;
}

// total / max degree over all queries: one block (a same-address atomic per query would serialise ~12 ns each)
// stats[0] = total edges, stats[1] = largest degree; count_over > 0: stats[2] = rows whose degree exceeds it (the dense
// formulation's count of the queries that would not fit the neighbour lists)
__global__ __launch_bounds__(1024) void degree_stats_kernel(size_t n_rows, const int32_t* __restrict__ nb_cnt,
                                                            int64_t* __restrict__ stats, int count_over) {
    __shared__ long long ssum[16];
    __shared__ int smax[16], sover[16];
    long long sum = 0; int mx = 0, over = 0;
    for (size_t r = threadIdx.x; r < n_rows; r += blockDim.x) {
        const int d = nb_cnt[r]; sum += d; mx = max(mx, d); over += (count_over > 0 && d > count_over) ? 1 : 0;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { sum += __shfl_xor(sum, o); mx = max(mx, __shfl_xor(mx, o)); over += __shfl_xor(over, o); }
    if ((threadIdx.x & 63) == 0) { ssum[threadIdx.x >> 6] = sum; smax[threadIdx.x >> 6] = mx; sover[threadIdx.x >> 6] = over; }
    __syncthreads();
    if (threadIdx.x == 0) {
        long long t = 0; int m = 0, ov = 0;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) { t += ssum[w]; m = max(m, smax[w]); ov += sover[w]; }
        stats[0] = t; stats[1] = m;
        if (count_over > 0) stats[2] = ov;
    }
}

int launch_degree_stats(hipStream_t s, size_t n_rows, const int32_t* nb_cnt, int64_t* stats, int count_over) {
    hipLaunchKernelGGL(degree_stats_kernel, dim3(1), dim3(1024), 0, s, n_rows, nb_cnt, stats, count_over);
    DAGL_LAUNCH_CHECK("degree_stats_kernel");
    return DAGL_OK;
}

int launch_refine(hipStream_t s, const RefineArgs& a) {
    if (a.splits * 2 * a.capseg > RF_MAX_CAND && a.splits * 2 > RF_MAX_CAND) {
        set_error("refine: too many segments"); return DAGL_ERR_INVALID;
    }
    const size_t nq = (size_t)a.B * a.L;
#ifdef DAGL_ABLATION
    RefineArgs at = a; at.times = getenv("DAGL_TIMES_FILE") ? dbg_times_buffer((nq + 3) / 4) : nullptr;
    hipLaunchKernelGGL(refine_kernel, dim3((unsigned)((nq + 3) / 4)), dim3(256), 0, s, at);
    if (at.times) dbg_times_dump(s, "refine_kernel", at.times, (nq + 3) / 4);
#else
    hipLaunchKernelGGL(refine_kernel, dim3((unsigned)((nq + 3) / 4)), dim3(256), 0, s, a);
#endif
    DAGL_LAUNCH_CHECK("refine_kernel");
    if (a.heavy_list != nullptr && a.mode == DAGL_MODE_ADAPTIVE) {
        hipLaunchKernelGGL(refine_heavy_kernel, dim3(256), dim3(64 * RF_HEAVY_WAVES), 0, s, a);
        DAGL_LAUNCH_CHECK("refine_heavy_kernel");
    }
    return DAGL_OK;
}
int refine_heavy_cap() { return RF_HEAVY_CAP; }

__global__ __launch_bounds__(256) void topk_policy_kernel(int64_t* stats, int32_t* policy, int32_t* gate, int32_t* redo_flags, int n_flags,
                                                          long long n_queries) {
    // the re-run pays when the redo pass would scan more than a fiftieth of the query GROUPS (a flagged query sends its whole
    // 128-query group through the fp32 scan, ~50x the screen's price per pair: 84 us per group at 256^2, 5 ms at 1024^2 where the
    // tight re-run of everything is 37; a few hundred flagged queries of a natural-image map are spread over most groups)
    __shared__ int go, n_groups;
    if (threadIdx.x == 0) n_groups = 0;
    __syncthreads();
    if (*policy == 0 && stats[2] > 0) {
        int loc = 0;
        for (int e = threadIdx.x; e < n_flags; e += blockDim.x) loc += redo_flags[e] != 0 ? 1 : 0;
        if (loc) atomicAdd(&n_groups, loc);
    }
    __syncthreads();
    if (threadIdx.x == 0) go = (*policy == 0 && (long long)n_groups * 50 > (long long)n_flags) ? 1 : 0;
    (void)n_queries;
    __syncthreads();
    if (!go) return;
    for (int e = threadIdx.x; e < n_flags; e += blockDim.x) redo_flags[e] = 0;
    __syncthreads();
    if (threadIdx.x == 0) { stats[0] = 0; stats[1] = 0; stats[2] = 0; *policy = 1; *gate = 1; }
}

int launch_topk_policy(hipStream_t s, int64_t* stats, int32_t* policy, int32_t* gate, int32_t* redo_flags, int n_flags, long long n_queries) {
    hipLaunchKernelGGL(topk_policy_kernel, dim3(1), dim3(256), 0, s, stats, policy, gate, redo_flags, n_flags, n_queries);
    DAGL_LAUNCH_CHECK("topk_policy_kernel");
    return DAGL_OK;
}

}  // namespace dagl
