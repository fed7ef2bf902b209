// Fixed-k neighbourhoods wider than the per-query lists (k > DAGL_MAX_TOPK; the reference's fixed-k variant takes any
// num_edge: top_k = min(num_edge, N), GReccR2b_3mh_1-checkpoint.py:242-250; a stray sibling uses 500).  No lists: every query
// is handled like the adaptive mode's flagged rows (overflow.hip, row_attend.h) -- its whole score row against all keys from the
// fp32 matrix cores, a batch of rows at a time, then per row
//   select   the k-th largest score by a 4 x 8-bit radix selection over the row (scores are >= 0: their bit patterns sort like
//            the values), with the number of keys to take AT that value (ties go to the lower key index, like the list path)
//   attend   mask = the k best (AND the adaptive test in the intersection mode), softmax over all N keys (masked keys count e^0),
//            weighted sum of the value patches, in 32 key chunks with softmax statistics of their own
//   combine  the chunks -> the query's aggregated row, degree, softmax mass
// and the fold over everything at the end.  Inference; the differentiable path of these modes is the dense formulation of
// dense_train.hip with the same selection (wide_select.h) as its mask.
#include <string.h>

#include "dagl_common.h"
#include "row_attend.h"
#include "wide_select.h"

namespace dagl {

__device__ __forceinline__ float wide_logit(float s, bool adaptive, float mtq, float bsq) {
    // top-k: softmax(10 S mask), mask in {0, 1} (GReccR2b_3mh_1-checkpoint.py:248-250); intersection: 10 S m, m = relu(S - mt + bs)
    return adaptive ? __fmul_rn(__fmul_rn(s, (s - mtq) + bsq), SOFTMAX_SCALE) : __fmul_rn(s, SOFTMAX_SCALE);
}

constexpr int WIDE_RANGES = ROW_CHUNKS * 4;          // (chunk, wave) key ranges of the attend kernel

// block = one row.  sel[slot] = {threshold key T, keys to take with key == T (0x7fffffff: all of them), -, -};
// eq_before[slot][r] = keys == T in the (chunk, wave) ranges before range r (only when not all are taken)
__global__ __launch_bounds__(256) void wide_select_kernel(WideArgs a) {
    __shared__ WideSelShared shs;
    __shared__ int sh_eq[WIDE_RANGES];
    const int slot = blockIdx.x, tid = threadIdx.x;
    if (a.served != nullptr && a.served[slot] != 0) return;        // (wide_list_kernel has done this row)
    const size_t ql = (size_t)a.b * a.g.L + a.r0 + slot;
    const float* row = a.scores + (size_t)slot * a.ldn;
    const bool adaptive = a.mode == DAGL_MODE_ADAPTIVE_TOPK;
    const float mtq = adaptive ? a.mt[ql] : 0.f, bsq = adaptive ? a.bs[ql] : 0.f;
    const int N = a.g.N;
    unsigned T, need, bin_count;
    wide_radix_select(row, N, a.k, adaptive, mtq, bsq, shs, T, need, bin_count);    // `need` of the `bin_count` keys equal to T are taken
    const bool all = need >= bin_count;
    if (tid == 0) {
        int32_t* o = a.sel + (size_t)slot * 4;
        o[0] = (int32_t)T; o[1] = all ? 0x7fffffff : (int32_t)need; o[2] = 0; o[3] = 0;
    }
    if (all) return;                                              // block-uniform
    // a tie at the k-th place: the lower key indices win -- count the equal keys per (chunk, wave) range of the attend kernel
    if (tid < WIDE_RANGES) sh_eq[tid] = 0;
    __syncthreads();
    for (int r = tid >> 6; r < WIDE_RANGES; r += 4) {
        int j0c, j1c, j0, j1; row_chunk_range(N, r >> 2, j0c, j1c); row_wave_range(j0c, j1c, r & 3, j0, j1);
        int cnt = 0;
        for (int j = j0 + (tid & 63); j < j1; j += 64) cnt += (wide_key(row[j], adaptive, mtq, bsq) == T) ? 1 : 0;
        cnt = wave_sum_i32(cnt);
        if ((tid & 63) == 0) sh_eq[r] = cnt;
    }
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        for (int r = 0; r < WIDE_RANGES; ++r) { a.eq_before[(size_t)slot * WIDE_RANGES + r] = run; run += sh_eq[r]; }
    }
}

__global__ __launch_bounds__(256) void wide_attend_kernel(WideArgs a) {
    __shared__ RowBlockShared sh;
    __shared__ int shc[4];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const RowCols cols = row_cols(lane);
    const int chunk = blockIdx.x;
    int j0c, j1c; row_chunk_range(a.g.N, chunk, j0c, j1c);
    int j0, j1; row_wave_range(j0c, j1c, w, j0, j1);
    const bool adaptive = a.mode == DAGL_MODE_ADAPTIVE_TOPK;
    const float4* vmb = reinterpret_cast<const float4*>(a.b2p + (size_t)a.b * a.g.Hp * a.g.Wp * CH);
    for (int slot = blockIdx.y; slot < a.R; slot += gridDim.y) {
        if (a.served != nullptr && a.served[slot] != 0) continue;  // (block-uniform)
        const size_t ql = (size_t)a.b * a.g.L + a.r0 + slot;
        const float* row = a.scores + (size_t)slot * a.ldn;
        const float mtq = adaptive ? a.mt[ql] : 0.f, bsq = adaptive ? a.bs[ql] : 0.f;
        const unsigned T = (unsigned)a.sel[(size_t)slot * 4];
        const int need = a.sel[(size_t)slot * 4 + 1];
        int eq_seen = (need == 0x7fffffff) ? 0 : a.eq_before[(size_t)slot * WIDE_RANGES + chunk * 4 + w];   // wave-uniform running rank
        RowAcc o; row_acc_clear(o);
        row_wave_walk(a.g, vmb, cols, lane, j0, j1, true,
                      [&](int j, bool in, float& l) {
                          const float s = in ? row[j] : 0.f;
                          const unsigned key = in ? wide_key(s, adaptive, mtq, bsq) : 0u;
                          const bool eq = in && key == T && key != 0u;
                          const unsigned long long eqb = __ballot(eq);
                          const int rank = eq_seen + __popcll(eqb & ((1ull << lane) - 1ull));
                          eq_seen += __popcll(eqb);
                          const bool pass = in && key != 0u && (key > T || (eq && rank < need));
                          if (pass) l = wide_logit(s, adaptive, mtq, bsq);
                          return pass;
                      }, o);
        if (lane == 0) shc[w] = o.cnt;
        __syncthreads();
        const int cnt_blk = shc[0] + shc[1] + shc[2] + shc[3];
        row_block_store(sh, cols, o, cnt_blk, a.part + ((size_t)slot * ROW_CHUNKS + chunk) * ROW_PART_FLOATS);
    }
}

__global__ __launch_bounds__(256) void wide_combine_kernel(WideArgs a) {
    __shared__ RowReduceShared sh;
    const int tid = threadIdx.x;
    for (int slot = blockIdx.x; slot < a.R; slot += gridDim.x) {
        if (a.served != nullptr && a.served[slot] != 0) continue;  // (block-uniform)
        const size_t ql = (size_t)a.b * a.g.L + a.r0 + slot;
        const float* part_row = a.part + (size_t)slot * ROW_CHUNKS * ROW_PART_FLOATS;
        const RowSum row = row_reduce(part_row, a.g.N, sh);
        if (tid < P / 4) reinterpret_cast<float4*>(a.agg)[ql * (P / 4) + tid] = row_combine(part_row, row, sh);
        if (tid == 0) {
            a.deg[ql] = row.deg;
            if (a.rowsum) a.rowsum[ql] = (float)(row.zs / row.Z);
        }
    }
}

// ---- round 5: k of a few hundred -- selection, softmax and weighted sum of a row in ONE block -----------------------------------------
// The three kernels above read a 256 KiB score row six times to find its k-th largest (4 radix passes + ties), once more to mask it
// (every one of the N scores walked again to find the k that pass) and hand partial rows through memory to a combine launch:
// 0.70 + 0.52-0.73 + 0.05 ms per 2048 rows at 256^2, whatever k (profiles/r05_topk_wide.log).  For k <= WL_KMAX a block now
//   A  samples every s-th score (4096 samples in the LDS) and takes the r-th largest sample, r a little above k / s (+ 3 sigma), as a
//      LOWER bound t of the k-th largest score -- a bit-wise search over the samples' sort keys;
//   B  reads the row ONCE: every wave compacts the scores >= t of its eighth of the row, in key order, into its segment of a
//      candidate list in the LDS (ballot + prefix: no atomics, the list is ordered by key whatever the timing);
//   C  finds the k-th largest among the candidates exactly (same search), ties to the lower key (the order of the list), and forms the
//      final list: map offsets + logits;
//   D  softmax over the list (+ (N - k) e^-M for the masked keys' e^0, dagl.py:259-261; fp64 exp / sums in a fixed order) and the
//      weighted sum of the value patches: wave w takes entries w, w + 8, .., a lane four float4 columns, sixteen gathers in flight; the
//      eight waves' partial rows are added in wave order.
// A row whose sample misleads (fewer than k candidates, or more than a segment holds: flat maps) is left to the kernels above
// (`served` = 0): same results, old speed.
constexpr int WL_KMAX = 1024;                        // largest k served from the list
constexpr int WL_SAMPLES = 4096;
constexpr int WL_WAVES = 8;                          // 512 threads: the block's phases are chains of memory round trips -- more waves, shorter chains
constexpr int WL_THREADS = 64 * WL_WAVES;
constexpr int WL_SEG = 512;                          // candidate slots per wave (an eighth of the row)
constexpr int WL_PER = WL_SAMPLES / WL_THREADS;      // keys a thread holds in the two selections (8)
static_assert(WL_WAVES * WL_SEG == WL_SAMPLES, "one register set serves both selections");
constexpr float WL_RESCORE_LOGIT = 2.0e4f;            // rows whose largest logit exceeds this get their selected keys' scores again, exactly
struct WideListShared {
    // one 32 KiB region, two lives: B / C the candidates ckey[8][512] | cscore[8][512]; D the waves' partial rows (the sample's keys live in registers)
    union {
        struct { int ckey[WL_WAVES][WL_SEG]; float cscore[WL_WAVES][WL_SEG]; } c;
        float4 part[WL_WAVES][P / 4];
    } u;
    int lst_of[WL_KMAX]; float lst_w[WL_KMAX];       // C / D: the selected keys' map offsets (float4 units) and logits -> weights (8 KiB)
    double dred[WL_THREADS];
    float fred[WL_WAVES];
    int ired[3 * WL_WAVES];
    int wcnt[WL_WAVES], wgt_n[WL_WAVES], weq_n[WL_WAVES];
};

// r-th largest of the block's sort keys, WL_PER per thread in registers (0 = no key; 0 when there are fewer than r keys): a bit-wise
// search from the highest bit the keys DIFFER in (scores of a row share sign, exponent and often the leading mantissa bits: ~a dozen
// steps saved, two block barriers each) down to bit `lo_bit` -- lo_bit > 0 gives a value BELOW the r-th largest with at least r keys
// above it, which is all the sample's lower bound needs.  Block-uniform result.
__device__ __forceinline__ unsigned wl_kth_largest(const unsigned (&key)[WL_PER], int r, int* ired, int lo_span = 32) {
    const int tid = threadIdx.x, w = tid >> 6;
    unsigned kor = 0u, kand = 0xffffffffu; int nz = 0;
#pragma unroll
    for (int u = 0; u < WL_PER; ++u) { if (key[u] != 0u) { kor |= key[u]; kand &= key[u]; ++nz; } }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { kor |= __shfl_xor(kor, o); kand &= __shfl_xor(kand, o); }
    nz = wave_sum_i32(nz);
    __syncthreads();                                              // (ired's previous use)
    if ((tid & 63) == 0) { ired[w] = nz; ired[WL_WAVES + w] = (int)kor; ired[2 * WL_WAVES + w] = (int)kand; }
    __syncthreads();
    int all_nz = 0;
#pragma unroll
    for (int ww = 0; ww < WL_WAVES; ++ww) { all_nz += ired[ww]; kor |= (unsigned)ired[WL_WAVES + ww]; kand &= (unsigned)ired[2 * WL_WAVES + ww]; }
    if (all_nz < r) return 0u;                                    // block-uniform
    const unsigned diff = kor ^ kand;
    if (diff == 0u) return kor;                                   // every key the same value
    const int top = 31 - __clz((int)diff);                        // highest bit two keys differ in; the bits above it are common
    unsigned prefix = (top == 31) ? 0u : (kand & ~((2u << top) - 1u));
    const int lo = (top + 1 - lo_span > 0) ? top + 1 - lo_span : 0;
    for (int bit = top; bit >= lo; --bit) {
        const unsigned t = prefix | (1u << bit);
        int cnt = 0;
#pragma unroll
        for (int u = 0; u < WL_PER; ++u) cnt += key[u] >= t ? 1 : 0;
        cnt = wave_sum_i32(cnt);
        __syncthreads();                                          // (the previous step's sums have been read)
        if ((tid & 63) == 0) ired[w] = cnt;
        __syncthreads();
        int all = 0;
#pragma unroll
        for (int ww = 0; ww < WL_WAVES; ++ww) all += ired[ww];
        if (all >= r) prefix = t;
    }
    return prefix;
}

__global__ __launch_bounds__(WL_THREADS) void wide_list_kernel(WideArgs a) {
    __shared__ WideListShared sh;
    const int slot = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const size_t ql = (size_t)a.b * a.g.L + a.r0 + slot;
    const float* __restrict__ row = a.scores + (size_t)slot * a.ldn;
    const bool adaptive = a.mode == DAGL_MODE_ADAPTIVE_TOPK;
    const float mtq = adaptive ? a.mt[ql] : 0.f, bsq = adaptive ? a.bs[ql] : 0.f;
    const int N = a.g.N, k = a.k;
    // ---- A: a lower bound of the k-th largest score from a sample --------------------------------------------------------------------
    // the sample: whole 128-byte lines (32 scores), one of every `stride` lines -- a score of every 64 bytes would touch EVERY line of
    // the row and read it a second time; which line of a group of `stride` rotates (5 g mod stride), so that the sample does not sit
    // on one band of image columns (a key index is a pixel in raster order)
    const int n_lines = (N + 31) / 32;
    const int stride = (n_lines * 32 + WL_SAMPLES - 1) / WL_SAMPLES;               // lines per sampled line
    const int n_groups = (n_lines + stride - 1) / stride;
    const int ns = n_groups * 32;                                                  // <= WL_SAMPLES
    unsigned sk[WL_PER];
#pragma unroll
    for (int u = 0; u < WL_PER; ++u) {                                             // (independent loads: all in flight)
        const int e = tid + WL_THREADS * u;
        const int g = e >> 5;
        int line = g * stride + (stride > 1 ? (5 * g) % stride : 0);
        if (line > n_lines - 1) line = n_lines - 1;
        const int j = line * 32 + (e & 31);
        sk[u] = (e < ns && j < N) ? wide_key(row[j], adaptive, mtq, bsq) : 0u;
    }
    // (neighbouring pixels' scores are correlated: the sample is worth fewer independent draws than it has entries -- a wide margin)
    int rs = k;
    if (stride > 1) { const float r = (float)k / (float)stride; rs = (int)(r + 6.0f * sqrtf(r) + 8.0f); }
    if (rs > ns) rs = ns;
    unsigned tl = wl_kth_largest(sk, rs, sh.ired, 12);          // (12 bits below the first differing one: a bound 2^-12 of the spread loose)
    if (tl < 1u) tl = 1u;                                           // (key 0 = "not a candidate")
    // ---- B: the candidates, per wave eighth, in key order -------------------------------------------------------------------------------
    {
        // a lane takes FOUR consecutive scores per load (16 bytes), eight loads in flight per wave (one load per round trip left a
        // wave's share at hundreds of dependent round trips); positions by a wave scan of the lanes' counts: still key order
        const int per = ((N + WL_WAVES - 1) / WL_WAVES + 255) / 256 * 256;
        const int j0 = min(w * per, N), j1 = min(j0 + per, N);
        const bool vec = (a.ldn % 4 == 0);                          // (16-byte aligned rows)
        int cw = 0;
        constexpr int UB = 8;
        for (int c0 = j0; c0 < j1; c0 += 256 * UB) {
            float4 v[UB];
#pragma unroll
            for (int ub = 0; ub < UB; ++ub) {
                const int j = c0 + 256 * ub + 4 * lane;
                // (streaming loads: the 537 MB of score rows per 2048 queries pass through once -- as ordinary loads they push the value
                // map, which the weighted sum behind them gathers from, out of the L2: k = 100 1.32 -> 1.26 ms)
                if (vec && j + 3 < j1) { typedef float wlf4 __attribute__((ext_vector_type(4))); const wlf4 t = __builtin_nontemporal_load(reinterpret_cast<const wlf4*>(row + j)); v[ub] = make_float4(t[0], t[1], t[2], t[3]); }
                else v[ub] = make_float4(j < j1 ? row[j] : 0.f, j + 1 < j1 ? row[j + 1] : 0.f, j + 2 < j1 ? row[j + 2] : 0.f, j + 3 < j1 ? row[j + 3] : 0.f);
            }
#pragma unroll
            for (int ub = 0; ub < UB; ++ub) {
                const int j = c0 + 256 * ub + 4 * lane;
                const float sc[4] = {v[ub].x, v[ub].y, v[ub].z, v[ub].w};
                bool ps[4]; int cnt = 0;
#pragma unroll
                for (int q = 0; q < 4; ++q) { ps[q] = j + q < j1 && wide_key(sc[q], adaptive, mtq, bsq) >= tl; cnt += ps[q] ? 1 : 0; }
                if (!__any(cnt != 0)) continue;                     // (wave-uniform)
                const int incl = wave_scan_incl_i32(cnt);
                int pos = cw + incl - cnt;
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (ps[q]) { if (pos < WL_SEG) { sh.u.c.ckey[w][pos] = j + q; sh.u.c.cscore[w][pos] = sc[q]; } ++pos; }
                cw += __builtin_amdgcn_readlane(incl, 63);
            }
        }
        if (lane == 0) sh.wcnt[w] = cw;
    }
    __syncthreads();
    int total = 0; bool fits = true;
#pragma unroll
    for (int ww = 0; ww < WL_WAVES; ++ww) { total += sh.wcnt[ww]; fits = fits && sh.wcnt[ww] <= WL_SEG; }
    // (tl == 1: every candidate of the row is in the list -- fewer than k of them is then the row's true degree)
    if (!fits || (tl > 1u && total < k)) { if (tid == 0) a.served[slot] = 0; return; }       // block-uniform: left to the three-kernel form
    // ---- C: the k best of the candidates, ties to the lower key -------------------------------------------------------------------------
    unsigned ck[WL_PER];
#pragma unroll
    for (int u = 0; u < WL_PER; ++u) {                              // thread (w, lane) holds its own wave's segment: positions lane + 64 u
        const int i = lane + 64 * u;
        ck[u] = i < sh.wcnt[w] ? wide_key(sh.u.c.cscore[w][i], adaptive, mtq, bsq) : 0u;
    }
    const int kk = total < k ? total : k;
    unsigned T = 0u;
    if (kk > 0) T = wl_kth_largest(ck, kk, sh.ired);
    {
        int ngt = 0, neq = 0;                                       // this wave's segment: keys above / at the threshold
#pragma unroll
        for (int u = 0; u < WL_PER; ++u) { ngt += (ck[u] > T) ? 1 : 0; neq += (ck[u] == T && ck[u] != 0u) ? 1 : 0; }
        ngt = wave_sum_i32(ngt); neq = wave_sum_i32(neq);
        if (lane == 0) { sh.wgt_n[w] = ngt; sh.weq_n[w] = neq; }
    }
    __syncthreads();
    int gt_all = 0;
#pragma unroll
    for (int ww = 0; ww < WL_WAVES; ++ww) gt_all += sh.wgt_n[ww];
    const int need = kk - gt_all;                                   // of the keys AT the threshold: the first `need` in key order
    {
        int eq_before = 0, sel_before = 0;
        for (int ww = 0; ww < w; ++ww) {
            const int take = min(max(need - eq_before, 0), sh.weq_n[ww]);
            sel_before += sh.wgt_n[ww] + take; eq_before += sh.weq_n[ww];
        }
        int eq_run = eq_before, pos_run = sel_before;
#pragma unroll
        for (int u = 0; u < WL_PER; ++u) {
            const int i = lane + 64 * u;
            const unsigned key = ck[u];
            const bool eq = key == T && key != 0u;
            const unsigned long long eqb = __ballot(eq);
            const int rank = eq_run + __popcll(eqb & ((1ull << lane) - 1ull));
            eq_run += __popcll(eqb);
            const bool pick = key != 0u && (key > T || (eq && rank < need));
            const unsigned long long pb = __ballot(pick);
            const int pos = pos_run + __popcll(pb & ((1ull << lane) - 1ull));
            pos_run += __popcll(pb);
            if (pick) {
                const int j = sh.u.c.ckey[w][i];
                const int jy = j / a.g.W, jx = j - jy * a.g.W;
                sh.lst_of[pos] = (jy * a.g.Wp + jx) * (CH / 4);
                sh.lst_w[pos] = wide_logit(sh.u.c.cscore[w][i], adaptive, mtq, bsq);
            }
        }
    }
    __syncthreads();                                                // (the candidates are dead from here: their memory holds the partial rows below)
    const int deg = kk;                                             // (every candidate has a non-zero key: kk of them are picked)
    // ---- D: softmax over the list, weighted sum ------------------------------------------------------------------------------------------
    float m = -1.f;
    for (int e = tid; e < deg; e += WL_THREADS) m = fmaxf(m, sh.lst_w[e]);
    m = wave_max_f32(m);
    if (lane == 0) sh.fred[w] = m;
    __syncthreads();
    float mf = sh.fred[0];
#pragma unroll
    for (int ww = 1; ww < WL_WAVES; ++ww) mf = fmaxf(mf, sh.fred[ww]);
    if (a.x != nullptr && mf > WL_RESCORE_LOGIT) {                  // (block-uniform)
        // Rows with logits of tens of thousands: the scores came from split-fp16 products (2^-22 relative: a logit of 1e5 is off by
        // 0.02, tools/scale_sweep.py at x 30 inputs: 1.1e-3 against the fp64 oracle where fp32 scores gave 1.6e-4) -- the selected keys
        // are scored again from the fp32 feature rows, accumulated in fp64.  A thread per key, its row read serially: a path for inputs
        // far outside anything trained features produce (logits 5-70), never taken otherwise.
        float* qrow = reinterpret_cast<float*>(&sh.u.part[0][0]);      // (dead until the partial rows below)
        const float* qsrc = a.wq + ((size_t)a.b * a.rows_q + a.r0 + slot) * DS;
        for (int c = tid; c < D; c += WL_THREADS) qrow[c] = qsrc[c];
        __syncthreads();
        float m2 = -1.f;
        for (int e = tid; e < deg; e += WL_THREADS) {
            const int po = sh.lst_of[e] / (CH / 4);
            const int jy = po / a.g.Wp, jx = po - jy * a.g.Wp;
            const float* xr = a.x + ((size_t)a.b * a.rows_x + (size_t)jy * a.g.W + jx) * DS;
            double acc = 0.0;
            for (int c = 0; c < D; c += 4) {
                const float4 v = *reinterpret_cast<const float4*>(xr + c);
                acc += (double)qrow[c] * (double)v.x + (double)qrow[c + 1] * (double)v.y + (double)qrow[c + 2] * (double)v.z + (double)qrow[c + 3] * (double)v.w;
            }
            const float lg = wide_logit((float)acc, adaptive, mtq, bsq);
            sh.lst_w[e] = lg;
            m2 = fmaxf(m2, lg);
        }
        m2 = wave_max_f32(m2);
        __syncthreads();                                            // (fred has been read by every thread)
        if (lane == 0) sh.fred[w] = m2;
        __syncthreads();
        mf = sh.fred[0];
#pragma unroll
        for (int ww = 1; ww < WL_WAVES; ++ww) mf = fmaxf(mf, sh.fred[ww]);
        __syncthreads();                                            // (qrow is dead: the partial rows may be written)
    }
    double M = (double)mf;
    if (deg < N) M = fmax(M, 0.0);
    double zloc = 0.0;
    for (int e = tid; e < WL_KMAX; e += WL_THREADS) {               // (fixed assignment: thread t sums entries t, t + 512 in that order)
        if (e < deg) { const double wv = exp((double)sh.lst_w[e] - M); sh.lst_w[e] = (float)wv; zloc += wv; }
    }
    sh.dred[tid] = zloc;
    __syncthreads();
    for (int st = WL_THREADS / 2; st > 0; st >>= 1) {               // fixed tree: the same sum on every run
        if (tid < st) sh.dred[tid] += sh.dred[tid + st];
        __syncthreads();
    }
    const double zs = sh.dred[0];
    const double Z = zs + (double)(N - deg) * exp(-M);
    const RowCols cols = row_cols(lane);
    const float4* vmb = reinterpret_cast<const float4*>(a.b2p + (size_t)a.b * a.g.Hp * a.g.Wp * CH);
    int coff[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) coff[u] = cols.kh[u] * a.g.Wp * (CH / 4) + cols.rem[u];
    float4 acc[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) acc[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    constexpr int GB = 4;                                           // entries in flight per wave (x 4 columns = 16 gathers per lane)
    for (int e0 = w; e0 < deg; e0 += WL_WAVES * GB) {
        float4 v[GB][4]; float wv[GB];
#pragma unroll
        for (int g2 = 0; g2 < GB; ++g2) {
            const int e = min(e0 + WL_WAVES * g2, deg - 1);
            const int of = sh.lst_of[e];
            wv[g2] = (e0 + WL_WAVES * g2 < deg) ? sh.lst_w[e] : 0.f;
#pragma unroll
            for (int u = 0; u < 4; ++u) v[g2][u] = vmb[of + coff[u]];
        }
#pragma unroll
        for (int g2 = 0; g2 < GB; ++g2)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                acc[u].x = fmaf(wv[g2], v[g2][u].x, acc[u].x); acc[u].y = fmaf(wv[g2], v[g2][u].y, acc[u].y);
                acc[u].z = fmaf(wv[g2], v[g2][u].z, acc[u].z); acc[u].w = fmaf(wv[g2], v[g2][u].w, acc[u].w);
            }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) if (cols.cv[u]) sh.u.part[w][lane + 64 * u] = acc[u];
    __syncthreads();
    if (tid < P / 4) {
        const float inv = (float)(1.0 / Z);
        float4 t = sh.u.part[0][tid];
#pragma unroll
        for (int ww = 1; ww < WL_WAVES; ++ww) { const float4 q = sh.u.part[ww][tid]; t.x += q.x; t.y += q.y; t.z += q.z; t.w += q.w; }   // wave order
        reinterpret_cast<float4*>(a.agg)[ql * (P / 4) + tid] = make_float4(t.x * inv, t.y * inv, t.z * inv, t.w * inv);
    }
    if (tid == 0) {
        a.deg[ql] = deg;
        if (a.rowsum) a.rowsum[ql] = (float)(zs / Z);
        a.served[slot] = 1;
    }
}

// rows per batch: 512 MiB of scores at most (as the adaptive mode's redo), whole 128-row tiles of the product
int topk_wide_rows(int N, int L) {
    const long long ldn = (N + 31) / 32 * 32;
    long long c = ((long long)512 << 20) / (ldn * 4);
    c = c / 128 * 128;
    if (c < 128) c = 128;
    if (c > 2048) c = 2048;
    const long long lr = (L + 127) / 128 * 128;
    return (int)(c < lr ? c : lr);
}
// split-fp16 copy of one image's feature rows (hi or lo): [feat_rows_h(rows) + a tile of slack][DSH] halfs
static size_t wide_split_bytes(int rows) { return align_up(((size_t)feat_rows_h(rows) + 256) * DSH * sizeof(uint16_t), 256); }
size_t topk_wide_workspace_bytes(int N, int L) {
    const size_t R = (size_t)topk_wide_rows(N, L);
    const size_t ldn = (size_t)(N + 31) / 32 * 32;
    return align_up(R * ldn * sizeof(float), 256) + align_up(R * ROW_CHUNKS * ROW_PART_FLOATS * sizeof(float), 256) +
           align_up(R * 4 * sizeof(int32_t), 256) + align_up(R * WIDE_RANGES * sizeof(int32_t), 256) + align_up(R * sizeof(int32_t), 256) +
           2 * wide_split_bytes(N) + 2 * wide_split_bytes(L) + 256;
}

int launch_topk_wide(hipStream_t s, int B, const Grid& g, int mode, int k, const float* wq, const float* x, const float* mt,
                     const float* bs, const float* b2p, void* ws, float* agg, int32_t* deg, float* rowsum, RangeTag range) {
    WideArgs a;
    memset(&a, 0, sizeof(a));
    a.g = g; a.mode = mode; a.k = k; a.mt = mt; a.bs = bs; a.b2p = b2p; a.agg = agg; a.deg = deg; a.rowsum = rowsum;
    a.wq = wq; a.x = x; a.rows_q = feat_rows(g.L); a.rows_x = feat_rows(g.N);
    const int Rmax = topk_wide_rows(g.N, g.L);
    a.ldn = (g.N + 31) / 32 * 32;
    char* p = static_cast<char*>(ws);
    a.scores = reinterpret_cast<float*>(p); p += align_up((size_t)Rmax * a.ldn * sizeof(float), 256);
    a.part = reinterpret_cast<float*>(p); p += align_up((size_t)Rmax * ROW_CHUNKS * ROW_PART_FLOATS * sizeof(float), 256);
    a.sel = reinterpret_cast<int32_t*>(p); p += align_up((size_t)Rmax * 4 * sizeof(int32_t), 256);
    a.eq_before = reinterpret_cast<int32_t*>(p); p += align_up((size_t)Rmax * WIDE_RANGES * sizeof(int32_t), 256);
    int32_t* served = reinterpret_cast<int32_t*>(p); p += align_up((size_t)Rmax * sizeof(int32_t), 256);
    uint16_t* xs_hi = reinterpret_cast<uint16_t*>(p); p += wide_split_bytes(g.N);
    uint16_t* xs_lo = reinterpret_cast<uint16_t*>(p); p += wide_split_bytes(g.N);
    uint16_t* qs_hi = reinterpret_cast<uint16_t*>(p); p += wide_split_bytes(g.L);
    uint16_t* qs_lo = reinterpret_cast<uint16_t*>(p); p += wide_split_bytes(g.L);
    unsigned* amax_words = reinterpret_cast<unsigned*>(p);              // [0] keys, [1] queries: bits of the image's largest feature
    const int rows_q = feat_rows(g.L), rows_x = feat_rows(g.N);
    // scores on the fp16 matrix cores with split operands (round 5; DAGL_WIDE_FP32_SCORES: the fp32 matrix cores as before): the image's
    // features as fp16 pairs, 64 x = hi + lo (dense.hip's copies: rows of 216 halfs, columns 196.. zero), three products per score
    // accumulated in fp32 -- >= 21 significant bits, as the projections -- at 0.36 instead of 0.58 ms per 2048 rows
#ifndef DAGL_WIDE_FP32_SCORES
    const bool split_scores = g.N >= 2048 && range.word != nullptr;      // (scan = "exact" has no range guard: fp32 products there)
#else
    const bool split_scores = false;
#endif
    const int rows_xh = feat_rows_h(g.N) + 256, rows_qh = feat_rows_h(g.L) + 256;
    for (int b = 0; b < B; ++b) {
        if (split_scores) {
            // (round 6) the features' split takes its power of two from the image's largest feature -- two small passes (53 + 3 MB at
            // 256^2: ~12 us of a 1.3 ms call) -- instead of the fixed 64: finite features of any size are served (a non-finite one still
            // sets the range word: the scale of an inf / NaN maximum is 1 and the split flags the value)
            DAGL_HIP_TRY(hipMemsetAsync(amax_words, 0, 2 * sizeof(unsigned), s));
            int rc = launch_absmax(s, (size_t)g.N * DS, x + (size_t)b * rows_x * DS, amax_words);
            if (rc) return rc;
            if ((rc = launch_absmax(s, (size_t)g.L * DS, wq + (size_t)b * rows_q * DS, amax_words + 1))) return rc;
            if ((rc = launch_feat_split(s, 1, g.N, rows_x, rows_xh, x + (size_t)b * rows_x * DS, xs_hi, xs_lo, range, amax_words))) return rc;
            if ((rc = launch_feat_split(s, 1, g.L, rows_q, rows_qh, wq + (size_t)b * rows_q * DS, qs_hi, qs_lo, range, amax_words + 1))) return rc;
        }
        for (int r0 = 0; r0 < g.L; r0 += Rmax) {
            const int R = (g.L - r0 < Rmax) ? g.L - r0 : Rmax;
            a.b = b; a.r0 = r0; a.R = R;
            if (split_scores) {
                Gemm16s gs;
                gs.M = R; gs.N = g.N; gs.K = 224; gs.k_valid = DSH;
                gs.a_hi = qs_hi + (size_t)r0 * DSH; gs.a_lo = qs_lo + (size_t)r0 * DSH; gs.lda = DSH; gs.a_rows = rows_qh - r0;
                gs.b_hi = xs_hi; gs.b_lo = xs_lo; gs.ldb = DSH; gs.b_rows = rows_xh;
                gs.C = a.scores; gs.ldc = a.ldn; gs.part = nullptr; gs.slices = 1; gs.scale_word = amax_words + 1; gs.scale_word_b = amax_words; gs.alpha0 = 1.0f;
                const int rc = launch_gemm16s(s, gs);
                if (rc) return rc;
            } else {
            // scores of the batch against all keys of its image: one product [R, 196] x [196, N] on the fp32 matrix cores
            // (chains of 48 products, partial sums added in fp32 -- the form the adaptive mode's flagged rows use)
            Gemm32 gm;
            gm.M = R; gm.N = g.N; gm.K = D; gm.batch = 1;
            gm.A = wq + ((size_t)b * rows_q + r0) * DS; gm.lda = DS; gm.sA = 0; gm.a_kc = 1;
            gm.B = x + (size_t)b * rows_x * DS; gm.ldb = DS; gm.sB = 0; gm.b_kc = 1;
            gm.C = a.scores; gm.ldc = a.ldn; gm.sC = 0;
            gm.alpha = 1.f; gm.beta = 0.f; gm.bias = nullptr; gm.relu = 0; gm.chunk_tiles = 3;
            int rc = launch_gemm32(s, gm);
            if (rc) return rc;
            }
            // k up to WL_KMAX: one block per row does everything from one read of the row; the rows it leaves alone (and larger k: all)
            // go through the three kernels behind it
            a.served = nullptr;
            if (k <= WL_KMAX) {
                a.served = served;
                hipLaunchKernelGGL(wide_list_kernel, dim3(R), dim3(WL_THREADS), 0, s, a);
                DAGL_LAUNCH_CHECK("wide_list_kernel");
            }
            hipLaunchKernelGGL(wide_select_kernel, dim3(R), dim3(256), 0, s, a);
            DAGL_LAUNCH_CHECK("wide_select_kernel");
            hipLaunchKernelGGL(wide_attend_kernel, dim3(ROW_CHUNKS, R < 64 ? R : 64), dim3(256), 0, s, a);
            DAGL_LAUNCH_CHECK("wide_attend_kernel");
            hipLaunchKernelGGL(wide_combine_kernel, dim3(R < 1024 ? R : 1024), dim3(256), 0, s, a);
            DAGL_LAUNCH_CHECK("wide_combine_kernel");
        }
    }
    return DAGL_OK;
}

}  // namespace dagl
