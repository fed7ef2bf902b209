// Shared by aggregate.hip (edge softmax kernels) and select.hip (the fused redo kernel of the top-k modes): the edge logit, fp64
// wave reductions, and the merge of a query's per-(chunk, half) k-best lists into its exact k best + weights.
#pragma once
#include "dagl_common.h"

namespace dagl {

// ------------------------------------------------------------------------------------------------------
// edge softmax
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float edge_logit(float s, float mtq, float bsq, bool adaptive) {
    // l = (S * m) * 10 in fp32, m = relu((S - mean*thr) + bias) or 1  -- the reference's operation order
    float m = 1.0f;
    if (adaptive) m = (s - mtq) + bsq;
    return __fmul_rn(__fmul_rn(s, m), SOFTMAX_SCALE);
}

// (all 64 lanes active at every call site; DPP forms: dagl_common.h)
__device__ __forceinline__ double wave_max_d(double v) { return wave_max_f64(v); }
__device__ __forceinline__ double wave_sum_d(double v) { return wave_sum_f64(v); }

// MODE 2/3: merge the per-(chunk, half) k-best candidate lists of a query into its exact k best
// (value descending, ties -> smaller key index), then weights.  One wave per query; candidates in LDS.
constexpr int TOPK_MAX_CAND = 1024;
__device__ __forceinline__ void edge_softmax_topk_unit(const EdgeArgs& a, int kslots, float (*cv)[TOPK_MAX_CAND],
                                                       int (*ci)[TOPK_MAX_CAND], size_t blk) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const size_t ql = blk * 4 + w;
    bool active = ql < (size_t)a.B * a.L;
    if (active && a.run_flags != nullptr) {
        const size_t b = ql / a.L, qg = (ql - b * a.L) / 128;
        active = a.run_flags[b * ((a.L + 127) / 128) + qg] != 0;
    }
    const int nc = a.splits * 2 * kslots;
    if (active) {
        for (int t = lane; t < nc; t += 64) {
            cv[w][t] = a.cand_val[ql * nc + t];
            ci[w][t] = a.cand_idx[ql * nc + t];
        }
    }
    __syncthreads();
    if (!active) return;
    const bool adaptive = (a.mode == DAGL_MODE_ADAPTIVE_TOPK);
    const float mtq = adaptive ? a.mt[ql] : 0.f, bsq = adaptive ? a.bs[ql] : 0.f;
    float my_s = 0.f; int my_key = -1;          // lane r keeps the r-th selected neighbour
    int n = 0;
    for (int r = 0; r < a.k; ++r) {
        float bv = -2.f; int bi = 0x7fffffff, bpos = -1;
        for (int t = lane; t < nc; t += 64) {
            const float v = cv[w][t]; const int id = ci[w][t];
            if (id >= 0 && (v > bv || (v == bv && id < bi))) { bv = v; bi = id; bpos = t; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(bv, o); const int oi = __shfl_xor(bi, o); const int op = __shfl_xor(bpos, o);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; bpos = op; }
        }
        if (bpos < 0) break;                     // fewer than k candidates exist (wave-uniform)
        if (lane == r) { my_s = bv; my_key = bi; }
        if (lane == 0) ci[w][bpos] = -1;         // taken
        __threadfence_block();                   // LDS write visible to the wave's next scan
        ++n;
    }
    const bool valid = lane < n;
    const float lg = valid ? edge_logit(my_s, mtq, bsq, adaptive) : 0.f;
    double M = wave_max_d(valid ? (double)lg : -1e300);
    if (n < a.N) M = fmax(M, 0.0);
    const double e = valid ? exp((double)lg - M) : 0.0;
    const double sum = wave_sum_d(e) + (double)(a.N - n) * exp(-M);
    if (valid) {
        a.nb_idx[ql * a.width + lane] = my_key;
        a.nb_wgt[ql * a.width + lane] = (float)(e / sum);
        if (a.nb_s != nullptr) a.nb_s[ql * a.width + lane] = my_s;
    }
    if (lane == 0) a.nb_cnt[ql] = n;
}

}  // namespace dagl
