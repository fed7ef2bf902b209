// Row-wise selection of the k best scores by radix selection over the scores' bit patterns (shared by the inference form of the wide
// top-k modes, topk_wide.hip, and their differentiable form, dense_train.hip).  Reference: top_k = min(num_edge, N) best scores of
// every query, GReccR2b_3mh_1-checkpoint.py:242-250 (CA_model-checkpoint.py:134-143 uses 500).
#pragma once
#include "dagl_common.h"

namespace dagl {

// sort key of a score: 0 = "not a candidate" (fails the adaptive test of the intersection mode), else bits + 1
// (scores are >= 0 -- sums of products of post-ReLU features --, so their bit patterns sort like the values)
__device__ __forceinline__ unsigned wide_key(float s, bool adaptive, float mtq, float bsq) {
    if (adaptive && !(((s - mtq) + bsq) > 0.f)) return 0u;
    return __float_as_uint(fmaxf(s, 0.f)) + 1u;
}

struct WideSelShared {
    unsigned hist[256];
    unsigned prefix, remaining, bin_count;
};

// One block of 256 threads, one score row: T = sort key of the k-th best score (4 x 8-bit digits, most significant first), `need` =
// how many of the `bin_count` keys equal to T belong to the k best (all of them when need >= bin_count).  Block-uniform results.
__device__ __forceinline__ void wide_radix_select(const float* __restrict__ row, int N, int k, bool adaptive, float mtq, float bsq,
                                                  WideSelShared& sh, unsigned& T, unsigned& need, unsigned& bin_count) {
    const int tid = threadIdx.x;
    if (tid == 0) { sh.prefix = 0u; sh.remaining = (unsigned)k; }
    bin_count = 0;
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        sh.hist[tid] = 0u;
        __syncthreads();
        const unsigned prefix = sh.prefix;
        // (the high digit -- sign and seven exponent bits -- is the same for nearly every score of a row: 64 lanes adding to one LDS
        // word serialise, 65 536 times.  There the wave counts its lanes per distinct digit first: one add per digit.  For the second
        // digit, with ~100 distinct values per wave, that loop was 3.6x SLOWER than the plain adds.)
        for (int j0 = 0; j0 < N; j0 += 256) {
            const int j = j0 + tid;
            unsigned key = 0u; bool act = false;
            if (j < N) { key = wide_key(row[j], adaptive, mtq, bsq); act = pass == 0 || (key >> (shift + 8)) == prefix; }
            const unsigned bin = (key >> shift) & 255u;
            if (pass == 0) {
                unsigned long long todo = __ballot(act);
                while (todo) {                                                  // wave-uniform loop: one round per distinct digit
                    const int first = __ffsll((long long)todo) - 1;
                    const unsigned b0 = (unsigned)__builtin_amdgcn_readlane((int)bin, first);
                    const unsigned long long same = __ballot(act && bin == b0);
                    if ((tid & 63) == first) atomicAdd(&sh.hist[b0], (unsigned)__popcll(same));
                    todo &= ~same;
                }
            } else if (act) atomicAdd(&sh.hist[bin], 1u);
        }
        __syncthreads();
        if (tid == 0) {
            unsigned rem = sh.remaining, cum = 0; int bin = 0;
            for (int bq = 255; bq >= 0; --bq) {
                if (cum + sh.hist[bq] >= rem) { bin = bq; break; }
                cum += sh.hist[bq];
            }
            // (fewer than k keys in all: bin 0 is reached with cum + hist[0] = N >= rem, k <= N by construction)
            sh.remaining = rem - cum; sh.prefix = (prefix << 8) | (unsigned)bin; sh.bin_count = sh.hist[bin];
        }
        __syncthreads();
        bin_count = sh.bin_count;
    }
    T = sh.prefix; need = sh.remaining;
}

}  // namespace dagl
