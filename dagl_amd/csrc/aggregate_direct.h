// gather + weighted sum of one query's neighbour list, value rows read on the fly from the padded NHWC value map
#pragma once
#include "dagl_common.h"

namespace dagl {

constexpr int AGG_STAGE = 256;                                 // list entries staged at a time
// (the body lives here because overflow.hip runs it inside a launch of its own: ovf_scores_aggregate_kernel)
__device__ __forceinline__ void aggregate_direct_block(const AggArgs& a, int b, int q, int* sh_of /* [AGG_STAGE] */, float* sh_w /* [AGG_STAGE] */) {
    const int r = threadIdx.x;
    const int C4 = P / 4;                                      // 196
    const size_t ql = (size_t)b * a.g.L + q;
    const int n = a.nb_cnt[ql];
    const size_t lo = a.row_off ? (size_t)a.row_off[ql] : ql * a.width;
    const int32_t* ip = a.nb_idx + lo;
    const float* wp = a.nb_wgt + lo;
    const int W = a.g.W, Wp = a.g.Wp;
    const int rc = r < C4 ? r : C4 - 1;                        // (threads 196..255 stage entries, then idle along)
    const int kh = rc / 28, rem = rc % 28;
    const float4* vm = reinterpret_cast<const float4*>(a.b2p + (size_t)b * a.g.Hp * a.g.Wp * CH) + (size_t)kh * Wp * (CH / 4) + rem;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int base = 0; base < n; base += AGG_STAGE) {
        const int m = min(n - base, AGG_STAGE);
        if (base > 0) __syncthreads();
        for (int e = r; e < m; e += 256) {
            const int id = ip[base + e];
            const int jy = id / W, jx = id - jy * W;
            sh_of[e] = (jy * Wp + jx) * (CH / 4);
            sh_w[e] = wp[base + e];
        }
        __syncthreads();
        int j = 0;
        constexpr int AL = 16;
        for (; j + AL <= m; j += AL) {
            float w[AL]; float4 v[AL];
#pragma unroll
            for (int u = 0; u < AL; ++u) { v[u] = vm[sh_of[j + u]]; w[u] = sh_w[j + u]; }
#pragma unroll
            for (int u = 0; u < AL; ++u) {
                acc.x = fmaf(w[u], v[u].x, acc.x); acc.y = fmaf(w[u], v[u].y, acc.y);
                acc.z = fmaf(w[u], v[u].z, acc.z); acc.w = fmaf(w[u], v[u].w, acc.w);
            }
        }
        constexpr int AU = 8;
        for (; j < m; j += AU) {
            float w[AU]; float4 v[AU];
#pragma unroll
            for (int u = 0; u < AU; ++u) { const int e = min(j + u, m - 1); v[u] = vm[sh_of[e]]; w[u] = sh_w[e]; }
#pragma unroll
            for (int u = 0; u < AU; ++u) {
                const bool ok = j + u < m;
                acc.x = ok ? fmaf(w[u], v[u].x, acc.x) : acc.x; acc.y = ok ? fmaf(w[u], v[u].y, acc.y) : acc.y;
                acc.z = ok ? fmaf(w[u], v[u].z, acc.z) : acc.z; acc.w = ok ? fmaf(w[u], v[u].w, acc.w) : acc.w;
            }
        }
    }
    // (count -1: a query redone on its own, its row comes from overflow.hip)
    if (r < C4 && n >= 0) reinterpret_cast<float4*>(a.agg)[ql * C4 + r] = acc;
}

}  // namespace dagl
