// One query's whole score row in the reference's dense form (DN_Gray/model/dagl.py:250-264): mask, softmax over all N keys
// (masked keys count e^0), weighted sum of the value patches straight from the value map -- cut into key chunks that run in
// parallel with softmax statistics of their own and are combined afterwards.  Shared by the per-query redo of the adaptive
// mode (overflow.hip) and the fixed-k neighbourhoods wider than the lists (topk_wide.hip).
#pragma once
#include "dagl_common.h"

namespace dagl {

constexpr int ROW_CHUNKS = 32;                                 // key chunks a row is cut into (one block each)
constexpr int ROW_PART_FLOATS = P + 8;                         // partial weighted sum (784) + {max logit, count, z (double), -}

__device__ __forceinline__ void row_chunk_range(int N, int chunk, int& j0, int& j1) {
    const int per = ((N + ROW_CHUNKS - 1) / ROW_CHUNKS + 255) / 256 * 256;
    j0 = chunk * per; j1 = j0 + per;
    if (j0 > N) j0 = N;
    if (j1 > N) j1 = N;
}
// the quarter of a chunk wave w of a block walks
__device__ __forceinline__ void row_wave_range(int j0c, int j1c, int w, int& j0, int& j1) {
    const int per_wave = ((j1c - j0c + 3) / 4 + 63) / 64 * 64;
    j0 = min(j0c + w * per_wave, j1c);
    j1 = (j0 + per_wave < j1c) ? j0 + per_wave : j1c;
}

// lane l of a wave owns the float4 columns l + 64 u of the 784-float row
struct RowCols { int kh[4], rem[4]; bool cv[4]; };
__device__ __forceinline__ RowCols row_cols(int lane) {
    RowCols c;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int r = lane + 64 * u;
        c.cv[u] = r < P / 4;
        const int rc = c.cv[u] ? r : 0;
        c.kh[u] = rc / 28; c.rem[u] = rc % 28;
    }
    return c;
}

// a wave's running result over its keys: weights are taken against the running maximum m of the keys seen so far
struct RowAcc { float4 acc[4]; float m; double z; int cnt; };
__device__ __forceinline__ void row_acc_clear(RowAcc& o) {
#pragma unroll
    for (int u = 0; u < 4; ++u) o.acc[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    o.m = -1.f; o.z = 0.0; o.cnt = 0;
}
__device__ __forceinline__ void row_take(const Grid& g, const float4* vmb, const RowCols& c, int key, float wv, RowAcc& o) {
    const int jy = key / g.W, jx = key - jy * g.W;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        if (!c.cv[u]) continue;
        const float4 v = vmb[((size_t)(jy + c.kh[u]) * g.Wp + jx) * (CH / 4) + c.rem[u]];
        o.acc[u].x = fmaf(wv, v.x, o.acc[u].x); o.acc[u].y = fmaf(wv, v.y, o.acc[u].y);
        o.acc[u].z = fmaf(wv, v.z, o.acc[u].z); o.acc[u].w = fmaf(wv, v.w, o.acc[u].w);
    }
}
// keys [j0, j1) in steps of 64: `eval(j, in_range, l)` says whether key j passes and gives its logit (>= 0); it is called by all
// 64 lanes together (it may use wave-wide operations).  The passing keys of a step are taken in ascending order, two at a time
// so that their loads are in flight together.  gather = false: statistics only.
template <typename EvalFn>
__device__ __forceinline__ void row_wave_walk(const Grid& g, const float4* vmb, const RowCols& c, int lane, int j0, int j1, bool gather,
                                              EvalFn eval, RowAcc& o) {
    for (int c0 = j0; c0 < j1; c0 += 64) {
        const int j = c0 + lane;
        float l = 0.f;
        const bool pass = eval(j, j < j1, l);
        unsigned long long bal = __ballot(pass);
        if (!bal) continue;
        o.cnt += __popcll(bal);
        const float mx = wave_max_f32(pass ? l : -1.f);
        if (mx > o.m) {                                              // wave-uniform: what has been summed so far shrinks
            const float sc = (o.m < 0.f) ? 0.f : expf(o.m - mx);
#pragma unroll
            for (int u = 0; u < 4; ++u) { o.acc[u].x *= sc; o.acc[u].y *= sc; o.acc[u].z *= sc; o.acc[u].w *= sc; }
            o.z *= (double)sc;
            o.m = mx;
        }
        const float wgt = pass ? expf(l - o.m) : 0.f;
        if (!gather) continue;
        while (bal) {
            const int b0 = __ffsll((long long)bal) - 1; bal &= bal - 1;
            const float w0 = __shfl(wgt, b0);
            if (bal) {
                const int b1 = __ffsll((long long)bal) - 1; bal &= bal - 1;
                const float w1 = __shfl(wgt, b1);
                row_take(g, vmb, c, c0 + b0, w0, o); row_take(g, vmb, c, c0 + b1, w1, o);
                o.z += (double)w0; o.z += (double)w1;
            } else {
                row_take(g, vmb, c, c0 + b0, w0, o);
                o.z += (double)w0;
            }
        }
    }
}

// the four waves' results of one (row, chunk), brought to the block's maximum, added in wave order and stored:
// pr[0 .. 783] partial row (only when keys passed), pr[784] = m, [785] = count, [786..787] = z (double)
struct RowBlockShared { float4 part[4][P / 4]; double z[4]; float m[4]; };            // 12.3 KiB
__device__ __forceinline__ void row_block_store(RowBlockShared& sh, const RowCols& c, const RowAcc& o, int cnt_blk, float* pr) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (lane == 0) { sh.m[w] = o.m; sh.z[w] = o.z; }
#pragma unroll
    for (int u = 0; u < 4; ++u) if (c.cv[u]) sh.part[w][lane + 64 * u] = o.acc[u];
    __syncthreads();
    const float mb = fmaxf(fmaxf(sh.m[0], sh.m[1]), fmaxf(sh.m[2], sh.m[3]));
    float scw[4];
#pragma unroll
    for (int ww = 0; ww < 4; ++ww) scw[ww] = (sh.m[ww] < 0.f) ? 0.f : expf(sh.m[ww] - mb);
    if (tid < P / 4 && cnt_blk > 0) {
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int ww = 0; ww < 4; ++ww) {
            const float4 v = sh.part[ww][tid];
            t.x = fmaf(scw[ww], v.x, t.x); t.y = fmaf(scw[ww], v.y, t.y); t.z = fmaf(scw[ww], v.z, t.z); t.w = fmaf(scw[ww], v.w, t.w);
        }
        reinterpret_cast<float4*>(pr)[tid] = t;
    }
    if (tid == 0) {
        double z = 0.0;
        for (int ww = 0; ww < 4; ++ww) z += sh.z[ww] * (double)scw[ww];
        pr[P] = mb; reinterpret_cast<int*>(pr + P)[1] = cnt_blk; *reinterpret_cast<double*>(pr + P + 2) = z;
    }
    __syncthreads();
}

// A row from its chunks (256 threads; fixed-order sums): M = largest logit (0 joins in when a key is masked), degree,
// Z = sum of z_c e^(m_c - M) + (N - degree) e^(-M)  (masked keys count e^0 each, dagl.py:259-261).  Afterwards
// sh_scale[c] = e^(m_c - M) (0: no passing key in chunk c, or all of it underflows).
struct RowSum { double M, Z, zs; int deg; };
struct RowReduceShared { float scale[256]; double d[256]; int i[256]; };
__device__ __forceinline__ RowSum row_reduce(const float* part_row /* [ROW_CHUNKS][ROW_PART_FLOATS] */, int N, RowReduceShared& sh) {
    const int tid = threadIdx.x;
    float m = -1.f; int cnt = 0; double z = 0.0;
    if (tid < ROW_CHUNKS) {
        const float* pr = part_row + (size_t)tid * ROW_PART_FLOATS + P;
        m = pr[0]; cnt = reinterpret_cast<const int*>(pr)[1]; z = *reinterpret_cast<const double*>(pr + 2);
    }
    __syncthreads();                                                             // (the arrays' previous use)
    sh.scale[tid] = m; sh.i[tid] = cnt;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {                                       // max and integer sum: order-free
        if (tid < st) { sh.scale[tid] = fmaxf(sh.scale[tid], sh.scale[tid + st]); sh.i[tid] += sh.i[tid + st]; }
        __syncthreads();
    }
    RowSum r;
    r.deg = sh.i[0];
    r.M = (double)sh.scale[0];
    if (r.deg < N) r.M = fmax(r.M, 0.0);
    __syncthreads();
    const float sc = (m < 0.f) ? 0.f : (float)exp((double)m - r.M);
    sh.scale[tid] = sc;
    sh.d[tid] = z * (double)sc;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {                                       // fixed tree: the same sum on every run
        if (tid < st) sh.d[tid] += sh.d[tid + st];
        __syncthreads();
    }
    r.zs = sh.d[0];
    r.Z = r.zs + (double)(N - r.deg) * exp(-r.M);
    return r;
}
// ... and the row itself: sum of the chunks' partial rows, scaled, in chunk order, / Z  (thread = float4 column, tid < 196)
__device__ __forceinline__ float4 row_combine(const float* part_row, const RowSum& r, const RowReduceShared& sh) {
    const int tid = threadIdx.x;
    const float inv = (float)(1.0 / r.Z);
    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int c = 0; c < ROW_CHUNKS; ++c) {
        const float sc = sh.scale[c];
        if (sc == 0.f) continue;                                                 // block-uniform
        const float4 v = reinterpret_cast<const float4*>(part_row + (size_t)c * ROW_PART_FLOATS)[tid];
        t.x = fmaf(sc, v.x, t.x); t.y = fmaf(sc, v.y, t.y); t.z = fmaf(sc, v.z, t.z); t.w = fmaf(sc, v.w, t.w);
    }
    t.x *= inv; t.y *= inv; t.z *= inv; t.w *= inv;
    return t;
}

}  // namespace dagl
