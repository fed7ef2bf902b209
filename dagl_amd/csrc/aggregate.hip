// Edge softmax, neighbour gather + weighted sum, fold/normalise.
//
//   edge softmax   A_ij = mask_b * softmax_j(10 * S_ij * m_ij)         (DN_Gray/model/dagl.py:259-261)
//                  evaluated on the neighbour list only: every non-neighbour contributes exp(0) to the
//                  denominator (the reference does NOT renormalise after masking), so
//                  A_ij = e^{l_ij - M} / (sum_nb e^{l - M} + (N - deg) e^{-M}),  M = max(max_nb l, [deg<N] 0)
//   gather         agg_i = sum_j A_ij * V_j                              (torch.mm(yi, pi), dagl.py:263-264)
//                  V_j read on the fly from the padded NHWC value map (7 x 448-B segments per neighbour)
//   fold           out = fold(agg) / fold(unfold(1))                     (dagl.py:265-272)
#include "dagl_common.h"
#include "aggregate_direct.h"
#include "topk_merge.h"

namespace dagl {


// MODE 0: adaptive single-pass lists (<= 64 entries, unordered) -> sorted by key index, weights.
// One wave per query.
__global__ __launch_bounds__(256) void edge_softmax_fast_kernel(EdgeArgs a) {
    const int lane = threadIdx.x & 63;
    const size_t ql = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ql >= (size_t)a.B * a.L) return;
    int n = a.cnt[ql];
    if (n > DAGL_FAST_CAP) n = DAGL_FAST_CAP;                 // overflowing rows are redone by the CSR path
    int key = 0x7fffffff; float s = 0.f;
    if (lane < n) { key = a.list_idx[ql * DAGL_FAST_CAP + lane]; s = a.list_val[ql * DAGL_FAST_CAP + lane]; }
    // bitonic sort (ascending key) across the 64 lanes: the append order of the atomics is not
    // deterministic, the summation order downstream must be
#pragma unroll
    for (int k2 = 2; k2 <= 64; k2 <<= 1) {
#pragma unroll
        for (int j = k2 >> 1; j > 0; j >>= 1) {
            const int okey = __shfl_xor(key, j);
            const float os = __shfl_xor(s, j);
            const bool up = ((lane & k2) == 0);
            const bool lower = ((lane & j) == 0);
            const bool take = (lower == up) ? (okey < key) : (okey > key);
            if (take) { key = okey; s = os; }
        }
    }
    const float mtq = a.mt[ql], bsq = a.bs[ql];
    const bool valid = lane < n;
    const float lg = valid ? edge_logit(s, mtq, bsq, true) : 0.f;
    double M = wave_max_d(valid ? (double)lg : -1e300);
    if (n < a.N) M = fmax(M, 0.0);
    const double e = valid ? exp((double)lg - M) : 0.0;
    const double sum = wave_sum_d(e) + (double)(a.N - n) * exp(-M);
    if (valid) {
        a.nb_idx[ql * a.width + lane] = key;
        a.nb_wgt[ql * a.width + lane] = (float)(e / sum);
        if (a.nb_s != nullptr) a.nb_s[ql * a.width + lane] = s;
    }
    if (lane == 0) a.nb_cnt[ql] = n;
}

// MODE 1: CSR lists of any length (already in deterministic order): weights written at the same offsets.
__global__ __launch_bounds__(256) void edge_softmax_csr_kernel(EdgeArgs a) {
    const int lane = threadIdx.x & 63;
    const size_t ql = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ql >= (size_t)a.B * a.L) return;
    const int64_t o0 = a.row_off[ql], o1 = a.row_off[ql + 1];
    const int n = (int)(o1 - o0);
    const float mtq = a.mt[ql], bsq = a.bs[ql];
    double M = -1e300;
    for (int64_t o = o0 + lane; o < o1; o += 64) M = fmax(M, (double)edge_logit(a.list_val[o], mtq, bsq, true));
    M = wave_max_d(M);
    if (n < a.N) M = fmax(M, 0.0);
    double sum = 0.0;
    for (int64_t o = o0 + lane; o < o1; o += 64) sum += exp((double)edge_logit(a.list_val[o], mtq, bsq, true) - M);
    sum = wave_sum_d(sum) + (double)(a.N - n) * exp(-M);
    for (int64_t o = o0 + lane; o < o1; o += 64) {
        a.nb_idx[o] = a.list_idx[o];
        a.nb_wgt[o] = (float)(exp((double)edge_logit(a.list_val[o], mtq, bsq, true) - M) / sum);
    }
    if (lane == 0) a.nb_cnt[ql] = n;
}

// one block = 4 queries; as the redo pass behind the screen (run_flags given) a small grid walks over all of them and
// skips the unflagged ones (see score_select_kernel)
__global__ __launch_bounds__(256) void edge_softmax_topk_kernel(EdgeArgs a, int kslots, unsigned n_blocks) {
    __shared__ float cv[4][TOPK_MAX_CAND];
    __shared__ int ci[4][TOPK_MAX_CAND];
    if (a.run_flags == nullptr) { edge_softmax_topk_unit(a, kslots, cv, ci, blockIdx.x); return; }
    if (a.run_count != nullptr && *a.run_count == 0) return;           // nothing flagged anywhere (the usual case)
    const int nqg = (a.L + 127) / 128;
    for (size_t blk = blockIdx.x; blk < n_blocks; blk += gridDim.x) {
        // block-uniform skip: none of the block's 4 queries sits in a flagged group
        const size_t q0 = blk * 4, q1 = (q0 + 3 < (size_t)a.B * a.L - 1) ? q0 + 3 : (size_t)a.B * a.L - 1;
        const size_t b0 = q0 / a.L, b1 = q1 / a.L;
        if (a.run_flags[b0 * nqg + (q0 - b0 * a.L) / 128] == 0 && a.run_flags[b1 * nqg + (q1 - b1 * a.L) / 128] == 0) continue;
        edge_softmax_topk_unit(a, kslots, cv, ci, blk);
    }
}

// per-query degree and softmax mass (parity tests: the reference's mask_b.sum(1) and A.sum(1))
__global__ void row_stats_kernel(size_t n_rows, const float* __restrict__ nb_wgt, const int32_t* __restrict__ nb_cnt,
                                 const int64_t* __restrict__ row_off, int width, int32_t* __restrict__ deg,
                                 float* __restrict__ rowsum) {
    const size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows) return;
    const int n = nb_cnt[r];
    if (n < 0) return;                                   // a query redone on its own: overflow.hip has written both already
    const float* w = nb_wgt + (row_off ? (size_t)row_off[r] : r * width);
    double s = 0.0;
    for (int j = 0; j < n; ++j) s += (double)w[j];
    if (deg) deg[r] = n;
    if (rowsum) rowsum[r] = (float)s;
}

int launch_row_stats(hipStream_t s, size_t n_rows, const float* nb_wgt, const int32_t* nb_cnt, const int64_t* row_off,
                     int width, int32_t* deg, float* rowsum) {
    hipLaunchKernelGGL(row_stats_kernel, dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0, s, n_rows, nb_wgt,
                       nb_cnt, row_off, width, deg, rowsum);
    DAGL_LAUNCH_CHECK("row_stats_kernel");
    return DAGL_OK;
}

int launch_edge_softmax(hipStream_t s, const EdgeArgs& a) {
    const size_t nq = (size_t)a.B * a.L;
    dim3 grid((unsigned)((nq + 3) / 4)), block(256);
    if (a.mode == DAGL_MODE_ADAPTIVE) {
        if (a.row_off == nullptr) hipLaunchKernelGGL(edge_softmax_fast_kernel, grid, block, 0, s, a);
        else hipLaunchKernelGGL(edge_softmax_csr_kernel, grid, block, 0, s, a);
    } else {
        const int ks = topk_slots(a.k);
        if (a.splits * 2 * ks > TOPK_MAX_CAND) {
            set_error("edge softmax: %d candidates per query exceed %d", a.splits * 2 * ks, TOPK_MAX_CAND);
            return DAGL_ERR_INVALID;
        }
        dim3 g2 = grid;
        if (a.run_flags != nullptr && g2.x > 128) g2.x = 128;            // redo pass: blocks walk over the queries
        hipLaunchKernelGGL(edge_softmax_topk_kernel, g2, block, 0, s, a, ks, grid.x);
    }
    DAGL_LAUNCH_CHECK("edge_softmax_kernel");
    return DAGL_OK;
}

// ------------------------------------------------------------------------------------------------------
// gather + weighted sum, value rows read on the fly from the padded NHWC value map
// ------------------------------------------------------------------------------------------------------
// block = one query, thread = float4 column r of its 784-float patch row: r = kh*28 + (kw*4 + c4).  The block first turns
// the list into map offsets in LDS (one key -> (row, column) division per ENTRY instead of one per entry and thread, and
// no global index load in front of every gather); the gather loop then reads offset and weight from LDS, so the
// compiler can keep a whole batch of gathers in flight.  Variable-length lists only -- the top-k modes take
// aggregate_fold_kernel.  A few queries of a long-tailed degree distribution come close to their 256 slots and set the
// kernel's duration: batches of 16 first, then batches of 8 with the slots past the list's end predicated off (clamped
// reads, no fma).  The fma order is the list order.
__global__ __launch_bounds__(256) void aggregate_direct_kernel(AggArgs a) {
    __shared__ int sh_of[AGG_STAGE];
    __shared__ float sh_w[AGG_STAGE];
    aggregate_direct_block(a, blockIdx.y, blockIdx.x, sh_of, sh_w);
}

int launch_aggregate_direct(hipStream_t s, const AggArgs& a) {
    dim3 grid((unsigned)a.g.L, a.B), block(256);
    hipLaunchKernelGGL(aggregate_direct_kernel, grid, block, 0, s, a);
    DAGL_LAUNCH_CHECK("aggregate_direct_kernel");
    return DAGL_OK;
}

// ------------------------------------------------------------------------------------------------------
// stand-alone gather over materialised value rows (the roofline-graded form: k*P*4 + k*8 + P*4 B/query)
// ------------------------------------------------------------------------------------------------------
template <int KU>
__global__ __launch_bounds__(256) void gather_rows_kernel(int L, int k, int P4, const int32_t* __restrict__ idx,
                                                          const float* __restrict__ wgt,
                                                          const float4* __restrict__ values,
                                                          float4* __restrict__ out) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)L * P4) return;
    const int q = (int)(t / P4), r = (int)(t % P4);
    const int32_t* ip = idx + (size_t)q * k;
    const float* wp = wgt + (size_t)q * k;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    int j = 0;
    for (; j + KU <= k; j += KU) {
        int id[KU]; float w[KU]; float4 v[KU];
#pragma unroll
        for (int u = 0; u < KU; ++u) { id[u] = ip[j + u]; w[u] = wp[j + u]; }
#pragma unroll
        for (int u = 0; u < KU; ++u) {
            const bool ok = id[u] >= 0;
            v[u] = values[(size_t)(ok ? id[u] : 0) * P4 + r];
            if (!ok) w[u] = 0.f;
        }
#pragma unroll
        for (int u = 0; u < KU; ++u) {
            acc.x = fmaf(w[u], v[u].x, acc.x); acc.y = fmaf(w[u], v[u].y, acc.y);
            acc.z = fmaf(w[u], v[u].z, acc.z); acc.w = fmaf(w[u], v[u].w, acc.w);
        }
    }
    for (; j < k; ++j) {
        const int id = ip[j];
        if (id < 0) continue;
        const float w = wp[j];
        const float4 v = values[(size_t)id * P4 + r];
        acc.x = fmaf(w, v.x, acc.x); acc.y = fmaf(w, v.y, acc.y);
        acc.z = fmaf(w, v.z, acc.z); acc.w = fmaf(w, v.w, acc.w);
    }
    out[t] = acc;
}

int launch_gather_fixed(hipStream_t s, int L, int k, int P_, const int32_t* idx, const float* wgt,
                        const float* values, float* out) {
    const int P4 = P_ / 4;
    const size_t items = (size_t)L * P4;
    dim3 grid((unsigned)((items + 255) / 256)), block(256);
    if (k % 8 == 0)
        hipLaunchKernelGGL(gather_rows_kernel<8>, grid, block, 0, s, L, k, P4, idx, wgt,
                           reinterpret_cast<const float4*>(values), reinterpret_cast<float4*>(out));
    else
        hipLaunchKernelGGL(gather_rows_kernel<4>, grid, block, 0, s, L, k, P4, idx, wgt,
                           reinterpret_cast<const float4*>(values), reinterpret_cast<float4*>(out));
    DAGL_LAUNCH_CHECK("gather_rows_kernel");
    return DAGL_OK;
}

// ------------------------------------------------------------------------------------------------------
// fold + overlap normalisation
// ------------------------------------------------------------------------------------------------------
// Query (r,c)'s aggregated 7x7 patch lands with its top-left corner at (4r-3, 4c-3) (fold grid: kernel 7,
// padding 3, stride 4 -- dagl.py:266), i.e. NOT where it was read (SAME grid, top-left 4r-pt).  A pixel is
// covered by <= 2 x 2 windows; the divisor fold(unfold(1)) (dagl.py:268-270) is that window count.
__global__ __launch_bounds__(256) void fold_kernel(Grid g, const float* __restrict__ agg, float* __restrict__ out,
                                                   int imgs, int heads, RangeTag range) {
    const int b = blockIdx.z;
    const int y = blockIdx.y;
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    // range guard: a kernel of THIS call met a value outside the split-fp16 range -> the numbers are worthless: NaN
    const bool poisoned = (range.word != nullptr && *range.word == range.tag) || (range.veto != nullptr && *range.veto == range.tag);
    if (range.done != nullptr && b == 0 && y == 0 && blockIdx.x == 0 && threadIdx.x == 0) *range.done = range.tag;
    if (x >= g.W) return;
    // rows r with 4r-3 <= y <= 4r+3
    int r0 = (y - 3 + QS - 1) / QS; if (y - 3 < 0) r0 = 0;
    int r1 = (y + 3) / QS; if (r1 > g.Lh - 1) r1 = g.Lh - 1;
    int c0 = (x - 3 + QS - 1) / QS; if (x - 3 < 0) c0 = 0;
    int c1 = (x + 3) / QS; if (c1 > g.Lw - 1) c1 = g.Lw - 1;
    float4 acc[CH / 4];
#pragma unroll
    for (int u = 0; u < CH / 4; ++u) acc[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int r = r0; r <= r1; ++r) {
        const int kh = y - (QS * r - 3);
        for (int c = c0; c <= c1; ++c) {
            const int kw = x - (QS * c - 3);
            const float4* p = reinterpret_cast<const float4*>(
                agg + ((size_t)b * g.L + (size_t)r * g.Lw + c) * P + (kh * KS + kw) * CH);
#pragma unroll
            for (int u = 0; u < CH / 4; ++u) {
                const float4 v = p[u];
                acc[u].x += v.x; acc[u].y += v.y; acc[u].z += v.z; acc[u].w += v.w;
            }
        }
    }
    float cnt = (float)((r1 - r0 + 1) * (c1 - c0 + 1));
    if (poisoned) cnt = __builtin_nanf("");
    const int head = b / imgs, img = b - head * imgs;
    float* o = out + ((size_t)img * heads + head) * CH * g.N + (size_t)y * g.W + x;
#pragma unroll
    for (int u = 0; u < CH / 4; ++u) {
        o[(size_t)(4 * u + 0) * g.N] = acc[u].x / cnt;
        o[(size_t)(4 * u + 1) * g.N] = acc[u].y / cnt;
        o[(size_t)(4 * u + 2) * g.N] = acc[u].z / cnt;
        o[(size_t)(4 * u + 3) * g.N] = acc[u].w / cnt;
    }
}

// ------------------------------------------------------------------------------------------------------
// gather + weighted sum + fold in one pass (short fixed-width lists: the top-k modes)
// ------------------------------------------------------------------------------------------------------
// thread = (pixel, 4 channels).  For each of the <= 2 x 2 queries whose 7x7 window covers the pixel it forms that query's
// aggregated value at the pixel's place in the window -- the same fma chain over the neighbours, in list order, that
// aggregate_direct_kernel runs -- and adds the windows in fold_kernel's order: the result is bit-identical to the two
// kernels, without the [L,784] rows ever going to memory (12.8 MB written and read back at 256^2) and one launch less.
// key -> (row, column) of the map: every thread does this for every neighbour of up to four queries (32 per thread, 8.4 M per
// launch at 256^2), and an integer division by a run-time W is ~25 instructions: float reciprocal + one correction step instead
// (exact: keys < 2^24, the product is off by less than one)
__device__ __forceinline__ void key_row_col(int id, int W, float inv_w, int& jy, int& jx) {
    jy = (int)((float)id * inv_w);
    jx = id - jy * W;
    if (jx < 0) { --jy; jx += W; } else if (jx >= W) { ++jy; jx -= W; }
}

__global__ __launch_bounds__(256) void aggregate_fold_kernel(AggArgs a, float* __restrict__ out, int imgs, int heads,
                                                             RangeTag range) {
    const Grid& g = a.g;
    const int b = blockIdx.z;
    // pixels in row-major order over the whole map (a block per row left 44 % of the lanes idle on a 72-pixel leaf tile)
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int pix = t >> 2, u = t & 3;
    bool poisoned = range.word != nullptr && *range.word == range.tag;
    if (range.done != nullptr && b == 0 && blockIdx.x == 0 && threadIdx.x == 0) *range.done = range.tag;
    if (a.unserved != nullptr && *a.unserved != 0) {        // DAGL_FLAG_NO_REDO: flagged query groups and no redo pass behind them
        poisoned = true;
        if (b == 0 && blockIdx.x == 0 && threadIdx.x == 0) *a.unserved_sticky = 1;
    }
    if (pix >= g.N) return;
    const int y = pix / g.W, x = pix - y * g.W;
    int r0 = (y - 3 + QS - 1) / QS; if (y - 3 < 0) r0 = 0;
    int r1 = (y + 3) / QS; if (r1 > g.Lh - 1) r1 = g.Lh - 1;
    int c0 = (x - 3 + QS - 1) / QS; if (x - 3 < 0) c0 = 0;
    int c1 = (x + 3) / QS; if (c1 > g.Lw - 1) c1 = g.Lw - 1;
    const float4* vm = reinterpret_cast<const float4*>(a.b2p + (size_t)b * g.Hp * g.Wp * CH) + u;
    const int W = g.W, Wp = g.Wp;
    const float inv_w = 1.0f / (float)W;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int r = r0; r <= r1; ++r) {
        const int kh = y - (QS * r - 3);
        for (int c = c0; c <= c1; ++c) {
            const int kw = x - (QS * c - 3);
            const size_t ql = (size_t)b * g.L + (size_t)r * g.Lw + c;
            const int n = a.nb_cnt[ql];
            const int32_t* ip = a.nb_idx + ql * a.width;
            const float* wp = a.nb_wgt + ql * a.width;
            float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
            int j = 0;
            for (; j + 4 <= n; j += 4) {
                int id[4]; float w[4]; float4 v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) { id[e] = ip[j + e]; w[e] = wp[j + e]; }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    int jy, jx; key_row_col(id[e], W, inv_w, jy, jx);
                    v[e] = vm[((jy + kh) * Wp + jx + kw) * (CH / 4)];
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    q.x = fmaf(w[e], v[e].x, q.x); q.y = fmaf(w[e], v[e].y, q.y);
                    q.z = fmaf(w[e], v[e].z, q.z); q.w = fmaf(w[e], v[e].w, q.w);
                }
            }
            for (; j < n; ++j) {
                const int id = ip[j]; const float w = wp[j];
                int jy, jx; key_row_col(id, W, inv_w, jy, jx);
                const float4 v = vm[((jy + kh) * Wp + jx + kw) * (CH / 4)];
                q.x = fmaf(w, v.x, q.x); q.y = fmaf(w, v.y, q.y);
                q.z = fmaf(w, v.z, q.z); q.w = fmaf(w, v.w, q.w);
            }
            acc.x += q.x; acc.y += q.y; acc.z += q.z; acc.w += q.w;
        }
    }
    float cnt = (float)((r1 - r0 + 1) * (c1 - c0 + 1));
    if (poisoned) cnt = __builtin_nanf("");
    const int head = b / imgs, img = b - head * imgs;
    float* o = out + ((size_t)img * heads + head) * CH * g.N + (size_t)y * g.W + x;
    o[(size_t)(4 * u + 0) * g.N] = acc.x / cnt;
    o[(size_t)(4 * u + 1) * g.N] = acc.y / cnt;
    o[(size_t)(4 * u + 2) * g.N] = acc.z / cnt;
    o[(size_t)(4 * u + 3) * g.N] = acc.w / cnt;
}

int launch_aggregate_fold(hipStream_t s, const AggArgs& a, float* out, int heads, RangeTag range) {
    dim3 grid((unsigned)(((size_t)a.g.N * 4 + 255) / 256), 1, a.B), block(256);
    hipLaunchKernelGGL(aggregate_fold_kernel, grid, block, 0, s, a, out, a.B / heads, heads, range);
    DAGL_LAUNCH_CHECK("aggregate_fold_kernel");
    return DAGL_OK;
}

int launch_fold(hipStream_t s, int B, const Grid& g, const float* agg, float* out, int heads, RangeTag range) {
    dim3 grid((g.W + 63) / 64, g.H, B), block(64);
    hipLaunchKernelGGL(fold_kernel, grid, block, 0, s, g, agg, out, B / heads, heads, range);
    DAGL_LAUNCH_CHECK("fold_kernel");
    return DAGL_OK;
}

// ------------------------------------------------------------------------------------------------------
// stage mix: out = conv1x1(cat(4 heads)) + x   (CES.forward, DN_Gray/model/dagl.py:114,116,118)
// ------------------------------------------------------------------------------------------------------
// fp32 MFMA 16x16x4: one wave = 64 outputs x MIX_TILES tiles of 16 pixels; A = the 64x64 mix weights in registers (loaded
// once per wave: with one tile per wave the 64 weight loads per lane outweighed the 16 pixel loads), B[channel][pixel]
// straight from the NCHW concat map (16 consecutive pixels of a channel plane = one 64-byte run, the wave's tiles are
// consecutive: whole lines get used); the result tile has pixels along the lanes, so the NCHW stores are 64-byte runs too.
constexpr int MIX_TILES = 4;
__global__ __launch_bounds__(256) void stage_mix_kernel(int HW, const float* __restrict__ cat, const float* __restrict__ x,
                                                        const float* __restrict__ mix_w, const float* __restrict__ mix_b,
                                                        float* __restrict__ out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, g = lane >> 4;
    const int b = blockIdx.y;
    const int pw = (blockIdx.x * 4 + wave) * (16 * MIX_TILES);
    if (pw >= HW) return;
    float w[4][16];                                            // B fragment (n, T): w[o = n*16 + i][c = 4T + g]
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int T = 0; T < 16; ++T) w[n][T] = mix_w[(n * 16 + i) * 64 + 4 * T + g];
    float bo[4][4];
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) bo[n][r] = mix_b[n * 16 + 4 * g + r];
    float av[MIX_TILES][16];
#pragma unroll
    for (int t = 0; t < MIX_TILES; ++t) {
        int px = pw + 16 * t + i; if (px >= HW) px = HW - 1;
        const float* cb = cat + (size_t)b * 64 * HW + px;
#pragma unroll
        for (int T = 0; T < 16; ++T) av[t][T] = cb[(size_t)(4 * T + g) * HW];
    }
#pragma unroll
    for (int t = 0; t < MIX_TILES; ++t) {
        f32x4 acc[4];
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int T = 0; T < 16; ++T)
#pragma unroll
            for (int n = 0; n < 4; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[n][T], av[t][T], acc[n], 0, 0, 0);
        // D[row = output n*16 + 4g + r][col = pixel]
        const int pp = pw + 16 * t + i;
#pragma unroll
        for (int n = 0; n < 4; ++n) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int o = n * 16 + 4 * g + r;
                if (pp < HW) {
                    const size_t idx = ((size_t)b * 64 + o) * HW + pp;
                    out[idx] = (acc[n][r] + bo[n][r]) + x[idx];
                }
            }
        }
    }
}

int launch_stage_mix(hipStream_t s, int B, int HW, const float* cat, const float* x, const float* mix_w,
                     const float* mix_b, float* out) {
    dim3 grid((HW + 64 * MIX_TILES - 1) / (64 * MIX_TILES), B), block(256);
    hipLaunchKernelGGL(stage_mix_kernel, grid, block, 0, s, HW, cat, x, mix_w, mix_b, out);
    DAGL_LAUNCH_CHECK("stage_mix_kernel");
    return DAGL_OK;
}

}  // namespace dagl
