// CE.forward for ANY patch geometry -- ksize, stride_1, stride_2 and inter_channels are constructor arguments of the reference
// block (DN_Gray/model/dagl.py:175-176) that its own builders never override (dagl.py:94-109), so every tuned kernel of this
// library has (7, 4, 1, 16) compiled in.  This file is the run-time-geometry route behind the same module: the reference's dense
// formulation (dagl.py:207-275) stage by stage on the fp32 matrix cores, nothing O(L N) beyond one chunk of score rows:
//
//   dagl.py:208-215   g (3x3) + theta (1x1 = centre tap) as ONE 2c-output product over the 3x3 patches of the zero-bordered NHWC
//                     input; thr / bias as one 2-output product over the stride_1 SAME ksize x ksize patches
//   dagl.py:216-243   unfold_patches_kernel (train_ops.hip) on the maps, any window / stride / channel count (multiple of 4)
//   dagl.py:248-249   rows x fc^T + bias, ReLU (gemm32.hip; Linear weights re-ordered once from Unfold's (c,kh,kw) to (kh,kw,c))
//   dagl.py:250       S chunk = Wq chunk x X^T (gemm32.hip), rows of at most 256 MiB at a time
//   dagl.py:256-261   gen_row_softmax_kernel: one block per query row -- row mean (fp64 sum), mask, softmax over ALL keys with the
//                     masked keys' e^0 in the denominator, not renormalised; the fixed-k variant's k best by radix selection
//                     (wide_select.h, ties to the lower key index), GReccR2b_3mh_1-checkpoint.py:242-250
//   dagl.py:263-264   A chunk x value rows (gemm32.hip); fixed-k modes with k <= 64: the row kernel writes (key, weight) lists in key order
//                     (ballot compaction, no atomics) and the sum is a gather over k value rows (gather_rows_kernel, aggregate.hip)
//   dagl.py:265-272   gen_fold_normalize_kernel: fold with padding = the stride_2 SAME grid's left pad and stride_1 (the reference
//                     folds query patches cut with the stride_1 SAME pad back with the stride_2 pad: reproduced), overlap count
//                     with its zero guard, NCHW out
//
// Correct and on the device, not tuned: the default geometry never comes here (dagl_amd/ce.py routes it to the tuned kernels).
#include "dagl_common.h"
#include "wide_select.h"

namespace dagl {

int launch_unfold_patches(hipStream_t s, int B, int Hp, int Wp, int C, int k, int stride, int oy, int ox, int oh, int ow,
                          const float* map, float* rows);
int launch_gather_fixed(hipStream_t s, int L, int k, int P_, const int32_t* idx, const float* wgt, const float* values, float* out);

namespace {

struct GenGeom {
    int B, Cin, H, W, ks, s1, s2, C;
    int PG, Hp, Wp;                 // border of the NHWC maps, padded extent
    int Lh, Lw, L, t1, l1;          // query grid (stride_1 SAME) and its top / left pad
    int Nh, Nw, N, t2, l2;          // key / value grid (stride_2 SAME)
    int P, D;                       // patch length ks*ks*C, feature length P / 4
    int fold_pad, fold_h, fold_w;   // F.fold's padding (dagl.py:243: paddings[0] = left pad of the stride_2 grid) and block grid
    long long ldn; int Lc;          // score-row stride (N rounded up to 4) and rows per chunk
};

inline void same_pad_1d(int n, int k, int stride, int& out, int& lead) {
    out = (n + stride - 1) / stride;
    int p = (out - 1) * stride + k - n; if (p < 0) p = 0;
    lead = p / 2;                                   // dagl.py:131-136: the odd unit goes to the bottom / right
}

inline GenGeom gen_geom(int B, int Cin, int H, int W, int ks, int s1, int s2, int C) {
    GenGeom g{};
    g.B = B; g.Cin = Cin; g.H = H; g.W = W; g.ks = ks; g.s1 = s1; g.s2 = s2; g.C = C;
    g.PG = ks > 2 ? ks - 1 : 1;
    g.Hp = H + 2 * g.PG; g.Wp = W + 2 * g.PG;
    same_pad_1d(H, ks, s1, g.Lh, g.t1); same_pad_1d(W, ks, s1, g.Lw, g.l1);
    same_pad_1d(H, ks, s2, g.Nh, g.t2); same_pad_1d(W, ks, s2, g.Nw, g.l2);
    g.L = g.Lh * g.Lw; g.N = g.Nh * g.Nw;
    g.P = ks * ks * C; g.D = g.P / 4;
    g.fold_pad = g.l2;                              // dagl.py:243 takes paddings[0] (the LEFT pad) for both axes
    g.fold_h = (H + 2 * g.fold_pad - ks) / s1 + 1;
    g.fold_w = (W + 2 * g.fold_pad - ks) / s1 + 1;
    g.ldn = ((long long)g.N + 3) / 4 * 4;
    long long lc = (64ll << 20) / g.ldn;            // 256 MiB of score rows
    if (lc < 1) lc = 1;
    if (lc > g.L) lc = g.L;
    g.Lc = (int)lc;
    return g;
}

constexpr int GEN_LIST_CAP_WORDS = 64;              // = GEN_LIST_MAX: (key, weight) list slots per row behind a chunk's score rows
struct GenPlan {
    size_t o_xp, o_wgt, o_bgt, o_wtb, o_btb, o_fc1, o_fc2, o_rows, o_y, o_b1p, o_b2p, o_tb, o_wq, o_x, o_s, o_agg, o_end;
};
inline GenPlan gen_plan(const GenGeom& g) {
    GenPlan p{};
    size_t off = 0;
    auto carve = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
    const size_t B = g.B, HW = (size_t)g.H * g.W, pix = (size_t)g.Hp * g.Wp;
    p.o_xp = carve(B * pix * g.Cin * 4);
    p.o_wgt = carve((size_t)2 * g.C * 9 * g.Cin * 4);
    p.o_bgt = carve((size_t)2 * g.C * 4);
    p.o_wtb = carve((size_t)2 * g.ks * g.ks * g.Cin * 4);
    p.o_btb = carve(8);
    p.o_fc1 = carve((size_t)g.D * g.P * 4);
    p.o_fc2 = carve((size_t)g.D * g.P * 4);
    // one row buffer serves every unfold in turn: 3x3 input patches, thr / bias patches, query / key / value patches
    size_t rows = B * HW * 9 * g.Cin;
    const size_t r_tb = B * g.L * (size_t)g.ks * g.ks * g.Cin, r_q = B * g.L * (size_t)g.P, r_k = B * g.N * (size_t)g.P;
    if (r_tb > rows) rows = r_tb;
    if (r_q > rows) rows = r_q;
    if (r_k > rows) rows = r_k;
    p.o_rows = carve(rows * 4);
    p.o_y = carve(B * HW * 2 * g.C * 4);
    p.o_b1p = carve(B * pix * g.C * 4);
    p.o_b2p = carve(B * pix * g.C * 4);
    p.o_tb = carve(B * g.L * 2 * 4);
    p.o_wq = carve(B * g.L * (size_t)g.D * 4);
    p.o_x = carve(B * g.N * (size_t)g.D * 4);
    p.o_s = carve((size_t)g.Lc * (g.ldn + 2 * GEN_LIST_CAP_WORDS) * 4);
    p.o_agg = carve(B * g.L * (size_t)g.P * 4);
    p.o_end = off;
    return p;
}

// NCHW [B,C,H,W] -> zero-bordered NHWC [B,H+2pg,W+2pg,C]; thread = one float4 of a padded pixel (border pixels get zeros)
__global__ __launch_bounds__(256) void gen_pad_nhwc_kernel(int C, int H, int W, int pg, const float* __restrict__ src,
                                                           float* __restrict__ dst) {
    const int b = blockIdx.y, c4n = C / 4, Hp = H + 2 * pg, Wp = W + 2 * pg;
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)Hp * Wp * c4n) return;
    const int pix = (int)(t / c4n), c4 = (int)(t - (size_t)pix * c4n);
    const int y = pix / Wp - pg, x = pix % Wp - pg;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (y >= 0 && y < H && x >= 0 && x < W) {
        const float* s = src + (((size_t)b * C + 4 * c4) * H + y) * W + x;
        const size_t hw = (size_t)H * W;
        v = make_float4(s[0], s[hw], s[2 * hw], s[3 * hw]);
    }
    reinterpret_cast<float4*>(dst + ((size_t)b * Hp * Wp + pix) * C)[c4] = v;
}

// weight [O, Cw, k, k] (conv weights; Linear weights over Unfold's (c,kh,kw) patch order, dagl.py:196-203) -> rows
// dst[o * ld + off + (kh*k + kw) * Cw + c]: the element order of unfold_patches_kernel
__global__ void gen_weight_rows_kernel(int O, int Cw, int k, const float* __restrict__ src, float* __restrict__ dst, long long ld,
                                       long long off) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t per = (size_t)Cw * k * k;
    if (t >= (size_t)O * per) return;
    const int o = (int)(t / per); const int e = (int)(t - (size_t)o * per);
    const int tap = e / Cw, c = e - tap * Cw;
    dst[(size_t)o * ld + off + e] = src[((size_t)o * Cw + c) * (k * k) + tap];
}

// y [B*H*W, 2c] (g | theta) -> the two zero-bordered NHWC maps (their borders were zeroed by a memset)
__global__ __launch_bounds__(256) void gen_split_maps_kernel(int C, int H, int W, int pg, const float* __restrict__ y,
                                                             float* __restrict__ b1p, float* __restrict__ b2p) {
    const int b = blockIdx.y, c4n = C / 4, Wp = W + 2 * pg, Hp = H + 2 * pg;
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)H * W * 2 * c4n) return;
    const int pix = (int)(t / (2 * c4n)), e = (int)(t - (size_t)pix * (2 * c4n));
    const int yy = pix / W, xx = pix - yy * W;
    const float4 v = reinterpret_cast<const float4*>(y + ((size_t)b * H * W + pix) * (2 * C))[e];
    float* m = e < c4n ? b1p : b2p;
    reinterpret_cast<float4*>(m + (((size_t)b * Hp + yy + pg) * Wp + xx + pg) * C)[e < c4n ? e : e - c4n] = v;
}

__device__ __forceinline__ double gen_block_sum(double v, double* sh) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}
__device__ __forceinline__ float gen_block_max(float v, float* sh) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
}

// MODE 0 adaptive (dagl.py:256-261), 1 the k best (0/1 mask), 2 both.  (T, jt, tie): wide_radix_select's k-th sort key, the last
// key index taken at that key when the k-th place is tied (ties go to the lower index)
template <int MODE>
__device__ __forceinline__ float gen_logit(float s, int j, float mtq, float bsq, float scale, unsigned T, int jt, bool& pass) {
    float m = 1.f;
    if (MODE != 1) { m = (s - mtq) + bsq; pass = m > 0.f; }          // expression order of dagl.py:256
    if (MODE != 0) {
        const unsigned key = wide_key(s, MODE == 2, mtq, bsq);
        pass = key != 0u && (key > T || (key == T && j <= jt));
    }
    return pass ? __fmul_rn(__fmul_rn(s, m), scale) : 0.f;            // (S m) softmax_scale, dagl.py:259-260
}

// one block per query row of the chunk: S row -> A row in place
// (abuf != sbuf: the differentiable path keeps S for its backward)
// LIST (fixed-k modes, k <= GEN_LIST_MAX): instead of the A row, the row's neighbours as a list of k (key, weight) pairs in key order,
// empty slots = -1 -- the weighted sum of dagl.py:263-264 is then a gather over k value rows (gather_rows_kernel), not an [Lc,N] x [N,P] product
constexpr int GEN_LIST_MAX = GEN_LIST_CAP_WORDS;
template <int MODE, bool LIST = false>
__global__ __launch_bounds__(256) void gen_row_softmax_kernel(int N, long long ldn, int L, int l0, int b, int k, float scale,
                                                              const float* sbuf, float* abuf, const float* __restrict__ thr,
                                                              const float* __restrict__ bias, int tstride, int32_t* __restrict__ deg,
                                                              int32_t* __restrict__ nb_idx = nullptr, float* __restrict__ nb_wgt = nullptr) {
    __shared__ double shd[4];
    __shared__ float shf[4];
    __shared__ WideSelShared shs;
    __shared__ int sh_cnt[4];
    __shared__ int sh_jt;
    const int lr = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const size_t ql = (size_t)b * L + l0 + lr;
    const float* row = sbuf + (size_t)lr * ldn;
    float* arow = abuf + (size_t)lr * ldn;
    float mtq = 0.f, bsq = 0.f;
    if (MODE != 1) {
        double sm = 0.0;
        for (int j = tid; j < N; j += 256) sm += (double)row[j];
        const float mean = (float)(gen_block_sum(sm, shd) / (double)N);           // yi.mean(dim=1), dagl.py:256
        mtq = __fmul_rn(mean, thr[ql * tstride]); bsq = bias[ql * tstride];
    }
    unsigned T = 0u; int jt = 0x7fffffff;
    if (MODE != 0) {
        unsigned need, bin_count;
        wide_radix_select(row, N, k, MODE == 2, mtq, bsq, shs, T, need, bin_count);
        if (need < bin_count) {                                    // block-uniform: a tie at the k-th place, the `need` lowest key indices win
            int run = 0;
            for (int j0 = 0; j0 < N; j0 += 256) {
                const int j = j0 + tid;
                const bool eq = j < N && wide_key(row[j], MODE == 2, mtq, bsq) == T;
                const unsigned long long eqb = __ballot(eq);
                if (lane == 0) sh_cnt[w] = __popcll(eqb);
                __syncthreads();
                int before = run;
                for (int u = 0; u < w; ++u) before += sh_cnt[u];
                const int rank = before + __popcll(eqb & ((1ull << lane) - 1ull));
                if (eq && rank == (int)need - 1) sh_jt = j;
                run += sh_cnt[0] + sh_cnt[1] + sh_cnt[2] + sh_cnt[3];
                __syncthreads();
                if (run >= (int)need) break;
            }
            jt = sh_jt;
        }
    }
    float mx = -__builtin_inff(); int cnt = 0;                     // the softmax runs over ALL keys: a masked key's logit is 0 (dagl.py:259: yi * mask)
    for (int j = tid; j < N; j += 256) {
        bool pass;
        const float l = gen_logit<MODE>(row[j], j, mtq, bsq, scale, T, jt, pass);
        cnt += pass ? 1 : 0;
        mx = fmaxf(mx, l);
    }
    const float M = gen_block_max(mx, shf);
    double z = 0.0;
    for (int j = tid; j < N; j += 256) {
        bool pass;
        const float l = gen_logit<MODE>(row[j], j, mtq, bsq, scale, T, jt, pass);
        z += (double)expf(l - M);
    }
    const float invz = (float)(1.0 / gen_block_sum(z, shd));
    if (LIST) {
        // ordered compaction of the (at most k) passing keys: positions by ballot + wave offsets, no atomics -- the list (and with it the
        // gather's summation order) is the same on every run
        int32_t* li = nb_idx + (size_t)lr * k;
        float* lw = nb_wgt + (size_t)lr * k;
        int run = 0;
        for (int j0 = 0; j0 < N; j0 += 256) {
            const int j = j0 + tid;
            bool pass = false; float l = 0.f;
            if (j < N) l = gen_logit<MODE>(row[j], j, mtq, bsq, scale, T, jt, pass);
            const unsigned long long pb = __ballot(pass);
            if (lane == 0) sh_cnt[w] = __popcll(pb);
            __syncthreads();
            int pos = run;
            for (int u = 0; u < w; ++u) pos += sh_cnt[u];
            pos += __popcll(pb & ((1ull << lane) - 1ull));
            if (pass && pos < k) { li[pos] = j; lw[pos] = expf(l - M) * invz; }
            run += sh_cnt[0] + sh_cnt[1] + sh_cnt[2] + sh_cnt[3];
            __syncthreads();
            if (run >= k) break;                                    // (block-uniform; at most k keys pass in the fixed-k modes)
        }
        for (int e = run + tid; e < k; e += 256) { li[e] = -1; lw[e] = 0.f; }
    } else {
        for (int j = tid; j < N; j += 256) {
            bool pass;
            const float l = gen_logit<MODE>(row[j], j, mtq, bsq, scale, T, jt, pass);
            arow[j] = pass ? expf(l - M) * invz : 0.f;             // softmax * mask_b, dagl.py:260-261
        }
        for (int j = N + tid; j < ldn; j += 256) arow[j] = 0.f;
    }
    const double Cn = gen_block_sum((double)cnt, shd);
    if (tid == 0 && deg != nullptr) deg[ql] = (int32_t)Cn;
}

// out[b,c,y,x] = sum over the fold's blocks covering (y,x) of agg[b, block, (kh,kw,c)] / max(count, 1 if 0)   (dagl.py:265-272)
__global__ __launch_bounds__(256) void gen_fold_normalize_kernel(int C, int H, int W, int ks, int s1, int pad, int fh, int fw,
                                                                 const float* __restrict__ agg, float* __restrict__ out) {
    const int b = blockIdx.y;
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)C * H * W) return;
    const int x = (int)(t % W); const int y = (int)((t / W) % H); const int c = (int)(t / ((size_t)W * H));
    const size_t P_ = (size_t)ks * ks * C;
    float acc = 0.f; int cnt = 0;
    for (int kh = 0; kh < ks; ++kh) {
        const int ty = y + pad - kh;
        if (ty < 0 || ty % s1 != 0 || ty / s1 >= fh) continue;
        for (int kw = 0; kw < ks; ++kw) {
            const int tx = x + pad - kw;
            if (tx < 0 || tx % s1 != 0 || tx / s1 >= fw) continue;
            acc += agg[((size_t)b * fh * fw + (size_t)(ty / s1) * fw + tx / s1) * P_ + (size_t)(kh * ks + kw) * C + c];
            ++cnt;
        }
    }
    out[(size_t)b * C * H * W + t] = acc / (float)(cnt == 0 ? 1 : cnt);             // out_mask += (out_mask == 0), dagl.py:271
}

Gemm32 gen_gemm(int M, int N, int K, const float* A, long long lda, const float* Bm, long long ldb, int b_kc, float* Cm, long long ldc,
                const float* bias, int relu) {
    Gemm32 g;
    g.M = M; g.N = N; g.K = K; g.batch = 1;
    g.A = A; g.lda = lda; g.sA = 0; g.a_kc = 1;
    g.B = Bm; g.ldb = ldb; g.sB = 0; g.b_kc = b_kc;
    g.C = Cm; g.ldc = ldc; g.sC = 0;
    g.alpha = 1.f; g.beta = 0.f; g.bias = bias; g.relu = relu;
    g.chunk_tiles = 7;                              // partial sums of 112 products added in fp32: shorter rounding chains
    return g;
}

inline dim3 gen_grid(size_t n, int B) { return dim3((unsigned)((n + 255) / 256), (unsigned)B); }

// ---- the differentiable form (autograd through dagl.py:250-272 for a module built with a non-default geometry) ------------------
// d agg[b, block (py,px), (kh,kw,c)] = d out[b, c, py*s1 - pad + kh, px*s1 - pad + kw] / count   (adjoint of gen_fold_normalize_kernel)
__global__ __launch_bounds__(256) void gen_unfold_out_kernel(int C, int H, int W, int ks, int s1, int pad, int fh, int fw,
                                                             const float* __restrict__ dout, float* __restrict__ dagg) {
    const int b = blockIdx.y;
    const size_t P_ = (size_t)ks * ks * C;
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)fh * fw * P_) return;
    const int blk = (int)(t / P_); const int e = (int)(t - (size_t)blk * P_);
    const int tap = e / C, c = e - tap * C, kh = tap / ks, kw = tap - kh * ks;
    const int py = blk / fw, px = blk - py * fw;
    const int y = py * s1 - pad + kh, x = px * s1 - pad + kw;
    float v = 0.f;
    if (y >= 0 && y < H && x >= 0 && x < W) {
        int cnt = 0;                                                            // how many blocks cover (y, x): as the forward counts them
        for (int a = 0; a < ks; ++a) {
            const int ty = y + pad - a;
            if (ty < 0 || ty % s1 != 0 || ty / s1 >= fh) continue;
            for (int q = 0; q < ks; ++q) {
                const int tx = x + pad - q;
                if (tx < 0 || tx % s1 != 0 || tx / s1 >= fw) continue;
                ++cnt;
            }
        }
        v = dout[(((size_t)b * C + c) * H + y) * W + x] / (float)(cnt == 0 ? 1 : cnt);
    }
    dagg[(size_t)b * fh * fw * P_ + t] = v;
}

// one block per query row of the chunk: sbuf = S row, abuf = A row, dabuf = d A row  ->  dabuf = d S row (without the row-mean term),
// d bias[q] = t = sum_j d l_j scale S_j,  d thr[q] = -t mean,  rcoef[q] = -t thr / N (the row-mean term: d S_j += rcoef for EVERY key j,
// applied as two rank-1 updates of d Wq / d X).  l_j = scale S_j m_j with m_j = S_j - mean thr + bias on the passing keys (a 0/1 mask
// in the fixed-k mode: MODE 1), y_j = mask_j e^{l_j} / Z  =>  d l_j = y_j (d y_j - sum_i y_i d y_i); a key passes iff its weight is
// not zero (a passing key whose weight underflowed has no gradient either way).
template <int MODE>
__global__ __launch_bounds__(256) void gen_row_backward_kernel(int N, long long ldn, int L, int l0, int b, float scale,
                                                               const float* __restrict__ sbuf, const float* __restrict__ abuf, float* __restrict__ dabuf,
                                                               const float* __restrict__ thr, const float* __restrict__ bias,
                                                               float* __restrict__ d_thr, float* __restrict__ d_bias, float* __restrict__ rcoef) {
    __shared__ double shd[4];
    const int lr = blockIdx.x, tid = threadIdx.x;
    const size_t ql = (size_t)b * L + l0 + lr;
    const float* srow = sbuf + (size_t)lr * ldn;
    const float* arow = abuf + (size_t)lr * ldn;
    float* drow = dabuf + (size_t)lr * ldn;
    float mean = 0.f, mtq = 0.f, bsq = 0.f;
    if (MODE != 1) {
        double sm = 0.0;
        for (int j = tid; j < N; j += 256) sm += (double)srow[j];
        mean = (float)(gen_block_sum(sm, shd) / (double)N);
        mtq = __fmul_rn(mean, thr[ql]); bsq = bias[ql];
    }
    double cs = 0.0;
    for (int j = tid; j < N; j += 256) cs += (double)arow[j] * (double)drow[j];
    const float c = (float)gen_block_sum(cs, shd);
    double ts = 0.0;
    for (int j = tid; j < N; j += 256) {
        const float a = arow[j], sj = srow[j];
        float ds = 0.f;
        if (a != 0.f) {
            const float dl = a * (drow[j] - c);
            if (MODE == 1) ds = dl * scale;
            else {
                const float m = (sj - mtq) + bsq;
                ds = dl * scale * (m + sj);
                ts += (double)(dl * scale) * (double)sj;
            }
        }
        drow[j] = ds;
    }
    for (int j = N + tid; j < ldn; j += 256) drow[j] = 0.f;
    if (MODE != 1) {
        const double t = gen_block_sum(ts, shd);
        if (tid == 0) {
            d_bias[ql] = (float)t;
            d_thr[ql] = (float)(-t * (double)mean);
            rcoef[ql] = (float)(-t * (double)thr[ql] / (double)N);
        }
    }
}

__global__ void gen_fill_kernel(size_t n, float v, float* __restrict__ dst) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) dst[t] = v;
}
// out[r, c] += (rowcoef ? rowcoef[r] : 1) * vec[c]
__global__ void gen_rank1_add_kernel(size_t rows, int cols, const float* __restrict__ rowcoef, const float* __restrict__ vec, float* __restrict__ out) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= rows * cols) return;
    const size_t r = t / cols; const int c = (int)(t - r * cols);
    out[t] += (rowcoef ? rowcoef[r] : 1.f) * vec[c];
}

struct GenCorePlan { size_t o_vrows, o_dvrows, o_s, o_a, o_da, o_agg, o_rc, o_ones, o_vec, o_end; };
inline GenCorePlan gen_core_plan(const GenGeom& g, bool backward) {
    GenCorePlan p{};
    size_t off = 0;
    auto carve = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
    const size_t B = g.B;
    p.o_vrows = carve(B * g.N * (size_t)g.P * 4);
    p.o_dvrows = backward ? carve(B * g.N * (size_t)g.P * 4) : 0;
    p.o_s = carve((size_t)g.Lc * (g.ldn + 2 * GEN_LIST_CAP_WORDS) * 4);
    p.o_a = backward ? carve((size_t)g.Lc * g.ldn * 4) : 0;
    p.o_da = backward ? carve((size_t)g.Lc * g.ldn * 4) : 0;
    p.o_agg = carve(B * g.L * (size_t)g.P * 4);
    p.o_rc = backward ? carve(B * g.L * 4) : 0;
    p.o_ones = backward ? carve((size_t)(g.N > g.L ? g.N : g.L) * 4) : 0;
    p.o_vec = backward ? carve((size_t)g.D * 4 * 2) : 0;
    p.o_end = off;
    return p;
}

// dagl.py:250-272 from the feature rows: a chunk of query rows at a time -- S chunk, row-wise mask / softmax, A chunk x value rows --,
// then fold + overlap count.  vrows [B,N,P] = the value patches (unfold of the zero-bordered value map).
int gen_core_forward(hipStream_t s, const GenGeom& g, float scale, int mode, int k, const float* wq, const float* x, const float* vrows,
                     const float* thr, const float* bias, int tstride, float* sbuf, float* agg, float* out, int32_t* degree) {
    int rc;
    const int kk = k < g.N ? k : g.N;                                      // top_k = min(num_edge, N), GReccR2b_3mh_1-checkpoint.py:243
    // the lists live in the tail of the score chunk's buffer (gen_geom sizes a chunk so that Lc rows of scores AND of lists fit)
    const bool lists = mode != DAGL_MODE_ADAPTIVE && kk <= GEN_LIST_MAX;
    int32_t* nb_idx = reinterpret_cast<int32_t*>(sbuf + (size_t)g.Lc * g.ldn);
    float* nb_wgt = reinterpret_cast<float*>(nb_idx + (size_t)g.Lc * GEN_LIST_MAX);
    for (int b = 0; b < g.B; ++b) {
        const float* Xb = x + (size_t)b * g.N * g.D;
        const float* Vb = vrows + (size_t)b * g.N * g.P;
        for (int l0 = 0; l0 < g.L; l0 += g.Lc) {
            const int lc = (g.L - l0 < g.Lc) ? g.L - l0 : g.Lc;
            const float* Wqc = wq + ((size_t)b * g.L + l0) * g.D;
            if ((rc = launch_gemm32(s, gen_gemm(lc, g.N, g.D, Wqc, g.D, Xb, g.D, 1, sbuf, g.ldn, nullptr, 0)))) return rc;
            float* aggc = agg + ((size_t)b * g.L + l0) * g.P;
            if (lists) {
                // fixed-k modes with short lists: (key, weight) lists out of the row kernel, then a gather over k value rows
                if (mode == DAGL_MODE_TOPK)
                    hipLaunchKernelGGL((gen_row_softmax_kernel<1, true>), dim3(lc), dim3(256), 0, s, g.N, g.ldn, g.L, l0, b, kk, scale, sbuf, sbuf, thr, bias,
                                       tstride, degree, nb_idx, nb_wgt);
                else
                    hipLaunchKernelGGL((gen_row_softmax_kernel<2, true>), dim3(lc), dim3(256), 0, s, g.N, g.ldn, g.L, l0, b, kk, scale, sbuf, sbuf, thr, bias,
                                       tstride, degree, nb_idx, nb_wgt);
                DAGL_LAUNCH_CHECK("gen_row_softmax_kernel");
                if ((rc = launch_gather_fixed(s, lc, kk, g.P, nb_idx, nb_wgt, Vb, aggc))) return rc;
                continue;
            }
#define GEN_ROWS(M_) hipLaunchKernelGGL((gen_row_softmax_kernel<M_>), dim3(lc), dim3(256), 0, s, g.N, g.ldn, g.L, l0, b, kk, scale, sbuf, sbuf, thr, bias, tstride, degree)
            if (mode == DAGL_MODE_ADAPTIVE) GEN_ROWS(0); else if (mode == DAGL_MODE_TOPK) GEN_ROWS(1); else GEN_ROWS(2);
#undef GEN_ROWS
            DAGL_LAUNCH_CHECK("gen_row_softmax_kernel");
            if ((rc = launch_gemm32(s, gen_gemm(lc, g.P, g.N, sbuf, g.ldn, Vb, g.P, 0, aggc, g.P, nullptr, 0)))) return rc;
        }
    }
    // fold + overlap count, dagl.py:265-272
    hipLaunchKernelGGL(gen_fold_normalize_kernel, gen_grid((size_t)g.C * g.H * g.W, g.B), dim3(256), 0, s, g.C, g.H, g.W, g.ks, g.s1, g.fold_pad,
                       g.fold_h, g.fold_w, agg, out);
    DAGL_LAUNCH_CHECK("gen_fold_normalize_kernel");
    return DAGL_OK;
}

}  // namespace

size_t ce_generic_workspace_bytes(int B, int Cin, int H, int W, int ks, int s1, int s2, int C) {
    return gen_plan(gen_geom(B, Cin, H, W, ks, s1, s2, C)).o_end + 256;
}

int ce_generic_check(int B, int Cin, int H, int W, int ks, int s1, int s2, int C, int mode, int k) {
    DAGL_REQUIRE(B >= 1 && H >= 1 && W >= 1, "dagl_ce_generic_forward: bad shape");
    DAGL_REQUIRE(ks >= 1 && ks <= 31 && s1 >= 1 && s2 >= 1, "dagl_ce_generic_forward: ksize in 1..31, strides >= 1");
    DAGL_REQUIRE(Cin >= 4 && Cin % 4 == 0 && C >= 4 && C % 4 == 0,
                 "dagl_ce_generic_forward: in_channels and inter_channels must be multiples of 4 (16-byte pixels of the NHWC maps)");
    DAGL_REQUIRE(mode == DAGL_MODE_ADAPTIVE || mode == DAGL_MODE_TOPK || mode == DAGL_MODE_ADAPTIVE_TOPK, "dagl_ce_generic_forward: bad mode");
    DAGL_REQUIRE(mode == DAGL_MODE_ADAPTIVE || k >= 1, "dagl_ce_generic_forward: k >= 1 in the top-k modes");
    const GenGeom g = gen_geom(B, Cin, H, W, ks, s1, s2, C);
    // F.fold (dagl.py:267) raises when its block grid does not hold exactly the L query patches
    DAGL_REQUIRE(g.fold_h >= 1 && g.fold_w >= 1 && (long long)g.fold_h * g.fold_w == (long long)g.L,
                 "dagl_ce_generic_forward: fold(kernel %d, padding %d, stride %d) of a %dx%d map has %dx%d blocks, the query grid %dx%d "
                 "(the reference's F.fold raises on this geometry too)", ks, g.fold_pad, s1, H, W, g.fold_h, g.fold_w, g.Lh, g.Lw);
    return DAGL_OK;
}

int launch_ce_generic(hipStream_t s, int B, int Cin, int H, int W, int ks, int s1, int s2, int C, float scale, int mode, int k,
                      const float* x, const float* g_w, const float* g_b, const float* th_w, const float* th_b, const float* thr_w,
                      const float* thr_b, const float* bias_w, const float* bias_b, const float* fc1_w, const float* fc1_b,
                      const float* fc2_w, const float* fc2_b, float* out, int32_t* degree, void* workspace) {
    const GenGeom g = gen_geom(B, Cin, H, W, ks, s1, s2, C);
    const GenPlan p = gen_plan(g);
    char* ws = reinterpret_cast<char*>(((uintptr_t)workspace + 255) / 256 * 256);
    auto F = [&](size_t off) { return reinterpret_cast<float*>(ws + off); };
    int rc;
    const size_t HW = (size_t)H * W;
    const bool heads = mode != DAGL_MODE_TOPK;

    // ---- weights in the unfold's element order ------------------------------------------------------------------------------
    const long long kg = 9ll * Cin;
    DAGL_HIP_TRY(hipMemsetAsync(F(p.o_wgt), 0, (size_t)2 * C * kg * 4, s));
    hipLaunchKernelGGL(gen_weight_rows_kernel, dim3((unsigned)(((size_t)C * kg + 255) / 256)), dim3(256), 0, s, C, Cin, 3, g_w, F(p.o_wgt), kg, 0ll);
    hipLaunchKernelGGL(gen_weight_rows_kernel, dim3((unsigned)(((size_t)C * Cin + 255) / 256)), dim3(256), 0, s, C, Cin, 1, th_w,
                       F(p.o_wgt) + (size_t)C * kg, kg, 4ll * Cin);                       // theta = the centre tap
    DAGL_HIP_TRY(hipMemcpyAsync(F(p.o_bgt), g_b, (size_t)C * 4, hipMemcpyDeviceToDevice, s));
    DAGL_HIP_TRY(hipMemcpyAsync(F(p.o_bgt) + C, th_b, (size_t)C * 4, hipMemcpyDeviceToDevice, s));
    hipLaunchKernelGGL(gen_weight_rows_kernel, dim3((unsigned)(((size_t)g.D * g.P + 255) / 256)), dim3(256), 0, s, g.D, C, ks, fc1_w, F(p.o_fc1), (long long)g.P, 0ll);
    hipLaunchKernelGGL(gen_weight_rows_kernel, dim3((unsigned)(((size_t)g.D * g.P + 255) / 256)), dim3(256), 0, s, g.D, C, ks, fc2_w, F(p.o_fc2), (long long)g.P, 0ll);
    const long long ktb = (long long)ks * ks * Cin;
    if (heads) {
        hipLaunchKernelGGL(gen_weight_rows_kernel, dim3((unsigned)((ktb + 255) / 256)), dim3(256), 0, s, 1, Cin, ks, thr_w, F(p.o_wtb), ktb, 0ll);
        hipLaunchKernelGGL(gen_weight_rows_kernel, dim3((unsigned)((ktb + 255) / 256)), dim3(256), 0, s, 1, Cin, ks, bias_w, F(p.o_wtb) + ktb, ktb, 0ll);
        DAGL_HIP_TRY(hipMemcpyAsync(F(p.o_btb), thr_b, 4, hipMemcpyDeviceToDevice, s));
        DAGL_HIP_TRY(hipMemcpyAsync(F(p.o_btb) + 1, bias_b, 4, hipMemcpyDeviceToDevice, s));
    }
    DAGL_LAUNCH_CHECK("gen_weight_rows_kernel");

    // ---- prologue convolutions, dagl.py:208-215 -----------------------------------------------------------------------------
    hipLaunchKernelGGL(gen_pad_nhwc_kernel, gen_grid((size_t)g.Hp * g.Wp * (Cin / 4), B), dim3(256), 0, s, Cin, H, W, g.PG, x, F(p.o_xp));
    DAGL_LAUNCH_CHECK("gen_pad_nhwc_kernel");
    if ((rc = launch_unfold_patches(s, B, g.Hp, g.Wp, Cin, 3, 1, g.PG - 1, g.PG - 1, H, W, F(p.o_xp), F(p.o_rows)))) return rc;
    if ((rc = launch_gemm32(s, gen_gemm((int)(B * HW), 2 * C, (int)kg, F(p.o_rows), kg, F(p.o_wgt), kg, 1, F(p.o_y), 2 * C, F(p.o_bgt), 0)))) return rc;
    DAGL_HIP_TRY(hipMemsetAsync(F(p.o_b1p), 0, (size_t)B * g.Hp * g.Wp * C * 4, s));
    DAGL_HIP_TRY(hipMemsetAsync(F(p.o_b2p), 0, (size_t)B * g.Hp * g.Wp * C * 4, s));
    hipLaunchKernelGGL(gen_split_maps_kernel, gen_grid(HW * 2 * (C / 4), B), dim3(256), 0, s, C, H, W, g.PG, F(p.o_y), F(p.o_b1p), F(p.o_b2p));
    DAGL_LAUNCH_CHECK("gen_split_maps_kernel");
    if (heads) {
        if ((rc = launch_unfold_patches(s, B, g.Hp, g.Wp, Cin, ks, s1, g.PG - g.t1, g.PG - g.l1, g.Lh, g.Lw, F(p.o_xp), F(p.o_rows)))) return rc;
        if ((rc = launch_gemm32(s, gen_gemm(B * g.L, 2, (int)ktb, F(p.o_rows), ktb, F(p.o_wtb), ktb, 1, F(p.o_tb), 2, F(p.o_btb), 0)))) return rc;
    }

    // ---- patch features, dagl.py:216-249 ------------------------------------------------------------------------------------
    if ((rc = launch_unfold_patches(s, B, g.Hp, g.Wp, C, ks, s1, g.PG - g.t1, g.PG - g.l1, g.Lh, g.Lw, F(p.o_b1p), F(p.o_rows)))) return rc;
    if ((rc = launch_gemm32(s, gen_gemm(B * g.L, g.D, g.P, F(p.o_rows), g.P, F(p.o_fc1), g.P, 1, F(p.o_wq), g.D, fc1_b, 1)))) return rc;
    if ((rc = launch_unfold_patches(s, B, g.Hp, g.Wp, C, ks, s2, g.PG - g.t2, g.PG - g.l2, g.Nh, g.Nw, F(p.o_b1p), F(p.o_rows)))) return rc;
    if ((rc = launch_gemm32(s, gen_gemm(B * g.N, g.D, g.P, F(p.o_rows), g.P, F(p.o_fc2), g.P, 1, F(p.o_x), g.D, fc2_b, 1)))) return rc;
    if ((rc = launch_unfold_patches(s, B, g.Hp, g.Wp, C, ks, s2, g.PG - g.t2, g.PG - g.l2, g.Nh, g.Nw, F(p.o_b2p), F(p.o_rows)))) return rc;   // value rows

    return gen_core_forward(s, g, scale, mode, k, F(p.o_wq), F(p.o_x), F(p.o_rows), F(p.o_tb), F(p.o_tb) + 1, 2, F(p.o_s), F(p.o_agg), out, degree);
}

size_t ce_generic_core_workspace_bytes(int B, int H, int W, int ks, int s1, int s2, int C, int backward) {
    return gen_core_plan(gen_geom(B, 4, H, W, ks, s1, s2, C), backward != 0).o_end + 256;
}
int ce_generic_border(int ks) { return ks > 2 ? ks - 1 : 1; }

// forward of the differentiable core: the feature rows and the zero-bordered NHWC value map (border = ce_generic_border(ks)) given
int launch_ce_generic_core_forward(hipStream_t s, int B, int H, int W, int ks, int s1, int s2, int C, float scale, int mode, int k,
                                   const float* wq, const float* x, const float* b2p, const float* thr, const float* bias, float* out,
                                   int32_t* degree, void* workspace) {
    const GenGeom g = gen_geom(B, 4, H, W, ks, s1, s2, C);
    const GenCorePlan p = gen_core_plan(g, false);
    char* ws = reinterpret_cast<char*>(((uintptr_t)workspace + 255) / 256 * 256);
    auto F = [&](size_t off) { return reinterpret_cast<float*>(ws + off); };
    int rc;
    if ((rc = launch_unfold_patches(s, B, g.Hp, g.Wp, C, ks, s2, g.PG - g.t2, g.PG - g.l2, g.Nh, g.Nw, b2p, F(p.o_vrows)))) return rc;
    return gen_core_forward(s, g, scale, mode, k, wq, x, F(p.o_vrows), thr, bias, 1, F(p.o_s), F(p.o_agg), out, degree);
}

int launch_fold_patches(hipStream_t s, int B, int Hp, int Wp, int C, int k, int stride, int oy, int ox, int oh, int ow, const float* drows,
                        float* dmap);

// backward: S and A are recomputed chunk by chunk (nothing O(L N) is kept between the passes)
int launch_ce_generic_core_backward(hipStream_t s, int B, int H, int W, int ks, int s1, int s2, int C, float scale, int mode, int k,
                                    const float* wq, const float* x, const float* b2p, const float* thr, const float* bias, const float* d_out,
                                    float* d_wq, float* d_x, float* d_b2p, float* d_thr, float* d_bias, void* workspace) {
    const GenGeom g = gen_geom(B, 4, H, W, ks, s1, s2, C);
    const GenCorePlan p = gen_core_plan(g, true);
    char* ws = reinterpret_cast<char*>(((uintptr_t)workspace + 255) / 256 * 256);
    auto F = [&](size_t off) { return reinterpret_cast<float*>(ws + off); };
    int rc;
    const bool heads = mode != DAGL_MODE_TOPK;
    const int kk = k < g.N ? k : g.N;
    if ((rc = launch_unfold_patches(s, B, g.Hp, g.Wp, C, ks, s2, g.PG - g.t2, g.PG - g.l2, g.Nh, g.Nw, b2p, F(p.o_vrows)))) return rc;
    hipLaunchKernelGGL(gen_unfold_out_kernel, gen_grid((size_t)g.L * g.P, B), dim3(256), 0, s, C, H, W, ks, s1, g.fold_pad, g.fold_h, g.fold_w,
                       d_out, F(p.o_agg));
    DAGL_LAUNCH_CHECK("gen_unfold_out_kernel");
    const size_t n_ones = (size_t)(g.N > g.L ? g.N : g.L);
    hipLaunchKernelGGL(gen_fill_kernel, dim3((unsigned)((n_ones + 255) / 256)), dim3(256), 0, s, n_ones, 1.0f, F(p.o_ones));
    DAGL_LAUNCH_CHECK("gen_fill_kernel");
    for (int b = 0; b < B; ++b) {
        const float* Xb = x + (size_t)b * g.N * g.D;
        const float* Vb = F(p.o_vrows) + (size_t)b * g.N * g.P;
        float* dVb = F(p.o_dvrows) + (size_t)b * g.N * g.P;
        float* dXb = d_x + (size_t)b * g.N * g.D;
        for (int l0 = 0; l0 < g.L; l0 += g.Lc) {
            const int lc = (g.L - l0 < g.Lc) ? g.L - l0 : g.Lc;
            const float* Wqc = wq + ((size_t)b * g.L + l0) * g.D;
            const float* dAggc = F(p.o_agg) + ((size_t)b * g.L + l0) * g.P;
            const float beta = l0 == 0 ? 0.f : 1.f;
            if ((rc = launch_gemm32(s, gen_gemm(lc, g.N, g.D, Wqc, g.D, Xb, g.D, 1, F(p.o_s), g.ldn, nullptr, 0)))) return rc;          // S
#define GEN_ROWS(M_) hipLaunchKernelGGL((gen_row_softmax_kernel<M_>), dim3(lc), dim3(256), 0, s, g.N, g.ldn, g.L, l0, b, kk, scale, F(p.o_s), F(p.o_a), thr, bias, 1, (int32_t*)nullptr)
            if (mode == DAGL_MODE_ADAPTIVE) GEN_ROWS(0); else if (mode == DAGL_MODE_TOPK) GEN_ROWS(1); else GEN_ROWS(2);                 // A
#undef GEN_ROWS
            DAGL_LAUNCH_CHECK("gen_row_softmax_kernel");
            if ((rc = launch_gemm32(s, gen_gemm(lc, g.N, g.P, dAggc, g.P, Vb, g.P, 1, F(p.o_da), g.ldn, nullptr, 0)))) return rc;       // d A = d agg V^T
            {                                                                                                                            // d V (+)= A^T d agg
                Gemm32 q = gen_gemm(g.N, g.P, lc, F(p.o_a), g.ldn, dAggc, g.P, 0, dVb, g.P, nullptr, 0);
                q.a_kc = 0; q.beta = beta;
                if ((rc = launch_gemm32(s, q))) return rc;
            }
            float* rc_q = F(p.o_rc) + (size_t)b * g.L;
#define GEN_BWD(M_) hipLaunchKernelGGL((gen_row_backward_kernel<M_>), dim3(lc), dim3(256), 0, s, g.N, g.ldn, g.L, l0, b, scale, F(p.o_s), F(p.o_a), F(p.o_da), \
                                       thr, bias, d_thr, d_bias, F(p.o_rc))
            if (heads) GEN_BWD(0); else GEN_BWD(1);                                                                                     // d S
#undef GEN_BWD
            DAGL_LAUNCH_CHECK("gen_row_backward_kernel");
            (void)rc_q;
            if ((rc = launch_gemm32(s, gen_gemm(lc, g.D, g.N, F(p.o_da), g.ldn, Xb, g.D, 0, d_wq + ((size_t)b * g.L + l0) * g.D, g.D, nullptr, 0)))) return rc;   // d Wq = d S X
            {                                                                                                                            // d X (+)= d S^T Wq
                Gemm32 q = gen_gemm(g.N, g.D, lc, F(p.o_da), g.ldn, Wqc, g.D, 0, dXb, g.D, nullptr, 0);
                q.a_kc = 0; q.beta = beta;
                if ((rc = launch_gemm32(s, q))) return rc;
            }
        }
        if (heads) {
            // the row-mean term of dagl.py:256: d S_lj += rcoef_l for every key j  =>  d Wq_l += rcoef_l sum_j X_j,  d X_j += sum_l rcoef_l Wq_l
            float* colsum = F(p.o_vec); float* u = F(p.o_vec) + g.D;
            if ((rc = launch_gemm32(s, gen_gemm(1, g.D, g.N, F(p.o_ones), g.N, Xb, g.D, 0, colsum, g.D, nullptr, 0)))) return rc;
            if ((rc = launch_gemm32(s, gen_gemm(1, g.D, g.L, F(p.o_rc) + (size_t)b * g.L, g.L, wq + (size_t)b * g.L * g.D, g.D, 0, u, g.D, nullptr, 0)))) return rc;
            hipLaunchKernelGGL(gen_rank1_add_kernel, dim3((unsigned)(((size_t)g.L * g.D + 255) / 256)), dim3(256), 0, s, (size_t)g.L, g.D,
                               F(p.o_rc) + (size_t)b * g.L, colsum, d_wq + (size_t)b * g.L * g.D);
            hipLaunchKernelGGL(gen_rank1_add_kernel, dim3((unsigned)(((size_t)g.N * g.D + 255) / 256)), dim3(256), 0, s, (size_t)g.N, g.D,
                               (const float*)nullptr, u, dXb);
            DAGL_LAUNCH_CHECK("gen_rank1_add_kernel");
        }
    }
    // d value map = fold(d value rows)
    return launch_fold_patches(s, B, g.Hp, g.Wp, C, ks, s2, g.PG - g.t2, g.PG - g.l2, g.Nh, g.Nw, F(p.o_dvrows), d_b2p);
}

}  // namespace dagl
