// Dense neighbourhoods under autograd: forward + backward of the graph core (DN_Gray/model/dagl.py:250-272) when a
// query keeps more keys than a fixed-width list holds (default-initialised thr/bias heads keep ~95 % of the keys: the
// regime a training run starts in).  This is the reference's dense formulation -- S = Wq X^T, mask, softmax over ALL
// keys, A V -- and exactly what autograd derives from it, chunked over the queries so that the [L,N] matrices exist only
// one chunk at a time:
//
//   forward, per chunk of queries          S = Wq X^T                               gemm32  (dagl.py:250)
//                                          A = softmax(10 S m) mask_b, m = relu(..)  dense_softmax_fwd_kernel (:256-261)
//                                          agg = A V                                gemm32  (:263-264),  then fold (:265-272)
//   backward, per chunk                    d A = d agg V^T                          gemm32
//                                          S recomputed; c = sum_j A d A; d l = A (d A - c);
//                                          d S = 10 (m + S) d l, d m = 10 S d l       dense_softmax_bwd_kernel
//                                          d Wq = d S X,  d X += d S^T Wq,  d V += A^T d agg        gemm32 x 3
//   plus the dense mean term of dagl.py:256: mu_l = Wq_l . Xbar  ->  d Wq_l += d mu_l Xbar, d X_j += (sum_l d mu_l Wq_l)/N.
// V = the unfolded value patches [N,784] of one image (materialised here, as the reference does at :224-231; the
// inference paths never do); d V is folded back onto the 16-channel map by a gather (49 taps per pixel, no atomics).
// All sums run in a fixed order: bit-reproducible gradients.
#include "dagl_common.h"
#include "wide_select.h"

namespace dagl {

constexpr size_t DT_CHUNK_FLOATS = (size_t)128 << 20;      // floats per [chunk, N] matrix (512 MiB)

struct DtPlan {
    int Lc, n_chunks, Bc;            // queries per chunk, chunks per image, images per group
    long long ldn;                   // leading dimension of the [L,N] chunk matrices (N rounded up to 32)
    size_t o_sbuf, o_abuf, o_vrows, o_dvrows, o_dagg, o_agg, o_b2p, o_colsum, o_mt, o_dmu, o_dxbar, o_deg, o_rowsum, o_sel, o_lse, o_mu, o_end;
    // split-fp16 backward (one chunk per image group only): hi / lo operand copies, each `..._h` halfs long (lo follows hi)
    bool h16;
    int Lp, kslices;                 // L rounded up to 32; split-K of d Wq
    int nk;                          // N rounded up to 32 kslices: the K extent of the copies whose rows run over the keys
    long long ldl, ldk;              // leading dimensions (halfs) of the copies whose rows run over the queries / the keys: extent + 64 -- a
                                     // power-of-two row stride (L = 1024: 2 KiB, N = 16384: 32 KiB) sends the 16 rows of every LDS-DMA piece
                                     // to the same memory channel (the products ran at a third of the fp32 ones' rate)
    size_t o_words, o_dgk, o_dgt, o_vk, o_xk, o_xt, o_wqk, o_wqt, o_dsk, o_dst, o_at, o_part;
    size_t dgk_h, dgt_h, vk_h, xk_h, xt_h, wqk_h, wqt_h, dsk_h, dst_h, at_h;
};

constexpr int DT_PK = 800;           // value-patch length 784 rounded up to the K step (50 taps of 16)
constexpr int DT_DK = 224;           // feature length 196 rounded up to the K step

static DtPlan dt_plan(int B, const Grid& g, bool backward) {
    DtPlan p;
    p.ldn = (g.N + 31) / 32 * 32;
    // (measured: skewing rows that are a multiple of 2 KiB long by 128 bytes does nothing for the row-per-block softmax kernels)
    long long lc = (long long)(DT_CHUNK_FLOATS / (size_t)p.ldn) / 128 * 128;
    if (lc < 128) lc = 128;
    p.Lc = (int)(lc < g.L ? lc : g.L);
    p.n_chunks = (g.L + p.Lc - 1) / p.Lc;
    p.Bc = 1;
    if (p.n_chunks == 1) {
        long long bc = (long long)(DT_CHUNK_FLOATS / ((size_t)p.Lc * p.ldn));
        p.Bc = (int)(bc < 1 ? 1 : (bc > B ? B : bc));
    }
    size_t off = 0;
    auto carve = [&](size_t bytes) { const size_t o = off; off = align_up(off + bytes, 256); return o; };
    const size_t chunk = (size_t)p.Bc * p.Lc * p.ldn * sizeof(float);
    p.o_sbuf = carve(chunk);
    p.o_abuf = backward ? carve(chunk) : 0;
    p.o_vrows = carve((size_t)p.Bc * g.N * P * sizeof(float));
    p.o_dvrows = backward ? carve((size_t)p.Bc * g.N * P * sizeof(float)) : 0;
    p.o_dagg = backward ? carve((size_t)B * g.L * P * sizeof(float)) : 0;
    p.o_agg = backward ? 0 : carve((size_t)B * g.L * P * sizeof(float));
    p.o_b2p = carve((size_t)B * g.Hp * g.Wp * CH * sizeof(float));
    p.o_colsum = carve((size_t)B * DS * sizeof(double));
    p.o_mt = carve((size_t)B * g.L * sizeof(float));
    p.o_dmu = carve((size_t)B * g.L * sizeof(float));
    p.o_dxbar = carve((size_t)B * D * sizeof(float));
    p.o_deg = carve((size_t)B * g.L * sizeof(int32_t));
    p.o_rowsum = carve((size_t)B * g.L * sizeof(float));
    p.o_sel = carve((size_t)B * g.L * 2 * sizeof(int32_t));      // wide top-k modes: (sort key of the k-th best score, last key index taken at it)
    p.o_lse = carve((size_t)B * g.L * 2 * sizeof(float));        // (their entry points keep the softmax statistics and the row means here)
    p.o_mu = carve((size_t)B * g.L * sizeof(float));
    p.h16 = backward && p.n_chunks == 1;
    p.Lp = (g.L + 31) / 32 * 32;
    p.kslices = 1;
    {
        // d Wq = d S X: (L / 128) x 2 output tiles per image -- split K (the keys) until the launch fills the chip
        const long long tiles = (long long)((g.L + 127) / 128) * 2 * p.Bc;
        while (p.kslices < 8 && tiles * p.kslices < 512 && g.N >= 512 * p.kslices) p.kslices *= 2;
    }
    p.nk = (g.N + 32 * p.kslices - 1) / (32 * p.kslices) * (32 * p.kslices);
    p.ldl = p.Lp + 64; p.ldk = p.nk + 64;
    if (p.h16) {
        const size_t bc = (size_t)p.Bc;
        p.dgk_h = bc * g.L * DT_PK; p.dgt_h = bc * P * p.ldl; p.vk_h = bc * g.N * DT_PK;
        p.xk_h = bc * g.N * DT_DK; p.xt_h = bc * D * p.ldk; p.wqk_h = bc * g.L * DT_DK; p.wqt_h = bc * D * p.ldl;
        p.dsk_h = bc * p.Lc * p.ldk; p.dst_h = bc * g.N * p.ldl; p.at_h = bc * g.N * p.ldl;
        p.o_words = carve(256);
        p.o_dgk = carve(2 * p.dgk_h * 2); p.o_dgt = carve(2 * p.dgt_h * 2); p.o_vk = carve(2 * p.vk_h * 2);
        p.o_xk = carve(2 * p.xk_h * 2); p.o_xt = carve(2 * p.xt_h * 2); p.o_wqk = carve(2 * p.wqk_h * 2); p.o_wqt = carve(2 * p.wqt_h * 2);
        p.o_dsk = carve(2 * p.dsk_h * 2); p.o_dst = carve(2 * p.dst_h * 2); p.o_at = carve(2 * p.at_h * 2);
        p.o_part = carve((size_t)p.kslices * bc * g.L * D * sizeof(float));
    }
    p.o_end = off;
    return p;
}

size_t dense_train_workspace_bytes(int B, const Grid& g, bool backward) { return dt_plan(B, g, backward).o_end; }

template <class T>
static T* dt_at(void* ws, size_t off) { return reinterpret_cast<T*>(static_cast<char*>(ws) + off); }

// mu[b,l] = Wq_l . colsum / N (fp64 dot, as query_thresholds_kernel), mt = mu * thr: one wave per query, dense rows
__global__ __launch_bounds__(256) void dt_thresholds_kernel(int L, int N, const float* __restrict__ wq_rows,
                                                            const double* __restrict__ colsum, const float* __restrict__ thr,
                                                            float* __restrict__ mt, float* __restrict__ mu) {
    const int b = blockIdx.y, lane = threadIdx.x & 63;
    const int l = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (l >= L) return;
    const float* q = wq_rows + ((size_t)b * L + l) * D;
    const double* cs = colsum + (size_t)b * DS;
    double acc = 0.0;
    for (int d = lane; d < D; d += 64) acc += (double)q[d] * cs[d];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) {
        const float mean = (float)(acc / (double)N);
        mu[(size_t)b * L + l] = mean;
        mt[(size_t)b * L + l] = mean * thr[(size_t)b * L + l];
    }
}

__device__ __forceinline__ double dt_block_sum(double v, double* sh) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}
__device__ __forceinline__ float dt_block_max(float v, float* sh) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
}

// MODE 0: the adaptive mask (dagl.py:256-261).  MODE 1: the k best scores of the row, a 0/1 mask: logits 10 S on them
// (GReccR2b_3mh_1-checkpoint.py:242-250).  MODE 2: both tests, m = relu(S - mean thr + bias) on the k best of the keys that pass.
// (T, jt) = sort key of the k-th best score and the last key index taken AT that key (ties go to the lower index): dt_select_kernel.
template <int MODE>
__device__ __forceinline__ float dt_logit(float s, int j, float mtq, float bsq, unsigned T, int jt, bool& pass, float& m) {
    if (MODE == 1) {
        const unsigned key = wide_key(s, false, 0.f, 0.f);
        pass = key > T || (key == T && j <= jt);
        m = 1.f;
        return pass ? __fmul_rn(s, SOFTMAX_SCALE) : 0.f;
    }
    m = (s - mtq) + bsq;                               // expression order of dagl.py:256
    pass = m > 0.f;
    if (MODE == 2) {
        const unsigned key = pass ? __float_as_uint(fmaxf(s, 0.f)) + 1u : 0u;
        pass = key != 0u && (key > T || (key == T && j <= jt));
    }
    return pass ? __fmul_rn(__fmul_rn(s, m), SOFTMAX_SCALE) : 0.f;       // (S m) 10, dagl.py:259-260
}

// wide top-k modes: one block per query row of the chunk -> sel[ql] = (T, jt)
template <bool ADAPTIVE>
__global__ __launch_bounds__(256) void dt_select_kernel(int N, long long ldn, int L, int l0, int Lc, int k, const float* __restrict__ sbuf,
                                                        const float* __restrict__ mt, const float* __restrict__ bs,
                                                        int32_t* __restrict__ sel, int b0) {
    __shared__ WideSelShared shs;
    __shared__ int sh_cnt[4];
    __shared__ int sh_jt;
    const int lr = blockIdx.x, bi = blockIdx.y, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const size_t ql = (size_t)(b0 + bi) * L + l0 + lr;
    const float* row = sbuf + ((size_t)bi * Lc + lr) * ldn;
    const float mtq = ADAPTIVE ? mt[ql] : 0.f, bsq = ADAPTIVE ? bs[ql] : 0.f;
    unsigned T, need, bin_count;
    wide_radix_select(row, N, k, ADAPTIVE, mtq, bsq, shs, T, need, bin_count);
    int jt = 0x7fffffff;
    if (need < bin_count) {                                        // block-uniform: a tie at the k-th place, the `need` lowest key indices win
        int run = 0;
        for (int j0 = 0; j0 < N; j0 += 256) {
            const int j = j0 + tid;
            const bool eq = j < N && wide_key(row[j], ADAPTIVE, mtq, bsq) == T;
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
    if (tid == 0) { sel[2 * ql] = (int32_t)T; sel[2 * ql + 1] = jt; }
}

// one block per query row: S row -> A row in place (dagl.py:256-261); saves the softmax shift and denominator
template <int MODE>
__global__ __launch_bounds__(256) void dense_softmax_fwd_kernel(int N, long long ldn, int L, int l0, int Lc,
                                                                float* __restrict__ sbuf, const float* __restrict__ mt,
                                                                const float* __restrict__ bs, float* __restrict__ lse,
                                                                int32_t* __restrict__ deg, float* __restrict__ rowsum, int b0,
                                                                const int32_t* __restrict__ sel) {
    __shared__ double shd[4];
    __shared__ float shf[4];
    const int lr = blockIdx.x, bi = blockIdx.y;
    const size_t ql = (size_t)(b0 + bi) * L + l0 + lr;
    float* row = sbuf + ((size_t)bi * Lc + lr) * ldn;
    const float mtq = MODE == 1 ? 0.f : mt[ql], bsq = MODE == 1 ? 0.f : bs[ql];
    const unsigned T = MODE == 0 ? 0u : (unsigned)sel[2 * ql];
    const int jt = MODE == 0 ? 0 : sel[2 * ql + 1];
    float mx = 0.f;                                    // masked keys have logit 0 (N > deg) -- and if all pass, max >= ... handled below
    int cnt = 0;
    for (int j = threadIdx.x; j < N; j += 256) {
        bool pass; float m;
        const float l = dt_logit<MODE>(row[j], j, mtq, bsq, T, jt, pass, m);
        cnt += pass ? 1 : 0;
        mx = pass ? fmaxf(mx, l) : mx;
    }
    // (logits of passing keys are positive: S > 0 and m > 0, or zero when S = 0; so max over all keys = max(0, ...) either way)
    const float M = dt_block_max(mx, shf);
    double z = 0.0;
    for (int j = threadIdx.x; j < N; j += 256) {
        bool pass; float m;
        const float l = dt_logit<MODE>(row[j], j, mtq, bsq, T, jt, pass, m);
        z += (double)expf(l - M);
    }
    const double Z = dt_block_sum(z, shd);
    const float invz = (float)(1.0 / Z);
    double rs = 0.0;
    for (int j = threadIdx.x; j < N; j += 256) {
        bool pass; float m;
        const float l = dt_logit<MODE>(row[j], j, mtq, bsq, T, jt, pass, m);
        const float a = pass ? expf(l - M) * invz : 0.f;
        row[j] = a;
        rs += (double)a;
    }
    for (int j = N + threadIdx.x; j < ldn; j += 256) row[j] = 0.f;           // pad columns: zero weights
    const double RS = dt_block_sum(rs, shd);
    const double C = dt_block_sum((double)cnt, shd);
    if (threadIdx.x == 0) {
        lse[2 * ql] = M; lse[2 * ql + 1] = (float)Z;
        deg[ql] = (int32_t)C; rowsum[ql] = (float)RS;
    }
}

// one block per query row: sbuf = recomputed S row, abuf = d A row  ->  sbuf = d S row, abuf = A row
template <int MODE>
__global__ __launch_bounds__(256) void dense_softmax_bwd_kernel(int N, long long ldn, int L, int l0, int Lc,
                                                                float* __restrict__ sbuf, float* __restrict__ abuf,
                                                                const float* __restrict__ mt, const float* __restrict__ bs,
                                                                const float* __restrict__ lse, const float* __restrict__ mu,
                                                                const float* __restrict__ thr, float* __restrict__ dthr,
                                                                float* __restrict__ dbias, float* __restrict__ dmu, int b0,
                                                                unsigned* __restrict__ ds_word, const int32_t* __restrict__ sel) {
    __shared__ double shd[4];
    const int lr = blockIdx.x, bi = blockIdx.y;
    const size_t ql = (size_t)(b0 + bi) * L + l0 + lr;
    float* srow = sbuf + ((size_t)bi * Lc + lr) * ldn;
    float* arow = abuf + ((size_t)bi * Lc + lr) * ldn;
    float ds_max = 0.f;
    const float mtq = MODE == 1 ? 0.f : mt[ql], bsq = MODE == 1 ? 0.f : bs[ql];
    const unsigned T = MODE == 0 ? 0u : (unsigned)sel[2 * ql];
    const int jt = MODE == 0 ? 0 : sel[2 * ql + 1];
    // The softmax's shift and denominator are formed HERE, from the recomputed scores -- not taken from the forward: the
    // forward may have run on the streamed split-fp16 kernel, whose scores differ from these in the last bits; logits of
    // several hundred turn that into ~1e-3 of every weight of the row, and the head biases' gradients (sums over all
    // queries that cancel to 1e-4 of their terms) are only right when the weights are normalised by their own sum.
    (void)lse;
    __shared__ float shf[4];
    float mx = 0.f;
#pragma unroll 16
    for (int j = threadIdx.x; j < N; j += 256) {
        bool pass; float m;
        const float l = dt_logit<MODE>(srow[j], j, mtq, bsq, T, jt, pass, m);
        mx = pass ? fmaxf(mx, l) : mx;
    }
    const float M = dt_block_max(mx, shf);
    double zz = 0.0;
#pragma unroll 16
    for (int j = threadIdx.x; j < N; j += 256) {
        bool pass; float m;
        const float l = dt_logit<MODE>(srow[j], j, mtq, bsq, T, jt, pass, m);
        zz += (double)expf(l - M);
    }
    const float invz = (float)(1.0 / dt_block_sum(zz, shd));
    double c = 0.0;
#pragma unroll 16
    for (int j = threadIdx.x; j < N; j += 256) {
        bool pass; float m;
        const float l = dt_logit<MODE>(srow[j], j, mtq, bsq, T, jt, pass, m);
        const float a = pass ? expf(l - M) * invz : 0.f;
        c += (double)a * (double)arow[j];
    }
    const float cf = (float)dt_block_sum(c, shd);
    double sdm = 0.0;
#pragma unroll 16
    for (int j = threadIdx.x; j < N; j += 256) {
        bool pass; float m;
        const float s = srow[j];
        const float l = dt_logit<MODE>(s, j, mtq, bsq, T, jt, pass, m);
        const float a = pass ? expf(l - M) * invz : 0.f;
        const float dl = a * (arow[j] - cf);           // non-neighbours: A = 0 and their logit is the constant 0
        const float ds = MODE == 1 ? SOFTMAX_SCALE * dl          // the 0/1 mask is a constant: d (10 S) / d S
                                   : SOFTMAX_SCALE * (m + s) * dl; // d S  (a = 0 -> 0)
        srow[j] = ds;
        ds_max = fmaxf(ds_max, fabsf(ds));
        arow[j] = a;
        sdm += (double)(SOFTMAX_SCALE * s * dl);       // d m
    }
    for (int j = N + threadIdx.x; j < ldn; j += 256) { srow[j] = 0.f; arow[j] = 0.f; }
    const float S = (float)dt_block_sum(sdm, shd);
    if (ds_word != nullptr) {                          // the split-fp16 products' scale for d S (order-independent integer maximum)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) ds_max = fmaxf(ds_max, __shfl_xor(ds_max, o));
        if ((threadIdx.x & 63) == 0 && ds_max > 0.f && ds_max < __builtin_inff()) atomicMax(ds_word, __float_as_uint(ds_max));
    }
    if (threadIdx.x == 0 && MODE != 1) {
        dbias[ql] = S;
        dthr[ql] = -mu[ql] * S;
        dmu[ql] = -thr[ql] * S;
    }
}

// (Measured and dropped: the row in registers -- S and d A read once, one expf per weight, 16-byte accesses; 512 threads x 32
// elements, 1 or 2 blocks per CU: 832 us against this kernel's 867 at [8, 1024 x 16384], whatever the variant.  2.1 GB in 0.83 ms;
// this kernel without its stores: 666 us -- one load in flight per thread and pass; the register form has all of a row's loads in
// flight but one block per CU, whose load / reduce / store phases do not overlap with anything: ~26 us per row and CU either way.)
static int launch_dense_softmax_bwd(hipStream_t s, int rows, int nb, int N, long long ldn, int L, int l0, int Lc, float* sbuf, float* abuf,
                                    const float* mt, const float* bs, const float* lse, const float* mu, const float* thr, float* dthr,
                                    float* dbias, float* dmu, int b0, unsigned* ds_word, int mode = DAGL_MODE_ADAPTIVE,
                                    const int32_t* sel = nullptr) {
#define DT_BWD(M_) hipLaunchKernelGGL(dense_softmax_bwd_kernel<M_>, dim3(rows, nb), dim3(256), 0, s, N, ldn, L, l0, Lc, sbuf, abuf, mt, bs, lse, \
                                      mu, thr, dthr, dbias, dmu, b0, ds_word, sel)
    if (mode == DAGL_MODE_TOPK) DT_BWD(1); else if (mode == DAGL_MODE_ADAPTIVE_TOPK) DT_BWD(2); else DT_BWD(0);
#undef DT_BWD
    DAGL_LAUNCH_CHECK("dense_softmax_bwd_kernel");
    return DAGL_OK;
}

// wide top-k modes: the rows' (T, jt) from the chunk's scores
static int launch_dt_select(hipStream_t s, int rows, int nb, int N, long long ldn, int L, int l0, int Lc, int k, int mode, const float* sbuf,
                            const float* mt, const float* bs, int32_t* sel, int b0) {
    if (mode == DAGL_MODE_ADAPTIVE_TOPK)
        hipLaunchKernelGGL(dt_select_kernel<true>, dim3(rows, nb), dim3(256), 0, s, N, ldn, L, l0, Lc, k, sbuf, mt, bs, sel, b0);
    else
        hipLaunchKernelGGL(dt_select_kernel<false>, dim3(rows, nb), dim3(256), 0, s, N, ldn, L, l0, Lc, k, sbuf, mt, bs, sel, b0);
    DAGL_LAUNCH_CHECK("dt_select_kernel");
    return DAGL_OK;
}

// out[r, c] += rs[r] * v[c] * scale   (the rank-one terms of the dense row mean)
__global__ void dt_rank1_add_kernel(size_t rows, int rows_per_batch, const float* __restrict__ rs, const float* __restrict__ vf,
                                    const double* __restrict__ vd, int vstride, float scale, float* __restrict__ out) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= rows * D) return;
    const size_t r = t / D; const int c = (int)(t - r * D);
    const size_t b = r / rows_per_batch;
    const float v = vf ? vf[b * vstride + c] : (float)(vd[b * vstride + c] * (double)scale);
    out[t] += (rs ? rs[r] : 1.0f) * (vf ? v * scale : v);
}

// d b2[b,c,y,x] = sum over the 49 keys whose window covers (y,x) of d V[key][(kh,kw,c)]; thread = pixel, fixed order
__global__ __launch_bounds__(256) void dt_fold_dv_kernel(Grid g, int nb, const float* __restrict__ dv, float* __restrict__ db2) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)nb * g.N) return;
    const size_t b = t / g.N; const size_t r = t - b * g.N;
    const int y = (int)(r / g.W), x = (int)(r - (size_t)y * g.W);
    float4 acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int kh = 0; kh < KS; ++kh) {
        const int jy = y + 3 - kh;
        if (jy < 0 || jy >= g.H) continue;
#pragma unroll
        for (int kw = 0; kw < KS; ++kw) {
            const int jx = x + 3 - kw;
            if (jx < 0 || jx >= g.W) continue;
            const float4* row = reinterpret_cast<const float4*>(dv + ((b * g.N + (size_t)jy * g.W + jx) * P + (kh * KS + kw) * CH));
#pragma unroll
            for (int q = 0; q < 4; ++q) { const float4 v = row[q]; acc[q].x += v.x; acc[q].y += v.y; acc[q].z += v.z; acc[q].w += v.w; }
        }
    }
    float* o = db2 + (b * CH * g.H + y) * g.W + x;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        o[(size_t)(4 * q + 0) * g.N] = acc[q].x; o[(size_t)(4 * q + 1) * g.N] = acc[q].y;
        o[(size_t)(4 * q + 2) * g.N] = acc[q].z; o[(size_t)(4 * q + 3) * g.N] = acc[q].w;
    }
}

static Gemm32 dt_gemm(int M, int N, int K, int batch, const float* A, long long lda, long long sA, int a_kc, const float* Bm,
                      long long ldb, long long sB, int b_kc, float* C, long long ldc, long long sC, float beta, bool grad = false) {
    Gemm32 g;
    g.M = M; g.N = N; g.K = K; g.batch = batch; g.A = A; g.lda = lda; g.sA = sA; g.a_kc = a_kc;
    g.B = Bm; g.ldb = ldb; g.sB = sB; g.b_kc = b_kc; g.C = C; g.ldc = ldc; g.sC = sC;
    g.alpha = 1.0f; g.beta = beta; g.bias = nullptr; g.relu = 0;
    // S: chains of 48 products; the long forward contraction A V: 128.  Gradient products (1e-3 bar, contractions of at
    // most a few thousand terms per chunk of queries / 16 384 keys) run unchunked: 140 instead of 236 registers, a third
    // block per CU.
    g.chunk_tiles = grad ? 0 : ((K == D) ? 3 : 8);
    return g;
}

static int dt_prepare(hipStream_t s, int B, const Grid& g, const DtPlan& p, void* ws, const float* wq_rows, const float* x_rows,
                      const float* b2, const float* thr, float* mu) {
    int rc;
    if ((rc = launch_pad_nhwc(s, B, g.H, g.W, b2, dt_at<float>(ws, p.o_b2p)))) return rc;
    if (thr == nullptr) return DAGL_OK;                  // (the fixed-k mode has no threshold heads)
    if ((rc = launch_colsum_rows(s, B, g.N, x_rows, dt_at<double>(ws, p.o_colsum)))) return rc;
    hipLaunchKernelGGL(dt_thresholds_kernel, dim3((g.L + 3) / 4, B), dim3(256), 0, s, g.L, g.N, wq_rows,
                       dt_at<double>(ws, p.o_colsum), thr, dt_at<float>(ws, p.o_mt), mu);
    DAGL_LAUNCH_CHECK("dt_thresholds_kernel");
    return DAGL_OK;
}

int launch_dense_train_forward(hipStream_t s, int B, const Grid& g, const float* wq_rows, const float* x_rows, const float* b2,
                               const float* thr, const float* bias, float* out, float* lse, float* mu, void* ws, size_t ws_bytes,
                               int64_t* stats_dev, int mode, int k) {
    const DtPlan p = dt_plan(B, g, false);
    if (lse == nullptr) lse = dt_at<float>(ws, p.o_lse);
    if (mu == nullptr) mu = dt_at<float>(ws, p.o_mu);
    int32_t* sel = dt_at<int32_t>(ws, p.o_sel);
    if (ws_bytes < p.o_end) { set_error("dense forward: workspace %zu B < required %zu B", ws_bytes, p.o_end); return DAGL_ERR_WORKSPACE; }
    int rc;
    if ((rc = dt_prepare(s, B, g, p, ws, wq_rows, x_rows, b2, thr, mu))) return rc;
    float* sbuf = dt_at<float>(ws, p.o_sbuf);
    float* vrows = dt_at<float>(ws, p.o_vrows);
    float* agg = dt_at<float>(ws, p.o_agg);
    const float* b2p = dt_at<float>(ws, p.o_b2p);
    const float* mt = dt_at<float>(ws, p.o_mt);
    int32_t* deg = dt_at<int32_t>(ws, p.o_deg);
    float* rowsum = dt_at<float>(ws, p.o_rowsum);
    for (int b0 = 0; b0 < B; b0 += p.Bc) {
        const int nb = (B - b0 < p.Bc) ? B - b0 : p.Bc;
        if ((rc = launch_unfold_values(s, nb, g, b2p + (size_t)b0 * g.Hp * g.Wp * CH, vrows))) return rc;
        for (int l0 = 0; l0 < g.L; l0 += p.Lc) {
            const int lc = (g.L - l0 < p.Lc) ? g.L - l0 : p.Lc;
            // S = Wq X^T
            if ((rc = launch_gemm32(s, dt_gemm(lc, g.N, D, nb, wq_rows + ((size_t)b0 * g.L + l0) * D, D, (long long)g.L * D, 1,
                                               x_rows + (size_t)b0 * g.N * D, D, (long long)g.N * D, 1,
                                               sbuf, p.ldn, (long long)p.Lc * p.ldn, 0.f)))) return rc;
            if (mode != DAGL_MODE_ADAPTIVE)
                if ((rc = launch_dt_select(s, lc, nb, g.N, p.ldn, g.L, l0, p.Lc, k, mode, sbuf, mt, bias, sel, b0))) return rc;
#define DT_FWD(M_) hipLaunchKernelGGL(dense_softmax_fwd_kernel<M_>, dim3(lc, nb), dim3(256), 0, s, g.N, p.ldn, g.L, l0, p.Lc, sbuf, mt, bias, \
                                      lse, deg, rowsum, b0, sel)
            if (mode == DAGL_MODE_TOPK) DT_FWD(1); else if (mode == DAGL_MODE_ADAPTIVE_TOPK) DT_FWD(2); else DT_FWD(0);
#undef DT_FWD
            DAGL_LAUNCH_CHECK("dense_softmax_fwd_kernel");
            // agg = A V
            if ((rc = launch_gemm32(s, dt_gemm(lc, P, g.N, nb, sbuf, p.ldn, (long long)p.Lc * p.ldn, 1,
                                               vrows, P, (long long)g.N * P, 0,
                                               agg + ((size_t)b0 * g.L + l0) * P, P, (long long)g.L * P, 0.f)))) return rc;
        }
    }
    if ((rc = launch_fold(s, B, g, agg, out))) return rc;
    if (stats_dev) if ((rc = launch_degree_stats(s, (size_t)B * g.L, deg, stats_dev))) return rc;
    return DAGL_OK;
}


// ---- split-fp16 operands of the backward's five products (gemm16s.hip) --------------------------------------------------------
// src fp32 [R][C] (leading dimension ld) -> hi / lo [R][Cp] halfs (row stride ldo), s x = hi + lo, pad columns zero; s from *word (largest magnitude
// of the tensor, fcg_scale_of) or `fixed` when word is null.  Thread = (row, octet of columns).
__global__ __launch_bounds__(256) void dt_split_rows_kernel(size_t R, int C, long long ld, int Cp, long long ldo, const float* __restrict__ src,
                                                            const unsigned* __restrict__ word, float fixed,
                                                            unsigned short* __restrict__ hi, unsigned short* __restrict__ lo) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int o8 = Cp / 8;
    if (t >= R * o8) return;
    const size_t r = t / o8; const int c0 = 8 * (int)(t - r * o8);
    const float sc = word ? fcg_scale_of(*word) : fixed;
    const float* p = src + r * ld + c0;
    unsigned short vh[8], vl[8];
    if (c0 + 8 <= C) {
        const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
        g16_split(a.x * sc, vh[0], vl[0]); g16_split(a.y * sc, vh[1], vl[1]); g16_split(a.z * sc, vh[2], vl[2]); g16_split(a.w * sc, vh[3], vl[3]);
        g16_split(b.x * sc, vh[4], vl[4]); g16_split(b.y * sc, vh[5], vl[5]); g16_split(b.z * sc, vh[6], vl[6]); g16_split(b.w * sc, vh[7], vl[7]);
    } else {
#pragma unroll
        for (int u = 0; u < 8; ++u) g16_split((c0 + u < C) ? p[u] * sc : 0.f, vh[u], vl[u]);
    }
    *reinterpret_cast<uint4*>(hi + r * ldo + c0) = *reinterpret_cast<const uint4*>(vh);
    *reinterpret_cast<uint4*>(lo + r * ldo + c0) = *reinterpret_cast<const uint4*>(vl);
}

// src fp32 [batch][R][C] (ld, batch stride ss) -> transposed hi / lo [batch][C][Rp] halfs (row stride ldo; Rp >= R a multiple of 32, pad zero).
// Block = 64 rows x 64 columns: 256-byte row reads, transposed in LDS, 128-byte runs written.
__global__ __launch_bounds__(256) void dt_split_transpose_kernel(int R, int C, long long ld, long long ss, int Rp, long long ldo, long long sd,
                                                                 const float* __restrict__ src, const unsigned* __restrict__ word,
                                                                 float fixed, unsigned short* __restrict__ hi,
                                                                 unsigned short* __restrict__ lo) {
    __shared__ __attribute__((aligned(16))) unsigned short th[64][64 + 8];
    __shared__ __attribute__((aligned(16))) unsigned short tl[64][64 + 8];
    const int r0 = blockIdx.x * 64, c0 = blockIdx.y * 64, b = blockIdx.z;
    const float sc = word ? fcg_scale_of(*word) : fixed;
    const float* sb = src + (long long)b * ss;
    float v[16];                                                             // all of a thread's 16 loads in flight, then the splits
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const int e = threadIdx.x + 256 * u, r = e >> 6, c = e & 63;
        v[u] = (r0 + r < R && c0 + c < C) ? sb[(long long)(r0 + r) * ld + c0 + c] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const int e = threadIdx.x + 256 * u, r = e >> 6, c = e & 63;
        g16_split(v[u] * sc, th[c][r], tl[c][r]);
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 64 * 8; e += 256) {                        // (column, octet of rows)
        const int c = e >> 3, r8 = e & 7;
        if (c0 + c >= C || r0 + 8 * r8 >= Rp) continue;
        const long long o = (long long)b * sd + (long long)(c0 + c) * ldo + r0 + 8 * r8;
        *reinterpret_cast<uint4*>(hi + o) = *reinterpret_cast<const uint4*>(&th[c][8 * r8]);
        *reinterpret_cast<uint4*>(lo + o) = *reinterpret_cast<const uint4*>(&tl[c][8 * r8]);
    }
}

// value patches of the zero-bordered map, split: out[b][n][(kh,kw,c)] (50 taps of 16 halfs: tap 49 = zero pad); thread = (key, tap)
__global__ __launch_bounds__(256) void dt_unfold_values_split_kernel(int H, int W, size_t total, const float* __restrict__ mp,
                                                                     const unsigned* __restrict__ word,
                                                                     unsigned short* __restrict__ hi, unsigned short* __restrict__ lo) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;                                                  // total = nb * N * 50
    const int Wp = W + 2 * PADPIX, Hp = H + 2 * PADPIX;
    const size_t key = t / 50; const int tap = (int)(t - key * 50);
    const size_t b = key / ((size_t)H * W); const int n = (int)(key - b * (size_t)H * W);
    const int y = n / W, x = n - y * W;
    const float sc = fcg_scale_of(*word);
    unsigned short vh[16], vl[16];
    if (tap < KS * KS) {
        const int kh = tap / KS, kw = tap - kh * KS;
        const float4* src = reinterpret_cast<const float4*>(mp + ((b * Hp + y + kh) * (size_t)Wp + x + kw) * CH);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 v = src[q];
            g16_split(v.x * sc, vh[4 * q], vl[4 * q]); g16_split(v.y * sc, vh[4 * q + 1], vl[4 * q + 1]);
            g16_split(v.z * sc, vh[4 * q + 2], vl[4 * q + 2]); g16_split(v.w * sc, vh[4 * q + 3], vl[4 * q + 3]);
        }
    } else {
#pragma unroll
        for (int u = 0; u < 16; ++u) { vh[u] = 0; vl[u] = 0; }
    }
    uint4* oh = reinterpret_cast<uint4*>(hi + t * 16); uint4* ol = reinterpret_cast<uint4*>(lo + t * 16);
    oh[0] = reinterpret_cast<const uint4*>(vh)[0]; oh[1] = reinterpret_cast<const uint4*>(vh)[1];
    ol[0] = reinterpret_cast<const uint4*>(vl)[0]; ol[1] = reinterpret_cast<const uint4*>(vl)[1];
}

static int dt_split_rows(hipStream_t s, size_t R, int C, long long ld, int Cp, long long ldo, const float* src, const unsigned* word,
                         float fixed, unsigned short* hi, size_t halfs) {
    const size_t items = R * (size_t)(Cp / 8);
    hipLaunchKernelGGL(dt_split_rows_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, s, R, C, ld, Cp, ldo, src, word, fixed, hi, hi + halfs);
    DAGL_LAUNCH_CHECK("dt_split_rows_kernel");
    return DAGL_OK;
}

static int dt_split_transpose(hipStream_t s, int nb, int R, int C, long long ld, long long ss, int Rp, long long ldo, const float* src,
                              const unsigned* word, float fixed, unsigned short* hi, size_t halfs) {
    hipLaunchKernelGGL(dt_split_transpose_kernel, dim3((Rp + 63) / 64, (C + 63) / 64, nb), dim3(256), 0, s, R, C, ld, ss, Rp, ldo,
                       (long long)C * ldo, src, word, fixed, hi, hi + halfs);
    DAGL_LAUNCH_CHECK("dt_split_transpose_kernel");
    return DAGL_OK;
}

static Gemm16s dt_gemm16(int M, int N, int K, int nb, const unsigned short* a, size_t a_halfs, long long lda, long long sA,
                         const unsigned short* b, size_t b_halfs, long long ldb, long long sB, float* C, long long ldc, long long sC,
                         const unsigned* wa, const unsigned* wb, float alpha0) {
    Gemm16s g;
    g.M = M; g.N = N; g.K = K; g.a_hi = a; g.a_lo = a + a_halfs; g.b_hi = b; g.b_lo = b + b_halfs; g.lda = lda; g.ldb = ldb;
    g.a_rows = M; g.b_rows = N; g.C = C; g.ldc = ldc; g.part = nullptr; g.slices = 1; g.scale_word = wa; g.scale_word_b = wb;
    g.alpha0 = alpha0; g.batch = nb; g.sA = sA; g.sB = sB; g.sC = sC;
    return g;
}

// The backward of one image group (all queries in one chunk) with its five products on the fp16 matrix cores, split operands:
// 580 GFLOP per head at [8, 128 x 128] ran at 85 TFLOP/s on the fp32 matrix cores (54 % of their peak) -- 6.8 ms per head and
// step, half of the adaptive-mode training step.  Operand copies: K-contiguous rows for both sides of every product, i.e.
// d agg, V, Wq, X by rows, and d agg, X, Wq, d S, A transposed; every tensor scaled by a power of two taken from its largest
// magnitude (A <= 1: 2^13).  The softmax backward between the products stays in fp32 on fp32 S and d A.
static int dt_backward_group16(hipStream_t s, const Grid& g, const DtPlan& p, void* ws, int b0, int nb, const float* wq_rows,
                               const float* x_rows, const float* thr, const float* bias, const float* lse, const float* mu,
                               float* dwq_rows, float* dx_rows, float* dthr, float* dbias) {
    int rc;
    unsigned* words = dt_at<unsigned>(ws, p.o_words);          // 0 d agg, 1 values, 2 X, 3 Wq, 4 d S
    float* sbuf = dt_at<float>(ws, p.o_sbuf);
    float* abuf = dt_at<float>(ws, p.o_abuf);
    float* dvrows = dt_at<float>(ws, p.o_dvrows);
    const float* dagg = dt_at<float>(ws, p.o_dagg) + (size_t)b0 * g.L * P;
    const float* b2p = dt_at<float>(ws, p.o_b2p) + (size_t)b0 * g.Hp * g.Wp * CH;
    const float* mt = dt_at<float>(ws, p.o_mt);
    float* dmu = dt_at<float>(ws, p.o_dmu);
    const float* wq = wq_rows + (size_t)b0 * g.L * D;
    const float* xr = x_rows + (size_t)b0 * g.N * D;
    auto H = [&](size_t o) { return dt_at<unsigned short>(ws, o); };
    const int L = g.L, N = g.N, Lp = p.Lp;
    const long long ldn = p.ldn, ldl = p.ldl, ldk = p.ldk;
    DAGL_HIP_TRY(hipMemsetAsync(words, 0, 256, s));
    if ((rc = launch_absmax(s, (size_t)nb * L * P, dagg, words + 0))) return rc;
    if ((rc = launch_absmax(s, (size_t)nb * g.Hp * g.Wp * CH, b2p, words + 1))) return rc;
    if ((rc = launch_absmax(s, (size_t)nb * N * D, xr, words + 2))) return rc;
    if ((rc = launch_absmax(s, (size_t)nb * L * D, wq, words + 3))) return rc;
    // operand copies that do not depend on the softmax
    if ((rc = dt_split_rows(s, (size_t)nb * L, P, P, DT_PK, DT_PK, dagg, words + 0, 1.f, H(p.o_dgk), p.dgk_h))) return rc;
    if ((rc = dt_split_transpose(s, nb, L, P, P, (long long)L * P, Lp, ldl, dagg, words + 0, 1.f, H(p.o_dgt), p.dgt_h))) return rc;
    {
        const size_t total = (size_t)nb * N * 50;
        hipLaunchKernelGGL(dt_unfold_values_split_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, g.H, g.W, total, b2p,
                           words + 1, H(p.o_vk), H(p.o_vk) + p.vk_h);
        DAGL_LAUNCH_CHECK("dt_unfold_values_split_kernel");
    }
    if ((rc = dt_split_rows(s, (size_t)nb * N, D, D, DT_DK, DT_DK, xr, words + 2, 1.f, H(p.o_xk), p.xk_h))) return rc;
    if ((rc = dt_split_transpose(s, nb, N, D, D, (long long)N * D, p.nk, ldk, xr, words + 2, 1.f, H(p.o_xt), p.xt_h))) return rc;
    if ((rc = dt_split_rows(s, (size_t)nb * L, D, D, DT_DK, DT_DK, wq, words + 3, 1.f, H(p.o_wqk), p.wqk_h))) return rc;
    if ((rc = dt_split_transpose(s, nb, L, D, D, (long long)L * D, Lp, ldl, wq, words + 3, 1.f, H(p.o_wqt), p.wqt_h))) return rc;
    const long long sS = (long long)p.Lc * ldn;
    // d A = d agg V^T ; S = Wq X^T
    if ((rc = launch_gemm16s(s, dt_gemm16(L, N, DT_PK, nb, H(p.o_dgk), p.dgk_h, DT_PK, (long long)L * DT_PK, H(p.o_vk), p.vk_h, DT_PK,
                                          (long long)N * DT_PK, abuf, ldn, sS, words + 0, words + 1, 1.f)))) return rc;
    if ((rc = launch_gemm16s(s, dt_gemm16(L, N, DT_DK, nb, H(p.o_wqk), p.wqk_h, DT_DK, (long long)L * DT_DK, H(p.o_xk), p.xk_h, DT_DK,
                                          (long long)N * DT_DK, sbuf, ldn, sS, words + 3, words + 2, 1.f)))) return rc;
    if ((rc = launch_dense_softmax_bwd(s, L, nb, N, ldn, L, 0, p.Lc, sbuf, abuf, mt, bias, lse, mu, thr, dthr, dbias, dmu, b0, words + 4))) return rc;
    // d S by rows and transposed, A transposed
    if ((rc = dt_split_rows(s, (size_t)nb * p.Lc, (int)ldn, ldn, p.nk, ldk, sbuf, words + 4, 1.f, H(p.o_dsk), p.dsk_h))) return rc;
    if ((rc = dt_split_transpose(s, nb, L, N, ldn, sS, Lp, ldl, sbuf, words + 4, 1.f, H(p.o_dst), p.dst_h))) return rc;
    if ((rc = dt_split_transpose(s, nb, L, N, ldn, sS, Lp, ldl, abuf, nullptr, 8192.f, H(p.o_at), p.at_h))) return rc;
    // d Wq = d S X (split over the keys)
    {
        Gemm16s q = dt_gemm16(L, D, p.nk / p.kslices, nb, H(p.o_dsk), p.dsk_h, ldk, (long long)p.Lc * ldk, H(p.o_xt), p.xt_h, ldk, (long long)D * ldk,
                              dwq_rows + (size_t)b0 * L * D, D, (long long)L * D, words + 4, words + 2, 1.f);
        q.slices = p.kslices; q.part = dt_at<float>(ws, p.o_part);
        if ((rc = launch_gemm16s(s, q))) return rc;
    }
    // d X = d S^T Wq ; d V = A^T d agg
    if ((rc = launch_gemm16s(s, dt_gemm16(N, D, Lp, nb, H(p.o_dst), p.dst_h, ldl, (long long)N * ldl, H(p.o_wqt), p.wqt_h, ldl,
                                          (long long)D * ldl, dx_rows + (size_t)b0 * N * D, D, (long long)N * D, words + 4, words + 3, 1.f)))) return rc;
    if ((rc = launch_gemm16s(s, dt_gemm16(N, P, Lp, nb, H(p.o_at), p.at_h, ldl, (long long)N * ldl, H(p.o_dgt), p.dgt_h, ldl,
                                          (long long)P * ldl, dvrows, P, (long long)N * P, nullptr, words + 0, 1.f / 8192.f)))) return rc;
    return DAGL_OK;
}

int launch_dense_train_backward(hipStream_t s, int B, const Grid& g, const float* wq_rows, const float* x_rows, const float* b2,
                                const float* thr, const float* bias, const float* lse, const float* mu_saved, const float* dout,
                                float* dwq_rows, float* dx_rows, float* db2, float* dthr, float* dbias, void* ws, size_t ws_bytes,
                                bool fp32_products, int mode, int k) {
    const DtPlan p = dt_plan(B, g, true);
    // the wide top-k modes re-select from the recomputed scores: the fp32 product of the forward, bit for bit (a split-fp16 S could
    // order two near-equal scores the other way round)
    const bool h16 = p.h16 && !fp32_products && mode == DAGL_MODE_ADAPTIVE;
    int32_t* sel = dt_at<int32_t>(ws, p.o_sel);
    if (ws_bytes < p.o_end) { set_error("dense backward: workspace %zu B < required %zu B", ws_bytes, p.o_end); return DAGL_ERR_WORKSPACE; }
    int rc;
    float* mu = dt_at<float>(ws, p.o_rowsum);                    // recomputed with the thresholds (same values as the forward's)
    if ((rc = dt_prepare(s, B, g, p, ws, wq_rows, x_rows, b2, thr, mu))) return rc;
    (void)mu_saved;
    float* sbuf = dt_at<float>(ws, p.o_sbuf);
    float* abuf = dt_at<float>(ws, p.o_abuf);
    float* vrows = dt_at<float>(ws, p.o_vrows);
    float* dvrows = dt_at<float>(ws, p.o_dvrows);
    float* dagg = dt_at<float>(ws, p.o_dagg);
    const float* b2p = dt_at<float>(ws, p.o_b2p);
    const float* mt = dt_at<float>(ws, p.o_mt);
    float* dmu = dt_at<float>(ws, p.o_dmu);
    float* dxbar = dt_at<float>(ws, p.o_dxbar);
    const double* colsum = dt_at<double>(ws, p.o_colsum);
    if ((rc = launch_unfold_dout(s, B, g, dout, dagg))) return rc;
    for (int b0 = 0; b0 < B; b0 += p.Bc) {
        const int nb = (B - b0 < p.Bc) ? B - b0 : p.Bc;
        if (h16) {
            if ((rc = dt_backward_group16(s, g, p, ws, b0, nb, wq_rows, x_rows, thr, bias, lse, mu, dwq_rows, dx_rows, dthr, dbias))) return rc;
        } else {
        if ((rc = launch_unfold_values(s, nb, g, b2p + (size_t)b0 * g.Hp * g.Wp * CH, vrows))) return rc;
        for (int l0 = 0; l0 < g.L; l0 += p.Lc) {
            const int lc = (g.L - l0 < p.Lc) ? g.L - l0 : p.Lc;
            const float* wq_c = wq_rows + ((size_t)b0 * g.L + l0) * D;
            const float* dagg_c = dagg + ((size_t)b0 * g.L + l0) * P;
            const long long sS = (long long)p.Lc * p.ldn;
            const float beta = (l0 == 0) ? 0.f : 1.f;
            // d A = d agg V^T ; S = Wq X^T
            if ((rc = launch_gemm32(s, dt_gemm(lc, g.N, P, nb, dagg_c, P, (long long)g.L * P, 1, vrows, P, (long long)g.N * P, 1,
                                               abuf, p.ldn, sS, 0.f, true)))) return rc;
            if ((rc = launch_gemm32(s, dt_gemm(lc, g.N, D, nb, wq_c, D, (long long)g.L * D, 1,
                                               x_rows + (size_t)b0 * g.N * D, D, (long long)g.N * D, 1, sbuf, p.ldn, sS, 0.f)))) return rc;
            if (mode != DAGL_MODE_ADAPTIVE)
                if ((rc = launch_dt_select(s, lc, nb, g.N, p.ldn, g.L, l0, p.Lc, k, mode, sbuf, mt, bias, sel, b0))) return rc;
            if ((rc = launch_dense_softmax_bwd(s, lc, nb, g.N, p.ldn, g.L, l0, p.Lc, sbuf, abuf, mt, bias, lse, mu, thr, dthr, dbias, dmu, b0,
                                               nullptr, mode, sel))) return rc;
            // d Wq = d S X
            if ((rc = launch_gemm32(s, dt_gemm(lc, D, g.N, nb, sbuf, p.ldn, sS, 1, x_rows + (size_t)b0 * g.N * D, D, (long long)g.N * D, 0,
                                               dwq_rows + ((size_t)b0 * g.L + l0) * D, D, (long long)g.L * D, 0.f, true)))) return rc;
            // d X (+)= d S^T Wq
            if ((rc = launch_gemm32(s, dt_gemm(g.N, D, lc, nb, sbuf, p.ldn, sS, 0, wq_c, D, (long long)g.L * D, 0,
                                               dx_rows + (size_t)b0 * g.N * D, D, (long long)g.N * D, beta, true)))) return rc;
            // d V (+)= A^T d agg
            if ((rc = launch_gemm32(s, dt_gemm(g.N, P, lc, nb, abuf, p.ldn, sS, 0, dagg_c, P, (long long)g.L * P, 0,
                                               dvrows, P, (long long)g.N * P, beta, true)))) return rc;
        }
        }
        {
            const size_t n = (size_t)nb * g.N;
            hipLaunchKernelGGL(dt_fold_dv_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, g, nb, dvrows,
                               db2 + (size_t)b0 * CH * g.N);
            DAGL_LAUNCH_CHECK("dt_fold_dv_kernel");
        }
    }
    // dense mean term: d Wq_l += d mu_l Xbar ;  d X_j += (sum_l d mu_l Wq_l) / N   (the fixed-k mode has no threshold)
    if (mode != DAGL_MODE_TOPK) {
        const size_t nq = (size_t)B * g.L * D, nk = (size_t)B * g.N * D;
        hipLaunchKernelGGL(dt_rank1_add_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, s, (size_t)B * g.L, g.L, dmu, nullptr,
                           colsum, DS, 1.0f / (float)g.N, dwq_rows);
        DAGL_LAUNCH_CHECK("dt_rank1_add_kernel");
        if ((rc = launch_dxbar(s, B, g.L, wq_rows, dmu, dxbar))) return rc;
        hipLaunchKernelGGL(dt_rank1_add_kernel, dim3((unsigned)((nk + 255) / 256)), dim3(256), 0, s, (size_t)B * g.N, g.N, nullptr, dxbar,
                           nullptr, D, 1.0f / (float)g.N, dx_rows);
        DAGL_LAUNCH_CHECK("dt_rank1_add_kernel");
    }
    return DAGL_OK;
}

}  // namespace dagl
