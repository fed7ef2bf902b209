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
size_t topk_wide_workspace_bytes(int N, int L) {
    const size_t R = (size_t)topk_wide_rows(N, L);
    const size_t ldn = (size_t)(N + 31) / 32 * 32;
    return align_up(R * ldn * sizeof(float), 256) + align_up(R * ROW_CHUNKS * ROW_PART_FLOATS * sizeof(float), 256) +
           align_up(R * 4 * sizeof(int32_t), 256) + align_up(R * WIDE_RANGES * sizeof(int32_t), 256);
}

int launch_topk_wide(hipStream_t s, int B, const Grid& g, int mode, int k, const float* wq, const float* x, const float* mt,
                     const float* bs, const float* b2p, void* ws, float* agg, int32_t* deg, float* rowsum) {
    WideArgs a;
    memset(&a, 0, sizeof(a));
    a.g = g; a.mode = mode; a.k = k; a.mt = mt; a.bs = bs; a.b2p = b2p; a.agg = agg; a.deg = deg; a.rowsum = rowsum;
    const int Rmax = topk_wide_rows(g.N, g.L);
    a.ldn = (g.N + 31) / 32 * 32;
    char* p = static_cast<char*>(ws);
    a.scores = reinterpret_cast<float*>(p); p += align_up((size_t)Rmax * a.ldn * sizeof(float), 256);
    a.part = reinterpret_cast<float*>(p); p += align_up((size_t)Rmax * ROW_CHUNKS * ROW_PART_FLOATS * sizeof(float), 256);
    a.sel = reinterpret_cast<int32_t*>(p); p += align_up((size_t)Rmax * 4 * sizeof(int32_t), 256);
    a.eq_before = reinterpret_cast<int32_t*>(p);
    const int rows_q = feat_rows(g.L), rows_x = feat_rows(g.N);
    for (int b = 0; b < B; ++b)
        for (int r0 = 0; r0 < g.L; r0 += Rmax) {
            const int R = (g.L - r0 < Rmax) ? g.L - r0 : Rmax;
            a.b = b; a.r0 = r0; a.R = R;
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
            hipLaunchKernelGGL(wide_select_kernel, dim3(R), dim3(256), 0, s, a);
            DAGL_LAUNCH_CHECK("wide_select_kernel");
            hipLaunchKernelGGL(wide_attend_kernel, dim3(ROW_CHUNKS, R < 64 ? R : 64), dim3(256), 0, s, a);
            DAGL_LAUNCH_CHECK("wide_attend_kernel");
            hipLaunchKernelGGL(wide_combine_kernel, dim3(R < 1024 ? R : 1024), dim3(256), 0, s, a);
            DAGL_LAUNCH_CHECK("wide_combine_kernel");
        }
    return DAGL_OK;
}

}  // namespace dagl
