// Long-tailed adaptive neighbourhoods behind the screen: the few queries whose neighbourhood does not fit the fixed-width
// list (more than DAGL_LIST_CAP passing keys, or more candidates than a screen segment has slots) are redone ONE BY ONE
// in the dense formulation of the reference (DN_Gray/model/dagl.py:250-264) instead of sending the whole call to the
// fp32 scan: an adaptive mask with a mean degree of ~8 has maxima of several hundred (measured at 256^2: mean 7.7,
// maximum 890, 10 of 4096 queries beyond 256), and those 10 rows are 10 x N scores, not L x N.
//   A  gather + scores     scores of every flagged query against ALL keys: a VALU kernel for up to OVF_SMALL flagged queries,
//                          beyond that ONE matrix product [flagged, 196] x [196, N] on the fp32 matrix cores; either way every
//                          key row is read once for all flagged queries (the row count lives in device memory: blocks past
//                          it exit)
//   B  stats / attend / combine (below): mask, softmax over all N keys (masked keys count e^0), weighted sum of the value
//                          patches straight from the value map; overwrites the query's aggregated row
// All kernels read the number of flagged queries from device memory and exit at once when it is zero.
#include "dagl_common.h"
#include "aggregate_direct.h"
#include "row_attend.h"

namespace dagl {

__device__ __forceinline__ float ovf_logit(float s, float mtq, float bsq, bool& pass) {
    const float m = (s - mtq) + bsq;                                      // dagl.py:256, same expression order
    pass = m > 0.f;
    return pass ? __fmul_rn(__fmul_rn(s, m), SOFTMAX_SCALE) : 0.f;
}

constexpr int OVF_SMALL = 16;     // up to this many flagged queries the scores come from a VALU kernel, beyond that from ONE matrix
                                  // product [flagged, 196] x [196, N] on the fp32 matrix cores
constexpr int OVF_GRID_B = 256;         // combine: fixed small grid (an empty call costs one wave of exits), flagged queries strided

// how many flagged queries this call redoes (0 when the list overflowed: the host, or the veto word, sends the call elsewhere --
// redoing 256 rows of a DENSE mask one by one costs 17 ms at 256^2), and in which form
__device__ __forceinline__ int ovf_rows_served(const OvfArgs& a) { const int nf = *a.count; return nf > a.cap ? 0 : nf; }
__device__ __forceinline__ int ovf_small_limit(const OvfArgs& a) { return a.cap < OVF_SMALL ? a.cap : OVF_SMALL; }


// Few flagged queries (<= OVF_SMALL; 10 at 256^2 mean degree 8): a 128-row matrix-core tile would be 90 % padding (45 us);
// instead four lanes share a key, each sums a quarter of the 49 float4 products per flagged row (query rows broadcast from
// LDS), two shuffle steps finish the dot product.  Every key row is read once for all flagged queries.
__device__ __forceinline__ void ovf_scores_small_block(const OvfArgs& a, int kblock, int b, float4 (*sq)[DS / 4]) {
    const int nf = ovf_rows_served(a);
    if (nf == 0 || nf > ovf_small_limit(a)) return;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int k = lane >> 2, qd = lane & 3;
    const long long key = ((long long)kblock * 4 + w) * 16 + k;
    const bool ok = key < a.g.N;
    const float4* xr = reinterpret_cast<const float4*>(a.x + ((size_t)b * a.rows_x + (ok ? key : 0)) * DS);
    // the lane's 13 pieces of the key row are requested together, ahead of the query rows (the row is read once, from HBM: one
    // round trip, not 13)
    constexpr int NC = (D / 4 + 3) / 4;                                          // 13
    float4 xs[NC];
#pragma unroll
    for (int u = 0; u < NC; ++u) {
        const int c = qd + 4 * u;
        const float4 raw = xr[c < D / 4 ? c : D / 4 - 1];                        // clamped, zeroed below (no predicated load)
        xs[u] = (c < D / 4) ? raw : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (int e = tid; e < nf * (DS / 4); e += 256) sq[e / (DS / 4)][e % (DS / 4)] = reinterpret_cast<const float4*>(a.qrows)[e];
    __syncthreads();
    float acc[OVF_SMALL];
#pragma unroll
    for (int r = 0; r < OVF_SMALL; ++r) acc[r] = 0.f;
#pragma unroll
    for (int u = 0; u < NC; ++u) {
        const int c = min(qd + 4 * u, D / 4 - 1);
        const float4 x = xs[u];
#pragma unroll
        for (int r = 0; r < OVF_SMALL; ++r)
            if (r < nf) {                                                        // wave-uniform
                const float4 q = sq[r][c];
                acc[r] = fmaf(q.x, x.x, fmaf(q.y, x.y, fmaf(q.z, x.z, fmaf(q.w, x.w, acc[r]))));
            }
    }
#pragma unroll
    for (int r = 0; r < OVF_SMALL; ++r)
        if (r < nf) {
            const float sum = quad_sum_f32(acc[r]);
            if (ok && qd == 0) a.scores[((size_t)b * a.cap + r) * a.ldn + key] = sum;
        }
}

// ONE launch for two independent jobs behind the refine kernels: blocks [0, score_blocks * B) score the flagged queries (few of
// them: the form above; an empty or long list: they exit), the rest are aggregate_direct_kernel's blocks -- the gather + weighted
// sum over every query's list, which skips the flagged rows (count -1).  Run one after the other the two cost 16.6 + 25 us at
// 256^2; both are chains of memory round trips that leave the CUs mostly idle, so they share them.
__global__ __launch_bounds__(256) void ovf_scores_aggregate_kernel(OvfArgs a, AggArgs ag, int score_blocks) {
    __shared__ float4 sq[OVF_SMALL][DS / 4];                                    // 12.75 KiB
    __shared__ int sh_of[AGG_STAGE];
    __shared__ float sh_w[AGG_STAGE];
    const int sb = score_blocks * a.B;
    if ((int)blockIdx.x < sb) { ovf_scores_small_block(a, blockIdx.x % score_blocks, blockIdx.x / score_blocks, sq); return; }
    const int q = blockIdx.x - sb;
    aggregate_direct_block(ag, q / ag.g.L, q % ag.g.L, sh_of, sh_w);
}

// Passing keys per (flagged query, key chunk), nothing else: what the call's statistics need.  A call that WAITS for its verdict
// runs this and the statistics block first, queues the read-back, and only then the gathers (list gather, attend, combine): the
// host round trip and the enqueueing of the caller's next launches run under them.
__global__ __launch_bounds__(256) void ovf_count_kernel(OvfArgs a) {
    __shared__ int shc[4];
    const int nf = ovf_rows_served(a);
    if (nf == 0) return;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    int j0c, j1c; row_chunk_range(a.g.N, blockIdx.x, j0c, j1c);
    int j0, j1; row_wave_range(j0c, j1c, w, j0, j1);
    for (int slot = blockIdx.y; slot < nf; slot += gridDim.y) {
        const size_t ql = (size_t)a.list[slot];
        const int b = (int)(ql / a.g.L);
        const float* row = a.scores + ((size_t)b * a.cap + slot) * a.ldn;
        const float mtq = a.mt[ql], bsq = a.bs[ql];
        int cnt = 0;
        for (int c0 = j0; c0 < j1; c0 += 64) {
            const int j = c0 + lane; bool pass = false;
            if (j < j1) (void)ovf_logit(row[j], mtq, bsq, pass);
            cnt += __popcll(__ballot(pass));
        }
        if (lane == 0) shc[w] = cnt;
        __syncthreads();
        if (tid == 0) reinterpret_cast<int*>(a.part + ((size_t)slot * ROW_CHUNKS + blockIdx.x) * ROW_PART_FLOATS + P)[1] = shc[0] + shc[1] + shc[2] + shc[3];
        __syncthreads();
    }
}

// Mask, softmax statistics and weighted sums of the flagged queries from their score rows.  A row is cut into OVF_CHUNKS key
// chunks: block (chunk c, lane y of 32) takes the flagged queries y, y + 32, ..; its four waves split the chunk (a row with
// hundreds of passing keys is a latency chain of value-patch gathers; one block per query left it at 290 us).
// Per (query, chunk): partial row (784), m = largest logit of the chunk's passing keys (-1: none), their number, z = sum e^(l - m).
// The gathers stop once the call's running edge count (a device word every block adds its counts to BEFORE gathering) passes
// `edge_limit`: the sum of all counts then exceeds it too, i.e. the verdict formed from the same counts (ovf_combine_kernel's
// statistics block; the host reads the same word) sends the call to the dense formulation and these rows are never used -- a
// value patch per edge costs ~1 ns here, the dense formulation 9 ps per PAIR.
__global__ __launch_bounds__(256) void ovf_attend_kernel(OvfArgs a) {
    __shared__ RowBlockShared sh;
    __shared__ int shc[4]; __shared__ int sh_over;
    const int nf = ovf_rows_served(a);
    if (nf == 0) return;
    // (a call that waits for its verdict knows the flagged rows' edges before this launch: beyond the limit the host, looking at
    // the same word, runs the dense formulation instead)
    if (a.edges_run == nullptr && a.flagged_edges != nullptr && *a.flagged_edges > a.edge_limit) return;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const RowCols cols = row_cols(lane);
    unsigned long long* edges_run = reinterpret_cast<unsigned long long*>(a.edges_run);
    const int chunk = blockIdx.x;
    int j0c, j1c; row_chunk_range(a.g.N, chunk, j0c, j1c);
    int j0, j1; row_wave_range(j0c, j1c, w, j0, j1);
    bool gather = true;
    for (int slot = blockIdx.y; slot < nf; slot += gridDim.y) {
        const size_t ql = (size_t)a.list[slot];
        const int b = (int)(ql / a.g.L);
        const float* row = a.scores + ((size_t)b * a.cap + slot) * a.ldn;
        const float mtq = a.mt[ql], bsq = a.bs[ql];
        const float4* vmb = reinterpret_cast<const float4*>(a.b2p + (size_t)b * a.g.Hp * a.g.Wp * CH);
        // the block's count for this (query, chunk) first, charged to the call's edge budget
        int cnt = 0;
        for (int c0 = j0; c0 < j1; c0 += 64) {
            const int j = c0 + lane; bool pass = false;
            if (j < j1) (void)ovf_logit(row[j], mtq, bsq, pass);
            cnt += __popcll(__ballot(pass));
        }
        if (lane == 0) shc[w] = cnt;
        if (tid == 0) sh_over = 0;
        __syncthreads();
        const int cnt_blk = shc[0] + shc[1] + shc[2] + shc[3];
        if (tid == 0 && cnt_blk > 0 && edges_run != nullptr &&
            (long long)(atomicAdd(edges_run, (unsigned long long)cnt_blk) + (unsigned long long)cnt_blk) > a.edge_limit) sh_over = 1;
        __syncthreads();
        if (sh_over) gather = false;
        RowAcc o; row_acc_clear(o);
        if (cnt_blk > 0)                                                         // block-uniform
            row_wave_walk(a.g, vmb, cols, lane, j0, j1, gather,
                          [&](int j, bool in, float& l) { bool pass = false; if (in) l = ovf_logit(row[j], mtq, bsq, pass); return pass; }, o);
        row_block_store(sh, cols, o, cnt_blk, a.part + ((size_t)slot * ROW_CHUNKS + chunk) * ROW_PART_FLOATS);
    }
}

// blocks 0 .. gridDim.x - 2: the flagged rows, combined from their chunks (row = sum of the chunks' partial rows, scaled, in chunk
// order, / Z); overwrites the aggregated row.  Last block: the call's statistics -- total edges and largest degree (the lists'
// counts, a flagged row counted with its true degree), the flagged rows' edges, and the verdict of the calls that do not wait.
__global__ __launch_bounds__(256) void ovf_combine_kernel(OvfArgs a, size_t n_rows, int64_t* __restrict__ stats, int32_t* veto, int32_t tag,
                                                          int what /* 1: the rows, 2: the statistics block, 3: both */) {
    __shared__ RowReduceShared sh;
    __shared__ long long sh_l[4][3];
    const int nf = ovf_rows_served(a);
    const int tid = threadIdx.x;
    const int C4 = P / 4;
    if ((what & 2) && blockIdx.x == gridDim.x - 1) {
        long long sum = 0, fl = 0; int mx = 0;
        for (size_t r = tid; r < n_rows; r += 256) { const int d = max(a.nb_cnt[r], 0); sum += d; mx = max(mx, d); }
        for (int slot = tid; slot < nf; slot += 256) {                           // a flagged row counts with its true degree
            int deg = 0;
            for (int c = 0; c < ROW_CHUNKS; ++c)
                deg += reinterpret_cast<const int*>(a.part + ((size_t)slot * ROW_CHUNKS + c) * ROW_PART_FLOATS + P)[1];
            sum += deg; mx = max(mx, deg); fl += deg;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { sum += __shfl_xor(sum, o); mx = max(mx, __shfl_xor(mx, o)); fl += __shfl_xor(fl, o); }
        if ((tid & 63) == 0) { sh_l[tid >> 6][0] = sum; sh_l[tid >> 6][1] = mx; sh_l[tid >> 6][2] = fl; }
        __syncthreads();
        if (tid == 0) {
            long long t = 0, f = 0; long long m = 0;
            for (int w = 0; w < 4; ++w) { t += sh_l[w][0]; m = max(m, sh_l[w][1]); f += sh_l[w][2]; }
            stats[0] = t; stats[1] = m;
            if (a.flagged_edges != nullptr) *a.flagged_edges = f;       // what redoing the flagged rows one by one gathers
            if (veto != nullptr) {
                // the host's verdict (capi.hip, adaptive mode) formed here for the calls that do not wait for it: were the overflowed
                // queries all redone in-stream?  If not, the fold NaN-fills this call's output and dagl_ce_range_check reports it.
                const long long ov = stats[2];
                const bool heavy = ov > 0 && ov <= a.cap && f > a.edge_limit;
                const bool mostly = ov * 2 > (long long)n_rows || heavy;
                const bool served = ov == 0 || (!mostly && ov <= a.cap);
                if (!served) *veto = tag;
            }
        }
        return;
    }
    if (!(what & 1)) return;
    const int row_blocks = (what & 2) ? gridDim.x - 1 : gridDim.x;
    for (int slot = blockIdx.x; slot < nf; slot += row_blocks) {
        const size_t ql = (size_t)a.list[slot];
        const float* part_row = a.part + (size_t)slot * ROW_CHUNKS * ROW_PART_FLOATS;
        const RowSum row = row_reduce(part_row, a.g.N, sh);
        if (tid < C4) reinterpret_cast<float4*>(a.agg)[ql * C4 + tid] = row_combine(part_row, row, sh);
        if (tid == 0) {
            if (a.dbg_deg) a.dbg_deg[ql] = row.deg;
            if (a.dbg_rowsum) a.dbg_rowsum[ql] = (float)(row.zs / row.Z);
        }
    }
}

// flagged queries a call can redo one by one: 512 MiB of score rows (every image's keys against all listed rows), at most
// 2048 -- at 256^2 that is half the queries, i.e. everything short of "most queries overflow" (where the whole call goes
// dense).  The per-row cost (a row of the product, ~a thousand value patches gathered) stays below the dense formulation's
// fixed cost up to there; the fp32 scan + CSR lists, which used to take over beyond 256 rows, was slower than either.
int overflow_cap(int N, int B) {
    const long long ldn = (N + 31) / 32 * 32;
    long long c = ((long long)512 << 20) / (ldn * 4 * (B > 0 ? B : 1));
    if (c > 2048) c = 2048;
    return (int)(c < 4 ? 4 : c);
}

// The redo of the flagged queries, queued right behind the refine kernels: (matrix-core scores when there are many,) VALU scores
// when there are few, rows, combine + statistics.  Every kernel reads the number of flagged queries from device memory and exits
// at once when it is zero.  Two orders:
//   ag given (a call that does not wait for its verdict): scores in one launch with the gather + weighted sum over everybody's
//     lists, attend, combine + statistics -- four launches;
//   ag null (a call that waits): scores, counts, statistics -- then the caller queues its read-back, the list gather, and
//     launch_overflow_apply (attend, combine): everything that gathers runs under the host's round trip.
static int ovf_scores(hipStream_t s, const OvfArgs& a, const AggArgs* ag) {
    // scores of the flagged queries against all keys when they are many: one product [flagged, 196] x [196, N] on the fp32 matrix
    // cores (chains of 48 products, partial sums added in fp32), batched over the images: every image's keys against ALL flagged
    // rows (a flagged query only reads the row block of its own image); the query rows were compacted by the refine kernels
    Gemm32 g;
    g.M = a.cap; g.N = a.g.N; g.K = D; g.batch = a.B;
    g.A = a.qrows; g.lda = DS; g.sA = 0; g.a_kc = 1;                          // the same flagged rows for every image
    g.B = a.x; g.ldb = DS; g.sB = (long long)a.rows_x * DS; g.b_kc = 1;
    g.C = a.scores; g.ldc = a.ldn; g.sC = (long long)a.cap * a.ldn;
    g.alpha = 1.f; g.beta = 0.f; g.bias = nullptr; g.relu = 0; g.chunk_tiles = 3;
    g.m_limit = a.count;                                                       // few rows (VALU form) or more than the list holds: none
    g.m_limit_floor = a.cap < OVF_SMALL ? a.cap : OVF_SMALL; g.m_limit_ceil = a.cap;
    const int rc = launch_gemm32(s, g);
    if (rc) return rc;
    const int score_blocks = (a.g.N + 63) / 64;
    hipLaunchKernelGGL(ovf_scores_aggregate_kernel, dim3((unsigned)(score_blocks * a.B + (ag ? a.g.L * a.B : 0))), dim3(256), 0, s, a,
                       ag ? *ag : AggArgs(), score_blocks);
    DAGL_LAUNCH_CHECK("ovf_scores_aggregate_kernel");
    return DAGL_OK;
}

int launch_overflow_rows(hipStream_t s, const OvfArgs& a, const AggArgs* ag, size_t n_rows, int64_t* stats, int32_t* veto, int32_t tag) {
    if (a.cap <= 0) return DAGL_ERR_INVALID;
    int rc = ovf_scores(s, a, ag);
    if (rc) return rc;
    const int gy = a.cap < 32 ? a.cap : 32;                               // flagged queries are strided over grid.y
    const int rb = a.cap < OVF_GRID_B ? a.cap : OVF_GRID_B;
    if (ag != nullptr) {
        hipLaunchKernelGGL(ovf_attend_kernel, dim3(ROW_CHUNKS, gy), dim3(256), 0, s, a);
        DAGL_LAUNCH_CHECK("ovf_attend_kernel");
        hipLaunchKernelGGL(ovf_combine_kernel, dim3(rb + 1), dim3(256), 0, s, a, n_rows, stats, veto, tag, 3);
        DAGL_LAUNCH_CHECK("ovf_combine_kernel");
    } else {
        hipLaunchKernelGGL(ovf_count_kernel, dim3(ROW_CHUNKS, gy), dim3(256), 0, s, a);
        DAGL_LAUNCH_CHECK("ovf_count_kernel");
        hipLaunchKernelGGL(ovf_combine_kernel, dim3(1), dim3(256), 0, s, a, n_rows, stats, veto, tag, 2);
        DAGL_LAUNCH_CHECK("ovf_combine_kernel");
    }
    return DAGL_OK;
}

// (the second half of the order for calls that wait: behind the read-back and the list gather)
int launch_overflow_apply(hipStream_t s, const OvfArgs& a) {
    if (a.cap <= 0) return DAGL_ERR_INVALID;
    const int gy = a.cap < 32 ? a.cap : 32;
    hipLaunchKernelGGL(ovf_attend_kernel, dim3(ROW_CHUNKS, gy), dim3(256), 0, s, a);
    DAGL_LAUNCH_CHECK("ovf_attend_kernel");
    hipLaunchKernelGGL(ovf_combine_kernel, dim3(a.cap < OVF_GRID_B ? a.cap : OVF_GRID_B), dim3(256), 0, s, a, (size_t)0, nullptr, nullptr, 0, 1);
    DAGL_LAUNCH_CHECK("ovf_combine_kernel");
    return DAGL_OK;
}

}  // namespace dagl
