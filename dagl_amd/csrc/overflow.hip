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

namespace dagl {

__device__ __forceinline__ float ovf_logit(float s, float mtq, float bsq, bool& pass) {
    const float m = (s - mtq) + bsq;                                      // dagl.py:256, same expression order
    pass = m > 0.f;
    return pass ? __fmul_rn(__fmul_rn(s, m), SOFTMAX_SCALE) : 0.f;
}

constexpr int OVF_SMALL = 16;     // up to this many flagged queries the scores come from a VALU kernel instead of the product
constexpr int OVF_GRID_B = 256;   // fixed small grid (an empty call costs one wave of exits): flagged queries are strided

// feature rows of the flagged queries -> compact [cap, DS] matrix (rows past the count: untouched, never used)
__global__ __launch_bounds__(256) void ovf_gather_kernel(OvfArgs a) {
    // more flagged queries than the list holds: the host is about to send the whole call to the dense formulation (or the
    // fp32 scan) -- nothing to do here (redoing 256 rows of a DENSE mask one by one costs 17 ms at 256^2)
    int nf = *a.count; if (nf > a.cap) nf = 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        a.eff[0] = nf;                                                          // what the attend kernels use
        a.eff[1] = (nf > OVF_SMALL) ? nf : 0;                                   // rows of the matrix-core product (few rows: VALU kernel)
    }
    const int slot = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (slot >= nf) return;
    const size_t ql = (size_t)a.list[slot];
    const int b = (int)(ql / a.g.L);
    const float* qrow = a.wq + ((size_t)b * a.rows_q + (ql - (size_t)b * a.g.L)) * DS;
    for (int c = lane; c < DS; c += 64) a.qrows[(size_t)slot * DS + c] = qrow[c];
}

// A' few flagged queries (<= OVF_SMALL; 10 at 256^2 mean degree 8): a 128-row matrix-core tile would be 90 % padding (45 us);
// instead four lanes share a key, each sums a quarter of the 49 float4 products per flagged row (query rows broadcast from
// LDS), two shuffle steps finish the dot product.  Every key row is still read once for all flagged queries.
__global__ __launch_bounds__(256) void ovf_scores_small_kernel(OvfArgs a) {
    __shared__ float4 sq[OVF_SMALL][DS / 4];                                    // 12.5 KiB
    const int nf = a.eff[0];
    if (nf == 0 || nf > OVF_SMALL) return;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    for (int e = tid; e < nf * (DS / 4); e += 256) sq[e / (DS / 4)][e % (DS / 4)] = reinterpret_cast<const float4*>(a.qrows)[e];
    __syncthreads();
    const int b = blockIdx.y;
    const int k = lane >> 2, qd = lane & 3;
    const long long key = ((long long)blockIdx.x * 4 + w) * 16 + k;
    const bool ok = key < a.g.N;
    const float4* xr = reinterpret_cast<const float4*>(a.x + ((size_t)b * a.rows_x + (ok ? key : 0)) * DS);
    float acc[OVF_SMALL];
#pragma unroll
    for (int r = 0; r < OVF_SMALL; ++r) acc[r] = 0.f;
    // the lane's 13 pieces of the key row are requested together (the row is read once, from HBM: one round trip, not 13)
    constexpr int NC = (D / 4 + 3) / 4;                                          // 13
    float4 xs[NC];
#pragma unroll
    for (int u = 0; u < NC; ++u) {
        const int c = qd + 4 * u;
        const float4 raw = xr[c < D / 4 ? c : D / 4 - 1];                        // clamped, zeroed below (no predicated load)
        xs[u] = (c < D / 4) ? raw : make_float4(0.f, 0.f, 0.f, 0.f);
    }
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

// A flagged query's score row is cut into OVF_CHUNKS key chunks, one block each: a row with hundreds of passing keys is a
// latency chain of value-patch gathers, and one block per query (10 blocks at 256^2) left it at 290 us.
//   B1 ovf_stats_kernel    per (query, chunk): largest logit and number of passing keys
//   B2 ovf_attend_kernel   per (query, chunk): with the row's M and degree from B1, z = sum of e^(l - M) over the chunk's
//                          passing keys and the UNNORMALISED weighted sum of their value patches
//   B3 ovf_combine_kernel  per query: Z = sum of z + (N - degree) e^(-M)  (masked keys count e^0 each, dagl.py:259-261),
//                          row = sum of the chunks' partial rows / Z in chunk order; overwrites the aggregated row
__device__ __forceinline__ void ovf_chunk_range(int N, int chunk, int& j0, int& j1) {
    const int per = ((N + OVF_CHUNKS - 1) / OVF_CHUNKS + 255) / 256 * 256;
    j0 = chunk * per; j1 = j0 + per;
    if (j0 > N) j0 = N;
    if (j1 > N) j1 = N;
}

__global__ __launch_bounds__(256) void ovf_stats_kernel(OvfArgs a) {
    __shared__ float shf[4];
    __shared__ int wcnt[4];
    const int nf = *a.eff;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    int j0, j1; ovf_chunk_range(a.g.N, blockIdx.x, j0, j1);
    for (int slot = blockIdx.y; slot < nf; slot += gridDim.y) {
        const size_t ql = (size_t)a.list[slot];
        const int b = (int)(ql / a.g.L);
        const float* row = a.scores + ((size_t)b * a.cap + slot) * a.ldn;
        const float mtq = a.mt[ql], bsq = a.bs[ql];
        float mx = -1.f; int cnt = 0;
        for (int j = j0 + tid; j < j1; j += 256) {
            bool pass; const float l = ovf_logit(row[j], mtq, bsq, pass);
            if (pass) { mx = fmaxf(mx, l); ++cnt; }
        }
        mx = wave_max_f32(mx); cnt = wave_sum_i32(cnt);
        if (lane == 0) { shf[w] = mx; wcnt[w] = cnt; }
        __syncthreads();
        if (tid == 0) {
            float* pr = a.part + ((size_t)slot * OVF_CHUNKS + blockIdx.x) * OVF_PART_FLOATS + P;
            pr[0] = fmaxf(fmaxf(shf[0], shf[1]), fmaxf(shf[2], shf[3]));
            reinterpret_cast<int*>(pr)[1] = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
        }
        __syncthreads();
    }
}

// the row's largest logit (0 joins in when a key is masked) and degree, from the chunks' partials (fixed order)
__device__ __forceinline__ void ovf_row_stats(const OvfArgs& a, int slot, double& M, int& deg) {
    float mx = -1.f; deg = 0;
    for (int c = 0; c < OVF_CHUNKS; ++c) {
        const float* pr = a.part + ((size_t)slot * OVF_CHUNKS + c) * OVF_PART_FLOATS + P;
        mx = fmaxf(mx, pr[0]); deg += reinterpret_cast<const int*>(pr)[1];
    }
    M = (double)mx;
    if (deg < a.g.N) M = fmax(M, 0.0);
}

__global__ __launch_bounds__(256) void ovf_attend_kernel(OvfArgs a) {
    // (a value patch per edge: rows that are dense cost ~1 ns per edge here, the dense formulation 9 ps per PAIR -- past the
    // limit the host, looking at the same word, runs that instead)
    if (a.flagged_edges != nullptr && *a.flagged_edges > a.edge_limit) return;
    __shared__ double shd[4];
    __shared__ float4 part[4][P / 4];                                       // the four waves' partial rows (12.25 KiB)
    const int nf = *a.eff;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int C4 = P / 4;
    int j0c, j1c; ovf_chunk_range(a.g.N, blockIdx.x, j0c, j1c);
    for (int slot = blockIdx.y; slot < nf; slot += gridDim.y) {
    const size_t ql = (size_t)a.list[slot];
    const int b = (int)(ql / a.g.L);
    const float* row = a.scores + ((size_t)b * a.cap + slot) * a.ldn;
    const float mtq = a.mt[ql], bsq = a.bs[ql];
    double M; int deg; ovf_row_stats(a, slot, M, deg);

    // No block barrier inside the sum: every wave walks its own quarter of the chunk in pieces of 64 keys (ballot of the
    // passing keys, ascending), lane l owns the float4 columns l + 64 u of the 784-float row; neighbours are taken two at a
    // time so that their loads are in flight together.  The four partial rows are added in wave order at the end.
    int kh[4], rem[4]; bool cv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int r = lane + 64 * u;
        cv[u] = r < C4;
        const int rc = cv[u] ? r : 0;
        kh[u] = rc / 28; rem[u] = rc % 28;
    }
    const float4* vmb = reinterpret_cast<const float4*>(a.b2p + (size_t)b * a.g.Hp * a.g.Wp * CH);
    float4 acc[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) acc[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    double z = 0.0;
    const int per_wave = ((j1c - j0c + 3) / 4 + 63) / 64 * 64;
    const int j0 = j0c + w * per_wave;
    const int j1 = (j0 + per_wave < j1c) ? j0 + per_wave : j1c;
    auto take = [&](int key, float wv) {
        const int jy = key / a.g.W, jx = key - jy * a.g.W;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (!cv[u]) continue;
            const float4 v = vmb[((size_t)(jy + kh[u]) * a.g.Wp + jx) * (CH / 4) + rem[u]];
            acc[u].x = fmaf(wv, v.x, acc[u].x); acc[u].y = fmaf(wv, v.y, acc[u].y);
            acc[u].z = fmaf(wv, v.z, acc[u].z); acc[u].w = fmaf(wv, v.w, acc[u].w);
        }
    };
    for (int c0 = j0; c0 < j1; c0 += 64) {
        const int j = c0 + lane;
        bool pass = false; float wgt = 0.f;
        if (j < j1) {
            const float l = ovf_logit(row[j], mtq, bsq, pass);
            if (pass) wgt = expf((float)((double)l - M));
        }
        unsigned long long bal = __ballot(pass);
        while (bal) {
            const int b0 = __ffsll((long long)bal) - 1; bal &= bal - 1;
            const float w0 = __shfl(wgt, b0);
            if (bal) {
                const int b1 = __ffsll((long long)bal) - 1; bal &= bal - 1;
                const float w1 = __shfl(wgt, b1);
                take(c0 + b0, w0); take(c0 + b1, w1);
                z += (double)w0; z += (double)w1;
            } else {
                take(c0 + b0, w0);
                z += (double)w0;
            }
        }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) if (cv[u]) part[w][lane + 64 * u] = acc[u];
    if (lane == 0) shd[w] = z;
    __syncthreads();
    float* pr = a.part + ((size_t)slot * OVF_CHUNKS + blockIdx.x) * OVF_PART_FLOATS;
    if (tid < C4) {
        float4 t = part[0][tid];
#pragma unroll
        for (int ww = 1; ww < 4; ++ww) { const float4 v = part[ww][tid]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
        reinterpret_cast<float4*>(pr)[tid] = t;
    }
    if (tid == 0) *reinterpret_cast<double*>(pr + P + 2) = (shd[0] + shd[1]) + (shd[2] + shd[3]);
    __syncthreads();
    }
}

__global__ __launch_bounds__(256) void ovf_combine_kernel(OvfArgs a) {
    if (a.flagged_edges != nullptr && *a.flagged_edges > a.edge_limit) return;
    const int nf = *a.eff;
    const int tid = threadIdx.x;
    const int C4 = P / 4;
    for (int slot = blockIdx.x; slot < nf; slot += gridDim.x) {
        const size_t ql = (size_t)a.list[slot];
        double M; int deg; ovf_row_stats(a, slot, M, deg);
        double zs = 0.0;
        for (int c = 0; c < OVF_CHUNKS; ++c)
            zs += *reinterpret_cast<const double*>(a.part + ((size_t)slot * OVF_CHUNKS + c) * OVF_PART_FLOATS + P + 2);
        const double Z = zs + (double)(a.g.N - deg) * exp(-M);
        const float inv = (float)(1.0 / Z);
        if (tid < C4) {
            float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int c = 0; c < OVF_CHUNKS; ++c) {
                const float4 v = reinterpret_cast<const float4*>(a.part + ((size_t)slot * OVF_CHUNKS + c) * OVF_PART_FLOATS)[tid];
                t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
            }
            t.x *= inv; t.y *= inv; t.z *= inv; t.w *= inv;
            reinterpret_cast<float4*>(a.agg)[ql * C4 + tid] = t;
        }
        if (tid == 0) {
            a.nb_cnt[ql] = deg;                                               // true degree (the list itself stays clipped)
            if (a.dbg_deg) a.dbg_deg[ql] = deg;
            if (a.dbg_rowsum) a.dbg_rowsum[ql] = (float)(zs / Z);
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

// total edges and largest degree of a call whose flagged rows have their chunk statistics but not yet their rows:
// the lists' counts, with every flagged row's clipped count replaced by its true degree (what ovf_combine_kernel will
// store in nb_cnt afterwards)
__global__ __launch_bounds__(1024) void degree_stats_flagged_kernel(size_t n_rows, const int32_t* __restrict__ nb_cnt,
                                                                    int64_t* __restrict__ stats, OvfArgs a, int32_t* veto, int32_t tag) {
    __shared__ long long ssum[16];
    __shared__ int smax[16];
    __shared__ long long sfl[16];
    long long sum = 0, fl = 0; int mx = 0;
    for (size_t r = threadIdx.x; r < n_rows; r += blockDim.x) { const int d = nb_cnt[r]; sum += d; mx = max(mx, d); }
    const int nf = *a.eff;
    for (int slot = threadIdx.x; slot < nf; slot += blockDim.x) {
        double M; int deg; ovf_row_stats(a, slot, M, deg);
        sum += deg - nb_cnt[a.list[slot]]; mx = max(mx, deg); fl += deg;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { sum += __shfl_xor(sum, o); mx = max(mx, __shfl_xor(mx, o)); fl += __shfl_xor(fl, o); }
    if ((threadIdx.x & 63) == 0) { ssum[threadIdx.x >> 6] = sum; smax[threadIdx.x >> 6] = mx; sfl[threadIdx.x >> 6] = fl; }
    __syncthreads();
    if (threadIdx.x == 0) {
        long long t = 0, f = 0; int m = 0;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) { t += ssum[w]; m = max(m, smax[w]); f += sfl[w]; }
        stats[0] = t; stats[1] = m;
        if (a.flagged_edges != nullptr) *a.flagged_edges = f;       // what redoing the flagged rows one by one would gather
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
}

int launch_degree_stats_flagged(hipStream_t s, size_t n_rows, const int32_t* nb_cnt, int64_t* stats, const OvfArgs& a, int32_t* veto,
                                int32_t tag) {
    hipLaunchKernelGGL(degree_stats_flagged_kernel, dim3(1), dim3(1024), 0, s, n_rows, nb_cnt, stats, a, veto, tag);
    DAGL_LAUNCH_CHECK("degree_stats_flagged_kernel");
    return DAGL_OK;
}

int launch_overflow_scores(hipStream_t s, const OvfArgs& a) {
    if (a.cap <= 0) return DAGL_OK;
    hipLaunchKernelGGL(ovf_gather_kernel, dim3((a.cap + 3) / 4), dim3(256), 0, s, a);
    DAGL_LAUNCH_CHECK("ovf_gather_kernel");
    // scores of the flagged queries against all keys: one product [flagged, 196] x [196, N] on the fp32 matrix cores
    // (chains of 48 products, partial sums added in fp32), batched over the images: every image's keys against ALL flagged
    // rows (a flagged query only reads the row block of its own image; flagged queries are few)
    {
        Gemm32 g;
        g.M = a.cap; g.N = a.g.N; g.K = D; g.batch = a.B;
        g.A = a.qrows; g.lda = DS; g.sA = 0; g.a_kc = 1;                          // the same flagged rows for every image
        g.B = a.x; g.ldb = DS; g.sB = (long long)a.rows_x * DS; g.b_kc = 1;
        g.C = a.scores; g.ldc = a.ldn; g.sC = (long long)a.cap * a.ldn;
        g.alpha = 1.f; g.beta = 0.f; g.bias = nullptr; g.relu = 0; g.chunk_tiles = 3; g.m_limit = a.eff + 1;
        const int rc = launch_gemm32(s, g);
        if (rc) return rc;
    }
    hipLaunchKernelGGL(ovf_scores_small_kernel, dim3((a.g.N + 63) / 64, a.B), dim3(256), 0, s, a);
    DAGL_LAUNCH_CHECK("ovf_scores_small_kernel");
    const int gy = a.cap < 32 ? a.cap : 32;                               // flagged queries are strided over grid.y
    hipLaunchKernelGGL(ovf_stats_kernel, dim3(OVF_CHUNKS, gy), dim3(256), 0, s, a);
    DAGL_LAUNCH_CHECK("ovf_stats_kernel");
    return DAGL_OK;
}

int launch_overflow_apply(hipStream_t s, const OvfArgs& a) {
    if (a.cap <= 0) return DAGL_OK;
    const int gy = a.cap < 32 ? a.cap : 32;
    hipLaunchKernelGGL(ovf_attend_kernel, dim3(OVF_CHUNKS, gy), dim3(256), 0, s, a);
    DAGL_LAUNCH_CHECK("ovf_attend_kernel");
    hipLaunchKernelGGL(ovf_combine_kernel, dim3(a.cap < OVF_GRID_B ? a.cap : OVF_GRID_B), dim3(256), 0, s, a);
    DAGL_LAUNCH_CHECK("ovf_combine_kernel");
    return DAGL_OK;
}

}  // namespace dagl
