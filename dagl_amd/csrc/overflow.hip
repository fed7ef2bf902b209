// Long-tailed adaptive neighbourhoods behind the screen: the few queries whose neighbourhood does not fit the fixed-width
// list (more than DAGL_LIST_CAP passing keys, or more candidates than a screen segment has slots) are redone ONE BY ONE
// in the dense formulation of the reference (DN_Gray/model/dagl.py:250-264) instead of sending the whole call to the
// fp32 scan: an adaptive mask with a mean degree of ~8 has maxima of several hundred (measured at 256^2: mean 7.7,
// maximum 890, 10 of 4096 queries beyond 256), and those 10 rows are 10 x N scores, not L x N.
//   A  gather + gemm32     scores of every flagged query against ALL keys as ONE matrix product [flagged, 196] x [196, N] on
//                          the fp32 matrix cores (every key row is read once for all flagged queries, not once per query; the
//                          row count lives in device memory: blocks past it exit)
//   B  ovf_attend_kernel   one block per flagged query: mask, softmax over all N keys (masked keys count e^0), weighted sum
//                          of the value patches straight from the value map; overwrites the query's aggregated row
// All kernels read the number of flagged queries from device memory and exit at once when it is zero.
#include "dagl_common.h"

namespace dagl {

constexpr int OVF_KEYS = 256;                    // keys per block of kernel A

__device__ __forceinline__ float ovf_logit(float s, float mtq, float bsq, bool& pass) {
    const float m = (s - mtq) + bsq;                                      // dagl.py:256, same expression order
    pass = m > 0.f;
    return pass ? __fmul_rn(__fmul_rn(s, m), SOFTMAX_SCALE) : 0.f;
}

constexpr int OVF_GRID_B = 256;   // fixed small grid (an empty call costs one wave of exits): flagged queries are strided

// feature rows of the flagged queries -> compact [cap, DS] matrix (rows past the count: untouched, never used)
__global__ __launch_bounds__(256) void ovf_gather_kernel(OvfArgs a) {
    // more flagged queries than the list holds: the host is about to send the whole call to the dense formulation (or the
    // fp32 scan) -- nothing to do here (redoing 256 rows of a DENSE mask one by one costs 17 ms at 256^2)
    int nf = *a.count; if (nf > a.cap) nf = 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) *a.eff = nf;                       // what the product and the attend kernel use
    const int slot = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (slot >= nf) return;
    const size_t ql = (size_t)a.list[slot];
    const int b = (int)(ql / a.g.L);
    const float* qrow = a.wq + ((size_t)b * a.rows_q + (ql - (size_t)b * a.g.L)) * DS;
    for (int c = lane; c < DS; c += 64) a.qrows[(size_t)slot * DS + c] = qrow[c];
}

__global__ __launch_bounds__(256) void ovf_attend_kernel(OvfArgs a) {
    __shared__ double shd[4];
    __shared__ float shf[4];
    __shared__ int lk[OVF_KEYS];
    __shared__ float lw[OVF_KEYS];
    __shared__ int wcnt[4];
    const int nf = *a.eff;
    for (int slot = blockIdx.x; slot < nf; slot += gridDim.x) {
    const size_t ql = (size_t)a.list[slot];
    const int b = (int)(ql / a.g.L);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int N = a.g.N;
    const float* row = a.scores + ((size_t)b * a.cap + slot) * a.ldn;
    const float mtq = a.mt[ql], bsq = a.bs[ql];

    // 1. degree, largest logit
    float mx = -1.f; int cnt = 0;
    for (int j = tid; j < N; j += 256) {
        bool pass; const float l = ovf_logit(row[j], mtq, bsq, pass);
        if (pass) { mx = fmaxf(mx, l); ++cnt; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { mx = fmaxf(mx, __shfl_xor(mx, o)); cnt += __shfl_xor(cnt, o); }
    if (lane == 0) { shf[w] = mx; wcnt[w] = cnt; }
    __syncthreads();
    const int deg = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
    double M = (double)fmaxf(fmaxf(shf[0], shf[1]), fmaxf(shf[2], shf[3]));
    if (deg < N) M = fmax(M, 0.0);
    __syncthreads();
    // 2. denominator over ALL keys (masked keys: e^(0 - M) each)
    double z = 0.0;
    for (int j = tid; j < N; j += 256) {
        bool pass; const float l = ovf_logit(row[j], mtq, bsq, pass);
        if (pass) z += (double)expf((float)((double)l - M));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) z += __shfl_xor(z, o);
    if (lane == 0) shd[w] = z;
    __syncthreads();
    const double Z = ((shd[0] + shd[1]) + (shd[2] + shd[3])) + (double)(N - deg) * exp(-M);
    __syncthreads();

    // 3. weighted sum of the value patches, neighbours in ascending key order, 256 keys per round
    const int C4 = P / 4;                                                 // thread r < 196 owns float4 column r = kh*28 + (kw*4 + c4)
    const int r = tid < C4 ? tid : 0;
    const int kh = r / 28, rem = r % 28;
    const float4* vm = reinterpret_cast<const float4*>(a.b2p + (size_t)b * a.g.Hp * a.g.Wp * CH) + rem;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    double rs = 0.0;
    for (int c0 = 0; c0 < N; c0 += OVF_KEYS) {
        const int j = c0 + tid;
        bool pass = false; float wgt = 0.f;
        if (j < N) {
            const float l = ovf_logit(row[j], mtq, bsq, pass);
            if (pass) wgt = (float)((double)expf((float)((double)l - M)) / Z);
        }
        const unsigned long long bal = __ballot(pass);
        if (lane == 0) wcnt[w] = (int)__popcll(bal);
        __syncthreads();
        int base = 0;
        for (int u = 0; u < w; ++u) base += wcnt[u];
        const int total = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
        if (pass) { const int pos = base + (int)__popcll(bal & ((1ull << lane) - 1ull)); lk[pos] = j; lw[pos] = wgt; }
        __syncthreads();
        if (tid < C4) {
            for (int e = 0; e < total; ++e) {
                const int key = lk[e]; const float wv = lw[e];
                const int jy = key / a.g.W, jx = key - jy * a.g.W;
                const float4 v = vm[((size_t)(jy + kh) * a.g.Wp + jx) * (CH / 4)];
                acc.x = fmaf(wv, v.x, acc.x); acc.y = fmaf(wv, v.y, acc.y); acc.z = fmaf(wv, v.z, acc.z); acc.w = fmaf(wv, v.w, acc.w);
            }
        }
        if (tid == 0) for (int e = 0; e < total; ++e) rs += (double)lw[e];
        __syncthreads();
    }
    if (tid < C4) reinterpret_cast<float4*>(a.agg)[ql * C4 + tid] = acc;
    if (tid == 0) {
        a.nb_cnt[ql] = deg;                                               // true degree (the list itself stays clipped)
        if (a.dbg_deg) a.dbg_deg[ql] = deg;
        if (a.dbg_rowsum) a.dbg_rowsum[ql] = (float)rs;
    }
    __syncthreads();
    }
}

// flagged queries a call can redo one by one: 64 MiB of score rows, at most 256 (beyond that the whole call goes dense)
int overflow_cap(int N) {
    const long long ldn = (N + 31) / 32 * 32;
    long long c = ((long long)64 << 20) / (ldn * 4);
    if (c > 256) c = 256;
    return (int)(c < 4 ? 4 : c);
}

int launch_overflow_rows(hipStream_t s, const OvfArgs& a) {
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
        g.alpha = 1.f; g.beta = 0.f; g.bias = nullptr; g.relu = 0; g.chunk_tiles = 3; g.m_limit = a.eff;
        const int rc = launch_gemm32(s, g);
        if (rc) return rc;
    }
    hipLaunchKernelGGL(ovf_attend_kernel, dim3(a.cap < OVF_GRID_B ? a.cap : OVF_GRID_B), dim3(256), 0, s, a);
    DAGL_LAUNCH_CHECK("ovf_attend_kernel");
    return DAGL_OK;
}

}  // namespace dagl
