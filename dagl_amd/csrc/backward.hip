// Backward of the graph core (everything of CE.forward after the projections, DN_Gray/model/dagl.py:250-272) for
// sparse neighbourhoods -- fixed-width neighbour lists as the forward keeps them (top-k modes, adaptive masks with at
// most DAGL_FAST_CAP neighbours).  The reference differentiates the dense formulation with autograd; the same
// gradients restricted to the neighbour lists are:
//   out = fold(agg)/cnt                      d agg[l,(kh,kw,c)] = d out[c, 4r-3+kh, 4c-3+kw] / cnt          (unfold)
//   agg_l = sum_t A_t V_{j_t}                d A_t = <d agg_l, V_{j_t}>,   d V_{j_t} += A_t d agg_l
//   A = softmax(l) * mask_b (not renormalised, non-neighbours have l = 0 and no gradient)
//                                            d l_t = A_t (d A_t - sum_u A_u d A_u)
//   l = 10 S m,  m = (S - mu thr) + bias  (adaptive)      d S_t = 10 (m_t + S_t) d l_t,   d m_t = 10 S_t d l_t
//                m = 1                    (top-k)         d S_t = 10 d l_t
//   per query:  d bias = sum_t d m_t,  d thr = -mu sum_t d m_t,  d mu = -thr sum_t d m_t
//   mu_l = Wq_l . mean_j X_j                 d Wq_l += d mu_l Xbar,   d X_j += (sum_l d mu_l Wq_l) / N   for ALL keys j
//   S_lj = Wq_l . X_j                        d Wq_l += sum_t d S_t X_{j_t},   d X_{j_t} += d S_t Wq_l
// Scatter-adds into d X and d b2 use fp32 atomics (their order is not fixed: gradients are reproducible only up to
// fp32 reassociation, like the reference's cuDNN/cuBLAS backward).
#include "dagl_common.h"

namespace dagl {

// d agg[b,l,(kh,kw,c)] = d out[b,c,y,x] / cnt(y,x),  (y,x) = (4r-3+kh, 4c-3+kw); zero outside the image
__global__ __launch_bounds__(256) void unfold_dout_kernel(Grid g, const float* __restrict__ dout, float* __restrict__ dagg) {
    const int b = blockIdx.y;
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;          // (l, kh, kw) ; 16 channels per thread
    if (t >= (size_t)g.L * KS * KS) return;
    const int l = (int)(t / (KS * KS)), kk = (int)(t % (KS * KS));
    const int kh = kk / KS, kw = kk % KS;
    const int r = l / g.Lw, c = l % g.Lw;
    const int y = QS * r - 3 + kh, x = QS * c - 3 + kw;
    float v[CH];
    if (y >= 0 && y < g.H && x >= 0 && x < g.W) {
        int r0 = (y < 3) ? 0 : (y - 3 + QS - 1) / QS; int r1 = (y + 3) / QS; if (r1 > g.Lh - 1) r1 = g.Lh - 1;
        int c0 = (x < 3) ? 0 : (x - 3 + QS - 1) / QS; int c1 = (x + 3) / QS; if (c1 > g.Lw - 1) c1 = g.Lw - 1;
        const float inv = 1.0f / (float)((r1 - r0 + 1) * (c1 - c0 + 1));
        const float* d = dout + (size_t)b * CH * g.N + (size_t)y * g.W + x;
#pragma unroll
        for (int ch = 0; ch < CH; ++ch) v[ch] = d[(size_t)ch * g.N] * inv;
    } else {
#pragma unroll
        for (int ch = 0; ch < CH; ++ch) v[ch] = 0.f;
    }
    float4* o = reinterpret_cast<float4*>(dagg + (((size_t)b * g.L + l) * P + kk * CH));
#pragma unroll
    for (int q = 0; q < CH / 4; ++q) o[q] = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
}

// one wave per query: d A, softmax / logit backward, per-query threshold gradients, d V scatter
__global__ __launch_bounds__(256) void edge_backward_kernel(BwdArgs a) {
    const int lane = threadIdx.x & 63;
    const size_t ql = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ql >= (size_t)a.B * a.g.L) return;
    const int b = (int)(ql / a.g.L);
    const int n = a.nb_cnt[ql];
    const bool adaptive = (a.mode != DAGL_MODE_TOPK);
    const float4* dg = reinterpret_cast<const float4*>(a.dagg + ql * P);
    // this lane's float4 columns of the 784-float patch row: r = lane + 64 u < 196
    float4 d4[4]; int voff[4]; bool ok[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int r = lane + 64 * u;
        ok[u] = r < P / 4;
        d4[u] = ok[u] ? dg[r] : make_float4(0.f, 0.f, 0.f, 0.f);
        const int kh = r / 28, rem = r % 28;                          // r = kh*28 + (kw*4 + c4)
        voff[u] = (kh * a.g.Wp) * (CH / 4) + rem;                     // float4 offset inside a key's window
    }
    const float4* vm = reinterpret_cast<const float4*>(a.b2p + (size_t)b * a.g.Hp * a.g.Wp * CH);
    float* dvm = a.db2p + (size_t)b * a.g.Hp * a.g.Wp * CH;

    // pass 1: d A_t (lane t keeps neighbour t's value; width <= 64)
    float my_dA = 0.f;
    for (int t = 0; t < n; ++t) {
        const int j = a.nb_idx[ql * a.width + t];
        const int jy = j / a.g.W, jx = j - jy * a.g.W;
        const float4* vj = vm + ((size_t)jy * a.g.Wp + jx) * (CH / 4);
        float acc = 0.f;
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (ok[u]) {
                const float4 v = vj[voff[u]];
                acc += d4[u].x * v.x + d4[u].y * v.y + d4[u].z * v.z + d4[u].w * v.w;
            }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
        if (lane == t) my_dA = acc;
    }
    const bool valid = lane < n;
    const float A = valid ? a.nb_wgt[ql * a.width + lane] : 0.f;
    const float S = valid ? a.nb_s[ql * a.width + lane] : 0.f;
    float sdot = A * my_dA;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sdot += __shfl_xor(sdot, o);
    const float dl = A * (my_dA - sdot);
    float dS, dm = 0.f;
    if (adaptive) {
        const float m = (S - a.mu[ql] * a.thr[ql]) + a.bs[ql];
        dS = SOFTMAX_SCALE * (m + S) * dl;
        dm = SOFTMAX_SCALE * S * dl;
    } else {
        dS = SOFTMAX_SCALE * dl;
    }
    if (valid) a.dS[ql * a.width + lane] = dS;
    if (adaptive) {
        float sdm = valid ? dm : 0.f;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sdm += __shfl_xor(sdm, o);
        if (lane == 0) {
            a.dbias[ql] = sdm;
            a.dthr[ql] = -a.mu[ql] * sdm;
            a.dmu[ql] = -a.thr[ql] * sdm;
        }
    }
    // pass 2: d V_{j_t} += A_t d agg_l   (scatter into the padded value-map gradient)
    for (int t = 0; t < n; ++t) {
        const int j = a.nb_idx[ql * a.width + t];
        const float w = a.nb_wgt[ql * a.width + t];
        const int jy = j / a.g.W, jx = j - jy * a.g.W;
        float* dvj = dvm + ((size_t)jy * a.g.Wp + jx) * CH;
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (ok[u]) {
                float* p = dvj + (size_t)voff[u] * 4;
                atomicAdd(p + 0, w * d4[u].x); atomicAdd(p + 1, w * d4[u].y);
                atomicAdd(p + 2, w * d4[u].z); atomicAdd(p + 3, w * d4[u].w);
            }
    }
}

// d Xbar[b,:] = sum_l d mu_l Wq_l[:]   (adaptive modes): one block per (b, column quad group); fixed-order partials
__global__ __launch_bounds__(256) void dxbar_kernel(int L, const float* __restrict__ wq_rows, const float* __restrict__ dmu,
                                                    float* __restrict__ dxbar) {
    __shared__ float part[4][D];
    const int b = blockIdx.x;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};                                  // columns lane + 64 u
    for (int l = w; l < L; l += 4) {
        const float g = dmu[(size_t)b * L + l];
        const float* q = wq_rows + ((size_t)b * L + l) * D;
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int c = lane + 64 * u; if (c < D) acc[u] += g * q[c]; }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int c = lane + 64 * u; if (c < D) part[w][c] = acc[u]; }
    __syncthreads();
    for (int c = threadIdx.x; c < D; c += 256) dxbar[(size_t)b * D + c] = (part[0][c] + part[1][c]) + (part[2][c] + part[3][c]);
}

// d X[b,j,:] = d Xbar[b,:]/N (adaptive) or 0 (top-k): the dense part of the key-feature gradient
__global__ void dx_init_kernel(size_t n, int N, const float* __restrict__ dxbar, float* __restrict__ dx) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (dxbar == nullptr) { dx[i] = 0.f; return; }
    const size_t row = i / D; const int c = (int)(i % D);
    const size_t b = row / N;
    dx[i] = dxbar[b * D + c] / (float)N;
}

// one wave per query: d Wq_l = sum_t d S_t X_{j_t} + d mu_l Xbar;  d X_{j_t} += d S_t Wq_l
__global__ __launch_bounds__(256) void feature_backward_kernel(BwdArgs a) {
    const int lane = threadIdx.x & 63;
    const size_t ql = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ql >= (size_t)a.B * a.g.L) return;
    const int b = (int)(ql / a.g.L);
    const int n = a.nb_cnt[ql];
    const float* q = a.wq_rows + ql * D;
    float qv[4], acc[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int c = lane + 64 * u; qv[u] = (c < D) ? q[c] : 0.f; acc[u] = 0.f; }
    for (int t = 0; t < n; ++t) {
        const int j = a.nb_idx[ql * a.width + t];
        const float ds = a.dS[ql * a.width + t];
        const float* xr = a.x_rows + ((size_t)b * a.g.N + j) * D;
        float* dxr = a.dx_rows + ((size_t)b * a.g.N + j) * D;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c = lane + 64 * u;
            if (c < D) { acc[u] += ds * xr[c]; atomicAdd(dxr + c, ds * qv[u]); }
        }
    }
    if (a.mode != DAGL_MODE_TOPK) {
        const float g = a.dmu[ql];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int c = lane + 64 * u; if (c < D) acc[u] += g * (float)(a.colsum[(size_t)b * DS + c] / (double)a.g.N); }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int c = lane + 64 * u; if (c < D) a.dwq_rows[ql * D + c] = acc[u]; }
}

// padded NHWC gradient map -> NCHW [B,16,H,W]
__global__ void unpad_nchw_kernel(int H, int W, const float* __restrict__ src, float* __restrict__ dst) {
    const int Hp = H + 2 * PADPIX, Wp = W + 2 * PADPIX;
    const int b = blockIdx.z, y = blockIdx.y;
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= W) return;
    const float* s = src + ((((size_t)b * Hp + y + PADPIX) * Wp) + x + PADPIX) * CH;
#pragma unroll
    for (int c = 0; c < CH; ++c) dst[(((size_t)b * CH + c) * H + y) * W + x] = s[c];
}

int launch_core_backward(hipStream_t s, const BwdArgs& a, float* dxbar_ws, float* db2_nchw) {
    const Grid& g = a.g;
    const size_t nq = (size_t)a.B * g.L;
    {   // d agg
        const size_t items = (size_t)g.L * KS * KS;
        hipLaunchKernelGGL(unfold_dout_kernel, dim3((unsigned)((items + 255) / 256), a.B), dim3(256), 0, s, g, a.dout, a.dagg);
        DAGL_LAUNCH_CHECK("unfold_dout_kernel");
    }
    DAGL_HIP_TRY(hipMemsetAsync(a.db2p, 0, (size_t)a.B * g.Hp * g.Wp * CH * sizeof(float), s));
    hipLaunchKernelGGL(edge_backward_kernel, dim3((unsigned)((nq + 3) / 4)), dim3(256), 0, s, a);
    DAGL_LAUNCH_CHECK("edge_backward_kernel");
    const bool adaptive = (a.mode != DAGL_MODE_TOPK);
    if (adaptive) {
        hipLaunchKernelGGL(dxbar_kernel, dim3(a.B), dim3(256), 0, s, g.L, a.wq_rows, a.dmu, dxbar_ws);
        DAGL_LAUNCH_CHECK("dxbar_kernel");
    }
    {
        const size_t n = (size_t)a.B * g.N * D;
        hipLaunchKernelGGL(dx_init_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, n, g.N,
                           adaptive ? dxbar_ws : nullptr, a.dx_rows);
        DAGL_LAUNCH_CHECK("dx_init_kernel");
    }
    hipLaunchKernelGGL(feature_backward_kernel, dim3((unsigned)((nq + 3) / 4)), dim3(256), 0, s, a);
    DAGL_LAUNCH_CHECK("feature_backward_kernel");
    hipLaunchKernelGGL(unpad_nchw_kernel, dim3((g.W + 63) / 64, g.H, a.B), dim3(64), 0, s, g.H, g.W, a.db2p, db2_nchw);
    DAGL_LAUNCH_CHECK("unpad_nchw_kernel");
    return DAGL_OK;
}

// ---- forward-side helpers of the training entry point -------------------------------------------------------------
// dense feature rows [B,rows,196] -> the scans' layouts: fp32 [B,rows_alloc,204] (+ zero pad columns) and bf16 [.,216]
__global__ void rows_to_feat_kernel(int rows, int rows_alloc, int rows_alloc_h, const float* __restrict__ src,
                                    float* __restrict__ feat, unsigned short* __restrict__ feat_h) {
    const int b = blockIdx.y;
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)rows * DPAD) return;
    const int r = (int)(t / DPAD), c = (int)(t % DPAD);
    const float v = (c < D) ? src[((size_t)b * rows + r) * D + c] : 0.f;
    if (c < DS) feat[((size_t)b * rows_alloc + r) * DS + c] = v;
    if (feat_h != nullptr) {
        unsigned u = __float_as_uint(v);
        u = (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
        feat_h[((size_t)b * rows_alloc_h + r) * DSH + c] = (unsigned short)u;
    }
}

int launch_rows_to_feat(hipStream_t s, int B, int rows, const float* src, float* feat, uint16_t* feat_h) {
    const size_t n = (size_t)rows * DPAD;
    hipLaunchKernelGGL(rows_to_feat_kernel, dim3((unsigned)((n + 255) / 256), B), dim3(256), 0, s, rows, feat_rows(rows),
                       feat_rows_h(rows), src, feat, feat_h);
    DAGL_LAUNCH_CHECK("rows_to_feat_kernel");
    return DAGL_OK;
}

// column sums of dense rows [B,N,196] -> fp64 [B,204] (fixed order: per-wave strided partials + butterfly + 4-way add)
__global__ __launch_bounds__(256) void colsum_rows_kernel(int N, const float* __restrict__ rows, double* __restrict__ colsum) {
    __shared__ double part[4];
    const int b = blockIdx.y, c = blockIdx.x;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    double t = 0.0;
    for (int j = threadIdx.x; j < N; j += 256) t += (double)rows[((size_t)b * N + j) * D + c];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o);
    if (lane == 0) part[w] = t;
    __syncthreads();
    if (threadIdx.x == 0) colsum[(size_t)b * DS + c] = (part[0] + part[1]) + (part[2] + part[3]);
}

int launch_colsum_rows(hipStream_t s, int B, int N, const float* rows, double* colsum) {
    hipLaunchKernelGGL(colsum_rows_kernel, dim3(D, B), dim3(256), 0, s, N, rows, colsum);
    DAGL_LAUNCH_CHECK("colsum_rows_kernel");
    return DAGL_OK;
}

}  // namespace dagl
