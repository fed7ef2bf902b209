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
// The two scatter-adds (d V into the value map, d S Wq into the key features) are turned into gathers: the edge list is
// sorted by key with a stable LSD radix sort (edge_radix_* below: 8 bits per pass, per-block digit histograms, one scan,
// a scatter that ranks equal digits by position -- edge ids ascend inside a key's run), so every output element is summed
// by one thread in a fixed order: no atomics whose order could matter, bit-reproducible gradients.
#include <cstring>

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

// one wave per query: d A, softmax / logit backward, per-query threshold gradients
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
}

// d Xbar[b,:] = sum_l d mu_l Wq_l[:]   (adaptive modes): one block of 16 waves per image, a wave takes rows w, w + 16, .. four at a
// time; fixed-order partials (four waves walking 256 rows each, one row in flight: 135 us for 0.8 MB)
__global__ __launch_bounds__(1024) void dxbar_kernel(int L, const float* __restrict__ wq_rows, const float* __restrict__ dmu,
                                                     float* __restrict__ dxbar) {
    __shared__ float part[16][D];
    const int b = blockIdx.x;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};                                  // columns lane + 64 u
    for (int l0 = w; l0 < L; l0 += 64) {
        float g[4], q[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int l = l0 + 16 * r;
            g[r] = (l < L) ? dmu[(size_t)b * L + l] : 0.f;
            const float* qr = wq_rows + ((size_t)b * L + (l < L ? l : 0)) * D;
#pragma unroll
            for (int u = 0; u < 4; ++u) { const int c = lane + 64 * u; q[r][u] = (c < D) ? qr[c] : 0.f; }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[u] += g[r] * q[r][u];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int c = lane + 64 * u; if (c < D) part[w][c] = acc[u]; }
    __syncthreads();
    for (int c = threadIdx.x; c < D; c += 1024) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += part[k][c];
        dxbar[(size_t)b * D + c] = t;
    }
}

// ---- edges sorted by key ------------------------------------------------------------------------------------------
// sort key = b*N + j (invalid list slots: B*N, sorted to the end), value = edge id = (b*L + l)*width + t
__global__ void edge_keys_kernel(BwdArgs a, uint32_t* __restrict__ keys, uint32_t* __restrict__ vals) {
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t E = (size_t)a.B * a.g.L * a.width;
    if (e >= E) return;
    const size_t ql = e / a.width; const int t = (int)(e - ql * a.width);
    const size_t b = ql / a.g.L;
    const bool valid = t < a.nb_cnt[ql];
    keys[e] = valid ? (uint32_t)(b * a.g.N + a.nb_idx[e]) : (uint32_t)((size_t)a.B * a.g.N);
    vals[e] = (uint32_t)e;
}

// run boundaries of the sorted keys: seg[2k] = first position of key k, seg[2k+1] = one past its last (0,0 = no edge)
__global__ void segment_bounds_kernel(size_t E, uint32_t n_keys, const uint32_t* __restrict__ keys, uint32_t* __restrict__ seg) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= E) return;
    const uint32_t k = keys[i];
    if (k >= n_keys) return;
    if (i == 0 || keys[i - 1] != k) seg[2 * (size_t)k] = (uint32_t)i;
    if (i + 1 == E || keys[i + 1] != k) seg[2 * (size_t)k + 1] = (uint32_t)(i + 1);
}

// ---- segmented reduction over the key-sorted edges ------------------------------------------------------------------
// Every key j with incoming edges gets one row of 980 floats = [ sum_e A_e d agg_{l_e}  (784: d V patch of key j) |
// sum_e d S_e Wq_{l_e} (196: d X_j without the dense term) ], stored at rowbuf[first sorted position of j's run].
// In-degrees are heavy-tailed (a few hub keys are neighbours of hundreds of queries), so the work is cut into fixed
// chunks of SEG_C sorted edges, one wave each; a run that crosses chunk boundaries leaves per-chunk partial rows that
// row_fixup_kernel adds up in chunk order.  Everything is summed in a fixed order: bit-reproducible.
constexpr int SEG_C = 16;
constexpr int ROWF = P + D;                 // 980 floats
constexpr int ROWQ = ROWF / 4;              // 245 float4: lane owns slots lane + 64 u, u < 4

__device__ __forceinline__ float4 f4_fma(float c, const float4& v, const float4& a) {
    return make_float4(a.x + c * v.x, a.y + c * v.y, a.z + c * v.z, a.w + c * v.w);
}

__global__ __launch_bounds__(256) void edge_segreduce_kernel(BwdArgs a, size_t E, uint32_t n_keys,
                                                              const uint32_t* __restrict__ keys, const uint32_t* __restrict__ eids,
                                                              const uint32_t* __restrict__ seg, float* __restrict__ rowbuf,
                                                              float* __restrict__ part /* [chunks][2][980] */) {
    const int lane = threadIdx.x & 63;
    const size_t chunk = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const size_t p0 = chunk * SEG_C;
    if (p0 >= E) return;
    const size_t p1 = (p0 + SEG_C < E) ? p0 + SEG_C : E;
    float4 acc[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) acc[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    uint32_t cur = keys[p0];
    auto flush = [&](uint32_t key, size_t pos_end) {
        // the run of `key` ends at sorted position pos_end (exclusive) as far as this chunk sees it
        if (key >= n_keys) return;                                    // invalid list slots
        const uint32_t e0 = seg[2 * (size_t)key], e1 = seg[2 * (size_t)key + 1];
        float* dst;
        if (e0 >= p0 && e1 <= p1) dst = rowbuf + (size_t)e0 * ROWF;           // run entirely inside this chunk
        else if (e0 < p0) dst = part + (chunk * 2 + 0) * ROWF;                 // run came in from an earlier chunk
        else dst = part + (chunk * 2 + 1) * ROWF;                              // run continues into later chunks
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int r = lane + 64 * u;
            if (r < ROWQ) reinterpret_cast<float4*>(dst)[r] = acc[u];
            acc[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    for (size_t p = p0; p < p1; ++p) {
        const uint32_t key = keys[p];
        if (key != cur) { flush(cur, p); cur = key; }
        if (key >= n_keys) break;                                     // sorted: only invalid slots follow
        const uint32_t eid = eids[p];
        const float w = a.nb_wgt[eid], ds = a.dS[eid];
        const size_t ql = eid / (uint32_t)a.width;
        const float4* dg = reinterpret_cast<const float4*>(a.dagg + ql * P);
        const float4* wq = reinterpret_cast<const float4*>(a.wq_rows + ql * D);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int r = lane + 64 * u;
            if (r < P / 4) acc[u] = f4_fma(w, dg[r], acc[u]);
            else if (r < ROWQ) acc[u] = f4_fma(ds, wq[r - P / 4], acc[u]);
        }
    }
    flush(cur, p1);
}

// runs that cross chunk boundaries: the chunk in which such a run STARTS adds up its tail partial and the head
// partials of the following chunks, in chunk order, into the run's row
__global__ __launch_bounds__(256) void row_fixup_kernel(size_t E, uint32_t n_keys, const uint32_t* __restrict__ keys,
                                                         const uint32_t* __restrict__ seg, float* __restrict__ rowbuf,
                                                         const float* __restrict__ part) {
    const int lane = threadIdx.x & 63;
    const size_t chunk = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const size_t p0 = chunk * SEG_C;
    if (p0 >= E) return;
    const size_t p1 = (p0 + SEG_C < E) ? p0 + SEG_C : E;
    const uint32_t key = keys[p1 - 1];                                // the only run that can leave through the end
    if (key >= n_keys) return;
    const uint32_t e0 = seg[2 * (size_t)key], e1 = seg[2 * (size_t)key + 1];
    if (e1 <= p1 || e0 < p0) return;                                  // ends here, or started in an earlier chunk
    const size_t c_last = ((size_t)e1 - 1) / SEG_C;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int r = lane + 64 * u;
        if (r >= ROWQ) continue;
        float4 t = reinterpret_cast<const float4*>(part + (chunk * 2 + 1) * ROWF)[r];
        for (size_t c = chunk + 1; c <= c_last; ++c) {
            const float4 h = reinterpret_cast<const float4*>(part + (c * 2 + 0) * ROWF)[r];
            t.x += h.x; t.y += h.y; t.z += h.z; t.w += h.w;
        }
        reinterpret_cast<float4*>(rowbuf + (size_t)e0 * ROWF)[r] = t;
    }
}

// d b2[b,c,y,x] = sum over the keys j whose 7x7 window covers (y,x) of (d V patch of j)[(kh,kw,c)]; thread = pixel
__global__ __launch_bounds__(256) void dvalue_fold_kernel(BwdArgs a, const uint32_t* __restrict__ seg,
                                                           const float* __restrict__ rowbuf, float* __restrict__ db2) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)a.B * a.g.N) return;
    const size_t b = t / a.g.N; const size_t r = t - b * a.g.N;
    const int y = (int)(r / a.g.W), x = (int)(r - (size_t)y * a.g.W);
    float4 acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int kh = 0; kh < KS; ++kh) {
        const int jy = y + 3 - kh;
        if (jy < 0 || jy >= a.g.H) continue;
#pragma unroll
        for (int kw = 0; kw < KS; ++kw) {
            const int jx = x + 3 - kw;
            if (jx < 0 || jx >= a.g.W) continue;
            const size_t key = b * a.g.N + (size_t)jy * a.g.W + jx;
            const uint2 e = *reinterpret_cast<const uint2*>(seg + 2 * key);
            if (e.y > e.x) {
                const float4* row = reinterpret_cast<const float4*>(rowbuf + (size_t)e.x * ROWF + (kh * KS + kw) * CH);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 v = row[q];
                    acc[q].x += v.x; acc[q].y += v.y; acc[q].z += v.z; acc[q].w += v.w;
                }
            }
        }
    }
    float* o = db2 + (b * CH * a.g.H + y) * a.g.W + x;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        o[(size_t)(4 * q + 0) * a.g.N] = acc[q].x; o[(size_t)(4 * q + 1) * a.g.N] = acc[q].y;
        o[(size_t)(4 * q + 2) * a.g.N] = acc[q].z; o[(size_t)(4 * q + 3) * a.g.N] = acc[q].w;
    }
}

// d X[b,j,:] = d Xbar[b,:]/N (adaptive modes) + (d X part of key j's row);  thread = (key, column quad)
__global__ __launch_bounds__(256) void dx_finalize_kernel(BwdArgs a, const uint32_t* __restrict__ seg,
                                                           const float* __restrict__ rowbuf, const float* __restrict__ dxbar) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    constexpr int Q = D / 4;                                   // 49 float4 per feature row
    if (t >= (size_t)a.B * a.g.N * Q) return;
    const size_t key = t / Q; const int c4 = (int)(t - key * Q);
    const size_t b = key / a.g.N;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (dxbar != nullptr) {
        const float4 xb = *reinterpret_cast<const float4*>(dxbar + b * D + c4 * 4);
        const float inv = 1.0f / (float)a.g.N;
        acc = make_float4(xb.x * inv, xb.y * inv, xb.z * inv, xb.w * inv);
    }
    const uint2 e = *reinterpret_cast<const uint2*>(seg + 2 * key);
    if (e.y > e.x) {
        const float4 v = *reinterpret_cast<const float4*>(rowbuf + (size_t)e.x * ROWF + P + c4 * 4);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    *reinterpret_cast<float4*>(a.dx_rows + key * D + c4 * 4) = acc;
}

// one wave per query: d Wq_l = sum_t d S_t X_{j_t} + d mu_l Xbar
__global__ __launch_bounds__(256) void dwq_gather_kernel(BwdArgs a) {
    const int lane = threadIdx.x & 63;
    const size_t ql = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ql >= (size_t)a.B * a.g.L) return;
    const int b = (int)(ql / a.g.L);
    const int n = a.nb_cnt[ql];
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int t = 0; t < n; ++t) {
        const int j = a.nb_idx[ql * a.width + t];
        const float ds = a.dS[ql * a.width + t];
        const float* xr = a.x_rows + ((size_t)b * a.g.N + j) * D;
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int c = lane + 64 * u; if (c < D) acc[u] += ds * xr[c]; }
    }
    if (a.mode != DAGL_MODE_TOPK) {
        const float g = a.dmu[ql];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c = lane + 64 * u;
            if (c < D) acc[u] += g * (float)(a.colsum[(size_t)b * DS + c] / (double)a.g.N);
        }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int c = lane + 64 * u; if (c < D) a.dwq_rows[ql * D + c] = acc[u]; }
}

int launch_unfold_dout(hipStream_t s, int B, const Grid& g, const float* dout, float* dagg) {
    const size_t items = (size_t)g.L * KS * KS;
    hipLaunchKernelGGL(unfold_dout_kernel, dim3((unsigned)((items + 255) / 256), B), dim3(256), 0, s, g, dout, dagg);
    DAGL_LAUNCH_CHECK("unfold_dout_kernel");
    return DAGL_OK;
}

int launch_dxbar(hipStream_t s, int B, int L, const float* wq_rows, const float* dmu, float* dxbar) {
    hipLaunchKernelGGL(dxbar_kernel, dim3(B), dim3(1024), 0, s, L, wq_rows, dmu, dxbar);
    DAGL_LAUNCH_CHECK("dxbar_kernel");
    return DAGL_OK;
}

static int key_bits(size_t n_keys) {             // radix bits that cover 0..n_keys (n_keys itself = the invalid marker)
    int bits = 1;
    while (((size_t)1 << bits) <= n_keys) ++bits;
    return bits;
}

// ---- stable LSD radix sort of (key, edge id) pairs, 8 bits per pass ------------------------------------------------------
// A block owns RS_TILE consecutive positions, visited in RS_ITEMS rounds of 256 (position = base + 256 round + thread): that
// order IS the stable order.  Pass = (1) per-block digit counts (LDS integer atomics: the counts do not depend on their
// order), stored digit-major; (2) one exclusive scan over [256 digits][blocks]; (3) scatter: an element's slot = its block's
// start for the digit + the digit's elements in earlier rounds + in earlier waves of this round + in lower lanes of its wave
// (eight ballots match the lanes that hold the same digit).
constexpr int RS_ITEMS = 8;
constexpr int RS_TILE = 256 * RS_ITEMS;

__global__ __launch_bounds__(256) void edge_radix_hist_kernel(size_t n, int shift, const uint32_t* __restrict__ keys,
                                                               uint32_t* __restrict__ blockhist, unsigned nblk) {
    __shared__ unsigned hist[256];
    hist[threadIdx.x] = 0u;
    __syncthreads();
    const size_t base = (size_t)blockIdx.x * RS_TILE;
#pragma unroll
    for (int r = 0; r < RS_ITEMS; ++r) {
        const size_t p = base + 256 * r + threadIdx.x;
        if (p < n) atomicAdd(&hist[(keys[p] >> shift) & 255u], 1u);
    }
    __syncthreads();
    blockhist[(size_t)threadIdx.x * nblk + blockIdx.x] = hist[threadIdx.x];
}

// exclusive scan of m values in place (one block; m = 256 x blocks is a few ten thousand)
__global__ __launch_bounds__(1024) void edge_radix_scan_kernel(size_t m, uint32_t* __restrict__ v) {
    __shared__ uint32_t part[1024];
    const int t = threadIdx.x;
    const size_t per = (m + 1023) / 1024;
    const size_t i0 = (size_t)t * per < m ? (size_t)t * per : m, i1 = i0 + per < m ? i0 + per : m;
    uint32_t s = 0;
    for (size_t i = i0; i < i1; ++i) s += v[i];
    part[t] = s;
    __syncthreads();
    if (t == 0) { uint32_t run = 0; for (int j = 0; j < 1024; ++j) { const uint32_t x = part[j]; part[j] = run; run += x; } }
    __syncthreads();
    uint32_t run = part[t];
    for (size_t i = i0; i < i1; ++i) { const uint32_t x = v[i]; v[i] = run; run += x; }
}

__global__ __launch_bounds__(256) void edge_radix_scatter_kernel(size_t n, int shift, const uint32_t* __restrict__ keys_in,
                                                                  const uint32_t* __restrict__ vals_in,
                                                                  uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out,
                                                                  const uint32_t* __restrict__ blockoff, unsigned nblk) {
    __shared__ unsigned start[256];            // next free slot of each digit for this block
    __shared__ unsigned wcount[4][256];        // this round's digit counts per wave
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    start[t] = blockoff[(size_t)t * nblk + blockIdx.x];
    const size_t base = (size_t)blockIdx.x * RS_TILE;
    for (int r = 0; r < RS_ITEMS; ++r) {
        wcount[0][t] = 0u; wcount[1][t] = 0u; wcount[2][t] = 0u; wcount[3][t] = 0u;
        __syncthreads();
        const size_t p = base + 256 * r + t;
        const bool have = p < n;
        const uint32_t key = have ? keys_in[p] : 0u, val = have ? vals_in[p] : 0u;
        const unsigned d = (key >> shift) & 255u;
        unsigned long long same = __ballot(have);                          // lanes of this wave holding the same digit
#pragma unroll
        for (int bit = 0; bit < 8; ++bit) {
            const unsigned long long m = __ballot((d >> bit) & 1u);
            same &= ((d >> bit) & 1u) ? m : ~m;
        }
        const unsigned below = (unsigned)__popcll(same & ((1ull << lane) - 1ull));
        if (have && below == 0u) wcount[w][d] = (unsigned)__popcll(same);   // the digit's lowest lane speaks for the wave
        __syncthreads();
        if (have) {
            unsigned pos = start[d] + below;
            for (int w2 = 0; w2 < w; ++w2) pos += wcount[w2][d];
            keys_out[pos] = key; vals_out[pos] = val;
        }
        __syncthreads();
        start[t] += wcount[0][t] + wcount[1][t] + wcount[2][t] + wcount[3][t];
        __syncthreads();
    }
}

static unsigned radix_blocks(size_t n) { return (unsigned)((n + RS_TILE - 1) / RS_TILE); }

// sorts (keys_a, vals_a) by the low `bits` bits of the key, ping-ponging with (keys_b, vals_b); *sorted_in_b tells where the
// result ended up
static int edge_radix_sort(hipStream_t s, size_t n, int bits, uint32_t* keys_a, uint32_t* vals_a, uint32_t* keys_b,
                           uint32_t* vals_b, uint32_t* blockhist, bool* sorted_in_b) {
    const unsigned nblk = radix_blocks(n);
    bool in_a = true;
    for (int shift = 0; shift < bits; shift += 8) {
        const uint32_t* ki = in_a ? keys_a : keys_b; const uint32_t* vi = in_a ? vals_a : vals_b;
        uint32_t* ko = in_a ? keys_b : keys_a; uint32_t* vo = in_a ? vals_b : vals_a;
        hipLaunchKernelGGL(edge_radix_hist_kernel, dim3(nblk), dim3(256), 0, s, n, shift, ki, blockhist, nblk);
        DAGL_LAUNCH_CHECK("edge_radix_hist_kernel");
        hipLaunchKernelGGL(edge_radix_scan_kernel, dim3(1), dim3(1024), 0, s, (size_t)256 * nblk, blockhist);
        DAGL_LAUNCH_CHECK("edge_radix_scan_kernel");
        hipLaunchKernelGGL(edge_radix_scatter_kernel, dim3(nblk), dim3(256), 0, s, n, shift, ki, vi, ko, vo, blockhist, nblk);
        DAGL_LAUNCH_CHECK("edge_radix_scatter_kernel");
        in_a = !in_a;
    }
    *sorted_in_b = !in_a;
    return DAGL_OK;
}

size_t edge_rowbuf_floats(size_t n_edges) { return n_edges * ROWF; }
size_t edge_part_floats(size_t n_edges) { return ((n_edges + SEG_C - 1) / SEG_C) * 2 * ROWF; }

size_t edge_sort_temp_bytes(size_t n_edges, size_t n_keys) {       // [256 digits][blocks] counts of one radix pass
    (void)n_keys;
    return (size_t)256 * radix_blocks(n_edges) * sizeof(uint32_t);
}

int launch_core_backward(hipStream_t s, const BwdArgs& a, const BwdSortWs& w, float* dxbar_ws, float* db2_nchw) {
    const Grid& g = a.g;
    const size_t nq = (size_t)a.B * g.L;
    const size_t E = nq * a.width, n_keys = (size_t)a.B * g.N;
    {   // d agg
        const size_t items = (size_t)g.L * KS * KS;
        hipLaunchKernelGGL(unfold_dout_kernel, dim3((unsigned)((items + 255) / 256), a.B), dim3(256), 0, s, g, a.dout, a.dagg);
        DAGL_LAUNCH_CHECK("unfold_dout_kernel");
    }
    hipLaunchKernelGGL(edge_backward_kernel, dim3((unsigned)((nq + 3) / 4)), dim3(256), 0, s, a);
    DAGL_LAUNCH_CHECK("edge_backward_kernel");
    // edges by key
    hipLaunchKernelGGL(edge_keys_kernel, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, s, a, w.keys_in, w.vals_in);
    DAGL_LAUNCH_CHECK("edge_keys_kernel");
    bool in_b = false;
    {
        const int rc = edge_radix_sort(s, E, key_bits(n_keys), w.keys_in, w.vals_in, w.keys_out, w.vals_out,
                                       static_cast<uint32_t*>(w.temp), &in_b);
        if (rc) return rc;
    }
    const uint32_t* skeys = in_b ? w.keys_out : w.keys_in;
    const uint32_t* svals = in_b ? w.vals_out : w.vals_in;
    DAGL_HIP_TRY(hipMemsetAsync(w.seg, 0, 2 * n_keys * sizeof(uint32_t), s));
    hipLaunchKernelGGL(segment_bounds_kernel, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, s, E, (uint32_t)n_keys, skeys, w.seg);
    DAGL_LAUNCH_CHECK("segment_bounds_kernel");
    {
        const size_t chunks = (E + SEG_C - 1) / SEG_C;
        hipLaunchKernelGGL(edge_segreduce_kernel, dim3((unsigned)((chunks + 3) / 4)), dim3(256), 0, s, a, E, (uint32_t)n_keys,
                           skeys, svals, w.seg, w.rowbuf, w.part);
        DAGL_LAUNCH_CHECK("edge_segreduce_kernel");
        hipLaunchKernelGGL(row_fixup_kernel, dim3((unsigned)((chunks + 3) / 4)), dim3(256), 0, s, E, (uint32_t)n_keys,
                           skeys, w.seg, w.rowbuf, w.part);
        DAGL_LAUNCH_CHECK("row_fixup_kernel");
        hipLaunchKernelGGL(dvalue_fold_kernel, dim3((unsigned)((n_keys + 255) / 256)), dim3(256), 0, s, a, w.seg, w.rowbuf, db2_nchw);
        DAGL_LAUNCH_CHECK("dvalue_fold_kernel");
    }
    const bool adaptive = (a.mode != DAGL_MODE_TOPK);
    if (adaptive) {
        hipLaunchKernelGGL(dxbar_kernel, dim3(a.B), dim3(1024), 0, s, g.L, a.wq_rows, a.dmu, dxbar_ws);
        DAGL_LAUNCH_CHECK("dxbar_kernel");
    }
    {
        const size_t n = n_keys * (D / 4);
        hipLaunchKernelGGL(dx_finalize_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a, w.seg, w.rowbuf,
                           adaptive ? dxbar_ws : nullptr);
        DAGL_LAUNCH_CHECK("dx_finalize_kernel");
    }
    hipLaunchKernelGGL(dwq_gather_kernel, dim3((unsigned)((nq + 3) / 4)), dim3(256), 0, s, a);
    DAGL_LAUNCH_CHECK("dwq_gather_kernel");
    return DAGL_OK;
}

// ---- forward-side helpers of the training entry point -------------------------------------------------------------
// dense feature rows [B,rows,196] -> the scans' layouts: fp32 [B,rows_alloc,204] (+ zero pad columns) and bf16 [.,216]
__global__ void rows_to_feat_kernel(int rows, int rows_alloc, int rows_alloc_h, const float* __restrict__ src,
                                    float* __restrict__ feat, unsigned short* __restrict__ feat_h, RangeTag range) {
    const int b = blockIdx.y;
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)rows * DPAD) return;
    const int r = (int)(t / DPAD), c = (int)(t % DPAD);
    const float v = (c < D) ? src[((size_t)b * rows + r) * D + c] : 0.f;
    if (range.word != nullptr && !(fabsf(v) < 3.0e38f)) *range.word = range.tag;
    if (c < DS) feat[((size_t)b * rows_alloc + r) * DS + c] = v;
    if (feat_h != nullptr) {
        unsigned u = __float_as_uint(v);
        u = (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
        feat_h[((size_t)b * rows_alloc_h + r) * DSH + c] = (unsigned short)u;
    }
}

int launch_rows_to_feat(hipStream_t s, int B, int rows, const float* src, float* feat, uint16_t* feat_h, RangeTag range) {
    const size_t n = (size_t)rows * DPAD;
    hipLaunchKernelGGL(rows_to_feat_kernel, dim3((unsigned)((n + 255) / 256), B), dim3(256), 0, s, rows, feat_rows(rows),
                       feat_rows_h(rows), src, feat, feat_h, range);
    DAGL_LAUNCH_CHECK("rows_to_feat_kernel");
    return DAGL_OK;
}

// column sums of dense rows [B,N,196] -> fp64 [B,204] (fixed order: per-thread strided partials + butterfly + 4-way add).
// Block = (image, four columns): a thread reads 16 bytes of a row (one column per block and 4 bytes per row and thread was 172 us
// for 103 MB at [8, 16384 x 196]: 196 blocks per image asking the L2 for the same lines).
__global__ __launch_bounds__(256) void colsum_rows_kernel(int N, const float* __restrict__ rows, double* __restrict__ colsum) {
    __shared__ double part[4][4];
    const int b = blockIdx.y, c4 = blockIdx.x;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    double t0 = 0.0, t1 = 0.0, t2 = 0.0, t3 = 0.0;
    const float4* base = reinterpret_cast<const float4*>(rows + (size_t)b * N * D) + c4;
#pragma unroll 8
    for (int j = threadIdx.x; j < N; j += 256) {
        const float4 v = base[(size_t)j * (D / 4)];
        t0 += (double)v.x; t1 += (double)v.y; t2 += (double)v.z; t3 += (double)v.w;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { t0 += __shfl_xor(t0, o); t1 += __shfl_xor(t1, o); t2 += __shfl_xor(t2, o); t3 += __shfl_xor(t3, o); }
    if (lane == 0) { part[w][0] = t0; part[w][1] = t1; part[w][2] = t2; part[w][3] = t3; }
    __syncthreads();
    if (threadIdx.x < 4) {
        const int u = threadIdx.x;
        colsum[(size_t)b * DS + 4 * c4 + u] = (part[0][u] + part[1][u]) + (part[2][u] + part[3][u]);
    }
}

// Batches (training: 8-32 images): block = (image, seven float4 columns = 112 contiguous bytes of a row), 1024 threads = 146 row
// subsets x 7 columns.  The form above has 49 blocks per image each taking 16 bytes of every 128-byte line (8 x the L2 traffic:
// 108 us for 103 MB at [8, 16384 x 196]); here a line is touched by two blocks at most.  Fixed order: per-thread strided partials,
// then the 146 subsets of a column one after the other.
constexpr int CSW_SUB = 146;
__global__ __launch_bounds__(1024) void colsum_rows_wide_kernel(int N, const float* __restrict__ rows, double* __restrict__ colsum) {
    __shared__ double part[CSW_SUB][7][4];                                        // 32 KiB
    const int b = blockIdx.y, c0 = blockIdx.x * 7;
    const int t = threadIdx.x;
    const int sub = t / 7, c4 = t - sub * 7;
    if (sub < CSW_SUB) {
        double t0 = 0.0, t1 = 0.0, t2 = 0.0, t3 = 0.0;
        const float4* base = reinterpret_cast<const float4*>(rows + (size_t)b * N * D) + c0 + c4;
#pragma unroll 8
        for (int j = sub; j < N; j += CSW_SUB) {
            const float4 v = base[(size_t)j * (D / 4)];
            t0 += (double)v.x; t1 += (double)v.y; t2 += (double)v.z; t3 += (double)v.w;
        }
        part[sub][c4][0] = t0; part[sub][c4][1] = t1; part[sub][c4][2] = t2; part[sub][c4][3] = t3;
    }
    __syncthreads();
    if (t < 28) {
        const int cc = t >> 2, u = t & 3;
        double a = 0.0;
        for (int q = 0; q < CSW_SUB; ++q) a += part[q][cc][u];
        colsum[(size_t)b * DS + 4 * (c0 + cc) + u] = a;
    }
}

int launch_colsum_rows(hipStream_t s, int B, int N, const float* rows, double* colsum) {
    static_assert(D / 4 == 49, "seven blocks of seven float4 columns");
    if (7 * B >= 48)
        hipLaunchKernelGGL(colsum_rows_wide_kernel, dim3(7, B), dim3(1024), 0, s, N, rows, colsum);
    else
        hipLaunchKernelGGL(colsum_rows_kernel, dim3(D / 4, B), dim3(256), 0, s, N, rows, colsum);
    DAGL_LAUNCH_CHECK("colsum_rows_kernel");
    return DAGL_OK;
}

}  // namespace dagl
