// Batched fp32 GEMM on the matrix cores: C[b] = alpha * A[b] B[b] + beta * C[b] (+ bias, relu), v_mfma_f32_32x32x2_f32.
//
// The training path's dense stages (dense_train.hip: S = Wq X^T, agg = A V, d A = d agg V^T, d Wq = d S X,
// d X = d S^T Wq, d V = A^T d agg -- the reference's torch.matmul / torch.mm of DN_Gray/model/dagl.py:250,263 and what
// autograd derives from them) are plain matrix products over operands that already sit in HBM in row-major form, so they
// share this one kernel.  The f32-input MFMA is bitwise an fmaf chain along K (one rounding per product, no wider
// accumulator): the numerics class of the reference's own fp32 GEMMs, which is what the gradient parity tests need.
//
// Tile 128 x 128 x 16, 4 waves (2 x 2), each wave 64 x 64 = 2 x 2 MFMA tiles; operands staged K-major in LDS
// (As[k][m], Bs[k][n]) so that a fragment read is 32 consecutive words per half-wave (conflict-free); global tiles are
// fetched as 16-byte vectors along whichever dimension is contiguous in memory and double-buffered through registers.
// Either operand may be stored "K-contiguous" (row-major M x K / N x K) or "K-major" (K x M / K x N), which covers NN, NT,
// TN without copies.  Products with few output tiles and a long K (the weight gradients: 196 x 784 outputs over 131072
// patch rows) are cut into K slices that write their own partial C, summed in slice order by gemm32_reduce_kernel:
// no atomics anywhere, deterministic.  Optional chunked accumulation keeps the fmaf chains short.
#include <stdlib.h>

#include "dagl_common.h"

namespace dagl {

constexpr int G_BM = 128, G_BN = 128, G_BK = 16, G_PAD = 4;     // (k-tiles of 32 measured slower: 824 vs 657 us on the fc2 forward --
                                                                   // the LDS then holds two blocks per CU instead of four)
constexpr int G_KT = G_BK / 4;                                     // threads along k of a K-contiguous operand tile
constexpr int G_VEC = G_BK / 8;                                    // float4 per thread and operand tile

// one operand tile: 128 (rows = M or N index) x G_BK (k) floats -> G_VEC float4 per thread
template <bool KC>
struct TileLoader {
    const float* base; long long ld; int rows, K, row0; bool vec;
    float4 v[G_VEC];
    __device__ __forceinline__ void load(int k0, int tid) {
        // Interior k range (block-uniform) and whole vectors: unconditional loads from clamped addresses, zeroed afterwards --
        // a predicated load makes the compiler wait for each one before it issues the next, and four exposed memory round
        // trips per tile are what this kernel used to cost (27 TFLOP/s on the fc2 weight gradient).
        const bool interior = vec && (k0 + G_BK <= K);
        if (KC) {                                   // element (r, k) at base[r * ld + k]: vectors along k
            if (interior) {
#pragma unroll
                for (int u = 0; u < G_VEC; ++u) {
                    const int r = row0 + tid / G_KT + (256 / G_KT) * u, k = k0 + 4 * (tid % G_KT);
                    const int rc = r < rows ? r : rows - 1;
                    const float4 t = *reinterpret_cast<const float4*>(base + (long long)rc * ld + k);
                    v[u] = (r < rows) ? t : make_float4(0.f, 0.f, 0.f, 0.f);
                }
                return;
            }
#pragma unroll
            for (int u = 0; u < G_VEC; ++u) {
                const int r = row0 + tid / G_KT + (256 / G_KT) * u, k = k0 + 4 * (tid % G_KT);
                float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
                if (r < rows) {
                    const float* p = base + (long long)r * ld + k;
                    if (vec && k + 3 < K) t = *reinterpret_cast<const float4*>(p);
                    else {
                        if (k < K) t.x = p[0];
                        if (k + 1 < K) t.y = p[1];
                        if (k + 2 < K) t.z = p[2];
                        if (k + 3 < K) t.w = p[3];
                    }
                }
                v[u] = t;
            }
        } else {                                    // element (r, k) at base[k * ld + r]: vectors along r
            if (interior && (rows & 3) == 0 && rows >= 4) {
#pragma unroll
                for (int u = 0; u < G_VEC; ++u) {
                    const int k = k0 + (tid >> 5) + 8 * u, r = row0 + 4 * (tid & 31);
                    const int rc = r < rows ? r : rows - 4;
                    const float4 t = *reinterpret_cast<const float4*>(base + (long long)k * ld + rc);
                    v[u] = (r < rows) ? t : make_float4(0.f, 0.f, 0.f, 0.f);
                }
                return;
            }
#pragma unroll
            for (int u = 0; u < G_VEC; ++u) {
                const int k = k0 + (tid >> 5) + 8 * u, r = row0 + 4 * (tid & 31);
                float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
                if (k < K) {
                    const float* p = base + (long long)k * ld + r;
                    if (vec && r + 3 < rows) t = *reinterpret_cast<const float4*>(p);
                    else {
                        if (r < rows) t.x = p[0];
                        if (r + 1 < rows) t.y = p[1];
                        if (r + 2 < rows) t.z = p[2];
                        if (r + 3 < rows) t.w = p[3];
                    }
                }
                v[u] = t;
            }
        }
    }
    __device__ __forceinline__ void store(float (*s)[G_BM + G_PAD], int tid) const {
        if (KC) {
#pragma unroll
            for (int u = 0; u < G_VEC; ++u) {
                const int r = tid / G_KT + (256 / G_KT) * u, k = 4 * (tid % G_KT);
                s[k][r] = v[u].x; s[k + 1][r] = v[u].y; s[k + 2][r] = v[u].z; s[k + 3][r] = v[u].w;
            }
        } else {
#pragma unroll
            for (int u = 0; u < G_VEC; ++u) {
                const int k = (tid >> 5) + 8 * u, r = 4 * (tid & 31);
                *reinterpret_cast<float4*>(&s[k][r]) = v[u];
            }
        }
    }
};

// CHUNK: chunked accumulation compiled in (a second set of 64 accumulator registers; without it a third block fits a CU)
// MLOOP: the launch's y extent is shorter than the row tiles and a block walks them (device-side row limit; its own instance: a loop
// around the tile keeps addresses live across it and spills ~50 registers, which the straight-line instances must not pay)
template <bool AKC, bool BKC, bool CHUNK, bool MLOOP = false>
__global__ __launch_bounds__(256, MLOOP ? 2 : (CHUNK ? 3 : 4)) void gemm32_kernel(Gemm32 g, int vecA, int vecB) {
    __shared__ __attribute__((aligned(16))) float As[2][G_BK][G_BM + G_PAD];
    __shared__ __attribute__((aligned(16))) float Bs[2][G_BK][G_BN + G_PAD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int i = lane & 31, h = lane >> 5;
    const int n0 = blockIdx.x * G_BN;
    const long long bz = blockIdx.z;
    if (g.m_limit != nullptr) {                     // row count known on the device only: blocks past it have nothing to do
        int ml = *g.m_limit;
        if (ml <= g.m_limit_floor || ml > g.m_limit_ceil) ml = 0;
        if (ml < g.M) g.M = ml;
    }
    // the grid's y extent may be SHORTER than the row tiles (launch_gemm32 with a device-side row limit: a product that usually has no
    // or few rows -- the adaptive mode's flagged queries -- used to launch 8192 blocks that looked at the limit and left: 4.9 us);
    // a block walks its row tiles
    for (int m0 = blockIdx.y * G_BM; m0 < g.M; m0 += gridDim.y * G_BM) {
    TileLoader<AKC> la{g.A + bz * g.sA, g.lda, g.M, g.K, m0, vecA != 0, {}};
    TileLoader<BKC> lb{g.B + bz * g.sB, g.ldb, g.N, g.K, n0, vecB != 0, {}};

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // split-K launch (launch_gemm32): slice bz covers k in [bz * K, min((bz + 1) * K, k_total)) and writes its own C slice
    if (g.k_total > 0) {
        const int k_eff = g.k_total - (int)bz * g.K;
        la.K = lb.K = k_eff < g.K ? k_eff : g.K;
    }
    const int K = la.K;
    f32x16 sum[CHUNK ? 2 : 1][CHUNK ? 2 : 1];
    if (CHUNK) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) sum[a][b][r] = 0.f;
    }
    const int nt = (K + G_BK - 1) / G_BK;
    la.load(0, tid); lb.load(0, tid);
    la.store(As[0], tid); lb.store(Bs[0], tid);
    __syncthreads();
    for (int t = 0; t < nt; ++t) {
        const int cur = t & 1;
#ifdef DAGL_ABLATION
        if (t + 1 < nt && !(g.variant & 1)) { la.load((t + 1) * G_BK, tid); lb.load((t + 1) * G_BK, tid); }
#else
        if (t + 1 < nt) { la.load((t + 1) * G_BK, tid); lb.load((t + 1) * G_BK, tid); }
#endif
#pragma unroll
        for (int half = 0; half < G_BK / 16; ++half) {
#pragma unroll
            for (int kk = 16 * half; kk < 16 * half + 16; kk += 2) {
                const float a0 = As[cur][kk + h][wm * 64 + i], a1 = As[cur][kk + h][wm * 64 + 32 + i];
                const float b0 = Bs[cur][kk + h][wn * 64 + i], b1 = Bs[cur][kk + h][wn * 64 + 32 + i];
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
            }
            if (CHUNK && (t * (G_BK / 16) + half + 1) % g.chunk_tiles == 0) {
                // chunked accumulation: a chain of chunk_tiles * 16 products per partial sum instead of K (the partial sums
                // are added in fp32): the rounding error of a long fmaf chain grows with its length
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b)
#pragma unroll
                        for (int r = 0; r < 16; ++r) { sum[a][b][r] += acc[a][b][r]; acc[a][b][r] = 0.f; }
            }
        }
#ifdef DAGL_ABLATION
        if (t + 1 < nt && !(g.variant & 2)) { la.store(As[cur ^ 1], tid); lb.store(Bs[cur ^ 1], tid); }
        if (!(g.variant & 4)) __syncthreads();
#else
        if (t + 1 < nt) { la.store(As[cur ^ 1], tid); lb.store(Bs[cur ^ 1], tid); }
        __syncthreads();
#endif
    }
    if (CHUNK) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][r] += sum[a][b][r];
    }

    // D[row = (r&3) + 8 (r>>2) + 4 h][col = i] of each 32 x 32 tile
    float* C = g.C + bz * g.sC;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int n = n0 + wn * 64 + b * 32 + i;
            if (n >= g.N) continue;
            const float bv = g.bias ? g.bias[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (m >= g.M) continue;
                float* c = C + (long long)m * g.ldc + n;
                float v = g.alpha * acc[a][b][r] + bv;
                if (g.beta != 0.f) v += g.beta * *c;
                if (g.relu) v = v > 0.f ? v : 0.f;
                *c = v;
            }
        }
    if (!MLOOP) break;
    __syncthreads();                               // (the tile buffers are about to be restaged)
    }
}

// C = epilogue(sum over slices, in slice order): the second half of a split-K product
__global__ void gemm32_reduce_kernel(int slices, int M, int N, const float* __restrict__ part, float* __restrict__ C, long long ldc,
                                     float alpha, float beta, const float* __restrict__ bias, int relu) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)M * N) return;
    const int m = (int)(t / N), n = (int)(t - (size_t)m * N);
    float v = 0.f;
    for (int sidx = 0; sidx < slices; ++sidx) v += part[(size_t)sidx * M * N + t];
    v = alpha * v + (bias ? bias[n] : 0.f);
    float* c = C + (long long)m * ldc + n;
    if (beta != 0.f) v += beta * *c;
    if (relu) v = v > 0.f ? v : 0.f;
    *c = v;
}

// slices a K-heavy product (few output tiles, long K: the weight gradients) is cut into so that the launch fills the chip
int gemm32_auto_slices(int M, int N, int K) {
    const long long tiles = (long long)((M + G_BM - 1) / G_BM) * ((N + G_BN - 1) / G_BN);
    if (tiles >= 256 || K < 4096) return 1;
    long long sl = (512 + tiles - 1) / tiles;
    const long long by_k = K / 1024;                       // at least 64 K-tiles per slice
    if (sl > by_k) sl = by_k;
    if (sl > 256) sl = 256;
    return (int)(sl < 1 ? 1 : sl);
}

int launch_gemm32(hipStream_t s, const Gemm32& g_in) {
    Gemm32 g = g_in;
#ifdef DAGL_ABLATION
    { static const int var = getenv("DAGL_GEMM_VARIANT") ? atoi(getenv("DAGL_GEMM_VARIANT")) : 0; g.variant = var; }
#endif
    if (g.M <= 0 || g.N <= 0 || g.batch <= 0) return DAGL_OK;
    if (g.K <= 0 || !g.A || !g.B || !g.C) { set_error("gemm32: bad argument"); return DAGL_ERR_INVALID; }
    const float alpha = g.alpha, beta = g.beta; const float* bias = g.bias; const int relu = g.relu;
    float* C = g.C; const long long ldc = g.ldc;
    const int slices = (g.slices > 1 && g.batch == 1 && g.scratch != nullptr) ? g.slices : 1;
    g.k_total = 0;
    if (slices > 1) {
        const int kc = ((g.K + slices - 1) / slices + G_BK - 1) / G_BK * G_BK;      // whole K-tiles per slice
        g.k_total = g.K; g.K = kc; g.batch = (g.k_total + kc - 1) / kc;
        g.sA = g.a_kc ? kc : (long long)kc * g.lda;
        g.sB = g.b_kc ? kc : (long long)kc * g.ldb;
        g.C = g.scratch; g.ldc = g.N; g.sC = (long long)g.M * g.N;
        g.alpha = 1.f; g.beta = 0.f; g.bias = nullptr; g.relu = 0;
    }
    const int vecA = ((uintptr_t)g.A % 16 == 0) && (g.lda % 4 == 0) && (g.sA % 4 == 0);
    const int vecB = ((uintptr_t)g.B % 16 == 0) && (g.ldb % 4 == 0) && (g.sB % 4 == 0);
    int tiles_m = (g.M + G_BM - 1) / G_BM;
    const bool mloop = g.m_limit != nullptr && tiles_m > 2 && g.a_kc && g.b_kc && g.chunk_tiles > 0;   // (the flagged rows' score product)
    if (mloop) tiles_m = 2;                                        // blocks walk their row tiles: gemm32_kernel<.., MLOOP>
    const dim3 grid((g.N + G_BN - 1) / G_BN, tiles_m, g.batch), block(256);
#define G32_LAUNCH(A_, B_) do { if (g.chunk_tiles > 0) hipLaunchKernelGGL((gemm32_kernel<A_, B_, true>), grid, block, 0, s, g, vecA, vecB); \
                                else hipLaunchKernelGGL((gemm32_kernel<A_, B_, false>), grid, block, 0, s, g, vecA, vecB); } while (0)
    if (mloop) hipLaunchKernelGGL((gemm32_kernel<true, true, true, true>), grid, block, 0, s, g, vecA, vecB);
    else if (g.a_kc && g.b_kc) G32_LAUNCH(true, true);
    else if (g.a_kc && !g.b_kc) G32_LAUNCH(true, false);
    else if (!g.a_kc && g.b_kc) G32_LAUNCH(false, true);
    else G32_LAUNCH(false, false);
#undef G32_LAUNCH
    DAGL_LAUNCH_CHECK("gemm32_kernel");
    if (slices > 1) {
        const size_t n = (size_t)g.M * g.N;
        hipLaunchKernelGGL(gemm32_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, g.batch, g.M, g.N, g.scratch, C,
                           ldc, alpha, beta, bias, relu);
        DAGL_LAUNCH_CHECK("gemm32_reduce_kernel");
    }
    return DAGL_OK;
}

}  // namespace dagl
