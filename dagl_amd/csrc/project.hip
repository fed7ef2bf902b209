// Patch projection: feat = relu(patch_rows . W^T + b), the two Linear(784->196)+ReLU layers applied
// to every 7x7x16 patch (DN_Gray/model/dagl.py:196-203, called at :248 (queries, fc1) and :249 (keys,
// fc2)) -- computed as an implicit GEMM straight from the zero-bordered NHWC map, so the two
// [N,784] unfold buffers of dagl.py:224-240 never exist.
//
// MFMA: v_mfma_f32_16x16x4_f32 (exact fp32 fma chain).  One wave owns 16 consecutive patches of one
// grid row x all 208 (=13x16, 196 real) output columns: 13 independent accumulators.
//   A (patches): read straight from global/L2: for a fixed (kh,kw) the 16 patches x 16 channels are one
//      contiguous 1-KiB run of the NHWC map (stride-1 grid) -- one coalesced dwordx4 per lane per step.
//   B (weights): the (kh,kw) slice [208 x 16] = 13 KiB is shared by the 4 waves of a block through LDS,
//      double buffered, filled by LDS-DMA (global_load_lds_dwordx4) from a pre-swizzled packed copy.
// K order inside a step is permuted (k-slot g of MFMA t <-> channel 4g+t) so that one 16-byte read per
// lane feeds 4 MFMAs for both operands; a dot product does not care.
#include "dagl_common.h"

namespace dagl {

constexpr int PJ_WAVES = 4;
constexpr int PJ_NT = DPAD / 16;                 // 13 column tiles
constexpr int PJ_STEPS = KS * KS;                // 49 (kh,kw) steps, 16 channels each
constexpr int PJ_SLICE = DPAD * CH;              // floats per weight slice (3328)

// Packed weight layout consumed by project_kernel: wp[step][o][slot'][4] with
// slot' = g ^ (((o>>3)&1)<<1)  (g = channel quad) -- makes the ds_read_b128 of the B fragment conflict-free.
__global__ void pack_fc_weight_kernel2(const float* __restrict__ w, float* __restrict__ wp) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= PJ_STEPS * PJ_SLICE) return;
    const int step = i / PJ_SLICE, r = i % PJ_SLICE;
    const int o = r / CH, sl = (r % CH) / 4, t = r % 4;
    const int g = sl ^ (((o >> 3) & 1) << 1);
    const int c = 4 * g + t;
    wp[i] = (o < D) ? w[(size_t)o * P + c * (KS * KS) + step] : 0.0f;
}

int launch_pack_fc_weight(hipStream_t s, const float* w, float* wp) {
    const int n = PJ_STEPS * PJ_SLICE;
    hipLaunchKernelGGL(pack_fc_weight_kernel2, dim3((n + 255) / 256), dim3(256), 0, s, w, wp);
    DAGL_LAUNCH_CHECK("pack_fc_weight_kernel");
    return DAGL_OK;
}

__device__ __forceinline__ void glds16(const float* gsrc, float* lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

struct ProjArgs {
    Grid gr;
    const float* map;            // padded NHWC key/query feature map
    const float* wp[2];          // packed weights: [0] keys (fc2), [1] queries (fc1)
    const float* bias[2];
    float* feat[2];              // outputs [B, rows_alloc, DS]
    uint16_t* feat_h[2];         // optional bf16 copies [B, rows_alloc_h, DSH] (round to nearest even)
    int rows_alloc[2];
    int rows_alloc_h[2];
    int n_items[2];              // 16-patch work items per image
    int segs[2];                 // work items per grid row
    int n_blocks_q;              // blocks [0, n_blocks_q) project queries, the rest keys
    double* colsum;              // [B, DS] key column sums (may be null)
    RangeTag range;              // a non-finite feature (NaN / inf input upstream) sets the call's word: the fp32 path has no RANGE, but the
                                 // ReLU below would turn a NaN into 0 and the selection drop an inf-poisoned key -- finite output from a
                                 // non-finite input, where dagl.py:207-275 returns NaN
};

// One launch covers both projections: the few query blocks (stride-4 grid, fc1) are dispatched first and
// run concurrently with the key blocks (stride-1 grid, fc2) instead of forming a second, under-filled launch.
// A wave owns PJ_MT = 2 tiles of 16 consecutive patches: every weight fragment read from LDS feeds 8 MFMAs.
constexpr int PJ_MT = 2;
constexpr int PJ_ROWS = 16 * PJ_MT;              // patches per wave

__global__ __launch_bounds__(256, 2) void project_kernel(ProjArgs pa) {
    __shared__ __attribute__((aligned(16))) float sB[2][PJ_SLICE];       // 26 KiB
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int i = lane & 15, g = lane >> 4;
    const int b = blockIdx.y;
    const Grid& gr = pa.gr;

    const bool queries = (int)blockIdx.x < pa.n_blocks_q;                // block-uniform
    const int which = queries ? 1 : 0;
    const int blk = queries ? blockIdx.x : blockIdx.x - pa.n_blocks_q;
    const float* __restrict__ wp = pa.wp[which];
    const int n_items = pa.n_items[which];
    const int segs_per_row = pa.segs[which];

    // work item of this wave: PJ_ROWS consecutive patches of one grid row
    int item = blk * PJ_WAVES + wave;
    const bool wave_valid = item < n_items;
    if (!wave_valid) item = n_items - 1;
    const int row_len = queries ? gr.Lw : gr.W;
    const int gy = item / segs_per_row;                       // grid row (query row r or pixel row y)
    const int gx0 = (item % segs_per_row) * PJ_ROWS;
    const float* abase[PJ_MT];
#pragma unroll
    for (int m = 0; m < PJ_MT; ++m) {
        int gx = gx0 + 16 * m + i;
        if (gx >= row_len) gx = row_len - 1;
        // top-left corner of the patch in padded-map coordinates
        const int py = queries ? (QS * gy - gr.pt + PADPIX) : gy;
        const int px = queries ? (QS * gx - gr.pl + PADPIX) : gx;
        abase[m] = pa.map + (((size_t)b * gr.Hp + py) * gr.Wp + px) * CH + 4 * g;
    }

    // acc: running chain of the current kernel row (7 steps x 16 channels = 112 terms); tot: sum of finished
    // rows.  Chunking the 784-term sum by kernel row cuts its rounding error ~2.5x (see select.hip).
    f32x4 acc[PJ_MT][PJ_NT], tot[PJ_MT][PJ_NT];
#pragma unroll
    for (int m = 0; m < PJ_MT; ++m)
#pragma unroll
        for (int n = 0; n < PJ_NT; ++n) { acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f}; tot[m][n] = acc[m][n]; }

    // B-fragment LDS offsets (floats): row o = n*16 + i, swizzled slot
    const int bslot = g ^ (((i >> 3) & 1) << 1);
    const int boff = i * CH + bslot * 4;

    // prologue: slice 0 -> sB[0]
    const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr_of(&sB[0][0]));
    for (int p = wave; p < PJ_NT; p += PJ_WAVES)
        glds16_asm(wp + (size_t)p * 256 + lane * 4, __builtin_amdgcn_readfirstlane(lds0 + p * 1024));
    float4 a_cur[PJ_MT];
#pragma unroll
    for (int m = 0; m < PJ_MT; ++m) a_cur[m] = *reinterpret_cast<const float4*>(abase[m]);
    dma_wait_all();
    __syncthreads();

    for (int step = 0; step < PJ_STEPS; ++step) {
        const int cur = step & 1;
        float4 a_nxt[PJ_MT];
#pragma unroll
        for (int m = 0; m < PJ_MT; ++m) a_nxt[m] = a_cur[m];
        if (step + 1 < PJ_STEPS) {
            const int ns = step + 1;
            const float* wsrc = wp + (size_t)ns * PJ_SLICE;
            const unsigned dst = lds0 + (cur ^ 1) * (PJ_SLICE * 4);
            for (int p = wave; p < PJ_NT; p += PJ_WAVES)
                glds16_asm(wsrc + (size_t)p * 256 + lane * 4, __builtin_amdgcn_readfirstlane(dst + p * 1024));
            const int kh = ns / KS, kw = ns % KS;
#pragma unroll
            for (int m = 0; m < PJ_MT; ++m)
                a_nxt[m] = *reinterpret_cast<const float4*>(abase[m] + ((size_t)kh * gr.Wp + kw) * CH);
        }
        const float* sb = &sB[cur][boff];
#pragma unroll
        for (int n = 0; n < PJ_NT; ++n) {
            const float4 bw = *reinterpret_cast<const float4*>(sb + n * 16 * CH);
#pragma unroll
            for (int m = 0; m < PJ_MT; ++m) {
                acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[m].x, bw.x, acc[m][n], 0, 0, 0);
                acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[m].y, bw.y, acc[m][n], 0, 0, 0);
                acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[m].z, bw.z, acc[m][n], 0, 0, 0);
                acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[m].w, bw.w, acc[m][n], 0, 0, 0);
            }
        }
        if ((step + 1) % KS == 0) {
#pragma unroll
            for (int m = 0; m < PJ_MT; ++m)
#pragma unroll
                for (int n = 0; n < PJ_NT; ++n) { tot[m][n] += acc[m][n]; acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
        }
#pragma unroll
        for (int m = 0; m < PJ_MT; ++m) a_cur[m] = a_nxt[m];
        dma_wait_all();
        __syncthreads();
    }

    // epilogue: D[row = 4g + r][col = n*16 + i]; bias + ReLU; zero columns 196..203
    float* fb = pa.feat[which] + (size_t)b * pa.rows_alloc[which] * DS;
    uint16_t* hb = pa.feat_h[which] ? pa.feat_h[which] + (size_t)b * pa.rows_alloc_h[which] * DSH : nullptr;
    const float* __restrict__ fbias = pa.bias[which];
    float csum[PJ_NT];
#pragma unroll
    for (int n = 0; n < PJ_NT; ++n) csum[n] = 0.f;
#pragma unroll
    for (int m = 0; m < PJ_MT; ++m) {
        const int grid_row_base = gy * row_len + gx0 + 16 * m;  // linear patch index of row 0 of this tile
#pragma unroll
        for (int n = 0; n < PJ_NT; ++n) {
            const int col = n * 16 + i;
            const float bv = (col < D) ? fbias[col] : 0.0f;
            float s = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int rr = 4 * g + r;
                const bool ok = wave_valid && (gx0 + 16 * m + rr < row_len);
                float v = tot[m][n][r] + bv;
                if (pa.range.word != nullptr && !(fabsf(v) <= 3.4e38f)) *pa.range.word = pa.range.tag;
                v = v > 0.f ? v : 0.f;
                if (col >= D) v = 0.f;
                if (ok && col < DS) fb[(size_t)(grid_row_base + rr) * DS + col] = v;
                if (ok && hb != nullptr) {
                    unsigned u = __float_as_uint(v);
                    u = (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;            // fp32 -> bf16, round to nearest even
                    hb[(size_t)(grid_row_base + rr) * DSH + col] = (uint16_t)u;
                }
                s += ok ? v : 0.f;
            }
            csum[n] += s;
        }
    }
    if (!queries && pa.colsum != nullptr) {
        // reduce over the 4 row groups (lanes i, i+16, i+32, i+48), then one fp64 atomic per column per wave
#pragma unroll
        for (int n = 0; n < PJ_NT; ++n) {
            float s = csum[n];
            s += __shfl_xor(s, 16);
            s += __shfl_xor(s, 32);
            const int col = n * 16 + i;
            if (g == 0 && col < D && wave_valid) atomicAdd(&pa.colsum[(size_t)b * DS + col], (double)s);
        }
    }
}

// which: bit 0 = keys, bit 1 = queries
int launch_project(hipStream_t s, int B, const Grid& g, int which, const float* map, const float* wp_keys,
                   const float* bias_keys, float* feat_keys, double* colsum, const float* wp_q,
                   const float* bias_q, float* feat_q, uint16_t* feat_keys_bf16, uint16_t* feat_q_bf16, RangeTag range) {
    ProjArgs pa;
    pa.range = range;
    pa.feat_h[0] = feat_keys_bf16; pa.feat_h[1] = feat_q_bf16;
    pa.rows_alloc_h[0] = feat_rows_h(g.N); pa.rows_alloc_h[1] = feat_rows_h(g.L);
    pa.gr = g; pa.map = map;
    pa.wp[0] = wp_keys; pa.bias[0] = bias_keys; pa.feat[0] = feat_keys;
    pa.wp[1] = wp_q; pa.bias[1] = bias_q; pa.feat[1] = feat_q;
    pa.rows_alloc[0] = feat_rows(g.N); pa.rows_alloc[1] = feat_rows(g.L);
    pa.segs[0] = (g.W + PJ_ROWS - 1) / PJ_ROWS; pa.segs[1] = (g.Lw + PJ_ROWS - 1) / PJ_ROWS;
    pa.n_items[0] = pa.segs[0] * g.H; pa.n_items[1] = pa.segs[1] * g.Lh;
    pa.colsum = colsum;
    const int nbq = (which & 2) ? (pa.n_items[1] + PJ_WAVES - 1) / PJ_WAVES : 0;
    const int nbk = (which & 1) ? (pa.n_items[0] + PJ_WAVES - 1) / PJ_WAVES : 0;
    pa.n_blocks_q = nbq;
    dim3 grid(nbq + nbk, B), block(256);
    hipLaunchKernelGGL(project_kernel, grid, block, 0, s, pa);
    DAGL_LAUNCH_CHECK("project_kernel");
    return DAGL_OK;
}

// mt[b,l] = (Wq[l,:] . colsum/N) * thr[b,l]   -- the "mean_j S[l,j] * thr" term of dagl.py:256.
// The row mean of S is linear in the key features, so it needs no pass over S.
__global__ void query_thresholds_kernel(int L, int N, int rows_alloc, const float* __restrict__ wq,
                                        const double* __restrict__ colsum, const float* __restrict__ thr,
                                        float* __restrict__ mt, float* __restrict__ mu_out, ThrFuse f) {
    const int b = blockIdx.y;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;     // one wave per query
    const int lane = threadIdx.x & 63;
    if (wave >= L) return;
    const float* q = wq + ((size_t)b * rows_alloc + wave) * DS;
    const double* cs = colsum + (size_t)b * DS;
    double acc = 0.0;
    for (int d = lane; d < D; d += 64) acc += (double)q[d] * cs[d];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) {
        const size_t ql = (size_t)b * L + wave;
        const float mean = (float)(acc / (double)N);
        float tv, bv = 0.f;
        if (f.part != nullptr) {
            // batch entry b = head * imgs + img; partials of a head: [4 groups][imgs][L][2]
            const int head = b / f.imgs_per_head, img = b - head * f.imgs_per_head;
            const size_t n = (size_t)f.imgs_per_head * L;
            const float2* pp = reinterpret_cast<const float2*>(f.part) + (size_t)head * 4 * n + (size_t)img * L + wave;
            const float2 p0 = pp[0], p1 = pp[n], p2 = pp[2 * n], p3 = pp[3 * n];
            tv = ((p0.x + p1.x) + (p2.x + p3.x)) + f.thr_b[head][0];
            bv = ((p0.y + p1.y) + (p2.y + p3.y)) + f.bias_b[head][0];
            f.thr_out[ql] = tv; f.bias_out[ql] = bv;
        } else {
            tv = thr[ql];
            if (f.theta_out != nullptr) bv = f.bias_out[ql];
        }
        const float m = mean * tv;
        mt[ql] = m;
        if (mu_out != nullptr) mu_out[ql] = mean;
        if (f.theta_out != nullptr) f.theta_out[ql] = adaptive_theta_of(m, bv);
        if (f.zero_out != nullptr) f.zero_out[ql] = 0.f;
    }
}

int launch_query_thresholds(hipStream_t s, int B, int L, int N, const float* wq, const double* colsum,
                            const float* thr, float* mt, float* mu_out, const ThrFuse* fuse) {
    dim3 grid((L + 3) / 4, B), block(256);
    const ThrFuse f = fuse ? *fuse : ThrFuse();
    hipLaunchKernelGGL(query_thresholds_kernel, grid, block, 0, s, L, N, feat_rows(L), wq, colsum, thr, mt, mu_out, f);
    DAGL_LAUNCH_CHECK("query_thresholds_kernel");
    return DAGL_OK;
}

}  // namespace dagl
