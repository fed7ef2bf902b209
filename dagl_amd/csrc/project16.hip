// Patch projection on the fp16 matrix cores with split operands -- same result class as the fp32 kernel of
// project.hip at a third of its matrix-core cycles per product and 16x its rate.
//
// Every fp32 operand is split into two fp16 numbers, a = a_hi + 2^-11 a_lo (a_hi = fp16(a), a_lo =
// fp16((a - a_hi) 2^11)): 22 significant bits.  A product keeps its three leading terms
//     a w  ~  a_hi w_hi  +  2^-11 (a_hi w_lo + a_lo w_hi)            (dropped: 2^-22 a_lo w_lo)
// Each fp16 x fp16 product is exact in fp32; v_mfma_f32_32x32x16_f16 sums 16 of them into an fp32 accumulator.
// The hh chain has 49 links (one per kernel tap) instead of the 784 of an fp32 fma chain, so its accumulation
// error is smaller, and the split itself costs 1.2e-5 normwise on the block output (measured against fp64; the
// fp32-input bound is 3e-6) -- well inside the reference's own fp32 noise of 2e-5..6e-5.
//
// Tiling: one wave = 32 consecutive patches x 224 (= 7 x 32, 196 real) outputs: accumulators hh[7], cross[7].
//   A: hi/lo fp16 NHWC maps, one 16-byte load per lane per tap and part (coalesced, L2 resident)
//   B: per tap the [224 outs][hi 16 | lo 16 | pad 8] fp16 slice (18 KiB) is shared by the 4 waves of a block
//      through LDS, double buffered by LDS-DMA; 80-byte rows (5 slots, odd) make the ds_read_b128 conflict-free.
#include "dagl_common.h"

namespace dagl {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

constexpr int P16_WAVES = 4;
constexpr int P16_NT = 7;                        // 32-wide output tiles (224 >= 196)
constexpr int P16_OUT = P16_NT * 32;             // 224
constexpr int P16_ROWH = 40;                     // halfs per output row of a slice: 16 hi + 16 lo + 8 pad
constexpr int P16_SLICE_H = 9216;                // halfs per tap slice: 224*40 = 8960 -> 18 KiB = 18 DMA pieces
constexpr int P16_PIECES = 18;
constexpr int P16_STEPS = KS * KS;               // 49 taps
constexpr float P16_LO_SCALE = 2048.0f;          // 2^11

__device__ __forceinline__ void split_f16(float a, unsigned short& hi, unsigned short& lo) {
    const _Float16 h = (_Float16)a;
    const _Float16 l = (_Float16)((a - (float)h) * P16_LO_SCALE);
    hi = __builtin_bit_cast(unsigned short, h);
    lo = __builtin_bit_cast(unsigned short, l);
}

// fp32 NHWC map -> hi / lo fp16 NHWC maps (same [B,Hp,Wp,16] geometry, 32 B per pixel each)
__global__ void split_map_kernel(size_t n, const float* __restrict__ src, unsigned short* __restrict__ hi,
                                 unsigned short* __restrict__ lo) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;          // float4 index
    if (i * 4 >= n) return;
    const float4 v = reinterpret_cast<const float4*>(src)[i];
    unsigned short h[4], l[4];
    split_f16(v.x, h[0], l[0]); split_f16(v.y, h[1], l[1]); split_f16(v.z, h[2], l[2]); split_f16(v.w, h[3], l[3]);
    reinterpret_cast<ushort4*>(hi)[i] = make_ushort4(h[0], h[1], h[2], h[3]);
    reinterpret_cast<ushort4*>(lo)[i] = make_ushort4(l[0], l[1], l[2], l[3]);
}

int launch_split_map(hipStream_t s, size_t n_floats, const float* src, uint16_t* hi, uint16_t* lo) {
    const size_t n4 = (n_floats + 3) / 4;
    hipLaunchKernelGGL(split_map_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, n_floats, src, hi, lo);
    DAGL_LAUNCH_CHECK("split_map_kernel");
    return DAGL_OK;
}

// fc weight [196,784] (c,kh,kw) -> packed [49 taps][P16_SLICE_H halfs]: row o = 40 halfs: w_hi[c=0..15], w_lo[c=0..15], pad
__global__ void pack_fc_weight16_kernel(const float* __restrict__ w, unsigned short* __restrict__ wp) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P16_STEPS * P16_SLICE_H) return;
    const int tap = i / P16_SLICE_H, r = i % P16_SLICE_H;
    const int o = r / P16_ROWH, e = r % P16_ROWH;
    unsigned short v = 0;
    if (o < D && e < 32) {
        const int c = e & 15;
        unsigned short hi, lo;
        split_f16(w[(size_t)o * P + c * (KS * KS) + tap], hi, lo);
        v = (e < 16) ? hi : lo;
    }
    wp[i] = v;
}

int launch_pack_fc_weight16(hipStream_t s, const float* w, uint16_t* wp) {
    const int n = P16_STEPS * P16_SLICE_H;
    hipLaunchKernelGGL(pack_fc_weight16_kernel, dim3((n + 255) / 256), dim3(256), 0, s, w, wp);
    DAGL_LAUNCH_CHECK("pack_fc_weight16_kernel");
    return DAGL_OK;
}

struct Proj16Args {
    Grid gr;
    const unsigned short* map_hi; const unsigned short* map_lo;     // [B,Hp,Wp,16] fp16
    const unsigned short* wp[2];                                    // packed weights: [0] keys (fc2), [1] queries (fc1)
    const float* bias[2];
    float* feat[2];                                                 // [B, rows_alloc, DS]
    uint16_t* feat_h[2];                                            // optional bf16 copies [B, rows_alloc_h, DSH]
    int rows_alloc[2], rows_alloc_h[2];
    int n_items[2], segs[2];                                        // 32-patch work items per image / per grid row
    int n_blocks_q;
    double* colsum;
};

// NT output tiles starting at tile n0: a block covers 4 x 32 patches x NT x 32 outputs.  The 7 tiles are split
// 4 + 3 over two blocks so that the accumulators (2 x NT x 16 registers) leave room for two blocks per CU.
//
// Pipeline: every operand of tap t+PD is in flight (LDS-DMA) while tap t is multiplied: a ring of P16_RING
// stages, each = the block's weight slice (12 KiB: 12 DMA pieces, 3 per wave) + 4 wave-private 2-KiB patch
// fragments (hi, lo: each lane DMA-copies exactly the 16 bytes it will read back).  All memory operations of the
// loop are asm LDS-DMAs, 5 per wave per tap, so the landing of tap t+1 is a counted s_waitcnt vmcnt(5*(PD-1)).
constexpr int P16_RING = 4;
constexpr int P16_PD = 3;                              // prefetch distance (taps)
constexpr int P16_STAGE_B = 12 * 1024;                 // bytes: weight slice region (>= NT*32*80)
constexpr int P16_STAGE = P16_STAGE_B + P16_WAVES * 2048;      // + per-wave patch fragments = 20 KiB

template <int N>
__device__ __forceinline__ void dma_wait_le() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int NT>
__device__ __forceinline__ void project16_body(const Proj16Args& pa, unsigned char* smem, int n0, int blk, bool queries) {
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int i = lane & 31, h = lane >> 5;
    const int b = blockIdx.y;
    const Grid& gr = pa.gr;
    const int which = queries ? 1 : 0;
    const unsigned short* __restrict__ wp = pa.wp[which] + (size_t)n0 * 32 * P16_ROWH;
    const int n_items = pa.n_items[which];
    const int segs_per_row = pa.segs[which];

    int item = blk * P16_WAVES + wave;
    const bool wave_valid = item < n_items;
    if (!wave_valid) item = n_items - 1;
    const int row_len = queries ? gr.Lw : gr.W;
    const int gy = item / segs_per_row;
    const int gx0 = (item % segs_per_row) * 32;
    int gx = gx0 + i;
    if (gx >= row_len) gx = row_len - 1;
    const int py = queries ? (QS * gy - gr.pt + PADPIX) : gy;
    const int px = queries ? (QS * gx - gr.pl + PADPIX) : gx;
    const size_t aoff = (((size_t)b * gr.Hp + py) * gr.Wp + px) * CH + 8 * h;      // halfs
    const unsigned short* ahi = pa.map_hi + aoff;
    const unsigned short* alo = pa.map_lo + aoff;

    f32x16 hh[NT], cx[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) { hh[n][r] = 0.f; cx[n][r] = 0.f; }

    const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr_of(smem));
    // issue the 5 DMAs of tap `t` (weights: pieces wave, wave+4, wave+8; patches: hi, lo) into ring stage t % RING
    auto issue = [&](int t) {
        const unsigned st = lds0 + (unsigned)(t % P16_RING) * P16_STAGE;
        const unsigned short* wsrc = wp + (size_t)t * P16_SLICE_H;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int p = wave + 4 * j;
            glds16_asm(reinterpret_cast<const float*>(wsrc + (size_t)p * 512 + lane * 8),
                       __builtin_amdgcn_readfirstlane(st + p * 1024));
        }
        const int kh = t / KS, kw = t - kh * KS;
        const size_t o = ((size_t)kh * gr.Wp + kw) * CH;
        const unsigned sa = st + P16_STAGE_B + wave * 2048;
        glds16_asm(reinterpret_cast<const float*>(ahi + o), __builtin_amdgcn_readfirstlane(sa));
        glds16_asm(reinterpret_cast<const float*>(alo + o), __builtin_amdgcn_readfirstlane(sa + 1024));
    };

#pragma unroll
    for (int t = 0; t < P16_PD; ++t) issue(t);
    dma_wait_le<5 * (P16_PD - 1)>();
    __syncthreads();

    // B fragment of local tile n: lane (j = i, h) reads row n*32 + i: hi at +8h, lo at +16+8h (halfs)
    const int boff = (i * P16_ROWH + 8 * h) * 2;                       // bytes
    for (int step = 0; step < P16_STEPS; ++step) {
        if (step + P16_PD < P16_STEPS) issue(step + P16_PD);
        const unsigned char* st = smem + (step % P16_RING) * P16_STAGE;
        const unsigned char* sa = st + P16_STAGE_B + wave * 2048 + lane * 16;
        const f16x8 fa_hi = __builtin_bit_cast(f16x8, *reinterpret_cast<const s16x8*>(sa));
        const f16x8 fa_lo = __builtin_bit_cast(f16x8, *reinterpret_cast<const s16x8*>(sa + 1024));
        const unsigned char* sb = st + boff;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const f16x8 w_hi = __builtin_bit_cast(f16x8, *reinterpret_cast<const s16x8*>(sb + n * 32 * P16_ROWH * 2));
            const f16x8 w_lo = __builtin_bit_cast(f16x8, *reinterpret_cast<const s16x8*>(sb + n * 32 * P16_ROWH * 2 + 32));
            hh[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa_hi, w_hi, hh[n], 0, 0, 0);
            cx[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa_hi, w_lo, cx[n], 0, 0, 0);
            cx[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa_lo, w_hi, cx[n], 0, 0, 0);
        }
        // tap step+1 must have landed (for every wave) before anyone reads it
        const int left = P16_STEPS - 1 - step;                          // taps still to compute after this one
        if (left >= P16_PD) dma_wait_le<5 * (P16_PD - 1)>();
        else if (left == 2) dma_wait_le<5>();
        else dma_wait_le<0>();
        __syncthreads();
    }

    // epilogue: D[row = patch (r&3)+8(r>>2)+4h][col = output (n0+n)*32 + i]
    float* fb = pa.feat[which] + (size_t)b * pa.rows_alloc[which] * DS;
    uint16_t* hb = pa.feat_h[which] ? pa.feat_h[which] + (size_t)b * pa.rows_alloc_h[which] * DSH : nullptr;
    const float* __restrict__ fbias = pa.bias[which];
    const int grid_row_base = gy * row_len + gx0;
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int col = (n0 + n) * 32 + i;
        const float bv = (col < D) ? fbias[col] : 0.0f;
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rr = (r & 3) + 8 * (r >> 2) + 4 * h;
            const bool ok = wave_valid && (gx0 + rr < row_len);
            float v = (hh[n][r] + cx[n][r] * (1.0f / P16_LO_SCALE)) + bv;
            v = v > 0.f ? v : 0.f;
            if (col >= D) v = 0.f;
            if (ok && col < DS) fb[(size_t)(grid_row_base + rr) * DS + col] = v;
            if (ok && hb != nullptr && col < DPAD) {
                unsigned u = __float_as_uint(v);
                u = (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;            // fp32 -> bf16, round to nearest even
                hb[(size_t)(grid_row_base + rr) * DSH + col] = (uint16_t)u;
            }
            s += ok ? v : 0.f;
        }
        if (!queries && pa.colsum != nullptr) {
            s += __shfl_xor(s, 32);                                   // the two row halves of the tile
            if (h == 0 && col < D && wave_valid) atomicAdd(&pa.colsum[(size_t)b * DS + col], (double)s);
        }
    }
}

__global__ __launch_bounds__(256, 2) void project16_kernel(Proj16Args pa) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[P16_RING * P16_STAGE];      // 80 KiB
    // blocks: [query blocks x 2 halves][key blocks x 2 halves]; half 0 = output tiles 0..3, half 1 = tiles 4..6
    const int bid = blockIdx.x;
    const bool queries = bid < 2 * pa.n_blocks_q;                       // block-uniform
    const int rel = queries ? bid : bid - 2 * pa.n_blocks_q;
    const int half = rel & 1, blk = rel >> 1;
    if (half == 0) project16_body<4>(pa, smem, 0, blk, queries);
    else project16_body<3>(pa, smem, 4, blk, queries);
}

int launch_project16(hipStream_t s, int B, const Grid& g, int which, const uint16_t* map_hi, const uint16_t* map_lo,
                     const uint16_t* wp_keys, const float* bias_keys, float* feat_keys, double* colsum,
                     const uint16_t* wp_q, const float* bias_q, float* feat_q, uint16_t* feat_keys_bf16,
                     uint16_t* feat_q_bf16) {
    Proj16Args pa;
    pa.gr = g; pa.map_hi = map_hi; pa.map_lo = map_lo;
    pa.wp[0] = wp_keys; pa.bias[0] = bias_keys; pa.feat[0] = feat_keys; pa.feat_h[0] = feat_keys_bf16;
    pa.wp[1] = wp_q; pa.bias[1] = bias_q; pa.feat[1] = feat_q; pa.feat_h[1] = feat_q_bf16;
    pa.rows_alloc[0] = feat_rows(g.N); pa.rows_alloc[1] = feat_rows(g.L);
    pa.rows_alloc_h[0] = feat_rows_h(g.N); pa.rows_alloc_h[1] = feat_rows_h(g.L);
    pa.segs[0] = (g.W + 31) / 32; pa.segs[1] = (g.Lw + 31) / 32;
    pa.n_items[0] = pa.segs[0] * g.H; pa.n_items[1] = pa.segs[1] * g.Lh;
    pa.colsum = colsum;
    const int nbq = (which & 2) ? (pa.n_items[1] + P16_WAVES - 1) / P16_WAVES : 0;
    const int nbk = (which & 1) ? (pa.n_items[0] + P16_WAVES - 1) / P16_WAVES : 0;
    pa.n_blocks_q = nbq;
    hipLaunchKernelGGL(project16_kernel, dim3(2 * (nbq + nbk), B), dim3(256), 0, s, pa);
    DAGL_LAUNCH_CHECK("project16_kernel");
    return DAGL_OK;
}

}  // namespace dagl
