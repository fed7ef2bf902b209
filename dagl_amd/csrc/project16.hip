// Patch projection on the fp16 matrix cores with split operands -- same result class as the fp32 kernel of
// project.hip at a fifth of its matrix-core cycles.
//
// Every fp32 operand is split into two fp16 numbers after an exact power-of-two pre-scaling that keeps the low part a
// normal fp16 number for ordinary magnitudes (16 a = a_hi + a_lo for activations, 1024 w = w_hi + w_lo for weights; the
// matrix cores keep fp16 denormals, so smaller values only lose relative, not absolute, precision): >= 21 significant
// bits.  A product keeps its three leading terms
//     a w  ~  a_hi w_hi + a_hi w_lo + a_lo w_hi                       (dropped: a_lo w_lo, 2^-22 relative)
// each an exact fp32 number inside v_mfma_f32_32x32x16_f16, all three accumulated into ONE fp32 accumulator (147 links
// per output instead of the 784 of an fp32 fma chain); the result is scaled back by 2^-14.  Measured cost of the split on
// the block output: 1.2e-5 normwise against fp64 -- inside the reference's own fp32 noise of 2e-5..6e-5.
// Range: |16 a| and |1024 w| must stay below 65504 (fp16); the exact scan (project.hip) has no such limit.
//
// Tiling (round 4; project16_body2): a work unit = 8 items of 32 consecutive patches (row-major order) = 256 patches, done by TWO
// blocks of 4 waves -- one per tile group (output tiles 0-3 / 4-6 of the 7 x 32 = 224 >= 196 outputs) --, a wave = 2 items x the
// group's tiles (8 or 6 accumulators): a weight fragment read from the LDS feeds two multiply chains, and a block fetches only its
// group's rows of a tap's weight slice.  (Round 3: a wave = 32 patches x all 7 tiles; per SIMD and tap 6.5 LDS-DMA pieces and 32
// fragment reads beside 42 multiplies; now 3.5 and 22.)  The overhang of the grid over one resident round is cut into single-tile
// blocks of 4 items (project16_body<1>).
//   A (patches): keys -- per item a 2-row ring of the hi / lo map rows its patches touch (38 pixels), by LDS-DMA: each
//                input pixel is fetched once per kernel ROW, not once per tap; queries (stride-4 grid) -- per tap each lane
//                DMA-copies the 16 bytes it reads back
//   B (weights): per tap the group's rows of the [224 outs][hi 16 | lo 16] fp16 slice (8 / 6 KiB) are shared by the 4 waves of a
//                block through a 4-stage LDS ring filled by LDS-DMA; the four 16-byte slots of a 64-byte row are stored at
//                slot ^ ((row >> 2) & 3), which makes the ds_read_b128 of 32 consecutive rows conflict-free unpadded
#include <stdlib.h>
#include <string.h>

#include "dagl_common.h"
#include "thr_bias4.h"

namespace dagl {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

constexpr int P16_NT = 7;                        // 32-wide output tiles (224 >= 196)
constexpr int P16_OUT = P16_NT * 32;             // 224
constexpr int P16_ROWH = 32;                     // halfs per output row of a slice: 16 hi + 16 lo (four 16-byte slots, swizzled)
constexpr int P16_SLICE_H = 7168;                // halfs per tap slice: 224*32 -> 14 KiB = 14 DMA pieces
constexpr int P16_STEPS = KS * KS;               // 49 taps
constexpr float P16_A_SCALE = 16.0f;              // 2^4: activations are split as 16 a = hi + lo
constexpr float P16_W_SCALE = 1024.0f;            // 2^10: weights as 1024 w = hi + lo

__device__ __forceinline__ void split_f16(float a, unsigned short& hi, unsigned short& lo) {
    const _Float16 h = (_Float16)a;
    const _Float16 l = (_Float16)(a - (float)h);
    hi = __builtin_bit_cast(unsigned short, h);
    lo = __builtin_bit_cast(unsigned short, l);
}

// fp32 NHWC map -> hi / lo fp16 NHWC maps (same [B,Hp,Wp,16] geometry, 32 B per pixel each)
__global__ void split_map_kernel(size_t n, const float* __restrict__ src, unsigned short* __restrict__ hi,
                                 unsigned short* __restrict__ lo, RangeTag range) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;          // float4 index
    if (i * 4 >= n) return;
    const float4 v = reinterpret_cast<const float4*>(src)[i];
    const float am = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))) * P16_A_SCALE;
    if (range.word != nullptr && !(am < RANGE_LIMIT)) *range.word = range.tag;
    unsigned short h[4], l[4];
    split_f16(v.x * P16_A_SCALE, h[0], l[0]); split_f16(v.y * P16_A_SCALE, h[1], l[1]);
    split_f16(v.z * P16_A_SCALE, h[2], l[2]); split_f16(v.w * P16_A_SCALE, h[3], l[3]);
    reinterpret_cast<ushort4*>(hi)[i] = make_ushort4(h[0], h[1], h[2], h[3]);
    reinterpret_cast<ushort4*>(lo)[i] = make_ushort4(l[0], l[1], l[2], l[3]);
}

// [B, feat_rows(n), DS] feature rows -> dense [B, n, 196] (training path: dagl_project_patches16); a call that left the
// split-fp16 range hands out NaN, never numbers computed from inf halves
__global__ __launch_bounds__(256) void feat_rows_out_kernel(int n, int rows_alloc, const float* __restrict__ feat,
                                                            float* __restrict__ out, RangeTag range) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;              // one float4 = 4 of a row's 196 columns
    const int b = blockIdx.y;
    const bool poisoned = range.word != nullptr && *range.word == range.tag;
    if (range.done != nullptr && b == 0 && t == 0) *range.done = range.tag;
    if (t >= (size_t)n * (D / 4)) return;
    const size_t row = t / (D / 4); const int c4 = (int)(t - row * (D / 4));
    float4 v = *reinterpret_cast<const float4*>(feat + ((size_t)b * rows_alloc + row) * DS + 4 * c4);
    if (poisoned) { const float q = __builtin_nanf(""); v = make_float4(q, q, q, q); }
    *reinterpret_cast<float4*>(out + ((size_t)b * n + row) * D + 4 * c4) = v;
}

int launch_feat_rows_out(hipStream_t s, int B, int n, const float* feat, float* rows_out, RangeTag range) {
    const size_t items = (size_t)n * (D / 4);
    hipLaunchKernelGGL(feat_rows_out_kernel, dim3((unsigned)((items + 255) / 256), B), dim3(256), 0, s, n, feat_rows(n), feat,
                       rows_out, range);
    DAGL_LAUNCH_CHECK("feat_rows_out_kernel");
    return DAGL_OK;
}

int launch_split_map(hipStream_t s, size_t n_floats, const float* src, uint16_t* hi, uint16_t* lo, RangeTag range) {
    const size_t n4 = (n_floats + 3) / 4;
    hipLaunchKernelGGL(split_map_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, n_floats, src, hi, lo, range);
    DAGL_LAUNCH_CHECK("split_map_kernel");
    return DAGL_OK;
}

// fc weight [196,784] (c,kh,kw) -> packed [49 taps][P16_SLICE_H halfs]: row o = 64 bytes = four 16-byte slots
// (hi c0-7, hi c8-15, lo c0-7, lo c8-15) stored at slot ^ ((o >> 2) & 3): the ds_read_b128 of 32 consecutive rows is then
// conflict-free without padding (rows r, r+4, r+8, r+12 of a 16-lane LDS group land in different slots)
// (rows_order: the weight comes as [196][49 taps][16 c] -- the differentiable path's layout -- instead of Linear's [196][c][tap])
__global__ void pack_fc_weight16_kernel(const float* __restrict__ w, unsigned short* __restrict__ wp, int rows_order) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P16_STEPS * P16_SLICE_H) return;
    const int tap = i / P16_SLICE_H, r = i % P16_SLICE_H;
    const int o = r / P16_ROWH, e = r % P16_ROWH;
    unsigned short v = 0;
    if (o < D) {
        const int slot = (e >> 3) ^ ((o >> 2) & 3);                  // logical slot stored at this physical position
        const int c = (slot & 1) * 8 + (e & 7);
        unsigned short hi, lo;
        const float wv = w[(size_t)o * P + (rows_order ? tap * CH + c : c * (KS * KS) + tap)] * P16_W_SCALE;
        split_f16(wv, hi, lo);
        v = (slot < 2) ? hi : lo;
        // range flag of these packed weights: the last 4 bytes of the buffer (cleared by launch_pack_fc_weight16)
        if (!(fabsf(wv) < RANGE_LIMIT)) *reinterpret_cast<int32_t*>(wp + P16_PACKED_HALFS - 2) = 1;
    }
    wp[i] = v;
}

int launch_pack_fc_weight16(hipStream_t s, const float* w, uint16_t* wp, bool rows_order) {
    const int n = P16_STEPS * P16_SLICE_H;
    DAGL_HIP_TRY(hipMemsetAsync(wp + P16_PACKED_HALFS - 2, 0, 4, s));
    hipLaunchKernelGGL(pack_fc_weight16_kernel, dim3((n + 255) / 256), dim3(256), 0, s, w, wp, rows_order ? 1 : 0);
    DAGL_LAUNCH_CHECK("pack_fc_weight16_kernel");
    return DAGL_OK;
}

// two features -> packed fp16 pairs of the split copy: DN_FS x = hi + lo (x >= 0: after the ReLU)
__device__ __forceinline__ void p16_split_pair(float a, float b, unsigned& hi, unsigned& lo, bool& bad) {
    const float v0 = a * DN_FS, v1 = b * DN_FS;
    bad |= !(v0 < RANGE_LIMIT) | !(v1 < RANGE_LIMIT);
    const _Float16 h0 = (_Float16)v0, h1 = (_Float16)v1;
    const _Float16 l0 = (_Float16)(v0 - (float)h0), l1 = (_Float16)(v1 - (float)h1);
    hi = (unsigned)__builtin_bit_cast(unsigned short, h0) | ((unsigned)__builtin_bit_cast(unsigned short, h1) << 16);
    lo = (unsigned)__builtin_bit_cast(unsigned short, l0) | ((unsigned)__builtin_bit_cast(unsigned short, l1) << 16);
}

struct Proj16Args {
    Grid gr;
    const unsigned short* map_hi; const unsigned short* map_lo;     // [B,Hp,Wp,16] fp16: 16 b1 = hi + lo
    const unsigned short* map_hi2; const unsigned short* map_lo2;   // the coarse tier 2^-8 b1 = hi + lo, or null (B1Tiers, dagl_common.h)
    const float* b1_amax; int amax_slots;                           // [heads][slots] largest |b1| of the conv blocks, or null: fine tier
    const unsigned short* wp[2];                                    // packed weights: [0] keys (fc2), [1] queries (fc1);
                                                                    // head h at + h * P16_PACKED_HALFS
    const float* bias[2][4];                                        // fc bias per head
    int imgs_per_head;                                              // batch index = head * imgs_per_head + image
    float* feat[2];                                                 // [B, rows_alloc, DS]
    uint16_t* feat_h[2];                                            // optional bf16 copies [B, rows_alloc_h, DSH]
    int tiled_h[2];                                                 // bf16 copy in the screen's FRAGMENT order instead of row-major (queries):
                                                                    // per 32 rows [rows 0..15 | 16..31][t 0..12][half 0..1][row][8 columns
                                                                    // 16t + 8half ..], 2 x 6.5 KiB at the tile's row-major place
                                                                    // (ScreenArgs::q_tiled); an epilogue pass = one contiguous half
    unsigned short* split_hi[2]; unsigned short* split_lo[2];       // optional split-fp16 copies [B, rows_alloc_s, DSH] (Split16Out)
    int rows_alloc_s[2];
    int rows_alloc[2], rows_alloc_h[2];
    int n_items[2], segs[2];                                        // 32-patch work items per image / per grid row
    int lin[2];                                                     // items = 32 consecutive patches in ROW-MAJOR order (across row ends)
                                                                    // instead of 32 patches of one row: no idle slots at the row ends
                                                                    // (a 72-pixel row used to cost 3 items = 96 slots)
    int n_blocks_q, n_blocks_k;                                     // 4-item blocks (the rows of colpart; the single-tile blocks)
    int units_q, units_k;                                           // 8-item units (two blocks each: tile groups 0-3 / 4-6)
    int n_full, n_split_groups, batch;                              // 1-D grid: full blocks, then 7 single-tile blocks per split group
    float* colpart;                                                 // [B, n_blocks_k, 224] per-block key column sums (or null)
    RangeTag range; int heads;                                      // range guard: the packed weights' flags feed the call's word
    unsigned long long* times;                                      // ablation builds: block phase stamps (debug.hip) or null
    // the thr / bias heads' blocks (thr_bias4.h) behind the projection's own: n_proj = blocks of the projection, then thr_x x thr_y x TB_GROUPS
    ThrHeadSet thr_hs; float* thr_part; int n_proj, thr_x, thr_y;
};

// Block = 4 waves (two blocks per CU, independent barriers).  All operands arrive by LDS-DMA issued from inline asm and are
// consumed behind COUNTED s_waitcnt vmcnt(N): the weight slice of tap t+PD is requested while tap t is multiplied.
#ifndef DAGL_P16_BW
#define DAGL_P16_BW 4
#endif
constexpr int P16_BW = DAGL_P16_BW;                    // waves per block (two blocks per CU: independent barriers)
constexpr int P16_BLOCKS_PER_CU = (P16_BW > 4) ? 1 : 2;
constexpr int P16_RING = 4;                            // weight stages
constexpr int P16_PD_KEYS = 3;                         // prefetch distance (taps): key blocks
constexpr int P16_PD_Q = 2;                            // query blocks (their per-tap patch stages leave room for 3 only)
constexpr int P16_QRING = 3;
constexpr int P16_STAGE_B = 14 * 1024;                 // bytes per weight stage (= 224*64, whole DMA pieces)
constexpr int P16_OFF_A = P16_RING * P16_STAGE_B;      // 56 KiB: patch region
constexpr int P16_APX = 44;                            // keys: staged pixels per map row: 32 + 6 of a row segment, + 6 for the wrap (below)
constexpr int P16_APART = P16_APX * 32;                // keys: one part (hi or lo) of a staged map row: 44 px x 32 B
constexpr int P16_AROW = 2 * P16_APART;                // keys: one staged map row per wave: hi | lo
constexpr int P16_LDS = P16_OFF_A + P16_QRING * P16_BW * 2048;      // 80 KiB (keys use 56 + 4 x 2 x 2.5 = 76)
static_assert(P16_BW * 2 * P16_AROW <= P16_QRING * P16_BW * 2048, "key row rings must fit the patch region");
static_assert(P16_BLOCKS_PER_CU * P16_LDS <= 160 * 1024, "resident blocks per CU");


// Which tier of the key / query map holds this head's values (B1Tiers, dagl_common.h).  Every wave reduces its head's slots itself, but
// not before it needs to: the slots are REQUESTED first thing (inline asm: up to four 16-byte loads per lane = 1024 slots, the oldest
// entries of the wave's memory queue, so the counted wait in front of the block's first barrier covers them), the block starts on the
// fine tier like every call up to round 5, and the verdict is formed behind that barrier -- no round trip in front of the first
// LDS-DMA.  A head on the coarse tier (|b1| >= 3750) requests its first patch rows / taps again from there: one round trip, once.
typedef float p16f4 __attribute__((ext_vector_type(4)));
struct P16Tier { const unsigned short* hi; const unsigned short* lo; float unscale; };
struct P16TierReq { p16f4 v[4]; };
__device__ __forceinline__ void p16_tier_request(const Proj16Args& pa, int head, int lane, P16TierReq& rq) {
    if (pa.b1_amax == nullptr) return;
    const float* sl = pa.b1_amax + (size_t)head * pa.amax_slots;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int base = 4 * (lane + 64 * j);
        if (base > pa.amax_slots - 4) base = pa.amax_slots - 4;          // (slots: a multiple of 4; a slot read twice changes no maximum)
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(rq.v[j]) : "v"(sl + base) : "memory");
    }
}
// (call it behind a wait that covers the requests -- dma_wait_le of the block's prologue -- and a barrier; block-uniform result)
__device__ __forceinline__ bool p16_tier_is_coarse(const Proj16Args& pa, int head, int lane, P16TierReq& rq) {
    if (pa.b1_amax == nullptr) return false;
    asm volatile("" : "+v"(rq.v[0]), "+v"(rq.v[1]), "+v"(rq.v[2]), "+v"(rq.v[3]));      // (ordered behind the wait: both are volatile)
    float m = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) m = fmaxf(fmaxf(m, fmaxf(rq.v[j][0], rq.v[j][1])), fmaxf(rq.v[j][2], rq.v[j][3]));
    if (pa.amax_slots > 1024) {                                  // (huge batches: the rest in the ordinary way)
        const float* sl = pa.b1_amax + (size_t)head * pa.amax_slots;
        for (int i = 1024 + lane; i < pa.amax_slots; i += 64) m = fmaxf(m, sl[i]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    return !(m * B1_FINE_SCALE < RANGE_LIMIT);
}

template <int NT, bool KEYS, int VAR>
__device__ __forceinline__ void project16_body(const Proj16Args& pa, unsigned char* smem, int n0, int blk, int b) {
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int i = lane & 31, h = lane >> 5;
    const Grid& gr = pa.gr;
    constexpr int which = KEYS ? 0 : 1;
    const int head = b / pa.imgs_per_head;
    P16TierReq tier_rq;
    p16_tier_request(pa, head, lane, tier_rq);
    P16Tier tier = {pa.map_hi, pa.map_lo, 1.0f / (P16_A_SCALE * P16_W_SCALE)};
    const unsigned short* __restrict__ wp = pa.wp[which] + (size_t)head * P16_PACKED_HALFS + (size_t)n0 * 32 * P16_ROWH;
    const int n_items = pa.n_items[which];
    const int segs_per_row = pa.segs[which];
    constexpr int PD = KEYS ? P16_PD_KEYS : P16_PD_Q;
    // the 14th KiB of a tap's slice holds output rows 208..223: padding whose products are never stored (feature rows end at
    // column 203, the bf16 copies at 207; an output column depends on its own weight row only) -- not fetched
    constexpr int PIECES = (NT == P16_NT) ? 13 : (NT * 32 * P16_ROWH * 2 + 1023) / 1024;   // 13 (NT=7) / 2 (NT=1)
    constexpr int PBASE = PIECES / P16_BW;                                      // weight pieces per wave per tap ...
    const bool extra = wave < (PIECES % P16_BW);                                // ... plus one for the first waves

    int item = blk * P16_BW + wave;
    const bool wave_valid = item < n_items;
    if (!wave_valid) item = n_items - 1;
    const int row_len = KEYS ? gr.W : gr.Lw;
    // the item's patches: lin -> n = 32 item + i in row-major order: nA of them in row gy from pixel gx0 on, the rest at the
    // start of row gy + 1 (rows are at least 32 wide in this mode); else 32 patches of row gy from gx0 on
    const bool lin = pa.lin[which] != 0;
    int gy, gx0, nA, base_row, lim;
    if (lin) {
        base_row = item * 32;
        gy = base_row / row_len; gx0 = base_row - gy * row_len;
        nA = row_len - gx0 < 32 ? row_len - gx0 : 32;
        lim = (KEYS ? gr.N : gr.L) - base_row;
    } else {
        gy = item / segs_per_row; gx0 = (item % segs_per_row) * 32;
        nA = 32; base_row = gy * row_len + gx0; lim = row_len - gx0;
    }
    const int off_i = (i < nA) ? i : i + 6;            // keys: the patch's pixel position in the staged row (segment B behind A's halo)
    const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr_of(smem));

    f32x16 hh[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) hh[n][r] = 0.f;

    auto issue_w = [&](int t) {                        // weight slice of tap t -> ring stage t % RING
        if (VAR == 6) return;
        if (VAR == 11 && (t & 1)) return;
        const unsigned st = lds0 + (unsigned)(t % P16_RING) * P16_STAGE_B;
        const unsigned short* wsrc = wp + (size_t)t * P16_SLICE_H;
#pragma unroll
        for (int j = 0; j < PBASE; ++j) {
            const int p = wave + P16_BW * j;
            glds16_asm(reinterpret_cast<const float*>(wsrc + (size_t)p * 512 + lane * 8),
                       __builtin_amdgcn_readfirstlane(st + p * 1024));
        }
        if (extra) {
            const int p = wave + P16_BW * PBASE;
            glds16_asm(reinterpret_cast<const float*>(wsrc + (size_t)p * 512 + lane * 8),
                       __builtin_amdgcn_readfirstlane(st + p * 1024));
        }
    };

    // ---- patch operand plumbing -------------------------------------------------------------------------
    const unsigned short* ahi = nullptr; const unsigned short* alo = nullptr;   // queries: this lane's patch corner
    size_t krow0 = 0;                                                            // keys: halfs offset of (row py, pixel gx0)
    if (KEYS) {
        krow0 = (size_t)b * gr.Hp * gr.Wp * CH;
    } else {
        int qy = gy, gx = gx0 + i;
        if (lin) { int q = base_row + i; if (q > gr.L - 1) q = gr.L - 1; qy = q / row_len; gx = q - qy * row_len; }
        else if (gx >= row_len) gx = row_len - 1;
        const int py = QS * qy - gr.pt + PADPIX, px = QS * gx - gr.pl + PADPIX;
        const size_t aoff = (((size_t)b * gr.Hp + py) * gr.Wp + px) * CH + 8 * h;
        ahi = tier.hi + aoff; alo = tier.lo + aoff;
    }
    auto issue_row = [&](int r) {                      // keys: kernel row r of the item (44 pixels, hi | lo) -> row buffer r & 1
        if (VAR == 6) return;                          // positions 0 .. nA+5: map row gy + r from pixel gx0; from nA+6 on: row gy+1+r from pixel 0
        const unsigned dst = lds0 + P16_OFF_A + wave * (2 * P16_AROW) + (r & 1) * P16_AROW;
#pragma unroll
        for (int j = 0; j < 4; ++j) {                  // pieces: hi px 0-31, hi px 32-43 (24 lanes), lo px 0-31, lo px 32-43
            const int p = (j & 1) * 32 + (lane >> 1);
            int row = gy + r, px = gx0 + p;
            if (p >= nA + 6) { row += 1; px = p - (nA + 6); }
            if (px > gr.Wp - 1) px = gr.Wp - 1;                                   // stay inside the map
            if (row > gr.Hp - 1) row = gr.Hp - 1;
            const unsigned short* src = ((j < 2) ? tier.hi : tier.lo) + krow0 + ((size_t)row * gr.Wp + px) * CH + 8 * (lane & 1);
            const unsigned d = dst + (j >> 1) * P16_APART + (j & 1) * 1024;
            if ((j & 1) == 0 || lane < 2 * (P16_APX - 32)) glds16_asm(reinterpret_cast<const float*>(src), __builtin_amdgcn_readfirstlane(d));
        }
    };
    auto issue_q = [&](int t) {                        // queries: the 16 B of tap t this lane will read back
        const int kh = t / KS, kw = t - kh * KS;
        const size_t o = ((size_t)kh * gr.Wp + kw) * CH;
        const unsigned sa = lds0 + P16_OFF_A + (unsigned)(t % P16_QRING) * (P16_BW * 2048) + wave * 2048;
        glds16_asm(reinterpret_cast<const float*>(ahi + o), __builtin_amdgcn_readfirstlane(sa));
        glds16_asm(reinterpret_cast<const float*>(alo + o), __builtin_amdgcn_readfirstlane(sa + 1024));
    };
    // landing of tap t+1: its weight pieces (and everything issued before them) are complete once at most the DMAs
    // issued after them are outstanding.  Patch-row pieces issued in between only make the wait stricter.
    constexpr int PER_Q = KEYS ? 0 : 2;
#define P16_WAIT(P) do { if (VAR == 4 || VAR == 6 || VAR == 11) break; if (extra) dma_wait_le<(P) * (PBASE + 1 + PER_Q)>(); else dma_wait_le<(P) * (PBASE + PER_Q)>(); } while (0)

    const int swz = (i >> 2) & 3;                                       // slot swizzle of row n*32 + i (n*32 does not change it)
    const int boff_hi = i * (P16_ROWH * 2) + ((h ^ swz) << 4);          // bytes: B fragment row n*32 + i, hi half h
    const int boff_lo = i * (P16_ROWH * 2) + (((2 + h) ^ swz) << 4);
    if (NT == 1 && KEYS) {
        // single-tile blocks (the split remainder of the grid) have 3 MFMAs per tap: a barrier per tap would leave them
        // latency-bound, so a ring stage holds the 7 taps of one kernel row (7 x 2 KiB) and there is one barrier per row
        auto issue_wrow = [&](int kh) {
            const unsigned st = lds0 + (unsigned)(kh % P16_RING) * P16_STAGE_B;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int p = wave + P16_BW * j;                      // 14 pieces: tap kw = p >> 1, half (p & 1) of its 2 KiB
                if (p < 2 * KS)
                    glds16_asm(reinterpret_cast<const float*>(wp + (size_t)(kh * KS + (p >> 1)) * P16_SLICE_H + (p & 1) * 512 + lane * 8),
                               __builtin_amdgcn_readfirstlane(st + p * 1024));
            }
        };
        const bool four = wave < 2;                                    // waves 0, 1 carry 4 weight pieces per row, waves 2, 3 carry 3
        f32x16 c1, c2;
#pragma unroll
        for (int r = 0; r < 16; ++r) { c1[r] = 0.f; c2[r] = 0.f; }
        issue_row(0); issue_wrow(0); issue_wrow(1);
        if (four) dma_wait_le<4>(); else dma_wait_le<3>();             // row 0 and weight row 0 landed
        __syncthreads();
        if (p16_tier_is_coarse(pa, head, lane, tier_rq)) {             // (block-uniform) the map lives in the coarse tier: row 0 again
            tier.hi = pa.map_hi2; tier.lo = pa.map_lo2; tier.unscale = 1.0f / (B1_COARSE_SCALE * P16_W_SCALE);
            issue_row(0);
            dma_wait_le<0>();
            __syncthreads();
        }
        for (int kh = 0; kh < KS; ++kh) {
            if (kh + 1 < KS) issue_row(kh + 1);
            if (kh + 2 < KS) issue_wrow(kh + 2);
            const unsigned char* sa0 = smem + P16_OFF_A + wave * (2 * P16_AROW) + (kh & 1) * P16_AROW + off_i * 32 + 16 * h;
            const unsigned char* sb = smem + (kh % P16_RING) * P16_STAGE_B;
#pragma unroll
            for (int kw = 0; kw < KS; ++kw) {
                const f16x8 fa_hi = __builtin_bit_cast(f16x8, *reinterpret_cast<const s16x8*>(sa0 + kw * 32));
                const f16x8 fa_lo = __builtin_bit_cast(f16x8, *reinterpret_cast<const s16x8*>(sa0 + kw * 32 + P16_APART));
                const f16x8 w_hi = __builtin_bit_cast(f16x8, *reinterpret_cast<const s16x8*>(sb + kw * 2048 + boff_hi));
                const f16x8 w_lo = __builtin_bit_cast(f16x8, *reinterpret_cast<const s16x8*>(sb + kw * 2048 + boff_lo));
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa_hi, w_lo, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa_lo, w_hi, c2, 0, 0, 0);
                hh[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa_hi, w_hi, hh[0], 0, 0, 0);
            }
            // next row's patches and weights must have landed; only the weight row issued above may still be in flight
            if (kh + 2 < KS) { if (four) dma_wait_le<4>(); else dma_wait_le<3>(); } else dma_wait_le<0>();
            __syncthreads();
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) hh[0][r] += c1[r] + c2[r];
    } else {
    if (KEYS) issue_row(0);
#pragma unroll
    for (int t = 0; t < PD; ++t) { issue_w(t); if (!KEYS) issue_q(t); }
    P16_WAIT(PD - 1);
    __syncthreads();
    if (p16_tier_is_coarse(pa, head, lane, tier_rq)) {                 // (block-uniform) the map lives in the coarse tier
        tier.hi = pa.map_hi2; tier.lo = pa.map_lo2; tier.unscale = 1.0f / (B1_COARSE_SCALE * P16_W_SCALE);
        if (!KEYS) {
            int qy = gy, gx = gx0 + i;
            if (lin) { int q = base_row + i; if (q > gr.L - 1) q = gr.L - 1; qy = q / row_len; gx = q - qy * row_len; }
            else if (gx >= row_len) gx = row_len - 1;
            const int py = QS * qy - gr.pt + PADPIX, px = QS * gx - gr.pl + PADPIX;
            const size_t aoff = (((size_t)b * gr.Hp + py) * gr.Wp + px) * CH + 8 * h;
            ahi = tier.hi + aoff; alo = tier.lo + aoff;
#pragma unroll
            for (int t = 0; t < PD; ++t) issue_q(t);
        } else {
            issue_row(0);
        }
        dma_wait_le<0>();
        __syncthreads();
    }
    dbg_stamp(pa.times, blockIdx.x, 1);

    if (VAR == 9) { if (hh[0][0] != 0.f) pa.feat[which][0] = 1.f; return; }
    auto compute = [&](int step) {
        const int kh = step / KS, kw = step - kh * KS;
        const unsigned char* sa;
        int lo_off;
        if (KEYS) { sa = smem + P16_OFF_A + wave * (2 * P16_AROW) + (kh & 1) * P16_AROW + (off_i + kw) * 32 + 16 * h; lo_off = P16_APART; }
        else { sa = smem + P16_OFF_A + (step % P16_QRING) * (P16_BW * 2048) + wave * 2048 + lane * 16; lo_off = 1024; }
        const f16x8 fa_hi = __builtin_bit_cast(f16x8, *reinterpret_cast<const s16x8*>(sa));
        const f16x8 fa_lo = __builtin_bit_cast(f16x8, *reinterpret_cast<const s16x8*>(sa + lo_off));
        const unsigned char* sb = smem + (step % P16_RING) * P16_STAGE_B;
        // all fragment reads of the tap first (one exposed LDS latency per tap instead of one per tile), then the
        // MFMAs grouped so that no instruction depends on its predecessor
        f16x8 w_hi[NT], w_lo[NT];
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            w_hi[n] = __builtin_bit_cast(f16x8, *reinterpret_cast<const s16x8*>(sb + n * 32 * P16_ROWH * 2 + boff_hi));
            w_lo[n] = __builtin_bit_cast(f16x8, *reinterpret_cast<const s16x8*>(sb + n * 32 * P16_ROWH * 2 + boff_lo));
        }
        if (VAR == 5) {
#pragma unroll
            for (int n = 0; n < NT; ++n) asm volatile("" :: "v"(fa_hi), "v"(fa_lo), "v"(w_hi[n]), "v"(w_lo[n]));
            return;
        }
        // the two cross terms (2^-11 of the main one) go into the same accumulator: small terms first
#pragma unroll
        for (int n = 0; n < NT; ++n) hh[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa_hi, w_lo[n], hh[n], 0, 0, 0);
#pragma unroll
        for (int n = 0; n < NT; ++n) hh[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa_lo, w_hi[n], hh[n], 0, 0, 0);
#pragma unroll
        for (int n = 0; n < NT; ++n) hh[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa_hi, w_hi[n], hh[n], 0, 0, 0);
    };
    // steady state: PD-1 younger taps stay in flight across the barrier
    for (int step = 0; step < ((VAR == 7) ? 1 : (VAR == 8) ? 23 : P16_STEPS - PD); ++step) {
        if (KEYS && (step % KS) == 0 && step / KS + 1 < KS) issue_row(step / KS + 1);     // one kernel row ahead
        issue_w(step + PD);
        if (!KEYS) issue_q(step + PD);
        compute(step);
        P16_WAIT(PD - 1);
        __syncthreads();
    }
    // drain: the last PD taps, nothing left to issue
    if (PD == 3) { compute(P16_STEPS - 3); P16_WAIT(1); __syncthreads(); }
    compute(P16_STEPS - 2); P16_WAIT(0); __syncthreads();
    compute(P16_STEPS - 1);
    __syncthreads();
    }
#undef P16_WAIT
    static_assert(PD == 2 || PD == 3, "the drain sequence above is written for PD = 2 or 3");

    dbg_stamp(pa.times, blockIdx.x, 2);
    // ---- epilogue: D[row = patch (r&3)+8(r>>2)+4h][col = output (n0+n)*32 + i] ----------------------------
    float* fb = pa.feat[which] + (size_t)b * pa.rows_alloc[which] * DS;
    uint16_t* hb = pa.feat_h[which] ? pa.feat_h[which] + (size_t)b * pa.rows_alloc_h[which] * DSH : nullptr;
    const float* __restrict__ fbias = pa.bias[which][head];
    const int grid_row_base = base_row;
    float colsum_r[NT];                                                 // keys: this lane's share of the column sums
    if (NT == P16_NT && VAR == 0) {
        // Full blocks: a lane holds 16 rows x 7 columns of its wave's 32 x 224 tile, one dword of a row per store -- 224 store
        // instructions per wave (fp32 + bf16), each covering two 128-byte (64-byte) row segments; with every CU storing at once
        // that was 14.6 us of the kernel (store-issue bound: MI355X_MICROARCH.md, epilogue store tail).  The wave's 32 feature
        // rows are CONTIGUOUS in memory (row stride = row length = 816 bytes; 432 for the bf16 copy), so the tile goes through
        // the LDS (the weight ring is dead now: 20 KiB per wave, two passes of 16 rows) and leaves as lane-linear 16-byte
        // stores: 26 + 14 store instructions per wave instead of 224, whole cache lines.
        float* stg = reinterpret_cast<float*>(smem) + wave * (16 * DS);     // [16 rows][204] floats = 13056 B per wave
#pragma unroll
        for (int n = 0; n < NT; ++n) colsum_r[n] = 0.f;
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const int col = n * 32 + i;
                const float bv = (col < D) ? fbias[col] : 0.0f;
#pragma unroll
                for (int r8 = 0; r8 < 8; ++r8) {
                    const int r = 8 * pass + r8;
                    const int rl = (r8 & 3) + 8 * (r8 >> 2) + 4 * h;              // row inside the pass: 0..15
                    const int rr = rl + 16 * pass;
                    float v = hh[n][r] * tier.unscale + bv;
                    v = v > 0.f ? v : 0.f;
                    if (col >= D) v = 0.f;
                    if (col < DS) stg[rl * DS + col] = v;
                    colsum_r[n] += (wave_valid && rr < lim) ? v : 0.f;
                }
            }
            // (the wave only reads back what it wrote itself: LDS operations of a wave execute in order, no barrier)
            const int rows_here = wave_valid ? (lim - 16 * pass < 16 ? (lim - 16 * pass < 0 ? 0 : lim - 16 * pass) : 16) : 0;
            const int n4 = rows_here * (DS / 4);                                  // float4 chunks of the contiguous rows
            float4* dst = reinterpret_cast<float4*>(fb + (size_t)(grid_row_base + 16 * pass) * DS);
            const float4* src = reinterpret_cast<const float4*>(stg);
#pragma unroll
            for (int j = 0; j < (16 * (DS / 4) + 63) / 64; ++j) {                 // 13
                const int e = lane + 64 * j;
                if (e < n4) dst[e] = src[e];
            }
            if (hb != nullptr) {
                // bf16 copy: rows of 216 halfs = 27 chunks of 8 columns; columns 196.. are zero
                const bool tiled = pa.tiled_h[which] != 0;                       // 26 x (16 rows x 16 B) per pass instead of 16 rows x 27
                const int n8 = tiled ? 26 * 16 : rows_here * (DSH / 8);
                uint4* dh = reinterpret_cast<uint4*>(hb + (size_t)(grid_row_base + (tiled ? 0 : 16 * pass)) * DSH);
#pragma unroll
                for (int j = 0; j < (16 * (DSH / 8) + 63) / 64; ++j) {            // 7
                    const int e = lane + 64 * j;
                    const int row = tiled ? (e & 15) : e / (DSH / 8);
                    const int c8 = tiled ? (e >> 4) : e - row * (DSH / 8);
                    if (e < n8 && row < rows_here) {
                        // columns 8 c8 .. + 7 of the staged row: two 16-byte reads (row stride 816 B = 51 x 16); past column 203: zeros
                        const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
                        const float4 lo4 = (8 * c8 < DS) ? *reinterpret_cast<const float4*>(stg + row * DS + 8 * c8) : z4;
                        const float4 hi4 = (8 * c8 + 4 < DS) ? *reinterpret_cast<const float4*>(stg + row * DS + 8 * c8 + 4) : z4;
                        const float f8[8] = {lo4.x, lo4.y, lo4.z, lo4.w, hi4.x, hi4.y, hi4.z, hi4.w};
                        unsigned short q[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            unsigned bits = __float_as_uint(f8[u]);
                            bits = (bits + 0x7FFFu + ((bits >> 16) & 1u)) >> 16;   // fp32 -> bf16, round to nearest even
                            q[u] = (unsigned short)bits;
                        }
                        dh[tiled ? 416 * pass + e : e] = make_uint4(q[0] | ((unsigned)q[1] << 16), q[2] | ((unsigned)q[3] << 16),
                                                                               q[4] | ((unsigned)q[5] << 16), q[6] | ((unsigned)q[7] << 16));
                    }
                }
            }
        }
    } else {
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int col = (n0 + n) * 32 + i;
        const float bv = (col < D) ? fbias[col] : 0.0f;
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rr = (r & 3) + 8 * (r >> 2) + 4 * h;
            const bool ok = wave_valid && (rr < lim);
            float v = hh[n][r] * tier.unscale + bv;
            v = v > 0.f ? v : 0.f;
            if (col >= D) v = 0.f;
            if (VAR == 1 && v != 12345.678f) continue;
            if (ok && col < DS) fb[(size_t)(grid_row_base + rr) * DS + col] = v;
            if (VAR != 3 && ok && hb != nullptr && col < DPAD) {
                unsigned u = __float_as_uint(v);
                u = (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;            // fp32 -> bf16, round to nearest even
                if (pa.tiled_h[which]) hb[(size_t)grid_row_base * DSH + (rr >> 4) * 3328 + ((col >> 3) * 16 + (rr & 15)) * 8 + (col & 7)] = (uint16_t)u;
                else hb[(size_t)(grid_row_base + rr) * DSH + col] = (uint16_t)u;
            }
            if (VAR == 0 && ok && pa.split_hi[which] != nullptr && col < DSH) {           // split-fp16 copy (see project16_body2): columns 196 .. 215 zero
                const float vs = v * DN_FS;
                if (!(vs < RANGE_LIMIT) && pa.range.word != nullptr) *pa.range.word = pa.range.tag;
                const _Float16 hv = (_Float16)vs;
                const size_t o = ((size_t)b * pa.rows_alloc_s[which] + grid_row_base + rr) * DSH + col;
                pa.split_hi[which][o] = __builtin_bit_cast(unsigned short, hv);
                pa.split_lo[which][o] = __builtin_bit_cast(unsigned short, (_Float16)(vs - (float)hv));
            }
            s += ok ? v : 0.f;
        }
        colsum_r[n] = s;
    }
    }
    float* csum = reinterpret_cast<float*>(smem);                       // [waves][NT*32] (everything else in the LDS is dead now)
    if (KEYS) {
        if (NT == P16_NT && VAR == 0) __syncthreads();                  // the staging regions are about to be overwritten
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            float s = colsum_r[n];
            s += __shfl_xor(s, 32);                                   // the two row halves of the tile
            if (h == 0) csum[wave * (NT * 32) + n * 32 + i] = s;
        }
    }
    if (KEYS && pa.colpart != nullptr) {
        // fixed-order block reduction of the key column sums (no atomics: the row mean must be reproducible)
        __syncthreads();
        if (tid < NT * 32) {
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < P16_BW; ++w) t += csum[w * (NT * 32) + tid];
            pa.colpart[((size_t)b * pa.n_blocks_k + blk) * P16_OUT + n0 * 32 + tid] = t;
        }
    }
}


// ---- round 4: 64 patches x (4 | 3) output tiles per wave ------------------------------------------------------------------------
// The round-3 shape (a wave = 32 patches x all 7 output tiles, two 4-wave blocks per CU, each streaming the WHOLE 13 KiB weight
// slice of a tap) spent as many issue cycles on LDS-DMA pieces and weight-fragment reads as on its multiplies: per SIMD and tap 42
// multiplies (1344 cycles) beside 6.5 DMA pieces (~110 cycles each) and 32 LDS reads -- 1.13 us per tap against 0.61 of matrix work.
// Here a wave carries TWO 32-patch items and the tiles of ONE tile group (0-3 or 4-6): a weight fragment feeds two multiplies
// chains, a block fetches only its group's rows of the slice (8 / 6 KiB), and the two blocks of a CU -- one per group in the usual
// dispatch -- fetch the slice ONCE between them: 3.5 pieces and 22 reads per SIMD and tap for the same 42 multiplies.
// 2 x 4 x 16 = 128 accumulator registers: two blocks per CU stay resident (independent barriers, as before).
constexpr int P16_PW = 2;                                   // 32-patch items per wave
constexpr int P16_UNIT = P16_BW * P16_PW;                   // items per block (a "unit" = 8 items = 256 patches)
constexpr int P16_G0 = 4, P16_G1 = P16_NT - P16_G0;         // tiles of the two groups
constexpr int P16_STAGE2_B = P16_G0 * 2048;                 // bytes per weight stage (the larger group)
constexpr int P16_OFF_A2 = P16_RING * P16_STAGE2_B;         // 32 KiB: patch region behind the weight ring
constexpr int P16_LDS2 = P16_OFF_A2 + P16_QRING * P16_BW * P16_PW * 2048;      // 80 KiB (keys use 32 + 8 x 5.5 = 76)
static_assert(P16_BW * P16_PW * 2 * P16_AROW <= P16_QRING * P16_BW * P16_PW * 2048, "key row rings must fit the patch region");
static_assert(2 * P16_LDS2 <= 160 * 1024 && P16_LDS2 <= P16_LDS, "two resident blocks per CU, inside the kernel's allocation");
static_assert(P16_BW * 16 * P16_G0 * 32 * 4 <= P16_OFF_A2, "the epilogue stages 16 rows x its columns per wave in the dead weight ring");

template <int NT, bool KEYS>
__device__ __forceinline__ void project16_body2(const Proj16Args& pa, unsigned char* smem, int n0, int unit, int b) {
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int i = lane & 31, h = lane >> 5;
    const Grid& gr = pa.gr;
    constexpr int which = KEYS ? 0 : 1;
    constexpr int PW = P16_PW;
    const int head = b / pa.imgs_per_head;
    P16TierReq tier_rq;
    p16_tier_request(pa, head, lane, tier_rq);
    P16Tier tier = {pa.map_hi, pa.map_lo, 1.0f / (P16_A_SCALE * P16_W_SCALE)};
    const unsigned short* __restrict__ wp = pa.wp[which] + (size_t)head * P16_PACKED_HALFS + (size_t)n0 * 32 * P16_ROWH;
    const int n_items = pa.n_items[which];
    const int segs_per_row = pa.segs[which];
    constexpr int PD = KEYS ? P16_PD_KEYS : P16_PD_Q;
    constexpr int PIECES = NT * 2;                                           // KiB of the group's rows of a tap slice (tiles 4-6: rows
                                                                              // 128-223, of which 208-223 are padding never stored)
    constexpr int PBASE = PIECES / P16_BW;
    const bool extra = wave < (PIECES % P16_BW);
    const int row_len = KEYS ? gr.W : gr.Lw;
    const bool lin = pa.lin[which] != 0;

    int gy[PW], gx0[PW], nA[PW], base_row[PW], lim[PW], off_i[PW];
    bool item_valid[PW];
#pragma unroll
    for (int it = 0; it < PW; ++it) {
        int item = (unit * P16_BW + wave) * PW + it;
        item_valid[it] = item < n_items;
        if (!item_valid[it]) item = n_items - 1;
        if (lin) {
            base_row[it] = item * 32;
            gy[it] = base_row[it] / row_len; gx0[it] = base_row[it] - gy[it] * row_len;
            nA[it] = row_len - gx0[it] < 32 ? row_len - gx0[it] : 32;
            lim[it] = (KEYS ? gr.N : gr.L) - base_row[it];
        } else {
            gy[it] = item / segs_per_row; gx0[it] = (item % segs_per_row) * 32;
            nA[it] = 32; base_row[it] = gy[it] * row_len + gx0[it]; lim[it] = row_len - gx0[it];
        }
        off_i[it] = (i < nA[it]) ? i : i + 6;        // keys: the patch's pixel position in the staged row (segment B behind A's halo)
    }
    const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr_of(smem));

    f32x16 hh[PW][NT];
#pragma unroll
    for (int it = 0; it < PW; ++it)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) hh[it][n][r] = 0.f;

    auto issue_w = [&](int t) {                        // the group's rows of tap t's weight slice -> ring stage t % RING
#if defined(DAGL_P16_HALFW)
        if (t & 1) return;               // (timing experiment, wrong results: half of the weight LDS-DMA)
#endif
#if defined(DAGL_P16_NOW)
        return;                          // (timing experiment, wrong results: no weight LDS-DMA at all)
#endif
        const unsigned st = lds0 + (unsigned)(t % P16_RING) * P16_STAGE2_B;
        const unsigned short* wsrc = wp + (size_t)t * P16_SLICE_H;
#if !defined(DAGL_P16_M0_PER_PIECE)
        // a wave's pieces are ADJACENT, so that two of them go out under one M0 value (glds16x2_asm); same pieces per wave as below
        if (PBASE == 2 && (PIECES % P16_BW) == 0) {
            const int p = 2 * wave;
            glds16x2_asm(reinterpret_cast<const float*>(wsrc + (size_t)p * 512 + lane * 8), __builtin_amdgcn_readfirstlane(st + p * 1024));
        } else if (PBASE == 1) {
            const int nx = PIECES % P16_BW;                                   // waves carrying two pieces
            if (extra) {
                const int p = 2 * wave;
                glds16x2_asm(reinterpret_cast<const float*>(wsrc + (size_t)p * 512 + lane * 8), __builtin_amdgcn_readfirstlane(st + p * 1024));
            } else {
                const int p = nx + wave;
                glds16_asm(reinterpret_cast<const float*>(wsrc + (size_t)p * 512 + lane * 8), __builtin_amdgcn_readfirstlane(st + p * 1024));
            }
        } else
#endif
        {
#pragma unroll
        for (int j = 0; j < PBASE; ++j) {
            const int p = wave + P16_BW * j;
            glds16_asm(reinterpret_cast<const float*>(wsrc + (size_t)p * 512 + lane * 8), __builtin_amdgcn_readfirstlane(st + p * 1024));
        }
        if (extra) {
            const int p = wave + P16_BW * PBASE;
            glds16_asm(reinterpret_cast<const float*>(wsrc + (size_t)p * 512 + lane * 8), __builtin_amdgcn_readfirstlane(st + p * 1024));
        }
        }
    };
    // ---- patch operand plumbing (per item) ---------------------------------------------------------------------------------------
    const unsigned short* ahi[PW]; const unsigned short* alo[PW];                // queries: this lane's patch corner
    const size_t krow0 = (size_t)b * gr.Hp * gr.Wp * CH;                          // keys: halfs offset of the image's map
#pragma unroll
    for (int it = 0; it < PW; ++it) {
        ahi[it] = alo[it] = nullptr;
        if (!KEYS) {
            int qy = gy[it], gx = gx0[it] + i;
            if (lin) { int q = base_row[it] + i; if (q > gr.L - 1) q = gr.L - 1; qy = q / row_len; gx = q - qy * row_len; }
            else if (gx >= row_len) gx = row_len - 1;
            const int py = QS * qy - gr.pt + PADPIX, px = QS * gx - gr.pl + PADPIX;
            const size_t aoff = (((size_t)b * gr.Hp + py) * gr.Wp + px) * CH + 8 * h;
            ahi[it] = tier.hi + aoff; alo[it] = tier.lo + aoff;
        }
    }
    auto issue_row = [&](int r) {                      // keys: kernel row r of both items (44 pixels, hi | lo) -> row buffer r & 1
#pragma unroll
        for (int it = 0; it < PW; ++it) {
#if defined(DAGL_P16_HALFROWS)
            if (it > 0) break;           // (timing experiment, wrong results: half of the key-row LDS-DMA -- what a block shared by both tile groups would issue)
#endif
            const unsigned dst = lds0 + P16_OFF_A2 + (wave * PW + it) * (2 * P16_AROW) + (r & 1) * P16_AROW;
#if !defined(DAGL_P16_M0_PER_PIECE)
            // pieces: hi px 0-31, hi px 32-43 (24 lanes), lo px 0-31, lo px 32-43 -- the two whole ones and the two partial ones each under ONE
            // M0 value (glds16x2_off_asm; hi and lo of a pixel sit at the same offset of their maps)
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int p = half * 32 + (lane >> 1);
                int row = gy[it] + r, px = gx0[it] + p;
                if (p >= nA[it] + 6) { row += 1; px = p - (nA[it] + 6); }
                if (px > gr.Wp - 1) px = gr.Wp - 1;                                   // stay inside the map
                if (row > gr.Hp - 1) row = gr.Hp - 1;
                const size_t o = krow0 + ((size_t)row * gr.Wp + px) * CH + 8 * (lane & 1);
                if (half == 0) glds16x2_off_asm<0, P16_APART>(tier.hi + o, tier.lo + o, __builtin_amdgcn_readfirstlane(dst));
                else if (lane < 2 * (P16_APX - 32)) glds16x2_off_asm<1024, P16_APART + 1024>(tier.hi + o, tier.lo + o, __builtin_amdgcn_readfirstlane(dst));
            }
#else
#pragma unroll
            for (int j = 0; j < 4; ++j) {              // pieces: hi px 0-31, hi px 32-43 (24 lanes), lo px 0-31, lo px 32-43
                const int p = (j & 1) * 32 + (lane >> 1);
                int row = gy[it] + r, px = gx0[it] + p;
                if (p >= nA[it] + 6) { row += 1; px = p - (nA[it] + 6); }
                if (px > gr.Wp - 1) px = gr.Wp - 1;                                   // stay inside the map
                if (row > gr.Hp - 1) row = gr.Hp - 1;
                const unsigned short* src = ((j < 2) ? tier.hi : tier.lo) + krow0 + ((size_t)row * gr.Wp + px) * CH + 8 * (lane & 1);
                const unsigned d = dst + (j >> 1) * P16_APART + (j & 1) * 1024;
                if ((j & 1) == 0 || lane < 2 * (P16_APX - 32)) glds16_asm(reinterpret_cast<const float*>(src), __builtin_amdgcn_readfirstlane(d));
            }
#endif
        }
    };
    auto issue_q = [&](int t) {                        // queries: the 16 B of tap t this lane will read back, both items
        const int kh = t / KS, kw = t - kh * KS;
        const size_t o = ((size_t)kh * gr.Wp + kw) * CH;
#pragma unroll
        for (int it = 0; it < PW; ++it) {
            const unsigned sa = lds0 + P16_OFF_A2 + (unsigned)(t % P16_QRING) * (P16_BW * PW * 2048) + (wave * PW + it) * 2048;
            glds16_asm(reinterpret_cast<const float*>(ahi[it] + o), __builtin_amdgcn_readfirstlane(sa));
            glds16_asm(reinterpret_cast<const float*>(alo[it] + o), __builtin_amdgcn_readfirstlane(sa + 1024));
        }
    };
    constexpr int PER_Q = KEYS ? 0 : 2 * PW;
#define P16_WAIT2(P) do { if (extra) dma_wait_le<(P) * (PBASE + 1 + PER_Q)>(); else dma_wait_le<(P) * (PBASE + PER_Q)>(); } while (0)

    const int swz = (i >> 2) & 3;                                       // slot swizzle of row n*32 + i (n*32 does not change it)
    const int boff_hi = i * (P16_ROWH * 2) + ((h ^ swz) << 4);          // bytes: B fragment row n*32 + i, hi half h
    const int boff_lo = i * (P16_ROWH * 2) + (((2 + h) ^ swz) << 4);

    if (KEYS) issue_row(0);
#pragma unroll
    for (int t = 0; t < PD; ++t) { issue_w(t); if (!KEYS) issue_q(t); }
    P16_WAIT2(PD - 1);
    __syncthreads();
    if (p16_tier_is_coarse(pa, head, lane, tier_rq)) {                 // (block-uniform) the map lives in the coarse tier
        tier.hi = pa.map_hi2; tier.lo = pa.map_lo2; tier.unscale = 1.0f / (B1_COARSE_SCALE * P16_W_SCALE);
        if (!KEYS) {
#pragma unroll
            for (int it = 0; it < PW; ++it) {
                const size_t d = ahi[it] - pa.map_hi;                  // this lane's patch corner, same offset in the other tier
                ahi[it] = tier.hi + d; alo[it] = tier.lo + d;
            }
#pragma unroll
            for (int t = 0; t < PD; ++t) issue_q(t);
        } else {
            issue_row(0);
        }
        dma_wait_le<0>();
        __syncthreads();
    }
    dbg_stamp(pa.times, blockIdx.x, 1);

    auto compute = [&](int step) {
        const int kh = step / KS, kw = step - kh * KS;
        f16x8 fa_hi[PW], fa_lo[PW];
#pragma unroll
        for (int it = 0; it < PW; ++it) {
            const unsigned char* sa;
            int lo_off;
            if (KEYS) { sa = smem + P16_OFF_A2 + (wave * PW + it) * (2 * P16_AROW) + (kh & 1) * P16_AROW + (off_i[it] + kw) * 32 + 16 * h; lo_off = P16_APART; }
            else { sa = smem + P16_OFF_A2 + (step % P16_QRING) * (P16_BW * PW * 2048) + (wave * PW + it) * 2048 + lane * 16; lo_off = 1024; }
            fa_hi[it] = __builtin_bit_cast(f16x8, *reinterpret_cast<const s16x8*>(sa));
            fa_lo[it] = __builtin_bit_cast(f16x8, *reinterpret_cast<const s16x8*>(sa + lo_off));
        }
        const unsigned char* sb = smem + (step % P16_RING) * P16_STAGE2_B;
        f16x8 w_hi[NT], w_lo[NT];
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            w_hi[n] = __builtin_bit_cast(f16x8, *reinterpret_cast<const s16x8*>(sb + n * 32 * P16_ROWH * 2 + boff_hi));
            w_lo[n] = __builtin_bit_cast(f16x8, *reinterpret_cast<const s16x8*>(sb + n * 32 * P16_ROWH * 2 + boff_lo));
        }
        // the two cross terms (2^-11 of the main one) go into the same accumulator: small terms first; a weight fragment feeds both items
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int it = 0; it < PW; ++it) hh[it][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa_hi[it], w_lo[n], hh[it][n], 0, 0, 0);
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int it = 0; it < PW; ++it) hh[it][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa_lo[it], w_hi[n], hh[it][n], 0, 0, 0);
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int it = 0; it < PW; ++it) hh[it][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa_hi[it], w_hi[n], hh[it][n], 0, 0, 0);
    };
#ifndef DAGL_P16_PIPE
#if defined(DAGL_ABLATION) && defined(DAGL_P16_PHASES)
    // (experiment) where a wave's loop time goes: shader clocks of (0) the requests, (1) fragment reads + multiplies issued, (2) the counted
    // wait for tap t + 1's pieces, (3) the barrier; wave 0's sums replace the block's four stamps
    unsigned long long ph[4] = {0, 0, 0, 0};
    unsigned long long pt = __builtin_amdgcn_s_memtime();
#define P16_PH(k) do { const unsigned long long t1_ = __builtin_amdgcn_s_memtime(); ph[k] += t1_ - pt; pt = t1_; } while (0)
#else
#define P16_PH(k) do { } while (0)
#endif
    for (int step = 0; step < P16_STEPS - PD; ++step) {
        if (KEYS && (step % KS) == 0 && step / KS + 1 < KS) issue_row(step / KS + 1);     // one kernel row ahead
        issue_w(step + PD);
        if (!KEYS) issue_q(step + PD);
        P16_PH(0);
        compute(step);
        __builtin_amdgcn_sched_barrier(0);
        P16_PH(1);
        P16_WAIT2(PD - 1);
        P16_PH(2);
        __syncthreads();
        P16_PH(3);
    }
#if defined(DAGL_ABLATION) && defined(DAGL_P16_PHASES)
    if (pa.times != nullptr && threadIdx.x == 0) {
        pa.times[(size_t)blockIdx.x * 4 + 0] = ph[0]; pa.times[(size_t)blockIdx.x * 4 + 1] = ph[1];
        pa.times[(size_t)blockIdx.x * 4 + 2] = ph[2]; pa.times[(size_t)blockIdx.x * 4 + 3] = ph[3];
    }
#endif
#undef P16_PH
    if (PD == 3) { compute(P16_STEPS - 3); P16_WAIT2(1); __syncthreads(); }
    compute(P16_STEPS - 2); P16_WAIT2(0); __syncthreads();
    compute(P16_STEPS - 1);
    __syncthreads();
#else
    // (experiment, round 6) the fragments of tap t + 1 are read right behind tap t's barrier -- which is what makes them readable -- and
    // under tap t's second and third multiply groups; tap t + 1 starts its first group from registers.  Two fragment sets, taps in pairs.
    (void)compute;
    struct Frag { f16x8 fa_hi[PW], fa_lo[PW], w_hi[NT], w_lo[NT]; };
    auto load = [&](int step, Frag& f) {
        const int kh = step / KS, kw = step - kh * KS;
#pragma unroll
        for (int it = 0; it < PW; ++it) {
            const unsigned char* sa;
            int lo_off;
            if (KEYS) { sa = smem + P16_OFF_A2 + (wave * PW + it) * (2 * P16_AROW) + (kh & 1) * P16_AROW + (off_i[it] + kw) * 32 + 16 * h; lo_off = P16_APART; }
            else { sa = smem + P16_OFF_A2 + (step % P16_QRING) * (P16_BW * PW * 2048) + (wave * PW + it) * 2048 + lane * 16; lo_off = 1024; }
            f.fa_hi[it] = __builtin_bit_cast(f16x8, *reinterpret_cast<const s16x8*>(sa));
            f.fa_lo[it] = __builtin_bit_cast(f16x8, *reinterpret_cast<const s16x8*>(sa + lo_off));
        }
        const unsigned char* sb = smem + (step % P16_RING) * P16_STAGE2_B;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            f.w_lo[n] = __builtin_bit_cast(f16x8, *reinterpret_cast<const s16x8*>(sb + n * 32 * P16_ROWH * 2 + boff_lo));
            f.w_hi[n] = __builtin_bit_cast(f16x8, *reinterpret_cast<const s16x8*>(sb + n * 32 * P16_ROWH * 2 + boff_hi));
        }
    };
    auto g1 = [&](const Frag& f) {
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int it = 0; it < PW; ++it) hh[it][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.fa_hi[it], f.w_lo[n], hh[it][n], 0, 0, 0);
    };
    auto g23 = [&](const Frag& f) {
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int it = 0; it < PW; ++it) hh[it][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.fa_lo[it], f.w_hi[n], hh[it][n], 0, 0, 0);
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int it = 0; it < PW; ++it) hh[it][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.fa_hi[it], f.w_hi[n], hh[it][n], 0, 0, 0);
    };
    // one tap: WAITP = taps whose requests may stay in flight behind it (-1: the last tap, nothing to wait for)
#define P16_TAP(STEP, CUR, NXT, ISSUE, WAITP)                                                                              \
    do {                                                                                                                   \
        if (ISSUE) {                                                                                                       \
            if (KEYS && ((STEP) % KS) == 0 && (STEP) / KS + 1 < KS) issue_row((STEP) / KS + 1);                            \
            issue_w((STEP) + PD);                                                                                          \
            if (!KEYS) issue_q((STEP) + PD);                                                                               \
        }                                                                                                                  \
        g1(CUR);                                                                                                           \
        if ((WAITP) >= 0) {                                                                                                \
            if ((WAITP) == 2) P16_WAIT2(2); else if ((WAITP) == 1) P16_WAIT2(1); else P16_WAIT2(0);                        \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      /* this wave's reads of the stage about to be refilled */ \
            __syncthreads();                                                                                               \
            load((STEP) + 1, NXT);                                                                                         \
        }                                                                                                                  \
        g23(CUR);                                                                                                          \
        if ((WAITP) >= 0) {          /* one fragment read behind every multiply (left alone, hipcc reads a fragment right before its use) */ \
            _Pragma("unroll") for (int pin_ = 0; pin_ < 2 * PW + 2 * NT; ++pin_) {                                         \
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                         \
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                         \
            }                                                                                                              \
            __builtin_amdgcn_sched_group_barrier(0x008, 2 * NT * PW - (2 * PW + 2 * NT), 0);                               \
        }                                                                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                                 \
    } while (0)
    Frag fA, fB;
    load(0, fA);
    static_assert(P16_STEPS == 49 && (PD == 2 || PD == 3), "the tap pairs below are written for 49 taps");
    for (int step = 0; step < 46; step += 2) {                 // taps 0 .. 45: steady for both prefetch distances
        P16_TAP(step, fA, fB, true, PD - 1);
        P16_TAP(step + 1, fB, fA, true, PD - 1);
    }
    if (PD == 2) P16_TAP(46, fA, fB, true, 1); else P16_TAP(46, fA, fB, false, 1);
    P16_TAP(47, fB, fA, false, 0);
    P16_TAP(48, fA, fB, false, -1);
    __syncthreads();
#undef P16_TAP
#endif
#undef P16_WAIT2
#if !defined(DAGL_P16_PHASES)
    dbg_stamp(pa.times, blockIdx.x, 2);
#endif

    // ---- epilogue: D[row = patch (r&3)+8(r>>2)+4h][col = output (n0+n)*32 + i]: this block's columns [c0, c0 + cw) of the rows ----
    float* fb = pa.feat[which] + (size_t)b * pa.rows_alloc[which] * DS;
    uint16_t* hb = pa.feat_h[which] ? pa.feat_h[which] + (size_t)b * pa.rows_alloc_h[which] * DSH : nullptr;
    const float* __restrict__ fbias = pa.bias[which][head];
    constexpr int SEG = NT * 32;                                        // staged columns per row
    const int c0 = n0 * 32;
    const int cw = (c0 + SEG <= DS) ? SEG : DS - c0;                    // fp32 columns stored (tiles 4-6: 128 .. 203)
    const int cwh = (c0 + SEG <= DPAD) ? SEG : DPAD - c0;               // bf16 columns with data (.. 207; the row-major copy pads to 216)
    float* stg = reinterpret_cast<float*>(smem) + wave * (16 * SEG);    // [16 rows][SEG] floats in the dead weight ring
    float colsum_r[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) colsum_r[n] = 0.f;
#pragma unroll
    for (int it = 0; it < PW; ++it) {
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const int col = c0 + n * 32 + i;
                const float bv = (col < D) ? fbias[col] : 0.0f;
#pragma unroll
                for (int r8 = 0; r8 < 8; ++r8) {
                    const int r = 8 * pass + r8;
                    const int rl = (r8 & 3) + 8 * (r8 >> 2) + 4 * h;              // row inside the pass: 0..15
                    const int rr = rl + 16 * pass;
                    float v = hh[it][n][r] * tier.unscale + bv;
                    v = v > 0.f ? v : 0.f;
                    if (col >= D) v = 0.f;
                    stg[rl * SEG + n * 32 + i] = v;
                    colsum_r[n] += (item_valid[it] && rr < lim[it]) ? v : 0.f;
                }
            }
            // (the wave only reads back what it wrote itself: LDS operations of a wave execute in order, no barrier)
            const int rows_here = item_valid[it] ? (lim[it] - 16 * pass < 16 ? (lim[it] - 16 * pass < 0 ? 0 : lim[it] - 16 * pass) : 16) : 0;
            {
                const int cpr = cw / 4;                                          // float4 chunks per row: 32 or 19
                float* dst = fb + (size_t)(base_row[it] + 16 * pass) * DS + c0;
#pragma unroll
                for (int j = 0; j < (16 * (SEG / 4) + 63) / 64; ++j) {
                    const int e = lane + 64 * j;
                    const int row = e / cpr, c4 = e - row * cpr;
                    if (row < rows_here)
                        *reinterpret_cast<float4*>(dst + (size_t)row * DS + 4 * c4) = *reinterpret_cast<const float4*>(stg + row * SEG + 4 * c4);
                }
            }
            if (hb != nullptr) {
                const bool tiled = pa.tiled_h[which] != 0;
                // bf16 copy, 8-column chunks: row-major rows of 216 halfs (chunks c0/8 .. ; columns 196.. zero, the last group pads to
                // 216) or the screen's fragment order (26 chunks of 16 rows per 16-row half, ScreenArgs::q_tiled)
                const int ch0 = c0 / 8;
                const int nch = tiled ? cwh / 8 : ((c0 + SEG >= DPAD) ? (DSH / 8 - ch0) : SEG / 8);      // 16 | 10 (tiled) / 16 | 11
#pragma unroll
                for (int j = 0; j < (16 * (SEG / 8) + 63) / 64; ++j) {
                    const int e = lane + 64 * j;
                    const int row = tiled ? (e & 15) : e / nch;
                    const int c8 = tiled ? (e >> 4) : e - row * nch;
                    if (c8 < nch && row < rows_here) {
                        const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
                        const float4 lo4 = (8 * c8 < SEG) ? *reinterpret_cast<const float4*>(stg + row * SEG + 8 * c8) : z4;
                        const float4 hi4 = (8 * c8 + 4 < SEG) ? *reinterpret_cast<const float4*>(stg + row * SEG + 8 * c8 + 4) : z4;
                        const float f8[8] = {lo4.x, lo4.y, lo4.z, lo4.w, hi4.x, hi4.y, hi4.z, hi4.w};
                        unsigned short q[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            unsigned bits = __float_as_uint(f8[u]);
                            bits = (bits + 0x7FFFu + ((bits >> 16) & 1u)) >> 16;   // fp32 -> bf16, round to nearest even
                            q[u] = (unsigned short)bits;
                        }
                        const uint4 pk = make_uint4(q[0] | ((unsigned)q[1] << 16), q[2] | ((unsigned)q[3] << 16),
                                                    q[4] | ((unsigned)q[5] << 16), q[6] | ((unsigned)q[7] << 16));
                        if (tiled) reinterpret_cast<uint4*>(hb + (size_t)base_row[it] * DSH)[416 * pass + (ch0 + c8) * 16 + row] = pk;
                        else reinterpret_cast<uint4*>(hb + (size_t)(base_row[it] + 16 * pass + row) * DSH)[ch0 + c8] = pk;
                    }
                }
            }
            if (pa.split_hi[which] != nullptr) {
                // split-fp16 copy for the streamed dense formulation: DN_FS x = hi + lo, row-major rows of 216 halfs like the
                // bf16 copy (the same chunks of the staged rows; what dense.hip's feat_split_kernel would produce)
                unsigned short* sh = pa.split_hi[which] + (size_t)b * pa.rows_alloc_s[which] * DSH;
                unsigned short* sl = pa.split_lo[which] + (size_t)b * pa.rows_alloc_s[which] * DSH;
                const int ch0 = c0 / 8;
                const int nch = (c0 + SEG >= DPAD) ? (DSH / 8 - ch0) : SEG / 8;
                bool bad = false;
#pragma unroll
                for (int j = 0; j < (16 * (SEG / 8) + 63) / 64; ++j) {
                    const int e = lane + 64 * j;
                    const int row = e / nch, c8 = e - row * nch;
                    if (row < rows_here) {
                        // (c8 < SEG / 8 here: the pad chunk 26 lies inside the last group's staged columns 128..223)
                        const float4 lo4 = *reinterpret_cast<const float4*>(stg + row * SEG + 8 * c8);
                        const float4 hi4 = *reinterpret_cast<const float4*>(stg + row * SEG + 8 * c8 + 4);
                        unsigned h01, l01, h23, l23, h45, l45, h67, l67;
                        p16_split_pair(lo4.x, lo4.y, h01, l01, bad); p16_split_pair(lo4.z, lo4.w, h23, l23, bad);
                        p16_split_pair(hi4.x, hi4.y, h45, l45, bad); p16_split_pair(hi4.z, hi4.w, h67, l67, bad);
                        const size_t o = (size_t)(base_row[it] + 16 * pass + row) * (DSH / 8) + ch0 + c8;
                        reinterpret_cast<uint4*>(sh)[o] = make_uint4(h01, h23, h45, h67);
                        reinterpret_cast<uint4*>(sl)[o] = make_uint4(l01, l23, l45, l67);
                    }
                }
                if (bad && pa.range.word != nullptr) *pa.range.word = pa.range.tag;
            }
        }
    }
    if (KEYS && pa.colpart != nullptr) {
        // fixed-order block reduction of the key column sums (no atomics: the row mean must be reproducible); the unit's sums go
        // to the first of its two rows of colpart (rows count 4-item blocks, as the single-tile blocks write them), the second is zero
        float* csum = reinterpret_cast<float*>(smem);
        __syncthreads();                                                // the staging regions are about to be overwritten
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            float sacc = colsum_r[n];
            sacc += __shfl_xor(sacc, 32);                             // the two row halves of the tile
            if (h == 0) csum[wave * SEG + n * 32 + i] = sacc;
        }
        __syncthreads();
        if (tid < SEG) {
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < P16_BW; ++w) t += csum[w * SEG + tid];
            pa.colpart[((size_t)b * pa.n_blocks_k + 2 * unit) * P16_OUT + c0 + tid] = t;
            pa.colpart[((size_t)b * pa.n_blocks_k + 2 * unit + 1) * P16_OUT + c0 + tid] = 0.f;
        }
    }
}

template <int VAR>
__global__ __launch_bounds__(64 * P16_BW, P16_BLOCKS_PER_CU) void project16_kernel(Proj16Args pa) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[P16_LDS];
    // 1-D grid.  Full blocks first (project16_body2: a wave owns 64 patches x the 4 or 3 output tiles of its block's tile group); a
    // grid that overhangs the resident-block capacity by a few blocks would cost a whole extra round, so the overhang comes last,
    // cut into single-tile blocks (project16_body<1>).
    const int bid = blockIdx.x;
    if (bid >= pa.n_proj) {                          // (block-uniform) a block of the thr / bias heads: independent of everything this launch computes
        const int t = bid - pa.n_proj;
        const int bx = t % pa.thr_x, r2 = t / pa.thr_x;
        thr_bias4_block(pa.gr, pa.thr_hs, pa.thr_part, bx, r2 % pa.thr_y, r2 / pa.thr_y, smem);
        return;
    }
    dbg_stamp(pa.times, bid, 0);
    if (bid == 0 && threadIdx.x < 2 * pa.heads && pa.range.word != nullptr) {        // weights packed from out-of-range values?
        const unsigned short* wpk = pa.wp[threadIdx.x & 1];
        if (wpk != nullptr &&
            *reinterpret_cast<const int32_t*>(wpk + (size_t)(threadIdx.x >> 1) * P16_PACKED_HALFS + P16_PACKED_HALFS - 2) != 0)
            *pa.range.word = pa.range.tag;
    }
    // units of 8 items (256 patches), two blocks each (tile groups 0-3 / 4-6), per image [query units][key units]; the last
    // n_split_groups key units of the last image come last, cut into 2 x 7 single-tile blocks of 4 items each (round-3 body)
    const int per_units = pa.units_q + pa.units_k;
    if (bid < pa.n_full) {
        // block -> (unit, tile group).  The 4-tile blocks carry 4/7 of the work: block b runs on XCD b % 8 (observed; used for speed
        // only), so "even ids = group 0" would hand the even XCDs nothing but 4-tile blocks.  Per XCD x the q-th block takes unit
        // 8 (q / 2) + x (the two groups of a unit back to back on ONE XCD: they read the same map rows through one L2), group
        // (q & 1) ^ ((q >> 5) & 1) -- the flip every 32 blocks so that the second wave of blocks lands a 3-tile block beside a
        // 4-tile one where CUs are filled round-robin.  Ids past the last multiple of 16: unit = id / 2, group = id & 1.
        int vunit, grp;
        if (bid < (pa.n_full & ~15)) {
            const int xcd = bid & 7, q = bid >> 3;
#ifndef DAGL_P16_BAND_UNITS
            vunit = 8 * (q >> 1) + xcd; grp = (q ^ (q >> 5)) & 1;
#else
            // (round 5, measured and not shipped: profiles/r05_ab_project16_band.log) an XCD owns a BAND of consecutive units =
            // consecutive image rows: a unit of row-major key patches needs 7 map rows, 6 of which its neighbour needs too -- with units
            // dealt round-robin every map row is fetched by 7 of the 8 L2s.  Fewer fetches, but +1 us: the re-fetches come from the
            // Infinity Cache and the band puts the 16 query units' blocks on one XCD
            vunit = xcd * ((pa.n_full & ~15) >> 4) + (q >> 1); grp = (q ^ (q >> 5)) & 1;
#endif
        } else { vunit = bid >> 1; grp = bid & 1; }
        const int b = vunit / per_units, unit = vunit - b * per_units;
        if (VAR != 0) {                                   // ablation builds keep the round-3 body for their variants
            if (unit < pa.units_q) { project16_body<P16_NT, false, VAR>(pa, smem, 0, 2 * unit + grp, b); }
            else { project16_body<P16_NT, true, VAR>(pa, smem, 0, 2 * (unit - pa.units_q) + grp, b); }
        } else if (unit < pa.units_q) {
            if (grp == 0) project16_body2<P16_G0, false>(pa, smem, 0, unit, b);
            else project16_body2<P16_G1, false>(pa, smem, P16_G0, unit, b);
        } else {
            if (grp == 0) project16_body2<P16_G0, true>(pa, smem, 0, unit - pa.units_q, b);
            else project16_body2<P16_G1, true>(pa, smem, P16_G0, unit - pa.units_q, b);
        }
    } else {
        const int sb = bid - pa.n_full;
        const int grp = sb / (2 * P16_NT), rest = sb - grp * (2 * P16_NT);
        const int half = rest / P16_NT, tile = rest - half * P16_NT;
        project16_body<1, true, VAR>(pa, smem, tile, 2 * (pa.units_k - pa.n_split_groups + grp) + half, pa.batch - 1);
    }
#if !defined(DAGL_P16_PHASES)
    dbg_stamp(pa.times, bid, 3);
#endif
#if defined(DAGL_ABLATION) && defined(DAGL_P16_HWID)
    // (experiment: which CU ran the block -- HW_ID (cu [11:8], sh [12], se [15:13]) | XCC_ID << 16 in place of the prologue stamp)
    if (pa.times != nullptr && threadIdx.x == 0)
        pa.times[(size_t)bid * 4 + 1] = (unsigned long long)(__builtin_amdgcn_s_getreg((31 << 11) | 4) & 0xffffu) |
                                        ((unsigned long long)(__builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xfu) << 16);
#endif
}

// colsum[b][col] = sum over key blocks of colpart[b][blk][col]: one wave per column, lane-strided partial sums
// + a fixed-order butterfly (deterministic), fp64
__global__ __launch_bounds__(256) void colsum_reduce_kernel(int n_blocks_k, const float* __restrict__ colpart,
                                                            double* __restrict__ colsum) {
    const int lane = threadIdx.x & 63;
    const int col = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int b = blockIdx.y;
    if (col >= D) return;
    double t = 0.0;
    for (int k = lane; k < n_blocks_k; k += 64) t += (double)colpart[((size_t)b * n_blocks_k + k) * P16_OUT + col];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o);
    if (lane == 0) colsum[(size_t)b * DS + col] = t;
}

static inline bool p16_keys_linear(const Grid& g) { return g.W >= 32; }       // a wrapped item ends inside the NEXT row
static inline int p16_key_items(const Grid& g) { return p16_keys_linear(g) ? (g.N + 31) / 32 : ((g.W + 31) / 32) * g.H; }
int project16_key_blocks(const Grid& g) { return 2 * ((p16_key_items(g) + P16_UNIT - 1) / P16_UNIT); }   // rows of colpart: two per unit

int launch_project16(hipStream_t s, int B, const Grid& g, int which, const uint16_t* map_hi, const uint16_t* map_lo,
                     const uint16_t* wp_keys, const float* const* bias_keys, float* feat_keys, double* colsum, float* colpart,
                     const uint16_t* wp_q, const float* const* bias_q, float* feat_q, uint16_t* feat_keys_bf16,
                     uint16_t* feat_q_bf16, int heads, RangeTag range, int q_tiled, const Split16Out* split,
                     const ThrHeadSet* thr_hs, int thr_head_imgs, float* thr_part, const B1Tiers* tiers) {
    Proj16Args pa;
    pa.map_hi2 = tiers ? tiers->hi2 : nullptr; pa.map_lo2 = tiers ? tiers->lo2 : nullptr;
    pa.b1_amax = (tiers && tiers->hi2) ? tiers->amax : nullptr; pa.amax_slots = tiers ? tiers->slots : 0;
    static_assert(TB4_LDS_BYTES <= P16_LDS, "the thr / bias blocks live in the projection's LDS");
    static_assert(P16_BW == 4, "thr_bias4_block (thr_bias4.h) is written for blocks of exactly 256 threads: tid + 256 j strides, part[4][..]");
    for (int w = 0; w < 2; ++w) {
        pa.split_hi[w] = split ? split->hi[w] : nullptr; pa.split_lo[w] = split ? split->lo[w] : nullptr;
        pa.rows_alloc_s[w] = split ? split->rows_alloc[w] : 0;
    }
    pa.tiled_h[0] = 0; pa.tiled_h[1] = q_tiled;
    pa.range = range; pa.heads = heads; pa.times = nullptr;
    pa.imgs_per_head = B / heads;
    for (int h = 0; h < 4; ++h) {
        pa.bias[0][h] = bias_keys ? bias_keys[h < heads ? h : 0] : nullptr;
        pa.bias[1][h] = bias_q ? bias_q[h < heads ? h : 0] : nullptr;
    }
    pa.gr = g; pa.map_hi = map_hi; pa.map_lo = map_lo;
    pa.wp[0] = wp_keys; pa.feat[0] = feat_keys; pa.feat_h[0] = feat_keys_bf16;
    pa.wp[1] = wp_q; pa.feat[1] = feat_q; pa.feat_h[1] = feat_q_bf16;
    pa.rows_alloc[0] = feat_rows(g.N); pa.rows_alloc[1] = feat_rows(g.L);
    pa.rows_alloc_h[0] = feat_rows_h(g.N); pa.rows_alloc_h[1] = feat_rows_h(g.L);
    pa.segs[0] = (g.W + 31) / 32; pa.segs[1] = (g.Lw + 31) / 32;
    pa.lin[0] = p16_keys_linear(g) ? 1 : 0; pa.lin[1] = 1;           // (queries gather per lane: any row length)
    pa.n_items[0] = p16_key_items(g); pa.n_items[1] = (g.L + 31) / 32;
    const int uq = (which & 2) ? (pa.n_items[1] + P16_UNIT - 1) / P16_UNIT : 0;
    const int uk = (which & 1) ? (pa.n_items[0] + P16_UNIT - 1) / P16_UNIT : 0;
    const int nbq = 2 * uq, nbk = 2 * uk;
    pa.units_q = uq; pa.units_k = uk;
    pa.n_blocks_q = nbq; pa.n_blocks_k = nbk;
    pa.colpart = (colsum != nullptr) ? colpart : nullptr;
    // resident capacity: two blocks per CU.  A remainder of at most half a round is cut into single-tile blocks.
    static const int cus = [] {
        int dev = 0, n = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        return n > 0 ? n : 256;
    }();
    const int cap = P16_BLOCKS_PER_CU * cus, total = 2 * B * (uq + uk), rem = total % cap;
    int groups = (rem > 0 && rem <= cap / 2) ? (rem + 1) / 2 : 0;        // units whose two blocks overhang the last full round
    if (groups > uk) groups = uk;
    pa.n_split_groups = groups; pa.n_full = total - 2 * groups; pa.batch = B;
    pa.n_proj = pa.n_full + 2 * P16_NT * groups;
    pa.thr_part = nullptr; pa.thr_x = pa.thr_y = 1; memset(&pa.thr_hs, 0, sizeof(pa.thr_hs));
    int n_thr = 0;
    if (thr_hs != nullptr && thr_part != nullptr && thr_head_imgs > 0) {
        pa.thr_hs = *thr_hs; pa.thr_part = thr_part; pa.thr_x = thr_bias4_grid_x(g); pa.thr_y = thr_head_imgs;
        n_thr = pa.thr_x * pa.thr_y * TB_GROUPS;
    }
    const dim3 grid(pa.n_proj + n_thr), block(64 * P16_BW);
#ifdef DAGL_ABLATION      // debug builds only: the variants give wrong results by construction
    if (getenv("DAGL_TIMES_FILE")) pa.times = dbg_times_buffer(grid.x);
    static const int var = getenv("DAGL_P16_VARIANT") ? atoi(getenv("DAGL_P16_VARIANT")) : 0;
    if (var == 1) hipLaunchKernelGGL(project16_kernel<1>, grid, block, 0, s, pa);
    else if (var == 3) hipLaunchKernelGGL(project16_kernel<3>, grid, block, 0, s, pa);
    else if (var == 4) hipLaunchKernelGGL(project16_kernel<4>, grid, block, 0, s, pa);
    else if (var == 5) hipLaunchKernelGGL(project16_kernel<5>, grid, block, 0, s, pa);
    else if (var == 6) hipLaunchKernelGGL(project16_kernel<6>, grid, block, 0, s, pa);
    else if (var == 7) hipLaunchKernelGGL(project16_kernel<7>, grid, block, 0, s, pa);
    else if (var == 8) hipLaunchKernelGGL(project16_kernel<8>, grid, block, 0, s, pa);
    else if (var == 9) hipLaunchKernelGGL(project16_kernel<9>, grid, block, 0, s, pa);
    else if (var == 11) hipLaunchKernelGGL(project16_kernel<11>, grid, block, 0, s, pa);
    else
#endif
    hipLaunchKernelGGL(project16_kernel<0>, grid, block, 0, s, pa);
    DAGL_LAUNCH_CHECK("project16_kernel");
#ifdef DAGL_ABLATION
    if (pa.times) dbg_times_dump(s, "project16_kernel", pa.times, grid.x);
#endif
    if (pa.colpart != nullptr && nbk > 0) {
        hipLaunchKernelGGL(colsum_reduce_kernel, dim3((D + 3) / 4, B), dim3(256), 0, s, nbk, colpart, colsum);
        DAGL_LAUNCH_CHECK("colsum_reduce_kernel");
    }
    return DAGL_OK;
}

}  // namespace dagl
