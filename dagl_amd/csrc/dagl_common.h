// Shared declarations of the device library (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/dagl_ce.h"

namespace dagl {

constexpr int KS = DAGL_KSIZE;        // 7
constexpr int QS = DAGL_QSTRIDE;      // 4
constexpr int CH = DAGL_CH;           // 16 channels of the feature maps
constexpr int P = DAGL_P;             // 784
constexpr int D = DAGL_D;             // 196
constexpr int DS = DAGL_DS;           // 204: feature row stride (floats); 51 16-B slots, odd => conflict-free b128 LDS reads
constexpr int DSH = 216;              // bf16 feature row stride (elements): 432 B = 27 16-B slots (odd)
constexpr int SKEYS = 64;             // keys per step of the bf16 screen
constexpr int DPAD = 208;             // fc output columns rounded up to 13 MFMA tiles of 16
constexpr int PADPIX = DAGL_PADPIX;   // 3
constexpr int KT = 32;                // keys per streaming tile (one 32x32 MFMA tile)
constexpr int QT = 32;                // queries per wave
constexpr float SOFTMAX_SCALE = 10.0f;

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#ifdef __HIPCC__
// LDS-DMA (global_load_lds_dwordx4): each lane copies 16 B from its own global address to
// LDS[lds_byte_addr + lane*16]; lds_byte_addr must be wave-uniform.  Issued from inline asm on purpose:
// hipcc treats its own LDS-DMA builtin as a pending LDS write and drains vmcnt(0) before the next ds_read,
// which serialises the prefetch of tile t+1 with the reads of tile t.  The caller owns the completion wait
// (dma_wait_all() before the barrier that publishes the buffer).
__device__ __forceinline__ void glds16_asm(const float* gsrc, unsigned lds_byte_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_byte_addr)
                 : "memory");
}
// two ADJACENT 1-KiB pieces under ONE M0 value (the second through the instruction's immediate offset, which advances the global
// and the LDS address alike): a wave that changes M0 between two requests waits for the first to leave the issue stage
__device__ __forceinline__ void glds16x2_asm(const float* gsrc, unsigned lds_byte_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
                 "global_load_lds_dwordx4 %1, off offset:1024\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_byte_addr)
                 : "memory");
}
// N pieces whose LDS destinations are 1 KiB apart under ONE M0 value, each with a global address of its own: piece j goes out with the
// immediate offset 1024 j -- which the hardware adds to the LDS AND the global address -- and the lane's address minus 1024 j.  What it
// buys: a wave that changes M0 between two requests waits for the first to leave the issue stage (~240-280 cycles a piece,
// profiles/r03_screen_ring_phases.log, r06_project16_anatomy.log); four requests behind one M0 value issue back to back.
__device__ __forceinline__ void glds16x4_any_asm(const void* s0, const void* s1, const void* s2, const void* s3, unsigned lds_byte_addr) {
    unsigned keep;
    const char* a1 = static_cast<const char*>(s1) - 1024;
    const char* a2 = static_cast<const char*>(s2) - 2048;
    const char* a3 = static_cast<const char*>(s3) - 3072;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
                 "global_load_lds_dwordx4 %2, off offset:1024\n\tglobal_load_lds_dwordx4 %3, off offset:2048\n\t"
                 "global_load_lds_dwordx4 %4, off offset:3072\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(s0), "v"(a1), "v"(a2), "v"(a3), "s"(lds_byte_addr)
                 : "memory");
}
__device__ __forceinline__ void glds16x2_any_asm(const void* s0, const void* s1, unsigned lds_byte_addr) {
    unsigned keep;
    const char* a1 = static_cast<const char*>(s1) - 1024;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
                 "global_load_lds_dwordx4 %2, off offset:1024\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(s0), "v"(a1), "s"(lds_byte_addr)
                 : "memory");
}
// two pieces under one M0 value at LDS offsets OFF0 / OFF1 (< 4096) from the M0 base
template <int OFF0, int OFF1>
__device__ __forceinline__ void glds16x2_off_asm(const void* s0, const void* s1, unsigned lds_byte_addr) {
    static_assert(OFF0 >= 0 && OFF0 < 4096 && OFF1 >= 0 && OFF1 < 4096, "13-bit immediate offset");
    unsigned keep;
    const char* a0 = static_cast<const char*>(s0) - OFF0;
    const char* a1 = static_cast<const char*>(s1) - OFF1;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off offset:%4\n\t"
                 "global_load_lds_dwordx4 %2, off offset:%5\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(a0), "v"(a1), "s"(lds_byte_addr), "n"(OFF0), "n"(OFF1)
                 : "memory");
}
// A whole 27-KiB key tile of the screen (27 pieces of 1 KiB) requested by ONE wave in one statement: wave-uniform source
// base in SGPRs + a 32-bit lane offset, four pieces per M0 value through the instruction's immediate offset (it advances the
// global AND the LDS address), M0 saved / restored once.  The per-piece form above costs the issuing wave ~280 cycles a piece
// in this kernel (phase clocks, profiles/r03_screen_ring_phases.log): every piece waits for the previous one to leave the
// issue stage before M0 may change.
__device__ __forceinline__ void glds_tile27_asm(const void* src_uniform, unsigned lane_byte_off, unsigned lds_byte_addr) {
    unsigned keep;
#define DAGL_G4 "global_load_lds_dwordx4 %1, %2\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024\n\t" \
                "global_load_lds_dwordx4 %1, %2 offset:2048\n\tglobal_load_lds_dwordx4 %1, %2 offset:3072\n\t"
#define DAGL_ADV "v_add_u32 %1, 0x1000, %1\n\ts_add_u32 m0, m0, 0x1000\n\ts_nop 0\n\t"
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 4\n\t"            // (M0 -> LDS-DMA, fresh SGPR base -> VMEM)
                 DAGL_G4 DAGL_ADV DAGL_G4 DAGL_ADV DAGL_G4 DAGL_ADV DAGL_G4 DAGL_ADV DAGL_G4 DAGL_ADV DAGL_G4 DAGL_ADV
                 "global_load_lds_dwordx4 %1, %2\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024\n\t"
                 "global_load_lds_dwordx4 %1, %2 offset:2048\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep), "+v"(lane_byte_off)
                 : "s"(src_uniform), "s"(lds_byte_addr)
                 : "memory");
#undef DAGL_G4
#undef DAGL_ADV
}
// the same tile with EIGHT pieces per M0 value: immediate offsets -4096 .. +3072 around an M0 base four pieces in (the 13-bit offset is
// signed for the LDS address as for the global one: 27 pieces under 4 M0 values instead of 7; the filter pass -0.4 us, r06_ab_m0_pairs.log)
__device__ __forceinline__ void glds_tile27_x8_asm(const void* src_uniform, unsigned lane_byte_off, unsigned lds_byte_addr) {
    unsigned keep;
    lane_byte_off += 4096u;
#define DAGL_G8 "global_load_lds_dwordx4 %1, %2 offset:-4096\n\tglobal_load_lds_dwordx4 %1, %2 offset:-3072\n\t" \
                "global_load_lds_dwordx4 %1, %2 offset:-2048\n\tglobal_load_lds_dwordx4 %1, %2 offset:-1024\n\t" \
                "global_load_lds_dwordx4 %1, %2\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024\n\t" \
                "global_load_lds_dwordx4 %1, %2 offset:2048\n\tglobal_load_lds_dwordx4 %1, %2 offset:3072\n\t"
#define DAGL_ADV8 "v_add_u32 %1, 0x2000, %1\n\ts_add_u32 m0, m0, 0x2000\n\ts_nop 0\n\t"
    asm volatile("s_mov_b32 %0, m0\n\ts_add_u32 m0, %3, 0x1000\n\ts_nop 4\n\t"
                 DAGL_G8 DAGL_ADV8 DAGL_G8 DAGL_ADV8 DAGL_G8 DAGL_ADV8
                 "global_load_lds_dwordx4 %1, %2 offset:-4096\n\tglobal_load_lds_dwordx4 %1, %2 offset:-3072\n\t"
                 "global_load_lds_dwordx4 %1, %2 offset:-2048\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep), "+v"(lane_byte_off)
                 : "s"(src_uniform), "s"(lds_byte_addr)
                 : "memory");
#undef DAGL_G8
#undef DAGL_ADV8
}
__device__ __forceinline__ void dma_wait_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// counted form: returns once at most N of this wave's vector-memory operations are outstanding (loads land in issue order, so
// everything but the N youngest has arrived)
template <int N>
__device__ __forceinline__ void dma_wait_le() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// phase stamps of a block (ablation builds only; 100 MHz constant clock): tools/block_times.py turns them into a timeline
__device__ __forceinline__ void dbg_stamp(unsigned long long* buf, int block, int slot) {
#ifdef DAGL_ABLATION
    if (buf != nullptr && threadIdx.x == 0) buf[(size_t)block * 4 + slot] = __builtin_amdgcn_s_memrealtime();
#endif
}
#ifdef DAGL_ABLATION
unsigned long long* dbg_times_buffer(size_t blocks);                 // lazily allocated device buffer (project16.hip)
void dbg_times_dump(hipStream_t s, const char* kernel, const unsigned long long* buf, size_t blocks);   // env DAGL_TIMES_FILE
#endif
__device__ __forceinline__ unsigned lds_addr_of(const void* p) {
    return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)p;
}
// LDS flag words of the barrier-free rings (screen.hip, project16.hip): 
// flag words are touched through explicit DS instructions: a volatile C++ access through the generic pointer became
// flat_load_dword sc0 sc1 + s_waitcnt vmcnt(0), i.e. every poll drained the wave's LDS-DMA requests and candidate stores
__device__ __forceinline__ unsigned lds_flag_load(unsigned byte_addr) {          // wave-uniform address -> wave-uniform value
    unsigned v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(byte_addr) : "memory");
    return (unsigned)__builtin_amdgcn_readfirstlane((int)v);
}
__device__ __forceinline__ void lds_flag_store(unsigned byte_addr, unsigned val) {
    asm volatile("ds_write_b32 %0, %1" ::"v"(byte_addr), "v"(val) : "memory");
}
__device__ __forceinline__ void lds_flag_add(unsigned byte_addr, unsigned val) {
    asm volatile("ds_add_u32 %0, %1" ::"v"(byte_addr), "v"(val) : "memory");
}

// ---- wave-wide reductions / scan on the DPP path -------------------------------------------------------------------------------
// __shfl_xor / __shfl_up compile to ds_bpermute_b32: every step is a round trip through the LDS crossbar (~100+ cycles of
// dependent latency); a 64-lane reduction is six of them, a refine wave did ~110 per query.  The same data movement as DPP
// modifiers of the ALU instruction itself (quad_perm, row_ror, row_bcast -- gfx9 encodings) costs a few cycles per step.
// ALL 64 lanes of the wave must be active at the call (wave-uniform control flow around it): a disabled source lane hands its
// neighbour the neutral value, not its own.  dagl_selftest_wave_ops checks the lot against a serial evaluation.
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ float dpp_keep_f(float v) {                 // lanes without a source (or in a masked row) keep their own value
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ double dpp_keep_d(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = (int)(unsigned)b, hi = (int)(unsigned)(b >> 32);
    const unsigned l2 = (unsigned)__builtin_amdgcn_update_dpp(lo, lo, CTRL, ROW_MASK, 0xf, false);
    const unsigned h2 = (unsigned)__builtin_amdgcn_update_dpp(hi, hi, CTRL, ROW_MASK, 0xf, false);
    return __longlong_as_double((long long)(((unsigned long long)h2 << 32) | l2));
}
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ int dpp_zero_i(int v) {                     // lanes without a source (or in a masked row) get 0
    return __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xf, false);
}
// maximum over the 64 lanes, the same value in every lane
__device__ __forceinline__ float wave_max_f32(float v) {
    v = fmaxf(v, dpp_keep_f<0xB1>(v));                                // quad_perm 1,0,3,2
    v = fmaxf(v, dpp_keep_f<0x4E>(v));                                // quad_perm 2,3,0,1
    v = fmaxf(v, dpp_keep_f<0x124>(v));                               // row_ror 4
    v = fmaxf(v, dpp_keep_f<0x128>(v));                               // row_ror 8: every lane holds its row's maximum
    v = fmaxf(v, dpp_keep_f<0x142, 0xa>(v));                          // row_bcast 15 into rows 1, 3
    v = fmaxf(v, dpp_keep_f<0x143, 0xc>(v));                          // row_bcast 31 into rows 2, 3
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ double wave_bcast63_d(double v) {
    const long long b = __double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, 63), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), 63);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ double wave_max_f64(double v) {
    v = fmax(v, dpp_keep_d<0xB1>(v)); v = fmax(v, dpp_keep_d<0x4E>(v)); v = fmax(v, dpp_keep_d<0x124>(v)); v = fmax(v, dpp_keep_d<0x128>(v));
    v = fmax(v, dpp_keep_d<0x142, 0xa>(v)); v = fmax(v, dpp_keep_d<0x143, 0xc>(v));
    return wave_bcast63_d(v);
}
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ double dpp_zero_d(double v) {              // lanes without a source (or in a masked row) get 0
    const long long b = __double_as_longlong(v);
    const unsigned l2 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)b, CTRL, ROW_MASK, 0xf, false);
    const unsigned h2 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(b >> 32), CTRL, ROW_MASK, 0xf, false);
    return __longlong_as_double((long long)(((unsigned long long)h2 << 32) | l2));
}
// sum over the 64 lanes (fixed association: pairs, quads, rows of 16, rows 0+1 / 2+3, all), the same value in every lane
__device__ __forceinline__ double wave_sum_f64(double v) {
    v += dpp_zero_d<0xB1>(v); v += dpp_zero_d<0x4E>(v); v += dpp_zero_d<0x124>(v); v += dpp_zero_d<0x128>(v);   // every lane: its row's sum
    v += dpp_zero_d<0x142, 0xa>(v); v += dpp_zero_d<0x143, 0xc>(v);                                         // lane 63: all four rows
    return wave_bcast63_d(v);
}
__device__ __forceinline__ int wave_sum_i32(int v) {                  // the same value in every lane
    v += dpp_zero_i<0xB1>(v); v += dpp_zero_i<0x4E>(v); v += dpp_zero_i<0x124>(v); v += dpp_zero_i<0x128>(v);
    v += dpp_zero_i<0x142, 0xa>(v); v += dpp_zero_i<0x143, 0xc>(v);
    return __builtin_amdgcn_readlane(v, 63);
}
// sum over each aligned group of four lanes (every lane of the group gets it)
__device__ __forceinline__ float quad_sum_f32(float v) {
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, false));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, false));
    return v;
}
// inclusive prefix sum over the lanes
__device__ __forceinline__ int wave_scan_incl_i32(int v) {
    v += dpp_zero_i<0x111>(v);                                        // row_shr 1
    v += dpp_zero_i<0x112>(v);                                        // row_shr 2
    v += dpp_zero_i<0x114>(v);                                        // row_shr 4
    v += dpp_zero_i<0x118>(v);                                        // row_shr 8: inclusive inside each row of 16
    v += dpp_zero_i<0x142, 0xa>(v);                                   // + row 0 (2) total into rows 1 (3)
    v += dpp_zero_i<0x143, 0xc>(v);                                   // + rows 0..1 total into rows 2, 3
    return v;
}

#endif

// thread-local error text
void set_error(const char* fmt, ...);
int hip_fail(hipError_t e, const char* what);

#define DAGL_HIP_TRY(expr)                                         \
    do {                                                           \
        hipError_t _e = (expr);                                    \
        if (_e != hipSuccess) return ::dagl::hip_fail(_e, #expr);  \
    } while (0)

#define DAGL_LAUNCH_CHECK(name)                                    \
    do {                                                           \
        hipError_t _e = hipGetLastError();                         \
        if (_e != hipSuccess) return ::dagl::hip_fail(_e, name);   \
    } while (0)

#define DAGL_REQUIRE(cond, ...)                                    \
    do {                                                           \
        if (!(cond)) {                                             \
            ::dagl::set_error(__VA_ARGS__);                        \
            return DAGL_ERR_INVALID;                               \
        }                                                          \
    } while (0)

// Range guard of the split-fp16 kernels (prologue.hip, project16.hip, dense.hip): operands are pre-scaled by powers of two
// and split into two fp16 numbers, which holds for |256 w_conv|, |1024 w_fc|, |64 feature| < 65504 and -- on the entry points that
// do not carry the map's two tiers (B1Tiers below; round 6) -- |16 x|, |16 b1| < 65504.  A kernel
// that meets a larger value stores the call's tag into the workspace's range word; the last kernel of the call (fold) then
// writes NaN instead of numbers computed from inf / NaN halves, and the host side re-runs the call on the fp32 path where
// it reads statistics back anyway (adaptive modes) or on request (dagl_ce_range_check).
// Blocks are handed to the 8 XCDs round-robin by their linear id (observed; used for speed only): block `bid` of `nblk` -> a logical
// index such that every XCD owns one CONTIGUOUS range of logical indices and walks it in order -- blocks that share an operand
// (the key chunk of the screen, the A tile of a GEMM's column tiles) meet in one L2.  Bijective for any nblk.
__device__ __forceinline__ int xcd_remap2(int bid, int nblk) {
    const int xcd = bid & 7, slot = bid >> 3;
    const int q = nblk >> 3, r = nblk & 7;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + slot;
}

struct RangeTag {
    int32_t* word = nullptr;     // workspace word: tag of the last call that left the fp16 range
    int32_t* done = nullptr;     // workspace word: tag of the last completed call (written by the fold)
    int32_t tag = 0;             // this call's tag (process-wide counter, never 0)
    const int32_t* veto = nullptr;   // workspace word (DAGL_FLAG_NO_WAIT): tag of the last adaptive call whose neighbourhoods the
                                     // in-stream kernels did NOT serve (too many / too heavy overflowed queries): same poison
};
constexpr float RANGE_LIMIT = 60000.0f;

struct Grid {            // geometry of one image
    int H, W;            // feature map
    int Hp, Wp;          // padded map (H+6, W+6)
    int Lh, Lw, L;       // stride-4 query grid
    int N;               // keys = H*W
    int pt, pl;          // SAME padding (top, left) of the stride-4 grid  (dagl.py:123-139)
};

inline Grid make_grid(int H, int W) {
    Grid g;
    g.H = H; g.W = W; g.Hp = H + 2 * PADPIX; g.Wp = W + 2 * PADPIX;
    g.Lh = (H + QS - 1) / QS; g.Lw = (W + QS - 1) / QS; g.L = g.Lh * g.Lw; g.N = H * W;
    int ph = (g.Lh - 1) * QS + KS - H; if (ph < 0) ph = 0;
    int pw = (g.Lw - 1) * QS + KS - W; if (pw < 0) pw = 0;
    g.pt = ph / 2; g.pl = pw / 2;
    return g;
}

inline int round_up(int x, int m) { return (x + m - 1) / m * m; }
inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// rows allocated for a [rows, DS] feature matrix: whole 32-row tiles plus one guard tile so that the
// 1-KiB LDS-DMA pieces of the last tile stay in bounds.
inline int feat_rows(int rows) { return round_up(rows, KT) + KT; }
inline int feat_rows_h(int rows) { return round_up(rows, SKEYS) + SKEYS; }   // bf16 copies (64-row steps)

struct ZeroList {                    // small regions cleared by one launch (16-byte granular), each repeated
    int n = 0;                       // `reps` times at a byte stride (one per image of the batch)
    void* ptr[12];
    size_t bytes[12];
    size_t stride[12];
    int reps[12];
    void add(void* p, size_t b, int r = 1, size_t st = 0) {
        if (b && r > 0) { ptr[n] = p; bytes[n] = b; reps[n] = r; stride[n] = st; ++n; }
    }
};
int launch_zero_regions(hipStream_t s, const ZeroList& z);
int launch_zero_borders(hipStream_t s, int B, int H, int W, float* m1, float* m2);

// ---- stage launchers (defined in the .hip files) ------------------------------------------------
int launch_pad_nhwc(hipStream_t s, int B, int H, int W, const float* src, float* dst);
int launch_pack_fc_weight(hipStream_t s, const float* w, float* wp);
// g / theta convolutions of up to four heads (a CES stage) in one launch: per-head input, packed weights, biases
// (round 6) the split-fp16 range of the activations is per call, not fixed:
//   * the INPUT x is split with a power-of-two scale of the block's own (a convolution is linear: the block scales its result back):
//     a block whose rows leave |16 x| < 60000 runs its strip a second time with 2^e, e from the largest |x| it met -- free for inputs
//     in range, twice the (latency-bound) strip otherwise;
//   * the key / query map b1 = g(x) needs ONE scale per head (a patch projection sums 49 pixels of the map in one accumulator) that is
//     only known once every block has finished: the kernel writes TWO tiers, 16 b1 = hi + lo (|b1| < 3750) and 2^-8 b1 = hi2 + lo2
//     (|b1| < 1.5e7), and each block its largest |b1| into a slot; project16_kernel reduces its head's slots (a wave each: no atomics,
//     nothing to clear, valid under HIP-graph replay) and multiplies the tier that holds the map.  Beyond 1.5e7 -- x of ~1e6 with
//     default-like weights; the reference's own fp32 logits 10 S m ~ b1^4 overflow at b1 ~ 2e8 -- and for non-finite input the range
//     word is set as before (NaN-filled output + sticky report).
constexpr float B1_FINE_SCALE = 16.0f, B1_COARSE_SCALE = 1.0f / 256.0f;
struct B1Tiers {
    uint16_t* hi2 = nullptr; uint16_t* lo2 = nullptr;      // coarse tier maps, same geometry as the fine ones
    float* amax = nullptr;                                  // [heads][slots]: largest |b1| of every wave of every conv block (inf for non-finite values)
    int slots = 0;                                          // 4 x conv blocks per head
};
struct ConvHeadSet { const float* x[4]; const unsigned char* w[4]; const float* gb[4]; const float* tb[4]; int imgs; B1Tiers tiers; };
int conv16_blocks_per_head(const Grid& g, int heads, int imgs);     // 4 x (grid of conv_pair16_kernel / heads): the slots of B1Tiers::amax per head
struct ThrHeadSet { const float* x[4]; const float* thr_w[4]; const float* bias_w[4]; int imgs; };     // thr / bias heads likewise
int launch_thr_bias_heads(hipStream_t s, int heads, int imgs, const Grid& g, const ThrHeadSet& hs,
                          float* thr_part /* per head [4][imgs][L][2] partial sums */);
int launch_conv_pair16_heads(hipStream_t s, int heads, int imgs, const Grid& g, const ConvHeadSet& hs, float* b2p,
                             uint16_t* b1_hi, uint16_t* b1_lo, uint32_t* clear_a, int clear_a_words, uint32_t* clear_b,
                             int clear_b_words, RangeTag range);
int launch_prologue(hipStream_t s, int B, const Grid& g, const float* x, const float* g_w, const float* g_b,
                    const float* th_w, const float* th_b, const float* thr_w, const float* thr_b,
                    const float* bias_w, const float* bias_b, float* b1p /* may be null */, float* b2p, float* thr,
                    float* bias, uint16_t* b1_hi /* optional fp16 split of b1 */, uint16_t* b1_lo,
                    float* thr_part /* scratch [4][B][L][2] floats when thr != null */,
                    bool borders_zero = false /* the maps' 3-pixel borders still hold the zeros of an earlier call */,
                    bool defer_thr_reduce = false /* leave the partial sums in thr_part: launch_query_thresholds finishes them */,
                    uint32_t* clear_a = nullptr, int clear_a_words = 0, uint32_t* clear_b = nullptr, int clear_b_words = 0
                    /* two small per-call regions (counters, flags) cleared by the first block of the conv kernel */,
                    RangeTag range = RangeTag(), const unsigned char* conv_w16 = nullptr /* packed g / theta weights (split-fp16 path) */,
                    bool skip_conv = false /* the g / theta convolutions of all heads were one launch_conv_pair16_heads */,
                    const B1Tiers* tiers = nullptr /* split-fp16 path: the coarse tier maps + the blocks' |b1| slots (see B1Tiers) */);
constexpr size_t CONV_W16_BYTES = (18 + 2) * 16 * 128 + 256;   // packed conv weights of one head + range flag
int launch_pack_conv_weight16(hipStream_t s, const float* g_w, const float* th_w, unsigned char* img);
int launch_zero_borders16(hipStream_t s, int B, int H, int W, uint16_t* m1, uint16_t* m2);
int launch_project(hipStream_t s, int B, const Grid& g, int which /* bit0 keys, bit1 queries */, const float* map,
                   const float* wp_keys, const float* bias_keys, float* feat_keys, double* colsum,
                   const float* wp_q, const float* bias_q, float* feat_q,
                   uint16_t* feat_keys_bf16 = nullptr, uint16_t* feat_q_bf16 = nullptr,
                   RangeTag range = RangeTag() /* set by a non-finite feature: the call's output is NaN-filled, as the reference's is */);
// fp16 split-operand projection (project16.hip)
int launch_split_map(hipStream_t s, size_t n_floats, const float* src, uint16_t* hi, uint16_t* lo, RangeTag range = RangeTag());
int launch_pack_fc_weight16(hipStream_t s, const float* w, uint16_t* wp, bool rows_order = false /* [196][tap][c] instead of [196][c][tap] */);
// optional extra outputs of project16: the features once more as split fp16 (64 x = hi + lo, rows of 216 halfs, columns 196.. zero) --
// what the streamed dense formulation (dense.hip) consumes; index 0 = keys, 1 = queries; [B, rows_alloc, 216]
constexpr float DN_FS = 64.0f;                        // pre-scaling of the split features: 64 x = hi + lo
struct Split16Out { uint16_t* hi[2]; uint16_t* lo[2]; int rows_alloc[2]; };
int launch_project16(hipStream_t s, int B, const Grid& g, int which, const uint16_t* map_hi, const uint16_t* map_lo,
                     const uint16_t* wp_keys, const float* const* bias_keys /*[heads]*/, float* feat_keys, double* colsum,
                     float* colpart, const uint16_t* wp_q, const float* const* bias_q /*[heads]*/, float* feat_q,
                     uint16_t* feat_keys_bf16, uint16_t* feat_q_bf16, int heads = 1, RangeTag range = RangeTag(),
                     int q_tiled = 0 /* bf16 query copy in the screen's fragment order (ScreenArgs::q_tiled) */,
                     const Split16Out* split = nullptr,
                     const ThrHeadSet* thr_hs = nullptr /* with thr_part: the thr / bias heads' partial sums (thr_bias4.h) as extra blocks of this launch */,
                     int thr_head_imgs = 0 /* heads x imgs of those heads */, float* thr_part = nullptr,
                     const B1Tiers* tiers = nullptr /* the map's coarse tier + the conv blocks' |b1| slots: the kernel picks the tier per head */);
int launch_feat_rows_out(hipStream_t s, int B, int n, const float* feat /* [B, feat_rows(n), DS] */, float* rows_out /* [B, n, 196] */,
                         RangeTag range);            // dense copy of the feature rows; NaN when the call left the fp16 range
int project16_key_blocks(const Grid& g);     // key blocks of project16: colpart is [B, key blocks, 224] floats
constexpr size_t P16_PACKED_HALFS = (size_t)49 * 7168 + 1024; // packed fp16 weights (halfs) + read slack; the last 4 bytes = range flag of the weights
// Optional extra duties of the thresholds kernel on the fused path (one launch instead of three): finish the thr / bias
// heads (fixed-order sum of the prologue's four channel-group partials + the conv bias) and derive the screen's
// candidate threshold of the adaptive mode.
struct ThrFuse {
    const float* part = nullptr;            // [heads][4][imgs][L][2] partial sums (thr_bias_kernel), or null = thr/bias are final
    const float* thr_b[4] = {nullptr, nullptr, nullptr, nullptr};
    const float* bias_b[4] = {nullptr, nullptr, nullptr, nullptr};
    int imgs_per_head = 1;
    float* thr_out = nullptr; float* bias_out = nullptr;     // [B,L] final values (written when part != null)
    float* theta_out = nullptr;             // [B,L] adaptive candidate threshold on the screened scores, or null
    float* zero_out = nullptr;              // [B,L] words to clear (the dense formulation's sampled row maxima: ScreenArgs::theta_max), or null
};
// Relative band of the bf16-screened scores (screen.hip): bf16 keeps 8 significant bits, so round-to-nearest moves a value
// by up to 2^-8 relative (attained just above a power of two) and a product of two rounded operands by up to
// (1 + 2^-8)^2 - 1 = 0.0078278; the fp32 accumulation of 196 non-negative terms adds < 196 * 2^-23 = 2.4e-5.  Every
// user of the band (adaptive_theta_of, screen_theta_kernel, refine_kernel, dense_attend_kernel) takes this constant.
constexpr float SCREEN_DELTA = 0.0079f;
// theta of the adaptive modes: S~ >= theta  <=  (S~ (1+DELTA) - mean*thr) + bias > 0, with slack for the fp32 rounding of
// either side (dagl.py:256 evaluates (S - mean*thr) + bias in fp32)
__host__ __device__ inline float adaptive_theta_of(float mt, float bs) {
    const double m = (double)mt, bb = (double)bs;
    double t = (m - bb) - 1e-5 * (fabs(m) + fabs(bb) + 1.0);
    t = t / (1.0 + (double)SCREEN_DELTA);
    t -= 1e-6 * fabs(t);
    return (float)t - 1e-30f;
}
int launch_query_thresholds(hipStream_t s, int B, int L, int N, const float* wq, const double* colsum,
                            const float* thr, float* mt, float* mu_out = nullptr, const ThrFuse* fuse = nullptr);
int launch_unfold_values(hipStream_t s, int B, const Grid& g, const float* b2p, float* rows);
int launch_gather_fixed(hipStream_t s, int L, int k, int P_, const int32_t* idx, const float* wgt,
                        const float* values, float* out);
// out is [B/heads, heads*16, H, W]: batch entry head*imgs + image lands in channels [head*16, head*16+16) of image
int launch_fold(hipStream_t s, int B, const Grid& g, const float* agg, float* out, int heads = 1, RangeTag range = RangeTag());
int launch_stage_mix(hipStream_t s, int B, int HW, const float* cat, const float* x, const float* mix_w,
                     const float* mix_b, float* out);
int launch_scores_dense(hipStream_t s, int B, int L, int N, const float* wq, const float* x, float* sc);

// selection
struct SelectArgs {
    int B, L, N, W;                 // W = feature-map width (key index -> pixel)
    const float* wq;                // [B, feat_rows(L), DS]
    const float* x;                 // [B, feat_rows(N), DS]
    const float* mt;                // [B, L]   mean*thr          (adaptive modes)
    const float* bs;                // [B, L]   bias              (adaptive modes)
    int mode, k;
    int splits;                     // key chunks per query tile
    int tiles_per_split;            // 32-key tiles per chunk
    // single-pass adaptive lists
    int32_t* cnt;                   // [B, L] total passing keys per query (atomic)
    int32_t* seg_cnt;               // [B, L, splits, 2] passing keys per (query, chunk, lane half)
    int32_t* list_idx;              // [B, L, FAST_CAP]        (fast path)   or CSR array (fill pass)
    float*   list_val;              // same shape: raw score S
    const int32_t* seg_rel;         // [B, L, splits, 2] cursor of a segment relative to its row (fill pass)
    const int64_t* row_off;         // [B*L+1] CSR row offsets (fill pass)
    // top-k candidates
    int32_t* cand_idx;              // [B, L, splits*2, k]
    float*   cand_val;
    const int32_t* run_flags;       // optional [B, ceil(L/128)]: only flagged query groups are processed
    const int64_t* run_count;       // with run_flags: number of flagged queries of this call; 0 = the launch exits after one load
};
int launch_score_select(hipStream_t s, const SelectArgs& a, int pass /*0 fast, 1 fill, 2 topk*/);

struct EdgeArgs {
    int B, L, N, mode, k;
    const float* mt; const float* bs;
    // inputs
    const int32_t* cnt;             // [B,L] degree (adaptive)
    const int32_t* list_idx; const float* list_val;      // fast lists [B,L,FAST_CAP] or CSR
    const int64_t* row_off;         // CSR: [B*L+1] offsets (NULL = fast lists)
    const int32_t* cand_idx; const float* cand_val; int splits;   // top-k candidates
    // outputs: per-query neighbour lists with softmax weights
    int32_t* nb_idx; float* nb_wgt; // fast/top-k: [B,L,width] ; CSR: same offsets as input
    int32_t* nb_cnt;                // [B,L] entries used
    int width;
    const int32_t* run_flags;       // optional [B, ceil(L/128)] (top-k merge only)
    const int64_t* run_count;       // with run_flags: number of flagged queries of this call (0 = exit at once)
    float* nb_s;                    // optional [B,L,width]: raw scores of the kept neighbours (saved for backward)
};
int launch_edge_softmax(hipStream_t s, const EdgeArgs& a);
// top-k modes behind the screen: exact scan + merge of the FLAGGED query groups in one launch (select.hip); barrier = a zeroed word
int launch_topk_redo(hipStream_t s, const SelectArgs& a, const EdgeArgs& e, int pass /*2 topk, 3 adaptive-topk*/, unsigned* barrier,
                     int32_t* policy = nullptr /* workspace policy word: set when the pass has work (any flagged query) */);

struct AggArgs {
    int B; Grid g;
    const float* b2p;               // padded NHWC value map [B,Hp,Wp,16]
    const int32_t* nb_idx; const float* nb_wgt; const int32_t* nb_cnt;
    const int64_t* row_off;         // CSR offsets or NULL (fixed width)
    int width;
    float* agg;                     // [B,L,784] (kh,kw,c)
    // DAGL_FLAG_NO_REDO (aggregate_fold_kernel): the call's count of flagged queries -- non-zero: the lists of their groups were never
    // redone, the output is NaN-filled and the sticky word set -- or null
    const int64_t* unserved; int32_t* unserved_sticky;
};
int launch_aggregate_direct(hipStream_t s, const AggArgs& a);
int launch_aggregate_fold(hipStream_t s, const AggArgs& a, float* out, int heads, RangeTag range);   // both at once (fixed-width lists)
int launch_row_stats(hipStream_t s, size_t n_rows, const float* nb_wgt, const int32_t* nb_cnt, const int64_t* row_off,
                     int width, int32_t* deg, float* rowsum);

// bf16 screen + exact refine (screen.hip)
struct ScreenArgs {
    int B, L, N, mode;
    const uint16_t* wqh; const uint16_t* xh;        // bf16 features [B, rows_*h, DSH]
    int q_tiled;                                    // wqh in fragment order: per 32 queries [queries 0..15 | 16..31][t 0..12][half][query]
                                                    // [8 columns] (2 x 6.5 KiB at the tile's row-major place): a wave's 13 fragment
                                                    // loads are two runs of 512 B each instead of 32 rows x 32 B (1.6 us of the
                                                    // screen kernels' 4.6 us prologue)
    int rows_qh, rows_xh;
    int splits, steps_per_split, n_steps, sample;
    int qblock;                     // queries per block: 256 (8 waves, two blocks per CU) or 512 (16 waves, one block per CU)
    float* gmax;                    // pass 0 out: [B, L, splits*2, gkeep] largest group maxima of S~ per segment
    int gkeep;                      // 4 (the lane's four largest) or 16 (all of them: queries with <= 32 segments)
    const float* theta;             // pass 1 in (top-k): per-query candidate threshold on S~
    const float* mt; const float* bs;               // pass 1 in (adaptive modes)
    int capseg;
    // pass 1 out: capseg 8-byte slots per (query, chunk, half) segment: slot 0 = {candidates found, 0}, slot 1 + e = {key,
    // screened score (sign bit set = upper bound only)} of candidate e, stored SLOT-MAJOR inside a query ([slot][segment]): the
    // refine wave (lane = segment) reads a slot of all segments as one contiguous run.  A segment holds capseg - 1 candidates;
    // a larger count means overflow.
    int2* cand;                     // [B, L, capseg, splits*2]
    const int32_t* run_flags;
    int variant;                    // debug ablations (DAGL_SCREEN_VARIANT): 1 no DMA, 2 no MFMA, 4 no epilogue
    unsigned long long* times;      // ablation builds: [blocks][4] 100 MHz stamps (entry, loop start, loop end, exit) or null
    // top-k threshold policy kept on the device (include/dagl_ce.h DAGL_FLAG_TIGHT_TOPK): *policy != 0 -> the kernels take
    // sample_tight / capseg_tight instead of sample / capseg; *gate == 0 -> the launch exits at once (the in-call re-run of a
    // cold workspace).  Null pointers: the host's values as they are.
    const int32_t* policy; const int32_t* gate; int sample_tight, capseg_tight;
    // top-k modes: a segment that is full spills into its QUERY's shared area (one atomic per spilled record -- the rare path): natural
    // images give a few queries one hot segment (a 128-slot segment asked for 160-200) while the query's total stays in the hundreds;
    // without the spill such a query sent its whole 128-query group through the fp32 redo pass (k = 50 at 256^2: 7.2 ms a call)
    int2* spill; unsigned* spill_cnt;       // [B, L, SCREEN_SPILL] records, [B, L] counts (zeroed by screen_theta_kernel); null: none
    int* theta_max;                         // pass 0, top-1 use: [B, L] words (zeroed) that take the query's largest sampled score by an integer
                                            // atomicMax instead of the group maxima going to gmax; pass 1 with seg_max then reads `theta` raw
    int seg_max;                            // pass 1: slot 0 of a segment = {candidates found, largest screened score among them (float bits)}
};
constexpr int SCREEN_SPILL = 256;
int launch_screen(hipStream_t s, const ScreenArgs& a, int pass);
int launch_screen_theta(hipStream_t s, int n_rows, int G, int k, const float* gmax, float* theta, const int32_t* gate = nullptr,
                        unsigned* spill_cnt = nullptr);
// top-k modes, cold workspace: after the first refine -- more than a fiftieth of the query GROUPS are flagged (overflowed candidate slots) under
// the sampled threshold: switch the workspace's policy word to the tight threshold, open the gate of the re-run launches and
// clear what the first pass left in the redo flags and counters
int launch_topk_policy(hipStream_t s, int64_t* stats, int32_t* policy, int32_t* gate, int32_t* redo_flags, int n_flags, long long n_queries);
int launch_adaptive_theta(hipStream_t s, size_t n, const float* mt, const float* bs, float* theta, bool both = false);

struct RefineArgs {
    int B, L, N, mode, k, splits, capseg, width;
    const float* wq; const float* x; int rows_q, rows_x;     // fp32 features
    const float* mt; const float* bs;
    unsigned long long* times;                     // ablation builds: block phase stamps (debug.hip) or null
    const int2* cand;                              // candidate records of the screen (ScreenArgs::cand)
    const float* theta;                            // candidate threshold per query
    int32_t* nb_idx; float* nb_wgt; int32_t* nb_cnt;
    int32_t* redo_flags;            // [B, n_qgroups_exact]: query groups (of 128) the exact kernel must redo
    int n_qgroups_exact;
    int64_t* stats;                 // [3]: total edges, max degree, overflowed queries
    int32_t* ovf_list; int32_t* ovf_count; int ovf_cap;    // adaptive mode: overflowed queries are listed for the per-query redo
    float* ovf_qrows;               // ... and their feature rows copied to [ovf_cap, DS] (OvfArgs::qrows)
    float* nb_s;                    // optional [B,L,width]: raw scores of the kept neighbours (saved for backward)
    int32_t* heavy_list; int32_t* heavy_count;      // adaptive mode: queries with many candidates, refined by a whole block each
    const int32_t* policy; const int32_t* gate; int capseg_tight;     // as in ScreenArgs
    const int2* spill; const unsigned* spill_cnt;                     // as in ScreenArgs
};
int launch_refine(hipStream_t s, const RefineArgs& a);
// dense formulation: per row a score whose logit is the softmax shift of dense_attend_kernel -- the exact row maximum where the
// bf16 bounds are too far apart, their upper bound otherwise (screen.hip rowmax_exact_kernel).  Uses B, L, N, splits, capseg, wq, x,
// rows_q, rows_x, mt, bs, cand (of a filter pass with ScreenArgs::seg_max), theta of the arguments.
int launch_rowmax_exact(hipStream_t s, const RefineArgs& a, float* smax);
int refine_heavy_cap();
int launch_degree_stats(hipStream_t s, size_t n_rows, const int32_t* nb_cnt, int64_t* stats,
                        int count_over = 0 /* > 0: stats[2] = rows with a larger degree */);

// per-query dense redo of the queries that overflowed the screened adaptive lists (overflow.hip)
struct OvfArgs {
    int B; Grid g;
    const float* x; int rows_x;                                // fp32 key features [B, rows, DS]
    const float* mt; const float* bs; const float* b2p;
    const int32_t* list; const int32_t* count; int cap;        // flagged queries (b*L + l), how many (device), list capacity
    const float* qrows;                                        // [cap, DS] feature rows of the flagged queries (written by the refine kernels)
    float* scores; long long ldn;                              // [B, cap, ldn] scores of the flagged queries against all keys (many rows only)
    float* agg; const int32_t* nb_cnt; int32_t* dbg_deg; float* dbg_rowsum;
    float* part;                                               // [cap, OVF_CHUNKS, OVF_PART_FLOATS] per-chunk partial results
    int64_t* edges_run;                                        // device word, zero at the start of the call: edges gathered so far
    int64_t* flagged_edges; long long edge_limit;              // device word: sum of the flagged rows' degrees (the statistics block);
                                                               // beyond the limit the weighted sums are NOT formed (the host sees the
                                                               // same word and sends the call to the dense formulation)
};
constexpr int OVF_CHUNKS = 32;                                 // = ROW_CHUNKS, ROW_PART_FLOATS (row_attend.h): per (row, chunk) a partial
constexpr int OVF_PART_FLOATS = P + 8;                         // weighted sum (784) + {max logit, count, z (double), -}
// flagged rows: scores, masks, weighted sums, combined rows -- and, sharing a launch with the scores, the gather + weighted sum
// over every query's list (`ag`: what launch_aggregate_direct does; flagged rows are skipped); then total edges / largest degree
// of the call (stats[0], [1]) and, for the calls that do not wait, *veto = tag when the host would have had to send the call elsewhere
struct AggArgs;
int launch_overflow_rows(hipStream_t s, const OvfArgs& a, const AggArgs* ag /* null: the caller gathers, then launch_overflow_apply */,
                         size_t n_rows, int64_t* stats, int32_t* veto = nullptr, int32_t tag = 0);
int launch_overflow_apply(hipStream_t s, const OvfArgs& a);     // (calls that wait for their verdict: attend + combine behind the read-back)
int overflow_cap(int N, int B);

// fixed-k neighbourhoods wider than the lists (k > DAGL_MAX_TOPK, topk_wide.hip): row-wise dense form, a batch of queries at a time
struct WideArgs {
    Grid g; int mode, k;
    const float* mt; const float* bs; const float* b2p;
    float* scores; long long ldn;                              // [R, ldn] scores of the batch against all keys of its image
    float* part;                                               // [R, ROW_CHUNKS, ROW_PART_FLOATS]
    int32_t* sel; int32_t* eq_before;                          // [R, 4] threshold key + ties to take; [R, 128] ties before each key range
    float* agg; int32_t* deg; float* rowsum;                   // [B*L, 784], [B*L], [B*L] or null
    int b, r0, R;                                              // image, first query and number of queries of the batch
    int32_t* served;                                           // [R] 1 = the row was done by wide_list_kernel (k <= 1024), or null
    const float* wq; const float* x; int rows_q, rows_x;       // fp32 feature rows [B, rows, DS] (wide_list_kernel: exact scores of the selected keys of rows
                                                               // with logits beyond WL_RESCORE_LOGIT), or null
};
size_t topk_wide_workspace_bytes(int N, int L);
int launch_topk_wide(hipStream_t s, int B, const Grid& g, int mode, int k, const float* wq /* [B, rows, DS] */, const float* x,
                     const float* mt, const float* bs, const float* b2p, void* ws, float* agg, int32_t* deg, float* rowsum,
                     RangeTag range = RangeTag());
// fp32 feature rows [B, rows_in, DS] -> split fp16 rows [B, rows_out, DSH] hi and lo, DN_FS x = hi + lo, columns 196.. zero (dense.hip)
int launch_feat_split(hipStream_t s, int B, int rows, int rows_in, int rows_out, const float* src, uint16_t* hi, uint16_t* lo, RangeTag range,
                      const unsigned* scale_word = nullptr /* the tensor's largest magnitude (launch_absmax): its power-of-two scale instead of DN_FS */);


int launch_row_degree(hipStream_t s, int n_rows, int splits2, const int32_t* seg_cnt, int32_t* seg_rel,
                      int32_t* deg, int64_t* stats /* [2]: total edges, max degree */);
int launch_row_scan(hipStream_t s, int n_rows, const int32_t* deg, int64_t* row_off);
int topk_slots(int k);

// dense adaptive neighbourhoods (dense.hip)
struct DenseArgs {
    int B; Grid g;
    const float* wq; const float* x; int rows_q, rows_x;       // fp32 features [B, rows, DS]
    const uint16_t *x_hi, *x_lo, *wq_hi, *wq_lo; int rows_xh, rows_qh;   // the same, split fp16 [B, rows_h, DSH] (64 x = hi + lo)
    const float* mt; const float* bs;                          // [B,L] mean*thr, bias
    const float* smax;                                         // [B,L] the score whose logit is the row's shift (rowmax_exact_kernel)
    const float* b2p;                                          // padded NHWC value map
    const uint16_t *v_hi, *v_lo;                               // the same, split fp16 (16 v = hi + lo), [B,Hp,Wp,16]
    int splits, tiles_per_split, n_tiles, tiles_per_row;       // 32-key tiles (row aligned), key ranges per 64-query group
    float* part_acc; float* part_m; double* part_z; int32_t* part_deg;   // per (split, query) partial results
    float* m_exact; int32_t* redo_blk; int pass;               // [B,L] exact largest logit of a row (written by the first combine), [B, ceil(L/64)]
                                                               // blocks of 64 queries to run again with it as their shift; pass 0 | 1
    int32_t* redo_count;                                       // how many blocks the first combine flagged (stats[DENSE_RERUN_STAT])
    int variant;                                               // debug ablations (DAGL_DENSE_VARIANT): 1 no A V, 2 no S, 4 no staging, 16 constant
                                                               // weights, 32 no zero-granule skip, 64 phase clocks
    unsigned* phase_out;                                       // ablation builds: [blocks][8 waves][8] shader clocks per phase, or null
};
size_t dense_workspace_bytes(int B, const Grid& g);
Split16Out dense_split_buffers(void* dense_ws, int B, const Grid& g);    // where launch_dense_attend expects the split features
void dense_guard_rows(ZeroList& zl, int B, const Grid& g, const Split16Out& so);   // the rows past N / L (guard tiles) must be zero
constexpr int DENSE_RERUN_STAT = 11;                                     // word of the call's statistics block that counts the re-run blocks
int launch_dense_attend(hipStream_t s, int B, const Grid& g, const float* wq, const float* x, const float* mt,
                        const float* bs, const float* smax, const float* b2p, void* ws, float* agg, int32_t* deg_out,
                        float* rowsum_out, int64_t* stats /* [0] += edges, [1] = max degree */, RangeTag range = RangeTag(),
                        float* lse_out = nullptr /* [B,L,2] {shift M, sum Z} for the backward */,
                        bool features_split = false /* the projection already wrote dense_split_buffers() */,
                        bool want_stats = true /* stats[0..2]: total edges, largest degree, rows beyond the lists' width */);

// graph-core backward (backward.hip)
struct BwdArgs {
    int B; Grid g; int mode, width;
    const float* wq_rows; const float* x_rows;          // dense feature rows [B,L,196], [B,N,196]
    const float* b2p;                                   // padded NHWC value map [B,Hp,Wp,16]
    const float* thr; const float* bs; const float* mu; // per query [B,L] (adaptive modes): threshold, bias, row mean
    const double* colsum;                               // [B,204] column sums of the key features (adaptive modes)
    const int32_t* nb_idx; const float* nb_wgt; const float* nb_s; const int32_t* nb_cnt;
    const float* dout;                                  // [B,16,H,W]
    float* dagg;                                        // ws [B,L,784]
    float* dS;                                          // ws [B,L,width]
    float* dmu;                                         // ws [B,L]
    float* dwq_rows; float* dx_rows; float* dthr; float* dbias;          // outputs
};
struct BwdSortWs {                                      // edges sorted by key (stable radix sort)
    uint32_t *keys_in, *keys_out, *vals_in, *vals_out;  // [B*L*width] each
    uint32_t* seg;                                      // [B*N, 2] run of every key in the sorted order
    void* temp; size_t temp_bytes;
    float* rowbuf;                                      // [B*L*width, 980] per-key sums (row of a key = first position of its run)
    float* part;                                        // [chunks, 2, 980] partial rows of runs that cross chunk boundaries
};
size_t edge_sort_temp_bytes(size_t n_edges, size_t n_keys);
size_t edge_rowbuf_floats(size_t n_edges);
size_t edge_part_floats(size_t n_edges);
int launch_core_backward(hipStream_t s, const BwdArgs& a, const BwdSortWs& w, float* dxbar_ws, float* db2_nchw);
int launch_rows_to_feat(hipStream_t s, int B, int rows, const float* src, float* feat, uint16_t* feat_h,
                        RangeTag range = RangeTag() /* a non-finite feature (NaN-filled by a projection that left its range)
                                                       poisons the call: a NaN score would silently drop out of the selection */);

// batched fp32 GEMM on the matrix cores (gemm32.hip): C[b] = alpha * A[b] B[b] + beta * C[b] (+ bias[n], relu)
struct Gemm32 {
    int M, N, K, batch;
    const float* A; long long lda, sA; int a_kc;      // a_kc: element (m,k) at A[m*lda + k], else at A[k*lda + m]
    const float* B; long long ldb, sB; int b_kc;      // b_kc: element (k,n) at B[n*ldb + k], else at B[k*ldb + n]
    float* C; long long ldc, sC;
    float alpha, beta;
    const float* bias; int relu;
    int chunk_tiles = 0;                              // > 0: partial sums of chunk_tiles * 16 products, added in fp32
    int slices = 1; float* scratch = nullptr;         // split-K (batch == 1): slices * M * N floats of scratch, fixed-order sum
    int k_total = 0;                                  // (set by launch_gemm32)
    const int32_t* m_limit = nullptr;                 // optional device word: only rows < *m_limit are computed ...
    int m_limit_floor = 0, m_limit_ceil = 0x7fffffff; // ... and none at all when the word is <= floor or > ceil
    int variant = 0;                                  // ablation builds (DAGL_GEMM_VARIANT): 1 no global loads, 2 no LDS stores, 4 no barrier, 8 no MFMA
};
int launch_gemm32(hipStream_t s, const Gemm32& g);
int gemm32_auto_slices(int M, int N, int K);

int launch_col_sum_final(hipStream_t s, int n_part, int cols, const double* part, float* out);
// ---- split-fp16 GEMM of the training products (gemm16s.hip): C[m][n] = alpha sum_k A[m][k] B[n][k], both operands K-contiguous
// fp16 pairs (s x = hi + lo), hi*hi + hi*lo + lo*hi in one fp32 accumulator ---------------------------------------------------
#ifdef __HIPCC__
__device__ __forceinline__ void g16_split(float a, unsigned short& hi, unsigned short& lo) {
    const _Float16 h = (_Float16)a;
    const _Float16 l = (_Float16)(a - (float)h);
    hi = __builtin_bit_cast(unsigned short, h);
    lo = __builtin_bit_cast(unsigned short, l);
}
// power of two that brings a tensor's largest magnitude into [2^13, 2^14): s = 2^(13 - floor(log2 max)); 1 for an all-zero tensor
__host__ __device__ __forceinline__ float fcg_scale_of(unsigned max_bits) {
    if (max_bits == 0u || max_bits >= 0x7f800000u) return 1.0f;          // zero (or no word), inf / NaN: nothing to rescue
    const int e = (int)(max_bits >> 23) - 127;                          // floor(log2 max) for normal numbers
    int se = 13 - e;
    if (se > 120) se = 120;
    if (se < -120) se = -120;
    union { unsigned u; float f; } cv; cv.u = (unsigned)(se + 127) << 23;
    return cv.f;
}
#endif
struct Gemm16s {
    int M, N;                                   // real output extent (stores are clipped to it)
    int K;                                      // contraction length of ONE slice (multiple of 32)
    const unsigned short *a_hi, *a_lo;          // [>= M rounded up to 128][lda] halfs, K-contiguous
    const unsigned short *b_hi, *b_lo;          // [>= N rounded up to 128][ldb]
    long long lda, ldb;                         // leading dimensions in halfs (multiples of 8)
    int a_rows, b_rows;                         // rows that exist: a tile's loads are clamped to them (clipped outputs only)
    float* C; long long ldc;                    // slices == 1: C[m][n] = alpha acc
    float* part;                                // slices > 1: part[slice][batch][M][N] raw accumulators
    int slices;
    const unsigned* scale_word; float alpha0;   // alpha = alpha0 / (fcg_scale_of(*scale_word) fcg_scale_of(*scale_word_b)); a null word = 1
    const unsigned* scale_word_b = nullptr;
    int batch = 1; long long sA = 0, sB = 0, sC = 0;   // batch > 1: operand strides in halfs, output stride in floats (grid.z = batch x slices)
    int k_valid = 0;                            // > 0: halfs of an operand row that exist inside a slice (multiple of 8, < K): 16-byte pieces from
                                                // there on are fetched from columns k_valid - 8 .. of the same row instead, which the caller
                                                // guarantees to be ZERO (feature rows of 216 halfs under a contraction of 224)
    int nt_store = 0;                           // results as streaming stores (set by launch_gemm16s for results beyond the caches)
    // (round 6) d rows of the stride-1 patch projection folded on the way out (gemm16s.hip fold_tile): with the weight's columns laid out
    // [kh][kw 8 (7 + a zero pad)][c 16], column tile kh of a row tile of 128 consecutive patches is one kernel ROW of their patches: the block
    // adds it over kw in the LDS (fixed order) and writes [tile][kh][rows of the tile][seg + 6][16] partial rows instead of 128 x 112 results
    // -- 61 MB instead of 411 MB at n = 131 072 --, which fcg_fold_kh_kernel adds over kh.  fold_seg = patches of an image row a tile
    // holds (min(ow, 128)), fold_rows = image rows per tile (128 / fold_seg)
    float* fold_part = nullptr; int fold_seg = 0, fold_rows = 0;
    // (round 6) the weight gradient's second operand straight from SHIFTED PLANES of the map instead of the transposed patch rows
    // (fcg_shift_planes_kernel: planes[kw][c][b Hp + y][x] = 16 map[b, y, x + kw, c], x < W = 2^im_logw; 61 MB instead of 470 MB at
    // [8, 128, 128]): row n = (kh, kw, c) of B, contraction index k = patch (b, py, px) -> planes[kw][c][b Hp + py + kh][px].  A slice
    // (K patches) must be whole image rows of ONE image: the block derives (b, py) of its first patch once, no per-lane division
    int b_implicit = 0, im_logw = 0, im_H = 0, im_Hp = 0; long long im_plane = 0;       // im_plane = B Hp W halfs per (kw, c) plane
    int n_loop = 1;                             // column tiles a block walks one after the other (slices == 1; set by launch_gemm16s): short
                                                // contractions (d rows of the projections: K = 224 = 7 steps) are one operand pipeline of
                                                // n_loop x K / 32 steps per block instead of a request latency + 7 steps + 64 KiB of stores
};
int launch_gemm16s(hipStream_t s, const Gemm16s& g);
int launch_absmax(hipStream_t s, size_t n, const float* x, unsigned* word);     // *word = max(*word, bits of max |x|); n % 4 == 0
   // out[c] = sum of part[p][c], fixed order
int launch_unfold_dout(hipStream_t s, int B, const Grid& g, const float* dout, float* dagg);
int launch_dxbar(hipStream_t s, int B, int L, const float* wq_rows, const float* dmu, float* dxbar);
// dense neighbourhoods under autograd (dense_train.hip): the dense formulation chunked over queries
size_t dense_train_workspace_bytes(int B, const Grid& g, bool backward);
// any patch geometry (generic.hip; ABI 406)
size_t ce_generic_workspace_bytes(int B, int Cin, int H, int W, int ks, int s1, int s2, int C);
int ce_generic_check(int B, int Cin, int H, int W, int ks, int s1, int s2, int C, int mode, int k);
int launch_ce_generic(hipStream_t s, int B, int Cin, int H, int W, int ks, int s1, int s2, int C, float scale, int mode, int k,
                      const float* x, const float* g_w, const float* g_b, const float* th_w, const float* th_b, const float* thr_w,
                      const float* thr_b, const float* bias_w, const float* bias_b, const float* fc1_w, const float* fc1_b,
                      const float* fc2_w, const float* fc2_b, float* out, int32_t* degree, void* workspace);
size_t ce_generic_core_workspace_bytes(int B, int H, int W, int ks, int s1, int s2, int C, int backward);
int ce_generic_border(int ks);
int launch_ce_generic_core_forward(hipStream_t s, int B, int H, int W, int ks, int s1, int s2, int C, float scale, int mode, int k,
                                   const float* wq, const float* x, const float* b2p, const float* thr, const float* bias, float* out,
                                   int32_t* degree, void* workspace);
int launch_ce_generic_core_backward(hipStream_t s, int B, int H, int W, int ks, int s1, int s2, int C, float scale, int mode, int k,
                                    const float* wq, const float* x, const float* b2p, const float* thr, const float* bias, const float* d_out,
                                    float* d_wq, float* d_x, float* d_b2p, float* d_thr, float* d_bias, void* workspace);
int launch_dense_train_forward(hipStream_t s, int B, const Grid& g, const float* wq_rows, const float* x_rows, const float* b2,
                               const float* thr, const float* bias, float* out, float* lse /*[B,L,2]*/, float* mu /*[B,L]*/,
                               void* ws, size_t ws_bytes, int64_t* stats_dev /* [2]: edges, max degree */,
                               int mode = DAGL_MODE_ADAPTIVE, int k = 0 /* the wide top-k modes: DAGL_MODE_TOPK (thr / bias null) / _ADAPTIVE_TOPK;
                                                                          lse / mu may be null (kept in the workspace) */);
int launch_dense_train_backward(hipStream_t s, int B, const Grid& g, const float* wq_rows, const float* x_rows, const float* b2,
                                const float* thr, const float* bias, const float* lse, const float* mu, const float* dout,
                                float* dwq_rows, float* dx_rows, float* db2, float* dthr, float* dbias, void* ws, size_t ws_bytes,
                                bool fp32_products = false,      // false: the five products on the fp16 matrix cores, split operands (one-chunk shapes)
                                int mode = DAGL_MODE_ADAPTIVE, int k = 0);
int launch_colsum_rows(hipStream_t s, int B, int N, const float* rows, double* colsum);              // per-lane list length used for a requested k (4/8/16/32)

}  // namespace dagl
