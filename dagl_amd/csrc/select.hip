// Similarity + neighbour selection, streamed: S = Wq . X^T is produced 32x32 tile by tile on the fp32
// matrix cores and consumed in registers; the [L,N] score matrix of the reference
// (torch.matmul, DN_Gray/model/dagl.py:250: 1 GiB at 256x256) and its ~6 elementwise passes
// (:256-261) never reach HBM.
//
// Per wave: 32 queries held as MFMA B fragments in registers (100 VGPRs), key tiles of 32 rows x 204
// floats streamed through LDS by LDS-DMA (global_load_lds_dwordx4, double buffered) and shared by the
// 4 waves (= 4 query tiles) of a block.  Operands are swapped -- D[key][query] = mfma(A = keys,
// B = queries) -- so that one lane ends up with 16 scores of ONE query: the adaptive threshold / the
// running k-th best of that query is a lane-local register, no cross-lane traffic in the hot loop.
//
// v_mfma_f32_32x32x2_f32 is an exact fp32 fma chain (k-ordered), so S here has the same rounding class
// as the reference's fp32 GEMM.  K order is permuted inside each group of 8 (k-slot h of MFMA m of
// group t <-> k = 8t + 4h + m) so that one ds_read_b128 per lane feeds 4 MFMAs.
//
// Passes (template PASS):
//   0  adaptive, single pass: keys with relu(S - mean*thr + bias) != 0 (dagl.py:256-257) are appended
//      to the query's list (<= DAGL_FAST_CAP slots, atomic cursor) and counted per (query, chunk, half)
//   1  adaptive, CSR fill: same test, written at precomputed per-lane cursors (no atomics, any degree)
//   2  top-k: per-lane sorted k-best lists (GReccR2b_3mh_1-checkpoint.py:242-246 semantics)
//   3  adaptive AND top-k
#include "dagl_common.h"
#include "topk_merge.h"

namespace dagl {

constexpr int SEL_WAVES = 4;
constexpr int TILE_FLOATS = KT * DS;             // 6528 floats = 26112 B of key features per tile
constexpr int TILE_PIECES = 26;                  // 1-KiB DMA pieces per tile (the last one is half used)
constexpr int TILE_LDS = TILE_PIECES * 256;      // floats reserved per LDS buffer
constexpr int KG = 25;                           // groups of 8 k-values (200 = 196 + 4 zeros)
constexpr int KCH = 5;                           // accumulation chunks (5 groups = 40 terms each)

// XCD-aware, bijective block remap: the blocks an XCD receives (bid % 8 == xcd) form one contiguous
// range of logical ids, so blocks that stream the same key chunk share that XCD's L2.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    const int xcd = bid & 7, slot = bid >> 3;
    const int q = nblk >> 3, r = nblk & 7;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + slot;
}

template <int K>
struct TopK {
    float v[K];
    int id[K];
    __device__ __forceinline__ void init() {
#pragma unroll
        for (int t = 0; t < K; ++t) { v[t] = -1.0f; id[t] = -1; }
    }
    __device__ __forceinline__ float kth() const { return v[K - 1]; }
    // insert (s, j) into the descending list; caller guarantees s > v[K-1]
    __device__ __forceinline__ void insert(float s, int j) {
#pragma unroll
        for (int t = K - 1; t > 0; --t) {
            const bool up = s > v[t - 1];            // new element goes above slot t-1: shift it down
            const bool here = !up && (s > v[t]);
            const float nv = up ? v[t - 1] : (here ? s : v[t]);
            const int ni = up ? id[t - 1] : (here ? j : id[t]);
            v[t] = nv; id[t] = ni;
        }
        if (s > v[0]) { v[0] = s; id[0] = j; }
    }
};

template <int PASS, int K>
__device__ __forceinline__ void score_select_unit(const SelectArgs& a, float (*sK)[TILE_LDS], int n_qgroups, int n_tiles,
                                                  int rows_q, int rows_x, int logical) {
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int i = lane & 31, h = lane >> 5;
    const int b = blockIdx.y;

    const int split = logical / n_qgroups;
    const int qg = logical % n_qgroups;
    const int tile0 = split * a.tiles_per_split;
    int tile1 = tile0 + a.tiles_per_split;
    if (tile1 > n_tiles) tile1 = n_tiles;

    const int q = (qg * SEL_WAVES + wave) * QT + i;           // this lane's query
    const bool qvalid = q < a.L;
    const int qc = qvalid ? q : a.L - 1;

    // query fragments: qf[t] = Wq[q][8t+4h .. 8t+4h+3]
    float4 qf[KG];
    {
        const float* qp = a.wq + ((size_t)b * rows_q + qc) * DS + 4 * h;
#pragma unroll
        for (int t = 0; t < KG; ++t) qf[t] = *reinterpret_cast<const float4*>(qp + 8 * t);
        // consume the loads here, so that hipcc's lazy vmcnt waits for them sit before the loop and not
        // between the MFMAs of every tile (where they would also wait for the in-flight LDS-DMA)
#pragma unroll
        for (int t = 0; t < KG; ++t)
            asm volatile("" : "+v"(qf[t].x), "+v"(qf[t].y), "+v"(qf[t].z), "+v"(qf[t].w));
    }
    float mtq = 0.f, bsq = 0.f;
    if (PASS != 2) {
        mtq = a.mt[(size_t)b * a.L + qc];
        bsq = a.bs[(size_t)b * a.L + qc];
    }

    const float* xb = a.x + (size_t)b * rows_x * DS;
    const size_t qlin = (size_t)b * a.L + qc;                 // linear query id
    const size_t seg = (qlin * a.splits + split) * 2 + h;     // (query, chunk, half) segment id

    int n_loc = 0;                                            // passing keys seen by this lane
    int64_t cursor = 0;
    if (PASS == 1) cursor = a.row_off[qlin] + a.seg_rel[seg];
    TopK<(PASS >= 2) ? K : 1> best;
    best.init();

    // prologue: first tile -> buffer 0
    const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr_of(&sK[0][0]));
    for (int p = wave; p < TILE_PIECES; p += SEL_WAVES)
        glds16_asm(xb + (size_t)tile0 * TILE_FLOATS + p * 256 + lane * 4,
                   __builtin_amdgcn_readfirstlane(lds0 + p * 1024));
    dma_wait_all();
    __syncthreads();

    for (int tile = tile0; tile < tile1; ++tile) {
        const int cur = (tile - tile0) & 1;
        if (tile + 1 < tile1) {
            const unsigned dst = lds0 + (cur ^ 1) * (TILE_LDS * 4);
            for (int p = wave; p < TILE_PIECES; p += SEL_WAVES)
                glds16_asm(xb + (size_t)(tile + 1) * TILE_FLOATS + p * 256 + lane * 4,
                           __builtin_amdgcn_readfirstlane(dst + p * 1024));
        }
        // The 196-term dot product is accumulated in KCH chunks of 40 terms (each an fp32 fma chain on the
        // matrix core) that are then added together: the rounding error of a sequential sum grows ~n^2/2
        // in variance, so chunking cuts it ~2x -- the logits 10*S*m amplify S errors by up to ~100x, and
        // this keeps the block inside the reference's own fp32 noise ("error budget").
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const float* kp = &sK[cur][i * DS + 4 * h];
#pragma unroll
        for (int c = 0; c < KCH; ++c) {
            f32x16 part;
#pragma unroll
            for (int r = 0; r < 16; ++r) part[r] = 0.f;
#pragma unroll
            for (int t = c * (KG / KCH); t < (c + 1) * (KG / KCH); ++t) {
                const float4 kf = *reinterpret_cast<const float4*>(kp + 8 * t);
                part = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.x, qf[t].x, part, 0, 0, 0);
                part = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.y, qf[t].y, part, 0, 0, 0);
                part = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.z, qf[t].z, part, 0, 0, 0);
                part = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.w, qf[t].w, part, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] += part[r];
        }
        // acc[r] = S[key = tile*32 + (r&3) + 8*(r>>2) + 4*h][query q]
        const int kbase = tile * KT + 4 * h;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float s = acc[r];
            const int key = kbase + (r & 3) + 8 * (r >> 2);
            bool pass = qvalid && (key < a.N);
            if (PASS != 2) {
                const float m = (s - mtq) + bsq;              // same expression order as dagl.py:256
                pass = pass && (m > 0.f);
            }
            if (PASS == 0) {
                if (pass) {
                    ++n_loc;
                    if (n_loc <= DAGL_FAST_CAP) {
                        const int pos = atomicAdd(&a.cnt[qlin], 1);
                        if (pos < DAGL_FAST_CAP) {
                            a.list_idx[qlin * DAGL_FAST_CAP + pos] = key;
                            a.list_val[qlin * DAGL_FAST_CAP + pos] = s;
                        }
                    }
                }
            } else if (PASS == 1) {
                if (pass) {
                    a.list_idx[cursor] = key;
                    a.list_val[cursor] = s;
                    ++cursor;
                }
            } else {
                pass = pass && (s > best.kth());
                if (__any(pass)) {
                    if (pass) best.insert(s, key);
                }
            }
        }
        dma_wait_all();          // next tile landed (it had the whole tile's MFMA time)
        __syncthreads();
    }

    if (PASS == 0) {
        if (qvalid) a.seg_cnt[seg] = n_loc;
    } else if (PASS >= 2) {
        if (qvalid) {
            const size_t o = seg * K;
#pragma unroll
            for (int t = 0; t < ((PASS >= 2) ? K : 1); ++t) {
                a.cand_idx[o + t] = best.id[t];
                a.cand_val[o + t] = best.v[t];
            }
        }
    }
}

// Kernel: one (query group, key chunk) unit per block -- or, as the REDO pass behind the screen (run_flags given, almost
// always nothing flagged), a small grid whose blocks walk over the units and skip the unflagged ones: an empty redo costs
// one wave of flag reads instead of the dispatch of a thousand blocks that exit.
template <int PASS, int K>
__global__ __launch_bounds__(256, (K > 32) ? 1 : 2) void score_select_kernel(SelectArgs a, int n_qgroups, int n_tiles,
                                                              int rows_q, int rows_x, int n_units) {
    __shared__ __attribute__((aligned(16))) float sK[2][TILE_LDS];      // 52 KiB
    if (a.run_flags == nullptr) {
        score_select_unit<PASS, K>(a, sK, n_qgroups, n_tiles, rows_q, rows_x, xcd_remap(blockIdx.x, gridDim.x));
        return;
    }
    if (a.run_count != nullptr && *a.run_count == 0) return;           // nothing flagged anywhere (the usual case)
    for (int unit = blockIdx.x; unit < n_units; unit += gridDim.x) {
        if (a.run_flags[blockIdx.y * n_qgroups + unit % n_qgroups] == 0) continue;  // block-uniform
        score_select_unit<PASS, K>(a, sK, n_qgroups, n_tiles, rows_q, rows_x, unit);
        __syncthreads();
    }
}

template <int PASS>
static int launch_pass(hipStream_t s, const SelectArgs& a, dim3 grid, int n_qgroups, int n_tiles, int rows_q,
                       int rows_x) {
    dim3 block(256);
    const int n_units = (int)grid.x;
    if (a.run_flags != nullptr && grid.x > 128) grid.x = 128;            // redo pass: blocks walk over the units
    if (PASS < 2) {
        hipLaunchKernelGGL((score_select_kernel<PASS, 1>), grid, block, 0, s, a, n_qgroups, n_tiles, rows_q,
                           rows_x, n_units);
    } else {
        switch (topk_slots(a.k)) {
            case 4:
                hipLaunchKernelGGL((score_select_kernel<PASS, 4>), grid, block, 0, s, a, n_qgroups, n_tiles,
                                   rows_q, rows_x, n_units);
                break;
            case 8:
                hipLaunchKernelGGL((score_select_kernel<PASS, 8>), grid, block, 0, s, a, n_qgroups, n_tiles,
                                   rows_q, rows_x, n_units);
                break;
            case 16:
                hipLaunchKernelGGL((score_select_kernel<PASS, 16>), grid, block, 0, s, a, n_qgroups, n_tiles,
                                   rows_q, rows_x, n_units);
                break;
            case 32:
                hipLaunchKernelGGL((score_select_kernel<PASS, 32>), grid, block, 0, s, a, n_qgroups, n_tiles,
                                   rows_q, rows_x, n_units);
                break;
            default:       // 33..64: per-lane lists of 64 (128 registers) next to the query fragments: one wave per SIMD (512
                           // registers); this scan only serves images under 2048 keys, scan = "exact" and the screen's rare redo
                hipLaunchKernelGGL((score_select_kernel<PASS, 64>), grid, block, 0, s, a, n_qgroups, n_tiles,
                                   rows_q, rows_x, n_units);
                break;
        }
    }
    DAGL_LAUNCH_CHECK("score_select_kernel");
    return DAGL_OK;
}

int topk_slots(int k) { return k <= 4 ? 4 : k <= 8 ? 8 : k <= 16 ? 16 : k <= 32 ? 32 : 64; }

int launch_score_select(hipStream_t s, const SelectArgs& a, int pass) {
    const int n_qgroups = (a.L + SEL_WAVES * QT - 1) / (SEL_WAVES * QT);
    const int n_tiles = (a.N + KT - 1) / KT;
    dim3 grid(n_qgroups * a.splits, a.B);
    const int rows_q = feat_rows(a.L), rows_x = feat_rows(a.N);
    switch (pass) {
        case 0: return launch_pass<0>(s, a, grid, n_qgroups, n_tiles, rows_q, rows_x);
        case 1: return launch_pass<1>(s, a, grid, n_qgroups, n_tiles, rows_q, rows_x);
        case 2: return launch_pass<2>(s, a, grid, n_qgroups, n_tiles, rows_q, rows_x);
        case 3: return launch_pass<3>(s, a, grid, n_qgroups, n_tiles, rows_q, rows_x);
    }
    set_error("launch_score_select: bad pass %d", pass);
    return DAGL_ERR_INVALID;
}

// REDO pass of the top-k modes behind the screen, both halves in ONE launch: the exact scan of the flagged query groups
// (score_select_kernel's redo form: per-(chunk, half) k-best lists) and, behind a grid barrier, the merge of those lists into
// the neighbour lists (edge_softmax_topk_kernel's redo form).  Almost always nothing is flagged and the launch exits after one
// load -- as two launches that cost two launch + load latencies (~3 us each) on every forward.  The barrier: at most 128 blocks of
// 256 threads (one or two fit a CU): all resident (blocks of other streams' kernels only delay them); the word is zeroed per call.
template <int PASS, int K>
__global__ __launch_bounds__(256, (K > 32) ? 1 : 2) void topk_redo_kernel(SelectArgs a, EdgeArgs e, int kslots, int n_qgroups, int n_tiles,
                                                                         int rows_q, int rows_x, int n_units, unsigned n_merge_blocks,
                                                                         unsigned* barrier, int32_t* policy) {
    __shared__ __attribute__((aligned(16))) float sK[2][TILE_LDS];      // 52 KiB; the merge's candidate arrays (32 KiB) reuse it
    static_assert(sizeof(float) * 2 * TILE_LDS >= (sizeof(float) + sizeof(int)) * 4 * TOPK_MAX_CAND, "merge arrays fit the scan's tiles");
    if (*a.run_count == 0) return;                                       // nothing flagged anywhere (the usual case)
    // the workspace's threshold policy (include/dagl_ce.h DAGL_FLAG_TIGHT_TOPK): this pass has work under the sampled threshold -> the
    // tight threshold from the next call on (sticky; this call pays the exact scan of its flagged groups once).  ANY flagged query is
    // enough: it sends its whole 128-query group through the fp32 scan of all keys (84 us per group at 256^2, 0.3 ms at 512^2) where the
    // tight threshold costs 25 us per call at 256^2 and nothing on smaller or larger maps.  (Up to round 4's last day the test was "more
    // than an eighth of the QUERIES": natural-image maps whose few hundred flagged queries are spread over most groups -- two of the
    // five 512^2 Set12 maps, leaf-tile batches -- stayed on the redo pass for good: 36.7 ms instead of 2.4, 1.1-2.0 instead of 0.64.)
    if (policy != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *policy = 1;
    bool wrote = false;
    for (int unit = blockIdx.x; unit < n_units; unit += gridDim.x) {
        if (a.run_flags[blockIdx.y * n_qgroups + unit % n_qgroups] == 0) continue;  // block-uniform
        score_select_unit<PASS, K>(a, sK, n_qgroups, n_tiles, rows_q, rows_x, unit);
        wrote = true;
        __syncthreads();
    }
    // every block's lists are in memory before any block merges.  A device-scope fence writes back / invalidates the XCD's whole L2
    // (~0.2 us each, serialised per XCD: 512 of them cost 110 us): release only by blocks that wrote lists, acquire (below) only
    // by blocks that are about to read some
    __syncthreads();
    if (threadIdx.x == 0) {
        if (wrote) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        const unsigned total = gridDim.x * gridDim.y;
        __hip_atomic_fetch_add(barrier, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while (__hip_atomic_load(barrier, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < total) {
            __builtin_amdgcn_s_sleep(8);
            if (++spins > (1u << 23)) __builtin_trap();                 // (seconds: a block that never became resident -- fail loudly, do not hang)
        }
    }
    __syncthreads();
    bool acquired = false;
    float (*cv)[TOPK_MAX_CAND] = reinterpret_cast<float (*)[TOPK_MAX_CAND]>(&sK[0][0]);
    int (*ci)[TOPK_MAX_CAND] = reinterpret_cast<int (*)[TOPK_MAX_CAND]>(&sK[0][0] + 4 * TOPK_MAX_CAND);
    const int nqg = (e.L + 127) / 128;
    for (size_t blk = blockIdx.y * gridDim.x + blockIdx.x; blk < n_merge_blocks; blk += (size_t)gridDim.x * gridDim.y) {
        // block-uniform skip: none of the block's 4 queries sits in a flagged group
        const size_t q0 = blk * 4, q1 = (q0 + 3 < (size_t)e.B * e.L - 1) ? q0 + 3 : (size_t)e.B * e.L - 1;
        const size_t b0 = q0 / e.L, b1 = q1 / e.L;
        if (e.run_flags[b0 * nqg + (q0 - b0 * e.L) / 128] == 0 && e.run_flags[b1 * nqg + (q1 - b1 * e.L) / 128] == 0) continue;
        if (!acquired) {                                                  // (block-uniform)
            if (threadIdx.x == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            __syncthreads();
            acquired = true;
        }
        edge_softmax_topk_unit(e, kslots, cv, ci, blk);
        __syncthreads();
    }
}

int launch_topk_redo(hipStream_t s, const SelectArgs& a, const EdgeArgs& e, int pass, unsigned* barrier, int32_t* policy) {
    if (a.run_flags == nullptr || a.run_count == nullptr || e.run_flags != a.run_flags || barrier == nullptr || (pass != 2 && pass != 3)) {
        set_error("launch_topk_redo: redo arguments missing");
        return DAGL_ERR_INVALID;
    }
    const int n_qgroups = (a.L + SEL_WAVES * QT - 1) / (SEL_WAVES * QT);
    const int n_tiles = (a.N + KT - 1) / KT;
    const int n_units = n_qgroups * a.splits;
    const int rows_q = feat_rows(a.L), rows_x = feat_rows(a.N);
    const int ks = topk_slots(a.k);
    if (e.splits * 2 * ks > TOPK_MAX_CAND) {
        set_error("edge softmax: %d candidates per query exceed %d", e.splits * 2 * ks, TOPK_MAX_CAND);
        return DAGL_ERR_INVALID;
    }
    const unsigned n_merge = (unsigned)(((size_t)e.B * e.L + 3) / 4);
    // The kernel meets at a grid-wide barrier: EVERY block must be resident.  128 blocks at most (K = 64 takes 407 registers: ONE
    // block per CU), the batch in grid.y -- and never more than the device (a partition in CPX / QPX mode, a CU-masked stream) can
    // hold of THIS instantiation: hipOccupancyMaxActiveBlocksPerMultiprocessor x the CU count, asked once per instantiation and
    // device.  Fewer resident slots than images: the two-launch form (no barrier).
    int dev = 0, cus = 0;
    DAGL_HIP_TRY(hipGetDevice(&dev));
    DAGL_HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    auto capacity = [&](const void* fn) -> int {
        int per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 256, 0) != hipSuccess) { (void)hipGetLastError(); return 0; }
        return per_cu * cus;
    };
    int cap = 0;
#define DAGL_REDO_FN(P_, K_) reinterpret_cast<const void*>(&topk_redo_kernel<P_, K_>)
#define DAGL_REDO_CAP(P_) switch (ks) { case 4: cap = capacity(DAGL_REDO_FN(P_, 4)); break; case 8: cap = capacity(DAGL_REDO_FN(P_, 8)); break; \
                                        case 16: cap = capacity(DAGL_REDO_FN(P_, 16)); break; case 32: cap = capacity(DAGL_REDO_FN(P_, 32)); break; \
                                        default: cap = capacity(DAGL_REDO_FN(P_, 64)); break; }
    {
        // (cached: the answer depends on the instantiation and the device only)
        static thread_local int cached[2][5][16];                              // [pass - 2][slot class][device] = capacity + 1
        const int kc = ks == 4 ? 0 : ks == 8 ? 1 : ks == 16 ? 2 : ks == 32 ? 3 : 4;
        int* slot = (dev >= 0 && dev < 16) ? &cached[pass - 2][kc][dev] : nullptr;
        if (slot && *slot > 0) cap = *slot - 1;
        else {
            if (pass == 2) { DAGL_REDO_CAP(2) } else { DAGL_REDO_CAP(3) }
            if (slot) *slot = cap + 1;
        }
    }
#undef DAGL_REDO_CAP
#undef DAGL_REDO_FN
    int limit = cap < 128 ? cap : 128;
    int gx = limit / (a.B > 0 ? a.B : 1);
    if (gx > n_units) gx = n_units;
    if (gx < 1) {                                                         // (a batch this large, or a device this small: the two-launch form)
        int rc = launch_score_select(s, a, pass);
        return rc ? rc : launch_edge_softmax(s, e);
    }
    const dim3 grid(gx, a.B), block(256);
#define DAGL_REDO(P_, K_) hipLaunchKernelGGL((topk_redo_kernel<P_, K_>), grid, block, 0, s, a, e, ks, n_qgroups, n_tiles, rows_q, rows_x, \
                                              n_units, n_merge, barrier, policy)
#define DAGL_REDO_K(P_) switch (ks) { case 4: DAGL_REDO(P_, 4); break; case 8: DAGL_REDO(P_, 8); break; case 16: DAGL_REDO(P_, 16); break; \
                                      case 32: DAGL_REDO(P_, 32); break; default: DAGL_REDO(P_, 64); break; }
    if (pass == 2) { DAGL_REDO_K(2) } else { DAGL_REDO_K(3) }
#undef DAGL_REDO_K
#undef DAGL_REDO
    DAGL_LAUNCH_CHECK("topk_redo_kernel");
    return DAGL_OK;
}


// ---- dense scores (tests only): S[b,l,n] = Wq[l,:] . X[n,:] -------------------------------------------
__global__ void scores_dense_kernel(int L, int N, int rows_q, int rows_x, const float* __restrict__ wq,
                                    const float* __restrict__ x, float* __restrict__ sc) {
    const int b = blockIdx.z;
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    const int l = blockIdx.y;
    if (n >= N) return;
    const float* qp = wq + ((size_t)b * rows_q + l) * DS;
    const float* xp = x + ((size_t)b * rows_x + n) * DS;
    float acc = 0.f;
    for (int d = 0; d < D; ++d) acc = fmaf(qp[d], xp[d], acc);
    sc[((size_t)b * L + l) * N + n] = acc;
}

int launch_scores_dense(hipStream_t s, int B, int L, int N, const float* wq, const float* x, float* sc) {
    dim3 grid((N + 255) / 256, L, B);
    hipLaunchKernelGGL(scores_dense_kernel, grid, dim3(256), 0, s, L, N, feat_rows(L), feat_rows(N), wq, x, sc);
    DAGL_LAUNCH_CHECK("scores_dense_kernel");
    return DAGL_OK;
}

// ---- degrees and CSR offsets ------------------------------------------------------------------------------
// row_degree_kernel: one wave per query: deg = sum of its (chunk, half) segment counts, seg_rel = exclusive
// prefix of the counts inside the row, plus max / total over all rows (integer atomics: order-independent).
__global__ __launch_bounds__(256) void row_degree_kernel(int n_rows, int s2, const int32_t* __restrict__ seg_cnt,
                                                         int32_t* __restrict__ seg_rel, int32_t* __restrict__ deg,
                                                         unsigned long long* __restrict__ stats) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n_rows) return;
    int run = 0;
    for (int j0 = 0; j0 < s2; j0 += 64) {
        const int j = j0 + lane;
        const int c = (j < s2) ? seg_cnt[(size_t)row * s2 + j] : 0;
        int incl = c;                                         // inclusive wave scan
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(incl, o);
            if (lane >= o) incl += t;
        }
        if (j < s2) seg_rel[(size_t)row * s2 + j] = run + incl - c;
        run += __shfl(incl, 63);
    }
    if (lane == 0) {
        deg[row] = run;
        atomicAdd(&stats[0], (unsigned long long)run);
        atomicMax(&stats[1], (unsigned long long)run);
    }
}

// row_scan_kernel: exclusive scan of deg over the rows (single block; only the dense CSR path needs it)
__global__ __launch_bounds__(1024) void row_scan_kernel(int n_rows, const int32_t* __restrict__ deg,
                                                        int64_t* __restrict__ row_off) {
    __shared__ int64_t part[1024];
    const int t = threadIdx.x, nt = blockDim.x;
    const int per = (n_rows + nt - 1) / nt;
    const int r0 = min(n_rows, t * per), r1 = min(n_rows, r0 + per);
    int64_t sum = 0;
    for (int r = r0; r < r1; ++r) sum += deg[r];
    part[t] = sum;
    __syncthreads();
    if (t == 0) {
        int64_t run = 0;
        for (int j = 0; j < nt; ++j) { const int64_t v = part[j]; part[j] = run; run += v; }
        row_off[n_rows] = run;
    }
    __syncthreads();
    int64_t off = part[t];
    for (int r = r0; r < r1; ++r) { row_off[r] = off; off += deg[r]; }
}

int launch_row_degree(hipStream_t s, int n_rows, int s2, const int32_t* seg_cnt, int32_t* seg_rel, int32_t* deg,
                      int64_t* stats) {
    DAGL_HIP_TRY(hipMemsetAsync(stats, 0, 2 * sizeof(int64_t), s));
    hipLaunchKernelGGL(row_degree_kernel, dim3((n_rows + 3) / 4), dim3(256), 0, s, n_rows, s2, seg_cnt, seg_rel, deg,
                       reinterpret_cast<unsigned long long*>(stats));
    DAGL_LAUNCH_CHECK("row_degree_kernel");
    return DAGL_OK;
}

int launch_row_scan(hipStream_t s, int n_rows, const int32_t* deg, int64_t* row_off) {
    hipLaunchKernelGGL(row_scan_kernel, dim3(1), dim3(1024), 0, s, n_rows, deg, row_off);
    DAGL_LAUNCH_CHECK("row_scan_kernel");
    return DAGL_OK;
}

}  // namespace dagl
