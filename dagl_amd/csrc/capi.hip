// extern "C" boundary (include/dagl_ce.h) and the orchestration of one block forward.
#include <stdarg.h>
#include <string.h>

#include <atomic>

#include "dagl_common.h"
#include "thr_bias4.h"

namespace dagl {

static thread_local char g_err[512] = "";
// tags the calls of this process for the range guard (RangeTag): the only process-wide state of the library
static std::atomic<uint32_t> g_call_tag{1};
static int32_t next_call_tag() {
    uint32_t t = g_call_tag.fetch_add(1, std::memory_order_relaxed) & 0x7fffffffu;
    return (int32_t)(t ? t : 1u);
}

// A call captured into a HIP graph carries its tag as a kernel argument: every replay has the SAME tag, and a sticky word that holds
// it (one replay left the fp16 range / was not served in-stream) would poison every later replay.  The first launch of a captured call
// therefore turns "this tag" into "some earlier call" -- still non-zero: the poll still reports it -- before the call's own kernels run.
constexpr int32_t STALE_TAG = 0x7ffffffe;
__global__ void retag_kernel(int32_t* word, int32_t* veto, int32_t tag) {
    if (word != nullptr && *word == tag) *word = STALE_TAG;
    if (veto != nullptr && *veto == tag) *veto = STALE_TAG;
}

// Top-k threshold policy words of a workspace (stats[9] policy, stats[10] gate, stats[12] owner cookie).  A call that is not
// "prepared" (first call, another shape, a train / eval alternation) used to clear the policy: a module that alternates between
// two shapes forgot "tight" on every call and paid sampled pass + policy kernel + tight re-run each time.  The learnt word now
// survives as long as the workspace still carries the cookie of this very geometry (a fresh or re-used buffer does not).
__global__ void policy_init_kernel(int64_t* stats, int64_t cookie, int32_t start_tight) {
    if (stats[12] != cookie) { stats[9] = start_tight; stats[12] = cookie; }
    // [13]: sticky "a DAGL_FLAG_NO_REDO call went unserved" -- only prepared (weights-packed) calls set or report it, and this kernel runs
    // on the non-prepared ones: cleared every time, so that a workspace handed on by torch's caching allocator, or shared by modules of
    // one shape (the cookie hashes the geometry, not the owner), cannot pass a stale bit to CE.range_ok()
    stats[13] = 0;
    stats[10] = 0;
}

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int hip_fail(hipError_t e, const char* what) {
    set_error("HIP error %d (%s) at %s", (int)e, hipGetErrorString(e), what);
    (void)hipGetLastError();
    return DAGL_ERR_HIP;
}

// 128 bytes of pinned host memory per calling thread for the small device->host read-backs (a pageable
// destination would make every hipMemcpyAsync a blocking staged copy).  Allocated once, never freed.
static int64_t* pinned_scratch() {
    static thread_local int64_t* p = nullptr;
    if (p == nullptr) {
        void* q = nullptr;
        if (hipHostMalloc(&q, 128, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        p = static_cast<int64_t*>(q);
    }
    return p;
}
static int read_back(hipStream_t s, const int64_t* dev, int n, int64_t* out) {
    int64_t* pin = pinned_scratch();
    int64_t* dst = pin ? pin : out;
    DAGL_HIP_TRY(hipMemcpyAsync(dst, dev, (size_t)n * sizeof(int64_t), hipMemcpyDeviceToHost, s));
    DAGL_HIP_TRY(hipStreamSynchronize(s));
    if (pin) for (int i = 0; i < n; ++i) out[i] = pin[i];
    return DAGL_OK;
}

// Early verdict: the copy is queued behind the kernels that produce the words and followed by an event; the caller queues
// the rest of its launches and then waits for the EVENT only, so the device keeps working through the host round trip
// (a stream synchronise would drain it, and the next call's first kernels would start on an idle device).
static hipEvent_t verdict_event() {
    static thread_local hipEvent_t ev[16] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) { (void)hipGetLastError(); return nullptr; }
    if (ev[dev] == nullptr && hipEventCreateWithFlags(&ev[dev], hipEventDisableTiming) != hipSuccess) {
        (void)hipGetLastError(); ev[dev] = nullptr;
    }
    return ev[dev];
}
static int read_back_begin(hipStream_t s, const int64_t* dev, int n, bool* pending) {
    int64_t* pin = pinned_scratch();
    hipEvent_t ev = pin ? verdict_event() : nullptr;
    *pending = false;
    if (ev == nullptr) return DAGL_OK;                      // no pinned memory / event: read_back_end synchronises the stream
    DAGL_HIP_TRY(hipMemcpyAsync(pin, dev, (size_t)n * sizeof(int64_t), hipMemcpyDeviceToHost, s));
    DAGL_HIP_TRY(hipEventRecord(ev, s));
    *pending = true;
    return DAGL_OK;
}
static int read_back_end(hipStream_t s, const int64_t* dev, int n, bool pending, int64_t* out) {
    if (!pending) return read_back(s, dev, n, out);
    DAGL_HIP_TRY(hipEventSynchronize(verdict_event()));
    const int64_t* pin = pinned_scratch();
    for (int i = 0; i < n; ++i) out[i] = pin[i];
    return DAGL_OK;
}

// ---- optional stage profile: hipEvents recorded at stage boundaries on the caller's stream ------------------
struct Profile {
    int max_calls = 0, n_calls = 0;
    int only_stage = -1;                 // >= 0: record just the two events around that stage (an event record costs
                                         // ~4 us of stream time; nine per call are 12 % of a 256^2 forward)
    hipEvent_t* ev = nullptr;            // [max_calls][DAGL_N_STAGES + 1]
};
static inline void prof_mark(Profile* p, hipStream_t s, int stage_boundary) {
    if (p && p->n_calls < p->max_calls &&
        (p->only_stage < 0 || stage_boundary == p->only_stage || stage_boundary == p->only_stage + 1))
        (void)hipEventRecord(p->ev[(size_t)p->n_calls * (DAGL_N_STAGES + 1) + stage_boundary], s);
}

// ---- execution plan: chunking of the key stream + workspace carve ---------------------------------------
struct Plan {
    Grid g;
    int B, mode, k, kslots;
    bool screen;                    // bf16 screen + exact refine (default) vs. the all-fp32 scan
    bool split16;                   // fp16 split-operand projection (default) vs. the fp32 MFMA projection
    int splits, tiles_per_split, n_tiles;                 // fp32 scan (select.hip)
    int s_splits, s_steps_per_split, s_steps, s_sample, s_qblock;   // bf16 screen (screen.hip)
    int s_gkeep = 4;                                      // group maxima per (query, chunk, half) segment handed to the threshold kernel (ScreenArgs::gkeep)
    int capseg, capseg_alloc;
    int s_sample_tight = 0, capseg_tight = 0;             // top-k modes behind the screen: the tight threshold's pair (DAGL_FLAG_TIGHT_TOPK)
    int width;                      // neighbour-list width of the fixed-width paths
    int ovf_cap;                    // adaptive lists behind the screen: queries that may be redone one by one (overflow.hip)
    bool wide = false;              // top-k modes, k > DAGL_MAX_TOPK: row-wise dense form (topk_wide.hip), no lists
    size_t o_wide = 0;
    // byte offsets into the workspace
    size_t o_b1p, o_b2p, o_wp1, o_wp2, o_x, o_wq, o_xh, o_wqh, o_colsum, o_mt, o_cnt, o_segcnt, o_segoff, o_rowoff,
        o_deg, o_stats, o_lidx, o_lval, o_cidx, o_cval, o_nbidx, o_nbwgt, o_nbcnt, o_agg, o_gmax, o_theta, o_smax, o_traw, o_scand, o_spill, o_spillcnt,
        o_scandv, o_ssegcnt, o_redo, o_ovflist, o_heavy, o_ovfq, o_ovfscores, o_ovfpart, o_thr, o_bias, o_thrpart, o_maphi, o_maplo, o_maphi2, o_maplo2, o_b1amax, o_wp1h, o_wp2h, o_convw, o_colpart, o_end;
};

static bool g_N_small(int H, int W);
static size_t carve(size_t& off, size_t bytes) {
    const size_t o = off;
    off = align_up(off + bytes, 256);
    return o;
}

constexpr int SCREEN_MIN_KEYS = 2048;     // below this the fp32 scan is launch-bound anyway
constexpr int SCREEN_CAPSEG = 16;
static bool g_N_small(int H, int W) { return (int64_t)H * W < SCREEN_MIN_KEYS; }

static int make_plan(int B, int H, int W, int mode_flags, int k, Plan& p, bool core = false) {
    const int mode = mode_flags & 0xff;
    const bool exact = (mode_flags & DAGL_FLAG_EXACT_SCAN) != 0;
    DAGL_REQUIRE(B >= 1 && H >= 1 && W >= 1, "dagl: bad shape B=%d H=%d W=%d", B, H, W);
    DAGL_REQUIRE((mode_flags & ~(0xff | DAGL_FLAG_EXACT_SCAN | DAGL_FLAG_WEIGHTS_PACKED | DAGL_FLAG_DENSE_HINT | DAGL_FLAG_NO_WAIT | DAGL_FLAG_TIGHT_TOPK | DAGL_FLAG_SAMPLED_TOPK | DAGL_FLAG_NO_REDO)) == 0 &&
                 (mode == DAGL_MODE_ADAPTIVE || mode == DAGL_MODE_TOPK || mode == DAGL_MODE_ADAPTIVE_TOPK),
                 "dagl: unknown mode 0x%x", mode_flags);
    if (mode != DAGL_MODE_ADAPTIVE) DAGL_REQUIRE(k >= 1, "dagl: k=%d < 1", k);
    DAGL_REQUIRE((int64_t)H * W < (1ll << 30), "dagl: image too large");
    if (mode != DAGL_MODE_ADAPTIVE && (int64_t)k > (int64_t)H * W) k = H * W;     // top_k = min(num_edge, N), GReccR2b_3mh_1-checkpoint.py:243
    // neighbourhoods wider than the lists: every query in the row-wise dense form (inference entry points only)
    p.wide = mode != DAGL_MODE_ADAPTIVE && k > DAGL_MAX_TOPK;
    if (p.wide) DAGL_REQUIRE(!core, "dagl_ce_core_forward: k=%d > %d: the differentiable path keeps lists of at most %d neighbours", k,
                             DAGL_MAX_TOPK, DAGL_MAX_TOPK);
    p.g = make_grid(H, W);
    p.B = B; p.mode = mode; p.k = k;
    p.kslots = (mode == DAGL_MODE_ADAPTIVE || p.wide) ? 0 : topk_slots(k);
    const Grid& g = p.g;
    p.screen = !exact && !p.wide && g.N >= SCREEN_MIN_KEYS;
    p.split16 = !exact;
    p.n_tiles = (g.N + KT - 1) / KT;
    const int n_qgroups = (g.L + 127) / 128;
    // fp32 scan: enough blocks for ~4 per CU, chunks of at least 8 tiles, candidate merge bounded for top-k
    int splits = (1024 + n_qgroups * B - 1) / (n_qgroups * B);
    const int max_by_tiles = (p.n_tiles + 7) / 8;
    if (splits > max_by_tiles) splits = max_by_tiles;
    if (p.kslots) { const int cap = 1024 / (2 * p.kslots); if (splits > cap) splits = cap; }
    if (splits < 1) splits = 1;
    p.tiles_per_split = (p.n_tiles + splits - 1) / splits;
    p.splits = (p.n_tiles + p.tiles_per_split - 1) / p.tiles_per_split;
    // adaptive lists: 256 slots behind the screen (mean degrees of ~8 come with maxima of ~100), 64 for the exact scan and for
    // the training entry point (its backward keeps one neighbour per lane)
    p.width = (mode == DAGL_MODE_ADAPTIVE) ? ((core || exact || g_N_small(H, W)) ? DAGL_FAST_CAP : DAGL_LIST_CAP) : (p.wide ? 1 : k);
    // bf16 screen: the key stream from L2 into LDS is what bounds it (LDS-DMA lands ~25 GB/s per CU, 6.4 TB/s over the chip;
    // at 256^2 sixteen groups of 256 queries stream the 28 MB of bf16 keys 16 times = 453 MB = 71 us against 46 us of matrix
    // work), so a block covers 512 queries (16 waves, one block per CU: every key tile is fetched half as often) whenever
    // that still gives one block per CU; otherwise 256 queries (8 waves, two blocks per CU).  Key chunks of >= 4 steps of 64
    // keys, <= 64 chunks.
    p.s_steps = (g.N + SKEYS - 1) / SKEYS;
    {
        const int mx = (p.s_steps + 3) / 4 > 64 ? 64 : (p.s_steps + 3) / 4;
        // 512-, 384- or 256-query blocks: whichever leaves the fewest idle query slots (a 72 x 72 tile has L = 324: 12 waves
        // instead of 16), the larger block on a tie; the 16- and 12-wave blocks only when they still give one block per CU
        int qb = 256;
        long long slots = (long long)((g.L + 255) / 256) * 256;
        for (int cand = 384; cand <= 512; cand += 128) {
            const long long nq = (g.L + cand - 1) / cand;
            if (nq * B * mx < 256) continue;
            if (nq * cand <= slots) { qb = cand; slots = nq * cand; }
        }
#ifdef DAGL_ABLATION
        { static const int e = [] { const char* v = getenv("DAGL_SCREEN_QBLOCK"); return v ? atoi(v) : 0; }(); if (e == 256 || e == 384 || e == 512) qb = e; }
        static const int target_env = [] { const char* e = getenv("DAGL_SCREEN_BLOCKS"); return e ? atoi(e) : 0; }();
        const int target = target_env > 0 ? target_env : (qb >= 384 ? 256 : 512);
#else
        const int target = (qb >= 384) ? 256 : 512;   // one resident round
#endif
        const int nqg = (g.L + qb - 1) / qb;
        int sp = (target + nqg * B - 1) / (nqg * B);
        // top-k modes: the threshold is the k-th largest of chunks x 2 x 16 group maxima per query -- large batches of small maps
        // (256 leaf tiles: ONE chunk) left fewer values than k = 50 (theta = 0: every key a candidate, every group on the redo pass)
        // -- and the fewer keys a group holds, the closer its maximum lies to the query's best keys: k / 4 chunks (k = 50: 416 groups
        // of a dozen keys on a 72 x 72 tile instead of 128 groups of forty: a third of the candidates)
        if (mode != DAGL_MODE_ADAPTIVE && sp < (k + 3) / 4) sp = (k + 3) / 4;
        if (sp > mx) sp = mx;
        if (sp < 1) sp = 1;
        p.s_qblock = qb;
        p.s_steps_per_split = (p.s_steps + sp - 1) / sp;
        p.s_splits = (p.s_steps + p.s_steps_per_split - 1) / p.s_steps_per_split;
        // top-k threshold from every 8th key tile: ~2 k s candidates per query reach refine, which thins them out with their
        // screened scores before any feature row is fetched (every 4th: 8 us more sampling for 3 us less filtering at 256^2)
        // (the candidate count grows like k x stride: the stride is kept at <= 64 / k so that a query's candidates stay well
        // inside its slots -- 1024^2, k = 16, every 8th tile sent most queries to the exact redo: 165 ms instead of 27)
        p.s_sample = p.s_steps_per_split >= 16 ? 8 : (p.s_steps_per_split >= 8 ? 4 : (p.s_steps_per_split >= 4 ? 2 : 1));
        { const int kk = (k > 8) ? k : 8; int cap = 64 / kk; if (cap < 1) cap = 1; if (p.s_sample > cap) p.s_sample = cap; }
#ifdef DAGL_ABLATION
        { static const int smp = [] { const char* e = getenv("DAGL_SCREEN_SAMPLE"); return e ? atoi(e) : 0; }(); if (smp > 0) p.s_sample = smp; }
#endif
    }
    // candidate slots per (query, chunk, half) segment: 16 when a query has many segments, up to 256 when it has few
    // (short key streams): ~1024 slots per query in total
    p.capseg = SCREEN_CAPSEG;
    while (p.capseg < 256 && 2 * p.capseg * p.s_splits * 2 <= 1024) p.capseg *= 2;
#ifdef DAGL_ABLATION
    { static const int cs = [] { const char* e = getenv("DAGL_SCREEN_CAPSEG"); return e ? atoi(e) : 0; }(); if (cs >= 4) p.capseg = cs; }
#endif
    // DAGL_FLAG_TIGHT_TOPK: the threshold from every second key tile and eight times the slots per segment (as far as 2 GiB of
    // records go; with fewer slots: from every tile) -- on the Set12 feature maps every 2nd tile + 128 slots serves six of
    // seven images at 0.25 ms, every tile + 64 slots the same six at 0.27, every 4th tile two (profiles/r03_real_features_topk.log).
    // The records are ALWAYS laid out for the larger count, so that a workspace serves both kinds of call with one layout.
    p.capseg_alloc = p.capseg;
    if (p.screen && mode != DAGL_MODE_ADAPTIVE) {
        while (p.capseg_alloc < 8 * p.capseg && p.capseg_alloc < 256 &&
               (size_t)B * g.L * p.s_splits * 2 * (2 * (size_t)p.capseg_alloc) * sizeof(int2) <= ((size_t)2 << 30))
            p.capseg_alloc *= 2;
        // the tight pair: forced by the flag, or taken by the kernels themselves once the workspace's policy word says so
        p.s_sample_tight = (p.capseg_alloc >= 8 * p.capseg && p.s_sample >= 2) ? 2 : 1;
        p.capseg_tight = p.capseg_alloc;
        if (mode_flags & DAGL_FLAG_TIGHT_TOPK) { p.s_sample = p.s_sample_tight; p.capseg = p.capseg_tight; }
    }

    const size_t BL = (size_t)B * g.L;
    size_t off = 0;
    const size_t map_b = (size_t)B * g.Hp * g.Wp * CH * sizeof(float);
    p.o_b1p = carve(off, map_b);
    p.o_b2p = carve(off, map_b);
    p.o_wp1 = carve(off, (size_t)DPAD * P * sizeof(float));
    p.o_wp2 = carve(off, (size_t)DPAD * P * sizeof(float));
    p.o_x = carve(off, (size_t)B * feat_rows(g.N) * DS * sizeof(float));
    p.o_wq = carve(off, (size_t)B * feat_rows(g.L) * DS * sizeof(float));
    p.o_colsum = carve(off, (size_t)B * DS * sizeof(double));
    p.o_mt = carve(off, BL * sizeof(float));
    p.o_cnt = carve(off, BL * sizeof(int32_t));
    p.o_segcnt = carve(off, BL * p.splits * 2 * sizeof(int32_t));
    p.o_segoff = carve(off, BL * p.splits * 2 * sizeof(int32_t));
    p.o_rowoff = carve(off, (BL + 1) * sizeof(int64_t));
    p.o_deg = carve(off, BL * sizeof(int32_t));
    p.o_stats = carve(off, 16 * sizeof(int64_t));         // [0..3] per-call counters, [4] range word, [5] tag of the last completed call,
                                                          // [6] redone rows, [7] their edges, [8] veto word (DAGL_FLAG_NO_WAIT),
                                                          // [9] top-k threshold policy (sticky), [10] gate of the in-call re-run
    if (mode == DAGL_MODE_ADAPTIVE) {
        p.o_lidx = carve(off, BL * DAGL_FAST_CAP * sizeof(int32_t));
        p.o_lval = carve(off, BL * DAGL_FAST_CAP * sizeof(float));
        p.o_cidx = p.o_cval = 0;
    } else {
        p.o_lidx = p.o_lval = 0;
        p.o_cidx = carve(off, BL * p.splits * 2 * p.kslots * sizeof(int32_t));
        p.o_cval = carve(off, BL * p.splits * 2 * p.kslots * sizeof(float));
    }
    p.o_nbidx = carve(off, BL * p.width * sizeof(int32_t));
    p.o_nbwgt = carve(off, BL * p.width * sizeof(float));
    p.o_nbcnt = carve(off, BL * sizeof(int32_t));
    p.o_agg = carve(off, BL * P * sizeof(float));
    p.o_thr = carve(off, BL * sizeof(float));
    p.o_bias = carve(off, BL * sizeof(float));
    p.o_thrpart = carve(off, 8 * BL * sizeof(float));                       // prologue: partial thr/bias sums of 4 channel groups
    p.o_maphi = p.o_maplo = p.o_maphi2 = p.o_maplo2 = p.o_b1amax = p.o_wp1h = p.o_wp2h = p.o_convw = p.o_colpart = 0;
    if (!exact) {
        p.o_maphi = carve(off, (size_t)B * g.Hp * g.Wp * CH * sizeof(uint16_t));
        p.o_maplo = carve(off, (size_t)B * g.Hp * g.Wp * CH * sizeof(uint16_t));
        // the coarse tier of the key / query map and the conv blocks' |b1| slots (B1Tiers, dagl_common.h): written by conv_pair16_kernel,
        // read by project16_kernel, so that activations beyond the fine tier's |b1| < 3750 are served by the same launches
        p.o_maphi2 = carve(off, (size_t)B * g.Hp * g.Wp * CH * sizeof(uint16_t));
        p.o_maplo2 = carve(off, (size_t)B * g.Hp * g.Wp * CH * sizeof(uint16_t));
        p.o_b1amax = carve(off, (size_t)conv16_blocks_per_head(g, 1, B) * sizeof(float));
        p.o_wp1h = carve(off, 4 * P16_PACKED_HALFS * sizeof(uint16_t));          // up to 4 heads (stage entry point)
        p.o_wp2h = carve(off, 4 * P16_PACKED_HALFS * sizeof(uint16_t));
        p.o_convw = carve(off, 4 * CONV_W16_BYTES);                                       // packed g / theta weights per head
        p.o_colpart = carve(off, (size_t)B * project16_key_blocks(g) * 224 * sizeof(float));
    }
    p.o_xh = p.o_wqh = p.o_gmax = p.o_theta = p.o_smax = p.o_traw = p.o_spill = p.o_spillcnt = p.o_scand = p.o_scandv = p.o_ssegcnt = p.o_redo = 0;
    if (p.screen) {
        p.o_xh = carve(off, (size_t)B * feat_rows_h(g.N) * DSH * sizeof(uint16_t));
        p.o_wqh = carve(off, (size_t)B * feat_rows_h(g.L) * DSH * sizeof(uint16_t));
        p.s_gkeep = (p.s_splits * 2 * 16 <= 512) ? 16 : 4;       // (screen_theta_kernel takes up to 512 values per query)
        p.o_gmax = carve(off, BL * p.s_splits * 2 * p.s_gkeep * sizeof(float));
        p.o_theta = carve(off, BL * sizeof(float));
        p.o_scand = carve(off, BL * p.s_splits * 2 * p.capseg_alloc * sizeof(int2));     // candidate records (count in slot 0)
        p.o_spill = p.o_spillcnt = 0;
        if (mode != DAGL_MODE_ADAPTIVE) {       // top-k modes: a query's shared area behind its segments (ScreenArgs::spill)
            p.o_spill = carve(off, BL * SCREEN_SPILL * sizeof(int2));
            p.o_spillcnt = carve(off, BL * sizeof(unsigned));
        }
        p.o_smax = (mode == DAGL_MODE_ADAPTIVE) ? carve(off, BL * sizeof(float)) : 0;      // dense formulation: the rows' shifts (as scores)
        p.o_traw = (mode == DAGL_MODE_ADAPTIVE) ? carve(off, BL * sizeof(float)) : 0;      // ... and their largest sampled screened scores
        p.o_redo = carve(off, (size_t)B * n_qgroups * sizeof(int32_t));
    }
    p.ovf_cap = 0; p.o_ovflist = p.o_heavy = p.o_ovfq = p.o_ovfscores = p.o_ovfpart = 0;
    if (p.screen && mode == DAGL_MODE_ADAPTIVE && !core) {
        p.ovf_cap = overflow_cap(g.N, B);
        p.o_ovflist = carve(off, (size_t)p.ovf_cap * sizeof(int32_t));
        p.o_heavy = carve(off, (size_t)refine_heavy_cap() * sizeof(int32_t));
        p.o_ovfq = carve(off, (size_t)p.ovf_cap * DS * sizeof(float));
        p.o_ovfscores = carve(off, (size_t)B * p.ovf_cap * ((g.N + 31) / 32 * 32) * sizeof(float));
        p.o_ovfpart = carve(off, (size_t)p.ovf_cap * OVF_CHUNKS * OVF_PART_FLOATS * sizeof(float));
    }
    if (p.wide) p.o_wide = carve(off, topk_wide_workspace_bytes(g.N, g.L));
    p.o_end = off;
    return DAGL_OK;
}

template <class T>
static T* at(void* ws, size_t off) { return reinterpret_cast<T*>(static_cast<char*>(ws) + off); }

static int check_device() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { set_error("dagl: no HIP device"); (void)hipGetLastError(); return DAGL_ERR_NO_DEVICE; }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) { set_error("dagl: cannot query device"); (void)hipGetLastError(); return DAGL_ERR_NO_DEVICE; }
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        set_error("dagl: device arch %s is not gfx950", prop.gcnArchName);
        return DAGL_ERR_NO_DEVICE;
    }
    return DAGL_OK;
}

struct FusedIn {                 // input of the fused-prologue entry points (all device pointers), one per head
    const float* x;              // [B,64,H,W] (shared by the heads of a stage)
    const float *g_w, *g_b, *th_w, *th_b, *thr_w, *thr_b, *bias_w, *bias_b;
    const float *fc1_w, *fc1_b, *fc2_w, *fc2_b;
};

struct CoreIn {                  // training entry point: features given, neighbour lists handed back for the backward
    const float* wq_rows; const float* x_rows;                   // [B,L,196], [B,N,196] dense rows (post-ReLU)
    int32_t* nb_idx; float* nb_wgt; float* nb_s; int32_t* nb_cnt; // [B,L,width] x3, [B,L]
    float* mu;                                                   // [B,L] row means of S (adaptive modes)
    float* lse;                                                  // dense core (streamed dense formulation): [B,L,2] = {softmax shift M, sum Z};
                                                                 // the list outputs are then unused (null)
};

static int ce_forward_impl(hipStream_t s, int B, int H, int W, const float* b1, const float* b2, const float* thr,
                           const float* bias, const float* fc1_w, const float* fc1_b, const float* fc2_w,
                           const float* fc2_b, int mode_flags, int k, float* out, void* ws, size_t ws_bytes,
                           dagl_ce_info* info, int32_t* dbg_deg, float* dbg_rowsum, float* dbg_agg,
                           Profile* prof = nullptr, const FusedIn* fin = nullptr, int heads = 1,
                           const CoreIn* core = nullptr) {
    // heads > 1 (stage entry point): `B` counts head x image pairs, batch entry = head * (B / heads) + image; fin[h]
    // carries head h's weights, `out` is the [B/heads, heads*16, H, W] concat map
    Plan p;
    int rc = make_plan(B, H, W, mode_flags, k, p, core != nullptr);
    if (rc) return rc;
    const int mode = p.mode;
    k = p.k;                                                     // (clamped to the number of keys)
    if (info) { info->required_bytes = (int64_t)p.o_end; info->total_edges = -1; info->max_degree = -1; info->path = 0;
                info->redone_queries = -1; info->range_fallback = 0; info->dense_rerun_blocks = 0; }
    DAGL_REQUIRE(out && (core || (fc1_w && fc1_b && fc2_w && fc2_b)), "dagl_ce_forward: null tensor pointer");
    if (core) {
        DAGL_REQUIRE(core->wq_rows && core->x_rows && b2 && (core->lse || (core->nb_idx && core->nb_wgt && core->nb_s && core->nb_cnt)),
                     "dagl_ce_core_forward: null tensor pointer");
        if (mode != DAGL_MODE_TOPK) DAGL_REQUIRE(thr && bias && core->mu, "dagl_ce_core_forward: thr/bias/mu required in adaptive modes");
    } else if (fin) {
        DAGL_REQUIRE(fin->x && fin->g_w && fin->g_b && fin->th_w && fin->th_b, "dagl_ce_forward_fused: null tensor pointer");
        if (mode != DAGL_MODE_TOPK)
            DAGL_REQUIRE(fin->thr_w && fin->thr_b && fin->bias_w && fin->bias_b, "dagl_ce_forward_fused: thr/bias heads required in adaptive modes");
    } else {
        DAGL_REQUIRE(b1 && b2, "dagl_ce_forward: null tensor pointer");
        if (mode != DAGL_MODE_TOPK) DAGL_REQUIRE(thr && bias, "dagl_ce_forward: thr/bias required in adaptive modes");
    }
    DAGL_REQUIRE(ws != nullptr && ((uintptr_t)ws % 256) == 0, "dagl_ce_forward: workspace must be 256-byte aligned");
    if (ws_bytes < p.o_end) {
        set_error("dagl_ce_forward: workspace %zu B < required %zu B", ws_bytes, p.o_end);
        return DAGL_ERR_WORKSPACE;
    }
    const Grid& g = p.g;
    const size_t BL = (size_t)B * g.L;
    const int n_qgroups = (g.L + 127) / 128;

    float* b1p = at<float>(ws, p.o_b1p);
    float* b2p = at<float>(ws, p.o_b2p);
    float* wp1 = at<float>(ws, p.o_wp1);
    float* wp2 = at<float>(ws, p.o_wp2);
    float* X = at<float>(ws, p.o_x);
    float* Wq = at<float>(ws, p.o_wq);
    uint16_t* Xh = p.screen ? at<uint16_t>(ws, p.o_xh) : nullptr;
    uint16_t* Wqh = p.screen ? at<uint16_t>(ws, p.o_wqh) : nullptr;
    double* colsum = at<double>(ws, p.o_colsum);
    float* mt = at<float>(ws, p.o_mt);
    int32_t* cnt = at<int32_t>(ws, p.o_cnt);
    int32_t* segcnt = at<int32_t>(ws, p.o_segcnt);
    int32_t* segrel = at<int32_t>(ws, p.o_segoff);
    int64_t* rowoff = at<int64_t>(ws, p.o_rowoff);
    int32_t* deg = at<int32_t>(ws, p.o_deg);
    int64_t* stats = at<int64_t>(ws, p.o_stats);
    int32_t* nbidx = core ? core->nb_idx : at<int32_t>(ws, p.o_nbidx);
    float* nbwgt = core ? core->nb_wgt : at<float>(ws, p.o_nbwgt);
    int32_t* nbcnt = core ? core->nb_cnt : at<int32_t>(ws, p.o_nbcnt);
    float* agg = at<float>(ws, p.o_agg);

    // range guard of the split-fp16 kernels (dagl_common.h RangeTag); the fp32 path and the training entry point have no
    // such range
    RangeTag rt;
    // (round 6: the fp32 path too -- it has no range, but a NaN feature must not become a 0 behind its ReLU, nor an inf-poisoned key drop out
    // of the selection: dagl.py:207-275 returns NaN for a non-finite input, and so does every path here; project.hip sets the word)
    {                               // (training entry points: the streamed dense core splits the features into fp16 halves too, and
                                    //  non-finite feature rows -- a poisoned projection upstream -- must not vanish in the selection)
        rt.word = reinterpret_cast<int32_t*>(stats + 4); rt.done = reinterpret_cast<int32_t*>(stats + 5); rt.tag = next_call_tag();
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s, &cap) != hipSuccess) { (void)hipGetLastError(); cap = hipStreamCaptureStatusNone; }
        if (cap == hipStreamCaptureStatusActive) {
            hipLaunchKernelGGL(retag_kernel, dim3(1), dim3(1), 0, s, rt.word,
                               mode == DAGL_MODE_ADAPTIVE ? reinterpret_cast<int32_t*>(stats + 8) : nullptr, rt.tag);
            DAGL_LAUNCH_CHECK("retag_kernel");
        }
    }

    // ---- stage 0: layout: zero-bordered NHWC maps, packed fc weights ------------------------------------
    prof_mark(prof, s, 0);
    ThrHeadSet thr_all = {};
    bool thr_in_proj = false;
    const int imgs = B / heads;
    // "prepared": the caller vouches that this workspace last served an identical call (same geometry, mode, weights):
    // packed weights, the maps' zero borders and the zero guard rows of the feature matrices are still in place, and
    // the few per-call counters are cleared by the prologue kernel itself -- three launches fewer
    const bool prepared = fin && p.split16 && (mode_flags & DAGL_FLAG_WEIGHTS_PACKED);
    if (rt.word != nullptr && !prepared) DAGL_HIP_TRY(hipMemsetAsync(stats + 4, 0, 2 * sizeof(int64_t), s));   // fresh workspace
    if (!prepared && mode == DAGL_MODE_ADAPTIVE) DAGL_HIP_TRY(hipMemsetAsync(stats + 8, 0, sizeof(int64_t), s));
    // top-k modes behind the screen: the threshold policy lives in the workspace (include/dagl_ce.h DAGL_FLAG_TIGHT_TOPK); a cold
    // workspace starts with the sampled threshold and a closed gate
    const bool topk_policy = p.screen && mode != DAGL_MODE_ADAPTIVE && !(mode_flags & (DAGL_FLAG_TIGHT_TOPK | DAGL_FLAG_SAMPLED_TOPK)) &&
                             p.capseg_tight > 0 && (p.capseg_tight != p.capseg || p.s_sample_tight != p.s_sample);
    int32_t* policy_w = reinterpret_cast<int32_t*>(stats + 9);
    int32_t* gate_w = reinterpret_cast<int32_t*>(stats + 10);
    if (p.screen && mode != DAGL_MODE_ADAPTIVE && !prepared) {
        // (forced thresholds -- DAGL_FLAG_TIGHT_TOPK / _SAMPLED_TOPK -- do not read the policy word; the cookie and the sticky words are kept all the same)
        // maps of up to 16 384 keys (128^2; the 72 x 72 leaf tiles of the tiled driver) START on the tight threshold: there it costs
        // nothing measurable on synthetic maps (sampling every second key tile of <= 256 is a few steps) and its better threshold
        // saves 12 % on natural-image leaf tiles even when nothing overflows (0.64 against 0.73 ms per batch of 64 tiles,
        // profiles/r04_topk_policy_real_features.log).  The word is kept while the workspace carries this geometry's cookie.
        uint64_t ck = 0x5DA6ull;
        for (const int v : {B, H, W, mode, k, p.s_splits, p.capseg, p.capseg_tight}) ck = ck * 0x100000001B3ull ^ (uint64_t)(uint32_t)v;
        hipLaunchKernelGGL(policy_init_kernel, dim3(1), dim3(1), 0, s, stats, (int64_t)(ck | 1ull), g.N <= 16384 ? 1 : 0);
        DAGL_LAUNCH_CHECK("policy_init_kernel");
    }
    if (core) {
        if ((rc = launch_pad_nhwc(s, B, H, W, b2, b2p))) return rc;
    } else if (fin) {
        float* thr_ws = at<float>(ws, p.o_thr);
        float* bias_ws = at<float>(ws, p.o_bias);
        const bool thr_heads = (mode != DAGL_MODE_TOPK);
        const size_t map_f = (size_t)imgs * g.Hp * g.Wp * CH;
        const bool conv_merged = heads > 1 && p.split16;       // a stage's heads: their g / theta convolutions are ONE launch
        // the thr / bias heads' partial sums are first read by query_thresholds_kernel, behind the projection: on the split-fp16 path their
        // blocks ride in the projection's launch (round 5; 13 us of every adaptive-mode call as a launch of their own)
        if (thr_heads) {
            for (int h2 = 0; h2 < heads; ++h2) { thr_all.x[h2] = fin[h2].x; thr_all.thr_w[h2] = fin[h2].thr_w; thr_all.bias_w[h2] = fin[h2].bias_w; }
            thr_all.imgs = imgs;
            thr_in_proj = p.split16 && thr_bias4_ok(g, thr_all, heads);
        }
        // the map's two tiers (B1Tiers): all heads of the call share the slot array, [head][blocks of that head]
        B1Tiers tiers_all;
        if (p.split16) {
            tiers_all.hi2 = at<uint16_t>(ws, p.o_maphi2); tiers_all.lo2 = at<uint16_t>(ws, p.o_maplo2);
            tiers_all.amax = at<float>(ws, p.o_b1amax); tiers_all.slots = conv16_blocks_per_head(g, heads, imgs);
        }
        for (int hd = 0; hd < heads; ++hd) {
            const FusedIn& f = fin[hd];
            B1Tiers tiers_hd;
            unsigned char* convw = p.split16 ? at<unsigned char>(ws, p.o_convw) + (size_t)hd * CONV_W16_BYTES : nullptr;
            if (convw && !(mode_flags & DAGL_FLAG_WEIGHTS_PACKED) && (rc = launch_pack_conv_weight16(s, f.g_w, f.th_w, convw))) return rc;
            if (conv_merged && hd == heads - 1) {
                ConvHeadSet hs = {};
                for (int h2 = 0; h2 < heads; ++h2) {
                    hs.x[h2] = fin[h2].x; hs.w[h2] = at<unsigned char>(ws, p.o_convw) + (size_t)h2 * CONV_W16_BYTES;
                    hs.gb[h2] = fin[h2].g_b; hs.tb[h2] = fin[h2].th_b;
                }
                hs.imgs = imgs;
                hs.tiers = tiers_all;
                if ((rc = launch_conv_pair16_heads(s, heads, imgs, g, hs, b2p, at<uint16_t>(ws, p.o_maphi), at<uint16_t>(ws, p.o_maplo),
                                                   prepared ? reinterpret_cast<uint32_t*>(stats) : nullptr, prepared ? 8 : 0,
                                                   (prepared && p.screen) ? reinterpret_cast<uint32_t*>(at<int32_t>(ws, p.o_redo)) : nullptr,
                                                   (prepared && p.screen) ? B * n_qgroups : 0, rt))) return rc;
                if (thr_heads && !thr_in_proj) {
                    ThrHeadSet th = {};
                    for (int h2 = 0; h2 < heads; ++h2) { th.x[h2] = fin[h2].x; th.thr_w[h2] = fin[h2].thr_w; th.bias_w[h2] = fin[h2].bias_w; }
                    th.imgs = imgs;
                    if ((rc = launch_thr_bias_heads(s, heads, imgs, g, th, at<float>(ws, p.o_thrpart)))) return rc;
                }
            }
            // default path: the key/query map is only ever consumed as split fp16 (project16), so the prologue
            // writes the hi / lo maps itself and no fp32 copy exists
            if ((rc = launch_prologue(s, imgs, g, f.x, f.g_w, f.g_b, f.th_w, f.th_b, f.thr_w, f.thr_b, f.bias_w, f.bias_b,
                                      p.split16 ? nullptr : b1p + hd * map_f, b2p + hd * map_f,
                                      (thr_heads && !thr_in_proj) ? thr_ws + (size_t)hd * imgs * g.L : nullptr,
                                      bias_ws + (size_t)hd * imgs * g.L,
                                      p.split16 ? at<uint16_t>(ws, p.o_maphi) + hd * map_f : nullptr,
                                      p.split16 ? at<uint16_t>(ws, p.o_maplo) + hd * map_f : nullptr,
                                      at<float>(ws, p.o_thrpart) + (size_t)hd * 8 * imgs * g.L,
                                      prepared, /*defer_thr_reduce=*/thr_heads,
                                      (prepared && hd == 0) ? reinterpret_cast<uint32_t*>(stats) : nullptr,
                                      (prepared && hd == 0) ? 8 : 0,
                                      (prepared && hd == 0 && p.screen) ? reinterpret_cast<uint32_t*>(at<int32_t>(ws, p.o_redo)) : nullptr,
                                      (prepared && hd == 0 && p.screen) ? B * n_qgroups : 0, rt, convw, conv_merged,
                                      p.split16 ? &(tiers_hd = B1Tiers{tiers_all.hi2 + hd * map_f, tiers_all.lo2 + hd * map_f, tiers_all.amax, tiers_all.slots}) : nullptr))) return rc;
        }
        thr = thr_ws; bias = bias_ws;
    } else {
        if ((rc = launch_pad_nhwc(s, B, H, W, b1, b1p))) return rc;
        if ((rc = launch_pad_nhwc(s, B, H, W, b2, b2p))) return rc;
    }
    uint16_t *map_hi = nullptr, *map_lo = nullptr, *wp1h = nullptr, *wp2h = nullptr;
    if (core) {
        // nothing to pack: the projections were done by the caller (under autograd)
    } else if (p.split16) {
        map_hi = at<uint16_t>(ws, p.o_maphi); map_lo = at<uint16_t>(ws, p.o_maplo);
        wp1h = at<uint16_t>(ws, p.o_wp1h); wp2h = at<uint16_t>(ws, p.o_wp2h);
        if (!fin)                                                // stock-conv entry point: split the padded fp32 map
            if ((rc = launch_split_map(s, (size_t)B * g.Hp * g.Wp * CH, b1p, map_hi, map_lo, rt))) return rc;
        for (int hd = 0; hd < heads && !(mode_flags & DAGL_FLAG_WEIGHTS_PACKED); ++hd) {
            if ((rc = launch_pack_fc_weight16(s, fin ? fin[hd].fc1_w : fc1_w, wp1h + (size_t)hd * P16_PACKED_HALFS))) return rc;
            if ((rc = launch_pack_fc_weight16(s, fin ? fin[hd].fc2_w : fc2_w, wp2h + (size_t)hd * P16_PACKED_HALFS))) return rc;
        }
    } else {
        DAGL_REQUIRE(heads == 1, "dagl: the stage entry point needs the default (screened) scan");
        if (!(mode_flags & DAGL_FLAG_WEIGHTS_PACKED)) {
            if ((rc = launch_pack_fc_weight(s, fc1_w, wp1))) return rc;
            if ((rc = launch_pack_fc_weight(s, fc2_w, wp2))) return rc;
        }
    }
    const int q_tiled = (p.screen && p.split16 && !core) ? 1 : 0;   // the projection writes the bf16 queries in the screen's fragment order
    // a call that goes straight to the streamed dense formulation (DAGL_FLAG_DENSE_HINT): the projection writes the split-fp16
    // features dense.hip consumes as well -- no separate splitting pass over the fp32 rows
    Split16Out split_out;
    const Split16Out* split_p = nullptr;
    if (!core && p.split16 && p.screen && mode == DAGL_MODE_ADAPTIVE && (mode_flags & DAGL_FLAG_DENSE_HINT)) {
        size_t off_dn = p.o_end;
        const size_t o_dn = carve(off_dn, dense_workspace_bytes(B, g));
        if (ws_bytes >= off_dn) { split_out = dense_split_buffers(at<char>(ws, o_dn), B, g); split_p = &split_out; }
    }
    ZeroList zl;
    if (split_p) dense_guard_rows(zl, B, g, split_out);
    if (prepared && split_p) { if ((rc = launch_zero_regions(s, zl))) return rc; }
    if (!prepared) {   // rows past the last patch (partial tile + guard tile) are streamed by the scans: keep them zero
        const int rx = feat_rows(g.N), rq = feat_rows(g.L);
        const int hx = feat_rows_h(g.N), hq = feat_rows_h(g.L);
        zl.add(X + (size_t)g.N * DS, (size_t)(rx - g.N) * DS * sizeof(float), B, (size_t)rx * DS * sizeof(float));
        zl.add(Wq + (size_t)g.L * DS, (size_t)(rq - g.L) * DS * sizeof(float), B, (size_t)rq * DS * sizeof(float));
        if (p.screen) {
            zl.add(Xh + (size_t)g.N * DSH, (size_t)(hx - g.N) * DSH * sizeof(uint16_t), B, (size_t)hx * DSH * sizeof(uint16_t));
            // (from the last PARTIAL 32-query tile on: in the fragment order its unused rows sit between the used ones)
            const int lq = g.L & ~31;
            zl.add(Wqh + (size_t)lq * DSH, (size_t)(hq - lq) * DSH * sizeof(uint16_t), B, (size_t)hq * DSH * sizeof(uint16_t));
        }
        zl.add(colsum, align_up((size_t)B * DS * sizeof(double), 16));
        zl.add(stats, 4 * sizeof(int64_t));
        if (p.screen) zl.add(at<int32_t>(ws, p.o_redo), align_up((size_t)B * n_qgroups * sizeof(int32_t), 16));
        if ((rc = launch_zero_regions(s, zl))) return rc;
    }

    // ---- stage 1: both projections, one launch -------------------------------------------------------------
    prof_mark(prof, s, 1);
    if (core) {
        if ((rc = launch_rows_to_feat(s, B, g.N, core->x_rows, X, Xh, rt))) return rc;
        if ((rc = launch_rows_to_feat(s, B, g.L, core->wq_rows, Wq, Wqh, rt))) return rc;
        if (mode != DAGL_MODE_TOPK)
            if ((rc = launch_colsum_rows(s, B, g.N, core->x_rows, colsum))) return rc;
    } else if (p.split16) {
        B1Tiers tiers_fused;               // (the fused entry points: conv_pair16_kernel wrote both tiers; dagl_ce_forward's split_map_kernel only the fine one)
        if (fin) {
            tiers_fused.hi2 = at<uint16_t>(ws, p.o_maphi2); tiers_fused.lo2 = at<uint16_t>(ws, p.o_maplo2);
            tiers_fused.amax = at<float>(ws, p.o_b1amax); tiers_fused.slots = conv16_blocks_per_head(g, heads, B / heads);
        }
        const float* b1s[4]; const float* b2s[4];
        for (int hd = 0; hd < 4; ++hd) {
            b1s[hd] = (fin && hd < heads) ? fin[hd].fc1_b : fc1_b;
            b2s[hd] = (fin && hd < heads) ? fin[hd].fc2_b : fc2_b;
        }
        if ((rc = launch_project16(s, B, g, 3, map_hi, map_lo, wp2h, b2s, X, (mode == DAGL_MODE_TOPK) ? nullptr : colsum,
                                   at<float>(ws, p.o_colpart), wp1h, b1s,
                                   Wq, Xh, Wqh, heads, rt, q_tiled, split_p,
                                   thr_in_proj ? &thr_all : nullptr, thr_in_proj ? B : 0, thr_in_proj ? at<float>(ws, p.o_thrpart) : nullptr,
                                   fin ? &tiers_fused : nullptr))) return rc;
    } else {
        if ((rc = launch_project(s, B, g, 3, b1p, wp2, fc2_b, X, colsum, wp1, fc1_b, Wq, Xh, Wqh, rt))) return rc;
    }

    // ---- stage 2: adaptive thresholds ----------------------------------------------------------------------
    bool fused_theta = false;
    prof_mark(prof, s, 2);
    SelectArgs sa;
    memset(&sa, 0, sizeof(sa));
    sa.B = B; sa.L = g.L; sa.N = g.N; sa.W = g.W; sa.wq = Wq; sa.x = X; sa.mode = mode; sa.k = k;
    sa.splits = p.splits; sa.tiles_per_split = p.tiles_per_split;
    EdgeArgs ea;
    memset(&ea, 0, sizeof(ea));
    ea.B = B; ea.L = g.L; ea.N = g.N; ea.mode = mode; ea.k = k; ea.splits = p.splits;
    ea.nb_idx = nbidx; ea.nb_wgt = nbwgt; ea.nb_cnt = nbcnt; ea.width = p.width;
    ea.nb_s = core ? core->nb_s : nullptr;
    AggArgs ag;
    memset(&ag, 0, sizeof(ag));
    ag.B = B; ag.g = g; ag.b2p = b2p; ag.nb_idx = nbidx; ag.nb_wgt = nbwgt; ag.nb_cnt = nbcnt; ag.width = p.width;
    ag.agg = agg;
    if (mode != DAGL_MODE_TOPK) {
        ThrFuse tf;
        if (fin) {                                   // finish the thr / bias heads here (their partial sums are per head)
            tf.part = at<float>(ws, p.o_thrpart); tf.imgs_per_head = imgs;
            for (int hd = 0; hd < heads; ++hd) { tf.thr_b[hd] = fin[hd].thr_b; tf.bias_b[hd] = fin[hd].bias_b; }
            tf.thr_out = at<float>(ws, p.o_thr); tf.bias_out = at<float>(ws, p.o_bias);
        } else {
            tf.bias_out = const_cast<float*>(bias);   // read only in this case
        }
        fused_theta = p.screen && mode == DAGL_MODE_ADAPTIVE;
        if (fused_theta) tf.theta_out = at<float>(ws, p.o_theta);
        if (p.screen && mode == DAGL_MODE_ADAPTIVE) tf.zero_out = at<float>(ws, p.o_traw);     // (run_dense's sampled pass takes maxima into it)
        if ((rc = launch_query_thresholds(s, B, g.L, g.N, Wq, colsum, thr, mt, core ? core->mu : nullptr, &tf))) return rc;
        sa.mt = mt; sa.bs = bias; ea.mt = mt; ea.bs = bias;
    }

    bool agg_done = false;           // the gather + weighted sum over the lists ran inside the overflow launches (adaptive, screened)
    bool ovf_active = false;         // set once the refine kernel has listed its overflowed queries (adaptive, screened)
    // the statistics read-back of the adaptive modes also carries the range word: a call that left the split-fp16 range is
    // re-run on the fp32 path right here (same arguments, DAGL_FLAG_EXACT_SCAN); without a read-back (top-k modes) the
    // poisoned output and dagl_ce_range_check report it
    auto rerun_exact = [&]() -> int {
        if (mode_flags & DAGL_FLAG_EXACT_SCAN) {      // already the fp32 path: the word was set by a NON-FINITE feature (round 6), the output is
            if (info) info->range_fallback = 1;       // NaN-filled as the reference's would be; nothing to re-run
            return DAGL_OK;
        }
        if (heads > 1) {            // stage entry point: no fp32 form of the four-head launch set; hand the call back (per-head path)
            if (info) { info->required_bytes = -1; info->range_fallback = 1; }
            set_error("dagl_ces_stage_forward: an operand left the split-fp16 range: use the per-head entry point");
            return DAGL_ERR_WORKSPACE;
        }
        if (info) info->range_fallback = 1;
        const int rc2 = ce_forward_impl(s, B, H, W, b1, b2, fin ? nullptr : thr, fin ? nullptr : bias, fc1_w, fc1_b, fc2_w, fc2_b,
                                        (mode_flags | DAGL_FLAG_EXACT_SCAN) & ~(DAGL_FLAG_WEIGHTS_PACKED | DAGL_FLAG_DENSE_HINT), k, out,
                                        ws, ws_bytes, info, dbg_deg, dbg_rowsum, dbg_agg, nullptr, fin, heads, nullptr);
        if (info) info->range_fallback = 1;
        return rc2;
    };
    // the per-query redo gathers one value patch per edge of a flagged row (~1 ns each), the dense formulation costs ~9 ps per
    // (query, key) PAIR whatever the mask: a call whose flagged rows hold more than 1/96 of all pairs goes dense
    const long long ovf_edge_limit = (long long)((double)BL * (double)g.N / 96.0);
    auto overflow_args = [&]() {
        OvfArgs oa;
        memset(&oa, 0, sizeof(oa));
        oa.B = B; oa.g = g; oa.x = X; oa.rows_x = feat_rows(g.N);
        oa.mt = mt; oa.bs = bias; oa.b2p = b2p; oa.list = at<int32_t>(ws, p.o_ovflist);
        oa.count = reinterpret_cast<const int32_t*>(stats + 3); oa.cap = p.ovf_cap;
        oa.qrows = at<float>(ws, p.o_ovfq); oa.scores = at<float>(ws, p.o_ovfscores); oa.ldn = (g.N + 31) / 32 * 32; oa.part = at<float>(ws, p.o_ovfpart);
        oa.agg = agg; oa.nb_cnt = nbcnt; oa.dbg_deg = dbg_deg; oa.dbg_rowsum = dbg_rowsum;
        oa.edges_run = stats;            // (stats[0]: zero since the start of the call, overwritten with the total by the statistics block)
        oa.flagged_edges = stats + 7; oa.edge_limit = ovf_edge_limit;
        return oa;
    };
    auto run_tail = [&](const AggArgs& ag2) -> int {
        int r;
        if (dbg_deg || dbg_rowsum)
            if ((r = launch_row_stats(s, BL, ag2.nb_wgt, ag2.nb_cnt, ag2.row_off, ag2.width, dbg_deg, dbg_rowsum))) return r;
        prof_mark(prof, s, 6);
        // short fixed-width lists and nobody asking for the aggregated rows: gather, weighted sum and fold in one kernel
        if (mode != DAGL_MODE_ADAPTIVE && ag2.row_off == nullptr && !ovf_active && !dbg_agg && !core) {
            if ((r = launch_aggregate_fold(s, ag2, out, heads, rt))) return r;
            prof_mark(prof, s, 7);                              // (the whole tail is booked on the gather stage)
            prof_mark(prof, s, 8);
            return DAGL_OK;
        }
        // (the few queries whose neighbourhood overflowed the lists are redone one by one, dense rows, overflow.hip; the list gather
        // skips them.  A call that does not wait had both done in the overflow launches; one that waits has its statistics on their
        // way to the host by now and queues the gathers here, under the round trip)
        if (!agg_done) {
            if ((r = launch_aggregate_direct(s, ag2))) return r;
            if (ovf_active) {           // (a call that waited for its verdict: the flagged rows' gathers behind the read-back)
                OvfArgs oa = overflow_args();
                oa.edges_run = nullptr;                     // (their edges are known: ovf_attend_kernel looks at flagged_edges)
                if ((r = launch_overflow_apply(s, oa))) return r;
            }
        }
        prof_mark(prof, s, 7);
        if (dbg_agg) DAGL_HIP_TRY(hipMemcpyAsync(dbg_agg, agg, BL * P * sizeof(float), hipMemcpyDeviceToDevice, s));
        if ((r = launch_fold(s, B, g, agg, out, heads, rt))) return r;
        prof_mark(prof, s, 8);
        return DAGL_OK;
    };

    if (p.wide) {
        // fixed-k neighbourhoods wider than the lists: scores, k-th largest, mask, softmax and weighted sum row by row
        prof_mark(prof, s, 3); prof_mark(prof, s, 4); prof_mark(prof, s, 5);
        if (heads > 1) { set_error("dagl_ces_stage_forward: k=%d > %d: use the per-head entry point", k, DAGL_MAX_TOPK); return DAGL_ERR_UNSUPPORTED; }
        if ((rc = launch_topk_wide(s, B, g, mode, p.k, Wq, X, mt, bias, b2p, at<char>(ws, p.o_wide), agg, deg, dbg_rowsum, rt))) return rc;
        prof_mark(prof, s, 6);
        if (dbg_deg) DAGL_HIP_TRY(hipMemcpyAsync(dbg_deg, deg, BL * sizeof(int32_t), hipMemcpyDeviceToDevice, s));
        if (dbg_agg) DAGL_HIP_TRY(hipMemcpyAsync(dbg_agg, agg, BL * P * sizeof(float), hipMemcpyDeviceToDevice, s));
        prof_mark(prof, s, 7);
        if ((rc = launch_fold(s, B, g, agg, out, heads, rt))) return rc;
        prof_mark(prof, s, 8);
        if (info) {
            if ((rc = launch_degree_stats(s, BL, deg, stats))) return rc;
            int64_t hs[5] = {0, 0, 0, 0, 0};
            if ((rc = read_back(s, stats, 5, hs))) return rc;
            if (rt.word != nullptr && (int32_t)hs[4] == rt.tag) return rerun_exact();
            info->path = 6; info->total_edges = hs[0]; info->max_degree = (int32_t)hs[1];
        }
        if (prof && prof->n_calls < prof->max_calls) ++prof->n_calls;
        return DAGL_OK;
    }

    // ---- stages 3-5: neighbour selection + edge softmax ------------------------------------------------------
    bool need_exact = !p.screen;
    int32_t* redo = nullptr;
    if (p.screen) {
        ScreenArgs sc;
        memset(&sc, 0, sizeof(sc));
        sc.B = B; sc.L = g.L; sc.N = g.N; sc.mode = mode; sc.wqh = Wqh; sc.xh = Xh; sc.q_tiled = q_tiled;
        sc.rows_qh = feat_rows_h(g.L); sc.rows_xh = feat_rows_h(g.N);
        sc.splits = p.s_splits; sc.steps_per_split = p.s_steps_per_split; sc.n_steps = p.s_steps; sc.sample = p.s_sample; sc.qblock = p.s_qblock;
        sc.gmax = at<float>(ws, p.o_gmax); sc.gkeep = p.s_gkeep; sc.theta = at<float>(ws, p.o_theta); sc.mt = mt; sc.bs = bias;
        sc.capseg = p.capseg; sc.cand = at<int2>(ws, p.o_scand);
        if (mode != DAGL_MODE_ADAPTIVE) { sc.spill = at<int2>(ws, p.o_spill); sc.spill_cnt = at<unsigned>(ws, p.o_spillcnt); }
        if (topk_policy) { sc.policy = policy_w; sc.sample_tight = p.s_sample_tight; sc.capseg_tight = p.capseg_tight; }
        redo = at<int32_t>(ws, p.o_redo);
#ifdef DAGL_ABLATION
        { static const int var = [] { const char* e = getenv("DAGL_SCREEN_VARIANT"); return e ? atoi(e) : 0; }(); sc.variant = var; }
#endif
        // most queries keep more keys than the screen has candidate slots for: the dense regime.  Lists are pointless
        // there; stream the dense formulation instead (dense.hip).  Reached after the screen found out, or directly when
        // the caller passes DAGL_FLAG_DENSE_HINT (its previous call on this module ended here): always correct, only
        // slower than the lists when the neighbourhoods are in fact sparse.
        auto run_dense = [&](bool features_split) -> int {
            size_t off = p.o_end;
            const size_t o_dn = carve(off, dense_workspace_bytes(B, g));
            if (info) { info->required_bytes = (int64_t)off; info->path = 4; }
            if (core && !core->lse) {
                if (info) info->required_bytes = -1;
                set_error("dagl_ce_core_forward: dense neighbourhoods (most queries keep more than %d keys) do not fit fixed-width lists: "
                          "use dagl_ce_core_dense_forward", DAGL_FAST_CAP);
                return DAGL_ERR_UNSUPPORTED;
            }
            if (ws_bytes < off) {
                set_error("dagl_ce_forward: dense neighbourhoods need workspace %zu B, have %zu B", off, ws_bytes);
                return DAGL_ERR_WORKSPACE;
            }
            // the softmax's shift, known up front: every row's largest score, exactly -- a top-1 screen (sampled pass -> theta = the
            // largest sampled S~ less the band -> filter pass) and the exact scores of its few candidates (rowmax_exact_kernel).  Up to
            // round 4: an upper bound from one full bf16 scan, whose 1.6 % became > 18 units of a logit beyond ~580 and sent whole
            // blocks through dense_attend_kernel a second time (every block of bench.py's default map: 1.52 ms for 0.81)
            ScreenArgs s1 = sc;
            s1.mode = DAGL_MODE_TOPK; s1.mt = nullptr; s1.bs = nullptr; s1.policy = nullptr; s1.gate = nullptr;
            s1.spill = nullptr; s1.spill_cnt = nullptr; s1.seg_max = 1;      // (no spill: a row with more candidates than slots keeps its upper bound)
            s1.theta_max = at<int>(ws, p.o_traw); s1.theta = at<float>(ws, p.o_traw);     // (zeroed by query_thresholds_kernel)
            if ((rc = launch_screen(s, s1, 0))) return rc;
            if ((rc = launch_screen(s, s1, 1))) return rc;
            float* smax = at<float>(ws, p.o_smax);
            RefineArgs r1;
            memset(&r1, 0, sizeof(r1));
            r1.B = B; r1.L = g.L; r1.N = g.N; r1.mode = DAGL_MODE_TOPK; r1.k = 1; r1.splits = p.s_splits; r1.capseg = p.capseg;
            r1.wq = Wq; r1.x = X; r1.rows_q = feat_rows(g.L); r1.rows_x = feat_rows(g.N);
            r1.cand = s1.cand; r1.theta = s1.theta; r1.mt = mt; r1.bs = bias;
            if ((rc = launch_rowmax_exact(s, r1, smax))) return rc;
            prof_mark(prof, s, 6);          // (stage "gather" of a dense call = value-map split + dense_attend_kernel + combine)
            if ((rc = launch_dense_attend(s, B, g, Wq, X, mt, bias, smax, b2p, at<char>(ws, o_dn), agg, dbg_deg, dbg_rowsum, stats, rt,
                                          core ? core->lse : nullptr, features_split, info != nullptr))) return rc;
            prof_mark(prof, s, 7);
            if (dbg_agg) DAGL_HIP_TRY(hipMemcpyAsync(dbg_agg, agg, BL * P * sizeof(float), hipMemcpyDeviceToDevice, s));
            if ((rc = launch_fold(s, B, g, agg, out, heads, rt))) return rc;
            prof_mark(prof, s, 8);
            if (info) {
                int64_t hd[DENSE_RERUN_STAT + 1] = {0};
                if ((rc = read_back(s, stats, DENSE_RERUN_STAT + 1, hd))) return rc;
                if (rt.word != nullptr && (int32_t)hd[4] == rt.tag) {
                    if (core) { info->range_fallback = 1; return DAGL_OK; }      // (output NaN-filled; dagl_ce_core_dense_forward re-runs the GEMM form)
                    return rerun_exact();
                }
                info->total_edges = hd[0]; info->max_degree = (int32_t)hd[1];
                info->redone_queries = hd[2];                       // queries whose degree exceeds the lists' width
                info->dense_rerun_blocks = (int32_t)hd[DENSE_RERUN_STAT];       // blocks of 64 queries dense_attend_kernel ran a second time
            }
            if (prof && prof->n_calls < prof->max_calls) ++prof->n_calls;
            return DAGL_OK;
        };
        if (mode == DAGL_MODE_ADAPTIVE && (mode_flags & DAGL_FLAG_DENSE_HINT) && (!core || core->lse)) {
            prof_mark(prof, s, 3); prof_mark(prof, s, 4); prof_mark(prof, s, 5);
            return run_dense(split_p != nullptr);
        }
        prof_mark(prof, s, 3);
        if (mode != DAGL_MODE_ADAPTIVE) {                       // top-k threshold from the sampling pass
            if ((rc = launch_screen(s, sc, 0))) return rc;
            if ((rc = launch_screen_theta(s, (int)BL, p.s_splits * 2 * p.s_gkeep, k, sc.gmax, at<float>(ws, p.o_theta), nullptr, sc.spill_cnt))) return rc;
        }
        if (mode != DAGL_MODE_TOPK && !fused_theta)             // adaptive threshold; the intersection mode takes the larger
            if ((rc = launch_adaptive_theta(s, BL, mt, bias, at<float>(ws, p.o_theta), mode == DAGL_MODE_ADAPTIVE_TOPK))) return rc;
        prof_mark(prof, s, 4);
        if ((rc = launch_screen(s, sc, 1))) return rc;
        prof_mark(prof, s, 5);
        RefineArgs ra;
        memset(&ra, 0, sizeof(ra));
        ra.B = B; ra.L = g.L; ra.N = g.N; ra.mode = mode; ra.k = k; ra.splits = p.s_splits; ra.capseg = p.capseg;
        ra.width = p.width; ra.wq = Wq; ra.x = X; ra.rows_q = feat_rows(g.L); ra.rows_x = feat_rows(g.N);
        ra.mt = mt; ra.bs = bias; ra.cand = sc.cand; ra.theta = sc.theta; ra.spill = sc.spill; ra.spill_cnt = sc.spill_cnt;
        ra.nb_idx = nbidx; ra.nb_wgt = nbwgt; ra.nb_cnt = nbcnt; ra.redo_flags = redo; ra.n_qgroups_exact = n_qgroups;
        ra.stats = stats; ra.nb_s = core ? core->nb_s : nullptr;
        if (p.ovf_cap > 0) {
            ra.ovf_list = at<int32_t>(ws, p.o_ovflist); ra.ovf_count = reinterpret_cast<int32_t*>(stats + 3); ra.ovf_cap = p.ovf_cap;
            ra.ovf_qrows = at<float>(ws, p.o_ovfq);
            ra.heavy_list = at<int32_t>(ws, p.o_heavy); ra.heavy_count = reinterpret_cast<int32_t*>(stats + 3) + 1;   // (cleared with the counters)
            ovf_active = true;
        }
        if (topk_policy) { ra.policy = policy_w; ra.capseg_tight = p.capseg_tight; }
        if ((rc = launch_refine(s, ra))) return rc;
        if (topk_policy && !prepared) {
            // cold workspace: did the sampled threshold overflow most queries' slots (natural-image features)?  Then the policy word
            // flips here and sampling, threshold, filter and refine run once more, tight, in this very call -- four launches that
            // exit at once otherwise -- instead of every query group taking the fp32 redo pass (2.7 ms at 256^2)
            if ((rc = launch_topk_policy(s, stats, policy_w, gate_w, redo, (int)(B * n_qgroups), (long long)BL))) return rc;
            ScreenArgs sc2 = sc; sc2.gate = gate_w;
            if ((rc = launch_screen(s, sc2, 0))) return rc;
            if ((rc = launch_screen_theta(s, (int)BL, p.s_splits * 2 * p.s_gkeep, k, sc2.gmax, at<float>(ws, p.o_theta), gate_w, sc2.spill_cnt))) return rc;
            // (the intersection mode takes the larger of the two thresholds: max(adaptive, theta) is idempotent, so the ungated
            // kernel is harmless when the re-run did not run)
            if (mode != DAGL_MODE_TOPK && !fused_theta)
                if ((rc = launch_adaptive_theta(s, BL, mt, bias, at<float>(ws, p.o_theta), true))) return rc;
            if ((rc = launch_screen(s, sc2, 1))) return rc;
            RefineArgs ra2 = ra; ra2.gate = gate_w;
            if ((rc = launch_refine(s, ra2))) return rc;
        }
        if (info) info->path = 3;
        if (mode == DAGL_MODE_ADAPTIVE) {
            // Dense neighbourhoods need host-side CSR sizing, so the verdict must be read back.  Everything it consists of
            // is known once the flagged queries' chunk statistics exist (their true degrees): the copy is queued there,
            // the optimistic gather, the flagged rows' weighted sums and the fold behind it, and the host waits for the
            // copy alone -- the device works through the round trip and through the caller's next launches.
            // DAGL_FLAG_NO_WAIT: the verdict is formed on the device, the call returns without reading it (no host round trip:
            // the adaptive forward can be captured into a HIP graph); an unserved call is NaN-filled, never wrong
            const bool no_wait = (mode_flags & DAGL_FLAG_NO_WAIT) && ovf_active && !dbg_deg && !dbg_rowsum && !dbg_agg && !core && heads == 1;
            if (no_wait) {
                if (rt.tag == 0) rt.tag = next_call_tag();
                rt.veto = reinterpret_cast<const int32_t*>(stats + 8);
            }
            if (ovf_active) {           // the flagged rows redone (three launches that exit at once when there are none) + the call's statistics
                const OvfArgs oa = overflow_args();
                // (a call that does not wait shares the scores' launch with the gather over the lists; one that waits keeps the gather
                // behind the read-back: the host round trip runs under it)
                if ((rc = launch_overflow_rows(s, oa, no_wait ? &ag : nullptr, BL, stats, no_wait ? reinterpret_cast<int32_t*>(stats + 8) : nullptr,
                                               rt.tag))) return rc;
                agg_done = no_wait;
            } else {
                if ((rc = launch_degree_stats(s, BL, nbcnt, stats))) return rc;
            }
            if (no_wait) {
                if ((rc = run_tail(ag))) return rc;
                if (prof && prof->n_calls < prof->max_calls) ++prof->n_calls;
                return DAGL_OK;                                  // info: path 3, statistics not read (-1)
            }
            bool pending = false;
            if ((rc = read_back_begin(s, stats, 8, &pending))) return rc;
            if ((rc = run_tail(ag))) return rc;
            int64_t hs[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            if ((rc = read_back_end(s, stats, 8, pending, hs))) return rc;
            if (rt.word != nullptr && (int32_t)hs[4] == rt.tag) {
                if (core) { if (info) info->range_fallback = 1; return DAGL_OK; }   // (training entry point: non-finite features; out is NaN-filled)
                return rerun_exact();
            }
            if (info) info->redone_queries = hs[2];
            // most queries overflow, or the flagged rows are too heavy to redo one by one (the attend kernels saw the same
            // word and left them alone): dense regime
            const bool heavy_rows = ovf_active && hs[2] > 0 && hs[2] <= p.ovf_cap && hs[7] > ovf_edge_limit;
            const bool mostly = hs[2] * 2 > (int64_t)BL || heavy_rows;
            if (hs[2] == 0 || (!mostly && ovf_active && hs[2] <= p.ovf_cap)) {   // (few overflowed queries: redone in-stream)
                if (info) { info->total_edges = hs[0]; info->max_degree = (int32_t)hs[1]; }
                if (prof && prof->n_calls < prof->max_calls) ++prof->n_calls;
                return DAGL_OK;
            }
            ovf_active = false; agg_done = false;
            if (mostly) return run_dense(false);
            need_exact = true;                                   // redo everything with the fp32 scan (CSR capable)
        } else {
            // top-k modes: query groups whose candidate slots overflowed are redone by the fp32 scan below; it
            // exits at once for every other group, so no host round trip is needed
            sa.run_flags = redo; ea.run_flags = redo; sa.run_count = stats + 2; ea.run_count = stats + 2;
            need_exact = true;
            if (info && (dbg_deg || dbg_rowsum || dbg_agg)) {        // debug entry point: report the overflow count
                int64_t hs[4] = {0, 0, 0, 0};
                if ((rc = read_back(s, stats, 4, hs))) return rc;
                info->redone_queries = hs[2];
            }
        }
    } else {
        prof_mark(prof, s, 3);
        prof_mark(prof, s, 4);
    }

    if (need_exact) {
        if (mode == DAGL_MODE_ADAPTIVE) {
            int32_t* lidx = at<int32_t>(ws, p.o_lidx);
            float* lval = at<float>(ws, p.o_lval);
            DAGL_HIP_TRY(hipMemsetAsync(cnt, 0, BL * sizeof(int32_t), s));
            sa.cnt = cnt; sa.seg_cnt = segcnt; sa.list_idx = lidx; sa.list_val = lval;
            if ((rc = launch_score_select(s, sa, 0))) return rc;
            if (!p.screen) prof_mark(prof, s, 5);
            if ((rc = launch_row_degree(s, (int)BL, p.splits * 2, segcnt, segrel, deg, stats))) return rc;
            int64_t hstats[2] = {0, 0};
            if ((rc = read_back(s, stats, 2, hstats))) return rc;
            if (info) { info->total_edges = hstats[0]; info->max_degree = (int32_t)hstats[1]; info->path = 0; }
            if (hstats[1] <= DAGL_FAST_CAP) {
                ea.cnt = deg; ea.list_idx = lidx; ea.list_val = lval; ea.row_off = nullptr;
                if ((rc = launch_edge_softmax(s, ea))) return rc;
            } else {
                // two-pass CSR: exact degrees are known, refill deterministically at per-lane cursors
                size_t off = p.o_end;
                const size_t e = (size_t)hstats[0];
                const size_t o_ci = carve(off, e * sizeof(int32_t));
                const size_t o_cv = carve(off, e * sizeof(float));
                const size_t o_ni = carve(off, e * sizeof(int32_t));
                const size_t o_nw = carve(off, e * sizeof(float));
                if (info) { info->required_bytes = (int64_t)off; info->path = 1; }
                if (core) {
                    if (info) info->required_bytes = -1;
                    set_error("dagl_ce_core_forward: dense neighbourhoods (max degree %lld > %d) do not fit fixed-width lists: "
                              "use dagl_ce_core_dense_forward", (long long)hstats[1], DAGL_FAST_CAP);
                    return DAGL_ERR_UNSUPPORTED;
                }
                if (ws_bytes < off) {
                    set_error("dagl_ce_forward: dense neighbourhoods (max degree %lld, %lld edges) need workspace %zu B, have %zu B",
                              (long long)hstats[1], (long long)hstats[0], off, ws_bytes);
                    return DAGL_ERR_WORKSPACE;
                }
                if ((rc = launch_row_scan(s, (int)BL, deg, rowoff))) return rc;
                sa.list_idx = at<int32_t>(ws, o_ci); sa.list_val = at<float>(ws, o_cv); sa.seg_rel = segrel; sa.row_off = rowoff;
                if ((rc = launch_score_select(s, sa, 1))) return rc;
                ea.cnt = deg; ea.list_idx = sa.list_idx; ea.list_val = sa.list_val; ea.row_off = rowoff;
                ea.nb_idx = at<int32_t>(ws, o_ni); ea.nb_wgt = at<float>(ws, o_nw);
                if ((rc = launch_edge_softmax(s, ea))) return rc;
                ag.nb_idx = ea.nb_idx; ag.nb_wgt = ea.nb_wgt; ag.row_off = rowoff;
            }
        } else {
            sa.cand_idx = at<int32_t>(ws, p.o_cidx); sa.cand_val = at<float>(ws, p.o_cval);
            ea.cand_idx = sa.cand_idx; ea.cand_val = sa.cand_val;
            // DAGL_FLAG_NO_REDO: the caller has seen this workspace's recent calls without redo work and does without the launch (4.7 us
            // that find nothing); a call that flagged a group after all is NaN-filled by the gather kernel and reported (sticky)
            const bool no_redo = p.screen && (mode_flags & DAGL_FLAG_NO_REDO) && fin && heads == 1 && !core && !dbg_deg && !dbg_rowsum && !dbg_agg;
            if (no_redo) {
                ag.unserved = stats + 2; ag.unserved_sticky = reinterpret_cast<int32_t*>(stats + 13);
            } else if (p.screen) {
                // redo pass behind the screen: scan + merge of the flagged groups in one launch (exits after one load when nothing
                // is flagged); its grid barrier counts in stats[3] (cleared with the call's counters, unused by the top-k modes)
                if ((rc = launch_topk_redo(s, sa, ea, mode == DAGL_MODE_TOPK ? 2 : 3, reinterpret_cast<unsigned*>(stats + 3),
                                           topk_policy ? policy_w : nullptr))) return rc;
            } else {
                if ((rc = launch_score_select(s, sa, mode == DAGL_MODE_TOPK ? 2 : 3))) return rc;
                prof_mark(prof, s, 5);
                if ((rc = launch_edge_softmax(s, ea))) return rc;
            }
            if (info && !p.screen) { info->path = 2; info->max_degree = k; }
        }
    }

    // ---- stages 6-7: gather + weighted sum, fold ---------------------------------------------------------------
    if ((rc = run_tail(ag))) return rc;
    if (prof && prof->n_calls < prof->max_calls) ++prof->n_calls;
    return DAGL_OK;
}

}  // namespace dagl

using namespace dagl;

extern "C" {

int dagl_version(void) { return DAGL_ABI_VERSION; }

const char* dagl_last_error(void) { return g_err; }

int dagl_device_check(void) { return check_device(); }

size_t dagl_ce_workspace_bytes(int B, int H, int W, int mode, int k) {
    Plan p;
    if (make_plan(B, H, W, mode, k, p)) return 0;
    return p.o_end;
}

int dagl_ce_forward(void* stream, int B, int H, int W, const float* b1, const float* b2, const float* thr,
                    const float* bias, const float* fc1_w, const float* fc1_b, const float* fc2_w,
                    const float* fc2_b, int mode, int k, float* out, void* workspace, size_t ws_bytes,
                    dagl_ce_info* info) {
    return ce_forward_impl((hipStream_t)stream, B, H, W, b1, b2, thr, bias, fc1_w, fc1_b, fc2_w, fc2_b, mode, k, out,
                           workspace, ws_bytes, info, nullptr, nullptr, nullptr);
}

int dagl_ce_forward_debug(void* stream, int B, int H, int W, const float* b1, const float* b2, const float* thr,
                          const float* bias, const float* fc1_w, const float* fc1_b, const float* fc2_w,
                          const float* fc2_b, int mode, int k, float* out, void* workspace, size_t ws_bytes,
                          dagl_ce_info* info, int32_t* deg_out, float* rowsum_out, float* agg_out) {
    return ce_forward_impl((hipStream_t)stream, B, H, W, b1, b2, thr, bias, fc1_w, fc1_b, fc2_w, fc2_b, mode, k, out,
                           workspace, ws_bytes, info, deg_out, rowsum_out, agg_out);
}

int dagl_profile_create(int max_calls, dagl_profile** out) {
    DAGL_REQUIRE(max_calls >= 1 && max_calls <= 4096 && out, "dagl_profile_create: bad argument");
    Profile* p = new Profile();
    p->max_calls = max_calls;
    const size_t n = (size_t)max_calls * (DAGL_N_STAGES + 1);
    p->ev = new hipEvent_t[n];
    for (size_t i = 0; i < n; ++i) {
        hipError_t e = hipEventCreate(&p->ev[i]);
        if (e != hipSuccess) {
            for (size_t j = 0; j < i; ++j) (void)hipEventDestroy(p->ev[j]);
            delete[] p->ev; delete p;
            return hip_fail(e, "hipEventCreate");
        }
    }
    *out = reinterpret_cast<dagl_profile*>(p);
    return DAGL_OK;
}

int dagl_profile_destroy(dagl_profile* prof) {
    Profile* p = reinterpret_cast<Profile*>(prof);
    if (!p) return DAGL_OK;
    const size_t n = (size_t)p->max_calls * (DAGL_N_STAGES + 1);
    for (size_t i = 0; i < n; ++i) (void)hipEventDestroy(p->ev[i]);
    delete[] p->ev; delete p;
    return DAGL_OK;
}

int dagl_profile_select_stage(dagl_profile* prof, int stage) {
    Profile* p = reinterpret_cast<Profile*>(prof);
    DAGL_REQUIRE(p && stage >= -1 && stage < DAGL_N_STAGES, "dagl_profile_select_stage: bad argument");
    p->only_stage = stage;
    p->n_calls = 0;
    return DAGL_OK;
}

int dagl_profile_reset(dagl_profile* prof) {
    Profile* p = reinterpret_cast<Profile*>(prof);
    DAGL_REQUIRE(p, "dagl_profile_reset: null profile");
    p->n_calls = 0;
    return DAGL_OK;
}

int dagl_profile_read(dagl_profile* prof, int* n_calls, float* stage_ms, int capacity_calls) {
    Profile* p = reinterpret_cast<Profile*>(prof);
    DAGL_REQUIRE(p && n_calls, "dagl_profile_read: null argument");
    *n_calls = p->n_calls;
    if (!stage_ms) return DAGL_OK;
    const int n = p->n_calls < capacity_calls ? p->n_calls : capacity_calls;
    for (int c = 0; c < n; ++c) {
        hipEvent_t* e = p->ev + (size_t)c * (DAGL_N_STAGES + 1);
        DAGL_HIP_TRY(hipEventSynchronize(e[p->only_stage < 0 ? DAGL_N_STAGES : p->only_stage + 1]));
        for (int st = 0; st < DAGL_N_STAGES; ++st) {
            float ms = 0.f;
            if (p->only_stage < 0 || st == p->only_stage) DAGL_HIP_TRY(hipEventElapsedTime(&ms, e[st], e[st + 1]));
            stage_ms[(size_t)c * DAGL_N_STAGES + st] = ms;
        }
    }
    return DAGL_OK;
}

int dagl_ce_forward_profiled(void* stream, int B, int H, int W, const float* b1, const float* b2, const float* thr,
                             const float* bias, const float* fc1_w, const float* fc1_b, const float* fc2_w,
                             const float* fc2_b, int mode, int k, float* out, void* workspace, size_t ws_bytes,
                             dagl_ce_info* info, dagl_profile* prof) {
    return ce_forward_impl((hipStream_t)stream, B, H, W, b1, b2, thr, bias, fc1_w, fc1_b, fc2_w, fc2_b, mode, k, out,
                           workspace, ws_bytes, info, nullptr, nullptr, nullptr, reinterpret_cast<Profile*>(prof));
}

int dagl_ce_forward_fused(void* stream, int B, int H, int W, const float* x, const float* g_w, const float* g_b,
                          const float* theta_w, const float* theta_b, const float* thr_w, const float* thr_b,
                          const float* bias_w, const float* bias_b, const float* fc1_w, const float* fc1_b,
                          const float* fc2_w, const float* fc2_b, int mode, int k, float* out, void* workspace,
                          size_t ws_bytes, dagl_ce_info* info, dagl_profile* prof) {
    FusedIn fin{x, g_w, g_b, theta_w, theta_b, thr_w, thr_b, bias_w, bias_b, fc1_w, fc1_b, fc2_w, fc2_b};
    return ce_forward_impl((hipStream_t)stream, B, H, W, nullptr, nullptr, nullptr, nullptr, fc1_w, fc1_b, fc2_w, fc2_b,
                           mode, k, out, workspace, ws_bytes, info, nullptr, nullptr, nullptr,
                           reinterpret_cast<Profile*>(prof), &fin);
}

size_t dagl_ces_stage_workspace_bytes(int B, int H, int W, int mode, int k) {
    Plan p;
    if (B < 1 || make_plan(4 * B, H, W, mode, k, p)) return 0;
    return p.o_end + align_up((size_t)B * 64 * H * W * sizeof(float), 256);
}

int dagl_ces_stage_forward(void* stream, int B, int H, int W, const float* x, const dagl_ce_weights* heads4,
                           const float* mix_w, const float* mix_b, int mode, int k, float* out, void* workspace,
                           size_t ws_bytes, dagl_ce_info* info, dagl_profile* prof) {
    DAGL_REQUIRE(B >= 1 && x && heads4 && mix_w && mix_b && out, "dagl_ces_stage_forward: bad argument");
    Plan p;
    int rc = make_plan(4 * B, H, W, mode, k, p);
    if (rc) return rc;
    const size_t cat_bytes = align_up((size_t)B * 64 * H * W * sizeof(float), 256);
    if (info) info->required_bytes = (int64_t)(p.o_end + cat_bytes);
    DAGL_REQUIRE(workspace != nullptr && ((uintptr_t)workspace % 256) == 0, "dagl_ces_stage_forward: workspace must be 256-byte aligned");
    if (ws_bytes < p.o_end + cat_bytes) {
        set_error("dagl_ces_stage_forward: workspace %zu B < required %zu B", ws_bytes, p.o_end + cat_bytes);
        return DAGL_ERR_WORKSPACE;
    }
    FusedIn fin[4];
    for (int h = 0; h < 4; ++h) {
        const dagl_ce_weights& w = heads4[h];
        fin[h] = FusedIn{x, w.g_w, w.g_b, w.theta_w, w.theta_b, w.thr_w, w.thr_b, w.bias_w, w.bias_b,
                         w.fc1_w, w.fc1_b, w.fc2_w, w.fc2_b};
        DAGL_REQUIRE(w.g_w && w.g_b && w.theta_w && w.theta_b && w.fc1_w && w.fc1_b && w.fc2_w && w.fc2_b,
                     "dagl_ces_stage_forward: head %d has a null weight pointer", h);
    }
    float* cat = reinterpret_cast<float*>(static_cast<char*>(workspace) + p.o_end);       // [B,64,H,W]
    // a dense adaptive neighbourhood asks for more workspace than planned: give the block everything up to the concat map
    rc = ce_forward_impl((hipStream_t)stream, 4 * B, H, W, nullptr, nullptr, nullptr, nullptr, fin[0].fc1_w, fin[0].fc1_b,
                         fin[0].fc2_w, fin[0].fc2_b, mode, k, cat, workspace, p.o_end, info, nullptr, nullptr, nullptr,
                         reinterpret_cast<Profile*>(prof), fin, 4);
    if (rc) {
        if (rc == DAGL_ERR_WORKSPACE && info) info->required_bytes = -1;    // dense neighbourhoods: use the per-head entry point
        return rc;
    }
    return launch_stage_mix((hipStream_t)stream, B, H * W, cat, x, mix_w, mix_b, out);
}

int dagl_ce_range_check(void* stream, int B, int H, int W, int mode, int k, void* workspace, size_t ws_bytes,
                        int* violated) {
    DAGL_REQUIRE(workspace && violated, "dagl_ce_range_check: null pointer");
    Plan p;
    int rc = make_plan(B, H, W, mode, k, p);
    if (rc) return rc;
    DAGL_REQUIRE(ws_bytes >= p.o_end && ((uintptr_t)workspace % 256) == 0, "dagl_ce_range_check: not the workspace of such a call");
    *violated = 0;
    if (mode & DAGL_FLAG_EXACT_SCAN) return DAGL_OK;                      // the fp32 path has no such range (and always waits)
    // ONE read-back (one synchronisation) of stats[2..13]: [2] flagged queries of the last call, [4] range word, [8] veto word, [9] top-k
    // policy, [13] unserved DAGL_FLAG_NO_REDO call
    int64_t hw[12] = {0};
    int64_t* st = reinterpret_cast<int64_t*>(static_cast<char*>(workspace) + p.o_stats);
    if ((rc = read_back((hipStream_t)stream, st + 2, 12, hw))) return rc;
    const bool topk_screen = (mode & 0xff) != DAGL_MODE_ADAPTIVE && p.screen;
    // sticky: the word keeps the tag of the last call that left the range until it is read here (calls that reuse a
    // prepared workspace do not clear it), so a poll every n-th call sees a violation of ANY call since the last poll
    if ((int32_t)hw[2] != 0) { *violated |= 1; DAGL_HIP_TRY(hipMemsetAsync(st + 4, 0, sizeof(int64_t), (hipStream_t)stream)); }
    // bit 2 (not sticky: the count is cleared by every call): the last call's redo pass of the top-k modes had work
    if (topk_screen && hw[0] > 0) *violated |= 4;
    if (topk_screen && (int32_t)hw[7] != 0) *violated |= 8;            // bit 3: the workspace's threshold policy word says "tight"
    // bit 4 (sticky): a DAGL_FLAG_NO_REDO call had flagged groups (its output is NaN-filled)
    if (topk_screen && (int32_t)hw[11] != 0) { *violated |= 16; DAGL_HIP_TRY(hipMemsetAsync(st + 13, 0, sizeof(int64_t), (hipStream_t)stream)); }
    // bit 1: a DAGL_FLAG_NO_WAIT call was not served in-stream (likewise sticky)
    if ((mode & 0xff) == DAGL_MODE_ADAPTIVE && (int32_t)hw[6] != 0) {
        *violated |= 2; DAGL_HIP_TRY(hipMemsetAsync(st + 8, 0, sizeof(int64_t), (hipStream_t)stream));
    }
    return DAGL_OK;
}

int dagl_ce_list_width(int mode, int k) {
    const int m = mode & 0xff;
    if (m == DAGL_MODE_ADAPTIVE) return DAGL_FAST_CAP;
    if ((m == DAGL_MODE_TOPK || m == DAGL_MODE_ADAPTIVE_TOPK) && k >= 1 && k <= DAGL_MAX_TOPK) return k;
    set_error("dagl_ce_list_width: bad mode 0x%x / k=%d", mode, k);
    return DAGL_ERR_INVALID;
}

int dagl_ce_core_forward(void* stream, int B, int H, int W, const float* wq_rows, const float* x_rows, const float* b2,
                         const float* thr, const float* bias, int mode, int k, float* out, int32_t* nb_idx,
                         float* nb_wgt, float* nb_s, int32_t* nb_cnt, float* mu, void* workspace, size_t ws_bytes,
                         dagl_ce_info* info) {
    CoreIn core{wq_rows, x_rows, nb_idx, nb_wgt, nb_s, nb_cnt, mu, nullptr};
    return ce_forward_impl((hipStream_t)stream, B, H, W, nullptr, b2, thr, bias, nullptr, nullptr, nullptr, nullptr, mode,
                           k, out, workspace, ws_bytes, info, nullptr, nullptr, nullptr, nullptr, nullptr, 1, &core);
}

static size_t backward_offsets(int B, const Grid& g, int width, size_t o[13], size_t* sort_temp) {
    size_t off = 0;
    const size_t BL = (size_t)B * g.L, E = BL * width, n_keys = (size_t)B * g.N;
    o[0] = carve(off, BL * P * sizeof(float));                                  // d agg
    o[1] = carve(off, (size_t)B * g.Hp * g.Wp * CH * sizeof(float));            // b2 padded NHWC
    o[2] = carve(off, E * sizeof(float));                                       // d S
    o[3] = carve(off, BL * sizeof(float));                                      // d mu
    o[4] = carve(off, (size_t)B * DS * sizeof(double) + (size_t)B * D * sizeof(float));   // column sums, d Xbar
    for (int i = 5; i < 9; ++i) o[i] = carve(off, E * sizeof(uint32_t));        // sort keys / edge ids, in and out
    o[9] = carve(off, 2 * n_keys * sizeof(uint32_t));                           // run of every key
    const size_t tb = edge_sort_temp_bytes(E, n_keys);
    o[10] = carve(off, tb);
    o[11] = carve(off, edge_rowbuf_floats(E) * sizeof(float));
    o[12] = carve(off, edge_part_floats(E) * sizeof(float));
    if (sort_temp) *sort_temp = tb;
    return off;
}

size_t dagl_ce_core_backward_workspace_bytes(int B, int H, int W, int mode, int k) {
    const int width = dagl_ce_list_width(mode, k);
    if (B < 1 || H < 1 || W < 1 || width < 0) return 0;
    size_t o[13];
    return backward_offsets(B, make_grid(H, W), width, o, nullptr);
}

int dagl_ce_core_backward(void* stream, int B, int H, int W, int mode, int k, const float* wq_rows, const float* x_rows,
                          const float* b2, const float* thr, const float* bias, const int32_t* nb_idx,
                          const float* nb_wgt, const float* nb_s, const int32_t* nb_cnt, const float* mu,
                          const float* d_out, float* d_wq_rows, float* d_x_rows, float* d_b2, float* d_thr,
                          float* d_bias, void* workspace, size_t ws_bytes) {
    const int width = dagl_ce_list_width(mode, k);
    if (width < 0) return width;
    const int m = mode & 0xff;
    DAGL_REQUIRE(B >= 1 && H >= 1 && W >= 1, "dagl_ce_core_backward: bad shape");
    DAGL_REQUIRE(wq_rows && x_rows && b2 && nb_idx && nb_wgt && nb_s && nb_cnt && d_out && d_wq_rows && d_x_rows && d_b2,
                 "dagl_ce_core_backward: null tensor pointer");
    if (m != DAGL_MODE_TOPK)
        DAGL_REQUIRE(thr && bias && mu && d_thr && d_bias, "dagl_ce_core_backward: thr/bias/mu and their gradients required in adaptive modes");
    DAGL_REQUIRE(workspace != nullptr && ((uintptr_t)workspace % 256) == 0, "dagl_ce_core_backward: workspace must be 256-byte aligned");
    const Grid g = make_grid(H, W);
    DAGL_REQUIRE((size_t)B * g.L * width < (1ull << 31) && (size_t)B * g.N < (1ull << 31), "dagl_ce_core_backward: batch too large for 32-bit edge ids");
    size_t o[13], sort_temp = 0;
    const size_t need = backward_offsets(B, g, width, o, &sort_temp);
    if (ws_bytes < need) {
        set_error("dagl_ce_core_backward: workspace %zu B < required %zu B", ws_bytes, need);
        return DAGL_ERR_WORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    BwdArgs a;
    memset(&a, 0, sizeof(a));
    a.B = B; a.g = g; a.mode = m; a.width = width;
    a.wq_rows = wq_rows; a.x_rows = x_rows; a.thr = thr; a.bs = bias; a.mu = mu;
    a.nb_idx = nb_idx; a.nb_wgt = nb_wgt; a.nb_s = nb_s; a.nb_cnt = nb_cnt; a.dout = d_out;
    a.dagg = at<float>(workspace, o[0]);
    float* b2p = at<float>(workspace, o[1]);
    a.b2p = b2p; a.dS = at<float>(workspace, o[2]); a.dmu = at<float>(workspace, o[3]);
    double* colsum = at<double>(workspace, o[4]);
    float* dxbar = reinterpret_cast<float*>(colsum + (size_t)B * DS);
    a.colsum = colsum;
    a.dwq_rows = d_wq_rows; a.dx_rows = d_x_rows; a.dthr = d_thr; a.dbias = d_bias;
    BwdSortWs w;
    w.keys_in = at<uint32_t>(workspace, o[5]); w.keys_out = at<uint32_t>(workspace, o[6]);
    w.vals_in = at<uint32_t>(workspace, o[7]); w.vals_out = at<uint32_t>(workspace, o[8]);
    w.seg = at<uint32_t>(workspace, o[9]); w.temp = at<void>(workspace, o[10]); w.temp_bytes = sort_temp;
    w.rowbuf = at<float>(workspace, o[11]); w.part = at<float>(workspace, o[12]);
    int rc;
    if ((rc = launch_pad_nhwc(s, B, H, W, b2, b2p))) return rc;
    if (m != DAGL_MODE_TOPK)
        if ((rc = launch_colsum_rows(s, B, g.N, x_rows, colsum))) return rc;
    return launch_core_backward(s, a, w, dxbar, d_b2);
}

// forward on the inference path's streamed kernel (split-fp16 S and A V in one pass over the keys) when the image has enough
// keys for the screen's machinery it borrows (row-maximum scan); the chunked fp32 GEMM formulation otherwise
static bool dense_core_streamed(int H, int W) { return (int64_t)H * W >= SCREEN_MIN_KEYS; }
static size_t dense_core_streamed_bytes(int B, int H, int W) {
    Plan p;
    if (make_plan(B, H, W, DAGL_MODE_ADAPTIVE | DAGL_FLAG_DENSE_HINT, 0, p, true)) return 0;
    size_t off = p.o_end;
    (void)carve(off, dense_workspace_bytes(B, p.g));
    return off;
}

size_t dagl_ce_core_dense_workspace_bytes(int B, int H, int W, int backward) {
    if (B < 1 || H < 1 || W < 1) return 0;
    const size_t gemm_form = dense_train_workspace_bytes(B, make_grid(H, W), backward != 0);
    if (backward || !dense_core_streamed(H, W)) return gemm_form;
    const size_t streamed = dense_core_streamed_bytes(B, H, W);
    return streamed > gemm_form ? streamed : gemm_form;
}

int dagl_ce_core_dense_forward(void* stream, int B, int H, int W, int flags, const float* wq_rows, const float* x_rows,
                               const float* b2, const float* thr, const float* bias, float* out, float* lse, float* mu,
                               void* workspace, size_t ws_bytes, dagl_ce_info* info) {
    DAGL_REQUIRE(B >= 1 && H >= 1 && W >= 1 && wq_rows && x_rows && b2 && thr && bias && out && lse && mu &&
                 (flags & ~DAGL_FLAG_EXACT_SCAN) == 0, "dagl_ce_core_dense_forward: bad argument");
    DAGL_REQUIRE(workspace != nullptr && ((uintptr_t)workspace % 256) == 0, "dagl_ce_core_dense_forward: workspace must be 256-byte aligned");
    const Grid g = make_grid(H, W);
    hipStream_t s = (hipStream_t)stream;
    bool left_range = false;
    if (dense_core_streamed(H, W) && !(flags & DAGL_FLAG_EXACT_SCAN)) {
        CoreIn core{wq_rows, x_rows, nullptr, nullptr, nullptr, nullptr, mu, lse};
        const int rc0 = ce_forward_impl(s, B, H, W, nullptr, b2, thr, bias, nullptr, nullptr, nullptr, nullptr,
                                        DAGL_MODE_ADAPTIVE | DAGL_FLAG_DENSE_HINT, 0, out, workspace, ws_bytes, info, nullptr, nullptr,
                                        nullptr, nullptr, nullptr, 1, &core);
        // a feature outside the split-fp16 range (|feature| >= 937): the streamed kernels NaN-filled `out`; a call that reads
        // its statistics back (info != NULL) notices and is re-run right here in the fp32 GEMM form, which has no such range
        if (rc0 != DAGL_OK || info == nullptr || !info->range_fallback) return rc0;
        left_range = true;
    }
    if (info) { info->required_bytes = (int64_t)dense_train_workspace_bytes(B, g, false); info->total_edges = -1;
                info->max_degree = -1; info->redone_queries = -1; info->path = 5; info->range_fallback = 0; info->dense_rerun_blocks = 0; }
    // the two statistics words live at the very end of the caller's buffer (past the plan)
    const size_t need = dense_train_workspace_bytes(B, g, false) + 256;
    if (ws_bytes < need) { set_error("dagl_ce_core_dense_forward: workspace %zu B < required %zu B", ws_bytes, need);
                           if (info) info->required_bytes = (int64_t)need; return DAGL_ERR_WORKSPACE; }
    int64_t* stats = reinterpret_cast<int64_t*>(static_cast<char*>(workspace) + need - 256);
    int rc = launch_dense_train_forward(s, B, g, wq_rows, x_rows, b2, thr, bias, out, lse, mu, workspace, need - 256,
                                        info ? stats : nullptr);
    if (rc) return rc;
    if (info) {
        int64_t hs[2] = {0, 0};
        if ((rc = read_back(s, stats, 2, hs))) return rc;
        info->total_edges = hs[0]; info->max_degree = (int32_t)hs[1];
        info->range_fallback = left_range ? 1 : 0;
    }
    return DAGL_OK;
}

int dagl_ce_core_dense_backward(void* stream, int B, int H, int W, int flags, const float* wq_rows, const float* x_rows, const float* b2,
                                const float* thr, const float* bias, const float* lse, const float* mu, const float* d_out,
                                float* d_wq_rows, float* d_x_rows, float* d_b2, float* d_thr, float* d_bias, void* workspace,
                                size_t ws_bytes) {
    DAGL_REQUIRE(B >= 1 && H >= 1 && W >= 1 && wq_rows && x_rows && b2 && thr && bias && lse && mu && d_out && d_wq_rows &&
                 d_x_rows && d_b2 && d_thr && d_bias && (flags & ~DAGL_FLAG_EXACT_SCAN) == 0, "dagl_ce_core_dense_backward: bad argument");
    DAGL_REQUIRE(workspace != nullptr && ((uintptr_t)workspace % 256) == 0, "dagl_ce_core_dense_backward: workspace must be 256-byte aligned");
    return launch_dense_train_backward((hipStream_t)stream, B, make_grid(H, W), wq_rows, x_rows, b2, thr, bias, lse, mu, d_out,
                                       d_wq_rows, d_x_rows, d_b2, d_thr, d_bias, workspace, ws_bytes, (flags & DAGL_FLAG_EXACT_SCAN) != 0);
}

// ---- top-k modes whose neighbourhoods exceed the lists (min(k, N) > DAGL_MAX_TOPK) under autograd: the dense formulation of
// dense_train.hip with the row-wise selection of topk_wide.hip as its mask ------------------------------------------------------
int dagl_ce_core_wide_forward(void* stream, int B, int H, int W, int mode, int k, const float* wq_rows, const float* x_rows,
                              const float* b2, const float* thr, const float* bias, float* out, void* workspace, size_t ws_bytes,
                              dagl_ce_info* info) {
    DAGL_REQUIRE(B >= 1 && H >= 1 && W >= 1 && wq_rows && x_rows && b2 && out && k >= 1 &&
                 (mode == DAGL_MODE_TOPK || mode == DAGL_MODE_ADAPTIVE_TOPK), "dagl_ce_core_wide_forward: bad argument");
    DAGL_REQUIRE(mode == DAGL_MODE_TOPK || (thr && bias), "dagl_ce_core_wide_forward: the intersection mode needs thr and bias");
    DAGL_REQUIRE(workspace != nullptr && ((uintptr_t)workspace % 256) == 0, "dagl_ce_core_wide_forward: workspace must be 256-byte aligned");
    const Grid g = make_grid(H, W);
    if (k > g.N) k = g.N;                                              // top_k = min(num_edge, N)
    const size_t need = dense_train_workspace_bytes(B, g, false) + 256;
    if (info) { info->required_bytes = (int64_t)need; info->total_edges = -1; info->max_degree = -1; info->redone_queries = -1;
                info->path = 5; info->range_fallback = 0; info->dense_rerun_blocks = 0; }
    if (ws_bytes < need) { set_error("dagl_ce_core_wide_forward: workspace %zu B < required %zu B", ws_bytes, need); return DAGL_ERR_WORKSPACE; }
    hipStream_t s = (hipStream_t)stream;
    int64_t* stats = reinterpret_cast<int64_t*>(static_cast<char*>(workspace) + need - 256);
    const bool heads = mode != DAGL_MODE_TOPK;
    int rc = launch_dense_train_forward(s, B, g, wq_rows, x_rows, b2, heads ? thr : nullptr, heads ? bias : nullptr, out, nullptr, nullptr,
                                        workspace, need - 256, info ? stats : nullptr, mode, k);
    if (rc) return rc;
    if (info) {
        int64_t hs[2] = {0, 0};
        if ((rc = read_back(s, stats, 2, hs))) return rc;
        info->total_edges = hs[0]; info->max_degree = (int32_t)hs[1];
    }
    return DAGL_OK;
}

int dagl_ce_core_wide_backward(void* stream, int B, int H, int W, int mode, int k, const float* wq_rows, const float* x_rows,
                               const float* b2, const float* thr, const float* bias, const float* d_out, float* d_wq_rows,
                               float* d_x_rows, float* d_b2, float* d_thr, float* d_bias, void* workspace, size_t ws_bytes) {
    DAGL_REQUIRE(B >= 1 && H >= 1 && W >= 1 && wq_rows && x_rows && b2 && d_out && d_wq_rows && d_x_rows && d_b2 && k >= 1 &&
                 (mode == DAGL_MODE_TOPK || mode == DAGL_MODE_ADAPTIVE_TOPK), "dagl_ce_core_wide_backward: bad argument");
    DAGL_REQUIRE(mode == DAGL_MODE_TOPK || (thr && bias && d_thr && d_bias),
                 "dagl_ce_core_wide_backward: the intersection mode needs thr, bias and their gradients");
    DAGL_REQUIRE(workspace != nullptr && ((uintptr_t)workspace % 256) == 0, "dagl_ce_core_wide_backward: workspace must be 256-byte aligned");
    const Grid g = make_grid(H, W);
    if (k > g.N) k = g.N;
    const bool heads = mode != DAGL_MODE_TOPK;
    return launch_dense_train_backward((hipStream_t)stream, B, g, wq_rows, x_rows, b2, heads ? thr : nullptr, heads ? bias : nullptr, nullptr,
                                       nullptr, d_out, d_wq_rows, d_x_rows, d_b2, heads ? d_thr : nullptr, heads ? d_bias : nullptr,
                                       workspace, ws_bytes, true, mode, k);
}

size_t dagl_gemm_f32_scratch_floats(int batch, int M, int N, int K) {
    if (batch != 1 || M < 1 || N < 1 || K < 1) return 0;
    const int sl = gemm32_auto_slices(M, N, K);
    return sl > 1 ? (size_t)sl * M * N : 0;
}

int dagl_gemm_f32(void* stream, int batch, int M, int N, int K, const float* A, long long lda, long long stride_a, int a_k_contiguous,
                  const float* B, long long ldb, long long stride_b, int b_k_contiguous, float* C, long long ldc, long long stride_c,
                  float alpha, float beta, const float* bias, int relu, int chunk_tiles, float* scratch) {
    DAGL_REQUIRE(batch >= 1 && M >= 0 && N >= 0 && K >= 1 && lda >= 1 && ldb >= 1 && ldc >= N && chunk_tiles >= 0, "dagl_gemm_f32: bad shape");
    Gemm32 g;
    g.M = M; g.N = N; g.K = K; g.batch = batch; g.A = A; g.lda = lda; g.sA = stride_a; g.a_kc = a_k_contiguous;
    g.B = B; g.ldb = ldb; g.sB = stride_b; g.b_kc = b_k_contiguous; g.C = C; g.ldc = ldc; g.sC = stride_c;
    g.alpha = alpha; g.beta = beta; g.bias = bias; g.relu = relu; g.chunk_tiles = chunk_tiles;
    if (scratch != nullptr && batch == 1) { g.slices = gemm32_auto_slices(M, N, K); g.scratch = scratch; }
    return launch_gemm32((hipStream_t)stream, g);
}

int dagl_ce_prologue(void* stream, int B, int H, int W, const float* x, const float* g_w, const float* g_b,
                     const float* theta_w, const float* theta_b, const float* thr_w, const float* thr_b,
                     const float* bias_w, const float* bias_b, float* b1_nhwc, float* b2_nhwc, float* thr, float* bias,
                     float* scratch) {
    DAGL_REQUIRE(B >= 1 && H >= 1 && W >= 1 && x && g_w && g_b && theta_w && theta_b && b1_nhwc && b2_nhwc,
                 "dagl_ce_prologue: bad argument");
    if (thr || bias)
        DAGL_REQUIRE(thr && bias && thr_w && thr_b && bias_w && bias_b && scratch,
                     "dagl_ce_prologue: thr/bias heads incomplete (weights, outputs and 8*B*L floats of scratch)");
    hipStream_t s = (hipStream_t)stream;
    const Grid g = make_grid(H, W);
    return launch_prologue(s, B, g, x, g_w, g_b, theta_w, theta_b, thr_w, thr_b, bias_w, bias_b, b1_nhwc, b2_nhwc,
                           thr, bias, nullptr, nullptr, scratch);
}

// (ABI 405) the same four convolutions with g / theta on the fp16 matrix cores (split operands, conv_pair16_kernel: a third of the fp32
// kernel's time) and the key / query map out in fp32 -- the differentiable path's forward.  The input is split with a power-of-two scale
// of each block's own (a second run of the strip when its rows leave |16 x| < 60000): no range on x; |w_conv| < 234 as everywhere.
size_t dagl_ce_prologue16_scratch_bytes(int B, int H, int W) {
    if (B < 1 || H < 1 || W < 1) return 0;
    const Grid g = make_grid(H, W);
    size_t off = 0;
    carve(off, CONV_W16_BYTES);
    carve(off, (size_t)conv16_blocks_per_head(g, 1, B) * sizeof(float));
    carve(off, 8 * (size_t)B * g.L * sizeof(float));
    return off;
}

int dagl_ce_prologue16(void* stream, int B, int H, int W, const float* x, const float* g_w, const float* g_b,
                       const float* theta_w, const float* theta_b, const float* thr_w, const float* thr_b,
                       const float* bias_w, const float* bias_b, float* b1_nhwc, float* b2_nhwc, float* thr, float* bias,
                       void* scratch, size_t scratch_bytes) {
    DAGL_REQUIRE(B >= 1 && H >= 1 && W >= 1 && x && g_w && g_b && theta_w && theta_b && b1_nhwc && b2_nhwc && scratch,
                 "dagl_ce_prologue16: bad argument");
    if (thr || bias)
        DAGL_REQUIRE(thr && bias && thr_w && thr_b && bias_w && bias_b, "dagl_ce_prologue16: thr/bias heads incomplete");
    DAGL_REQUIRE(scratch_bytes >= dagl_ce_prologue16_scratch_bytes(B, H, W) && ((uintptr_t)scratch % 256) == 0,
                 "dagl_ce_prologue16: scratch %zu B (256-byte aligned), need %zu B", scratch_bytes, dagl_ce_prologue16_scratch_bytes(B, H, W));
    hipStream_t s = (hipStream_t)stream;
    const Grid g = make_grid(H, W);
    size_t off = 0;
    unsigned char* convw = at<unsigned char>(scratch, carve(off, CONV_W16_BYTES));
    B1Tiers tiers;                                   // (slots only: the fp32 map has no tiers; they switch the per-block input scale on)
    tiers.slots = conv16_blocks_per_head(g, 1, B);
    tiers.amax = at<float>(scratch, carve(off, (size_t)tiers.slots * sizeof(float)));
    float* thr_part = at<float>(scratch, carve(off, 8 * (size_t)B * g.L * sizeof(float)));
    int rc;
    if ((rc = launch_pack_conv_weight16(s, g_w, theta_w, convw))) return rc;
    return launch_prologue(s, B, g, x, g_w, g_b, theta_w, theta_b, thr_w, thr_b, bias_w, bias_b, b1_nhwc, b2_nhwc,
                           thr, bias, nullptr, nullptr, thr_part, false, false, nullptr, 0, nullptr, 0, RangeTag(), convw, false, &tiers);
}

// ---- the 7x7x16 -> 196 patch Linear (+ReLU) of the differentiable path's FORWARD on the inference kernels: split the map,
// pack the weight, project (split-fp16 matrix cores, no unfolded rows), copy the feature rows out densely ---------------
static void pp16_carve(int B, const Grid& g, int n, size_t& o_hi, size_t& o_lo, size_t& o_wp, size_t& o_feat, size_t& o_range,
                       size_t& total) {
    size_t off = 0;
    const size_t map_h = (size_t)B * g.Hp * g.Wp * CH * sizeof(uint16_t);
    o_hi = carve(off, map_h); o_lo = carve(off, map_h);
    o_wp = carve(off, P16_PACKED_HALFS * sizeof(uint16_t));
    o_feat = carve(off, (size_t)B * feat_rows(n) * DS * sizeof(float));
    o_range = carve(off, 2 * sizeof(int64_t));                        // range word, completion word (RangeTag)
    total = off;
}

size_t dagl_project_patches16_scratch_bytes(int B, int H, int W, int queries) {
    if (B < 1 || H < 1 || W < 1) return 0;
    const Grid g = make_grid(H, W);
    size_t a, b2, c, d, e, total;
    pp16_carve(B, g, queries ? g.L : g.N, a, b2, c, d, e, total);
    return total;
}

int dagl_project_patches16(void* stream, int B, int H, int W, int queries, const float* map_nhwc, const float* w_rows,
                           const float* fc_bias, float* rows_out, void* scratch, size_t scratch_bytes) {
    DAGL_REQUIRE(B >= 1 && H >= 1 && W >= 1 && map_nhwc && w_rows && fc_bias && rows_out && scratch,
                 "dagl_project_patches16: bad argument");
    DAGL_REQUIRE((reinterpret_cast<uintptr_t>(scratch) & 255) == 0, "dagl_project_patches16: scratch must be 256-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    const Grid g = make_grid(H, W);
    const int n = queries ? g.L : g.N;
    size_t o_hi, o_lo, o_wp, o_feat, o_range, total;
    pp16_carve(B, g, n, o_hi, o_lo, o_wp, o_feat, o_range, total);
    DAGL_REQUIRE(scratch_bytes >= total, "dagl_project_patches16: scratch %zu B, need %zu B", scratch_bytes, total);
    uint16_t* hi = at<uint16_t>(scratch, o_hi);
    uint16_t* lo = at<uint16_t>(scratch, o_lo);
    uint16_t* wp = at<uint16_t>(scratch, o_wp);
    float* feat = at<float>(scratch, o_feat);
    int rc;
    // range guard (|16 map| , |1024 w| < 65504): the split / pack / projection kernels store this call's tag into the range
    // word when they meet a larger value, and the copy-out below then writes NaN instead of numbers formed from inf halves
    RangeTag rt;
    int64_t* words = at<int64_t>(scratch, o_range);
    DAGL_HIP_TRY(hipMemsetAsync(words, 0, 2 * sizeof(int64_t), s));
    rt.word = reinterpret_cast<int32_t*>(words); rt.done = reinterpret_cast<int32_t*>(words + 1); rt.tag = next_call_tag();
    if ((rc = launch_split_map(s, (size_t)B * g.Hp * g.Wp * CH, map_nhwc, hi, lo, rt))) return rc;
    if ((rc = launch_pack_fc_weight16(s, w_rows, wp, /*rows_order=*/true))) return rc;
    const float* bias1[1] = {fc_bias};
    if (queries) rc = launch_project16(s, B, g, 2, hi, lo, nullptr, nullptr, nullptr, nullptr, nullptr, wp, bias1, feat, nullptr, nullptr, 1, rt);
    else rc = launch_project16(s, B, g, 1, hi, lo, wp, bias1, feat, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 1, rt);
    if (rc) return rc;
    // [B, feat_rows(n), DS] -> [B, n, 196]
    return launch_feat_rows_out(s, B, n, feat, rows_out, rt);
}

int dagl_pad_nhwc(void* stream, int B, int H, int W, const float* src_nchw, float* dst_nhwc) {
    DAGL_REQUIRE(B >= 1 && H >= 1 && W >= 1 && src_nchw && dst_nhwc, "dagl_pad_nhwc: bad argument");
    return launch_pad_nhwc((hipStream_t)stream, B, H, W, src_nchw, dst_nhwc);
}

int dagl_pack_fc_weight(void* stream, const float* w, float* w_packed) {
    DAGL_REQUIRE(w && w_packed, "dagl_pack_fc_weight: null pointer");
    return launch_pack_fc_weight((hipStream_t)stream, w, w_packed);
}

int dagl_feat_rows(int rows) { return feat_rows(rows); }

int dagl_project_patches(void* stream, int B, int H, int W, int queries, const float* map_nhwc,
                         const float* w_packed, const float* fc_bias, float* feat, double* colsum) {
    DAGL_REQUIRE(B >= 1 && H >= 1 && W >= 1 && map_nhwc && w_packed && fc_bias && feat,
                 "dagl_project_patches: bad argument");
    const Grid g = make_grid(H, W);
    hipStream_t s = (hipStream_t)stream;
    const int rows = queries ? g.L : g.N, ra = feat_rows(rows);
    for (int b = 0; b < B; ++b)
        DAGL_HIP_TRY(hipMemsetAsync(feat + ((size_t)b * ra + rows) * DS, 0, (size_t)(ra - rows) * DS * sizeof(float), s));
    if (colsum) DAGL_HIP_TRY(hipMemsetAsync(colsum, 0, (size_t)B * DS * sizeof(double), s));
    if (queries) return launch_project(s, B, g, 2, map_nhwc, nullptr, nullptr, nullptr, nullptr, w_packed, fc_bias, feat);
    return launch_project(s, B, g, 1, map_nhwc, w_packed, fc_bias, feat, colsum, nullptr, nullptr, nullptr);
}

int dagl_query_thresholds(void* stream, int B, int L, int N, const float* wq, const double* colsum,
                          const float* thr, float* mt) {
    DAGL_REQUIRE(B >= 1 && L >= 1 && N >= 1 && wq && colsum && thr && mt, "dagl_query_thresholds: bad argument");
    return launch_query_thresholds((hipStream_t)stream, B, L, N, wq, colsum, thr, mt);
}

int dagl_gather_aggregate(void* stream, int L, int k, int P_, const int32_t* idx, const float* wgt,
                          const float* values, float* out) {
    DAGL_REQUIRE(L >= 0 && k >= 1 && P_ >= 4 && (P_ % 4) == 0, "dagl_gather_aggregate: bad shape L=%d k=%d P=%d", L, k, P_);
    if (L == 0) return DAGL_OK;
    DAGL_REQUIRE(idx && wgt && values && out, "dagl_gather_aggregate: null pointer");
    DAGL_REQUIRE(((uintptr_t)values % 16) == 0 && ((uintptr_t)out % 16) == 0, "dagl_gather_aggregate: 16-byte alignment required");
    return launch_gather_fixed((hipStream_t)stream, L, k, P_, idx, wgt, values, out);
}

int dagl_unfold_values(void* stream, int B, int H, int W, const float* b2_nhwc, float* rows) {
    DAGL_REQUIRE(B >= 1 && H >= 1 && W >= 1 && b2_nhwc && rows, "dagl_unfold_values: bad argument");
    return launch_unfold_values((hipStream_t)stream, B, make_grid(H, W), b2_nhwc, rows);
}

int dagl_fold_normalize(void* stream, int B, int H, int W, const float* agg, float* out) {
    DAGL_REQUIRE(B >= 1 && H >= 1 && W >= 1 && agg && out, "dagl_fold_normalize: bad argument");
    return launch_fold((hipStream_t)stream, B, make_grid(H, W), agg, out);
}

int dagl_scores_dense(void* stream, int B, int L, int N, const float* wq, const float* x, float* sc) {
    DAGL_REQUIRE(B >= 1 && L >= 1 && N >= 1 && wq && x && sc, "dagl_scores_dense: bad argument");
    return launch_scores_dense((hipStream_t)stream, B, L, N, wq, x, sc);
}

// (ABI 406) any patch geometry: csrc/generic.hip
size_t dagl_ce_generic_workspace_bytes(int B, int Cin, int H, int W, int ksize, int stride_1, int stride_2, int inter_channels) {
    if (B < 1 || Cin < 4 || H < 1 || W < 1 || ksize < 1 || ksize > 31 || stride_1 < 1 || stride_2 < 1 || inter_channels < 4) return 0;
    return dagl::ce_generic_workspace_bytes(B, Cin, H, W, ksize, stride_1, stride_2, inter_channels);
}

int dagl_ce_generic_forward(void* stream, int B, int Cin, int H, int W, int ksize, int stride_1, int stride_2, int inter_channels,
                            float softmax_scale, int mode, int k, const float* x, const float* g_w, const float* g_b,
                            const float* theta_w, const float* theta_b, const float* thr_w, const float* thr_b, const float* bias_w,
                            const float* bias_b, const float* fc1_w, const float* fc1_b, const float* fc2_w, const float* fc2_b,
                            float* out, int32_t* degree, void* workspace, size_t workspace_bytes) {
    int rc = dagl::ce_generic_check(B, Cin, H, W, ksize, stride_1, stride_2, inter_channels, mode, k);
    if (rc) return rc;
    DAGL_REQUIRE(softmax_scale > 0.f, "dagl_ce_generic_forward: softmax_scale must be positive");
    DAGL_REQUIRE(x && g_w && g_b && theta_w && theta_b && fc1_w && fc1_b && fc2_w && fc2_b && out && workspace,
                 "dagl_ce_generic_forward: null pointer");
    if (mode != DAGL_MODE_TOPK) DAGL_REQUIRE(thr_w && thr_b && bias_w && bias_b, "dagl_ce_generic_forward: thr / bias heads missing");
    const size_t need = dagl::ce_generic_workspace_bytes(B, Cin, H, W, ksize, stride_1, stride_2, inter_channels);
    if (workspace_bytes < need) { dagl::set_error("dagl_ce_generic_forward: workspace %zu bytes, %zu needed", workspace_bytes, need); return DAGL_ERR_WORKSPACE; }
    if ((rc = check_device())) return rc;
    return dagl::launch_ce_generic((hipStream_t)stream, B, Cin, H, W, ksize, stride_1, stride_2, inter_channels, softmax_scale, mode, k, x,
                                   g_w, g_b, theta_w, theta_b, thr_w, thr_b, bias_w, bias_b, fc1_w, fc1_b, fc2_w, fc2_b, out, degree, workspace);
}

int dagl_ce_generic_border(int ksize) { return dagl::ce_generic_border(ksize); }

size_t dagl_ce_generic_core_workspace_bytes(int B, int H, int W, int ksize, int stride_1, int stride_2, int inter_channels, int backward) {
    if (B < 1 || H < 1 || W < 1 || ksize < 1 || ksize > 31 || stride_1 < 1 || stride_2 < 1 || inter_channels < 4) return 0;
    return dagl::ce_generic_core_workspace_bytes(B, H, W, ksize, stride_1, stride_2, inter_channels, backward);
}

int dagl_ce_generic_core_forward(void* stream, int B, int H, int W, int ksize, int stride_1, int stride_2, int inter_channels,
                                 float softmax_scale, int mode, int k, const float* wq_rows, const float* x_rows, const float* b2p,
                                 const float* thr, const float* bias, float* out, int32_t* degree, void* workspace, size_t workspace_bytes) {
    int rc = dagl::ce_generic_check(B, 4, H, W, ksize, stride_1, stride_2, inter_channels, mode, k);
    if (rc) return rc;
    DAGL_REQUIRE(softmax_scale > 0.f && wq_rows && x_rows && b2p && out && workspace, "dagl_ce_generic_core_forward: bad argument");
    if (mode != DAGL_MODE_TOPK) DAGL_REQUIRE(thr && bias, "dagl_ce_generic_core_forward: thr / bias missing");
    const size_t need = dagl::ce_generic_core_workspace_bytes(B, H, W, ksize, stride_1, stride_2, inter_channels, 0);
    if (workspace_bytes < need) { dagl::set_error("dagl_ce_generic_core_forward: workspace %zu bytes, %zu needed", workspace_bytes, need); return DAGL_ERR_WORKSPACE; }
    if ((rc = check_device())) return rc;
    return dagl::launch_ce_generic_core_forward((hipStream_t)stream, B, H, W, ksize, stride_1, stride_2, inter_channels, softmax_scale, mode, k,
                                                wq_rows, x_rows, b2p, thr, bias, out, degree, workspace);
}

int dagl_ce_generic_core_backward(void* stream, int B, int H, int W, int ksize, int stride_1, int stride_2, int inter_channels,
                                  float softmax_scale, int mode, int k, const float* wq_rows, const float* x_rows, const float* b2p,
                                  const float* thr, const float* bias, const float* d_out, float* d_wq, float* d_x, float* d_b2p,
                                  float* d_thr, float* d_bias, void* workspace, size_t workspace_bytes) {
    int rc = dagl::ce_generic_check(B, 4, H, W, ksize, stride_1, stride_2, inter_channels, mode, k);
    if (rc) return rc;
    DAGL_REQUIRE(softmax_scale > 0.f && wq_rows && x_rows && b2p && d_out && d_wq && d_x && d_b2p && workspace,
                 "dagl_ce_generic_core_backward: bad argument");
    if (mode != DAGL_MODE_TOPK) DAGL_REQUIRE(thr && bias && d_thr && d_bias, "dagl_ce_generic_core_backward: thr / bias (and their gradients) missing");
    const size_t need = dagl::ce_generic_core_workspace_bytes(B, H, W, ksize, stride_1, stride_2, inter_channels, 1);
    if (workspace_bytes < need) { dagl::set_error("dagl_ce_generic_core_backward: workspace %zu bytes, %zu needed", workspace_bytes, need); return DAGL_ERR_WORKSPACE; }
    if ((rc = check_device())) return rc;
    return dagl::launch_ce_generic_core_backward((hipStream_t)stream, B, H, W, ksize, stride_1, stride_2, inter_channels, softmax_scale, mode, k,
                                                 wq_rows, x_rows, b2p, thr, bias, d_out, d_wq, d_x, d_b2p, d_thr, d_bias, workspace);
}

}  // extern "C"
