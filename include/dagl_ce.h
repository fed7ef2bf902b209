/*
 * dagl_ce.h -- C ABI of the MI355X-native patch-graph attention head.
 *
 * One data-parallel hot path is implemented behind this boundary: the body of
 * the reference method CE.forward (DN_Gray/model/dagl.py:207-275, identical
 * forks in CAR/, Demosaic/, DN_Real/), i.e. everything after the four stock
 * prologue convolutions (dagl.py:208-215) up to and including the batch
 * concatenation (dagl.py:274).
 *
 * The reference has no FFI: its boundary is the Python call self.cX_Y(x) in
 * CES.forward (dagl.py:114,116,118).  The host-side mirror of that call is
 * dagl_amd.CE (an nn.Module with the reference's constructor, parameter names
 * and forward signature); it reaches the device code only through the entry
 * points declared here (ctypes), so that the same library can be bound from
 * any other host language.  INTEGRATION.md shows the binding.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to fp32 / int32 data unless stated;
 *   - buffers are caller-owned; nothing is allocated or freed in here;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream);
 *     all work is enqueued on it; entry points that must read a device value
 *     back (noted below) wait for that copy (an event or a synchronise of that stream), nothing else ever does;
 *   - return value 0 = OK, negative = DAGL_ERR_*; dagl_last_error() gives a
 *     thread-local message; no entry point aborts the process;
 *   - the library holds no global mutable state and is re-entrant.
 *
 * Fixed hyper-parameters (constructor defaults the reference never overrides,
 * dagl.py:175-176, CES passes only in_channels, dagl.py:94-109):
 *   ksize 7, query stride 4, key/value stride 1, inter_channels 16,
 *   softmax_scale 10, patch length P = 16*7*7 = 784, feature length D = 196.
 */
#ifndef DAGL_CE_H
#define DAGL_CE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DAGL_KSIZE      7
#define DAGL_QSTRIDE    4
#define DAGL_CH         16            /* inter_channels                         */
#define DAGL_P          784           /* patch row length  = 16*7*7             */
#define DAGL_D          196           /* similarity feature length = P/4        */
#define DAGL_DS         204           /* row stride (floats) of feature rows    */
#define DAGL_PADPIX     3             /* zero border of the padded NHWC maps    */

/* neighbour-selection modes */
#define DAGL_MODE_ADAPTIVE       0    /* shipped: relu(S - mean*thr + bias) != 0   (dagl.py:256-257) */
#define DAGL_MODE_TOPK           1    /* fixed-k variant (GReccR2b_3mh_1-checkpoint.py:242-250)      */
#define DAGL_MODE_ADAPTIVE_TOPK  2    /* adaptive mask intersected with the k best scores            */

/* OR-ed into `mode`: scan all L*N scores on the fp32 matrix cores instead of screening them in bf16 and
 * refining the survivors (both give the same neighbours)                             */
#define DAGL_FLAG_EXACT_SCAN     0x100
/* OR-ed into `mode`: this workspace last served an identical call -- same memory, same (B,H,W,mode,k), unchanged
 * fc1 / fc2 / g / theta weights -- and nothing else wrote to it since.  The call then reuses what that call left behind
 * (packed fc and convolution weights, the zero borders of the padded maps, the zero guard rows of the feature matrices)
 * instead of rebuilding it: four launches fewer per forward (inference loops)                                */
#define DAGL_FLAG_WEIGHTS_PACKED 0x200
/* OR-ed into `mode` (adaptive mode, >= 2048 keys): expect dense neighbourhoods -- go straight to the streamed dense
 * formulation (info->path 4) instead of trying per-query lists first.  A performance hint only: the result is the
 * same either way; info->total_edges tells the caller whether the hint still pays                              */
#define DAGL_FLAG_DENSE_HINT     0x400

/* OR-ed into `mode` (adaptive mode behind the bf16 screen): do not wait for the call's verdict on the host.  The few queries
 * whose neighbourhood overflows the lists are redone in-stream as always; whether that covered the call (it does not when
 * most queries overflow or the flagged rows are too heavy: the host would have sent the call to the dense formulation) is
 * decided on the device: an unserved call's output is NaN-filled -- never wrong numbers -- and dagl_ce_range_check reports
 * it (bit 1, sticky).  No host synchronisation at all: such a call can be captured into a HIP graph.  info is not filled
 * beyond required_bytes / path.  Callers use it once a workspace's calls have been served in-stream before (dagl_amd.CE).   */
#define DAGL_FLAG_NO_WAIT        0x800

/* OR-ed into `mode` (top-k modes behind the bf16 screen): take the candidate threshold from every SECOND key tile instead of every
 * 8th and give a query's candidate segments eight times the slots (every tile, where a gigabyte of records does not hold that many).  The sampled threshold is as good as the true k-th best on maps
 * whose scores are spread evenly (the synthetic benchmark features); on natural-image features the k-th best of an eighth of the
 * keys lies 4-25 % below the true one, hundreds to thousands of keys pass it, the slots overflow and most query groups land on
 * the fp32 redo pass (2.6 ms instead of 0.25 at 256^2, tools/time_real_image.py).  Costs half a pass more of the screen's
 * matrix work (+25 us at 256^2); the result is the same either way.
 * Round 4: WITHOUT either flag the choice is made ON THE DEVICE and kept in the workspace ("policy" word): a call whose sampled
 * threshold let ANY query overflow its candidate slots (its whole 128-query group takes the fp32 redo pass: 84 us per group at
 * 256^2) switches the workspace to the tight threshold for good (the word is sticky; kernels read it at their start: no host
 * poll, valid under HIP-graph replay).  On a cold workspace (no DAGL_FLAG_WEIGHTS_PACKED: the first call of a shape) a call with
 * more than a fiftieth of its query GROUPS flagged re-runs sampling, filter and refine with the tight threshold in-stream (four
 * gated launches that exit at once otherwise) instead of sending those groups to the fp32 redo pass: ~0.4 ms instead of 2.7 at
 * 256^2.  Maps of up to 16 384 keys start on the tight threshold (it costs nothing there).  A segment whose slots are full spills
 * into its query's shared area (256 records) before the query is flagged: the redo pass is left to maps flat enough for
 * thousands of candidates per query.  DAGL_FLAG_TIGHT_TOPK forces the tight threshold, DAGL_FLAG_SAMPLED_TOPK the
 * sampled one (tests, benchmarks of that path); dagl_ce_range_check's bit 2 still reports whether the last call's redo pass
 * had work.                                                                                                                    */
#define DAGL_FLAG_TIGHT_TOPK     0x1000
#define DAGL_FLAG_SAMPLED_TOPK   0x2000
/* OR-ed into `mode` (top-k modes behind the bf16 screen, dagl_ce_forward_fused): do NOT queue the fp32 redo pass behind the refine
 * kernel.  On a warm workspace that pass is one launch that finds nothing flagged and exits (4.7 us of a 0.23 ms call at 256^2:
 * the launch, not the grid).  Without it a call whose screen DID flag a query group (candidate slots and spill area full: maps flat
 * enough for thousands of candidates per query) cannot be served: its output is NaN-filled by the last kernel -- never wrong
 * numbers -- and dagl_ce_range_check reports it (bit 4, sticky) like a range violation.  For callers that poll: dagl_amd.CE sets it
 * once a poll has found the workspace's recent calls without redo work and drops it for good when a poll reports bit 4.       */
#define DAGL_FLAG_NO_REDO        0x4000

#define DAGL_MAX_TOPK            64   /* widest per-query neighbour LIST of the top-k modes (the fixed-k variant defaults to
                                         num_edge = 50, GReccR2b_3mh_1-checkpoint.py:155,243); k > N = H*W means every key:
                                         top_k = min(k, N) (:243), the lists are then min(k, N) wide.  min(k, N) beyond it
                                         (a stray sibling uses min(500, N), CA_model-checkpoint.py:134-143): the inference entry
                                         points take every query's score row in the dense form instead -- scores of 2048 queries
                                         at a time on the fp32 matrix cores, k-th largest per row by radix selection (ties to the
                                         lower key index, as in the lists), mask / softmax / weighted sum row-wise (info.path 6;
                                         ~5 ms per head at 256^2, k = 500); the training entry points (dagl_ce_core_forward,
                                         lists handed to the backward) and dagl_ces_stage_forward return an error for it      */
#define DAGL_FAST_CAP            64   /* per-query slots of the single-pass adaptive path (fp32 scan, training lists) */
#define DAGL_LIST_CAP           256   /* per-query slots of the screened adaptive path (inference): long-tailed degrees */

/* error codes */
#define DAGL_OK                   0
#define DAGL_ERR_INVALID         -1   /* bad argument (shape, mode, k, null pointer, alignment)      */
#define DAGL_ERR_WORKSPACE       -2   /* workspace too small; info->required_bytes says how much     */
#define DAGL_ERR_HIP             -3   /* a HIP call / launch failed (message has hipGetErrorString)  */
#define DAGL_ERR_NO_DEVICE       -4   /* no gfx950 device visible                                    */
#define DAGL_ERR_UNSUPPORTED     -5   /* training entry point: neighbourhood denser than DAGL_FAST_CAP */

typedef struct dagl_profile dagl_profile;     /* opaque stage profile, see dagl_profile_* below */

typedef struct dagl_ce_info {
    int64_t required_bytes;   /* workspace this call needed (valid on OK and on ERR_WORKSPACE)      */
    int64_t total_edges;      /* sum of degrees over all queries of the batch                       */
    int64_t redone_queries;   /* bf16 screen: queries whose candidate slots / list overflowed and were redone
                                 (-1 = not read back; the debug entry point and the adaptive mode read it);
                                 path 4: queries whose degree exceeds the neighbour lists' width (256)     */
    int32_t max_degree;       /* largest per-query degree                                           */
    int32_t path;             /* 0 = fp32 scan, single-pass lists, 1 = fp32 scan, two-pass CSR (some degree >
                                 FAST_CAP), 2 = fp32 scan, per-lane top-k lists, 3 = bf16 screen + refine,
                                 4 = dense neighbourhoods: streamed dense formulation (no lists),
                                 5 = dense formulation under autograd (dagl_ce_core_dense_forward),
                                 6 = top-k modes with min(k, N) > DAGL_MAX_TOPK: row-wise dense form (no lists) */
    int32_t range_fallback;   /* 1 = an operand left the range of the split-fp16 kernels (|activation| >= 3750, see
                                 below: weights, dense-regime features, |b1| >= 1.5e7, non-finite input) and the call was re-run on the fp32 path */
    int32_t dense_rerun_blocks; /* path 4 / 5: blocks of 64 queries the streamed dense formulation ran a second time (rows whose
                                 exact maximum the top-1 screen could not serve: flat maps); 0 otherwise (ABI 404; was `reserved`) */
} dagl_ce_info;

/* Range of the default (split-fp16) path.
 * Activations (round 6): the fused entry points (dagl_ce_forward_fused, dagl_ces_stage_forward) have NO fixed limit on the input
 * and serve |b1| = |g(x)| < 1.5e7 with the same launches -- the prologue splits x with a power-of-two scale of each block's own
 * and writes the key / query map in two tiers (16 b1 and 2^-8 b1 as fp16 pairs), the projection multiplies the tier that holds the
 * head's values (csrc/dagl_common.h B1Tiers): finite in, finite out, on the first call, without a host round trip, under HIP-graph
 * replay.  (The reference's own fp32 logits 10 S m ~ b1^4 overflow at b1 ~ 2e8.)  dagl_ce_forward (maps computed by the caller)
 * and the training entry points keep the fine tier: |b1| < 3750.  Still fixed: |w_conv| < 234, |w_fc| < 58 (checked when the
 * weights are packed; dagl_amd.CE moves a module with larger weights to DAGL_FLAG_EXACT_SCAN before its first call) and
 * |features| < 937 in the dense regime.  A call that meets an operand beyond these (or a non-finite one) never returns numbers
 * computed from it: its output is NaN-filled by the last kernel, and
 *   - the adaptive modes, which read statistics back anyway, notice and re-run the call on the fp32 path at once
 *     (info->range_fallback = 1; the result is the exact scan's);
 *   - the top-k modes have no host round trip: dagl_ce_range_check(workspace) tells (one synchronisation) whether ANY
 *     call on that workspace left the range since the last check (the word is sticky and cleared by the check that
 *     reports it); re-run with DAGL_FLAG_EXACT_SCAN (no range limit).
 *   - the training entry points (dagl_project_patches16, dagl_ce_core_dense_forward) NaN-fill their outputs likewise; the
 *     dense forward re-runs itself in its fp32 form when it reads statistics back (info != NULL, range_fallback = 1).   */
int dagl_ce_range_check(void* stream, int B, int H, int W, int mode, int k, void* workspace, size_t ws_bytes,
                        int* violated /* bit 0: range, bit 1: an unserved DAGL_FLAG_NO_WAIT call, bit 2 (top-k modes, not sticky):
                                         the LAST call's redo pass had flagged query groups (see DAGL_FLAG_TIGHT_TOPK), bit 3 (top-k
                                         modes): the workspace's threshold policy word has switched to the tight threshold, bit 4 (sticky):
                                         a DAGL_FLAG_NO_REDO call had flagged query groups (its output is NaN-filled) */);

/* ---- library ------------------------------------------------------------------------------- */
/* ABI version of THIS header: bumped whenever a struct or a signature declared here changes (round 3: dagl_ce_info is 40
 * bytes, dagl_ce_prologue takes `scratch`, dagl_ce_core_dense_forward takes `flags`, k <= 64; round 4: DAGL_FLAG_SAMPLED_TOPK,
 * the workspace layout carries the top-k policy words; 403: dagl_ce_core_wide_forward / _backward; 404:
 * dagl_ce_info.dense_rerun_blocks in the place of `reserved`; 405: dagl_fc_grad16_dmap).  A caller compares
 * dagl_version() with the DAGL_ABI_VERSION it was built against and refuses a mismatch (dagl_amd/_lib.py does).           */
#define DAGL_ABI_VERSION 406
int         dagl_version(void);                 /* DAGL_ABI_VERSION of the library = 10000*major + 100*minor + patch */
const char* dagl_last_error(void);              /* thread-local, never NULL                         */
int         dagl_device_check(void);            /* OK iff the current HIP device is gfx950          */

/* Measurement aid (bench.py's roofline line; replaces nothing in the reference): what the matrix pipes SUSTAIN on this device
 * under the screen's multiply stream alone -- `blocks` blocks of 16 waves, `steps` x 26 v_mfma_f32_32x32x16_bf16 per wave,
 * operands in registers.  clocks[2*b] = shader clocks, clocks[2*b+1] = 100 MHz ticks of block b's loop (device memory,
 * 2*blocks uint64); sink = one float of device memory.  FLOP of a launch = blocks * 16 * steps * 26 * 32768; time it with
 * events on `stream`.                                                                                                      */
int dagl_probe_mfma_bf16(void* stream, int blocks, int steps, unsigned long long* clocks, float* sink);
/* Self-test (replaces nothing in the reference): the wave-wide reductions / scan the per-query kernels run on the DPP path
 * (dagl_amd/csrc/dagl_common.h) against a serial evaluation; *mismatches_dev (one int of device memory) = lanes that disagree. */
int dagl_selftest_wave_ops(void* stream, int* mismatches_dev);

/* ---- whole block: replaces dagl.py:216-274 (CE.forward after its prologue convs) -------------- */

/* Bytes of workspace dagl_ce_forward needs for the single-pass / top-k paths (a two-pass CSR
 * fallback may ask for more through DAGL_ERR_WORKSPACE + info->required_bytes).                  */
size_t dagl_ce_workspace_bytes(int B, int H, int W, int mode, int k);

/*
 * b1   [B,16,H,W]  g(b)      key/query feature map   (dagl.py:208)
 * b2   [B,16,H,W]  theta(b)  value feature map       (dagl.py:209)
 * thr  [B,L]       thr_conv(same_pad(b))             (dagl.py:213-214)   L = ceil(H/4)*ceil(W/4)
 * bias [B,L]       bias_conv(same_pad(b))            (dagl.py:215)
 *      (thr/bias may be NULL in DAGL_MODE_TOPK)
 * fc1_w [196,784] fc1_b [196]  query projection  (dagl.py:196-199), element order (c,kh,kw)
 * fc2_w [196,784] fc2_b [196]  key   projection  (dagl.py:200-203)
 * out  [B,16,H,W]  the block's return value          (dagl.py:274)
 * The adaptive mode waits on the host for its verdict (overflowed queries, degrees): a small copy + event queued
 * behind the neighbour refinement, in front of the gather / fold -- those are still running when the call returns;
 * calls that fall back to the dense formulation / CSR lists synchronise `stream` once more for their counters.
 */
int dagl_ce_forward(void* stream, int B, int H, int W,
                    const float* b1, const float* b2, const float* thr, const float* bias,
                    const float* fc1_w, const float* fc1_b, const float* fc2_w, const float* fc2_b,
                    int mode, int k, float* out,
                    void* workspace, size_t ws_bytes, dagl_ce_info* info);

/* The whole of CE.forward (dagl.py:207-275) from the block's input: the four prologue convolutions are
 * computed in here too (prologue.hip), so no stock conv runs on the path.
 *   x [B,64,H,W];  g_w [16,64,3,3] g_b [16];  theta_w [16,64,1,1] theta_b [16];
 *   thr_w / bias_w [1,64,7,7], thr_b / bias_b [1]  (may be NULL in DAGL_MODE_TOPK)
 *   prof: optional stage profile (NULL = none)                                                    */
int dagl_ce_forward_fused(void* stream, int B, int H, int W, const float* x,
                          const float* g_w, const float* g_b, const float* theta_w, const float* theta_b,
                          const float* thr_w, const float* thr_b, const float* bias_w, const float* bias_b,
                          const float* fc1_w, const float* fc1_b, const float* fc2_w, const float* fc2_b,
                          int mode, int k, float* out,
                          void* workspace, size_t ws_bytes, dagl_ce_info* info, dagl_profile* prof);

/* One CES stage (DN_Gray/model/dagl.py:114,116,118): the four heads that share the input x, the 1x1 mixing
 * convolution over their concatenation and the residual,
 *     out = conv1x1(cat(head_1(x), .., head_4(x))) + x,          x, out: [B,64,H,W]
 * as ONE launch set: the heads become a batch dimension (4x fewer launches, 4x more blocks per launch).
 * Dense adaptive neighbourhoods (degree > DAGL_FAST_CAP) are not served here: the call returns
 * DAGL_ERR_WORKSPACE with info->required_bytes = -1 and the caller falls back to four dagl_ce_forward_fused
 * calls + its own mix.  mix_w [64,64,1,1], mix_b [64].                                               */
typedef struct dagl_ce_weights {
    const float *g_w, *g_b, *theta_w, *theta_b, *thr_w, *thr_b, *bias_w, *bias_b, *fc1_w, *fc1_b, *fc2_w, *fc2_b;
} dagl_ce_weights;
size_t dagl_ces_stage_workspace_bytes(int B, int H, int W, int mode, int k);
int dagl_ces_stage_forward(void* stream, int B, int H, int W, const float* x, const dagl_ce_weights* heads4,
                           const float* mix_w, const float* mix_b, int mode, int k, float* out,
                           void* workspace, size_t ws_bytes, dagl_ce_info* info, dagl_profile* prof);

/* Same call with optional per-query read-outs for parity tests (any of them may be NULL):
 *   deg_out [B,L] int32  neighbours per query     (reference: (mask != 0).sum(1), dagl.py:257)
 *   rowsum_out [B,L]     sum_j A_ij               (reference: softmax mass left after masking, :261)
 *   agg_out [B,L,784]    aggregated query patches (reference: torch.mm(yi,pi), :263-264), element
 *                        order (kh,kw,c)                                                           */
int dagl_ce_forward_debug(void* stream, int B, int H, int W,
                          const float* b1, const float* b2, const float* thr, const float* bias,
                          const float* fc1_w, const float* fc1_b, const float* fc2_w, const float* fc2_b,
                          int mode, int k, float* out,
                          void* workspace, size_t ws_bytes, dagl_ce_info* info,
                          int32_t* deg_out, float* rowsum_out, float* agg_out);

/* ---- stage profile: hipEvents recorded at the stage boundaries on the caller's stream ---------
 * The benchmark times its steps with these (no synchronisation is added to the timed region;
 * dagl_profile_read waits for the recorded events afterwards).  Stage order:                     */
#define DAGL_N_STAGES        8
#define DAGL_STAGE_LAYOUT    0   /* pad/NHWC maps + fc weight pack                                  */
#define DAGL_STAGE_PROJECT   1   /* fc2 projection of the N key patches (+ column sums) and fc1
                                    projection of the L query patches: one launch                  */
#define DAGL_STAGE_THRESH    2   /* per-query adaptive thresholds                                   */
#define DAGL_STAGE_SAMPLE    3   /* top-k screen: sampled group maxima -> per-query threshold       */
#define DAGL_STAGE_SELECT    4   /* the full L*N scan: bf16 screen filter, or the fp32 select       */
#define DAGL_STAGE_EDGE      5   /* refine / candidate merge / degree scan + edge softmax           */
#define DAGL_STAGE_GATHER    6   /* neighbour gather + weighted sum                                 */
#define DAGL_STAGE_FOLD      7   /* fold + overlap normalisation                                    */
int dagl_profile_create(int max_calls, dagl_profile** out);
int dagl_profile_destroy(dagl_profile* prof);
int dagl_profile_reset(dagl_profile* prof);
/* stage >= 0: record only the two events around that stage (the others read back as 0 ms); -1 = all boundaries.
 * Also resets the recorded calls. */
int dagl_profile_select_stage(dagl_profile* prof, int stage);
/* stage_ms [capacity_calls][DAGL_N_STAGES] (host memory, may be NULL to query n_calls only) */
int dagl_profile_read(dagl_profile* prof, int* n_calls, float* stage_ms, int capacity_calls);
int dagl_ce_forward_profiled(void* stream, int B, int H, int W,
                             const float* b1, const float* b2, const float* thr, const float* bias,
                             const float* fc1_w, const float* fc1_b, const float* fc2_w, const float* fc2_b,
                             int mode, int k, float* out,
                             void* workspace, size_t ws_bytes, dagl_ce_info* info, dagl_profile* prof);

/* ---- training: graph core with the features given, and its backward ---------------------------
 * Replaces what autograd records for dagl.py:250-272 (score matrix, adaptive mask, non-renormalised softmax,
 * aggregation, fold) when the block is trained (DN_Gray/trainer.py:44-50 loss.backward()).  The host computes
 * the projections itself (under its own autograd: wq_rows = relu(fc1(unfold4(b1))) [B,L,196],
 * x_rows = relu(fc2(unfold1(b1))) [B,N,196], dagl.py:240-249) and gets back the fixed-width neighbour lists
 * (width = dagl_ce_list_width(mode,k)) the backward needs.  Sparse neighbourhoods only: an adaptive mask that
 * keeps more than DAGL_FAST_CAP keys for some query returns DAGL_ERR_UNSUPPORTED.
 * Workspace of the forward: dagl_ce_workspace_bytes(B,H,W,mode,k).                                          */
int    dagl_ce_list_width(int mode, int k);
int    dagl_ce_core_forward(void* stream, int B, int H, int W,
                            const float* wq_rows, const float* x_rows, const float* b2 /* [B,16,H,W] */,
                            const float* thr, const float* bias /* [B,L], NULL in top-k mode */,
                            int mode, int k, float* out /* [B,16,H,W] */,
                            int32_t* nb_idx, float* nb_wgt, float* nb_s /* [B,L,width] each */,
                            int32_t* nb_cnt /* [B,L] */, float* mu /* [B,L] row means, adaptive modes */,
                            void* workspace, size_t ws_bytes, dagl_ce_info* info);
size_t dagl_ce_core_backward_workspace_bytes(int B, int H, int W, int mode, int k);
/* Gradients of a scalar loss w.r.t. wq_rows, x_rows, b2, thr, bias given d_out = dL/d out.  d_x_rows and d_b2
 * are gathered over the edge list sorted by key (stable radix sort: bit-reproducible); d_thr / d_bias may be NULL
 * in top-k mode.  The selection itself (which keys are neighbours) carries no gradient, as in the reference.  */
int    dagl_ce_core_backward(void* stream, int B, int H, int W, int mode, int k,
                             const float* wq_rows, const float* x_rows, const float* b2,
                             const float* thr, const float* bias,
                             const int32_t* nb_idx, const float* nb_wgt, const float* nb_s, const int32_t* nb_cnt,
                             const float* mu, const float* d_out,
                             float* d_wq_rows, float* d_x_rows, float* d_b2, float* d_thr, float* d_bias,
                             void* workspace, size_t ws_bytes);

/* Dense neighbourhoods under autograd (an adaptive mask that keeps more than DAGL_FAST_CAP keys for some query -- the
 * regime of default-initialised thr/bias heads, i.e. where a training run starts): the reference's dense formulation
 * (dagl.py:250-264) and the gradients autograd derives from it, chunked over the queries so that the [L,N] matrices
 * exist one chunk at a time; matrix products on the fp32 matrix cores (bitwise fmaf chains) or, by default in the backward, on
 * the fp16 ones with split operands.  Adaptive mode only
 * (the top-k modes always have fixed-width lists).  Same operands as dagl_ce_core_forward / _backward; instead of
 * neighbour lists the forward hands back `lse` [B,L,2] = (softmax shift, denominator) per query and `mu` [B,L].
 * info (may be NULL; non-NULL costs one host synchronisation): path 5, total_edges, max_degree.
 * flags: DAGL_FLAG_EXACT_SCAN = the chunked fp32 GEMM form whatever the size (no split-fp16 range limit); 0 = the
 * streamed split-fp16 kernel for images of >= 2048 keys (|feature| < 937; beyond it `out` is NaN-filled, and a call
 * with info != NULL re-runs itself in the GEMM form and sets info->range_fallback).                              */
size_t dagl_ce_core_dense_workspace_bytes(int B, int H, int W, int backward);
int    dagl_ce_core_dense_forward(void* stream, int B, int H, int W, int flags /* 0 or DAGL_FLAG_EXACT_SCAN */,
                                  const float* wq_rows, const float* x_rows, const float* b2,
                                  const float* thr, const float* bias, float* out, float* lse, float* mu,
                                  void* workspace, size_t ws_bytes, dagl_ce_info* info);
/* backward flags: 0 = the five matrix products of the gradients on the fp16 matrix cores with split operands (every tensor scaled
 * by a power of two from its own largest magnitude: no range limit; shapes whose [L,N] matrices fit one chunk, else fp32);
 * DAGL_FLAG_EXACT_SCAN = on the fp32 matrix cores.                                                                         */
int    dagl_ce_core_dense_backward(void* stream, int B, int H, int W, int flags,
                                   const float* wq_rows, const float* x_rows, const float* b2,
                                   const float* thr, const float* bias, const float* lse, const float* mu,
                                   const float* d_out,
                                   float* d_wq_rows, float* d_x_rows, float* d_b2, float* d_thr, float* d_bias,
                                   void* workspace, size_t ws_bytes);

/* The top-k modes with min(k, N) > DAGL_MAX_TOPK under autograd (the fixed-k variant takes any num_edge: top_k = min(num_edge, N),
 * GReccR2b_3mh_1-checkpoint.py:242-250; CA_model-checkpoint.py:134-143 uses 500): no lists -- the dense formulation above with
 * the row-wise selection as its mask: per query the k best scores (ties at the k-th place to the lower key index, as everywhere),
 * DAGL_MODE_TOPK: logits 10 S on them, a 0/1 mask (thr / bias and their gradients may be NULL); DAGL_MODE_ADAPTIVE_TOPK: the k best of
 * the keys that pass the adaptive test, m = relu(S - mean thr + bias).  Matrix products on the fp32 matrix cores (the backward
 * re-selects from the recomputed scores, which are the forward's bit for bit).  Workspace: dagl_ce_core_dense_workspace_bytes.
 * info (may be NULL; non-NULL costs one host synchronisation): path 5, total_edges, max_degree.                                    */
int    dagl_ce_core_wide_forward(void* stream, int B, int H, int W, int mode, int k,
                                 const float* wq_rows, const float* x_rows, const float* b2,
                                 const float* thr, const float* bias, float* out,
                                 void* workspace, size_t ws_bytes, dagl_ce_info* info);
int    dagl_ce_core_wide_backward(void* stream, int B, int H, int W, int mode, int k,
                                  const float* wq_rows, const float* x_rows, const float* b2,
                                  const float* thr, const float* bias, const float* d_out,
                                  float* d_wq_rows, float* d_x_rows, float* d_b2, float* d_thr, float* d_bias,
                                  void* workspace, size_t ws_bytes);

/* ---- stages (each callable on its own: unit parity tests and the benchmark use them) --------- */

/* Batched fp32 matrix product on the matrix cores, the building block of the dense training stages:
 * C[b] = alpha * A[b] B[b] + beta * C[b] (+ bias[n], relu).  A is logically M x K, stored row-major
 * (a_k_contiguous: element (m,k) at A[m*lda + k]) or K-major (A[k*lda + m]); B is logically K x N, stored
 * as N x K rows (b_k_contiguous: B[n*ldb + k]) or K x N rows (B[k*ldb + n]).  Replaces torch.matmul / torch.mm
 * of dagl.py:250,263 and their autograd counterparts.  chunk_tiles > 0: partial sums over chunk_tiles * 16 products
 * added in fp32 (shorter rounding chains).  scratch (batch == 1 only, may be NULL): dagl_gemm_f32_scratch_floats()
 * floats; when given, a product with few output tiles and a long K (a weight gradient) is cut into K slices that fill
 * the chip, summed in slice order.  Deterministic either way (no atomics).                                        */
size_t dagl_gemm_f32_scratch_floats(int batch, int M, int N, int K);
int dagl_gemm_f32(void* stream, int batch, int M, int N, int K,
                  const float* A, long long lda, long long stride_a, int a_k_contiguous,
                  const float* B, long long ldb, long long stride_b, int b_k_contiguous,
                  float* C, long long ldc, long long stride_c, float alpha, float beta, const float* bias, int relu,
                  int chunk_tiles, float* scratch);

/* Stages of the differentiable path besides the matrix products (train_ops.hip): a convolution / Linear over patches
 * under autograd is  unfold -> dagl_gemm_f32 (+ bias, ReLU);  d weight = d Z^T rows,  d rows = d Z weight,
 * d map = fold(d rows).  Replaces nn.Conv2d / nn.Linear / nn.Unfold of dagl.py:208-249 in the training path.
 *   unfold: rows[b, (py,px), (kh,kw,c)] = map[b, oy + py*stride + kh, ox + px*stride + kw, c]   (map [B,Hp,Wp,C], C % 4 == 0)
 *   fold:   the adjoint, written for every pixel of the [B,Hp,Wp,C] map (gather form, no atomics)
 *   copy4:  generic strided 4-D copy (layout changes NCHW <-> zero-bordered NHWC and their adjoints)             */
int dagl_unfold_patches(void* stream, int B, int Hp, int Wp, int C, int k, int stride, int oy, int ox, int oh, int ow,
                        const float* map, float* rows);
int dagl_fold_patches(void* stream, int B, int Hp, int Wp, int C, int k, int stride, int oy, int ox, int oh, int ow,
                      const float* d_rows, float* d_map);
int dagl_copy4(void* stream, int n0, int n1, int n2, int n3, const float* src, long long s0, long long s1, long long s2,
               long long s3, float* dst, long long d0, long long d1, long long d2, long long d3);
int dagl_relu_backward(void* stream, size_t n, const float* y, const float* dy, float* dz);   /* dz = dy * (y > 0)   */
/* single-parameter PReLU of the trunk's ResBlocks (DN_Gray/model/common.py:59-79, act = nn.PReLU(); DN_Gray/model/dagl.py:27-35,
 * 76-90): y = x > 0 ? x : a x; backward dx = dy (x > 0 ? 1 : a), da = sum dy x [x <= 0] with a fixed-order fp64 reduction
 * (deterministic).  n = element count, a multiple of 4; tensors contiguous and 16-byte aligned.                            */
int    dagl_prelu_forward(void* stream, size_t n, const float* x, const float* a, float* y);
size_t dagl_prelu_scratch_bytes(size_t n);
int    dagl_prelu_backward(void* stream, size_t n, const float* x, const float* dy, const float* a, float* dx, float* da, void* scratch);
size_t dagl_col_sum_scratch_bytes(size_t rows, int cols);
int dagl_col_sum(void* stream, size_t rows, int cols, const float* src, float* out, void* scratch);   /* out[c] = sum_r src[r,c], cols <= 256 */

/* The two gradient products of a patch projection (fc1 / fc2 under autograd, dagl.py:248-249) on the fp16 matrix cores with
 * split operands (gemm16s.hip) -- what dagl_unfold_patches + two dagl_gemm_f32 calls compute, at several times the rate:
 *   d_w    [196,784]   = dz^T rows            (rows = the 7x7x16 patches of map_nhwc on the grid (stride, oy, ox, oh, ow),
 *                                              element order (kh,kw,c); they are never materialised in fp32)
 *   d_rows [B*oh*ow,784] = dz w_rows          (the caller folds them back: dagl_fold_patches)
 *   d_b    [196]       = column sums of dz
 * dz = dy (y > 0) when the layer's output y is given (ReLU backward, fused with the pass that takes dz's largest magnitude
 * and its column sums), dz = dy when y is NULL; y, dy [B*oh*ow,196]; w_rows [196,784] in (kh,kw,c) order; any output may be NULL.
 * Holds for |16 map|, |1024 w| < 65504 (the forward projection's own range); dz is rescaled per call by a power of two taken
 * from its largest magnitude.  Fixed summation order (split-K slices added in slice order): bit-reproducible.             */
size_t dagl_fc_grad16_scratch_bytes(int B, int oh, int ow);
int dagl_fc_grad16(void* stream, int B, int Hp, int Wp, int stride, int oy, int ox, int oh, int ow, const float* map_nhwc,
                   const float* w_rows, const float* y, const float* dy, float* d_w, float* d_b, float* d_rows, void* scratch,
                   size_t scratch_bytes);
/* (ABI 405) The same with the gradient of the MAP as its output: d_map [B,Hp,Wp,16] = fold(d_rows) as dagl_fold_patches computes it
 * (autograd through `fc(unfold(map))`, dagl.py:240-249), the rows folded inside the product -- over kw in the block's LDS, over kh by a
 * light second kernel, every sum in a fixed order -- so that the [n, 784] fp32 rows (411 MB at n = 131 072) are neither written nor read
 * back.  Geometries: dagl_fc_grad16_dmap_ok(stride, ow) (stride 1 and 16 / 32 / 64 or a multiple of 128 patches per row); others:
 * dagl_fc_grad16 + dagl_fold_patches.                                                                                          */
int    dagl_fc_grad16_dmap_ok(int stride, int ow);
size_t dagl_fc_grad16_dmap_scratch_bytes(int B, int oh, int ow);
int dagl_fc_grad16_dmap(void* stream, int B, int Hp, int Wp, int stride, int oy, int ox, int oh, int ow, const float* map_nhwc,
                        const float* w_rows, const float* y, const float* dy, float* d_w, float* d_b, float* d_map, void* scratch,
                        size_t scratch_bytes);

/* Backward of the first two convolutions of the block (g 3x3 and theta 1x1, 64 -> 16: dagl.py:208-209 under loss.backward(),
 * DN_Gray/trainer.py:48-57) straight on the maps -- no patch rows (conv_grad.hip): tap-wise products on the fp32 matrix cores,
 * fixed summation order (bit-reproducible).
 *   x [B,64,H,W] (the forward's input), d_b1p / d_b2p [B,H+6,W+6,16] = gradients of the zero-bordered NHWC maps (interior read),
 *   g_w [16,64,3,3], th_w [16,64] (needed for d_x only)
 *   d_x [B,64,H,W] or NULL;  d_g_w [16,64,3,3], d_g_b [16], d_th_w [16,64], d_th_b [16]: all four or all NULL
 *   scratch: dagl_conv_pair_backward_scratch_bytes(B,H,W) bytes (parameter gradients only)
 * dagl_conv_pair_backward_supported: W a multiple of 4 and <= 256 (rows are staged whole); other shapes: unfold + dagl_gemm_f32. */
int    dagl_conv_pair_backward_supported(int B, int H, int W);
size_t dagl_conv_pair_backward_scratch_bytes(int B, int H, int W);
int    dagl_conv_pair_backward(void* stream, int B, int H, int W, const float* x, const float* d_b1p, const float* d_b2p,
                               const float* g_w, const float* th_w, float* d_x, float* d_g_w, float* d_g_b, float* d_th_w,
                               float* d_th_b, void* scratch);

/* The four prologue convolutions alone (dagl.py:208-215): b1/b2 as zero-bordered NHWC maps
 * [B,H+6,W+6,16], thr/bias [B,L] (both NULL = skip the two 7x7 heads).  `scratch` = 8*B*L floats of
 * caller-owned device memory for the heads' partial sums (may be NULL without the heads): like every
 * entry point, nothing is allocated in here.                                                       */
int dagl_ce_prologue(void* stream, int B, int H, int W, const float* x,
                     const float* g_w, const float* g_b, const float* theta_w, const float* theta_b,
                     const float* thr_w, const float* thr_b, const float* bias_w, const float* bias_b,
                     float* b1_nhwc, float* b2_nhwc, float* thr, float* bias, float* scratch);

/* (ABI 405) The same with g / theta on the fp16 matrix cores (split operands: >= 21 significant bits, a third of the fp32 kernel's time):
 * the differentiable path's forward.  No range on x (per-block power-of-two input scale); |w_conv| < 234.  `scratch`: 256-byte aligned,
 * dagl_ce_prologue16_scratch_bytes(B, H, W) bytes (packed weights, per-block statistics, the heads' partial sums).                  */
size_t dagl_ce_prologue16_scratch_bytes(int B, int H, int W);
int dagl_ce_prologue16(void* stream, int B, int H, int W, const float* x,
                       const float* g_w, const float* g_b, const float* theta_w, const float* theta_b,
                       const float* thr_w, const float* thr_b, const float* bias_w, const float* bias_b,
                       float* b1_nhwc, float* b2_nhwc, float* thr, float* bias, void* scratch, size_t scratch_bytes);

/* NCHW [B,16,H,W] -> zero-bordered NHWC [B,H+6,W+6,16]  (patch unfold without materialising
 * patches: replaces same_padding/extract_image_patches, dagl.py:123-169, for all three uses).    */
int dagl_pad_nhwc(void* stream, int B, int H, int W, const float* src_nchw, float* dst_nhwc);

/* fc weight [196,784] in the reference's (c,kh,kw) column order -> 208*784 floats in the packed
 * layout the projection kernel streams: [49 (kh,kw)][208 outputs, rows 196.. zero][16 channels,
 * 16-byte quads XOR-swizzled for conflict-free LDS reads].                                       */
int dagl_pack_fc_weight(void* stream, const float* w, float* w_packed);

/* Patch projection = Linear(784->196)+ReLU applied to every 7x7x16 patch of the padded map
 * (dagl.py:248 for queries, :249 for keys), as an implicit GEMM.
 *   queries != 0 : stride-4 SAME grid (L rows);  0 : stride-1 grid (N = H*W rows)
 *   feat   [B, rows_alloc, 204]  rows_alloc = dagl_feat_rows(rows); columns 196..203 and the
 *          rows beyond `rows` are written as zero
 *   colsum [B,204] (keys only, may be NULL): sum over the N rows of each column (for the row
 *          mean of the score matrix, dagl.py:256)                                                */
int dagl_project_patches(void* stream, int B, int H, int W, int queries,
                         const float* map_nhwc, const float* w_packed, const float* fc_bias,
                         float* feat, double* colsum);
int dagl_feat_rows(int rows);          /* rows rounded up to the streaming tile + 1 guard tile     */

/* The same Linear(784->196)+ReLU over patches on the split-fp16 matrix cores (the inference kernels: the map is split into
 * fp16 hi / lo, the weight packed, no unfolded patch rows exist), for the FORWARD of the differentiable path:
 *   map_nhwc  zero-bordered [B,H+6,W+6,16] fp32;  w_rows [196, 49 taps, 16 c] (the patch order of dagl_unfold_patches);
 *   rows_out  [B, n, 196] dense (n = L for queries != 0, N = H*W otherwise);  scratch: 256-byte aligned device memory of
 *   dagl_project_patches16_scratch_bytes(B,H,W,queries) bytes.  Holds for |16 map| < 65504, |1024 w| < 65504.  */
size_t dagl_project_patches16_scratch_bytes(int B, int H, int W, int queries);
int dagl_project_patches16(void* stream, int B, int H, int W, int queries, const float* map_nhwc, const float* w_rows,
                           const float* fc_bias, float* rows_out, void* scratch, size_t scratch_bytes);

/* Per-query threshold pieces of the adaptive mask (dagl.py:256):
 *   mt[b,l] = mean_j S[l,j] * thr[b,l]   with mean_j S[l,j] = Wq[l,:] . (colsum/N)
 *   (bias is used as is).                                                                        */
int dagl_query_thresholds(void* stream, int B, int L, int N, const float* wq, const double* colsum,
                          const float* thr, float* mt);

/* Gather + weighted sum over fixed-width neighbour lists: the sparse form of torch.mm(yi, pi)
 * (dagl.py:263-264; the north-star's batched_index_select + weighted sum).
 *   idx [L,k] int32 (row index into values, <0 = empty slot), wgt [L,k], values [n_rows,P],
 *   out [L,P];  P must be a multiple of 4, pointers 16-byte aligned.
 * Algorithmic bytes per query: k*P*4 (rows) + k*8 (idx,wgt) + P*4 (out).                        */
int dagl_gather_aggregate(void* stream, int L, int k, int P,
                          const int32_t* idx, const float* wgt, const float* values, float* out);

/* Materialise value rows [B,N,784] (element order (kh,kw,c)) from the padded NHWC value map --
 * what Unfold does at dagl.py:224-231; only the stand-alone gather benchmark / tests need it.   */
int dagl_unfold_values(void* stream, int B, int H, int W, const float* b2_nhwc, float* rows);

/* Fold the aggregated query patches back and divide by the overlap count (dagl.py:265-272).
 *   agg [B,L,784] element order (kh,kw,c);  out [B,16,H,W]                                       */
int dagl_fold_normalize(void* stream, int B, int H, int W, const float* agg, float* out);

/* Dense score matrix S = Wq . X^T for tests (dagl.py:250): s [B,L,N].                            */
int dagl_scores_dense(void* stream, int B, int L, int N, const float* wq, const float* x, float* s);

/* (ABI 406) CE.forward for ANY patch geometry.  ksize, stride_1, stride_2 and inter_channels are constructor arguments of the
 * reference block (DN_Gray/model/dagl.py:175-176; its own builders, dagl.py:94-109, never override them, so every tuned kernel has
 * (7, 4, 1, 16) compiled in).  This entry runs the reference's dense formulation, dagl.py:207-275, with the geometry as run-time
 * arguments -- unfold + fp32 matrix-core products, one chunk of score rows (<= 256 MiB) at a time, row-wise mask / softmax, fold
 * with the reference's padding rule (dagl.py:243: the stride_2 SAME grid's left pad) -- csrc/generic.hip.  Replaces the Python
 * call self.cX_Y(x) of a CE built with non-default arguments (dagl.py:114-118).
 *   x [B,Cin,H,W] fp32 NCHW;  g_w [c,Cin,3,3], theta_w [c,Cin,1,1], thr_w / bias_w [1,Cin,ksize,ksize] (may be NULL in
 *   DAGL_MODE_TOPK), fc1_w / fc2_w [P/4, P] over Unfold's (c,kh,kw) patch order with P = ksize^2 c -- the state_dict's own
 *   layouts, nothing pre-packed;  out [B,c,H,W];  degree [B*L] int32 or NULL (neighbours per query);  Cin and c multiples of 4.
 *   softmax_scale: dagl.py:175 (10 by default), any positive value.  k: the fixed-k modes' num_edge (min(k, N) keys are kept).
 * A geometry whose F.fold block grid does not hold exactly the L query patches raises in the reference; here DAGL_ERR_INVALID.  */
size_t dagl_ce_generic_workspace_bytes(int B, int Cin, int H, int W, int ksize, int stride_1, int stride_2, int inter_channels);
int dagl_ce_generic_forward(void* stream, int B, int Cin, int H, int W, int ksize, int stride_1, int stride_2, int inter_channels,
                            float softmax_scale, int mode, int k, const float* x,
                            const float* g_w, const float* g_b, const float* theta_w, const float* theta_b,
                            const float* thr_w, const float* thr_b, const float* bias_w, const float* bias_b,
                            const float* fc1_w, const float* fc1_b, const float* fc2_w, const float* fc2_b,
                            float* out, int32_t* degree, void* workspace, size_t workspace_bytes);

/* (ABI 406) The differentiable graph core, dagl.py:250-272, for any patch geometry (autograd through a CE built with non-default
 * arguments; the convolutions and patch projections around it are dagl_unfold_patches + dagl_gemm_f32 with their adjoints):
 *   wq_rows [B,L,P/4], x_rows [B,N,P/4] the feature rows (relu(fc(patches)), dagl.py:248-249);  b2p the value map, zero-bordered NHWC
 *   [B, H+2*border, W+2*border, c] with border = dagl_ce_generic_border(ksize);  thr / bias [B,L] (NULL in DAGL_MODE_TOPK);
 *   out / d_out [B,c,H,W].  The backward recomputes S and A chunk by chunk; it writes d_wq, d_x, d_b2p (every element) and, in the
 *   adaptive modes, d_thr / d_bias.  Workspace: dagl_ce_generic_core_workspace_bytes(..., backward).                                 */
int    dagl_ce_generic_border(int ksize);
size_t dagl_ce_generic_core_workspace_bytes(int B, int H, int W, int ksize, int stride_1, int stride_2, int inter_channels, int backward);
int dagl_ce_generic_core_forward(void* stream, int B, int H, int W, int ksize, int stride_1, int stride_2, int inter_channels,
                                 float softmax_scale, int mode, int k, const float* wq_rows, const float* x_rows, const float* b2p,
                                 const float* thr, const float* bias, float* out, int32_t* degree, void* workspace, size_t workspace_bytes);
int dagl_ce_generic_core_backward(void* stream, int B, int H, int W, int ksize, int stride_1, int stride_2, int inter_channels,
                                  float softmax_scale, int mode, int k, const float* wq_rows, const float* x_rows, const float* b2p,
                                  const float* thr, const float* bias, const float* d_out, float* d_wq, float* d_x, float* d_b2p,
                                  float* d_thr, float* d_bias, void* workspace, size_t workspace_bytes);

#ifdef __cplusplus
}
#endif
#endif /* DAGL_CE_H */
