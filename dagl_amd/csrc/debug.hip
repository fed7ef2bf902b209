// Ablation builds only (-DDAGL_ABLATION): block phase stamps of the matrix-core kernels.  A kernel given a stamp buffer
// writes four 100 MHz time stamps per block (entry, loop start, loop end, exit); with DAGL_TIMES_FILE set the launcher
// synchronises and appends them to that file, one line per launch: "<kernel> <blocks> t0 t1 t2 t3 t0 t1 ...".
// tools/block_times.py turns the lines into a timeline (dispatch ramp, prologue, loop, epilogue, tail).
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include "dagl_common.h"

namespace dagl {

#ifdef DAGL_ABLATION
unsigned long long* dbg_times_buffer(size_t blocks) {
    static unsigned long long* buf = nullptr;
    static size_t cap = 0;
    if (blocks > cap) {
        if (buf) (void)hipFree(buf);
        if (hipMalloc(&buf, blocks * 4 * sizeof(unsigned long long)) != hipSuccess) { buf = nullptr; cap = 0; return nullptr; }
        cap = blocks;
    }
    return buf;
}

void dbg_times_dump(hipStream_t s, const char* kernel, const unsigned long long* buf, size_t blocks) {
    const char* path = getenv("DAGL_TIMES_FILE");
    if (!path || !buf) return;
    static int skip = getenv("DAGL_TIMES_SKIP") ? atoi(getenv("DAGL_TIMES_SKIP")) : 0;   // dump calls passed over first (warm clocks)
    static int budget = 48;                                   // dump calls recorded per process
    if (skip > 0) { --skip; return; }
    if (budget-- <= 0) return;
    if (hipStreamSynchronize(s) != hipSuccess) return;
    std::vector<unsigned long long> h(blocks * 4);
    if (hipMemcpy(h.data(), buf, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) return;
    FILE* f = fopen(path, "a");
    if (!f) return;
    fprintf(f, "%s %zu", kernel, blocks);
    for (unsigned long long v : h) fprintf(f, " %llu", v);
    fprintf(f, "\n");
    fclose(f);
}
#endif

// ---- what the matrix pipes sustain: the screen's multiply loop and nothing else ------------------------------------------------
// The nominal bf16 peak (2.5 PFLOP/s) is 1024 SIMDs x one v_mfma_f32_32x32x16_bf16 per 32 cycles x 2.4 GHz.  Under exactly this
// instruction stream the part does not hold 2.4 GHz (profiles/r03_screen_ring_clock_and_factors.log: 1.85-1.9 GHz in
// screen_ring_kernel's loop), so bench.py prices the screen against BOTH: the nominal peak, and the rate this probe reaches on the
// same box -- one block per CU, 16 waves, two accumulator chains per wave, 26 multiplies per step like the screen, operands in
// registers, no LDS, no memory.  Per block: shader clocks (s_memtime) and 100 MHz ticks (s_memrealtime) of wave 0's loop.
typedef __bf16 probe_bf16x8 __attribute__((ext_vector_type(8)));
typedef short probe_s16x8 __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(1024, 1) void mfma_probe_kernel(int steps, unsigned long long* __restrict__ clocks, float* __restrict__ sink) {
    const int lane = threadIdx.x & 63;
    probe_bf16x8 q[13], k0, k1;
    {
        probe_s16x8 v;
#pragma unroll
        for (int t = 0; t < 13; ++t) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (short)(0x3C00 + ((lane * 37 + t * 11 + e * 5) & 0x1FF));       // bf16 in [2^-7, 2^-5)
            q[t] = __builtin_bit_cast(probe_bf16x8, v);
        }
        k0 = q[3]; k1 = q[7];
    }
    f32x16 a0, a1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { a0[r] = 0.f; a1[r] = 0.f; }
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < steps; ++it) {
#pragma unroll
        for (int t = 0; t < 13; ++t) {
            a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k0, q[t], a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k1, q[t], a1, 0, 0, 0);
        }
        // keep the sums bounded (and the loop from being folded): two instructions per 26 multiplies
        a0[0] *= 0.5f; a1[0] *= 0.5f;
    }
    float t = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) t += a0[r] + a1[r];
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (t == 12345.678f) sink[0] = t;                        // (never true in practice; keeps the accumulators alive)
    if (threadIdx.x == 0) { clocks[2 * blockIdx.x] = c1 - c0; clocks[2 * blockIdx.x + 1] = r1 - r0; }
}

// ---- self-test of the DPP wave primitives (dagl_common.h) against a serial evaluation through the LDS --------------------------
__global__ __launch_bounds__(256) void wave_ops_selftest_kernel(unsigned seed, int* __restrict__ mismatches) {
    __shared__ float sf[256];
    __shared__ double sd[256];
    __shared__ int si[256];
    const int tid = threadIdx.x, lane = tid & 63, w0 = tid & ~63;
    unsigned x = seed * 2654435761u + (blockIdx.x * 256u + tid) * 40503u + 12345u;
    x ^= x >> 13; x *= 0x5bd1e995u; x ^= x >> 15;
    const float f = (float)(int)(x & 0xFFFFu) * 0.37f - 9000.f;
    const double d = (double)(int)((x >> 8) & 0xFFFFu) * 1.0e-3 - 20.0;
    const int n = (int)((x >> 20) & 0x3FFu);
    sf[tid] = f; sd[tid] = d; si[tid] = n;
    __syncthreads();
    float rmax = -__builtin_inff(); double dmax = -1e300, dsum = 0.0; int isum = 0, scan = 0;
    for (int l = 0; l < 64; ++l) {
        rmax = fmaxf(rmax, sf[w0 + l]); dmax = fmax(dmax, sd[w0 + l]); isum += si[w0 + l];
        if (l <= lane) scan += si[w0 + l];
    }
    // the sum's association: pairs, quads, rows of 16 (ror 4 then ror 8), rows 0+1 | 2+3, halves -- evaluate it the same way
    double rows[4];
    for (int r = 0; r < 4; ++r) {
        double qd[4];
        for (int q = 0; q < 4; ++q) {
            const double* p = &sd[w0 + 16 * r + 4 * q];
            qd[q] = (p[0] + p[1]) + (p[2] + p[3]);
        }
        rows[r] = (qd[0] + qd[1]) + (qd[2] + qd[3]);
    }
    dsum = (rows[0] + rows[1]) + (rows[2] + rows[3]);
    const float quad = (sf[tid & ~3] + sf[(tid & ~3) + 1]) + (sf[(tid & ~3) + 2] + sf[(tid & ~3) + 3]);
    int bad = 0;
    bad += wave_max_f32(f) != rmax;
    bad += wave_max_f64(d) != dmax;
    bad += fabs(wave_sum_f64(d) - dsum) > 1e-9 * (1.0 + fabs(dsum));
    bad += wave_sum_i32(n) != isum;
    bad += wave_scan_incl_i32(n) != scan;
    bad += fabsf(quad_sum_f32(f) - quad) > 1e-3f * (1.f + fabsf(quad));
    if (bad) atomicAdd(mismatches, bad);
}

}  // namespace dagl

extern "C" int dagl_selftest_wave_ops(void* stream, int* mismatches_dev) {
    using namespace dagl;
    if (!mismatches_dev) { set_error("dagl_selftest_wave_ops: bad argument"); return DAGL_ERR_INVALID; }
    DAGL_HIP_TRY(hipMemsetAsync(mismatches_dev, 0, sizeof(int), (hipStream_t)stream));
    hipLaunchKernelGGL(wave_ops_selftest_kernel, dim3(64), dim3(256), 0, (hipStream_t)stream, 20240u, mismatches_dev);
    DAGL_LAUNCH_CHECK("wave_ops_selftest_kernel");
    return DAGL_OK;
}

extern "C" int dagl_probe_mfma_bf16(void* stream, int blocks, int steps, unsigned long long* clocks, float* sink) {
    using namespace dagl;
    if (blocks < 1 || steps < 1 || !clocks || !sink) { set_error("dagl_probe_mfma_bf16: bad argument"); return DAGL_ERR_INVALID; }
    hipLaunchKernelGGL(mfma_probe_kernel, dim3(blocks), dim3(1024), 0, (hipStream_t)stream, steps, clocks, sink);
    DAGL_LAUNCH_CHECK("mfma_probe_kernel");
    return DAGL_OK;
}
