// Run ON THE GPU BOX: what costs dense_attend_kernel's multiplying waves their duty.  One block per CU shaped like the kernel's: 8
// multiplying waves (two per SIMD; per "tile" 2 k-blocks x 3 column tiles x 6 v_mfma_f32_32x32x16_f16 on 6 accumulators, 4
// transposing LDS reads per column tile, prefetched one ahead) and optionally 4 more waves (one per SIMD) with the producers' stream
// (per tile 28 ds_read_b128, 42 v_mfma_f32_16x16x32_f16 in 14 fenced slots, ~18 VALU operations per slot, s_setprio 2).  Features are
// switched on one by one (bit mask F): 1 = one s_barrier per tile, 2 = the weights' reads per k-block (4 ds_read_b128 + ballot +
// uniform branch), 4 = the producer waves, 8 = the producers' multiplies too (4 without 8: their LDS reads and VALU only),
// 16 = LDS-DMA pieces (7 per producer, 1.5 per multiplying wave and tile, from an L2-resident buffer); 32 / 64 = the producers
// without their VALU work / without their LDS reads; 128 = producers at normal priority.
//   hipcc --offload-arch=gfx950 -O3 tools/dense_loop_bench.hip -o /tmp/dlb && /tmp/dlb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));
typedef short s4v __attribute__((__vector_size__(4 * sizeof(short))));
typedef short s8 __attribute__((ext_vector_type(8)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ s4v tr16(unsigned a) { return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4v*)(uintptr_t)a); }
__device__ __forceinline__ u4 lds128(unsigned a) { return *(const __attribute__((address_space(3))) u4*)(uintptr_t)a; }
__device__ __forceinline__ void glds16(const char* g, unsigned voff, unsigned lds) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %1\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "s"(g), "v"(voff), "s"(lds) : "memory");
}

template <int F>
__global__ __launch_bounds__(768) void kern(unsigned* out, const char* gbuf, int tiles) {
    __shared__ __attribute__((aligned(1024))) unsigned short sm[65536];           // 128 KiB
    for (int i = threadIdx.x; i < 65536; i += blockDim.x) sm[i] = (unsigned short)(0x3c00 + (i & 7));
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned lds0 = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)sm;
    const bool producer = wave >= 8;
    unsigned long long t0 = 0, t1 = 0;
    if (!producer) {
        f16v acc[6];
        for (int a = 0; a < 6; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
        const unsigned base = lds0 + (unsigned)(((lane & 15) >> 2) * 32 + (lane & 3) * 8 + (lane >> 5) * 576 + wave * 1024);
        const unsigned pbase = lds0 + 98304u + (unsigned)(lane * 80);
        s4v f0 = tr16(base), f1 = tr16(base + 128), f2 = tr16(base + 6144), f3 = tr16(base + 6272);
        h8 ph[2], pl[2];
        for (int e = 0; e < 8; ++e) { ph[0][e] = (_Float16)(0.5f + lane * 0.001f); ph[1][e] = (_Float16)0.25f; pl[0][e] = (_Float16)0.125f; pl[1][e] = (_Float16)1.5f; }
        t0 = __builtin_readcyclecounter();
        for (int t = 0; t < tiles; ++t) {
            if (F & 16) { glds16(gbuf, (unsigned)(lane * 16 + wave * 1024), lds0 + 110592u + (unsigned)(wave * 1024)); }
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                if (F & 2) {
                    const u4 a0 = lds128(pbase + 16 * kb), a1 = lds128(pbase + 16 * kb + 32), a2 = lds128(pbase + 5120 + 16 * kb), a3 = lds128(pbase + 5120 + 16 * kb + 32);
                    const bool nz = __builtin_amdgcn_ballot_w64(((a0.x | a0.y) | (a0.z | a0.w) | (a2.x | a2.y)) != 0u) != 0ull;
                    ph[0] = __builtin_bit_cast(h8, a0); pl[0] = __builtin_bit_cast(h8, a1); ph[1] = __builtin_bit_cast(h8, a2); pl[1] = __builtin_bit_cast(h8, a3);
                    if (!nz) continue;
                }
#pragma unroll
                for (int ct = 0; ct < 3; ++ct) {
                    const unsigned an = base + (unsigned)((((t * 6 + kb * 3 + ct) & 7) * 576));
                    const s4v n0 = tr16(an), n1 = tr16(an + 128), n2 = tr16(an + 6144), n3 = tr16(an + 6272);
                    __builtin_amdgcn_sched_barrier(0);
                    const s8 vh = {f0[0], f0[1], f0[2], f0[3], f1[0], f1[1], f1[2], f1[3]}, vl = {f2[0], f2[1], f2[2], f2[3], f3[0], f3[1], f3[2], f3[3]};
                    const h8 v_hi = __builtin_bit_cast(h8, vh), v_lo = __builtin_bit_cast(h8, vl);
                    acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(v_hi, pl[0], acc[ct], 0, 0, 0); acc[3 + ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(v_hi, pl[1], acc[3 + ct], 0, 0, 0);
                    acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(v_lo, ph[0], acc[ct], 0, 0, 0); acc[3 + ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(v_lo, ph[1], acc[3 + ct], 0, 0, 0);
                    acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(v_hi, ph[0], acc[ct], 0, 0, 0); acc[3 + ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(v_hi, ph[1], acc[3 + ct], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    f0 = n0; f1 = n1; f2 = n2; f3 = n3;
                }
            }
            if (F & 16) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (F & 1) __syncthreads();
        }
        t1 = __builtin_readcyclecounter();
        float s = 0.f;
        for (int a = 0; a < 6; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
        if (s == 12345.678f) out[8192] = 1;
    } else {
        if (!(F & 4)) {                                     // absent producers: only keep the barrier count
            if (F & 1) for (int t = 0; t < tiles; ++t) __syncthreads();
        } else {
            if (!(F & 128)) __builtin_amdgcn_s_setprio(2);
            f4v s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0;
            h8 q0, q1;
            for (int e = 0; e < 8; ++e) { q0[e] = (_Float16)(0.5f + lane * 0.001f); q1[e] = (_Float16)0.25f; }
            const unsigned kb_addr = lds0 + 65536u + (unsigned)((lane & 15) * 448 + (lane >> 4) * 16);
            float x = 0.3f + lane * 1e-3f, z = 0.f;
            t0 = __builtin_readcyclecounter();
            for (int t = 0; t < tiles; ++t) {
                if (F & 16) for (int j = 0; j < 7; ++j) glds16(gbuf, (unsigned)(lane * 16 + j * 1024 + (wave - 8) * 8192), lds0 + 32768u + (unsigned)((wave - 8) * 7168 + j * 1024));
                u4 ka[3], kb_[3];                           // fragment pairs two slots ahead, as in the kernel
                auto kfrag = [&](int sl) {
                    ka[sl % 3] = u4{1u, 2u, 3u, (unsigned)sl}; kb_[sl % 3] = u4{5u, 6u, 7u, (unsigned)t};
                    if (!(F & 64)) { ka[sl % 3] = lds128(kb_addr + (unsigned)(64 * sl)); kb_[sl % 3] = lds128(kb_addr + 14336u + (unsigned)(64 * sl)); }
                };
                kfrag(0); kfrag(1);
#pragma unroll
                for (int sl = 0; sl < 14; ++sl) {
                    if (sl + 2 < 14) kfrag(sl + 2);
                    __builtin_amdgcn_sched_barrier(0);
                    const u4 k0 = ka[sl % 3], k1 = kb_[sl % 3];
                    if (F & 8) {
                        s0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, k0), q1, s0, 0, 0, 0);
                        s1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, k0), q0, s1, 0, 0, 0);
                        s2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, k1), q0, s2, 0, 0, 0);
                    } else {
                        z += __builtin_bit_cast(float, k0.x ^ k1.y);
                    }
                    if (!(F & 32))
#pragma unroll
                    for (int u = 0; u < 4; ++u) { x = __builtin_fmaf(x, 1.0001f, 0.01f); z += __expf(fminf(x - 3.f, 0.f)); x = x > 2.f ? x - 1.f : x; }
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (F & 16) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (F & 1) __syncthreads();
            }
            t1 = __builtin_readcyclecounter();
            if (s0[0] + s1[1] + s2[2] + z == 12345.678f) out[8192] = 1;
        }
    }
    if (lane == 0) out[blockIdx.x * 12 + wave] = (unsigned)(t1 - t0);
}

template <int F>
void run(const char* name, unsigned* d, const char* g) {
    const int tiles = 256;
    std::vector<unsigned> h(256 * 12);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((kern<F>), dim3(256), dim3(768), 0, 0, d, g, tiles);
    hipEventRecord(e0);
    hipLaunchKernelGGL((kern<F>), dim3(256), dim3(768), 0, 0, d, g, tiles);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    double s = 0; for (int b = 0; b < 256; ++b) for (int w = 0; w < 8; ++w) s += h[b * 12 + w];
    s /= 256.0 * 8 * tiles;
    const double pipe = 72 * 32 + ((F & 8) ? 42 * 16 : 0);           // matrix cycles per SIMD and tile
    printf("%-78s %7.0f ticks per tile, %6.3f ms -> %5.2f us per tile; matrix work per SIMD and tile %4.0f cycles\n", name, s, ms, ms * 1e3 / tiles, pipe);
}

int main() {
    unsigned* d; hipMalloc(&d, 9000 * 4);
    char* g; hipMalloc(&g, 1 << 20); hipMemset(g, 0, 1 << 20);
    run<0>("multiplying waves alone (2 per SIMD), no barrier", d, g);
    run<1>("+ one barrier per tile", d, g);
    run<2>("weights' reads per k-block, no barrier", d, g);
    run<3>("weights' reads + barrier", d, g);
    run<3 | 4>("+ producer waves: LDS reads + VALU only", d, g);
    run<3 | 4 | 8>("+ the producers' multiplies", d, g);
    run<3 | 4 | 8 | 16>("+ LDS-DMA pieces", d, g);
    run<2 | 4 | 8 | 16>("everything without the barrier", d, g);
    run<1 | 4 | 8>("barrier + producers with multiplies, no weights' reads", d, g);
    run<3 | 4 | 32>("producers: LDS reads only (no VALU, no multiplies)", d, g);
    run<3 | 4 | 64>("producers: VALU only (no LDS reads, no multiplies)", d, g);
    run<3 | 4 | 64 | 128>("producers: VALU only, normal priority", d, g);
    run<3 | 4 | 128>("producers: LDS reads + VALU, normal priority", d, g);
    run<3 | 4 | 8 | 128>("producers: LDS reads + VALU + multiplies, normal priority", d, g);
    run<3 | 4 | 8 | 32>("producers: LDS reads + multiplies, no VALU", d, g);
    run<3 | 4 | 8 | 64>("producers: VALU + multiplies, no LDS reads", d, g);
    return 0;
}
