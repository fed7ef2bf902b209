// Run ON THE GPU BOX: issue rate of v_mfma_f32_32x32x16_f16 by dependency pattern, with and without the LDS fragment reads of
// dense.hip's A V block.  One block per CU, W waves per SIMD; per wave the shader clocks (s_memtime) of a loop of 64 blocks x 6
// multiplies.   hipcc --offload-arch=gfx950 -O3 tools/mfma_chain_bench.hip -o /tmp/mcb && /tmp/mcb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef short s4v __attribute__((__vector_size__(4 * sizeof(short))));
typedef short s8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ s4v tr16(unsigned a) { return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4v*)(uintptr_t)a); }

// MODE 0: one accumulator, 6 dependent multiplies per block; 1: two accumulators, 3 + 3 (one chain after the other);
// 2: two accumulators interleaved; 3: four accumulators interleaved; LDS: fragments re-read from the LDS every block (4 tr reads,
// prefetched one block ahead)
template <int MODE, int LDS>
__global__ __launch_bounds__(512) void kern(unsigned* out, int iters) {
    __shared__ __attribute__((aligned(1024))) unsigned short sm[32768];
    for (int i = threadIdx.x; i < 32768; i += blockDim.x) sm[i] = (unsigned short)(0x3c00 + (i & 7));
    __syncthreads();
    const int lane = threadIdx.x & 63;
    f16v acc[4];
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    h8 p0, p1, p2, p3;
    for (int e = 0; e < 8; ++e) { p0[e] = (_Float16)(0.5f + lane * 0.001f); p1[e] = (_Float16)0.25f; p2[e] = (_Float16)0.125f; p3[e] = (_Float16)1.5f; }
    const unsigned base = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)sm + (unsigned)(((lane & 15) >> 2) * 32 + (lane & 3) * 8 + (lane >> 5) * 576);
    s4v f0 = tr16(base), f1 = tr16(base + 128), f2 = tr16(base + 6144), f3 = tr16(base + 6272);
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        s4v n0 = f0, n1 = f1, n2 = f2, n3 = f3;
        if (LDS) { const unsigned a = base + (unsigned)(((it & 7) * 576)); n0 = tr16(a); n1 = tr16(a + 128); n2 = tr16(a + 6144); n3 = tr16(a + 6272); }
        __builtin_amdgcn_sched_barrier(0);
        const s8 vh = {f0[0], f0[1], f0[2], f0[3], f1[0], f1[1], f1[2], f1[3]}, vl = {f2[0], f2[1], f2[2], f2[3], f3[0], f3[1], f3[2], f3[3]};
        const h8 v_hi = __builtin_bit_cast(h8, vh), v_lo = __builtin_bit_cast(h8, vl);
        if (MODE == 0) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(v_hi, p0, acc[0], 0, 0, 0); acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(v_lo, p1, acc[0], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(v_hi, p1, acc[0], 0, 0, 0); acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(v_hi, p2, acc[0], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(v_lo, p3, acc[0], 0, 0, 0); acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(v_hi, p3, acc[0], 0, 0, 0);
        } else if (MODE == 1) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(v_hi, p0, acc[0], 0, 0, 0); acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(v_lo, p1, acc[0], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(v_hi, p1, acc[0], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(v_hi, p2, acc[1], 0, 0, 0); acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(v_lo, p3, acc[1], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(v_hi, p3, acc[1], 0, 0, 0);
        } else if (MODE == 2) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(v_hi, p0, acc[0], 0, 0, 0); acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(v_hi, p2, acc[1], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(v_lo, p1, acc[0], 0, 0, 0); acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(v_lo, p3, acc[1], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(v_hi, p1, acc[0], 0, 0, 0); acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(v_hi, p3, acc[1], 0, 0, 0);
        } else {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(v_hi, p0, acc[0], 0, 0, 0); acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(v_hi, p2, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(v_lo, p1, acc[2], 0, 0, 0); acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(v_lo, p3, acc[3], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(v_hi, p1, acc[0], 0, 0, 0); acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(v_hi, p3, acc[1], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        f0 = n0; f1 = n1; f2 = n2; f3 = n3;
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    if (s == 12345.678f) out[4096] = 1;
    if (lane == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = (unsigned)(t1 - t0);
}

template <int MODE, int LDS>
void run(const char* name, int waves, unsigned* d) {
    const int iters = 512;
    std::vector<unsigned> h(256 * 8);
    hipLaunchKernelGGL((kern<MODE, LDS>), dim3(256), dim3(64 * waves), 0, 0, d, iters);
    hipLaunchKernelGGL((kern<MODE, LDS>), dim3(256), dim3(64 * waves), 0, 0, d, iters);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    double s = 0; for (int b = 0; b < 256; ++b) for (int w = 0; w < waves; ++w) s += h[b * 8 + w];
    s /= 256.0 * waves;
    printf("%-44s %d waves/SIMD: %.1f clocks per multiply per wave, pipe duty %.2f\n", name, waves / 4, s / (iters * 6.0), 32.0 * (waves / 4) / (s / (iters * 6.0)));
}

int main() {
    unsigned* d; hipMalloc(&d, 5000 * 4);
    for (int waves : {4, 8}) {
        run<0, 0>("one accumulator, 6 dependent", waves, d);
        run<1, 0>("two accumulators, 3 + 3 sequential", waves, d);
        run<2, 0>("two accumulators interleaved", waves, d);
        run<3, 0>("four accumulators interleaved", waves, d);
        run<1, 1>("3 + 3 sequential + 4 tr reads per block", waves, d);
        run<2, 1>("interleaved + 4 tr reads per block", waves, d);
    }
    return 0;
}
