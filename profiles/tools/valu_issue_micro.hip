// valu_issue_micro.hip -- what a SIMD of gfx950 issues: wave64 FP32 VALU instructions per shader cycle, at 1 .. 4 wavefronts per
// SIMD, independent and dependent streams (one ceiling for bench.py's valu_issue_busy_frac).
//
//   hipcc --offload-arch=gfx950 -O2 profiles/tools/valu_issue_micro.hip -o valu_issue_micro && ./valu_issue_micro
//
// One workgroup per CU (a 100 KB LDS allocation keeps a second one out), 4 * w wavefronts per workgroup = w per SIMD.  Every
// wavefront runs LOOPS x 64 v_fma_f32 -- eight independent chains (one or three source registers), or ONE chain -- between two s_memtime stamps
// (shader cycles, MI355X_MICROARCH.md).  The SIMD's rate = w * instructions / cycles of the slowest wavefront on it; printed as
// the median over the workgroups, with the shader clock (s_memtime against s_memrealtime's 100 MHz) beside it.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>

constexpr int LOOPS = 4096;

// eight independent chains, one source register per instruction (no operand-fetch conflicts)
#define FMA8S                                                                                                          \
    "v_fma_f32 %0, %0, %0, %0\n\tv_fma_f32 %1, %1, %1, %1\n\tv_fma_f32 %2, %2, %2, %2\n\tv_fma_f32 %3, %3, %3, %3\n\t"         \
    "v_fma_f32 %4, %4, %4, %4\n\tv_fma_f32 %5, %5, %5, %5\n\tv_fma_f32 %6, %6, %6, %6\n\tv_fma_f32 %7, %7, %7, %7\n\t"
// eight independent chains, three source registers per instruction (x_i, c, d): what real code looks like
#define FMA8T                                                                                                          \
    "v_fma_f32 %0, %0, %8, %9\n\tv_fma_f32 %1, %1, %8, %9\n\tv_fma_f32 %2, %2, %8, %9\n\tv_fma_f32 %3, %3, %8, %9\n\t"         \
    "v_fma_f32 %4, %4, %8, %9\n\tv_fma_f32 %5, %5, %8, %9\n\tv_fma_f32 %6, %6, %8, %9\n\tv_fma_f32 %7, %7, %8, %9\n\t"
// ONE chain: every instruction waits for the one before it
#define FMA1x8                                                                                                         \
    "v_fma_f32 %0, %0, %0, %0\n\tv_fma_f32 %0, %0, %0, %0\n\tv_fma_f32 %0, %0, %0, %0\n\tv_fma_f32 %0, %0, %0, %0\n\t"         \
    "v_fma_f32 %0, %0, %0, %0\n\tv_fma_f32 %0, %0, %0, %0\n\tv_fma_f32 %0, %0, %0, %0\n\tv_fma_f32 %0, %0, %0, %0\n\t"

template <int KIND>  // 0: eight chains, one source; 1: eight chains, three sources; 2: one chain
__global__ __launch_bounds__(1024) void valu_kernel(float *sink, unsigned long long *cycles, unsigned long long *real, float c, float d) {
    extern __shared__ float lds[];
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    __syncthreads();  // all wavefronts of the CU start together
    unsigned long long t0, t1, r0, r1;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0)::"memory");
    for (int i = 0; i < LOOPS; ++i) {
        if constexpr (KIND == 2) {
            asm volatile(FMA1x8 FMA1x8 FMA1x8 FMA1x8 FMA1x8 FMA1x8 FMA1x8 FMA1x8
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)
                         : "v"(c), "v"(d));
        } else if constexpr (KIND == 1) {
            asm volatile(FMA8T FMA8T FMA8T FMA8T FMA8T FMA8T FMA8T FMA8T
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)
                         : "v"(c), "v"(d));
        } else {
            asm volatile(FMA8S FMA8S FMA8S FMA8S FMA8S FMA8S FMA8S FMA8S
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)
                         : "v"(c), "v"(d));
        }
    }
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1)::"memory");
    if ((threadIdx.x & 63) == 0) {
        const int w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        cycles[w] = t1 - t0;
        real[w] = r1 - r0;
    }
    if (x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 == 12345.678f) sink[0] = lds[threadIdx.x & 7];
}

template <int KIND>
static void run(int w, float *sink, unsigned long long *d_cyc, unsigned long long *d_real, int cus) {
    const int wpb = 4 * w;
    std::vector<unsigned long long> cyc((size_t)cus * wpb), real(cyc.size());
    const size_t lds = 100 * 1024;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&valu_kernel<KIND>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(valu_kernel<KIND>, dim3(cus), dim3(64 * wpb), lds, 0, sink, d_cyc, d_real, 1.0f, 0.5f);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(cyc.data(), d_cyc, cyc.size() * 8, hipMemcpyDeviceToHost);
    (void)hipMemcpy(real.data(), d_real, real.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> rate, mhz;
    const double insts = (double)LOOPS * 64.0;
    for (int b = 0; b < cus; ++b) {
        unsigned long long worst = 0;
        for (int i = 0; i < wpb; ++i) worst = std::max(worst, cyc[(size_t)b * wpb + i]);
        rate.push_back(w * insts / (double)worst);
        mhz.push_back((double)cyc[(size_t)b * wpb] / (double)real[(size_t)b * wpb] * 100.0);
    }
    std::sort(rate.begin(), rate.end());
    std::sort(mhz.begin(), mhz.end());
    const double r = rate[rate.size() / 2];
    printf("%-14s waves_per_simd %d  wave64_valu_insts_per_cycle_per_simd %.4f  cycles_per_inst_per_simd %.3f  cycles_per_inst_per_wave %.3f  shader_clock_MHz %.0f\n",
           KIND == 2 ? "one_chain" : (KIND == 1 ? "8_chains_3src" : "8_chains_1src"), w, r, 1.0 / r, w / r, mhz[mhz.size() / 2]);
}

int main() {
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    float *sink;
    unsigned long long *d_cyc, *d_real;
    (void)hipMalloc(reinterpret_cast<void **>(&sink), 64);
    (void)hipMalloc(reinterpret_cast<void **>(&d_cyc), (size_t)cus * 16 * 8);
    (void)hipMalloc(reinterpret_cast<void **>(&d_real), (size_t)cus * 16 * 8);
    printf("# %s, %d CUs; %d x 64 v_fma_f32 per wavefront between two s_memtime stamps; median over the CUs\n", p.gcnArchName, cus, LOOPS);
    for (int w = 1; w <= 4; ++w) run<0>(w, sink, d_cyc, d_real, cus);
    for (int w = 1; w <= 4; ++w) run<1>(w, sink, d_cyc, d_real, cus);
    for (int w = 1; w <= 4; ++w) run<2>(w, sink, d_cyc, d_real, cus);
    return 0;
}
