// Test tool (not part of the product): fills the private-segment (scratch) backing store of every wavefront slot of the
// device with a signalling pattern, so that a kernel which reloads a register spill slot BEFORE storing to it in the same
// launch -- the failure class found twice in round 3 -- computes with 0xFFFFFFFF (a NaN / a huge integer) on the FIRST
// launch instead of with whatever the previous launch happened to leave there.  Built on demand by
// tests/fuzz_random_models.py (`hipcc --offload-arch=gfx950 -shared -fPIC`), called between launches.
#include <hip/hip_runtime.h>
#include <cstdint>

namespace {
constexpr int kWords = 512;  // 2 KiB of scratch per lane: more than any kernel of the library uses (<= 412 B)
__global__ __launch_bounds__(64) void poison_kernel(uint32_t *sink, uint32_t pattern, int rounds) {
    volatile uint32_t a[kWords];
    for (int i = 0; i < kWords; ++i) a[i] = pattern;
    // keep the wavefront resident for a while so that the launch occupies every slot of the chip at once
    uint32_t acc = 0;
    for (int r = 0; r < rounds; ++r)
        for (int i = threadIdx.x & 7; i < kWords; i += 8) acc += a[i];
    if (acc == 0x12345u) sink[0] = acc;  // (never true: keeps the loop)
}
}  // namespace

extern "C" int poison_scratch(void *stream, uint32_t pattern) {
    static uint32_t *sink = nullptr;
    if (!sink && hipMalloc(reinterpret_cast<void **>(&sink), 64) != hipSuccess) return -1;
    // 256 CUs x 32 wave slots would be 8192 waves; launch twice that so that every slot is taken at least once
    hipLaunchKernelGGL(poison_kernel, dim3(16384), dim3(64), 0, (hipStream_t)stream, sink, pattern, 40);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
