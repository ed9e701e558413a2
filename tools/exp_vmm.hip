// Does the placement effect follow the physical fragment size?  Device memory through HIP's virtual-memory API: physical chunks of a chosen
// size mapped back to back into one reserved range.   hipcc -O2 -shared -fPIC --offload-arch=gfx950 -o tools/bin/libvmm.so tools/exp_vmm.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

extern "C" size_t vmm_granularity(int recommended) {
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t g = 0;
    if (hipMemGetAllocationGranularity(&g, &prop, recommended ? hipMemAllocationGranularityRecommended : hipMemAllocationGranularityMinimum) != hipSuccess) return 0;
    return g;
}

// bytes rounded up to whole chunks; chunk = 0: one physical allocation
static size_t g_align = 0;
extern "C" void vmm_set_align(size_t a) { g_align = a; }
extern "C" void* vmm_alloc(size_t bytes, size_t chunk) {
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t g = 0;
    if (hipMemGetAllocationGranularity(&g, &prop, hipMemAllocationGranularityRecommended) != hipSuccess || g == 0) return nullptr;
    if (chunk == 0) chunk = (bytes + g - 1) / g * g;
    chunk = (chunk + g - 1) / g * g;
    const size_t n = (bytes + chunk - 1) / chunk, total = n * chunk;
    void* base = nullptr;
    if (hipMemAddressReserve(&base, total, g_align, nullptr, 0) != hipSuccess) return nullptr;
    for (size_t i = 0; i < n; ++i) {
        hipMemGenericAllocationHandle_t h;
        if (hipMemCreate(&h, chunk, &prop, 0) != hipSuccess) { fprintf(stderr, "hipMemCreate failed at chunk %zu\n", i); return nullptr; }
        if (hipMemMap((char*)base + i * chunk, chunk, 0, h, 0) != hipSuccess) { fprintf(stderr, "hipMemMap failed\n"); return nullptr; }
        (void)hipMemRelease(h);   // (stays alive through the mapping)
    }
    hipMemAccessDesc d = {};
    d.location.type = hipMemLocationTypeDevice;
    d.location.id = 0;
    d.flags = hipMemAccessFlagsProtReadWrite;
    if (hipMemSetAccess(base, total, &d, 1) != hipSuccess) { fprintf(stderr, "hipMemSetAccess failed\n"); return nullptr; }
    return base;
}

// Per-chunk write rate: n chunks of `chunk` bytes (separate physical allocations), each filled `reps` times; out[i] = microseconds per fill of chunk i
extern "C" int vmm_probe(size_t chunk, int n, int reps, float* out) {
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    void* base = nullptr;
    if (hipMemAddressReserve(&base, chunk * n, 0, nullptr, 0) != hipSuccess) return 1;
    for (int i = 0; i < n; ++i) {
        hipMemGenericAllocationHandle_t h;
        if (hipMemCreate(&h, chunk, &prop, 0) != hipSuccess) return 2;
        if (hipMemMap((char*)base + (size_t)i * chunk, chunk, 0, h, 0) != hipSuccess) return 3;
        (void)hipMemRelease(h);
    }
    hipMemAccessDesc d = {};
    d.location.type = hipMemLocationTypeDevice;
    d.location.id = 0;
    d.flags = hipMemAccessFlagsProtReadWrite;
    if (hipMemSetAccess(base, chunk * n, &d, 1) != hipSuccess) return 4;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int w = 0; w < 200; ++w) hipMemsetAsync(base, 0, chunk, 0);   // clock ramp
    for (int i = 0; i < n; ++i) {
        char* p = (char*)base + (size_t)i * chunk;
        hipMemsetAsync(p, 1, chunk, 0);
        hipEventRecord(e0, 0);
        for (int r = 0; r < reps; ++r) hipMemsetAsync(p, r, chunk, 0);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        out[i] = ms * 1e3f / reps;
    }
    return 0;
}
