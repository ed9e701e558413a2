// tools/exp_vmmalloc.hip -- an allocation built from SMALL physical chunks mapped in a chosen order (tools/placement_vmm.py).
// Question (profiles/r03_notes.md 1, 11): a buffer's write "mode" belongs to the allocation, not to an address range, and a
// 72 GiB hipMalloc is slow at every offset -- is a physically CONTIGUOUS buffer the slow one (the 256 concurrent write
// streams of a persistent kernel sit at regular strides, i.e. in a fixed channel / bank relation), and does a buffer whose
// physical chunks are scattered behave the same on every box?
//   hipcc -O2 -shared -fPIC --offload-arch=gfx950 tools/exp_vmmalloc.hip -o tools/bin/libvmmalloc.so
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <random>
#include <vector>

namespace {
struct Region {
    void* va;
    size_t bytes, chunk;
    std::vector<hipMemGenericAllocationHandle_t> handles;
};
std::vector<Region> g_regions;
int fail(const char* what, hipError_t e) {
    std::fprintf(stderr, "exp_vmmalloc: %s: %s\n", what, hipGetErrorString(e));
    return 1;
}
}  // namespace

#define VCK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return fail(#x, e_); } while (0)

extern "C" {

// order: 0 = chunks mapped in creation order, 1 = pseudo-random permutation (seed), 2 = reversed,
//        3 = interleave of the two halves of the creation order (chunk i <- i/2 + (i&1) * n/2)
// pool > n_chunks: create `pool` chunks, map a random subset (the others are released) -- scatters over more physical memory.
int vmm_alloc(void** out, size_t bytes, size_t chunk, int order, unsigned seed, int pool_factor_x100) {
    hipMemAllocationProp prop{};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t gran = 0;
    VCK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum));
    if (chunk < gran) chunk = gran;
    chunk = (chunk + gran - 1) / gran * gran;
    const size_t n = (bytes + chunk - 1) / chunk;
    size_t pool = n * (size_t)std::max(100, pool_factor_x100) / 100;
    Region r{};
    r.bytes = n * chunk;
    r.chunk = chunk;
    VCK(hipMemAddressReserve(&r.va, r.bytes, 2ull << 20, nullptr, 0));
    std::vector<hipMemGenericAllocationHandle_t> all(pool);
    for (size_t i = 0; i < pool; ++i) VCK(hipMemCreate(&all[i], chunk, &prop, 0));
    std::vector<size_t> idx(pool);
    for (size_t i = 0; i < pool; ++i) idx[i] = i;
    if (order == 1) {
        std::mt19937 rng(seed);
        std::shuffle(idx.begin(), idx.end(), rng);
    } else if (order == 2) {
        std::reverse(idx.begin(), idx.end());
    } else if (order == 3) {
        for (size_t i = 0; i < pool; ++i) idx[i] = i / 2 + (i & 1) * (pool / 2);
    }
    hipMemAccessDesc acc{};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    for (size_t i = 0; i < n; ++i) {
        VCK(hipMemMap((char*)r.va + i * chunk, chunk, 0, all[idx[i]], 0));
        r.handles.push_back(all[idx[i]]);
    }
    for (size_t i = n; i < pool; ++i) VCK(hipMemRelease(all[idx[i]]));
    VCK(hipMemSetAccess(r.va, r.bytes, &acc, 1));
    *out = r.va;
    g_regions.push_back(r);
    return 0;
}

int vmm_free(void* p) {
    for (size_t k = 0; k < g_regions.size(); ++k) {
        if (g_regions[k].va != p) continue;
        Region& r = g_regions[k];
        VCK(hipMemUnmap(r.va, r.bytes));
        for (auto h : r.handles) VCK(hipMemRelease(h));
        VCK(hipMemAddressFree(r.va, r.bytes));
        g_regions.erase(g_regions.begin() + (long)k);
        return 0;
    }
    return 1;
}

size_t vmm_granularity(int recommended) {
    hipMemAllocationProp prop{};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t gran = 0;
    hipMemGetAllocationGranularity(&gran, &prop, recommended ? hipMemAllocationGranularityRecommended : hipMemAllocationGranularityMinimum);
    return gran;
}
}
