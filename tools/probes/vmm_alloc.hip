// Device memory with a chosen PHYSICAL make-up, for tools/probes/vmm_backing.py: hipMalloc, or a virtual range built with the
// virtual-memory API from one physical handle / 2 MiB (granularity) chunks mapped in creation order / in a shuffled order /
// interleaved with chunks that are left unmapped (every other created chunk is skipped, so neighbours are not adjacent physically).
//   hipcc --offload-arch=gfx950 -O2 -shared -fPIC tools/probes/vmm_alloc.hip -o /tmp/libvmm_alloc.so
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <numeric>
#include <vector>

#define CK(e)                                                                       \
    do {                                                                            \
        hipError_t err_ = (e);                                                      \
        if (err_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(err_)); return 1; } \
    } while (0)

static std::vector<hipMemGenericAllocationHandle_t> g_keep;      // spare handles stay alive (never unmapped; the probe process is short)

extern "C" int vmm_alloc(size_t bytes, int mode, unsigned long long seed, void** out, size_t* granularity) {
    *out = nullptr;
    if (mode == 0) { CK(hipMalloc(out, bytes)); return 0; }
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t gran = 0;
    CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum));
    if (granularity) *granularity = gran;
    const size_t size = (bytes + gran - 1) / gran * gran;
    void* ptr = nullptr;
    CK(hipMemAddressReserve(&ptr, size, 0, nullptr, 0));
    if (mode == 1) {
        hipMemGenericAllocationHandle_t h;
        CK(hipMemCreate(&h, size, &prop, 0));
        CK(hipMemMap(ptr, size, 0, h, 0));
    } else {
        const size_t n = size / gran, created = mode == 4 ? 2 * n : n;
        std::vector<hipMemGenericAllocationHandle_t> hs(created);
        for (size_t i = 0; i < created; ++i) CK(hipMemCreate(&hs[i], gran, &prop, 0));
        std::vector<size_t> order(n);
        for (size_t i = 0; i < n; ++i) order[i] = mode == 4 ? 2 * i : i;
        if (mode == 3) {
            unsigned long long s = seed * 0x9E3779B97F4A7C15ull + 12345;
            for (size_t i = n - 1; i > 0; --i) { s = s * 6364136223846793005ull + 1442695040888963407ull; std::swap(order[i], order[(size_t)((s >> 33) % (i + 1))]); }
        }
        for (size_t i = 0; i < n; ++i) CK(hipMemMap(static_cast<char*>(ptr) + i * gran, gran, 0, hs[order[i]], 0));
        if (mode == 4) for (size_t i = 0; i < n; ++i) g_keep.push_back(hs[2 * i + 1]);
    }
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess(ptr, size, &acc, 1));
    *out = ptr;
    return 0;
}
