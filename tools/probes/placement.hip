// Where a write stream lands PHYSICALLY, and what that is worth.  Round 4 found the small workload's fused step 11 % slower when
// its 61 MB obs block is a piece of a large allocation (torch's cached 25.8 GB block) than when it is an allocation of its own -
// with identical TLB and per-TCC-channel counters (profiles/r4_context_pmc.json).  This probe reproduces the two store
// patterns with nothing else around them and backs the destination in different ways:
//   malloc_own      hipMalloc of exactly the bytes written
//   malloc_in_big   the first bytes of one hipMalloc of 26 GB
//   vmm_one         virtual-memory API: one physical handle of the whole size
//   vmm_2m_order    2 MiB physical chunks (granularity), created and mapped in order
//   vmm_2m_shuffle  the same chunks mapped in a shuffled order (a pseudo-random permutation of the 2 MiB pieces)
// Patterns:
//   regions  256 workgroups x 256 threads, each streaming its own contiguous 240 KB region in 4 KB passes, all resident at
//            once (BASELINE config 2's fused LinearObs expansion: 61.4 MB per launch)
//   front    the obs kernel's shape at 4096 x 512: 768-thread workgroups, two 12 KB rows each, XCD-grouped dispatch order,
//            8 GiB per launch (a third of the real block)
//   hipcc --offload-arch=gfx950 -O3 tools/probes/placement.hip -o /tmp/placement && /tmp/placement
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CK(e)                                                                                  \
    do {                                                                                       \
        hipError_t err_ = (e);                                                                 \
        if (err_ != hipSuccess) { std::fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #e, hipGetErrorString(err_)); std::exit(2); } \
    } while (0)

__global__ __launch_bounds__(256) void regions_kernel(f32x4* dst, unsigned region_f4, float v) {
    f32x4* o = dst + (size_t)blockIdx.x * region_f4 + threadIdx.x;
    const f32x4 x = {v, v, v, v};
    // rotated start, as the library does (D2D_TUNE_STEP_OBS_ROTATE)
    const unsigned passes = region_f4 / 256u, p0 = (blockIdx.x * 29u) % passes;
    for (unsigned p = p0; p < passes; ++p) __builtin_nontemporal_store(x, o + (size_t)p * 256u);
    for (unsigned p = 0; p < p0; ++p) __builtin_nontemporal_store(x, o + (size_t)p * 256u);
}

__global__ __launch_bounds__(768) void front_kernel(f32x4* dst, unsigned chunks, float v) {
    const unsigned bid = blockIdx.x, lane8 = bid & 7u, rest = bid >> 3;
    const unsigned chunk = rest % chunks, env = (rest / chunks) * 8u + lane8;
    f32x4* o = dst + ((size_t)env * 512u + (size_t)chunk * 2u) * 768u + threadIdx.x;
    const f32x4 x = {v, v, v, v};
    __builtin_nontemporal_store(x, o);
    __builtin_nontemporal_store(x, o + 768);
}

struct Backing {
    const char* name;
    void* ptr = nullptr;
    size_t bytes = 0;
    std::vector<hipMemGenericAllocationHandle_t> handles;
    bool vmm = false;
    void* owner = nullptr;        // malloc_in_big: the big block
};

static size_t g_gran = 2u << 20;

static Backing make(const char* kind, size_t bytes, int dev) {
    Backing b;
    b.name = kind;
    b.bytes = bytes;
    const std::string k = kind;
    if (k == "malloc_own") { CK(hipMalloc(&b.ptr, bytes)); return b; }
    if (k == "malloc_in_big") { CK(hipMalloc(&b.owner, (size_t)26 << 30)); b.ptr = b.owner; return b; }
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = dev;
    CK(hipMemGetAllocationGranularity(&g_gran, &prop, hipMemAllocationGranularityMinimum));
    const size_t size = (bytes + g_gran - 1) / g_gran * g_gran;
    b.vmm = true;
    b.bytes = size;
    CK(hipMemAddressReserve(&b.ptr, size, 0, nullptr, 0));
    if (k == "vmm_one") {
        hipMemGenericAllocationHandle_t h;
        CK(hipMemCreate(&h, size, &prop, 0));
        CK(hipMemMap(b.ptr, size, 0, h, 0));
        b.handles.push_back(h);
    } else {
        const size_t n = size / g_gran;
        b.handles.resize(n);
        for (size_t i = 0; i < n; ++i) CK(hipMemCreate(&b.handles[i], g_gran, &prop, 0));
        std::vector<size_t> order(n);
        std::iota(order.begin(), order.end(), (size_t)0);
        if (k == "vmm_2m_shuffle") {
            unsigned long long s = 0x9E3779B97F4A7C15ull;
            for (size_t i = n - 1; i > 0; --i) { s = s * 6364136223846793005ull + 1442695040888963407ull; std::swap(order[i], order[(size_t)((s >> 33) % (i + 1))]); }
        }
        for (size_t i = 0; i < n; ++i) CK(hipMemMap(static_cast<char*>(b.ptr) + i * g_gran, g_gran, 0, b.handles[order[i]], 0));
    }
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess(b.ptr, size, &acc, 1));
    return b;
}

static void release(Backing& b) {
    if (b.vmm) {
        CK(hipMemUnmap(b.ptr, b.bytes));
        for (auto h : b.handles) CK(hipMemRelease(h));
        CK(hipMemAddressFree(b.ptr, b.bytes));
    } else {
        CK(hipFree(b.owner ? b.owner : b.ptr));
    }
}

int main(int argc, char** argv) {
    int dev = 0;
    CK(hipSetDevice(dev));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const char* kinds[] = {"malloc_own", "malloc_in_big", "vmm_one", "vmm_2m_order", "vmm_2m_shuffle"};
    const int rounds = argc > 1 ? std::atoi(argv[1]) : 3;
    for (int pattern = 0; pattern < 2; ++pattern) {
        const unsigned region_f4 = 15360;                                   // 240 KB: 4 envs x 50 rows x 75 float4 rounded to whole passes
        const size_t bytes = pattern == 0 ? (size_t)256 * region_f4 * 16 : (size_t)8 << 30;
        for (int r = 0; r < rounds; ++r)
            for (const char* kind : kinds) {
                Backing b = make(kind, bytes, dev);
                const int launches = pattern == 0 ? 200 : 5;
                const unsigned chunks = 256, envs = (unsigned)(bytes / ((size_t)512 * 768 * 16)) & ~7u;
                for (int k = -3; k < launches; ++k) {
                    if (k == 0) CK(hipEventRecord(e0, st));
                    if (pattern == 0) hipLaunchKernelGGL(regions_kernel, dim3(256), dim3(256), 0, st, static_cast<f32x4*>(b.ptr), region_f4, (float)k);
                    else hipLaunchKernelGGL(front_kernel, dim3(envs * chunks), dim3(768), 0, st, static_cast<f32x4*>(b.ptr), chunks, (float)k);
                }
                CK(hipEventRecord(e1, st));
                CK(hipEventSynchronize(e1));
                float ms = 0.f;
                CK(hipEventElapsedTime(&ms, e0, e1));
                const double per = ms / launches, written = pattern == 0 ? (double)bytes : (double)envs * 512 * 768 * 16;
                std::printf("{\"pattern\": \"%s\", \"backing\": \"%s\", \"round\": %d, \"us_per_launch\": %.2f, \"GBps\": %.1f, \"granularity\": %zu}\n",
                            pattern == 0 ? "regions (config 2 expansion)" : "front (obs kernel geometry, 8 GiB)", kind, r, per * 1e3, written / per / 1e6, g_gran);
                std::fflush(stdout);
                release(b);
            }
    }
    return 0;
}
