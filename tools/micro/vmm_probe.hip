// vmm_probe.hip -- which form of HIP's virtual memory management calls this runtime accepts (the owned-rows work matrix, mogp_model.h:RowBacked)
// build: hipcc --offload-arch=gfx950 -o vmm_probe vmm_probe.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
static int bad = 0;
#define SAY(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { printf("    %-80s -> %s\n", #x, hipGetErrorString(e__)); (void)hipGetLastError(); ++bad; } } while (0)
__global__ void k_touch(double* p, size_t n, double v) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) p[i] = v; }
struct Run { size_t g0, ng; hipMemGenericAllocationHandle_t h; };
static void pattern(const char* name, size_t gran, std::vector<Run> runs, int per_granule_handles = 0) {
    hipMemAllocationProp prop{};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    size_t total = 0;
    for (auto& r : runs) total = std::max(total, (r.g0 + r.ng) * gran);
    total = (total + (2u << 20) - 1) / (2u << 20) * (2u << 20);
    bad = 0;
    void* base = nullptr;
    SAY(hipMemAddressReserve(&base, total, 2u << 20, nullptr, 0));
    hipMemAccessDesc acc{};
    acc.location.type = hipMemLocationTypeDevice; acc.location.id = 0; acc.flags = hipMemAccessFlagsProtReadWrite;
    std::vector<hipMemGenericAllocationHandle_t> hs;
    int idx = 0;
    for (auto& r : runs) {
        const int before = bad;
        if (!per_granule_handles) {
            SAY(hipMemCreate(&r.h, r.ng * gran, &prop, 0));
            SAY(hipMemMap((char*)base + r.g0 * gran, r.ng * gran, 0, r.h, 0));
            SAY(hipMemSetAccess((char*)base + r.g0 * gran, r.ng * gran, &acc, 1));
        } else {
            for (size_t g = 0; g < r.ng; ++g) {
                hipMemGenericAllocationHandle_t h;
                SAY(hipMemCreate(&h, gran, &prop, 0));
                SAY(hipMemMap((char*)base + (r.g0 + g) * gran, gran, 0, h, 0));
                hs.push_back(h);
                if (per_granule_handles == 2) SAY(hipMemSetAccess((char*)base + (r.g0 + g) * gran, gran, &acc, 1));
            }
            if (per_granule_handles == 1) SAY(hipMemSetAccess((char*)base + r.g0 * gran, r.ng * gran, &acc, 1));
        }
        if (bad != before) printf("    ^ run %d (granule %zu, %zu granules)\n", idx, r.g0, r.ng);
        ++idx;
    }
    const int setup_bad = bad;
    if (!setup_bad)
        for (auto& r : runs) {
            double* p = (double*)((char*)base + r.g0 * gran); size_t n = r.ng * gran / 8;
            SAY(hipMemset(p, 0, r.ng * gran));
            k_touch<<<(unsigned)((n + 255) / 256), 256>>>(p, n, 1.0);
        }
    SAY(hipDeviceSynchronize());
    printf("%-58s gran %8zu: %s\n", name, gran, setup_bad ? "FAILED" : (bad ? "set up, but the kernels failed" : "ok"));
    if (!per_granule_handles) for (auto& r : runs) { (void)hipMemUnmap((char*)base + r.g0 * gran, r.ng * gran); (void)hipMemRelease(r.h); }
    else { size_t k = 0; for (auto& r : runs) for (size_t g = 0; g < r.ng; ++g) { (void)hipMemUnmap((char*)base + (r.g0 + g) * gran, gran); (void)hipMemRelease(hs[k++]); } }
    (void)hipMemAddressFree(base, total);
}
int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    (void)hipSetDevice(0);
    const size_t M2 = 2u << 20;
    pattern("A four runs of 1", M2, {{0, 1, {}}, {2, 1, {}}, {4, 1, {}}, {6, 1, {}}});
    pattern("B four runs of 4", M2, {{0, 4, {}}, {8, 4, {}}, {16, 4, {}}, {60, 4, {}}});
    pattern("C runs of 3 5 7 2", M2, {{0, 3, {}}, {8, 5, {}}, {16, 7, {}}, {60, 2, {}}});
    pattern("D 4 4 1 4 (the first probe)", M2, {{0, 4, {}}, {8, 4, {}}, {16, 1, {}}, {60, 4, {}}});
    pattern("E one run of 1", M2, {{5, 1, {}}});
    pattern("F 16 1 16", M2, {{0, 16, {}}, {20, 1, {}}, {40, 16, {}}});
    pattern("G 4 4 4 4 4 4 4 4 (eight runs)", M2, {{0, 4, {}}, {8, 4, {}}, {16, 4, {}}, {24, 4, {}}, {32, 4, {}}, {40, 4, {}}, {48, 4, {}}, {56, 4, {}}});
    pattern("H cyclic rows of 3 MB: 0 2 4 6 8 10", 3u << 20, {{0, 1, {}}, {2, 1, {}}, {4, 1, {}}, {6, 1, {}}, {8, 1, {}}, {10, 1, {}}});
    pattern("I cyclic rows of 8 MB: 1 5 9 13 17 21 25 29", 8u << 20, {{1, 1, {}}, {5, 1, {}}, {9, 1, {}}, {13, 1, {}}, {17, 1, {}}, {21, 1, {}}, {25, 1, {}}, {29, 1, {}}});
    pattern("J as D, one handle per granule, access per run", M2, {{0, 4, {}}, {8, 4, {}}, {16, 1, {}}, {60, 4, {}}}, 1);
    pattern("K as D, one handle per granule, access per granule", M2, {{0, 4, {}}, {8, 4, {}}, {16, 1, {}}, {60, 4, {}}}, 2);
    pattern("L as I, 2 MB handles, access per granule", M2, {{4, 4, {}}, {20, 4, {}}, {36, 4, {}}, {52, 4, {}}, {68, 4, {}}, {84, 4, {}}}, 2);
    pattern("M rows of 32 MB, every 8th (configs[2] on 8 ranks)", 32u << 20, {{3, 1, {}}, {11, 1, {}}, {19, 1, {}}, {27, 1, {}}, {35, 1, {}}, {43, 1, {}}});
    return 0;
}
