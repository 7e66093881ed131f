// CU-mask probe: which (xcc, se, sh, cu) does each workgroup land on for a given hipExtStreamCreateWithCUMask mask, and how does
// the throughput of a fixed ALU-bound kernel change.  Build: hipcc -O2 --offload-arch=gfx950 cumask.hip -o cumask
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <set>
#include <map>
#include <vector>
#include <string>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ void probe(uint32_t* out, int iters, double* sink) {
    uint32_t hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    double a = threadIdx.x * 1e-3, b = 1.0000001;
    for (int i = 0; i < iters; ++i) a = a * b + 1e-9;
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = hwid; out[2 * blockIdx.x + 1] = xcc; }
    if (a == 123.456) *sink = a;
}

static void run(const char* name, hipStream_t s, int grid, int iters, uint32_t* d, double* sink) {
    std::vector<uint32_t> h(2 * grid);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    probe<<<grid, 256, 0, s>>>(d, iters, sink);      // warm
    CK(hipStreamSynchronize(s));
    CK(hipEventRecord(e0, s));
    probe<<<grid, 256, 0, s>>>(d, iters, sink);
    CK(hipEventRecord(e1, s));
    CK(hipStreamSynchronize(s));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost));
    std::set<uint32_t> cus; std::map<int, int> per_xcc;
    for (int i = 0; i < grid; ++i) {
        uint32_t hw = h[2 * i], x = h[2 * i + 1] & 0xf;
        uint32_t cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        uint32_t key = (x << 12) | (se << 8) | (sh << 4) | cu;
        if (cus.insert(key).second) per_xcc[x]++;
    }
    printf("%-34s grid %5d  %8.3f ms  distinct CUs %3zu  per-xcc:", name, grid, ms, cus.size());
    for (auto& kv : per_xcc) printf(" %d:%d", kv.first, kv.second);
    printf("\n");
}

int main() {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    printf("device %s CUs %d\n", p.gcnArchName, p.multiProcessorCount);
    uint32_t* d; double* sink; CK(hipMalloc(&d, 1 << 20)); CK(hipMalloc(&sink, 8));
    hipStream_t s0; CK(hipStreamCreate(&s0));
    const int iters = 200000;
    run("unmasked grid=64", s0, 64, iters, d, sink);
    run("unmasked grid=256", s0, 256, iters, d, sink);
    run("unmasked grid=2048", s0, 2048, iters, d, sink);
    struct M { const char* name; std::vector<uint32_t> m; };
    std::vector<M> masks;
    auto mk = [](auto f) { std::vector<uint32_t> m(8, 0); for (int i = 0; i < 256; ++i) if (f(i)) m[i / 32] |= 1u << (i % 32); return m; };
    masks.push_back({"first 128 bits", mk([](int i) { return i < 128; })});
    masks.push_back({"first 248 bits", mk([](int i) { return i < 248; })});
    masks.push_back({"all but bits 0..7", mk([](int i) { return i >= 8; })});
    masks.push_back({"all but i%32==0", mk([](int i) { return i % 32 != 0; })});
    masks.push_back({"all but i%32<2", mk([](int i) { return i % 32 >= 2; })});
    masks.push_back({"only bits 0..7", mk([](int i) { return i < 8; })});
    masks.push_back({"only i%32==0", mk([](int i) { return i % 32 == 0; })});
    masks.push_back({"only bits 0..31", mk([](int i) { return i < 32; })});
    masks.push_back({"all 256", mk([](int i) { return true; })});
    for (auto& mm : masks) {
        hipStream_t s; 
        hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)mm.m.size(), mm.m.data());
        if (e != hipSuccess) { printf("%s: create failed %s\n", mm.name, hipGetErrorString(e)); continue; }
        run(mm.name, s, 2048, iters, d, sink);
        CK(hipStreamDestroy(s));
    }
    return 0;
}
