// gemm_timing.hip -- where does a rank-512 update's workgroup spend its life?  Per-workgroup wall-clock stamps (entry, k loop start, k loop
// end, stores acknowledged) and the CU each ran on, for one 2048-tile launch.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DGEMM_TIMING -Imogptk_amd/csrc -Iinclude tools/micro/gemm_timing.hip -o tools/micro/gemm_timing
#include "../../mogptk_amd/csrc/linalg.hip"
#include <cstdio>
#include <map>
#include <algorithm>
namespace mogp { void set_error(const std::string& m) { fprintf(stderr, "error: %s\n", m.c_str()); }
int hip_fail(hipError_t e, const char* what, const char* file, int line) { fprintf(stderr, "%s: %s (%s:%d)\n", what, hipGetErrorString(e), file, line); return -1; } }
using namespace mogp;
__global__ void k_fill(double* p, size_t n, double v) { for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v * (double)((i * 2654435761u) & 1023) / 1024.0; }
int main() {
    const int n = 8192, K = 512;
    double *A, *C;
    hipMalloc(&A, (size_t)n * n * 8); hipMalloc(&C, (size_t)n * (n + 64) * 8);
    hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, A, (size_t)n * n, 1e-3);
    hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, C, (size_t)n * n, 1.0);
    hipDeviceSynchronize();
    for (int nt : {8, 64, 68}) {
        GemmArgs g{};
        g.A = A; g.lda = K; g.a_kmajor = 0; g.B = A; g.ldb = K; g.b_kmajor = 0;
        g.C = C; g.ldc = n; g.alpha = -1.0; g.beta = 1.0; g.mode = GM_RECT; g.mt = nt == 8 ? 31 : 32; g.nt = nt; g.K = K;
        if (nt == 68) { g.nt = 64; g.beta = 0.0; printf("-- beta = 0: no read of C --\n"); }
        if (nt == 66) { g.nt = 64; g.ldc = n + 16; printf("-- ldc = 8192 + 16 --\n"); }
        if (nt == 67) { g.nt = 64; g.ldc = 128; printf("-- ldc = 128 (a tile is 128 KB contiguous; tiles overlap, timing only) --\n"); }
        const int tiles = g.mt * g.nt;
        for (int rep = 0; rep < 3; ++rep) launch_gemm(g, 0);
        hipDeviceSynchronize();
        std::vector<unsigned long long> t(8 * 8192);
        hipMemcpyFromSymbol(t.data(), HIP_SYMBOL(g_gemm_tim), t.size() * 8);
        unsigned long long tmin = ~0ull, tmax = 0;
        double pro = 0, loop = 0, epi = 0, karg = 0, cld = 0;
        std::map<unsigned long long, std::vector<std::pair<unsigned long long, unsigned long long>>> percu;
        for (int b = 0; b < tiles; ++b) {
            const unsigned long long* r = &t[8 * b];
            tmin = std::min(tmin, r[0]); tmax = std::max(tmax, r[3]);
            pro += (double)(r[1] - r[0]); karg += (double)(r[6] - r[0]); cld += (double)(r[7] - r[6]); loop += (double)(r[2] - r[1]); epi += (double)(r[3] - r[2]);
            const unsigned long long hw = r[4], cu = ((r[5] & 15) << 16) | (hw & 0xff00);      // XCC, SE/SH/CU bits
            percu[cu].push_back({r[0], r[3]});
        }
        printf("tiles %d: kernel arguments %.2f us, C tile read %.2f us (both inside the prologue)\n", tiles, karg / tiles / 100.0, cld / tiles / 100.0);
        printf("tiles %d: span %.1f us; per workgroup: prologue %.2f us, k loop %.2f us, epilogue %.2f us; %zu CUs seen\n", tiles, (tmax - tmin) / 100.0,
               pro / tiles / 100.0, loop / tiles / 100.0, epi / tiles / 100.0, percu.size());
        {   // C read time against the workgroup's start time (50 us buckets)
            double sum[16] = {0}; int cnt[16] = {0};
            for (int b = 0; b < tiles; ++b) { const unsigned long long* r = &t[8 * b]; int k = (int)((r[0] - tmin) / 5000); if (k > 15) k = 15; sum[k] += (double)(r[7] - r[6]); ++cnt[k]; }
            printf("   C read (us) by start time bucket of 50 us [count]:");
            for (int k = 0; k < 16; ++k) if (cnt[k]) printf(" %d:%.1f[%d]", k * 50, sum[k] / cnt[k] / 100.0, cnt[k]);
            printf("\n");
        }
        // occupancy of each CU over the span: time with 0 / 1 / 2 workgroups resident
        double occ[3] = {0, 0, 0};
        for (auto& kv : percu) {
            std::vector<std::pair<unsigned long long, int>> ev;
            for (auto& iv : kv.second) { ev.push_back({iv.first, +1}); ev.push_back({iv.second, -1}); }
            std::sort(ev.begin(), ev.end());
            unsigned long long last = tmin; int lvl = 0;
            for (auto& e : ev) { occ[std::min(lvl, 2)] += (double)(e.first - last); last = e.first; lvl += e.second; }
            occ[0] += (double)(tmax - last);
        }
        const double tot = occ[0] + occ[1] + occ[2];
        printf("   CU time with 0 / 1 / 2 workgroups resident: %.1f%% / %.1f%% / %.1f%%\n", 100 * occ[0] / tot, 100 * occ[1] / tot, 100 * occ[2] / tot);
    }
    return 0;
}
