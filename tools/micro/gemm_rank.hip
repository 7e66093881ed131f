// gemm_rank.hip -- rank-512 update microbenchmark: the launches that bound the fused factorisation + inversion at N = 8192.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -Imogptk_amd/csrc -Iinclude tools/micro/gemm_rank.hip -o tools/micro/gemm_rank
#include "../../mogptk_amd/csrc/linalg.hip"
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>
#include <algorithm>
#include <cmath>
namespace mogp { void set_error(const std::string& m) { fprintf(stderr, "error: %s\n", m.c_str()); }
int hip_fail(hipError_t e, const char* what, const char* file, int line) { fprintf(stderr, "%s: %s (%s:%d)\n", what, hipGetErrorString(e), file, line); return -1; } }
using namespace mogp;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__global__ void k_fill(double* p, size_t n, double v) { for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v * (double)((i * 2654435761u) & 1023) / 1024.0; }
// full-entropy mantissas (what real factorisation data look like): splitmix64 -> [0, 1)
__global__ void k_fill_rand(double* p, size_t n, double v, unsigned long long seed) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned long long z = (i + seed) * 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
        p[i] = v * ((double)(z >> 11) * (1.0 / 9007199254740992.0) - 0.5);
    }
}
static float timeit(GemmArgs& g, int reps) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    if (g.sk_spans) g.sk_epoch++;
    launch_gemm(g, 0); hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < reps; ++i) { if (g.sk_spans) g.sk_epoch++; launch_gemm(g, 0); }
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / reps;
}
int main(int argc, char** argv) {
    const int n = 8192, K = 512;
    double *A, *C;
    CK(hipMalloc(&A, (size_t)(n + 256) * n * 8)); CK(hipMalloc(&C, (size_t)(n + 256) * n * 8));
    if (argc > 1 && std::string(argv[1]) == "sk") {             // stream-K form against the tile-per-workgroup form: same result?  how fast?
        const int slots = argc > 2 ? atoi(argv[2]) : 512;
        double *C2, *ws; unsigned* flags;
        CK(hipMalloc(&C2, (size_t)n * n * 8)); CK(hipMalloc(&ws, (size_t)1024 * 128 * 128 * 8)); CK(hipMalloc(&flags, 1024 * 4)); CK(hipMemset(flags, 0, 1024 * 4));
        hipLaunchKernelGGL(k_fill_rand, dim3(2048), dim3(256), 0, 0, A, (size_t)n * n, 2e-2, 1ull);
        { GemmArgs g{}; g.A = A; g.lda = K; g.B = A; g.ldb = K; g.C = C; g.ldc = n; g.alpha = -1.0; g.beta = 1.0; g.mode = GM_RECT; g.mt = 32; g.nt = 64; g.K = K; timeit(g, 150); }   // clocks up
        std::vector<double> h1((size_t)n * n), h2((size_t)n * n);
        unsigned epoch = 1;
        struct Case { const char* name; int mode, mt, nt, K, akm, bkm, b0; };
        const Case cases[] = {
            {"rect 32x64", GM_RECT, 32, 64, 512, 0, 0, 0}, {"rect 31x8", GM_RECT, 31, 8, 512, 0, 0, 0}, {"rect 4x4", GM_RECT, 4, 4, 512, 0, 0, 0},
            {"rect 1x782 K=1920", GM_RECT, 1, 60, 1920, 0, 1, 0},
            {"lower 64 kmajor", GM_LOWER, 64, 64, 512, 1, 1, 0}, {"lower 64 kmajor fresh>=60", GM_LOWER, 64, 64, 512, 1, 1, 61}, {"lower 48", GM_LOWER, 48, 48, 512, 0, 0, 0},
            {"lower 32", GM_LOWER, 32, 32, 512, 0, 0, 0}, {"lower 16", GM_LOWER, 16, 16, 512, 0, 0, 0}, {"lower 4", GM_LOWER, 4, 4, 512, 0, 0, 0},
            {"rect_lower 56x4", GM_RECT_LOWER, 56, 4, 512, 0, 0, 0}, {"rect_lower 8x4", GM_RECT_LOWER, 8, 4, 512, 0, 0, 0},
            {"khi_j 56x4", GM_KHI_J, 56, 4, 512, 0, 0, 0}, {"khi_j 4x4", GM_KHI_J, 4, 4, 512, 0, 0, 0}, {"klo_j 40x4", GM_KLO_J, 40, 4, 512, 0, 1, 0},
            {"rect 32x28 b kmajor", GM_RECT, 32, 28, 512, 0, 1, 0},
        };
        for (const Case& c : cases) {
            GemmArgs g{};
            g.A = A; g.lda = c.akm ? n : c.K; g.a_kmajor = c.akm; g.B = A + 4096; g.ldb = c.bkm ? n : c.K; g.b_kmajor = c.bkm;
            g.ldc = n; g.alpha = -1.0; g.beta = 1.0; g.mode = c.mode; g.mt = c.mt; g.nt = c.nt; g.K = c.K; g.beta0_from = c.b0;
            const int T = c.mode == GM_LOWER ? c.mt * (c.mt + 1) / 2 : c.mt * c.nt;
            const long long tot = (long long)T * (c.K / 16);
            double t[2];
            for (int sk = 0; sk < 2; ++sk) {
                double* Cx = sk ? C2 : C;
                hipLaunchKernelGGL(k_fill_rand, dim3(2048), dim3(256), 0, 0, Cx, (size_t)n * n, 2.0, 77ull);
                g.C = Cx;
                g.sk_spans = sk ? (int)std::min<long long>(slots, tot / 4) : 0; g.sk_ws = ws; g.sk_flags = flags; g.sk_info = nullptr;
                g.sk_epoch = ++epoch;
                launch_gemm(g, 0);
                CK(hipDeviceSynchronize());
                CK(hipMemcpy((sk ? h2 : h1).data(), Cx, (size_t)n * n * 8, hipMemcpyDeviceToHost));
                g.sk_epoch = epoch; t[sk] = timeit(g, 10) * 1e3; epoch = g.sk_epoch;
            }
            double dmax = 0, vmax = 0; size_t nd = 0;
            for (size_t i = 0; i < (size_t)n * n; ++i) { const double d = fabs(h1[i] - h2[i]); if (d > dmax) dmax = d; if (fabs(h1[i]) > vmax) vmax = fabs(h1[i]); nd += d != 0.0; }
            printf("%-28s tiles %5d  tile-per-wg %8.1f us  stream-K(%d) %8.1f us  x%.2f | max |diff| %.2e (max |C| %.2f), %zu entries differ\n", c.name, T, t[0], g.sk_spans, t[1], t[0] / t[1], dmax, vmax, nd);
        }
        return 0;
    }
    if (argc > 1 && std::string(argv[1]) == "series") {          // the same two launches over and over: does the rate depend on how long the process has run?
        hipLaunchKernelGGL(k_fill_rand, dim3(2048), dim3(256), 0, 0, A, (size_t)n * n, 2e-3, 1ull);
        hipLaunchKernelGGL(k_fill_rand, dim3(2048), dim3(256), 0, 0, C, (size_t)n * n, 2.0, 77ull);
        CK(hipDeviceSynchronize());
        hipEvent_t t0, t1; hipEventCreate(&t0); hipEventCreate(&t1); hipEventRecord(t0);
        for (int it = 0; it < 30; ++it) {
            GemmArgs g{};
            g.A = A; g.lda = K; g.a_kmajor = 0; g.B = A; g.ldb = K; g.b_kmajor = 0;
            g.C = C; g.ldc = n; g.alpha = -1e-3; g.beta = 1.0; g.mode = GM_RECT; g.mt = 32; g.nt = 64; g.K = K;
            const float ms = timeit(g, 5);
            GemmArgs h = g; h.lda = h.ldb = n; h.a_kmajor = h.b_kmajor = 1; h.mode = GM_LOWER; h.mt = h.nt = 64;
            const float ms2 = timeit(h, 5);
            hipEventRecord(t1); hipEventSynchronize(t1); float el; hipEventElapsedTime(&el, t0, t1);
            printf("t=%7.1f ms  rect 32x64 %7.1f us %5.1f TF | lower kmajor %7.1f us %5.1f TF\n", el, ms * 1e3, gemm_flops(g, nullptr) / ms / 1e9, ms2 * 1e3, gemm_flops(h, nullptr) / ms2 / 1e9);
        }
        return 0;
    }
    hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, A, (size_t)n * n, 1e-3);
    hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, C, (size_t)n * n, 1.0);
    CK(hipDeviceSynchronize());
    {   // the chip's clocks take ~35 ms of load to ramp (52 -> 64 TFLOP/s on the same launch, "series" mode above): warm up before timing anything
        GemmArgs g{};
        g.A = A; g.lda = K; g.a_kmajor = 0; g.B = A; g.ldb = K; g.b_kmajor = 0;
        g.C = C; g.ldc = n; g.alpha = -1.0; g.beta = 1.0; g.mode = GM_RECT; g.mt = 32; g.nt = 64; g.K = K;
        timeit(g, 150);
        hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, C, (size_t)n * n, 1.0);
    }
    {
    for (int nt : {8, 16, 32, 64, 65})  {
        GemmArgs g{};
        g.A = A; g.lda = K; g.a_kmajor = 0; g.B = A; g.ldb = K; g.b_kmajor = 0;
        g.C = C; g.ldc = n; g.alpha = -1.0; g.beta = 1.0; g.mode = GM_RECT; g.mt = 32; g.nt = nt; g.K = K;
        if (nt == 8) g.mt = 31;
        if (nt == 65) { g.nt = 64; g.beta = 0.0; printf("beta = 0: "); }
        const float ms = timeit(g, 10);
        printf("rect %dx%d tiles %d : %8.1f us %6.1f TF\n", g.mt, nt, g.mt * nt, ms * 1e3, gemm_flops(g, nullptr) / ms / 1e9);
    }
    {   // long-K tiles: the asymptotic MFMA rate (prologue / epilogue amortised)
        GemmArgs g{};
        g.A = A; g.lda = 4096; g.a_kmajor = 0; g.B = A; g.ldb = 4096; g.b_kmajor = 0;
        g.C = C; g.ldc = n; g.alpha = -1.0; g.beta = 1.0; g.mode = GM_RECT; g.mt = 16; g.nt = 16; g.K = 4096;
        float ms = timeit(g, 5);
        printf("rect 16x16 K=4096 tiles 256 : %8.1f us %6.1f TF\n", ms * 1e3, gemm_flops(g, nullptr) / ms / 1e9);
        g.nt = 32;
        ms = timeit(g, 5);
        printf("rect 16x32 K=4096 tiles 512 : %8.1f us %6.1f TF\n", ms * 1e3, gemm_flops(g, nullptr) / ms / 1e9);
    }
    }
    const int mts[] = {64, 60, 48, 32, 22, 16};
    for (int layout = 0; layout < 2; ++layout)
        for (int mt : mts)
            {
                GemmArgs g{};
                g.A = A; g.lda = layout ? n : K; g.a_kmajor = layout; g.B = A; g.ldb = g.lda; g.b_kmajor = layout;
                g.C = C; g.ldc = n; g.alpha = -1.0; g.beta = 1.0; g.mode = GM_LOWER; g.mt = g.nt = mt; g.K = K;
                const float ms = timeit(g, 10);
                const double fl = gemm_flops(g, nullptr);
                printf("layout %s lower mt %2d tiles %4d : %8.1f us %6.1f TF\n", layout ? "kmajor" : "kcontig", mt, mt * (mt + 1) / 2, ms * 1e3, fl / ms / 1e9);
            }
    for (int layout = 0; layout < 2; ++layout) {       // rectangular, both layouts
        GemmArgs g{};
        g.A = A; g.lda = layout ? n : K; g.a_kmajor = layout; g.B = A; g.ldb = g.lda; g.b_kmajor = layout;
        g.C = C; g.ldc = n; g.alpha = -1.0; g.beta = 1.0; g.mode = GM_RECT; g.mt = 32; g.nt = 64; g.K = K;
        const float ms = timeit(g, 10);
        printf("layout %s rect 32x64 : %8.1f us %6.1f TF\n", layout ? "kmajor" : "kcontig", ms * 1e3, gemm_flops(g, nullptr) / ms / 1e9);
    }
    // rectangular Wt update shape: rem x k0 tiles
    {
        GemmArgs g{};
        g.A = A; g.lda = K; g.a_kmajor = 0; g.B = A; g.ldb = n; g.b_kmajor = 1;
        g.C = C; g.ldc = n; g.alpha = -1.0; g.beta = 1.0; g.mode = GM_RECT; g.mt = 32; g.nt = 28; g.K = K;
        const float ms = timeit(g, 10);
        printf("rect 32x28 : %8.1f us %6.1f TF\n", ms * 1e3, gemm_flops(g, nullptr) / ms / 1e9);
    }
    // leading dimension: does the power-of-two row stride (64 KB at N = 8192) cost anything?  (C tile read / write and k-major operands)
    for (int pad : {0, 16, 32, 64, 144}) {
        const int ld = n + pad;
        GemmArgs g{};
        g.A = A; g.lda = K; g.a_kmajor = 0; g.B = A; g.ldb = K; g.b_kmajor = 0;
        g.C = C; g.ldc = ld; g.alpha = -1.0; g.beta = 1.0; g.mode = GM_RECT; g.mt = 32; g.nt = 64; g.K = K;
        float ms = timeit(g, 10);
        printf("ld %d: rect 32x64 kcontig %8.1f us %6.1f TF", ld, ms * 1e3, gemm_flops(g, nullptr) / ms / 1e9);
        g.A = A; g.lda = ld; g.a_kmajor = 1; g.B = A; g.ldb = ld; g.b_kmajor = 1; g.mode = GM_LOWER; g.mt = g.nt = 64;
        ms = timeit(g, 10);
        printf(" | lower 64 kmajor %8.1f us %6.1f TF\n", ms * 1e3, gemm_flops(g, nullptr) / ms / 1e9);
    }
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, A, (size_t)n * n, 1e-3);
        hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, C, (size_t)n * n, 1.0);
        CK(hipDeviceSynchronize());
        for (int nt : {32, 64, 64, 64}) {
            GemmArgs g{};
            g.A = A; g.lda = K; g.a_kmajor = 0; g.B = A; g.ldb = K; g.b_kmajor = 0;
            g.C = C; g.ldc = n; g.alpha = -1.0; g.beta = 1.0; g.mode = GM_RECT; g.mt = 32; g.nt = nt; g.K = K;
            const float ms = timeit(g, 10);
            printf("refilled (10-bit A, 10-bit C) rep %d: rect %dx%d : %8.1f us %6.1f TF\n", rep, g.mt, nt, ms * 1e3, gemm_flops(g, nullptr) / ms / 1e9);
        }
    }
    // operand data: the fills above have 10-bit mantissas; the chip's clock under MFMA load depends on how many bits toggle
    for (int pass = 0; pass < 3; ++pass) {
        const char* what = pass == 0 ? "A 10-bit, C as left" : (pass == 1 ? "A random, C random" : "A zero, C zero");
        if (pass == 1) {
            hipLaunchKernelGGL(k_fill_rand, dim3(2048), dim3(256), 0, 0, A, (size_t)n * n, 2e-3, 1ull);
            hipLaunchKernelGGL(k_fill_rand, dim3(2048), dim3(256), 0, 0, C, (size_t)n * n, 2.0, 77ull);
        }
        if (pass == 2) { CK(hipMemset(A, 0, (size_t)n * n * 8)); CK(hipMemset(C, 0, (size_t)n * n * 8)); }
        CK(hipDeviceSynchronize());
        GemmArgs g{};
        g.A = A; g.lda = K; g.a_kmajor = 0; g.B = A; g.ldb = K; g.b_kmajor = 0;
        g.C = C; g.ldc = n; g.alpha = -1.0; g.beta = 1.0; g.mode = GM_RECT; g.mt = 32; g.nt = 64; g.K = K;
        float ms = timeit(g, 10);
        printf("data %-20s: rect 32x64 %8.1f us %6.1f TF", what, ms * 1e3, gemm_flops(g, nullptr) / ms / 1e9);
        g.lda = g.ldb = n; g.a_kmajor = g.b_kmajor = 1; g.mode = GM_LOWER; g.mt = g.nt = 64;
        ms = timeit(g, 10);
        printf(" | lower 64 kmajor %8.1f us %6.1f TF", ms * 1e3, gemm_flops(g, nullptr) / ms / 1e9);
        g.lda = g.ldb = 4096; g.a_kmajor = g.b_kmajor = 0; g.mode = GM_RECT; g.mt = 16; g.nt = 32; g.K = 4096;
        ms = timeit(g, 5);
        printf(" | K=4096 16x32 %8.1f us %6.1f TF\n", ms * 1e3, gemm_flops(g, nullptr) / ms / 1e9);
    }
    return 0;
}
