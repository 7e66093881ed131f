// gemm_rank.hip -- rank-512 update microbenchmark: the launches that bound the fused factorisation + inversion at N = 8192.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -Imogptk_amd/csrc -Iinclude tools/micro/gemm_rank.hip -o tools/micro/gemm_rank
#include "../../mogptk_amd/csrc/linalg.hip"
#include <cstdio>
#include <cstdlib>
namespace mogp { void set_error(const std::string& m) { fprintf(stderr, "error: %s\n", m.c_str()); }
int hip_fail(hipError_t e, const char* what, const char* file, int line) { fprintf(stderr, "%s: %s (%s:%d)\n", what, hipGetErrorString(e), file, line); return -1; } }
using namespace mogp;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__global__ void k_fill(double* p, size_t n, double v) { for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v * (double)((i * 2654435761u) & 1023) / 1024.0; }
static float timeit(GemmArgs& g, int reps) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    launch_gemm(g, 0); hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < reps; ++i) launch_gemm(g, 0);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / reps;
}
int main(int argc, char** argv) {
    const int n = 8192, K = 512;
    double *A, *C;
    CK(hipMalloc(&A, (size_t)n * n * 8)); CK(hipMalloc(&C, (size_t)n * n * 8));
    hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, A, (size_t)n * n, 1e-3);
    hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, C, (size_t)n * n, 1.0);
    CK(hipDeviceSynchronize());
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
    // rectangular Wt update shape: rem x k0 tiles
    {
        GemmArgs g{};
        g.A = A; g.lda = K; g.a_kmajor = 0; g.B = A; g.ldb = n; g.b_kmajor = 1;
        g.C = C; g.ldc = n; g.alpha = -1.0; g.beta = 1.0; g.mode = GM_RECT; g.mt = 32; g.nt = 28; g.K = K;
        const float ms = timeit(g, 10);
        printf("rect 32x28 : %8.1f us %6.1f TF\n", ms * 1e3, gemm_flops(g, nullptr) / ms / 1e9);
    }
    return 0;
}
