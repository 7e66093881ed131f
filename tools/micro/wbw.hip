// wbw.hip -- write-bandwidth microbenchmark: what can the Gram build's store pattern reach on MI355X?
//   hipcc -O3 --offload-arch=gfx950 tools/micro/wbw.hip -o tools/micro/wbw && tools/micro/wbw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d2_t __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void fill_linear(double* p, size_t n2) {          // n2 = number of 16-byte pieces
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (size_t)gridDim.x * blockDim.x)
        reinterpret_cast<d2_t*>(p)[i] = (d2_t){1.0, 2.0};
}
__global__ void copy_linear(const double* a, double* p, size_t n2) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (size_t)gridDim.x * blockDim.x)
        reinterpret_cast<d2_t*>(p)[i] = reinterpret_cast<const d2_t*>(a)[i];
}
// lower-triangle tiles of an n x n matrix, TR x TC tile per 256-thread workgroup, each thread stores 16-byte pieces
template <int TR, int TC, bool NT>
__global__ void fill_tiles(double* p, int64_t ld, int ntc) {  // ntc = tile columns per tile row of the square (TC wide)
    // tile index -> (ti, tj) with tj*TC <= ti*TR + TR - 1   (row-major over the lower part)
    int b = blockIdx.x, ti = 0;
    // rows of tiles: row ti has cnt = (ti*TR + TR + TC - 1) / TC tiles
    while (true) { int cnt = (ti * TR + TR + TC - 1) / TC; if (cnt > ntc) cnt = ntc; if (b < cnt) break; b -= cnt; ++ti; }
    const int tj = b;
    constexpr int PPR = TC / 2;                     // 16-byte pieces per tile row
    for (int idx = threadIdx.x; idx < TR * PPR; idx += 256) {
        const int r = idx / PPR, c = (idx % PPR) * 2;
        d2_t* q = reinterpret_cast<d2_t*>(p + (int64_t)(ti * TR + r) * ld + tj * TC + c);
        if (NT) __builtin_nontemporal_store((d2_t){1.0, 2.0}, q); else *q = (d2_t){1.0, 2.0};
    }
}
// the Gram kernel's thread map: 64x64 tile, thread (rg, cg) stores rows rg*4+m, columns cg*4 .. +3 as two 16-byte pieces
__global__ void fill_gram_map(double* p, int64_t ld) {
    int b = blockIdx.x, ti = 0;
    while (b > ti) { b -= ti + 1; ++ti; }
    const int tj = b, cg = threadIdx.x & 15, rg = threadIdx.x >> 4;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        double* o = p + (int64_t)(ti * 64 + rg * 4 + m) * ld + tj * 64 + cg * 4;
        *reinterpret_cast<d2_t*>(o) = (d2_t){1.0, 2.0};
        *reinterpret_cast<d2_t*>(o + 2) = (d2_t){3.0, 4.0};
    }
}
template <typename F> static float timeit(F f, int reps) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / reps;
}
template <int TR, int TC, bool NT> static void run_tiles(double* p, int n, const char* name) {
    int ntr = n / TR, ntc = n / TC, tiles = 0;
    for (int ti = 0; ti < ntr; ++ti) { int cnt = (ti * TR + TR + TC - 1) / TC; if (cnt > ntc) cnt = ntc; tiles += cnt; }
    const double bytes = (double)tiles * TR * TC * 8;
    float ms = timeit([&]() { hipLaunchKernelGGL((fill_tiles<TR, TC, NT>), dim3(tiles), dim3(256), 0, 0, p, (int64_t)n, ntc); }, 20);
    printf("%-34s %7d tiles %8.1f MB %8.1f us %7.2f TB/s\n", name, tiles, bytes / 1e6, ms * 1e3, bytes / ms / 1e9);
}
int main() {
    const int n = 8192;
    double *p, *q;
    CK(hipMalloc(&p, (size_t)n * n * 8)); CK(hipMalloc(&q, (size_t)n * n * 8));
    CK(hipMemset(p, 0, (size_t)n * n * 8)); CK(hipMemset(q, 0, (size_t)n * n * 8));
    for (size_t mb : {268, 536}) {
        size_t n2 = mb * 1000000 / 16;
        for (int grid : {2048, 8192, 65536}) {
            float ms = timeit([&]() { hipLaunchKernelGGL(fill_linear, dim3(grid), dim3(256), 0, 0, p, n2); }, 20);
            printf("fill_linear %zu MB grid %6d          %8.1f us %7.2f TB/s\n", mb, grid, ms * 1e3, n2 * 16.0 / ms / 1e9);
        }
        float ms = timeit([&]() { hipLaunchKernelGGL(copy_linear, dim3(8192), dim3(256), 0, 0, q, p, n2); }, 20);
        printf("copy_linear %zu MB (read+write)      %8.1f us %7.2f TB/s (sum of both directions)\n", mb, ms * 1e3, 2 * n2 * 16.0 / ms / 1e9);
    }
    float ms = timeit([&]() { hipMemsetAsync(p, 0, 268000000, 0); }, 20);
    printf("hipMemsetAsync 268 MB                 %8.1f us %7.2f TB/s\n", ms * 1e3, 268e6 / ms / 1e9);
    {
        int tiles = 128 * 129 / 2;
        float t = timeit([&]() { hipLaunchKernelGGL(fill_gram_map, dim3(tiles), dim3(256), 0, 0, p, (int64_t)n); }, 20);
        printf("%-34s %7d tiles %8.1f MB %8.1f us %7.2f TB/s\n", "gram thread map 64x64", tiles, tiles * 32768.0 / 1e6, t * 1e3, tiles * 32768.0 / t / 1e9);
    }
    run_tiles<64, 64, false>(p, n, "tiles 64x64");
    run_tiles<64, 64, true>(p, n, "tiles 64x64 nontemporal");
    run_tiles<32, 128, false>(p, n, "tiles 32x128");
    run_tiles<16, 256, false>(p, n, "tiles 16x256");
    run_tiles<16, 256, true>(p, n, "tiles 16x256 nontemporal");
    run_tiles<8, 512, false>(p, n, "tiles 8x512");
    run_tiles<4, 1024, false>(p, n, "tiles 4x1024");
    run_tiles<128, 128, false>(p, n, "tiles 128x128");
    return 0;
}
