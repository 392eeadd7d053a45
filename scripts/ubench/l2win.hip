// micro-benchmark (round 3): random 4-byte gathers / atomics when every workgroup stays inside a WINDOW of the table that
// its XCD's 4 MiB L2 can hold (workgroup b runs on XCD b % 8: window = b % 8 of `win_bytes` each), against the same accesses
// spread over the whole table, and against random LDS reads.  Decides whether an "L2-window" second level can replace
// the LDS slices for tables beyond 2048 LDS slices.
//   hipcc --offload-arch=gfx950 -O3 -o l2win l2win.hip && ./l2win
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ uint64_t splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ULL;
    uint64_t z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

// OP 0 gather (plain load), 1 atomicAdd agent scope (no return), 2 atomicAdd workgroup scope (no return), 3 atomicOr workgroup scope
// window of workgroup b: words [ (b % nwin) * win_words, + win_words )
template <int OP>
__global__ __launch_bounds__(256) void k_win(uint32_t *tab, uint64_t win_words, uint32_t nwin, uint64_t per_thread, uint64_t seed, unsigned long long *sink)
{
    uint32_t *w = tab + (uint64_t)(blockIdx.x % nwin) * win_words;
    uint32_t acc = 0;
    const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    uint64_t r = splitmix64(seed + t);
#pragma unroll 8
    for (uint64_t i = 0; i < per_thread; ++i) {
        r = r * 6364136223846793005ULL + 1442695040888963407ULL;
        const uint64_t idx = __umul64hi(r, win_words);
        if (OP == 0) acc += w[idx];
        else if (OP == 1) __hip_atomic_fetch_add(w + idx, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else if (OP == 2) __hip_atomic_fetch_add(w + idx, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else __hip_atomic_fetch_or(w + idx, 1u << (uint32_t)(r & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    if (OP == 0 && acc == 0x12345678u) atomicAdd(sink, 1ULL);
}

// random LDS reads: BYTES per read 4 / 2 / 1, image of `img_bytes`
template <int BYTES>
__global__ __launch_bounds__(1024) void k_lds(uint64_t per_thread, uint32_t img_bytes, uint64_t seed, unsigned long long *sink)
{
    extern __shared__ uint32_t smem[];
    for (uint32_t i = threadIdx.x; i < img_bytes / 4; i += 1024) smem[i] = i * 2654435761u;
    __syncthreads();
    uint32_t acc = 0;
    uint64_t r = splitmix64(seed + blockIdx.x * 1024 + threadIdx.x);
    const uint32_t n = img_bytes / BYTES;
#pragma unroll 8
    for (uint64_t i = 0; i < per_thread; ++i) {
        r = r * 6364136223846793005ULL + 1442695040888963407ULL;
        const uint32_t idx = (uint32_t)__umul64hi(r, (uint64_t)n);
        if (BYTES == 4) acc += smem[idx];
        else if (BYTES == 2) acc += reinterpret_cast<const uint16_t *>(smem)[idx];
        else acc += reinterpret_cast<const uint8_t *>(smem)[idx];
    }
    if (acc == 0x12345678u) atomicAdd(sink, 1ULL);
}

template <class F>
static float time_ms(F f, int reps = 5)
{
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    f();
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) f();
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms;
    CHECK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

int main()
{
    const uint64_t tab_bytes = 1ULL << 30;
    uint32_t *tab;
    unsigned long long *sink;
    CHECK(hipMalloc(&tab, tab_bytes));
    CHECK(hipMemset(tab, 0, tab_bytes));
    CHECK(hipMalloc(&sink, 8));
    CHECK(hipMemset(sink, 0, 8));
    const uint64_t probes = 70ULL << 20;
    const char *names[4] = {"gather", "atomicAdd agent", "atomicAdd workgroup-scope", "atomicOr workgroup-scope"};
    for (int wgs_per_cu = 4; wgs_per_cu <= 8; wgs_per_cu *= 2) {
        const uint32_t grid = 256 * wgs_per_cu;
        const uint64_t per_thread = probes / ((uint64_t)grid * 256);
        printf("== grid %u x 256 threads, %llu probes per thread (%.1f M probes)\n", grid, (unsigned long long)per_thread, grid * 256.0 * per_thread / 1e6);
        // windows per XCD: nwin = 8 -> one window per XCD; nwin = 1 -> everyone shares one window; whole table: nwin = 1, win = table
        struct Cfg { uint64_t win_bytes; uint32_t nwin; const char *what; } cfgs[] = {
            {512 << 10, 8, "8 windows x 512 KiB (one per XCD)"}, {1 << 20, 8, "8 windows x 1 MiB"},   {2 << 20, 8, "8 windows x 2 MiB"},
            {3 << 20, 8, "8 windows x 3 MiB"},                   {4 << 20, 8, "8 windows x 4 MiB"},   {8 << 20, 8, "8 windows x 8 MiB"},
            {2 << 20, 1, "one 2 MiB window, all XCDs"},          {32 << 20, 1, "whole 32 MiB table"}, {1 << 30, 1, "whole 1 GiB table"},
            {2 << 20, 64, "64 windows x 2 MiB (8 per XCD)"},     {1 << 20, 256, "256 windows x 1 MiB (32 per XCD: one per CU)"},
        };
        for (int op = 0; op < 4; ++op) {
            for (auto &c : cfgs) {
                const uint64_t ww = c.win_bytes / 4;
                float ms = 0;
                switch (op) {
                    case 0: ms = time_ms([&] { hipLaunchKernelGGL(k_win<0>, dim3(grid), dim3(256), 0, 0, tab, ww, c.nwin, per_thread, 1234, sink); }); break;
                    case 1: ms = time_ms([&] { hipLaunchKernelGGL(k_win<1>, dim3(grid), dim3(256), 0, 0, tab, ww, c.nwin, per_thread, 1234, sink); }); break;
                    case 2: ms = time_ms([&] { hipLaunchKernelGGL(k_win<2>, dim3(grid), dim3(256), 0, 0, tab, ww, c.nwin, per_thread, 1234, sink); }); break;
                    default: ms = time_ms([&] { hipLaunchKernelGGL(k_win<3>, dim3(grid), dim3(256), 0, 0, tab, ww, c.nwin, per_thread, 1234, sink); }); break;
                }
                printf("%-28s %-48s %8.3f ms  %7.1f G probes/s\n", names[op], c.what, ms, grid * 256.0 * per_thread / ms / 1e6);
            }
        }
    }
    // LDS
    {
        const uint64_t per_thread = 4096;
        for (uint32_t img : {65536u, 131072u}) {
            CHECK(hipFuncSetAttribute((const void *)k_lds<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
            CHECK(hipFuncSetAttribute((const void *)k_lds<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
            CHECK(hipFuncSetAttribute((const void *)k_lds<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
            float m4 = time_ms([&] { hipLaunchKernelGGL(k_lds<4>, dim3(256), dim3(1024), img, 0, per_thread, img, 99, sink); });
            float m2 = time_ms([&] { hipLaunchKernelGGL(k_lds<2>, dim3(256), dim3(1024), img, 0, per_thread, img, 99, sink); });
            float m1 = time_ms([&] { hipLaunchKernelGGL(k_lds<1>, dim3(256), dim3(1024), img, 0, per_thread, img, 99, sink); });
            const double n = 256.0 * 1024 * per_thread;
            printf("LDS random reads, %u KiB image, one 1024-thread workgroup per CU: b32 %.1f  u16 %.1f  u8 %.1f G probes/s\n", img >> 10, n / m4 / 1e6, n / m2 / 1e6,
                   n / m1 / 1e6);
        }
    }
    return 0;
}
