// micro-benchmark: what a pass over a table far larger than the Infinity Cache can reach on MI355X, in the shapes the table-pass
// kernels use (k_nib_apply's fold, k_nib_gather's slice load, k_win_fold): one 1024-thread workgroup per 1 MiB slice,
//   MODE 0  read only (16-byte loads, U in flight per lane, OR-reduced)            -- the lookups' slice load
//   MODE 1  read-modify-write in place (load 16 B, add, store 16 B)                -- the updates' fold
//   MODE 2  as 1, but every piece is loaded first into registers in two halves of the slice (deeper pipeline: loads of half 2 under stores of half 1)
// with LDS bytes per workgroup forcing 1 or 2 (or more) workgroups per CU, and plain / nontemporal accesses.
//   hipcc --offload-arch=gfx950 -O3 -o tabpass tabpass.hip && ./tabpass
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int U, bool NT, int NTHR>
__global__ __launch_bounds__(NTHR) void k_pass(uint32_t *tab, uint64_t slice_words, uint32_t nslices, uint32_t *sink)
{
    extern __shared__ uint32_t smem[];
    uint32_t acc = 0;
    for (uint32_t b = blockIdx.x; b < nslices; b += gridDim.x) {
        u32x4 *base = reinterpret_cast<u32x4 *>(tab + (uint64_t)b * slice_words);
        const uint32_t pieces = (uint32_t)(slice_words / 4);
        for (uint32_t p0 = threadIdx.x; p0 < pieces; p0 += NTHR * U) {
            u32x4 t[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t pc = p0 + (uint32_t)u * NTHR;
                t[u] = NT ? __builtin_nontemporal_load(base + pc) : base[pc];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t pc = p0 + (uint32_t)u * NTHR;
                if (MODE == 0) {
                    acc |= t[u].x | t[u].y | t[u].z | t[u].w;
                } else {
                    t[u].x += 1; t[u].y += 2; t[u].z += 3; t[u].w += 4;
                    if (NT) __builtin_nontemporal_store(t[u], base + pc);
                    else base[pc] = t[u];
                }
            }
        }
    }
    if (MODE == 0 && acc == 0x12345u) sink[0] = acc;
    if (threadIdx.x == 0xFFFFFF) smem[0] = acc;
}

template <class F>
static double run(F launch, int iters = 5)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    launch(); hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < iters; ++i) launch();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms * 1e3 / iters;  // us
}

template <int MODE, int U, bool NT, int NTHR>
static void one(uint32_t *tab, uint64_t words, uint32_t *sink, size_t lds, int grid_mult, const char *tag)
{
    const uint64_t slice_words = 1u << 18;  // 1 MiB
    const uint32_t nslices = (uint32_t)(words / slice_words);
    auto kern = k_pass<MODE, U, NT, NTHR>;
    hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const uint32_t grid = grid_mult ? 256u * grid_mult : nslices;
    const double us = run([&] { hipLaunchKernelGGL(kern, dim3(grid), dim3(NTHR), lds, 0, tab, slice_words, nslices, sink); });
    const double bytes = (double)words * 4 * (MODE == 0 ? 1 : 2);
    printf("%-10s mode=%d U=%d nt=%d thr=%4d lds=%6zu grid=%5u  %8.1f us  %6.2f TB/s\n", tag, MODE, U, (int)NT, NTHR, lds, grid, us, bytes / us * 1e-6);
    fflush(stdout);
}

int main(int argc, char **argv)
{
    const uint64_t words = (argc > 1 ? strtoull(argv[1], nullptr, 0) : (1ULL << 28));  // 2^28 counters = 1 GiB
    uint32_t *tab, *sink;
    hipMalloc(&tab, words * 4);
    hipMalloc(&sink, 64);
    hipMemset(tab, 0, words * 4);
    printf("table %.0f MiB\n", words * 4.0 / (1 << 20));
    const size_t L1 = 128 << 10, L2 = 64 << 10, L4 = 32 << 10;
    // read only
    one<0, 2, false, 1024>(tab, words, sink, L1, 0, "read");
    one<0, 4, false, 1024>(tab, words, sink, L1, 0, "read");
    one<0, 8, false, 1024>(tab, words, sink, L1, 0, "read");
    one<0, 8, true, 1024>(tab, words, sink, L1, 0, "read");
    one<0, 8, false, 1024>(tab, words, sink, L1, 1, "read-pers");
    one<0, 8, false, 1024>(tab, words, sink, L2, 0, "read");
    one<0, 8, true, 1024>(tab, words, sink, L2, 0, "read");
    one<0, 8, false, 1024>(tab, words, sink, L2, 2, "read-pers");
    one<0, 8, false, 512>(tab, words, sink, L4, 0, "read");
    one<0, 16, false, 512>(tab, words, sink, L4, 0, "read");
    one<0, 8, false, 256>(tab, words, sink, 0, 0, "read");
    one<0, 8, true, 256>(tab, words, sink, 0, 0, "read");
    // read-modify-write in place
    one<1, 2, false, 1024>(tab, words, sink, L1, 0, "rmw");
    one<1, 4, false, 1024>(tab, words, sink, L1, 0, "rmw");
    one<1, 8, false, 1024>(tab, words, sink, L1, 0, "rmw");
    one<1, 8, true, 1024>(tab, words, sink, L1, 0, "rmw");
    one<1, 8, false, 1024>(tab, words, sink, L1, 1, "rmw-pers");
    one<1, 2, false, 1024>(tab, words, sink, L2, 0, "rmw");
    one<1, 4, false, 1024>(tab, words, sink, L2, 0, "rmw");
    one<1, 8, false, 1024>(tab, words, sink, L2, 0, "rmw");
    one<1, 8, true, 1024>(tab, words, sink, L2, 0, "rmw");
    one<1, 8, false, 1024>(tab, words, sink, L2, 2, "rmw-pers");
    one<1, 8, false, 512>(tab, words, sink, L4, 0, "rmw");
    one<1, 8, false, 256>(tab, words, sink, 0, 0, "rmw");
    one<1, 8, true, 256>(tab, words, sink, 0, 0, "rmw");
    one<1, 4, false, 256>(tab, words, sink, 0, 0, "rmw");
    return 0;
}
