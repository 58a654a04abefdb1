// Can a wave of gfx950 stream 64-bit lane masks to memory with scalar stores (s_store_dwordx2) beside dense VALU work, and what does
// it cost?  Each wave runs `iters` rounds of 16 compare instructions (the masks) + 64 integer adds; variant 1 also stores every mask
// with a scalar store, variant 2 stores one dword per lane with a vector buffer store per 4 masks instead (what the fill does today).
// The host checks every stored mask.   hipcc -w --offload-arch=gfx950 -O3 -o scalar_store scalar_store.hip && ./scalar_store
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

template <int VAR>
__global__ __launch_bounds__(512) void k(unsigned long long* out, uint32_t* vout, int iters, int32_t c0) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t wid = blockIdx.x * (blockDim.x >> 6) + wave;
    unsigned long long* base = out + (size_t)wid * iters * 16;
    uint32_t* vbase = vout + (size_t)wid * iters * 4 * 64;
    int32_t a[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) a[q] = lane * (q + 3) + c0;
    for (int it = 0; it < iters; ++it) {
        uint32_t acc = 0;
#pragma unroll
        for (int m = 0; m < 16; ++m) {
#pragma unroll
            for (int q = 0; q < 2; ++q) { a[(m + q) & 7] += a[(m + q + 3) & 7] ^ c0; asm volatile("" : "+v"(a[(m + q) & 7])); }
            const unsigned long long mask = __ballot(((a[m & 7] >> 3) & 1) != 0);
            if (VAR == 1) {
                const unsigned long long* p = base + (size_t)it * 16 + m;
                asm volatile("s_store_dwordx2 %0, %1, 0x0" :: "s"(mask), "s"(p) : "memory");
            }
            else if (VAR == 2) {
                acc = (acc << 8) | ((uint32_t)((a[m & 7] >> 3) & 1));
                if ((m & 3) == 3) { vbase[((size_t)it * 4 + (m >> 2)) * 64 + lane] = acc; acc = 0; }
            }
            else { if (mask == 0x123456789abcdefull) out[0] = mask; }
        }
    }
    if (VAR == 1) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_dcache_wb" ::: "memory");
    int32_t s = 0;
#pragma unroll
    for (int q = 0; q < 8; ++q) s ^= a[q];
    if (s == 0x7ffffff1) out[1] = s;
}

int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount, waves = 8, iters = 10000;
    const size_t nmask = (size_t)cus * 2 * (waves / 2) * iters * 16;      // 2 blocks of 4 waves per CU
    unsigned long long* d; uint32_t* v; const size_t nv = (size_t)cus * 2 * (waves / 2) * iters * 4 * 64;      // dwords of the vector variant
    if (hipMalloc(&d, nmask * 8 + 64) != hipSuccess || hipMalloc(&v, nv * 4 + 64) != hipSuccess) { printf("allocation failed\n"); return 1; }
    for (int var = 0; var < 3; ++var) {
        hipMemset(d, 0, nmask * 8);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        auto launch = [&](int it) { if (var == 0) hipLaunchKernelGGL(k<0>, dim3(cus * 2), dim3(256), 0, 0, d, v, it, 5); else if (var == 1) hipLaunchKernelGGL(k<1>, dim3(cus * 2), dim3(256), 0, 0, d, v, it, 5); else hipLaunchKernelGGL(k<2>, dim3(cus * 2), dim3(256), 0, 0, d, v, it, 5); };
        launch(10); hipEventRecord(e0, 0); launch(iters); hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        printf("variant %d (%s): %.3f ms for %d rounds of 16 masks (v_cmp + 2 other VALU each) per wave, %d waves per CU: %.1f ns per round\n", var,
               var == 0 ? "no stores" : var == 1 ? "scalar store per mask" : "vector store per 4 masks", ms, iters, waves, ms * 1e6 / iters);
        fflush(stdout);
        if (var == 1) {
            std::vector<unsigned long long> h(nmask); hipMemcpy(h.data(), d, nmask * 8, hipMemcpyDeviceToHost);
            // recompute on the host
            size_t bad = 0;
            for (uint32_t wid = 0; wid < (uint32_t)cus * 8 && wid < 8; ++wid) {      // the first 8 waves
                int32_t a[64][8];
                for (int l = 0; l < 64; ++l) for (int q = 0; q < 8; ++q) a[l][q] = l * (q + 3) + 5;
                for (int it = 0; it < iters; ++it) for (int m = 0; m < 16; ++m) {
                    unsigned long long mask = 0;
                    for (int l = 0; l < 64; ++l) { for (int q = 0; q < 2; ++q) a[l][(m + q) & 7] += a[l][(m + q + 3) & 7] ^ 5; if ((a[l][m & 7] >> 3) & 1) mask |= 1ull << l; }
                    if (h[((size_t)wid * iters + it) * 16 + m] != mask) ++bad;
                }
            }
            printf("   masks of the first 8 waves checked on the host: %zu wrong of %d\n", bad, 8 * iters * 16);
        }
    }
    return 0;
}
