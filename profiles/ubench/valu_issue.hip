// How many cycles does one SIMD of gfx950 need per wave64 integer VALU instruction, with one wave on it and with two?
// (MI355X_MICROARCH.md: 2 cycles per instruction for the SIMD, 4 for a single wave's stream.)  The fill kernels run two waves
// of 256 registers per SIMD, so the VALU roofline of bench.py is the two-wave figure measured here.
//   hipcc --offload-arch=gfx950 -O3 -o valu_issue valu_issue.hip && ./valu_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int MIX>
__global__ __launch_bounds__(1024) void spin(int32_t* out, int iters, int32_t c0) {
    int32_t a[8], b[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { a[k] = threadIdx.x * (k + 1); b[k] = c0 + k; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (MIX == 0) { a[k] = a[k] + b[k]; asm volatile("" : "+v"(a[k])); }                          // v_add_u32
                else if (MIX == 1) { a[k] = a[k] > b[k] ? a[k] : b[k]; asm volatile("" : "+v"(a[k])); b[k] += 1; asm volatile("" : "+v"(b[k])); }   // v_max_i32, v_add
                else { const int32_t e = a[k] + c0, o = b[k] + (c0 + 1); const bool w = (e >> 16) >= (o >> 16); a[k] = w ? e : o; b[k] = w ? o : e; asm volatile("" : "+v"(a[k]), "+v"(b[k])); }   // the chain step of the fill: 2 add, cmp (sdwa, high halves), 2 cndmask
            }
        }
    }
    int32_t s = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += a[k] ^ b[k];
    if (s == 0x7fffffff) out[0] = s;
}

template <int MIX>
static void run(const char* name, int per_iter, int cus, double ghz) {
    int32_t* d; hipMalloc(&d, 4);
    const int iters = 20000;
    for (int waves = 4; waves <= 16; waves *= 2) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(spin<MIX>, dim3(cus), dim3(64 * waves), 0, 0, d, 100, 3);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(spin<MIX>, dim3(cus), dim3(64 * waves), 0, 0, d, iters, 3);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        const double instr_per_simd = (double)iters * 64 * per_iter * (waves / 4.0);
        printf("%-28s waves/SIMD %d: %8.3f ms  -> %.2f cycles per instruction per SIMD at %.1f GHz (%.0f instr per wave)\n", name, waves / 4, ms, ms * 1e-3 * ghz * 1e9 / instr_per_simd, ghz, (double)iters * 64 * per_iter);
    }
    hipFree(d);
}

int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const double ghz = p.clockRate / 1e6;
    printf("%s: %d CUs, %.2f GHz nominal\n", p.name, p.multiProcessorCount, ghz);
    run<0>("v_add_u32", 1, p.multiProcessorCount, ghz);
    run<1>("v_max_i32 + v_add_u32", 2, p.multiProcessorCount, ghz);
    run<2>("chain step (2 add, sdwa cmp, 2 cndmask)", 5, p.multiProcessorCount, ghz);
    return 0;
}
