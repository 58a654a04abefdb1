// How many cycles does one SIMD of gfx950 need per wave64 INTEGER VALU instruction, with 1, 2, 3 and 4 waves resident on it?
// MI355X_MICROARCH.md gives 2 cycles per wave64 v_fma_f32 for the SIMD and 4 for a single wave's stream.  The fill kernels are
// integer add / compare / select / max code with two waves of <= 256 registers per SIMD; bench.py prices their VALU roofline
// with the figure this program measures.  Every wave runs `iters` rounds of 64 independent instructions of one kind; the launch
// is timed with events, the shader clock is read inside the kernel (s_memtime against the 100 MHz wall clock), so the cycles are
// real cycles, not nominal ones.
//   hipcc -w --offload-arch=gfx950 -O3 -o valu_issue valu_issue.hip && ./valu_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int MIX>
__global__ __launch_bounds__(1024) void spin(int32_t* out, unsigned long long* clk, int iters, int32_t c0) {
    int32_t a[8], b[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { a[k] = threadIdx.x * (k + 1); b[k] = c0 + k; }
    const unsigned long long t0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (MIX == 0) { a[k] = a[k] + b[k]; asm volatile("" : "+v"(a[k])); }                                                  // v_add_u32
                else if (MIX == 1) { a[k] = a[k] > b[k] ? a[k] : b[k]; asm volatile("" : "+v"(a[k])); }                                // v_max_i32
                else if (MIX == 2) { const bool w = (a[k] >> 16) >= (b[k] >> 16); a[k] = w ? a[k] : b[k]; asm volatile("" : "+v"(a[k])); }   // v_cmp (sdwa, high halves) + v_cndmask_b32
                else if (MIX == 3) { a[k] = __builtin_amdgcn_update_dpp(0, a[k], 0x111, 0xF, 0xF, false); asm volatile("" : "+v"(a[k])); } // v_mov_b32_dpp row_shr:1
                else { const int32_t e = a[k] + c0, o = b[k] + (c0 + 1); const bool w = (e >> 16) >= (o >> 16); a[k] = w ? e : o; b[k] = w ? o : e; asm volatile("" : "+v"(a[k]), "+v"(b[k])); }   // the insertion chain's step: 2 add, cmp, 2 cndmask
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    int32_t s = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += a[k] ^ b[k];
    if (s == 0x7fffffff) out[0] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = w1 - w0; }
}

template <int MIX>
static void run(const char* name, int per_iter, int cus) {
    int32_t* d; unsigned long long* clk; hipMalloc(&d, 4); hipMalloc(&clk, 16);
    const int iters = 20000;
    for (int waves = 4; waves <= 16; waves += 4) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(spin<MIX>, dim3(cus), dim3(64 * waves), 0, 0, d, clk, 100, 3);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(spin<MIX>, dim3(cus), dim3(64 * waves), 0, 0, d, clk, iters, 3);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
        const double instr_per_simd = (double)iters * 64 * per_iter * (waves / 4.0);
        const double ghz = (double)h[0] / ((double)h[1] * 10.0);                 // shader cycles per ns (wall clock: 100 MHz), over the first wave's loop
        // the launch's duration prices the SIMD (the waves of a SIMD are served oldest first: the first wave's own loop takes the
        // same time whatever else is resident, so its duration says nothing about the SIMD's rate)
        printf("%-46s %d wave(s) per SIMD: launch %7.3f ms = %5.2f shader cycles per instruction per SIMD at the measured %.2f GHz; the oldest wave alone: %5.2f\n",
               name, waves / 4, ms, ms * 1e-3 * ghz * 1e9 / instr_per_simd, ghz, (double)h[0] / ((double)iters * 64 * per_iter));
        hipEventDestroy(e0); hipEventDestroy(e1);
    }
    hipFree(d); hipFree(clk);
}

int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("%s: %d CUs, %.2f GHz nominal\n", p.name, p.multiProcessorCount, p.clockRate / 1e6);
    run<0>("v_add_u32", 1, p.multiProcessorCount);
    run<1>("v_max_i32", 1, p.multiProcessorCount);
    run<2>("v_cmp_ge_i32_sdwa + v_cndmask_b32", 2, p.multiProcessorCount);
    run<3>("v_mov_b32 + v_mov_b32_dpp row_shr:1", 2, p.multiProcessorCount);
    run<4>("chain step (2 v_add, v_cmp sdwa, 2 v_cndmask)", 5, p.multiProcessorCount);
    return 0;
}
