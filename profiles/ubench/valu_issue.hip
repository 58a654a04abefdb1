// How many cycles does one SIMD of gfx950 need per wave64 instruction, by instruction class, with 1, 2, 3 and 4 waves resident?
// MI355X_MICROARCH.md gives 2 cycles per wave64 v_fma_f32 for the SIMD (SIMD-32) and 4 for a single wave's stream.  The fill
// kernels are integer add / compare / select / max code with two waves of <= 256 registers per SIMD; bench.py prices their
// issue roofline with the figure this program measures.
//
// CONTROL ROWS (round 3): the same harness on v_fma_f32 / v_add_f32 / v_pk_fma_f32 / packed 16-bit integer / three-operand integer
// instructions / s_nop, so that the harness is SEEN to reproduce the guide's 2-cycle figure for the class that has it — and
// what each integer class really gets.  A mixed row (vector + scalar instructions interleaved in every wave) shows whether a
// scalar instruction takes an issue turn of its own when two waves share a SIMD.
//
// Every wave runs `iters` rounds of 64 independent instructions of one kind (inline asm: the compiler cannot fold, fuse or
// re-select them); the launch is timed with events, the shader clock is read inside the kernel (s_memtime against the 100 MHz
// s_memrealtime), so the cycles are real cycles, not nominal ones.
//   hipcc -w --offload-arch=gfx950 -O3 -o valu_issue valu_issue.hip && ./valu_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

enum Mix { ADD_U32, MAX_I32, CMP_CNDMASK, DPP_MOV, CHAIN, FMA_F32, ADD_F32, PK_FMA_F32, PK_ADD_U16, PK_MAX_I16, MAX3_I32, ADD3_U32, LSHL_OR, CNDMASK_SGPR,
           S_NOP, V_S_MIX, AND_B32, CMP_ONLY, MOV_B32 };

template <int MIX>
__global__ __launch_bounds__(1024) void spin(int32_t* out, unsigned long long* clk, int iters, int32_t c0) {
    int32_t a[8], b[8];
    float fa[8], fb[8];
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 pa[8], pb[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { a[k] = threadIdx.x * (k + 1); b[k] = c0 + k; fa[k] = (float)a[k]; fb[k] = 1.0f + (float)k * 1e-3f; pa[k] = f2{fa[k], fb[k]}; pb[k] = f2{fb[k], 1.0f}; }
    unsigned long long smask = 0x5555555555555555ull ^ (unsigned long long)c0; asm volatile("" : "+s"(smask));
    int32_t sacc = c0; asm volatile("" : "+s"(sacc));
    const unsigned long long t0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (MIX == ADD_U32) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[k]) : "v"(b[k]));
                else if (MIX == MAX_I32) asm volatile("v_max_i32 %0, %0, %1" : "+v"(a[k]) : "v"(b[k]));
                else if (MIX == CMP_CNDMASK) { const bool w = (a[k] >> 16) >= (b[k] >> 16); a[k] = w ? a[k] : b[k]; asm volatile("" : "+v"(a[k])); }   // v_cmp (sdwa, high halves) + v_cndmask_b32
                else if (MIX == DPP_MOV) { a[k] = __builtin_amdgcn_update_dpp(0, a[k], 0x111, 0xF, 0xF, false); asm volatile("" : "+v"(a[k])); } // v_mov_b32 + v_mov_b32_dpp row_shr:1
                else if (MIX == CHAIN) { const int32_t e = a[k] + c0, o = b[k] + (c0 + 1); const bool w = (e >> 16) >= (o >> 16); a[k] = w ? e : o; b[k] = w ? o : e; asm volatile("" : "+v"(a[k]), "+v"(b[k])); }   // the insertion chain's step: 2 add, cmp, 2 cndmask
                else if (MIX == FMA_F32) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(fa[k]) : "v"(fb[k]));
                else if (MIX == ADD_F32) asm volatile("v_add_f32 %0, %0, %1" : "+v"(fa[k]) : "v"(fb[k]));
                else if (MIX == PK_FMA_F32) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(pa[k]) : "v"(pb[k]));
                else if (MIX == PK_ADD_U16) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(a[k]) : "v"(b[k]));
                else if (MIX == PK_MAX_I16) asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(a[k]) : "v"(b[k]));
                else if (MIX == MAX3_I32) asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b[k]), "v"(b[(k + 1) & 7]));
                else if (MIX == ADD3_U32) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b[k]), "v"(b[(k + 1) & 7]));
                else if (MIX == LSHL_OR) asm volatile("v_lshl_or_b32 %0, %0, 8, %1" : "+v"(a[k]) : "v"(b[k]));
                else if (MIX == CNDMASK_SGPR) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b[k]), "s"(smask));
                else if (MIX == S_NOP) asm volatile("s_nop 0");
                else if (MIX == V_S_MIX) { asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[k]) : "v"(b[k])); asm volatile("s_add_i32 %0, %0, 3" : "+s"(sacc) : : "scc"); }     // one vector, one scalar, interleaved
                else if (MIX == AND_B32) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[k]) : "v"(b[k]));
                else if (MIX == CMP_ONLY) { unsigned long long m; asm volatile("v_cmp_gt_i32_e64 %0, %1, %2" : "=s"(m) : "v"(a[k]), "v"(b[k])); smask ^= m; }                       // v_cmp into an SGPR pair (+ one s_xor_b64)
                else if (MIX == MOV_B32) asm volatile("v_mov_b32 %0, %1" : "=v"(a[k]) : "v"(b[k]));
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    int32_t s = sacc ^ (int32_t)smask;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += a[k] ^ b[k] ^ (int32_t)fa[k] ^ (int32_t)pa[k].x ^ (int32_t)pa[k].y;
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
    printf("-- control rows: floating point and packed classes, three-operand integer, s_nop\n");
    run<FMA_F32>("v_fma_f32", 1, p.multiProcessorCount);
    run<ADD_F32>("v_add_f32", 1, p.multiProcessorCount);
    run<PK_FMA_F32>("v_pk_fma_f32", 1, p.multiProcessorCount);
    run<PK_ADD_U16>("v_pk_add_u16", 1, p.multiProcessorCount);
    run<PK_MAX_I16>("v_pk_max_i16", 1, p.multiProcessorCount);
    run<MAX3_I32>("v_max3_i32", 1, p.multiProcessorCount);
    run<ADD3_U32>("v_add3_u32", 1, p.multiProcessorCount);
    run<LSHL_OR>("v_lshl_or_b32", 1, p.multiProcessorCount);
    run<AND_B32>("v_and_b32", 1, p.multiProcessorCount);
    run<MOV_B32>("v_mov_b32", 1, p.multiProcessorCount);
    run<CNDMASK_SGPR>("v_cndmask_b32_e64 (mask in an SGPR pair)", 1, p.multiProcessorCount);
    run<CMP_ONLY>("v_cmp_gt_i32_e64 -> SGPR pair + s_xor_b64", 2, p.multiProcessorCount);
    run<S_NOP>("s_nop 0", 1, p.multiProcessorCount);
    run<V_S_MIX>("v_add_u32 + s_add_i32 interleaved", 2, p.multiProcessorCount);
    printf("-- the fill kernel's classes (round 2)\n");
    run<ADD_U32>("v_add_u32", 1, p.multiProcessorCount);
    run<MAX_I32>("v_max_i32", 1, p.multiProcessorCount);
    run<CMP_CNDMASK>("v_cmp_ge_i32_sdwa + v_cndmask_b32", 2, p.multiProcessorCount);
    run<DPP_MOV>("v_mov_b32 + v_mov_b32_dpp row_shr:1", 2, p.multiProcessorCount);
    run<CHAIN>("chain step (2 v_add, v_cmp sdwa, 2 v_cndmask)", 5, p.multiProcessorCount);
    return 0;
}
