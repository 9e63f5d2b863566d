// profiles/micro/valu_rate.hip -- issue rate of the VALU instruction kinds k_fusedPass is made of, on one MI355X:
// v_mul_f32 / v_add_f32 / v_fma_f32 (VOP2, VGPR operands), v_mul_f32 with an SGPR operand, v_cmp_*_e64 writing an SGPR pair,
// v_cndmask_b32 reading it, v_rcp_f32, v_mul_lo_u32.  Every SIMD gets `waves` resident waves (1..8) running ITERS x 64 independent
// instructions of one kind; cycles per wave-instruction per SIMD = elapsed cycles * SIMDs / total wave-instructions.
// build: hipcc --offload-arch=gfx950 -O2 -o valu_rate valu_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define ITERS 2048
#define R16(X) X X X X X X X X X X X X X X X X
template <int KIND>
__global__ void __launch_bounds__(256) k(float* out, float sv, int n) {
    float a[8];
    for (int i = 0; i < 8; ++i) a[i] = out[(threadIdx.x + i * 64) & 1023] + (float)i;
    unsigned long long m = 0;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p[5];
    for (int i = 0; i < 5; ++i) { p[i].x = a[i]; p[i].y = a[(i + 3) & 7]; }
    f2 sp; sp.x = sv; sp.y = sv * 1.5f;
    for (int it = 0; it < n; ++it) {
        if (KIND == 0) { R16(asm volatile("v_mul_f32 %0, %0, %1\n v_mul_f32 %2, %2, %1\n v_mul_f32 %3, %3, %1\n v_mul_f32 %4, %4, %1" : "+v"(a[0]), "+v"(a[4]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]) :: );) }
        if (KIND == 1) { R16(asm volatile("v_fma_f32 %0, %0, %1, %0\n v_fma_f32 %2, %2, %1, %2\n v_fma_f32 %3, %3, %1, %3\n v_fma_f32 %4, %4, %1, %4" : "+v"(a[0]), "+v"(a[4]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]) :: );) }
        if (KIND == 2) { R16(asm volatile("v_mul_f32 %0, %4, %0\n v_mul_f32 %1, %4, %1\n v_mul_f32 %2, %4, %2\n v_mul_f32 %3, %4, %3" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]) : "s"(sv) : );) }
        if (KIND == 3) { R16(asm volatile("v_cmp_lt_f32 s[20:21], %0, %1\n v_cmp_lt_f32 s[22:23], %1, %2\n v_cmp_lt_f32 s[24:25], %2, %3\n v_cmp_lt_f32 s[26:27], %3, %0" :: "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]) : "s20","s21","s22","s23","s24","s25","s26","s27");) }
        if (KIND == 4) { R16(asm volatile("v_cndmask_b32 %0, %0, %4, %5\n v_cndmask_b32 %1, %1, %4, %5\n v_cndmask_b32 %2, %2, %4, %5\n v_cndmask_b32 %3, %3, %4, %5" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]) : "v"(a[4]), "s"(m) : );) }
        if (KIND == 5) { R16(asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]) :: );) }
        if (KIND == 6) { R16(asm volatile("v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]) : "v"(a[4]) : );) }
        if (KIND == 7) { R16(asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %2, %2, %3, vcc\n v_cmp_lt_f32 vcc, %1, %0\n v_cndmask_b32 %3, %3, %2, vcc" :: "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]) : "vcc");) }
        if (KIND == 8) { R16(asm volatile("v_sub_f32 %0, %0, %1\n v_add_f32 %2, %2, %1\n v_sub_f32 %3, %3, %1\n v_add_f32 %4, %4, %1" : "+v"(a[0]), "+v"(a[4]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]) :: );) }
        if (KIND == 9) { R16(asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4" : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]) : "v"(p[4]) : );) }
        if (KIND == 10) { R16(asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4" : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]) : "v"(p[4]) : );) }
        if (KIND == 11) { R16(asm volatile("v_pk_fma_f32 %0, %0, %4, %0\n v_pk_fma_f32 %1, %1, %4, %1\n v_pk_fma_f32 %2, %2, %4, %2\n v_pk_fma_f32 %3, %3, %4, %3" : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]) : "v"(p[4]) : );) }
        if (KIND == 12) { R16(asm volatile("v_pk_mul_f32 %0, %0, %4 op_sel_hi:[1,0]\n v_pk_mul_f32 %1, %1, %4 op_sel_hi:[1,0]\n v_pk_mul_f32 %2, %2, %4 op_sel_hi:[1,0]\n v_pk_mul_f32 %3, %3, %4 op_sel_hi:[1,0]" : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]) : "v"(p[4]) : );) }
        if (KIND == 13) { R16(asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4" : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]) : "s"(sp) : );) }
        m += (unsigned long long)it;
    }
    float s = 0; for (int i = 0; i < 8; ++i) s += a[i];
    for (int i = 0; i < 5; ++i) s += p[i].x + p[i].y;
    if (s == 12345.678f) out[0] = s + (float)m;
}
template <int KIND> double run(float* d, int waves_per_simd) {
    // 256 CUs x 4 SIMDs; blocks of 256 threads = 4 waves = one per SIMD; waves_per_simd blocks per CU
    const int blocks = 256 * waves_per_simd;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 8);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, ITERS);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_simd = (double)waves_per_simd * ITERS * 64.0;
    return ms * 1e-3 * 2.4e9 / instr_per_simd;   // cycles per wave-instruction per SIMD at a nominal 2.4 GHz
}
int main() {
    float* d; hipMalloc(&d, 4096); hipMemset(d, 0, 4096);
    const char* names[] = {"v_mul_f32 vgpr", "v_fma_f32", "v_mul_f32 sgpr", "v_cmp_f32 -> sgpr", "v_cndmask sgpr-mask", "v_rcp_f32", "v_mul_lo_u32", "v_cmp vcc + v_cndmask vcc (pair)", "v_add/v_sub_f32"};
    for (int w : {1, 2, 4, 6, 8}) {
        printf("waves/SIMD %d:", w);
        printf(" mul %.2f", run<0>(d, w)); printf(" fma %.2f", run<1>(d, w)); printf(" mul_s %.2f", run<2>(d, w)); printf(" cmp_s %.2f", run<3>(d, w));
        printf(" cndmask_s %.2f", run<4>(d, w)); printf(" rcp %.2f", run<5>(d, w)); printf(" mul_lo %.2f", run<6>(d, w)); printf(" cmp+cnd_vcc %.2f", run<7>(d, w));
        printf(" addsub %.2f", run<8>(d, w));
        printf(" | pk_mul %.2f pk_add %.2f pk_fma %.2f pk_mul_opsel %.2f pk_mul_sgpr %.2f\n", run<9>(d, w), run<10>(d, w), run<11>(d, w), run<12>(d, w), run<13>(d, w));
    }
    (void)names;
    return 0;
}
