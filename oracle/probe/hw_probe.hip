// oracle/probe/hw_probe.hip -- TEST INFRASTRUCTURE: measures, on the MI355X itself, the hardware functions AMD's OpenCL C built-in
// library (opencl.bc / ocml.bc of the ROCm in this image) is made of and whose bits no specification fixes:
//   v_rsq_f32   (normalize() = v * rsqrt(dot(v,v)); rsqrt = llvm.amdgcn.rsq with a 2^24 pre-scale below 2^-126)
//   v_min_f32 / v_max_f32 / v_med3_f32 on NaN and signed-zero operands (min/max/fmin/fmax = llvm.minnum/maxnum, clamp = fmed3)
// so that the CPU checker (oracle/cl_numerics.h) can model them exactly.
//   hw_probe <out-dir>
//     rsq_delta.bin   int8[2][2^23]: bits(rsq(x)) - bits(1.0f / sqrtf(x)), both operations correctly rounded binary32, for x = 1.m (parity 0) and 2 * 1.m (parity 1)
//     report.txt      exhaustive check that rsq(x) for EVERY other float is the same table entry scaled by the power of two,
//                     special values, and the min/max/med3 truth table
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// reference points the CPU can reproduce bit for bit: correctly rounded binary32 sqrt and division (hipcc's default expansions)
__device__ inline float base_rsqrt(float x) { return 1.0f / sqrtf(x); }
__device__ inline float base_sqrt(float x) { return sqrtf(x); }
__device__ inline float ocml_rsqrt(float x) {   // __ocml_rsqrt_f32 with denormals enabled (ocml.bc)
    const bool tiny = x < 1.17549435e-38f;
    const float r = __builtin_amdgcn_rsqf(tiny ? x * 16777216.0f : x);
    return tiny ? r * 4096.0f : r;
}

// v_sqrt_f32 as AMD's length() reaches it (opencl.bc: llvm.sqrt.f32 !fpmath 3.0 -> the bare instruction for x >= 2^-126, and
// ldexp(v_sqrt_f32(ldexp(x, 32)), -16) below)
__device__ inline float ocml_sqrt_approx(float x) {
    const bool tiny = x < 1.17549435e-38f;
    const float r = __builtin_amdgcn_sqrtf(tiny ? ldexpf(x, 32) : x);
    return tiny ? ldexpf(r, -16) : r;
}
__global__ void k_table_sqrt(int8_t* delta, int* worst) {   // x in [1, 4)
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t bits = (127u + (i >> 23)) << 23 | (i & 0x7FFFFFu);
    float x = __uint_as_float(bits);
    int d = (int)__float_as_uint(__builtin_amdgcn_sqrtf(x)) - (int)__float_as_uint(base_sqrt(x));
    delta[i] = (int8_t)d;
    atomicMax(worst, d < 0 ? -d : d);
}
__global__ void k_check_sqrt(const int8_t* delta, unsigned long long* bad, uint32_t* first_bad) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0 || i > 0x7F7FFFFFull) return;
    const float x = __uint_as_float((uint32_t)i);
    const float got = ocml_sqrt_approx(x);
    float xs = x; int e2 = 0;
    if (x < 1.17549435e-38f) { xs = ldexpf(x, 32); e2 = 32; }
    uint32_t b = __float_as_uint(xs);
    int e = (int)(b >> 23) - 127;
    int par = e & 1;
    int k = (e - par) / 2;
    uint32_t idx = ((uint32_t)par << 23) | (b & 0x7FFFFFu);
    float xm = __uint_as_float(((127u + par) << 23) | (b & 0x7FFFFFu));
    float t = __uint_as_float(__float_as_uint(base_sqrt(xm)) + (int)delta[idx]);
    float want = ldexpf(t, k - e2 / 2);
    if (__float_as_uint(want) != __float_as_uint(got)) { atomicAdd(bad, 1ull); atomicMin(first_bad, (uint32_t)i); }
}

__global__ void k_table(int8_t* delta, int* worst) {   // x in [1, 4)
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;          // 0 .. 2^24-1: parity << 23 | mantissa
    uint32_t bits = (127u + (i >> 23)) << 23 | (i & 0x7FFFFFu);
    float x = __uint_as_float(bits);
    int d = (int)__float_as_uint(__builtin_amdgcn_rsqf(x)) - (int)__float_as_uint(base_rsqrt(x));
    delta[i] = (int8_t)d;
    atomicMax(worst, d < 0 ? -d : d);
}

// every positive normal / denormal float: rsq(x) must equal the table entry of its (exponent parity, mantissa) times 2^-(e/2)
__global__ void k_check(const int8_t* delta, unsigned long long* bad, uint32_t* first_bad) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;   // bit patterns 1 .. 0x7F7FFFFF
    if (i == 0 || i > 0x7F7FFFFFull) return;
    const float x = __uint_as_float((uint32_t)i);
    const float got = ocml_rsqrt(x);
    // model
    float xs = x; int e2 = 0;
    if (x < 1.17549435e-38f) { xs = x * 16777216.0f; e2 = 24; }          // exact scaling
    uint32_t b = __float_as_uint(xs);
    int e = (int)(b >> 23) - 127;                                            // unbiased
    int par = e & 1;                                                         // e = 2k + par
    int k = (e - par) / 2;
    uint32_t idx = ((uint32_t)par << 23) | (b & 0x7FFFFFu);
    float xm = __uint_as_float(((127u + par) << 23) | (b & 0x7FFFFFu));     // in [1,4)
    float t = __uint_as_float(__float_as_uint(base_rsqrt(xm)) + (int)delta[idx]);
    float want = ldexpf(t, -k + e2 / 2);
    if (__float_as_uint(want) != __float_as_uint(got)) { atomicAdd(bad, 1ull); atomicMin(first_bad, (uint32_t)i); }
}

__global__ void k_ops(const float* a, const float* b, const float* c, float* out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[4 * i + 0] = __builtin_fminf(a[i], b[i]);                 // llvm.minnum -> v_min_f32
    out[4 * i + 1] = __builtin_fmaxf(a[i], b[i]);                 // llvm.maxnum -> v_max_f32
    out[4 * i + 2] = __builtin_amdgcn_fmed3f(a[i], b[i], c[i]);   // v_med3_f32
    out[4 * i + 3] = ocml_rsqrt(a[i]);
}

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: hw_probe <out-dir>\n"); return 2; }
    std::string dir = argv[1];
    int8_t* d_delta; int* d_worst; unsigned long long* d_bad; uint32_t* d_first;
    CK(hipMalloc(&d_delta, 1u << 24)); CK(hipMalloc(&d_worst, 4)); CK(hipMalloc(&d_bad, 8)); CK(hipMalloc(&d_first, 4));
    CK(hipMemset(d_worst, 0, 4)); CK(hipMemset(d_bad, 0, 8)); CK(hipMemset(d_first, 0xFF, 4));
    hipLaunchKernelGGL(k_table, dim3((1u << 24) / 256), dim3(256), 0, 0, d_delta, d_worst);
    hipLaunchKernelGGL(k_check, dim3((0x7F800000u + 255) / 256), dim3(256), 0, 0, d_delta, d_bad, d_first);
    CK(hipDeviceSynchronize());
    std::vector<int8_t> delta(1u << 24);
    int worst; unsigned long long bad; uint32_t first;
    CK(hipMemcpy(delta.data(), d_delta, 1u << 24, hipMemcpyDeviceToHost));
    CK(hipMemcpy(&worst, d_worst, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&bad, d_bad, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(&first, d_first, 4, hipMemcpyDeviceToHost));
    FILE* f = fopen((dir + "/rsq_delta.bin").c_str(), "wb"); fwrite(delta.data(), 1, delta.size(), f); fclose(f);
    FILE* r = fopen((dir + "/report.txt").c_str(), "w");
    long hist[9] = {0};
    for (int8_t v : delta) { int k = v + 4; if (k < 0) k = 0; if (k > 8) k = 8; hist[k]++; }
    fprintf(r, "v_rsq_f32 vs 1.0f/sqrtf(x) over [1,4): max |delta| = %d ulp; histogram of delta -4..+4:", worst);
    for (int k = 0; k < 9; ++k) fprintf(r, " %ld", hist[k]);
    fprintf(r, "\nscaling check over all %u positive finite bit patterns (incl. denormals through ocml's 2^24 pre-scale): %llu mismatches (first bit pattern 0x%08x)\n",
            0x7F7FFFFFu, bad, first);
    {   // the same for v_sqrt_f32
        CK(hipMemset(d_worst, 0, 4)); CK(hipMemset(d_bad, 0, 8)); CK(hipMemset(d_first, 0xFF, 4));
        hipLaunchKernelGGL(k_table_sqrt, dim3((1u << 24) / 256), dim3(256), 0, 0, d_delta, d_worst);
        hipLaunchKernelGGL(k_check_sqrt, dim3((0x7F800000u + 255) / 256), dim3(256), 0, 0, d_delta, d_bad, d_first);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(delta.data(), d_delta, 1u << 24, hipMemcpyDeviceToHost));
        CK(hipMemcpy(&worst, d_worst, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&bad, d_bad, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(&first, d_first, 4, hipMemcpyDeviceToHost));
        FILE* g = fopen((dir + "/sqrt_delta.bin").c_str(), "wb"); fwrite(delta.data(), 1, delta.size(), g); fclose(g);
        long h2[9] = {0};
        for (int8_t v : delta) { int k = v + 4; if (k < 0) k = 0; if (k > 8) k = 8; h2[k]++; }
        fprintf(r, "v_sqrt_f32 vs sqrtf(x) over [1,4): max |delta| = %d ulp; histogram of delta -4..+4:", worst);
        for (int k = 0; k < 9; ++k) fprintf(r, " %ld", h2[k]);
        fprintf(r, "\nscaling check over all positive finite bit patterns (denormals through the 2^32 pre-scale of length()): %llu mismatches (first bit pattern 0x%08x)\n", bad, first);
        printf("sqrt: worst %d ulp, %llu scaling mismatches\n", worst, bad);
    }
    // truth tables
    const float Z = 0.0f, NZ = -0.0f, I = INFINITY, N = NAN;
    float sn; { uint32_t u = 0x7FA00000u; memcpy(&sn, &u, 4); }
    float nn; { uint32_t u = 0xFFC00001u; memcpy(&nn, &u, 4); }
    std::vector<float> vals = {Z, NZ, 1.0f, -1.0f, 255.0f, 300.0f, I, -I, N, sn, nn, 1e-40f, -1e-40f};
    std::vector<float> A, B, Cc;
    for (float a : vals) for (float b : vals) for (float c : {0.0f, 1.0f, 255.0f, N}) { A.push_back(a); B.push_back(b); Cc.push_back(c); }
    int n = (int)A.size();
    float *da, *db, *dc, *dout;
    CK(hipMalloc(&da, n * 4)); CK(hipMalloc(&db, n * 4)); CK(hipMalloc(&dc, n * 4)); CK(hipMalloc(&dout, n * 16));
    CK(hipMemcpy(da, A.data(), n * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(db, B.data(), n * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dc, Cc.data(), n * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_ops, dim3((n + 63) / 64), dim3(64), 0, 0, da, db, dc, dout, n);
    std::vector<float> out(4 * n);
    CK(hipMemcpy(out.data(), dout, n * 16, hipMemcpyDeviceToHost));
    auto u = [](float x) { uint32_t v; memcpy(&v, &x, 4); return v; };
    fprintf(r, "a b c | minnum(a,b) maxnum(a,b) med3(a,b,c) rsqrt(a)   (bit patterns)\n");
    for (int i = 0; i < n; ++i)
        fprintf(r, "%08x %08x %08x | %08x %08x %08x %08x\n", u(A[i]), u(B[i]), u(Cc[i]), u(out[4 * i]), u(out[4 * i + 1]), u(out[4 * i + 2]), u(out[4 * i + 3]));
    fclose(r);
    printf("worst %d ulp, %llu scaling mismatches\n", worst, bad);
    return 0;
}
