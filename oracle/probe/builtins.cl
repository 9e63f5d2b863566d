// oracle/probe/builtins.cl -- TEST INFRASTRUCTURE.  One kernel per OpenCL C built-in (and operator) the reference's kernels use
// (nm -u of the compiled code.cl: dot cross normalize length distance fabs sqrt sin cos fmin fmax min max mad clamp; plus / and the
// float -> int conversions), compiled by the SAME toolchain and options as the reference itself (oracle/Makefile `ref_gpu`), so
// that tests/test_ref_gpu.py can compare the CPU model (oracle/cl_numerics.h) and the HIP kernels' numerics layer
// (csrc/pt_numerics.hpp) against what AMD's OpenCL C library really returns on the MI355X, argument by argument.
__kernel void b_sqrt(__global const float* a, __global float* o, uint n) { size_t i = get_global_id(0); if (i < n) o[i] = sqrt(a[i]); }
__kernel void b_sin(__global const float* a, __global float* o, uint n) { size_t i = get_global_id(0); if (i < n) o[i] = sin(a[i]); }
__kernel void b_cos(__global const float* a, __global float* o, uint n) { size_t i = get_global_id(0); if (i < n) o[i] = cos(a[i]); }
__kernel void b_fabs(__global const float* a, __global float* o, uint n) { size_t i = get_global_id(0); if (i < n) o[i] = fabs(a[i]); }
__kernel void b_f2i(__global const float* a, __global int* o, uint n) { size_t i = get_global_id(0); if (i < n) o[i] = (int)a[i]; }
__kernel void b_f2u(__global const float* a, __global uint* o, uint n) { size_t i = get_global_id(0); if (i < n) o[i] = (uint)a[i]; }
__kernel void b_div(__global const float* a, __global const float* b, __global float* o, uint n) { size_t i = get_global_id(0); if (i < n) o[i] = a[i] / b[i]; }
__kernel void b_fmin(__global const float* a, __global const float* b, __global float* o, uint n) { size_t i = get_global_id(0); if (i < n) o[i] = fmin(a[i], b[i]); }
__kernel void b_fmax(__global const float* a, __global const float* b, __global float* o, uint n) { size_t i = get_global_id(0); if (i < n) o[i] = fmax(a[i], b[i]); }
__kernel void b_min(__global const float* a, __global const float* b, __global float* o, uint n) { size_t i = get_global_id(0); if (i < n) o[i] = min(a[i], b[i]); }
__kernel void b_max(__global const float* a, __global const float* b, __global float* o, uint n) { size_t i = get_global_id(0); if (i < n) o[i] = max(a[i], b[i]); }
__kernel void b_mad(__global const float* a, __global const float* b, __global const float* c, __global float* o, uint n) { size_t i = get_global_id(0); if (i < n) o[i] = mad(a[i], b[i], c[i]); }
__kernel void b_clamp(__global const float* a, __global const float* b, __global const float* c, __global float* o, uint n) { size_t i = get_global_id(0); if (i < n) o[i] = clamp(a[i], b[i], c[i]); }
// the contraction the front end applies to a*b+c / a*b-c / c-a*b / a*b+c*d inside one expression
__kernel void b_muladd(__global const float* a, __global const float* b, __global const float* c, __global float* o, uint n) {
    size_t i = get_global_id(0);
    if (i < n) { o[4*i] = a[i]*b[i] + c[i]; o[4*i+1] = a[i]*b[i] - c[i]; o[4*i+2] = c[i] - a[i]*b[i]; o[4*i+3] = a[i]*b[i] + c[i]*a[i]; }
}
// float3 arguments arrive as three consecutive floats
__kernel void b_dot(__global const float* a, __global const float* b, __global float* o, uint n) {
    size_t i = get_global_id(0); if (i < n) o[i] = dot((float3)(a[3*i], a[3*i+1], a[3*i+2]), (float3)(b[3*i], b[3*i+1], b[3*i+2])); }
__kernel void b_cross(__global const float* a, __global const float* b, __global float* o, uint n) {
    size_t i = get_global_id(0); if (i < n) { float3 r = cross((float3)(a[3*i], a[3*i+1], a[3*i+2]), (float3)(b[3*i], b[3*i+1], b[3*i+2])); o[3*i] = r.x; o[3*i+1] = r.y; o[3*i+2] = r.z; } }
__kernel void b_length(__global const float* a, __global float* o, uint n) {
    size_t i = get_global_id(0); if (i < n) o[i] = length((float3)(a[3*i], a[3*i+1], a[3*i+2])); }
__kernel void b_distance(__global const float* a, __global const float* b, __global float* o, uint n) {
    size_t i = get_global_id(0); if (i < n) o[i] = distance((float3)(a[3*i], a[3*i+1], a[3*i+2]), (float3)(b[3*i], b[3*i+1], b[3*i+2])); }
__kernel void b_normalize(__global const float* a, __global float* o, uint n) {
    size_t i = get_global_id(0); if (i < n) { float3 r = normalize((float3)(a[3*i], a[3*i+1], a[3*i+2])); o[3*i] = r.x; o[3*i+1] = r.y; o[3*i+2] = r.z; } }
