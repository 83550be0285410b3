// ed_device.h -- device-side building blocks shared by the CDNA4 kernels.
//
// Everything here is `__device__ __forceinline__`; whether a*b+c contracts to an FMA is decided
// by the translation unit that includes it: deform_exact.hip / spline_filter.hip are compiled
// with -ffp-contract=off (reference evaluation order, bit-comparable with the x86-64 reference
// build, which has no FMA), deform_fast.hip with the default fast contraction.
//
// Reference lines restated by each helper are cited next to it (paths relative to
// /root/reference/elasticdeform/).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "edhip.h"

// Environment switches (kernel selection, ablations, debugging aids) exist only in EDHIP_EXPERIMENTS
// builds (`make EXPERIMENTS=1`, used by tools/ on the GPU box).  The shipped library never calls
// getenv: no environment variable can change what it launches or returns.
#ifdef EDHIP_EXPERIMENTS
#include <cstdlib>
inline const char* ed_env(const char* name) { return getenv(name); }
#else
inline const char* ed_env(const char*) { return nullptr; }
#endif

namespace ed {

// ---------------------------------------------------------------------------------------------
// element access.  (double)*(T*)p  -- deform.c:282-285
// ---------------------------------------------------------------------------------------------
// bfloat16 <-> float: the upper 16 bits of a binary32; round to nearest even on the way down
__device__ __forceinline__ float bf16_to_float(uint16_t b) { return __uint_as_float((uint32_t)b << 16); }
__device__ __forceinline__ uint16_t float_to_bf16(float f)
{
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u)
        return (uint16_t)((u >> 16) | 0x40);                  // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

// bytes per element of an edhip_dtype (0 for an unknown code)
__host__ __device__ __forceinline__ int dtype_size(int dt)
{
    switch (dt) {
    case EDHIP_BOOL: case EDHIP_U8: case EDHIP_I8: return 1;
    case EDHIP_U16: case EDHIP_I16: case EDHIP_F16: case EDHIP_BF16: return 2;
    case EDHIP_U32: case EDHIP_I32: case EDHIP_F32: return 4;
    case EDHIP_U64: case EDHIP_I64: case EDHIP_F64: return 8;
    default: return 0;
    }
}

__device__ __forceinline__ double load_as_double(const char* p, int dt)
{
    switch (dt) {
    case EDHIP_F16: return (double)(float)*(const _Float16*)p;
    case EDHIP_BF16: return (double)bf16_to_float(*(const uint16_t*)p);
    case EDHIP_BOOL:
    case EDHIP_U8: return (double)*(const uint8_t*)p;
    case EDHIP_I8: return (double)*(const int8_t*)p;
    case EDHIP_U16: return (double)*(const uint16_t*)p;
    case EDHIP_I16: return (double)*(const int16_t*)p;
    case EDHIP_U32: return (double)*(const uint32_t*)p;
    case EDHIP_I32: return (double)*(const int32_t*)p;
    case EDHIP_U64: return (double)*(const uint64_t*)p;
    case EDHIP_I64: return (double)*(const int64_t*)p;
    case EDHIP_F32: return (double)*(const float*)p;
    default: return *(const double*)p;
    }
}

// double -> integer conversions with the x86-64 semantics the reference was built with: the
// conversion goes through a signed 32-bit (8/16-bit targets) or signed 64-bit (32/64-bit targets)
// truncating convert and is then narrowed, so out-of-range values of the narrow types wrap
// instead of saturating (matters for integer images pushed through the prefilter).
__device__ __forceinline__ int32_t trunc_i32(double t)
{
    // cvttsd2si r32: out-of-range and NaN give the "integer indefinite" 0x80000000
    return (t >= -2147483648.0 && t < 2147483648.0) ? (int32_t)t : (int32_t)0x80000000;
}
__device__ __forceinline__ int64_t trunc_i64(double t)
{
    // cvttsd2si r64: out-of-range and NaN give 0x8000000000000000
    return (t >= -9223372036854775808.0 && t < 9223372036854775808.0)
               ? (int64_t)t
               : (int64_t)0x8000000000000000ull;
}
__device__ __forceinline__ uint64_t trunc_u64(double t)
{
    // gcc: values below 2^63 use the signed convert, the rest subtract 2^63 first and flip bit 63
    return t < 9223372036854775808.0
               ? (uint64_t)trunc_i64(t)
               : (uint64_t)trunc_i64(t - 9223372036854775808.0) ^ 0x8000000000000000ull;
}

// plain C cast store: the spline filters' line-buffer write-back (from_nd_image.c:422-431) and
// the bool / float / double stores of the forward pass (deform.c:287-290)
__device__ __forceinline__ void store_cast(char* p, int dt, double t)
{
    switch (dt) {
    case EDHIP_F16: *(_Float16*)p = (_Float16)(float)t; break;
    case EDHIP_BF16: *(uint16_t*)p = float_to_bf16((float)t); break;
    case EDHIP_BOOL:
    case EDHIP_U8: *(uint8_t*)p = (uint8_t)trunc_i32(t); break;
    case EDHIP_I8: *(int8_t*)p = (int8_t)trunc_i32(t); break;
    case EDHIP_U16: *(uint16_t*)p = (uint16_t)trunc_i32(t); break;
    case EDHIP_I16: *(int16_t*)p = (int16_t)trunc_i32(t); break;
    case EDHIP_U32: *(uint32_t*)p = (uint32_t)trunc_i64(t); break;
    case EDHIP_I32: *(int32_t*)p = trunc_i32(t); break;
    case EDHIP_U64: *(uint64_t*)p = trunc_u64(t); break;
    case EDHIP_I64: *(int64_t*)p = trunc_i64(t); break;
    case EDHIP_F32: *(float*)p = (float)t; break;
    default: *(double*)p = t; break;
    }
}

// forward store with the per-dtype rounding / clamping of deform.c:292-306,906-919
__device__ __forceinline__ void store_forward(char* p, int dt, double t)
{
    double lo = 0.0, hi = 0.0;
    int is_signed = 0;
    switch (dt) {
    case EDHIP_BOOL: *(uint8_t*)p = (uint8_t)trunc_i32(t); return;
    case EDHIP_F32: *(float*)p = (float)t; return;
    case EDHIP_F64: *(double*)p = t; return;
    case EDHIP_F16:
    case EDHIP_BF16: store_cast(p, dt, t); return;
    case EDHIP_U8: hi = 255.0; break;
    case EDHIP_U16: hi = 65535.0; break;
    case EDHIP_U32: hi = 4294967295.0; break;
    case EDHIP_U64: hi = 18446744073709551615.0; break;
    case EDHIP_I8: lo = -128.0; hi = 127.0; is_signed = 1; break;
    case EDHIP_I16: lo = -32768.0; hi = 32767.0; is_signed = 1; break;
    case EDHIP_I32: lo = -2147483648.0; hi = 2147483647.0; is_signed = 1; break;
    case EDHIP_I64: lo = -9223372036854775808.0; hi = 9223372036854775807.0; is_signed = 1; break;
    default: return;
    }
    if (is_signed)
        t = t > 0 ? t + 0.5 : t - 0.5;
    else
        t = t > 0 ? t + 0.5 : 0.0;
    t = t > hi ? hi : t;
    t = t < lo ? lo : t;
    store_cast(p, dt, t);
}

// ---------------------------------------------------------------------------------------------
// boundary map of a real coordinate, legacy SciPy (<= 1.5) semantics -- deform.c:47-128
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double map_coordinate(double c, int64_t len, int mode)
{
    // reference arithmetic (x86-64, no FMA): keep the products and sums separate in every TU
#pragma clang fp contract(off)
    if (c < 0) {
        switch (mode) {
        case EDHIP_MODE_MIRROR:
            if (len <= 1) {
                c = 0;
            } else {
                const int64_t period = 2 * len - 2;
                c = (double)(period * (int64_t)(-c / (double)period)) + c;
                c = c <= (double)(1 - len) ? c + (double)period : -c;
            }
            break;
        case EDHIP_MODE_REFLECT:
            if (len <= 1) {
                c = 0;
            } else {
                const int64_t period = 2 * len;
                if (c < (double)(-period))
                    c = (double)(period * (int64_t)(-c / (double)period)) + c;
                c = c < (double)(-len) ? c + (double)period : -c - 1;
            }
            break;
        case EDHIP_MODE_WRAP:
            if (len <= 1) {
                c = 0;
            } else {
                const int64_t period = len - 1;
                c += (double)(period * ((int64_t)(-c / (double)period) + 1));
            }
            break;
        case EDHIP_MODE_NEAREST: c = 0; break;
        default: c = -1; break;   // constant
        }
    } else if (c > (double)(len - 1)) {
        switch (mode) {
        case EDHIP_MODE_MIRROR:
            if (len <= 1) {
                c = 0;
            } else {
                const int64_t period = 2 * len - 2;
                c -= (double)(period * (int64_t)(c / (double)period));
                if (c >= (double)len)
                    c = (double)period - c;
            }
            break;
        case EDHIP_MODE_REFLECT:
            if (len <= 1) {
                c = 0;
            } else {
                const int64_t period = 2 * len;
                c -= (double)(period * (int64_t)(c / (double)period));
                if (c >= (double)len)
                    c = (double)period - c - 1;
            }
            break;
        case EDHIP_MODE_WRAP:
            if (len <= 1) {
                c = 0;
            } else {
                const int64_t period = len - 1;
                c -= (double)(period * (int64_t)(c / (double)period));
            }
            break;
        case EDHIP_MODE_NEAREST: c = (double)(len - 1); break;
        default: c = -1; break;   // constant
        }
    }
    return c;
}

// mirror map of an integer tap index onto [0, len): applied to every tap of a window that sticks
// out of the axis, whatever the boundary mode -- deform.c:668-683 (grid), :795-810 (inputs)
__device__ __forceinline__ int64_t mirror_index(int64_t idx, int64_t len)
{
    if (len <= 1)
        return 0;
    const int64_t period = 2 * len - 2;
    if (idx < 0) {
        idx = period * (-idx / period) + idx;
        idx = idx <= 1 - len ? idx + period : -idx;
    } else if (idx >= len) {
        idx -= period * (idx / period);
        if (idx >= len)
            idx = period - idx;
    }
    return idx;
}

// first tap of the (order+1)-wide window around c -- deform.c:657-661,784-788
__device__ __forceinline__ int64_t window_start(double c, int order)
{
    // reference arithmetic (x86-64, no FMA): keep the products and sums separate in every TU
#pragma clang fp contract(off)
    return (int64_t)((order & 1) ? floor(c) : floor(c + 0.5)) - order / 2;
}

// B-spline basis weights at c, fp64, the reference's closed forms and its "last weight = 1 - sum
// of the others" rule -- deform.c:160-268.  w must hold order+1 doubles; order 0 writes nothing.
__device__ __forceinline__ void spline_weights(double x, int order, double* w)
{
    // reference arithmetic (x86-64, no FMA): keep the products and sums separate in every TU
#pragma clang fp contract(off)
    x -= floor((order & 1) ? x : x + 0.5);
    double y = x, z = 1.0 - x, t;
    switch (order) {
    case 1: w[0] = 1.0 - x; break;
    case 2:
        w[1] = 0.75 - x * x;
        y = 0.5 - x;
        w[0] = 0.5 * y * y;
        break;
    case 3:
        w[1] = (y * y * (y - 2.0) * 3.0 + 4.0) / 6.0;
        w[2] = (z * z * (z - 2.0) * 3.0 + 4.0) / 6.0;
        w[0] = z * z * z / 6.0;
        break;
    case 4:
        t = x * x;
        w[2] = t * (t * 0.25 - 0.625) + 115.0 / 192.0;
        y = 1.0 + x;
        w[1] = y * (y * (y * (5.0 - y) / 6.0 - 1.25) + 5.0 / 24.0) + 55.0 / 96.0;
        w[3] = z * (z * (z * (5.0 - z) / 6.0 - 1.25) + 5.0 / 24.0) + 55.0 / 96.0;
        y = 0.5 - x;
        t = y * y;
        w[0] = t * t / 24.0;
        break;
    case 5:
        t = y * y;
        w[2] = t * (t * (0.25 - y / 12.0) - 0.5) + 0.55;
        t = z * z;
        w[3] = t * (t * (0.25 - z / 12.0) - 0.5) + 0.55;
        y += 1.0;
        w[1] = y * (y * (y * (y * (y / 24.0 - 0.375) + 1.25) - 1.75) + 0.625) + 0.425;
        z += 1.0;
        w[4] = z * (z * (z * (z * (z / 24.0 - 0.375) + 1.25) - 1.75) + 0.625) + 0.425;
        y = 1.0 - x;
        t = y * y;
        w[0] = y * t * t / 120.0;
        break;
    default: return;
    }
    double last = 1.0;
    for (int i = 0; i < order; ++i)
        last -= w[i];
    w[order] = last;
}

// control-point coordinate of output index o on one axis -- deform.c:643,655
__device__ __forceinline__ double control_coordinate(int64_t ncp, int64_t o_plus_off, int64_t in_len)
{
    // reference arithmetic (x86-64, no FMA): keep the products and sums separate in every TU
#pragma clang fp contract(off)
    return (double)(ncp - 1) * (double)o_plus_off / (double)(in_len - 1);
}

}  // namespace ed

// 16-bit float storage next to float32 arithmetic (EDHIP_F16 / EDHIP_BF16 on one side of a float32 pass):
// kind 1 = IEEE half, 2 = bfloat16; widening is exact, narrowing rounds to nearest even (what a torch cast does)
__device__ __forceinline__ float widen16(unsigned bits, int kind)
{
    return kind == 2 ? __uint_as_float(bits << 16) : (float)__builtin_bit_cast(_Float16, (unsigned short)bits);
}
__device__ __forceinline__ unsigned narrow16(float x, int kind)
{
    if (kind == 2) {
        unsigned u = __float_as_uint(x);
        if ((u & 0x7fffffffu) > 0x7f800000u)
            return (u >> 16) | 0x40u;                 // NaN stays NaN (quiet)
        u += 0x7fffu + ((u >> 16) & 1u);
        return u >> 16;
    }
    return (unsigned)__builtin_bit_cast(unsigned short, (_Float16)x);
}


// Experiment switches of the kernels (EDHIP_TILE_DBG bits, timestamp buffers): live in the profiling build
// (make EXPERIMENTS=1) only -- in the shipped library the tests below are the constant `false` and the code
// behind them is not compiled in.
#ifdef EDHIP_EXPERIMENTS
#define ED_DBG(word, bits) (((word) & (bits)) != 0)
#define ED_DBG_PTR(ptr) ((ptr) != nullptr)
#else
#define ED_DBG(word, bits) false
#define ED_DBG_PTR(ptr) false
#endif
