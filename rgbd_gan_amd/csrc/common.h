// Shared helpers for the gfx950 kernels of librgbdgan_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/rgbd_gan_hip.h"
#include "rgbd_debug.h"

typedef __bf16 bf16_t;
typedef bf16_t bf16x8 __attribute__((ext_vector_type(8)));
typedef bf16_t bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

void rgbd_set_error(const char* fmt, ...);

// Zero `bytes` (multiple of 4) with a kernel launch instead of hipMemsetAsync: identical in eager mode, and inside
// a captured HIP graph it is an ordinary kernel node in the dependency chain (memset nodes proved unreliable on
// replay with ROCm 7.2).  Returns hipSuccess or the launch error.
hipError_t rgbd_zero_async(void* ptr, size_t bytes, hipStream_t stream);

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE property of a kernel: reserved once per (kernel, device) -- a
// process that drives a second GPU must set it there too (a per-process flag left that device at the 64 KB default).
bool rgbd_reserve_lds(const void* fn, int bytes);

#define RGBD_REQUIRE(cond, ...)            \
    do {                                   \
        if (!(cond)) {                     \
            rgbd_set_error(__VA_ARGS__);   \
            return -1;                     \
        }                                  \
    } while (0)

#define RGBD_CHECK_LAUNCH(name)                                                       \
    do {                                                                              \
        hipError_t e_ = hipGetLastError();                                            \
        if (e_ != hipSuccess) {                                                       \
            rgbd_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));     \
            return -2;                                                                \
        }                                                                             \
    } while (0)

__device__ __forceinline__ float bf16_bits_to_f32(unsigned short b) {
    return __uint_as_float(((unsigned int)b) << 16);
}
__device__ __forceinline__ unsigned short f32_to_bf16_bits(float f) {
    bf16_t h = (bf16_t)f;  // v_cvt_pk_bf16_f32: round-to-nearest-even, NaN stays NaN
    return __builtin_bit_cast(unsigned short, h);
}
typedef bf16_t bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned int pack_bf16x2(float lo, float hi) {
    // ONE v_cvt_pk_bf16_f32 for the pair (two scalar conversions + shift + or took four VALU slots: a conv epilogue
    // packs 32 pairs per lane while the matrix pipes wait)
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(unsigned int, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ float bf16_lo(unsigned int w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf16_hi(unsigned int w) { return __uint_as_float(w & 0xffff0000u); }

// ---- MXFP8 element / scale arithmetic (csrc/mxfp8.hip states the format; shared with the conv epilogues that emit it)
__device__ __forceinline__ unsigned mx8_scale_of(float amax) {          // E8M0 byte of a block with this largest magnitude
    const unsigned bits = __float_as_uint(amax);
    const int E = (int)((bits >> 23) & 0xffu) + ((bits & 0x7fffffu) > 0x600000u ? 1 : 0);
    return (unsigned)(E > 8 ? E - 8 : 0);
}
__device__ __forceinline__ float mx8_inv_scale(unsigned s) {            // 2^(127 - s), always a normal number (s <= 247)
    return __uint_as_float((254u - s) << 23);
}
__device__ __forceinline__ float mx8_clamp(float v) {                   // NaN stays NaN (both comparisons are false)
    v = v > 448.f ? 448.f : v;
    return v < -448.f ? -448.f : v;
}
__device__ __forceinline__ unsigned mx8_pack4(float a, float b, float c, float d) {
    int w = __builtin_amdgcn_cvt_pk_fp8_f32(mx8_clamp(a), mx8_clamp(b), 0, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(mx8_clamp(c), mx8_clamp(d), w, true);
    return (unsigned)w;
}

// MXFP8 copy of 8 consecutive channels held as bf16 pairs (what a lane of the 8-lanes-per-pixel elementwise kernels stores):
// lanes l ^ 1, l ^ 2 hold the rest of the 32-channel block; q points at this lane's 8 bytes, s at the block's scale byte
// (written by the block's first lane).  Bit for bit rgbd_quantize_mxfp8 of the stored tensor.
__device__ __forceinline__ void mx8_emit8(const u32x4& v, unsigned char* q, unsigned char* s, bool block_leader) {
    float f[8];
#pragma unroll
    for (int k = 0; k < 4; ++k) { f[2 * k] = bf16_lo(v[k]); f[2 * k + 1] = bf16_hi(v[k]); }
    float amax = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) amax = fmaxf(amax, fabsf(f[k]));
    amax = fmaxf(amax, __shfl_xor(amax, 1));
    amax = fmaxf(amax, __shfl_xor(amax, 2));
    const unsigned sc = mx8_scale_of(amax);
    const float inv = mx8_inv_scale(sc);
    const u32x2 out = {mx8_pack4(f[0] * inv, f[1] * inv, f[2] * inv, f[3] * inv),
                       mx8_pack4(f[4] * inv, f[5] * inv, f[6] * inv, f[7] * inv)};
    *reinterpret_cast<u32x2*>(q) = out;
    if (block_leader) *s = (unsigned char)sc;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

static inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }
