// MXFP8 operands for the block-scaled fp8 MFMA of gfx950 (v_mfma_scale_f32_16x16x128_f8f6f4, BASELINE configuration 5:
// the 256x256 networks, the blocks the reference keeps commented out at net.py:181-183,192-194).
//
// Format (OCP microscaling, "MXFP8 E4M3"): along the REDUCTION index of the convolution (the channels of an NHWC
// activation tensor; Cin of the fprop weight image, Cout of the dgrad image) every 32 consecutive elements share one
// E8M0 scale byte, value = e4m3(q) * 2^(s - 127).  For a block with largest magnitude amax (biased fp32 exponent E):
//     s = max(E - 8 + (mantissa(amax) > 1.75), 0)     (8 = exponent of the largest e4m3 binade, 256..448 = 1.75 * 2^8)
//     q = e4m3_rne(clamp(x * 2^(127 - s), -448, 448))
// so amax lands in (224, 448]: the block's largest element never saturates (the MX specification's plain floor(log2 amax)
// rule would clip the (448, 512) part of the top binade by up to 12.5 %; the clamp stays for NaN / Inf hygiene).  Everything
// is local to the block: no amax history, no per-tensor state, bit-reproducible, and restated exactly in oracle/mxfp8.py.
// Hardware facts this relies on, measured with scripts/hw/mfma_f8_probe.hip (profiles/r04/mfma_f8_probe.txt):
//   * v_cvt_pk_fp8_f32 rounds to nearest even and returns NaN (0x7f) above 464 -- it does NOT saturate, hence the clamp;
//   * the instruction's K index of byte j of lane l's 32-byte operand is 16 (l >> 4) + j for j < 16 and
//     64 + 16 (l >> 4) + (j - 16) above, and the scale of K block b (32 wide) is the byte lane (row, l >> 4 = b) supplies:
//     with the operand read as the two 16-byte LDS chunks (l >> 4) and 4 + (l >> 4) of a 128-byte row, memory order IS K
//     order and the blocks are the natural contiguous ones.
#include "common.h"

namespace {

// x (rows, C) bf16 -> q (rows, C) e4m3, scales (rows, C / 32) e8m0.  C % 128 == 0.  A thread owns 8 consecutive
// elements (one 16-byte load, one 8-byte store), four neighbouring lanes one block, sixteen one scale dword.
__global__ __launch_bounds__(256) void quantize_mx8_kernel(const unsigned short* __restrict__ x, unsigned char* __restrict__ q,
                                                           unsigned char* __restrict__ scales, long n8) {
    const int lane = threadIdx.x & 63;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {    // n8 % 16 == 0: whole 16-lane groups
        const u32x4 v = *reinterpret_cast<const u32x4*>(x + i * 8);
        float f[8];
#pragma unroll
        for (int k = 0; k < 4; ++k) { f[2 * k] = bf16_lo(v[k]); f[2 * k + 1] = bf16_hi(v[k]); }
        float amax = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) amax = fmaxf(amax, fabsf(f[k]));
        amax = fmaxf(amax, __shfl_xor(amax, 1));
        amax = fmaxf(amax, __shfl_xor(amax, 2));
        const unsigned s = mx8_scale_of(amax);
        const float inv = mx8_inv_scale(s);
        u32x2 out = {mx8_pack4(f[0] * inv, f[1] * inv, f[2] * inv, f[3] * inv),
                     mx8_pack4(f[4] * inv, f[5] * inv, f[6] * inv, f[7] * inv)};
        *reinterpret_cast<u32x2*>(q + i * 8) = out;
        const int g0 = lane & ~15;
        const unsigned sd = (unsigned)__shfl((int)s, g0) | ((unsigned)__shfl((int)s, g0 + 4) << 8) |
                            ((unsigned)__shfl((int)s, g0 + 8) << 16) | ((unsigned)__shfl((int)s, g0 + 12) << 24);
        if ((lane & 15) == 0) *reinterpret_cast<unsigned*>(scales + (i >> 4) * 4) = sd;
    }
}

// Master weights (cout, cin, 3, 3) fp32 -> both MXFP8 images of a 3x3 convolution, `scale` (inv_c) folded in first:
//   fprop image [tap][cout][cin] e4m3 + [tap][cout][cin/32] e8m0     (blocks along cin)
//   dgrad image [8 - tap][cin][cout] e4m3 + [8 - tap][cin][cout/32]  (blocks along cout)
// A workgroup takes a 32 x 32 (co, ci) tile with all nine taps through LDS (the master is read in 1152-byte runs), then
// every thread quantises whole blocks: 288 along ci, 288 along co.
constexpr int MXP_T = 32;
__global__ __launch_bounds__(256) void pack_weights_mx8_multi_kernel(const rgbd_pack_mx8_desc* __restrict__ descs, int n) {
    __shared__ float tile[MXP_T][MXP_T * 9 + 1];          // [co][ci * 9 + tap], odd pitch
    int d = 0;
    for (int i = 1; i < n; ++i)
        if (descs[i].block_begin <= (int)blockIdx.x) d = i;
    const rgbd_pack_mx8_desc D = descs[d];
    const int nblk = (d + 1 < n ? descs[d + 1].block_begin : (int)gridDim.x) - D.block_begin;
    const int tiles_ci = D.cin / MXP_T, tiles = tiles_ci * (D.cout / MXP_T);
    unsigned char* fq = (unsigned char*)D.wf_q;
    unsigned char* fs = (unsigned char*)D.wf_s;
    unsigned char* dq = (unsigned char*)D.wd_q;
    unsigned char* ds = (unsigned char*)D.wd_s;
    for (int t = (int)blockIdx.x - D.block_begin; t < tiles; t += nblk) {
        const int co0 = (t / tiles_ci) * MXP_T, ci0 = (t % tiles_ci) * MXP_T;
        for (int e = threadIdx.x; e < MXP_T * MXP_T * 9; e += 256) {
            const int r = e / (MXP_T * 9), c = e - r * (MXP_T * 9);
            tile[r][c] = D.w[((long)(co0 + r) * D.cin + ci0) * 9 + c] * D.scale;
        }
        __syncthreads();
        for (int e = threadIdx.x; e < 2 * 9 * MXP_T; e += 256) {
            const bool along_ci = e < 9 * MXP_T;
            const int idx = along_ci ? e : e - 9 * MXP_T;
            const int tap = idx / MXP_T, r = idx - tap * MXP_T;        // r: the co (fprop image) / ci (dgrad image) of this block
            if (along_ci ? fq == nullptr : dq == nullptr) continue;
            float v[32];
            float amax = 0.f;
#pragma unroll
            for (int k = 0; k < 32; ++k) {
                v[k] = along_ci ? tile[r][k * 9 + tap] : tile[k][r * 9 + tap];
                amax = fmaxf(amax, fabsf(v[k]));
            }
            const unsigned s = mx8_scale_of(amax);
            const float inv = mx8_inv_scale(s);
            unsigned w8[8];
#pragma unroll
            for (int k = 0; k < 8; ++k)
                w8[k] = mx8_pack4(v[4 * k] * inv, v[4 * k + 1] * inv, v[4 * k + 2] * inv, v[4 * k + 3] * inv);
            unsigned char* qdst;
            unsigned char* sdst;
            if (along_ci) {
                const long row = (long)tap * D.cout + co0 + r;
                qdst = fq + row * D.cin + ci0;
                sdst = fs + row * (D.cin / 32) + ci0 / 32;
            } else {
                const long row = (long)(8 - tap) * D.cin + ci0 + r;
                qdst = dq + row * D.cout + co0;
                sdst = ds + row * (D.cout / 32) + co0 / 32;
            }
            *reinterpret_cast<u32x4*>(qdst) = u32x4{w8[0], w8[1], w8[2], w8[3]};
            *reinterpret_cast<u32x4*>(qdst + 16) = u32x4{w8[4], w8[5], w8[6], w8[7]};
            *sdst = (unsigned char)s;
        }
        __syncthreads();
    }
}

}  // namespace

extern "C" int rgbd_quantize_mxfp8(const void* x, void* q, void* scales, int64_t rows, int C, void* stream) {
    RGBD_REQUIRE(x && q && scales, "rgbd_quantize_mxfp8: null pointer");
    RGBD_REQUIRE(rows > 0 && C > 0 && C % 128 == 0, "rgbd_quantize_mxfp8: need rows > 0 and C a multiple of 128 (C=%d)", C);
    const long n8 = rows * (long)C / 8;
    const long blocks = (n8 + 255) / 256;
    quantize_mx8_kernel<<<(unsigned)(blocks < 8192 ? blocks : 8192), 256, 0, (hipStream_t)stream>>>(
        (const unsigned short*)x, (unsigned char*)q, (unsigned char*)scales, n8);
    RGBD_CHECK_LAUNCH("quantize_mx8_kernel");
    return 0;
}

extern "C" int rgbd_pack_weights_mxfp8_multi(const rgbd_pack_mx8_desc* descs_device, int n, int total_blocks, void* stream) {
    RGBD_REQUIRE(descs_device && n > 0 && total_blocks > 0, "rgbd_pack_weights_mxfp8_multi: bad arguments");
    pack_weights_mx8_multi_kernel<<<total_blocks, 256, 0, (hipStream_t)stream>>>(descs_device, n);
    RGBD_CHECK_LAUNCH("pack_weights_mx8_multi_kernel");
    return 0;
}
