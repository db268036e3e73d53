// Implicit-GEMM convolution engine for gfx950 (bf16 MFMA, fp32 accumulate), NHWC activations.
//
// fprop / dgrad : D[co][pixel] = sum_{tap,ci} Wp[tap][co][ci] * X[pixel + tap][ci]
//   A operand = packed weights (K-contiguous rows), B operand = gathered pixels (channels contiguous in
//   NHWC), so both MFMA operands are plain 16-byte LDS reads and the accumulator holds 4 consecutive output
//   channels per lane (8-byte NHWC stores).  The im2col matrix is never materialised: each K step gathers one
//   filter tap x 64 input channels for 128 output pixels straight from the activation tensor; nearest-2x
//   upsampling (rescale.py:4-5) is folded into that gather's address arithmetic.
//   Roofline: MFMA (bf16 dense ~2.5 PFLOP/s) for Cin,Cout >= 128; the 64-channel 128x128 layers sit at
//   ~290-380 FLOP/B, i.e. at the HBM/MFMA ridge, so they are HBM-bound unless epilogues stay fused.
//   Kernels: conv_fprop_kernel (generic gather, any K / pad, split-K for the 4x4 / 8x8 layers); conv3x3_sp_kernel (3x3
//   pad-1 on images >= 16x16, the hot one: halo patch per 64-channel slice, LDS-DMA staging, register-pipelined
//   fragments); conv3x3_patch_kernel (round 1's register-staged version of the same tiling, kept as the A/B reference).
//
// wgrad : dW[tap][co][ci] = sum_pixel dY[pixel][co] * X[pixel + tap][ci]
//   The reduction index is the pixel, which is the *strided* index of both NHWC operands.  gfx950's
//   ds_read_b64_tr_b16 transposes while reading LDS, so both operands are staged pixel-major exactly as they
//   sit in HBM and read K(pixel)-contiguous for v_mfma_f32_32x32x16_bf16.  One workgroup keeps all 9 taps of
//   a 64x64 (co,ci) tile in registers (144 accumulator VGPRs per lane), stages an 8x16-pixel patch of dY and the
//   10x18 halo patch of X once, and sweeps its share of patches; partial sums leave as plain fp32 stores into
//   one slab per workgroup (two 128-byte row segments per wave instruction) and a second kernel sums the slabs.
//   Kernels: conv_wgrad9_body (3x3 on 8x16 patches, the hot one: every wave runs all nine taps over a three-row register
//   window of X fragments, operands by LDS-DMA), conv_wgrad_body (1x1, small images, and the tap-split A/B reference).
#include "common.h"

#include <stdlib.h>
#include <atomic>
#include <type_traits>

namespace {

struct ConvArgs {
    const unsigned short* x;
    const unsigned short* wp;
    const float* bias;
    const unsigned short* resid;
    unsigned short* y;
    int B, Hin, Win, Cin, Cout, KH, KW, pad, ups, Hout, Wout, lrelu_ch;
    float slope;
    long M;
    int x_bytes, w_bytes;
    int ptiles, wgs_per_ntile;      // patch kernel: pixel tiles per output-channel tile, persistent workgroups per N tile
    int pool_sum;                   // patch kernel: write 2x2 SUMS of the result, y is (B,Hout/2,Wout/2,Cout)
    unsigned short* ypool;          // patch kernel: also write the 2x2 AVERAGE of y to (B,Hout/2,Wout/2,Cout) (or null)
    int ksplit;                     // gather kernel: workgroups per output tile along K (1 = no split)
    float* partial;                 // gather kernel, ksplit > 1: [ksplit][M][Cout] fp32 partial sums
    // pipelined 3x3 kernel, MASKED variant: the result times lrelu'(.) of a GIVEN activation output of y's shape (the
    // activation gradient of the layer in front, taken where the input gradient is produced), and its weighted column
    // sums colsum[co] += sum_b row_scale[b] sum_pixels y (that layer's bias gradient; row_scale null = 1)
    const unsigned short* mask_y;
    float* colsum;
    const float* row_scale;
    unsigned short* y2;             // MASKED, optional second output: y + row_scale2[b] * mask_y (what was stored, plus the
    const float* row_scale2;        //   per-sample multiple of the activation tile the epilogue holds anyway)
    // pipelined 3x3 kernel, STATS variant: stats[b][co][0..1] += (sum y, sum y^2) over the image's pixels, as 64-bit
    // integers in units of 2^-32 (integer adds commute: the result does not depend on which workgroup adds first)
    long long* stats;
    // pipelined 3x3 kernel, MX variant (block-scaled fp8 MFMA): x and wp are e4m3 bytes, xs / ws their E8M0 block scales
    // ((B,Hin,Win,Cin/32) and [9][Cout][Cin/32]: one byte per 32 consecutive input channels, csrc/mxfp8.hip)
    const unsigned char* xs;
    const unsigned char* ws;
    int xs_bytes, ws_bytes;
    // pipelined 3x3 kernel, any form: MXFP8 copies of what the epilogue stores (blocks of 32 along Cout: the next convolution's
    // reduction channels), so that the consumer needs no quantiser pass -- of y (yq / ys) and of the pooled output (ypq / yps)
    unsigned char* yq;
    unsigned char* ys;
    unsigned char* ypq;
    unsigned char* yps;
};

__device__ __forceinline__ u32x4 ldg16(const unsigned short* p) { return *reinterpret_cast<const u32x4*>(p); }

template <int BN>
__global__ __launch_bounds__(256, 2) void conv_fprop_kernel(ConvArgs a) {
    constexpr int BM = 128;                  // output pixels per workgroup
    constexpr int P_BYTES = BM * 128;        // 128 rows x 64 bf16
    constexpr int W_BYTES = BN * 128;
    constexpr int STAGE = P_BYTES + W_BYTES;
    constexpr int WAVES_CO = BN / 64;        // 64 output channels per wave along N
    constexpr int WAVES_PX = 4 / WAVES_CO;
    constexpr int PX_PER_WAVE = BM / WAVES_PX;
    constexpr int TPX = PX_PER_WAVE / 16;
    constexpr int WROWS = BN / 32;           // weight rows staged per thread
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int n_tiles = a.Cout / BN;
    // XCD-aware tile order (cdna_hip_programming.md T1, bijective form): blocks are dealt round-robin over the 8
    // XCDs, so give each XCD a contiguous run of tiles -- neighbours share pixel halos and weight panels in its L2.
    unsigned bid = blockIdx.x;
    {
        const unsigned nwg = gridDim.x, xcd = bid & 7u, q8 = nwg >> 3, r8 = nwg & 7u;
        bid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    }
    // split-K: the tiny layers (4x4 .. 16x16 images) have a handful of output tiles and a long K loop; one
    // workgroup would pull its whole ~1 MB operand panel through a single CU's L1.  ksplit workgroups share a tile,
    // each walks a slice of the K steps and leaves fp32 partial sums for conv_splitk_finish_kernel.
    const int sidx = (int)(bid % (unsigned)a.ksplit);
    bid /= (unsigned)a.ksplit;
    const int nt = bid % n_tiles;
    const long mt = bid / n_tiles;
    const long m0 = mt * BM;
    const int n0 = nt * BN;
    const int wave_co = (wid / WAVES_PX) * 64;
    const int wave_px = (wid % WAVES_PX) * PX_PER_WAVE;

    // ---- gather bookkeeping: this thread stages rows prow, prow+32, .. at 16-byte chunk `chunk`
    const int chunk = tid & 7;
    const int prow = tid >> 3;
    const int swz = (chunk ^ (prow & 7)) << 4;   // row & 7 == prow & 7 for every row this thread touches
    const int HWo = a.Hout * a.Wout;
    const int Hup = a.ups ? 2 * a.Hin : a.Hin;
    const int Wup = a.ups ? 2 * a.Win : a.Win;
    int pb[4], ph[4], pw[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const long m = m0 + prow + 32 * i;
        if (m < a.M) {
            const int b = (int)(m / HWo);
            const int r = (int)(m - (long)b * HWo);
            const int ho = r / a.Wout;
            pb[i] = b; ph[i] = ho; pw[i] = r - ho * a.Wout;
        } else {
            pb[i] = 0; ph[i] = -0x40000000; pw[i] = 0;
        }
    }
    const int nkc = a.Cin >> 6;
    const int nk_all = a.KH * a.KW * nkc;
    const int kbeg = (int)((long)sidx * nk_all / a.ksplit);
    const int nk = (int)((long)(sidx + 1) * nk_all / a.ksplit);     // this workgroup's K steps: [kbeg, nk)

    // Register prefetch, three rotating sets: tiles k+1, k+2, k+3 are in registers / in flight while tile k is
    // multiplied out of LDS (two LDS buffers, one barrier per K step).  The layers that use this kernel are the tiny
    // 4x4 / 8x8 ones, where a handful of workgroups walk a long K loop alone on their CU: latency, not bandwidth.
    // Offsets are 32-bit (host checks sizes).
    u32x4 regP[3][4], regW[3][WROWS];
    const auto xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(a.x), 0, a.x_bytes, 0x00020000);
    const auto wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(a.wp), 0, a.w_bytes, 0x00020000);
    auto load_tile = [&](int kt, u32x4 (&rP)[4], u32x4 (&rW)[WROWS]) {
        // K steps past the end load zeros (out-of-range offsets): the loop below then needs no early exit, every
        // step is unconditional and hipcc keeps counted vmcnt waits around the whole unrolled body
        const unsigned dead = kt >= nk ? 0x80000000u : 0u;
        const int tap = kt / nkc;
        const int c0 = (kt - tap * nkc) << 6;
        const int kh = tap / a.KW;
        const int dh = kh - a.pad, dw = (tap - kh * a.KW) - a.pad;
#pragma unroll
        for (int i = 0; i < WROWS; ++i) {
            const unsigned off = (unsigned)((tap * a.Cout + n0 + prow + 32 * i) * a.Cin + c0 + chunk * 8) * 2u;
            rW[i] = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, off | dead, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int hi = ph[i] + dh, wi = pw[i] + dw;
            const bool ok = (unsigned)hi < (unsigned)Hup && (unsigned)wi < (unsigned)Wup;
            const int hs = a.ups ? (hi >> 1) : hi, ws = a.ups ? (wi >> 1) : wi;
            // range-checked buffer load: out-of-frame lanes read zeros, no branch and no select after the load
            const unsigned off = ok ? (unsigned)(((pb[i] * a.Hin + hs) * a.Win + ws) * a.Cin + c0 + chunk * 8) * 2u
                                    : 0x80000000u;
            rP[i] = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, off | dead, 0, 0);
        }
    };
    auto store_tile = [&](int buf, const u32x4 (&rP)[4], const u32x4 (&rW)[WROWS]) {
        unsigned char* base = smem + buf * STAGE;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            *reinterpret_cast<u32x4*>(base + (prow + 32 * i) * 128 + swz) = rP[i];
#pragma unroll
        for (int i = 0; i < WROWS; ++i)
            *reinterpret_cast<u32x4*>(base + P_BYTES + (prow + 32 * i) * 128 + swz) = rW[i];
    };

    f32x4 acc[4][TPX];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < TPX; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int r16 = lane & 15, q = lane >> 4;
    auto compute = [&](int buf) {
        const unsigned char* pbase = smem + buf * STAGE;
        const unsigned char* wbase = pbase + P_BYTES;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int off = ((4 * s + q) ^ (r16 & 7)) << 4;
            bf16x8 af[4], bfr[TPX];
#pragma unroll
            for (int t = 0; t < 4; ++t)
                af[t] = *reinterpret_cast<const bf16x8*>(wbase + (wave_co + t * 16 + r16) * 128 + off);
#pragma unroll
            for (int t = 0; t < TPX; ++t)
                bfr[t] = *reinterpret_cast<const bf16x8*>(pbase + (wave_px + t * 16 + r16) * 128 + off);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < TPX; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
    };

    load_tile(kbeg + 0, regP[0], regW[0]);
    load_tile(kbeg + 1, regP[1], regW[1]);
    load_tile(kbeg + 2, regP[2], regW[2]);
    store_tile(0, regP[0], regW[0]);
    __syncthreads();
    for (int kt = 0; kt < nk - kbeg; kt += 3) {
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int k = kt + r;                                                // kbeg + k >= nk: multiplies zero tiles
            compute(k & 1);
            store_tile((k + 1) & 1, regP[(r + 1) % 3], regW[(r + 1) % 3]);     // tile k+1, requested two steps ago
            load_tile(kbeg + k + 3, regP[r], regW[r]);                           // set r (tile k) is in LDS already
            __syncthreads();
        }
    }

    if (a.partial) {     // split-K: raw fp32 partial sums, epilogue in conv_splitk_finish_kernel
        float* pp = a.partial + (long)sidx * a.M * a.Cout;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int co = n0 + wave_co + i * 16 + 4 * q;
#pragma unroll
            for (int j = 0; j < TPX; ++j) {
                const long m = m0 + wave_px + j * 16 + r16;
                if (m < a.M) *reinterpret_cast<f32x4*>(pp + m * a.Cout + co) = acc[i][j];
            }
        }
        return;
    }

    // ---- epilogue: bias -> residual -> leaky ReLU (first lrelu_ch channels) -> bf16 NHWC
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int co = n0 + wave_co + i * 16 + 4 * q;
        float bv[4] = {0.f, 0.f, 0.f, 0.f};
        if (a.bias) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(a.bias + co);
            bv[0] = t[0]; bv[1] = t[1]; bv[2] = t[2]; bv[3] = t[3];
        }
        const bool act = co < a.lrelu_ch;   // lrelu_ch is a multiple of 4 by construction (channel groups of 64)
#pragma unroll
        for (int j = 0; j < TPX; ++j) {
            const long m = m0 + wave_px + j * 16 + r16;
            if (m < a.M) {
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = acc[i][j][r] + bv[r];
                const long o = m * a.Cout + co;
                if (a.resid) {
                    const u32x2 rr = *reinterpret_cast<const u32x2*>(a.resid + o);
                    v[0] += bf16_lo(rr[0]); v[1] += bf16_hi(rr[0]);
                    v[2] += bf16_lo(rr[1]); v[3] += bf16_hi(rr[1]);
                }
                if (act) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = v[r] > 0.f ? v[r] : v[r] * a.slope;
                }
                u32x2 out = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
                *reinterpret_cast<u32x2*>(a.y + o) = out;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ small images (4x4, 8x8)
// 3x3 pad-1 convolutions on 4x4 / 8x8 images (generator blocks 0-1, discriminator blocks 1-0 and their backward passes:
// 25 launches per step, 2.4 / 0.6 GFLOP each).  The gather kernel above walks a long K loop in which every step waits
// for global loads issued three steps earlier (~0.9 us per step): 14-19 us per launch at 4-13 % of the MFMA peak.  These
// layers are one load round trip long if nothing depends on anything: a workgroup takes a 128-pixel tile (whole images:
// 2 of 8x8, 8 of 4x4) x 64 output channels x ONE 64-channel input slice, requests its whole operand set at once -- the
// images with their zero halo (25-37 KB) and the nine 64x64 weight tiles (72 KB) -- and after a single wait runs the nine
// taps out of LDS (144 MFMAs per wave).  Grid = (M / 128) x (Cout / 64) x (Cin / 64) workgroups (256 for 8x8 at B = 32),
// fp32 partial sums per input slice into the split-K scratch, epilogue by conv_splitk_finish_kernel as before.
template <int S>   // image side: 4 or 8
__global__ __launch_bounds__(256) void conv3x3_small_kernel(ConvArgs a) {
    constexpr int HP = S + 2, IM = 128 / (S * S), NH = IM * HP * HP;     // halo side, images per tile, halo pixels per tile
    constexpr int X_BYTES = NH * 128;
    constexpr int XL = (NH * 8 + 255) / 256, WL = 9 * 64 * 8 / 256;      // 16-byte pieces per thread: 7 or 9, and 18
    extern __shared__ __attribute__((aligned(16))) unsigned char ssm[];  // [X_BYTES] halo images, [W_BYTES] nine weight tiles
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int nci = a.Cin >> 6, nco = a.Cout >> 6;
    unsigned bid = blockIdx.x;
    const int slice = (int)(bid % (unsigned)nci);
    bid /= (unsigned)nci;
    const int ct = (int)(bid % (unsigned)nco);
    const int mt = (int)(bid / (unsigned)nco);
    const int ci0 = slice * 64, co0 = ct * 64, img0 = mt * IM;
    const auto xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(a.x), 0, a.x_bytes, 0x00020000);
    const auto wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(a.wp), 0, a.w_bytes, 0x00020000);

    // ---- everything this workgroup will ever read, requested at once (range-checked loads: halo pixels read zeros)
    u32x4 rx[XL], rw[WL];
#pragma unroll
    for (int i = 0; i < XL; ++i) {
        const int pc = tid + 256 * i, row = pc >> 3, chunk = pc & 7;
        const int im = row / (HP * HP), rem = row - im * (HP * HP);
        const int hy = rem / HP, hx = rem - hy * HP;
        const bool ok = row < NH && hy >= 1 && hy <= S && hx >= 1 && hx <= S;
        const unsigned off = (unsigned)(((((img0 + im) * S + hy - 1) * S + hx - 1) * a.Cin + ci0 + chunk * 8) * 2);
        rx[i] = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, ok ? off : 0x80000000u, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < WL; ++i) {
        const int pc = tid + 256 * i, row = pc >> 3, chunk = pc & 7;       // row = tap * 64 + output channel
        const int tap = row >> 6, co = row & 63;
        rw[i] = __builtin_amdgcn_raw_buffer_load_b128(
            wrsrc, (unsigned)((((tap * a.Cout + co0 + co) * a.Cin) + ci0 + chunk * 8) * 2), 0, 0);
    }
#pragma unroll
    for (int i = 0; i < XL; ++i) {
        const int pc = tid + 256 * i, row = pc >> 3, chunk = pc & 7;
        const int hx = row % HP;
        if (row < NH) *reinterpret_cast<u32x4*>(ssm + row * 128 + ((chunk ^ (hx & 7)) << 4)) = rx[i];
    }
#pragma unroll
    for (int i = 0; i < WL; ++i) {
        const int pc = tid + 256 * i, row = pc >> 3, chunk = pc & 7;
        *reinterpret_cast<u32x4*>(ssm + X_BYTES + row * 128 + ((chunk ^ (row & 7)) << 4)) = rw[i];
    }
    __syncthreads();

    // ---- wave w: pixels 32 w .. 32 w + 31 of the tile x 64 output channels, nine taps x two k halves
    const int r16 = lane & 15, q = lane >> 4;
    int prow[2], pcol[2];                         // halo row index of tap (0,0) and image column of this lane's two pixels
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int p = 32 * wid + 16 * j + r16;
        const int im = p / (S * S), y = (p / S) % S, x = p % S;
        prow[j] = (im * HP + y) * HP + x;
        pcol[j] = x;
    }
    f32x4 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const int kh = tap / 3, kw = tap - 3 * kh;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            bf16x8 af[4], bfr[2];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = tap * 64 + 16 * i + r16;
                af[i] = *reinterpret_cast<const bf16x8*>(ssm + X_BYTES + row * 128 + (((4 * h + q) ^ (row & 7)) << 4));
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int row = prow[j] + kh * HP + kw;
                bfr[j] = *reinterpret_cast<const bf16x8*>(ssm + row * 128 + (((4 * h + q) ^ ((pcol[j] + kw) & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
    }
    float* pp = a.partial + (long)slice * a.M * a.Cout;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int co = co0 + 16 * i + 4 * q;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const long m = (long)mt * 128 + 32 * wid + 16 * j + r16;
            *reinterpret_cast<f32x4*>(pp + m * a.Cout + co) = acc[i][j];
        }
    }
}

// Epilogue of the split-K path: sum the K-slice partials, then bias -> residual -> leaky ReLU -> bf16 NHWC.
__global__ __launch_bounds__(256) void conv_splitk_finish_kernel(const float* __restrict__ partial, int S, long M,
                                                                 int Cout, const float* __restrict__ bias,
                                                                 const unsigned short* __restrict__ resid, int lrelu_ch,
                                                                 float slope, unsigned short* __restrict__ y) {
    const long total = M * Cout;
    for (long e = ((long)blockIdx.x * 256 + threadIdx.x) * 4; e < total; e += (long)gridDim.x * 1024) {
        const int co = (int)(e % Cout);
        f32x4 v = *reinterpret_cast<const f32x4*>(partial + e);
        for (int s2 = 1; s2 < S; ++s2) v += *reinterpret_cast<const f32x4*>(partial + (long)s2 * total + e);
        if (bias) v += *reinterpret_cast<const f32x4*>(bias + co);
        if (resid) {
            const u32x2 rr = *reinterpret_cast<const u32x2*>(resid + e);
            v[0] += bf16_lo(rr[0]); v[1] += bf16_hi(rr[0]);
            v[2] += bf16_lo(rr[1]); v[3] += bf16_hi(rr[1]);
        }
        if (co < lrelu_ch) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = v[r] > 0.f ? v[r] : v[r] * slope;
        }
        u32x2 out = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
        *reinterpret_cast<u32x2*>(y + e) = out;
    }
}

#ifdef RGBD_DEBUG_BUILD     // A/B reference of round 1: `python -m rgbd_gan_amd.build --debug` only (rgbd_debug.h)
// ------------------------------------------------------------------------------------------------ 3x3 patch kernel
// The hot fprop/dgrad kernel for 3x3 pad-1 convolutions on images of 16x16 and larger.
//
// One 512-thread workgroup (8 waves, one per CU) owns a 16x16 patch of output pixels x BN output channels.
// Per 64-channel slice of the input it stages the 18x18 halo patch ONCE (10x10 when the nearest-2x upsample
// is folded in) and runs all nine filter taps out of LDS: the im2col expansion (9x re-read of every input
// pixel) happens in LDS addressing, not in L2/HBM traffic.  Per K step (one tap x 64 channels) only the BN x 64
// weight tile is fetched (L2-resident, three LDS buffers, register prefetch two steps ahead).  Global->LDS
// traffic per K step drops from 32 KB per 128 pixels (gather kernel above) to ~20 KB per 256 pixels.
//   MFMA operands: A = weight rows (K contiguous), B = halo-patch pixel rows.  A lane's LDS address is
//   (per-lane term depending only on the horizontal tap) + (compile-time row offset), so the unrolled tap loop
//   needs six address VGPRs; the 16-byte chunk index is XORed with (halo column & 7), which keeps ds_read_b128
//   conflict-free for every tap shift (the halo width is even, so row parity == column parity).
template <int BN, bool UPS>
__global__ __launch_bounds__(512, 2) void conv3x3_patch_kernel(ConvArgs a) {
    constexpr int HPW = UPS ? 10 : 18;            // halo patch width (and height)
    constexpr int NROWS = HPW * HPW;
    constexpr int P_BYTES = 18 * 18 * 128;
    constexpr int W_BYTES = BN * 128;
    constexpr int WAVES_CO = BN / 64;
    constexpr int WAVES_PX = 8 / WAVES_CO;
    constexpr int PX_PER_WAVE = 256 / WAVES_PX;   // 64 or 32 pixels = 4 or 2 patch rows
    constexpr int TPX = PX_PER_WAVE / 16;
    constexpr int WP = BN / 64;                   // weight pieces (16 B) staged per thread per K step
    constexpr int PP = (NROWS * 8 + 511) / 512;   // halo pieces per thread per channel slice (6 or 2)
    extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];
    unsigned char* const patch_lds = dsm;                       // [2][P_BYTES]
    unsigned char* const w_lds = dsm + 2 * P_BYTES;             // [3][W_BYTES]

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    // ---- persistent tile loop: this workgroup owns output-channel tile `nt` and the pixel tiles [pt_begin, pt_end).
    //      Consecutive logical workgroups get consecutive tile ranges and, through the XCD remap, share an L2
    //      (neighbouring patches share halos; all of them share the weight tiles).
    unsigned bid = blockIdx.x;
    {
        const unsigned nwg = gridDim.x, xcd = bid & 7u, q8 = nwg >> 3, r8 = nwg & 7u;
        bid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    }
    const int nt = bid / a.wgs_per_ntile;
    const int slot = bid - nt * a.wgs_per_ntile;
    const int pt_begin = (int)((long)slot * a.ptiles / a.wgs_per_ntile);
    const int pt_end = (int)((long)(slot + 1) * a.ptiles / a.wgs_per_ntile);
    const int tiles_x = a.Wout >> 4, tiles_per_img = tiles_x * (a.Hout >> 4);
    const int n0 = nt * BN;
    const int wave_co = (wid / WAVES_PX) * 64;
    const int wave_py = (wid % WAVES_PX) * TPX;                  // first patch row of this wave

    const auto xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(a.x), 0, a.x_bytes, 0x00020000);
    const auto wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(a.wp), 0, a.w_bytes, 0x00020000);

    const int nc = a.Cin >> 6;
    const int g_total = (pt_end - pt_begin) * nc;               // (tile, channel slice) pairs of this workgroup
    if (g_total <= 0) return;

    // ---- halo staging: LDS destinations are fixed per thread; sources depend on the tile and are recomputed per
    //      slice (once per nine K steps).  Channel-slice / tap offsets are wave-uniform scalars.
    int pdst[PP], prow_hy[PP], prow_hx[PP];
#pragma unroll
    for (int i = 0; i < PP; ++i) {
        const int p = tid + 512 * i;
        const int row = p >> 3, ch8 = p & 7;
        const int hy = row / HPW, hx = row - hy * HPW;
        prow_hy[i] = row < NROWS ? hy : -100000;
        prow_hx[i] = hx;
        pdst[i] = row < NROWS ? row * 128 + ((ch8 ^ (hx & 7)) << 4) : -1;
    }
    const int pch8 = (tid & 7) * 8;
    const int wrow = tid >> 3, wch8 = tid & 7;
    const int wdst = wrow * 128 + ((wch8 ^ (wrow & 7)) << 4);
    // LDS row (64*w + 16*t + m) of the weight tile holds output channel 64*w + 16*(m>>2) + 4*t + (m&3): with the
    // 16x16 MFMA accumulator layout (lane q = lane>>4 holds rows 4q..4q+3 of each 16-row tile t) every lane then
    // owns 16 CONSECUTIVE output channels of its pixel -> 16-byte NHWC stores, four lanes cover a 128-byte line
    const int wm = wrow & 15, wt = (wrow >> 4) & 3;
    const int wperm = (wrow & ~63) + 16 * (wm >> 2) + 4 * wt + (wm & 3);
    const unsigned wsrc0 = (unsigned)(((n0 + wperm) * a.Cin + wch8 * 8) * 2);
    const int tap_stride = a.Cout * a.Cin * 2;
    const int row64_stride = 64 * a.Cin * 2;

    u32x4 Pr[PP], Wr[3][WP];
    auto tile_origin = [&](int pt, int& b, int& y0, int& x0) {
        b = pt / tiles_per_img;
        const int rem = pt - b * tiles_per_img;
        const int ty = rem / tiles_x;
        y0 = ty << 4;
        x0 = (rem - ty * tiles_x) << 4;
    };
    auto load_patch = [&](int g) {                              // g: global (tile, slice) index, clamped by caller
        const int ti = g / nc, c = g - ti * nc;
        int b, y0, x0;
        tile_origin(pt_begin + ti, b, y0, x0);
        const int hy0 = UPS ? (y0 >> 1) - 1 : y0 - 1, hx0 = UPS ? (x0 >> 1) - 1 : x0 - 1;
#pragma unroll
        for (int i = 0; i < PP; ++i) {
            const int yy = hy0 + prow_hy[i], xx = hx0 + prow_hx[i];
            const bool ok = (unsigned)yy < (unsigned)a.Hin && (unsigned)xx < (unsigned)a.Win;
            const unsigned off = ok ? (unsigned)((((b * a.Hin + yy) * a.Win + xx) * a.Cin + pch8) * 2) : 0x80000000u;
            Pr[i] = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, off, c * 128, 0);
        }
    };
    auto store_patch = [&](int buf) {
#pragma unroll
        for (int i = 0; i < PP; ++i)
            if (pdst[i] >= 0) *reinterpret_cast<u32x4*>(patch_lds + buf * P_BYTES + pdst[i]) = Pr[i];
    };
    auto load_w = [&](int c, int t, u32x4 (&R)[WP]) {
#pragma unroll
        for (int i = 0; i < WP; ++i)
            R[i] = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wsrc0, t * tap_stride + i * row64_stride + c * 128, 0);
    };
    auto store_w = [&](int buf, const u32x4 (&R)[WP]) {
#pragma unroll
        for (int i = 0; i < WP; ++i) *reinterpret_cast<u32x4*>(w_lds + buf * W_BYTES + wdst + i * 64 * 128) = R[i];
    };

    f32x4 acc[4][TPX];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < TPX; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int r16 = lane & 15, q = lane >> 4;
    int aoff[2], boff[3][2];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
        aoff[s2] = (wave_co + r16) * 128 + (((4 * s2 + q) ^ (r16 & 7)) << 4);
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const int colx = UPS ? ((r16 + kw - 1) >> 1) + 1 : r16 + kw;
            const int row0 = UPS ? (wave_py >> 1) : wave_py;
            boff[kw][s2] = (row0 * HPW + colx) * 128 + (((4 * s2 + q) ^ (colx & 7)) << 4);
        }
    }

    // Pixel operands live in registers across the three VERTICAL taps of a filter column: output row j at vertical tap
    // kh reads halo row j + kh, so the wave's TPX rows need only TPX + 2 distinct halo rows per filter column (fewer
    // behind the folded upsample) -- read once when the column starts (kh == 0), reused by kh = 1, 2.  The tap loop
    // therefore runs column-major (t = 3 kw + kh).  LDS reads per channel slice: 72 (weights) + 6 NR (pixels) instead
    // of 72 + 18 TPX.
    constexpr int NR = UPS ? TPX / 2 + 2 : TPX + 2;
    bf16x8 brow[NR][2];
    auto compute = [&](const unsigned char* pbuf, const unsigned char* wbuf, int kh, int kw) {
        if (kh == 0) {
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int r = 0; r < NR; ++r)
                    brow[r][s2] = *reinterpret_cast<const bf16x8*>(pbuf + boff[kw][s2] + r * HPW * 128);
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            bf16x8 af[4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
                af[i] = *reinterpret_cast<const bf16x8*>(wbuf + aoff[s2] + i * 16 * 128);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < TPX; ++j) {
                    const int rj = UPS ? ((j + kh - 1) >> 1) + 1 : j + kh;   // compile-time after unrolling
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], brow[rj][s2], acc[i][j], 0, 0, 0);
                }
        }
    };

    // bias of the tile's output channels: parked in LDS once (the output-channel tile is fixed for the whole persistent
    // loop) -- a global load inside the per-tile epilogue would wait behind, and so drain, the prefetched next tile, and
    // 16 registers per lane held across the MFMA loop would push the kernel past 224 VGPRs (see the launch bounds)
    float* const bias_lds = reinterpret_cast<float*>(dsm + 2 * P_BYTES + 3 * W_BYTES);   // [BN], visible after the prologue barrier
    if (tid < BN) bias_lds[tid] = a.bias ? a.bias[n0 + tid] : 0.f;

    auto epilogue = [&](int pt) {       // bias -> residual -> leaky ReLU -> bf16 NHWC, then clear the accumulators
        int b, y0, x0;
        tile_origin(pt, b, y0, x0);
        const int co = n0 + wave_co + 16 * q;            // this lane's 16 consecutive output channels
        const bool act = co < a.lrelu_ch;
        if (a.pool_sum) {
            // adjoint of the nearest-2x upsample in front of a generator conv (rescale.py:4-5): the input gradient
            // leaves as 2x2 sums at half resolution.  Rows pair up inside the lane (wave_py is even), columns between
            // lanes r16 and r16^1 (one DPP quad permute per value); even lanes store.
            const int Hp = a.Hout >> 1, Wp = a.Wout >> 1;
#pragma unroll
            for (int j = 0; j < TPX; j += 2) {
                float v[16];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float t = acc[i][j][r] + acc[i][j + 1][r];
                        const int o = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0xB1, 0xF, 0xF, false);
                        v[4 * i + r] = t + __builtin_bit_cast(float, o);
                    }
                if ((r16 & 1) == 0) {
                    const int yy = (y0 + wave_py + j) >> 1, xx = (x0 + r16) >> 1;
                    const long o = (((long)b * Hp + yy) * Wp + xx) * a.Cout + co;
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        u32x4 out = {pack_bf16x2(v[8 * h + 0], v[8 * h + 1]), pack_bf16x2(v[8 * h + 2], v[8 * h + 3]),
                                     pack_bf16x2(v[8 * h + 4], v[8 * h + 5]), pack_bf16x2(v[8 * h + 6], v[8 * h + 7])};
                        *reinterpret_cast<u32x4*>(a.y + o + 8 * h) = out;
                    }
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                    acc[i][j + 1] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
            return;
        }
        float ps[16];                   // ypool: running 2x2 sums of the bf16-rounded outputs of a row pair
#pragma unroll
        for (int j = 0; j < TPX; ++j) {
            const int yy = y0 + wave_py + j, xx = x0 + r16;
            const long o = (((long)b * a.Hout + yy) * a.Wout + xx) * a.Cout + co;
            float v[16];
            if (a.bias) {               // (uniform: input-gradient launches have no bias and skip 16 VALU slots per row)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const f32x4 bq = *reinterpret_cast<const f32x4*>(bias_lds + wave_co + 16 * q + 4 * i);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[4 * i + r] = acc[i][j][r] + bq[r];
                }
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[4 * i + r] = acc[i][j][r];
            }
            if (a.resid) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const u32x4 rr = *reinterpret_cast<const u32x4*>(a.resid + o + 8 * h);
#pragma unroll
                    for (int w2 = 0; w2 < 4; ++w2) {
                        v[8 * h + 2 * w2] += bf16_lo(rr[w2]);
                        v[8 * h + 2 * w2 + 1] += bf16_hi(rr[w2]);
                    }
                }
            }
            if (act) {
#pragma unroll
                for (int k2 = 0; k2 < 16; ++k2) v[k2] = v[k2] > 0.f ? v[k2] : v[k2] * a.slope;
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                u32x4 out = {pack_bf16x2(v[8 * h + 0], v[8 * h + 1]), pack_bf16x2(v[8 * h + 2], v[8 * h + 3]),
                             pack_bf16x2(v[8 * h + 4], v[8 * h + 5]), pack_bf16x2(v[8 * h + 6], v[8 * h + 7])};
                *reinterpret_cast<u32x4*>(a.y + o + 8 * h) = out;
                if (a.ypool) {          // the block's downscale2x (rescale.py:12-13) of what was just stored
#pragma unroll
                    for (int w2 = 0; w2 < 4; ++w2) {
                        const float lo = bf16_lo(out[w2]), hi = bf16_hi(out[w2]);
                        ps[8 * h + 2 * w2] = (j & 1) ? ps[8 * h + 2 * w2] + lo : lo;
                        ps[8 * h + 2 * w2 + 1] = (j & 1) ? ps[8 * h + 2 * w2 + 1] + hi : hi;
                    }
                }
            }
            if (a.ypool && (j & 1)) {
#pragma unroll
                for (int k2 = 0; k2 < 16; ++k2) {
                    const int o2 = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, ps[k2]), 0xB1, 0xF, 0xF, false);
                    ps[k2] = 0.25f * (ps[k2] + __builtin_bit_cast(float, o2));
                }
                if ((r16 & 1) == 0) {
                    const long op = (((long)b * (a.Hout >> 1) + (yy >> 1)) * (a.Wout >> 1) + (xx >> 1)) * a.Cout + co;
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        u32x4 out = {pack_bf16x2(ps[8 * h + 0], ps[8 * h + 1]), pack_bf16x2(ps[8 * h + 2], ps[8 * h + 3]),
                                     pack_bf16x2(ps[8 * h + 4], ps[8 * h + 5]), pack_bf16x2(ps[8 * h + 6], ps[8 * h + 7])};
                        *reinterpret_cast<u32x4*>(a.ypool + op + 8 * h) = out;
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };

    // ---- prologue (once per workgroup; later tiles are prefetched under the previous tile's K steps)
    load_patch(0);
    load_w(0, 0, Wr[0]);          // step t multiplies filter tap (kh, kw) = (t % 3, t / 3): weight image 3 (t % 3) + t / 3
    load_w(0, 3, Wr[1]);
    load_w(0, 6, Wr[2]);
    store_patch(0);
    store_w(0, Wr[0]);
    load_patch(min(1, g_total - 1));
    __syncthreads();
    // ---- main loop over (tile, channel slice) pairs; nine statically unrolled tap steps each.  Every memory
    //      operation in the tap loop is unconditional (indices clamp / wrap) so hipcc keeps counted vmcnt waits
    //      across the barriers.  Weight tile k+3 is requested while tile k is multiplied.
    int c = 0, pt = pt_begin;
    for (int g = 0; g < g_total; ++g) {
        const unsigned char* pbuf = patch_lds + (g & 1) * P_BYTES;
        const int c_next = c + 1 == nc ? 0 : c + 1;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            compute(pbuf, w_lds + (t % 3) * W_BYTES, t % 3, t / 3);
            store_w((t + 1) % 3, Wr[(t + 1) % 3]);
            if (t == 6) store_patch((g + 1) & 1);
            load_w(t + 3 >= 9 ? c_next : c, 3 * (((t + 3) % 9) % 3) + ((t + 3) % 9) / 3, Wr[t % 3]);   // Wr[t % 3] went to LDS one step ago
            if (t == 7) {
                asm volatile("" ::: "memory");
                load_patch(min(g + 2, g_total - 1));
            }
            __syncthreads();
        }
        if (c_next == 0) {             // last slice of this pixel tile: write it out while the next tile streams in
            epilogue(pt);
            ++pt;
        }
        c = c_next;
    }
}

#endif  // RGBD_DEBUG_BUILD

// ------------------------------------------------------------------------------------------------ 3x3 pipelined kernel
// The same tiling and LDS images as conv3x3_patch_kernel (16x16 output pixels x BN output channels per 512-thread
// workgroup, 18x18 halo patch per 64-channel slice, one BN x 64 weight tile per filter tap), restructured so that the
// matrix pipe never waits for staging or for LDS latency:
//   * global -> LDS by LDS-DMA (`buffer_load_dwordx4 ... offen lds`): no staging registers, no ds_write pass.  The DMA
//     destination is lane-linear (M0 base + 16 B x lane), so the XOR chunk swizzle of both images is applied to each
//     lane's SOURCE address; out-of-image halo rows use an out-of-range buffer offset, which the hardware turns into
//     zeros written to LDS (scripts/hw/lds_dma_probe.hip).  The DMAs are inline asm, i.e. invisible to hipcc's waitcnt
//     bookkeeping: each wave retires its own pieces with a COUNTED s_waitcnt vmcnt(N) and the step's barrier publishes
//     them.  Weight tile t+3 is requested when tile t's buffer falls free (three buffers: two steps to land); the next
//     slice's halo patch streams in over steps 0-3 of the current one.
//   * operand fragments are software-pipelined through registers at HALF-step granularity: a K step (one tap x 64
//     channels) is two blocks of 16 (BN=128) MFMAs over k 0-31 and k 32-63; while a block runs, the ds_reads of the NEXT
//     block's fragments are in flight into the registers the PREVIOUS block has finished with (A: 4 fragments per
//     half; B: the halo rows a wave needs live in TPX row slots per half -- vertical tap kh+1 drops one row and adds one,
//     a new filter column replaces all of them).  A wave therefore never waits on LDS latency, and since the fragment
//     sets are only 32 + 32 registers the whole kernel stays near 170 VGPRs.
//   * ONE barrier per K step (after the k 0-31 block): it publishes weight tile t+1 (and, at step 8, the next halo
//     patch) and certifies that every wave has finished reading tile t, whose buffer the DMA of tile t+3 then reuses.
//     SIMD partners (wave w and w+4) issue MFMAs concurrently; DMA issue (~60 cycles per piece) and fragment reads of
//     one partner run under the other's MFMAs.
//   DMA roles: a wave's memory operations retire in issue order, so a weight piece (L2 hit, needed two steps later)
//   queued behind a halo piece (HBM latency, needed next slice) would inherit its latency: waves 0,1,6,7 issue ONLY
//   weight pieces, waves 2,3,4,5 ONLY halo pieces (one of each role per SIMD).
//   Hazards (B_t = the barrier inside step t):
//     RAW  W(t+1) is first read right after B_t (A fragments of block (t+1, k 0-31)); it was issued after B_(t-2) and is
//          retired by the issuing wave's vmcnt(pieces of W(t+2)) before that wave arrives at B_t.
//     WAR  W(t+3) goes into W(t)'s buffer after B_t; W(t)'s last reads (k 32-63 fragments, issued at the start of step t)
//          are retired by the lgkmcnt(0) in front of B_t.
//     The halo patch of slice g+1 is issued after B_0..B_3 of slice g into the other patch buffer (last read before
//     B_8 of slice g-1), retired by the halo waves' vmcnt(0) in front of B_8, first read right after B_8.
// M0 (the DMA's LDS base) is written here.  LLVM reserves M0 on AMDGPU, so the clobber below is accepted but not
// tracked (-Winline-asm says so); what makes this safe is that nothing hipcc generates for gfx950 in this file reads
// M0 -- tests/test_isa_cpu.py checks the emitted code: every M0 reference is one of these moves, consumed by the
// buffer_load ... lds right behind it.
__device__ __forceinline__ void lds_dma16(u32x4 rsrc, unsigned voff, unsigned soff, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                 :: "s"(lds_dst), "v"(voff), "s"(rsrc), "s"(soff) : "memory", "m0");
}
// 4 bytes per lane (the E8M0 scale dwords of the MX variant): lane l lands at lds_dst + 4 l
__device__ __forceinline__ void lds_dma4(u32x4 rsrc, unsigned voff, unsigned soff, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, %3 offen lds"
                 :: "s"(lds_dst), "v"(voff), "s"(rsrc), "s"(soff) : "memory", "m0");
}
#define RGBD_PP_BARRIER()                         \
    do {                                          \
        __builtin_amdgcn_sched_barrier(0);        \
        asm volatile("s_barrier" ::: "memory");   \
        __builtin_amdgcn_sched_barrier(0);        \
    } while (0)

// scheduling hint for one block: one fragment read (of the NEXT block) behind each of the first MFMAs -- LDS instructions
// issue beside the matrix pipe, so a wave that alternates them keeps the pipe busy on its own
#define RGBD_SP_INTERLEAVE()                                     \
    do {                                                         \
        _Pragma("unroll") for (int k_ = 0; k_ < 8; ++k_) {       \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   \
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   \
        }                                                        \
        __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);       \
    } while (0)

// EPI: 0 plain epilogue, 1 MASKED (activation gradient of the layer in front + its bias gradient, ConvArgs::mask_y),
//      2 STATS (per-(sample, channel) sum y, sum y^2 of what is stored: the instance-norm statistics of the AdaIN behind
//        this conv, ConvArgs::stats)
// MX: the MXFP8 form (BASELINE configuration 5).  Operands are e4m3 bytes with one E8M0 scale byte per 32 input channels
//   (csrc/mxfp8.hip), multiplied by v_mfma_scale_f32_16x16x128_f8f6f4 -- twice the K per matrix-pipe cycle of the bf16 form.
//   A 128-byte LDS row is now 128 channels, so a K step is one tap x 128 channels and the LDS images, the DMA pieces, the
//   bytes read per step and a step's matrix-pipe cycles (16 MFMAs x 32) are those of the bf16 kernel: same tiling, same
//   barriers, same epilogue.  What differs: a lane's 32-byte operand is LDS chunks q and 4 + q of its row (= the
//   instruction's K order, scripts/hw/mfma_f8_probe.hip), each operand carries a scale byte read from two more LDS images
//   (one dword per halo pixel / weight row, filled by 4-byte LDS-DMA pieces), and fragments are pipelined a whole step
//   ahead at QUARTER granularity: the A fragment of 16-channel tile i of step t+1 lands in the registers quarter i of
//   step t has finished with; B rows sit in a ring of row slots, a row's slot being free again before its next tenant's
//   read is issued (NSLOT below).
// EMIT: the epilogue also writes MXFP8 copies of what it stores (ConvArgs::yq ...) -- its own instantiation, because the
//   wide MX form has not a register to spare and the pointers alone tip it into spilling.
template <int BN, bool UPS, int KO = 0, int EPI = 0, bool MX = false, bool EMIT = false>   // KO: timing knock-outs (wrong results): 1 no weight DMA, 2 no halo DMA, 3 no LDS reads, 4 no MFMAs, 6 MFMAs and barriers only
__global__ __launch_bounds__(512, 2) void conv3x3_sp_kernel(ConvArgs a) {
    constexpr bool MASKED = EPI == 1, STATS = EPI == 2;
    constexpr int EB = MX ? 1 : 2;                // bytes per operand element: a 128-byte slice is 64 bf16 or 128 e4m3 channels
    static_assert(!MX || KO == 0, "the knock-outs exist for the bf16 form only");
    constexpr int HPW = UPS ? 10 : 18;            // halo patch width (and height)
    constexpr int NROWS = HPW * HPW;
    constexpr int P_PIECES = (NROWS * 128 + 1023) / 1024;   // 1-KiB DMA pieces per halo patch (41 or 13)
    constexpr int P_BYTES = P_PIECES * 1024;
    constexpr int W_BYTES = BN * 128;
    constexpr int WAVES_CO = BN / 64;
    constexpr int WAVES_PX = 8 / WAVES_CO;
    constexpr int PX_PER_WAVE = 256 / WAVES_PX;   // 64 or 32 pixels = 4 or 2 patch rows
    constexpr int TPX = PX_PER_WAVE / 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];
    unsigned char* const patch_lds = dsm;                       // [2][P_BYTES]
    unsigned char* const w_lds = dsm + 2 * P_BYTES;             // [3][W_BYTES]
    const unsigned lds0 = (unsigned)(size_t)dsm;                // LDS byte address of dsm (low half of the flat address)
    // MX: scale images behind the bias: one dword (the slice's four block scales) per halo pixel / per weight row
    constexpr int S_PIECES = (NROWS + 63) / 64;                 // 256-byte DMA pieces of a halo patch's scale dwords (6 or 2)
    constexpr int XS_BYTES = S_PIECES * 256, WS_BYTES = BN * 4;
    constexpr int XS_OFF = 2 * P_BYTES + 3 * W_BYTES + BN * 4, WS_OFF = XS_OFF + 2 * XS_BYTES;
    unsigned char* const xs_lds = dsm + XS_OFF;                 // [2][XS_BYTES]
    unsigned char* const ws_lds = dsm + WS_OFF;                 // [3][WS_BYTES]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned bid = blockIdx.x;
    {
        const unsigned nwg = gridDim.x, xcd = bid & 7u, q8 = nwg >> 3, r8 = nwg & 7u;
        bid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    }
    const int nt = bid / a.wgs_per_ntile;
    const int slot = bid - nt * a.wgs_per_ntile;
    const int pt_begin = (int)((long)slot * a.ptiles / a.wgs_per_ntile);
    const int pt_end = (int)((long)(slot + 1) * a.ptiles / a.wgs_per_ntile);
    const int tiles_x = a.Wout >> 4, tiles_per_img = tiles_x * (a.Hout >> 4);
    const int n0 = nt * BN;
    const int wave_co = (wid / WAVES_PX) * 64;
    const int wave_py = (wid % WAVES_PX) * TPX;                  // first patch row of this wave

    const int nc = a.Cin / (128 / EB);
    const int g_total = (pt_end - pt_begin) * nc;               // (tile, channel slice) pairs of this workgroup
    if (g_total <= 0) return;
    if (KO == 8) {                      // KO 8: workgroups start 0 / 1 / 2 / 3 x ~0.5 us apart (are the epilogues' store bursts
        for (unsigned k_ = 0; k_ < (blockIdx.x & 3u); ++k_) __builtin_amdgcn_s_sleep(16);   // the stall? timing experiment)
    }

    // ---- DMA roles: weight waves 0,1,6,7, halo waves 2,3,4,5 (wave w and w+4 share a SIMD: one of each per SIMD)
    // (KO 41 / 42, debug A/B: the weight role on the OLDER half, waves 0-3, / on the younger half.  In-kernel stamps,
    // profiles/r05/sp_stamps.txt: the older waves win the issue arbitration, finish a step early and wait ~650 cycles at its
    // barrier while the younger ones are still at work -- DMA issue is cheapest where that wait is)
    const bool w_role = KO == 41 || KO == 31 ? wid < 4 : KO == 42 ? wid >= 4 : ((wid >> 1) & 1) == (wid >> 2);
    const int ridx = (KO == 41 || KO == 42 || KO == 31) ? (wid & 3) : (wid & 1) + 2 * (wid >> 2);     // 0..3 within the role
    constexpr int WPW4 = BN / 32;                                // weight pieces per weight wave per K step (BN / 8 / 4)
    constexpr int PPW4 = (P_PIECES + 3) / 4;                     // halo pieces per halo wave per slice (11 or 4)
    constexpr int PPS = (PPW4 + 3) / 4;                          // ... issued per step (3 or 1), in steps 0 .. PSTEPS-1
    constexpr int PSTEPS = (PPW4 + PPS - 1) / PPS;               // 4
    constexpr int NOFF = PPW4 > WPW4 ? PPW4 : WPW4;
    static_assert(PPW4 <= 12, "halo border masks are packed 5 bits x 6 pieces x 2 registers");
    // Per-lane source offsets, constant for the whole kernel.
    //   halo wave: piece i covers patch rows 8 pi .. 8 pi + 7 (pi = ridx + 4 i, clamped: the last pieces are issued twice
    //     with identical bytes so that every wave issues the same count); lane l fills LDS chunk (l & 7) of row
    //     8 pi + (l >> 3) with SOURCE chunk (l & 7) ^ (halo column & 7).  doff = offset relative to the patch's
    //     top-left halo pixel; the tile-dependent part goes into the (scalar) soffset, against a descriptor whose base
    //     is one halo row + one pixel BEFORE the tensor (never dereferenced: those lanes are masked off below).
    //     Out-of-image lanes are known per tile class: pm packs 5 bits per piece (row 0, last row, column 0, last
    //     column, beyond the patch) that are ANDed with the tile's border bits.
    //   weight wave: piece i = LDS rows 8 (4 ridx + i) .. + 7; LDS row (64 w + 16 t + m) holds output channel
    //     64 w + 16 (m >> 2) + 4 t + (m & 3) (see conv3x3_patch_kernel: every lane then owns 16 consecutive channels)
    unsigned doff[NOFF], pm[2] = {0u, 0u};
#pragma unroll
    for (int i = 0; i < NOFF; ++i) {
        unsigned wv = 0, pv = 0;
        if (i < WPW4) {
            const int r = (ridx * WPW4 + i) * 8 + (lane >> 3);
            const int wm = r & 15, wt = (r >> 4) & 3;
            const int wperm = (r & ~63) + 16 * (wm >> 2) + 4 * wt + (wm & 3);
            wv = (unsigned)((n0 + wperm) * a.Cin * EB + (((lane & 7) ^ (r & 7)) << 4));
        }
        if (i < PPW4) {
            const int pi = ridx + 4 * i < P_PIECES ? ridx + 4 * i : P_PIECES - 1;
            const int row = pi * 8 + (lane >> 3);
            const int hy = row / HPW, hx = row - hy * HPW;
            pv = (unsigned)((hy * a.Win + hx) * a.Cin * EB + (((lane & 7) ^ (hx & 7)) << 4));
            const unsigned m = (hy == 0 ? 1u : 0u) | (hy == HPW - 1 ? 2u : 0u) | (hx == 0 ? 4u : 0u) |
                               (hx == HPW - 1 ? 8u : 0u) | (row >= NROWS ? 16u : 0u);
            pm[i / 6] |= m << (5 * (i % 6));
        }
        doff[i] = w_role ? wv : pv;
    }
    const int tap_stride = a.Cout * a.Cin * EB;
    const unsigned x_lead = (unsigned)((a.Win + 1) * a.Cin * EB);
    const unsigned long xp = (unsigned long)a.x - x_lead, wpp = (unsigned long)a.wp;
    const u32x4 xrsrc = {(unsigned)xp, (unsigned)(xp >> 32) & 0xffffu, (unsigned)a.x_bytes + x_lead, 0x00020000u};
    const u32x4 wrsrc = {(unsigned)wpp, (unsigned)(wpp >> 32) & 0xffffu, (unsigned)a.w_bytes, 0x00020000u};
    // MX: per-lane source offsets of the scale pieces (4 bytes per lane, lane l -> dword 64 piece + l of the image).
    //   halo wave: scale piece i covers halo pixels 64 spi .. + 63 (spi = ridx + 4 i, clamped: duplicates carry the same
    //     bytes), border classes as for the data pieces;  weight wave: piece ridx % (BN / 64) = LDS rows 64 p .. + 63
    //     (every piece is fetched by two -- BN = 64: all four -- weight waves, identical bytes, so that the waves of a role
    //     issue the same number of operations per tile and one counted wait serves them all)
    constexpr int SPW4 = (S_PIECES + 3) / 4;                     // scale pieces per halo wave per slice (2 or 1)
    const int sgrp = a.Cin >> 5;                                 // scale bytes per pixel / per weight row
    unsigned sdoff[SPW4], spm = 0u, wsv = 0u;
    u32x4 xsrsrc = {0u, 0u, 0u, 0u}, wsrsrc = {0u, 0u, 0u, 0u};
    if (MX) {
#pragma unroll
        for (int i = 0; i < SPW4; ++i) {
            const int spi = ridx + 4 * i < S_PIECES ? ridx + 4 * i : S_PIECES - 1;
            const int idx = spi * 64 + lane;
            const int hy = idx / HPW, hx = idx - hy * HPW;
            sdoff[i] = (unsigned)((hy * a.Win + hx) * sgrp);
            const unsigned m = (hy == 0 ? 1u : 0u) | (hy == HPW - 1 ? 2u : 0u) | (hx == 0 ? 4u : 0u) |
                               (hx == HPW - 1 ? 8u : 0u) | (idx >= NROWS ? 16u : 0u);
            spm |= m << (5 * i);
        }
        {
            const int r = (ridx % (BN / 64)) * 64 + lane;
            const int wm = r & 15, wt = (r >> 4) & 3;
            const int wperm = (r & ~63) + 16 * (wm >> 2) + 4 * wt + (wm & 3);
            wsv = (unsigned)((n0 + wperm) * sgrp);
        }
        const unsigned xs_lead = (unsigned)((a.Win + 1) * sgrp);
        const unsigned long xsp = (unsigned long)a.xs - xs_lead, wsp = (unsigned long)a.ws;
        xsrsrc = u32x4{(unsigned)xsp, (unsigned)(xsp >> 32) & 0xffffu, (unsigned)a.xs_bytes + xs_lead, 0x00020000u};
        wsrsrc = u32x4{(unsigned)wsp, (unsigned)(wsp >> 32) & 0xffffu, (unsigned)a.ws_bytes, 0x00020000u};
    }

    auto tile_origin = [&](int pt, int& b, int& y0, int& x0) {
        b = pt / tiles_per_img;
        const int rem = pt - b * tiles_per_img;
        const int ty = rem / tiles_x;
        y0 = ty << 4;
        x0 = (rem - ty * tiles_x) << 4;
    };
    // scalar part of a halo patch's source: (offset of the tile's first pixel (+ channel slice), border bits)
    auto patch_scalar = [&](int pt, int c, unsigned& soff, unsigned& border) {
        int b, y0, x0;
        tile_origin(pt, b, y0, x0);
        const int sy = UPS ? y0 >> 1 : y0, sx = UPS ? x0 >> 1 : x0;
        soff = (unsigned)((((b * a.Hin + sy) * a.Win + sx) * a.Cin) * EB + c * 128);
        border = (y0 == 0 ? 1u : 0u) | (y0 + 16 == a.Hout ? 2u : 0u) | (x0 == 0 ? 4u : 0u) | (x0 + 16 == a.Wout ? 8u : 0u) | 16u;
    };
    auto dma_patch = [&](int i, int buf, unsigned soff, unsigned border) {     // halo waves only
        const int pi = ridx + 4 * i < P_PIECES ? ridx + 4 * i : P_PIECES - 1;
        const bool ok = (pm[i / 6] & (border << (5 * (i % 6)))) == 0u;
        lds_dma16(xrsrc, ok ? doff[i] : 0x80000000u, soff, lds0 + (unsigned)(buf * P_BYTES + pi * 1024));
    };
    auto dma_w_piece = [&](int c, int tap, int buf, int i) {                     // weight waves only
        lds_dma16(wrsrc, doff[i], (unsigned)(tap * tap_stride + c * 128),
                  lds0 + (unsigned)(2 * P_BYTES + buf * W_BYTES + (ridx * WPW4 + i) * 1024));
    };
    auto dma_w = [&](int c, int tap, int buf) {
#pragma unroll
        for (int i = 0; i < WPW4; ++i) dma_w_piece(c, tap, buf, i);
    };
    // MX: the scale pieces.  A patch's scale source differs from its data source by the element size only:
    // soff = pixel * Cin + 128 c  ->  pixel * (Cin / 32) + 4 c = soff / 32
    auto dma_patch_scale = [&](int i, int buf, unsigned soff, unsigned border) {     // halo waves only
        const int spi = ridx + 4 * i < S_PIECES ? ridx + 4 * i : S_PIECES - 1;
        const bool ok = (spm & (border << (5 * i))) == 0u;
        lds_dma4(xsrsrc, ok ? sdoff[i] : 0x80000000u, soff >> 5, lds0 + (unsigned)(XS_OFF + buf * XS_BYTES + spi * 256));
    };
    auto dma_w_scale = [&](int c, int tap, int buf) {                                  // weight waves only
        lds_dma4(wsrsrc, wsv, (unsigned)(tap * a.Cout * sgrp + 4 * c),
                 lds0 + (unsigned)(WS_OFF + buf * WS_BYTES + (ridx % (BN / 64)) * 256));
    };

    f32x4 acc[4][TPX];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < TPX; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int r16 = lane & 15, q = lane >> 4;
    int aoff[2], boff[3][2];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
        aoff[s2] = (wave_co + r16) * 128 + (((4 * s2 + q) ^ (r16 & 7)) << 4);
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const int colx = UPS ? ((r16 + kw - 1) >> 1) + 1 : r16 + kw;
            const int row0 = UPS ? (wave_py >> 1) : wave_py;
            boff[kw][s2] = (row0 * HPW + colx) * 128 + (((4 * s2 + q) ^ (colx & 7)) << 4);
        }
    }
    constexpr int NR = UPS ? TPX / 2 + 2 : TPX + 2;     // halo rows a wave needs per filter column
    // output row j at vertical tap kh reads halo row rowidx(j, kh) (the folded upsample halves the row index)
    auto rowidx = [](int j, int kh) { return UPS ? ((j + kh - 1) >> 1) + 1 : j + kh; };
    bf16x8 brow[TPX][2], af[2][4];                      // [row slot = halo row % TPX][k half], [k half][16-channel tile]
    // A fragments of weight tile `wbuf`, k half h
    auto load_a = [&](const unsigned char* wbuf, int h) {
#pragma unroll
        for (int i = 0; i < 4; ++i) af[h][i] = *reinterpret_cast<const bf16x8*>(wbuf + aoff[h] + i * 16 * 128);
    };
    // B fragments: the halo rows (k half h) that tap (kh, kw) uses and tap (kh - 1, kw) did not; a row's slot is
    // (row % TPX): the row it replaces was last used one vertical tap earlier (or by the previous filter column)
    auto load_b = [&](const unsigned char* pbuf, int kh, int kw, int h) {
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            bool used = false, prev = false;
#pragma unroll
            for (int j = 0; j < TPX; ++j) {
                used = used || rowidx(j, kh) == r;
                prev = prev || (kh > 0 && rowidx(j, kh - 1) == r);
            }
            if (used && !prev) brow[r % TPX][h] = *reinterpret_cast<const bf16x8*>(pbuf + boff[kw][h] + r * HPW * 128);
        }
    };
    auto mfma_quarter = [&](int kh, int h, int i) {             // 16-channel tile i of the wave's 64 x TPX rows
#pragma unroll
        for (int j = 0; j < TPX; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[h][i], brow[rowidx(j, kh) % TPX][h], acc[i][j], 0, 0, 0);
    };
    auto mfma_block = [&](int kh, int h) {
#pragma unroll
        for (int i = 0; i < 4; ++i) mfma_quarter(kh, h, i);
    };
    // ---- MX fragments.  A: one 32-byte operand + scale byte per 16-channel tile, slot i refilled (for the next step)
    //      right after quarter i.  B: halo row r of filter column kw lives in slot (kw NR + r) % NSLOT.  A row is read one
    //      step before its first use (the rows a column starts with: during the last step of the column before); NSLOT
    //      is the smallest divisor-compatible ring in which the previous tenant of a slot has had its last use by then:
    //        TPX 4: rows 0-3 enter during (kw-1, kh 2), row 4 during (kw, 0), row 5 during (kw, 1); row r is last used at
    //               kh = min(r, 2); tenant ord + 9 = (kw+1, r+3) or (kw+2, r-3) enters at least one step later;
    //        TPX 4 behind the folded upsample: 4 rows per column, rows 0-2 at kh 0, row 3 at kh 2: ring of 6;
    //        TPX 2: every row of a slice has its own slot (12 or 9 rows).
    constexpr int NSLOT = !MX ? 1 : TPX == 4 ? (UPS ? 6 : 9) : 3 * NR;
    static_assert(!MX || (3 * NR) % NSLOT == 0, "a slice's rows must map to the same slots in every slice");
    i32x8 afm[MX ? 4 : 1], bm[NSLOT];
    int sam[MX ? 4 : 1], sbm[NSLOT];
    const int sa_off = (wave_co + r16) * 4 + q;                         // scale byte of this lane's K block, weight row r16
    int sb_off[3];
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
        const int colx = UPS ? ((r16 + kw - 1) >> 1) + 1 : r16 + kw;
        sb_off[kw] = ((UPS ? (wave_py >> 1) : wave_py) * HPW + colx) * 4 + q;
    }
    auto load_a_mx = [&](int wb, int i) {                                // tile i's fragment of weight buffer wb
        const unsigned char* wbuf = w_lds + wb * W_BYTES;
        // (the two chunks of a lane differ in chunk bit 2 = address bit 6, which the XOR swizzle never touches with a carry:
        // aoff[1] == aoff[0] ^ 64 -- one short-lived VALU result instead of a register held across the kernel)
        const u32x4 lo = *reinterpret_cast<const u32x4*>(wbuf + aoff[0] + i * 16 * 128);
        const u32x4 hi = *reinterpret_cast<const u32x4*>(wbuf + (aoff[0] ^ 64) + i * 16 * 128);
        afm[i] = i32x8{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
        sam[i] = (int)ws_lds[wb * WS_BYTES + sa_off + i * 64];
    };
    auto load_b_mx = [&](int pb, int kh, int kw, int part) {             // the rows tap (kh, kw) uses and (kh - 1, kw) did not,
        const unsigned char* pbuf = patch_lds + pb * P_BYTES;            // every third of them (part 0..2, or all: part < 0)
        int cnt = 0;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            bool used = false, prev = false;
#pragma unroll
            for (int j = 0; j < TPX; ++j) {
                used = used || rowidx(j, kh) == r;
                prev = prev || (kh > 0 && rowidx(j, kh - 1) == r);
            }
            if (used && !prev && (part < 0 || cnt++ % 3 == part)) {
                const int sl = (kw * NR + r) % NSLOT;
                const u32x4 lo = *reinterpret_cast<const u32x4*>(pbuf + boff[kw][0] + r * HPW * 128);
                const u32x4 hi = *reinterpret_cast<const u32x4*>(pbuf + (boff[kw][0] ^ 64) + r * HPW * 128);
                bm[sl] = i32x8{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
                sbm[sl] = (int)xs_lds[pb * XS_BYTES + (UPS ? sb_off[kw] : sb_off[0] + 4 * kw) + r * HPW * 4];
            }
        }
    };
    // The scaled-MFMA builtin is not treated as convergent by the compiler, which then sinks every MFMA of the unrolled
    // slice past the role branches to the accumulators' first real use, the epilogue (72 MFMAs in one block behind the last
    // barrier, all fragments spilled on the way).  An empty volatile asm that "rewrites" a quarter's accumulators one
    // quarter later keeps each MFMA within a quarter of where it is written; by then its result has long retired, so the
    // asm costs no wait states.
    auto pin_quarter = [&](int i) {
#pragma unroll
        for (int j = 0; j < TPX; ++j) asm volatile("" : "+v"(acc[i][j]));
    };
    auto mfma_quarter_mx = [&](int kh, int kw, int i) {
#pragma unroll
        for (int j = 0; j < TPX; ++j) {
            const int sl = (kw * NR + rowidx(j, kh)) % NSLOT;
            acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(afm[i], bm[sl], acc[i][j], 0, 0, 0, sam[i], 0, sbm[sl]);
        }
    };

    float* const bias_lds = reinterpret_cast<float*>(dsm + 2 * P_BYTES + 3 * W_BYTES);   // [BN]
    if (tid < BN) bias_lds[tid] = a.bias ? a.bias[n0 + tid] : 0.f;

    // TR (the MX form, which has no registers to spare): the epilogue's per-lane sums (16 channels x this lane's pixel column)
    // do not live across the main loop -- each tile's 16 values are reduced over the 16 pixel lanes at the end of its
    // epilogue by a transposing butterfly (round with distance d: a lane keeps the half of its values whose channel bit
    // equals its lane bit and adds the partner's copy of them; after four rounds lane r16 holds channel r16's total) and
    // added to ONE register per statistic.  The bf16 form keeps its 16 (32) registers and reduces once at the end.
    constexpr bool TR = MX;
    auto transpose_reduce16 = [&](const float (&v)[16]) {
        float w8[8], w4[4], w2[2];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const bool hi = (r16 & 8) != 0;
            w8[k] = (hi ? v[k + 8] : v[k]) + __shfl_xor(hi ? v[k] : v[k + 8], 8);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const bool hi = (r16 & 4) != 0;
            w4[k] = (hi ? w8[k + 4] : w8[k]) + __shfl_xor(hi ? w8[k] : w8[k + 4], 4);
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const bool hi = (r16 & 2) != 0;
            w2[k] = (hi ? w4[k + 2] : w4[k]) + __shfl_xor(hi ? w4[k] : w4[k + 2], 2);
        }
        const bool hi = (r16 & 1) != 0;
        return (hi ? w2[1] : w2[0]) + __shfl_xor(hi ? w2[0] : w2[1], 1);
    };
    float cs[(MASKED && !TR) ? 16 : 1]; // MASKED: this lane's share of the weighted column sums, over all of its tiles
    float cs_acc = 0.f, st1_acc = 0.f, st2_acc = 0.f;      // TR: channel (co + r16)'s totals
#pragma unroll
    for (int k2 = 0; k2 < ((MASKED && !TR) ? 16 : 1); ++k2) cs[k2] = 0.f;
    // STATS: this lane's share of (sum y, sum y^2) of its 16 channels over the tiles of image st_b walked so far; a
    // workgroup's tiles are consecutive, so it flushes once per image it touches (and at the end)
    float st1[(STATS && !TR) ? 16 : 1], st2[(STATS && !TR) ? 16 : 1];
    int st_b = -1;
#pragma unroll
    for (int k2 = 0; k2 < ((STATS && !TR) ? 16 : 1); ++k2) { st1[k2] = 0.f; st2[k2] = 0.f; }
    auto stats_flush = [&]() {          // 16 pixel columns -> per-wave totals; lane r16 then owns channel co + r16
        float v1 = 0.f, v2 = 0.f;
        if constexpr (TR) {
            v1 = st1_acc; v2 = st2_acc;
            st1_acc = 0.f; st2_acc = 0.f;
        } else {
#pragma unroll
            for (int k2 = 0; k2 < 16; ++k2) {
                float t1 = st1[k2], t2 = st2[k2];
                t1 += __shfl_xor(t1, 1); t2 += __shfl_xor(t2, 1);
                t1 += __shfl_xor(t1, 2); t2 += __shfl_xor(t2, 2);
                t1 += __shfl_xor(t1, 4); t2 += __shfl_xor(t2, 4);
                t1 += __shfl_xor(t1, 8); t2 += __shfl_xor(t2, 8);
                v1 = r16 == k2 ? t1 : v1;
                v2 = r16 == k2 ? t2 : v2;
                st1[k2] = 0.f; st2[k2] = 0.f;
            }
        }
        unsigned long long* dst = reinterpret_cast<unsigned long long*>(a.stats) +
                                  ((long)st_b * a.Cout + n0 + wave_co + 16 * q + r16) * 2;
        // a NaN / Inf activation or a sum beyond the fixed-point range (2^31) must not come out as a finite, wrong statistic:
        // such a wave raises the second moment to INT64_MAX instead (any later add wraps it negative) and the consumer
        // (adain_strip_sum) turns a negative or absurd second moment into NaN -- the layer's output shows the fault
        if (!(fabsf(v1) < 1e9f) || !(v2 < 1e9f)) {
            atomicMax(reinterpret_cast<long long*>(dst) + 1, 0x7fffffffffffffffLL);
        } else {
            atomicAdd(dst, (unsigned long long)__double2ll_rn((double)v1 * 4294967296.0));
            atomicAdd(dst + 1, (unsigned long long)__double2ll_rn((double)v2 * 4294967296.0));
        }
    };

    // MXFP8 copy of 16 stored channels of one pixel (two u32x4 of bf16 pairs): this lane and lane ^ 16 (q ^ 1: the same pixel,
    // the neighbouring 16 channels) form one 32-channel block; exactly rgbd_quantize_mxfp8 of the stored tensor
    auto emit_mx8 = [&](unsigned char* qbase, unsigned char* sbase, long o, const u32x4& lo4, const u32x4& hi4) {
        float f[16];
#pragma unroll
        for (int w2 = 0; w2 < 4; ++w2) {
            f[2 * w2] = bf16_lo(lo4[w2]); f[2 * w2 + 1] = bf16_hi(lo4[w2]);
            f[8 + 2 * w2] = bf16_lo(hi4[w2]); f[8 + 2 * w2 + 1] = bf16_hi(hi4[w2]);
        }
        float amax = 0.f;
#pragma unroll
        for (int k2 = 0; k2 < 16; ++k2) amax = fmaxf(amax, fabsf(f[k2]));
        amax = fmaxf(amax, __shfl_xor(amax, 16));
        const unsigned sc = mx8_scale_of(amax);
        const float inv = mx8_inv_scale(sc);
        const u32x4 qv = {mx8_pack4(f[0] * inv, f[1] * inv, f[2] * inv, f[3] * inv),
                          mx8_pack4(f[4] * inv, f[5] * inv, f[6] * inv, f[7] * inv),
                          mx8_pack4(f[8] * inv, f[9] * inv, f[10] * inv, f[11] * inv),
                          mx8_pack4(f[12] * inv, f[13] * inv, f[14] * inv, f[15] * inv)};
        *reinterpret_cast<u32x4*>(qbase + o) = qv;
        if ((q & 1) == 0) sbase[o >> 5] = (unsigned char)sc;
    };

    auto epilogue = [&](int pt) {       // bias -> residual -> leaky ReLU (or a given mask) -> bf16 NHWC, then clear the accumulators
        if (KO == 9 && a.B >= 0) {      // KO 9: no epilogue (the main loop alone; the opaque condition keeps the MFMAs live)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < TPX; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            return;
        }
        int b, y0, x0;
        tile_origin(pt, b, y0, x0);
        const int co = n0 + wave_co + 16 * q;            // this lane's 16 consecutive output channels
        const bool act = co < a.lrelu_ch;
        const float cw = MASKED && a.colsum && a.row_scale ? a.row_scale[b] : 1.f;
        const float cw2 = MASKED && a.y2 ? a.row_scale2[b] : 0.f;
        if (STATS) st_b = b;
        if (a.pool_sum) {
            const int Hp = a.Hout >> 1, Wp = a.Wout >> 1;
#pragma unroll
            for (int j = 0; j < TPX; j += 2) {
                float v[16];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float t = acc[i][j][r] + acc[i][j + 1][r];
                        const int o = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0xB1, 0xF, 0xF, false);
                        v[4 * i + r] = t + __builtin_bit_cast(float, o);
                    }
                if ((r16 & 1) == 0) {
                    const int yy = (y0 + wave_py + j) >> 1, xx = (x0 + r16) >> 1;
                    const long o = (((long)b * Hp + yy) * Wp + xx) * a.Cout + co;
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        u32x4 out = {pack_bf16x2(v[8 * h + 0], v[8 * h + 1]), pack_bf16x2(v[8 * h + 2], v[8 * h + 3]),
                                     pack_bf16x2(v[8 * h + 4], v[8 * h + 5]), pack_bf16x2(v[8 * h + 6], v[8 * h + 7])};
                        *reinterpret_cast<u32x4*>(a.y + o + 8 * h) = out;
                    }
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                    acc[i][j + 1] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
            return;
        }
        float ps[16];                   // ypool: running 2x2 sums of the bf16-rounded outputs of a row pair
        float tl0[TR && (MASKED || STATS) ? 16 : 1], tl1[TR && STATS ? 16 : 1];     // TR: this tile's sums
#pragma unroll
        for (int k2 = 0; k2 < (TR && (MASKED || STATS) ? 16 : 1); ++k2) tl0[k2] = 0.f;
#pragma unroll
        for (int k2 = 0; k2 < (TR && STATS ? 16 : 1); ++k2) tl1[k2] = 0.f;
#pragma unroll
        for (int j = 0; j < TPX; ++j) {
            const int yy = y0 + wave_py + j, xx = x0 + r16;
            const long o = (((long)b * a.Hout + yy) * a.Wout + xx) * a.Cout + co;
            float v[16];
            if (a.bias) {               // (uniform: input-gradient launches have no bias and skip 16 VALU slots per row)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const f32x4 bq = *reinterpret_cast<const f32x4*>(bias_lds + wave_co + 16 * q + 4 * i);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[4 * i + r] = acc[i][j][r] + bq[r];
                }
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[4 * i + r] = acc[i][j][r];
            }
            if (a.resid) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const u32x4 rr = *reinterpret_cast<const u32x4*>(a.resid + o + 8 * h);
#pragma unroll
                    for (int w2 = 0; w2 < 4; ++w2) {
                        v[8 * h + 2 * w2] += bf16_lo(rr[w2]);
                        v[8 * h + 2 * w2 + 1] += bf16_hi(rr[w2]);
                    }
                }
            }
            if (act) {                  // 0 <= slope <= 1 (checked by the launcher): max(v, slope v) IS the leaky ReLU, bit for
#pragma unroll                  // bit incl. -0 and NaN, in two VALU slots per value instead of three
                for (int k2 = 0; k2 < 16; ++k2) {       // (asm: fmaxf() puts a canonicalising v_max in front of every max)
                    const float sv = v[k2] * a.slope;
                    asm("v_max_f32 %0, %1, %2" : "=v"(v[k2]) : "v"(v[k2]), "v"(sv));
                }
            }
            u32x4 mk[MASKED ? 2 : 1];
            if (MASKED) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const u32x4 mm = *reinterpret_cast<const u32x4*>(a.mask_y + o + 8 * h);
                    mk[h] = mm;
#pragma unroll
                    for (int w2 = 0; w2 < 4; ++w2) {
                        v[8 * h + 2 * w2] = bf16_lo(mm[w2]) > 0.f ? v[8 * h + 2 * w2] : v[8 * h + 2 * w2] * a.slope;
                        v[8 * h + 2 * w2 + 1] = bf16_hi(mm[w2]) > 0.f ? v[8 * h + 2 * w2 + 1] : v[8 * h + 2 * w2 + 1] * a.slope;
                    }
                }
            }
            u32x4 stored[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                u32x4 out = {pack_bf16x2(v[8 * h + 0], v[8 * h + 1]), pack_bf16x2(v[8 * h + 2], v[8 * h + 3]),
                             pack_bf16x2(v[8 * h + 4], v[8 * h + 5]), pack_bf16x2(v[8 * h + 6], v[8 * h + 7])};
                if (KO != 7 || a.B < 0) *reinterpret_cast<u32x4*>(a.y + o + 8 * h) = out;   // KO 7: the epilogue without its stores
                stored[h] = out;
                if (MASKED) {           // column sums of what was stored (the rounded values, as the separate pass took them)
#pragma unroll
                    for (int w2 = 0; w2 < 4; ++w2) {
                        float* const cacc = TR ? tl0 : cs;
                        cacc[TR || MASKED ? 8 * h + 2 * w2 : 0] += cw * bf16_lo(out[w2]);
                        cacc[TR || MASKED ? 8 * h + 2 * w2 + 1 : 0] += cw * bf16_hi(out[w2]);
                    }
                    if (a.y2) {         // rgbd_axpy_rows_bf16 of (what was stored, the activation tile): the injection operand
                        u32x4 o2;
#pragma unroll
                        for (int w2 = 0; w2 < 4; ++w2)
                            o2[w2] = pack_bf16x2(bf16_lo(out[w2]) + cw2 * bf16_lo(mk[h][w2]),
                                                 bf16_hi(out[w2]) + cw2 * bf16_hi(mk[h][w2]));
                        *reinterpret_cast<u32x4*>(a.y2 + o + 8 * h) = o2;
                    }
                }
                if (STATS) {
#pragma unroll
                    for (int w2 = 0; w2 < 4; ++w2) {
                        const float lo = bf16_lo(out[w2]), hi = bf16_hi(out[w2]);
                        float* const s1p = TR ? tl0 : st1;
                        float* const s2p = TR ? tl1 : st2;
                        s1p[8 * h + 2 * w2] += lo;
                        s2p[8 * h + 2 * w2] += lo * lo;
                        s1p[8 * h + 2 * w2 + 1] += hi;
                        s2p[8 * h + 2 * w2 + 1] += hi * hi;
                    }
                }
                if (a.ypool) {          // the block's downscale2x (rescale.py:12-13) of what was just stored
#pragma unroll
                    for (int w2 = 0; w2 < 4; ++w2) {
                        const float lo = bf16_lo(out[w2]), hi = bf16_hi(out[w2]);
                        ps[8 * h + 2 * w2] = (j & 1) ? ps[8 * h + 2 * w2] + lo : lo;
                        ps[8 * h + 2 * w2 + 1] = (j & 1) ? ps[8 * h + 2 * w2 + 1] + hi : hi;
                    }
                }
            }
            if constexpr (EMIT) {
                if (a.yq) emit_mx8(a.yq, a.ys, o, stored[0], stored[1]);
            }
            if (a.ypool && (j & 1)) {
#pragma unroll
                for (int k2 = 0; k2 < 16; ++k2) {
                    const int o2 = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, ps[k2]), 0xB1, 0xF, 0xF, false);
                    ps[k2] = 0.25f * (ps[k2] + __builtin_bit_cast(float, o2));
                }
                if ((r16 & 1) == 0) {
                    const long op = (((long)b * (a.Hout >> 1) + (yy >> 1)) * (a.Wout >> 1) + (xx >> 1)) * a.Cout + co;
                    u32x4 pst[2];
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        u32x4 out = {pack_bf16x2(ps[8 * h + 0], ps[8 * h + 1]), pack_bf16x2(ps[8 * h + 2], ps[8 * h + 3]),
                                     pack_bf16x2(ps[8 * h + 4], ps[8 * h + 5]), pack_bf16x2(ps[8 * h + 6], ps[8 * h + 7])};
                        *reinterpret_cast<u32x4*>(a.ypool + op + 8 * h) = out;
                        pst[h] = out;
                    }
                    if constexpr (EMIT) {
                        if (a.ypq) emit_mx8(a.ypq, a.yps, op, pst[0], pst[1]);
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if constexpr (TR && MASKED) cs_acc += transpose_reduce16(tl0);
        if constexpr (TR && STATS) {
            st1_acc += transpose_reduce16(tl0);
            st2_acc += transpose_reduce16(tl1);
        }
        // one flush per TILE, not per (workgroup, image): a tile's fp32 partial sums do not depend on which workgroup walks it,
        // and the fixed-point integer adds behind them are associative -- the statistics (and with them the generator's
        // forward) are bit-identical whatever grid the launch was sized for (`cus`).  Per-image flushes made them depend on the
        // tile-to-workgroup assignment at fp32 rounding level (1e-7), which bf16 roundings downstream amplified to 1e-4 in the
        // mapping network's gradient between a 240- and a 256-workgroup launch.
        if constexpr (STATS) stats_flush();
    };

    // ---- prologue: halo patch of slice 0, weight tiles of steps 0-2 (step t multiplies filter tap
    //      (kh, kw) = (t % 3, t / 3), i.e. weight image 3 (t % 3) + t / 3), then the fragments of block (0, k 0-31)
    if (w_role) {
        dma_w(0, 0, 0);
        dma_w(0, 3, 1);
        dma_w(0, 6, 2);
        if (MX) {
            dma_w_scale(0, 0, 0);
            dma_w_scale(0, 3, 1);
            dma_w_scale(0, 6, 2);
        }
    } else {
        unsigned soff, border;
        patch_scalar(pt_begin, 0, soff, border);
#pragma unroll
        for (int i = 0; i < PPW4; ++i) dma_patch(i, 0, soff, border);
        if (MX) {
#pragma unroll
            for (int i = 0; i < SPW4; ++i) dma_patch_scale(i, 0, soff, border);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (MX) {
        load_a_mx(0, 0);
        load_a_mx(0, 1);
        load_a_mx(0, 2);
        load_b_mx(0, 0, 0, -1);
    } else {
        load_a(w_lds, 0);
        load_b(patch_lds, 0, 0, 0);
    }

    int c = 0, pt = pt_begin;
    bool w_waited = false;             // weight waves: tile 1 of this slice was already waited for (in front of an epilogue)
    unsigned long long stamp_sum[5] = {0, 0, 0, 0, 0};   // KO 30 (debug): cycles in block 1 / wait + barrier / block 2 + DMA of the steps with / without halo DMA / epilogue
    constexpr int NWT = MX ? WPW4 + 1 : WPW4;          // a weight wave's DMA operations per tile (MX: + the scale piece)
    for (int g = 0; g < g_total; ++g) {
        const unsigned char* pbuf = patch_lds + (g & 1) * P_BYTES;
        const unsigned char* pnext = patch_lds + ((g + 1) & 1) * P_BYTES;
        const int c_next = c + 1 == nc ? 0 : c + 1;
        const bool last = g + 1 >= g_total;
        const int pt_next = (c_next == 0 && !last) ? pt + 1 : pt;
        unsigned nsoff, nborder;
        patch_scalar(pt_next, c_next, nsoff, nborder);
        if constexpr (MX) {
            // One K step = tap (kh, kw) x 128 channels = four quarters of TPX MFMAs (one 16-channel tile each).
            //   quarter 0 runs in front of B_t: it needs nothing B_t publishes; A_3 of THIS step is read under it
            //   B_t: weight tile t + 1 (+ its scales) has landed in every weight wave; at t = 8 the next halo patch too
            //   quarters 1-3: A_0..2 and the new B rows of step t + 1 arrive; the DMAs of weight tile t + 3 (into tile t's
            //   buffer: its last reads, A_3 of step t, are retired by the lgkmcnt(0) in front of B_t) and of the next halo
            //   patch go out one or two per quarter.  Hazards otherwise as in the bf16 form (the kernel's header).
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int kh = t % 3, kw = t / 3;
                const int tn = (t + 1) % 9, khn = tn % 3, kwn = tn / 3;
                load_a_mx(t % 3, 3);
                mfma_quarter_mx(kh, kw, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, TPX, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (t > 0) pin_quarter(3);
                if (w_role) {
                    if (!(t == 0 && w_waited)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NWT) : "memory");
                } else if (t == 8) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                RGBD_PP_BARRIER();
#pragma unroll
                for (int i = 1; i < 4; ++i) {
                    load_a_mx(tn % 3, i - 1);
                    load_b_mx(t == 8 ? (g + 1) & 1 : g & 1, khn, kwn, i - 1);
                    mfma_quarter_mx(kh, kw, i);
#pragma unroll
                    for (int k_ = 0; k_ < TPX; ++k_) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    pin_quarter(i - 1);
                    const int cw = t + 3 >= 9 ? c_next : c, tapw = 3 * (((t + 3) % 9) % 3) + ((t + 3) % 9) / 3;
                    if (w_role) {
                        if (i < 3) {
#pragma unroll
                            for (int k_ = (i - 1) * (WPW4 / 2); k_ < i * (WPW4 / 2); ++k_) dma_w_piece(cw, tapw, t % 3, k_);
                        } else {
                            dma_w_scale(cw, tapw, t % 3);
                        }
                    } else if (t < PSTEPS) {
                        if (i - 1 < PPS && t * PPS + i - 1 < PPW4) dma_patch(t * PPS + i - 1, (g + 1) & 1, nsoff, nborder);
                    } else if (t == PSTEPS) {
                        if (i - 1 < SPW4) dma_patch_scale(i - 1, (g + 1) & 1, nsoff, nborder);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            pin_quarter(3);
        } else {
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            constexpr int dummy = 0;
            (void)dummy;
            const int kh = t % 3, kw = t / 3;
            const int tn = (t + 1) % 9, khn = tn % 3, kwn = tn / 3;
            unsigned long long ts0 = 0, ts1 = 0, ts2 = 0;
            if (KO == 30 || KO == 31) ts0 = __builtin_amdgcn_s_memtime();
            // ---- block (t, k 0-31); meanwhile the k 32-63 fragments of this step arrive
            if (KO != 3 && KO != 6) {
                load_a(w_lds + (t % 3) * W_BYTES, 1);
                load_b(pbuf, kh, kw, 1);
            }
            if (KO != 4) mfma_block(kh, 0);
            RGBD_SP_INTERLEAVE();
            __builtin_amdgcn_sched_barrier(0);
            if (KO == 30 || KO == 31) ts1 = __builtin_amdgcn_s_memtime();
            // ---- B_t.  Weight waves: all but the youngest tile (t + 2) have landed, i.e. tile t + 1; halo waves: the
            //      whole next patch (issued in steps 0-3: HBM latency under load is a few steps), in front of B_8
            if (w_role) {
                if (!(t == 0 && w_waited)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(KO == 1 || KO == 6 ? 0 : WPW4) : "memory");
            } else if (t == 8) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            RGBD_PP_BARRIER();
            if (KO == 30 || KO == 31) ts2 = __builtin_amdgcn_s_memtime();
            // ---- block (t, k 32-63): the k 0-31 fragments of step t + 1 arrive, and the DMAs go out one per quarter of
            //      the block (an LDS-DMA holds its wave for ~60 cycles: spread out, the SIMD partner's MFMAs cover it):
            //      weight tile t + 3 into tile t's buffer / the next halo pieces
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (KO != 3 && KO != 6) {
                    if (i == 0) load_a(w_lds + (tn % 3) * W_BYTES, 0);
                    if (i == 1) load_b(t == 8 ? pnext : pbuf, khn, kwn, 0);
                }
                if (KO != 4) mfma_quarter(kh, 1, i);
#pragma unroll
                for (int k_ = 0; k_ < 4; ++k_) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (w_role) {
                    if (i < WPW4 && KO != 1 && KO != 6)
                        dma_w_piece(t + 3 >= 9 ? c_next : c, 3 * (((t + 3) % 9) % 3) + ((t + 3) % 9) / 3, t % 3, i);
                } else if (t < PSTEPS && i < PPS && t * PPS + i < PPW4 && KO != 2 && KO != 6) {
                    dma_patch(t * PPS + i, (g + 1) & 1, nsoff, nborder);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (KO == 30 || KO == 31) {
                const unsigned long long ts3 = __builtin_amdgcn_s_memtime();
                stamp_sum[0] += ts1 - ts0; stamp_sum[1] += ts2 - ts1; stamp_sum[t < PSTEPS ? 2 : 3] += ts3 - ts2;
            }
        }
        }
        w_waited = false;
        if (c_next == 0) {             // last slice of this pixel tile: write it out.  The weight waves first retire tile 1
                                       // of the next slice: behind the epilogue's stores a counted wait would also wait
                                       // for those
            if (w_role) {
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(KO == 1 || KO == 6 ? 0 : NWT) : "memory");
                w_waited = true;
            }
            unsigned long long te0 = 0;
            if (KO == 30 || KO == 31) te0 = __builtin_amdgcn_s_memtime();
            epilogue(pt);
            if (KO == 30 || KO == 31) stamp_sum[4] += __builtin_amdgcn_s_memtime() - te0;
        }
        c = c_next;
        pt = pt_next;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // no DMA may land in LDS after the workgroup has gone
#ifdef RGBD_DEBUG_BUILD
    if ((KO == 30 || KO == 31) && a.partial && lane == 0) {
        unsigned* o = reinterpret_cast<unsigned*>(a.partial) + 8 * 1024 + (blockIdx.x * 8 + wid) * 5;
#pragma unroll
        for (int k_ = 0; k_ < 5; ++k_) o[k_] = (unsigned)stamp_sum[k_];
    }
#endif
    if (MASKED && a.colsum) {
        // 16 pixel columns (lanes r16) -> one value per channel per wave; the WAVES_PX waves that share channels meet
        // through LDS (idle by now: every wave has retired its DMAs, the barrier says so for all of them); then ONE fp32
        // atomic per channel per workgroup, BN consecutive addresses per instruction (4 lanes x 16 instructions per wave
        // on the same 64 addresses, all workgroups at once, serialised for 0.8 ms)
        if constexpr (!TR) {
#pragma unroll
            for (int k2 = 0; k2 < 16; ++k2) {
                float t = cs[k2];
                t += __shfl_xor(t, 1);
                t += __shfl_xor(t, 2);
                t += __shfl_xor(t, 4);
                t += __shfl_xor(t, 8);
                cs[k2] = t;
            }
        }
        __syncthreads();
        float* const red = reinterpret_cast<float*>(dsm);              // [WAVES_PX][BN]
        if constexpr (TR) {
            red[(wid % WAVES_PX) * BN + wave_co + 16 * q + r16] = cs_acc;
        } else if (r16 == 0) {
#pragma unroll
            for (int k2 = 0; k2 < 16; ++k2) red[(wid % WAVES_PX) * BN + wave_co + 16 * q + k2] = cs[k2];
        }
        __syncthreads();
        if (tid < BN) {
            float t = 0.f;
#pragma unroll
            for (int w2 = 0; w2 < WAVES_PX; ++w2) t += red[w2 * BN + tid];
            atomicAdd(a.colsum + n0 + tid, t);
        }
    }
}

#ifdef RGBD_DEBUG_BUILD
// ------------------------------------------------------------------------------------------------ 3x3 dual-workgroup kernel
// (DEBUG library only: the measured A/B of round 5, not a shipped dataflow -- see "Outcome" below.)
// conv3x3_dw_kernel: the pipelined kernel's dataflow (halo patch per channel slice, LDS-DMA staging with source-side
// swizzles and counted waits, fragments pipelined through registers, one barrier per K step) re-tiled so that TWO
// independent workgroups live on every CU:
//   * 256 threads = 4 waves, one per SIMD; a workgroup still owns 16x16 output pixels x BN output channels, every wave
//     4 patch rows (64 pixels) x ALL BN channels: BN / 16 x 4 accumulator tiles = 128 registers for BN = 128.  With at
//     most 256 registers per wave two workgroups share a CU's SIMDs, i.e. half of the register file holds accumulators
//     -- the "second accumulator set" the 8-wave kernel has no room for, owned by another workgroup.
//   * channel slices of 32 (64-byte LDS rows): 2 x 21 KiB of halo patch + 3 x BN x 64 B of weight tiles + bias =
//     66.5 KiB per workgroup, 133 of the CU's 160 KiB for the pair.
//   * the two workgroups of a CU share nothing and synchronise with nobody but themselves: one's epilogue, barrier waits,
//     DMA issue and LDS latency can run under the other's MFMAs.
//   A K step = one filter tap x 32 channels = BN / 16 x 4 v_mfma_f32_16x16x32_bf16 per wave (the same 32 MFMAs between
//   two barriers as a wave of the 8-wave kernel at BN = 128); a fragment is ONE ds_read_b128 (lane (r16, q): row r16,
//   16-byte chunk q = k 8q .. 8q+7).
//   LDS images: rows of 64 bytes, four 16-byte chunks; chunk position p of row R holds source chunk p ^ f(R) with
//     f = 2 ((r16 >> 2) & 1) for weight rows and 2 ((hx >> 2) & 1) for the halo pixel in patch column hx: the sixteen
//     lanes that ds_read_b128 serves together (rows R0 .. R0+15 of one chunk column pair, ANY R0: a filter column shifts
//     it) then touch sixteen different 16-byte bank groups -- four rows apart f flips, which separates the two rows of a
//     bank class that read the same chunk (MI355X_MICROARCH.md, LDS: lane groups {0-3,12-15,20-27}, ...).
//   DMA roles: waves 0,1 weight pieces (BN / 32 per step each), waves 2,3 halo pieces (11 per slice each, steps 0-3): a
//     wave's memory operations retire in issue order, so an L2-hit weight piece must not queue behind an HBM halo piece.
//   Pipeline per step t (tap (t % 3, t / 3), A tiles in four quarters, double-buffered at quarter granularity: NI / 2
//     register slots): see the comment at the main loop.  Halo row r of filter column kw lives in register slot
//     (kw NR + r) % NSLOT (conv3x3_sp_kernel's MX form, same proof).
//   Sum order differs from conv3x3_sp_kernel (32-channel slices), so outputs agree with it to fp32 accumulation noise
//   (1-2 in 10^4 bf16 outputs one ulp apart), not byte for byte; run to run the kernel is bit-reproducible.
// Outcome (profiles/r05/ab_conv_dw*.txt, dw_census.txt): level with the 8-wave kernel on every shape of
//   the step (within +-3 % once the measurement order is rotated; +5-7 % only on the upsampling 128 -> 64 layer), so it is
//   not shipped.  The census shows why the premise fails: the CU's two workgroups DO run side by side (512 of 512
//   co-resident pairs), but issue arbitration favours the older one, which finishes 20-25 % earlier and leaves the younger
//   a tail alone on the CU; with priorities alternated per tile both end together and the launch gains 1-3 %.  In-kernel
//   stamps: a wave spends 12-31 % of its time in the step's wait + barrier and 17-20 % (Cin = 64) in the epilogue whether or
//   not another workgroup is there to fill the matrix pipe -- per step the pair needs ~1300 cycles where the MFMAs need 1024.
//   BN = 128 needs 128 accumulator + 16 A + 36 B registers before addresses: the fused epilogues spill inside the main loop.
template <int BN, bool UPS, int EPI = 0, int KO = 0>      // KO (debug library; timing only, wrong results): 1 no weight DMA, 2 no halo DMA, 3 no LDS reads, 6 all three, 9 no epilogue
__global__ __launch_bounds__(256, 2) void conv3x3_dw_kernel(ConvArgs a) {
    constexpr bool MASKED = EPI == 1, STATS = EPI == 2;
    constexpr int HPW = UPS ? 10 : 18;            // halo patch width (and height)
    constexpr int NROWS = HPW * HPW;
    constexpr int P_PIECES = (NROWS * 64 + 1023) / 1024;     // 1-KiB DMA pieces (16 rows) per halo patch: 21 or 7
    constexpr int P_BYTES = P_PIECES * 1024;
    constexpr int W_BYTES = BN * 64;
    constexpr int NI = BN / 16;                   // 16-channel A tiles per wave: 8 or 4
    constexpr int QT = NI / 4;                    // ... per quarter of a step
    constexpr int NH = BN / 64;                   // 64-channel halves: the epilogue's unit (a lane owns 16 consecutive channels of each)
    constexpr int TPX = 4;                        // patch rows per wave
    extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];
    unsigned char* const patch_lds = dsm;                       // [2][P_BYTES]
    unsigned char* const w_lds = dsm + 2 * P_BYTES;             // [3][W_BYTES]
    const unsigned lds0 = (unsigned)(size_t)dsm;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned bid = blockIdx.x;
    {
        const unsigned nwg = gridDim.x, xcd = bid & 7u, q8 = nwg >> 3, r8 = nwg & 7u;
        bid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    }
    const int nt = bid / a.wgs_per_ntile;
    const int slot = bid - nt * a.wgs_per_ntile;
    const int pt_begin = (int)((long)slot * a.ptiles / a.wgs_per_ntile);
    const int pt_end = (int)((long)(slot + 1) * a.ptiles / a.wgs_per_ntile);
    const int tiles_x = a.Wout >> 4, tiles_per_img = tiles_x * (a.Hout >> 4);
    const int n0 = nt * BN;
    const int wave_py = wid * TPX;                               // first patch row of this wave

    const int nc = a.Cin >> 5;
    const int g_total = (pt_end - pt_begin) * nc;               // (tile, channel slice) pairs of this workgroup
    if (g_total <= 0) return;
#ifdef RGBD_DEBUG_BUILD
    unsigned long long census_t0 = 0;
    if (a.partial) census_t0 = __builtin_amdgcn_s_memrealtime();
#endif

    const bool w_role = wid < 2;
    const int ridx = wid & 1;                                    // 0..1 within the role
    constexpr int WPW = NI / 2;                                  // weight pieces per weight wave per K step (4 or 2)
    constexpr int PPW = (P_PIECES + 1) / 2;                      // halo pieces per halo wave per slice (11 or 4)
    constexpr int PPS = (PPW + 3) / 4;                           // ... issued per step (3 or 1), in steps 0 .. PSTEPS-1
    constexpr int PSTEPS = (PPW + PPS - 1) / PPS;                // 4
    constexpr int NOFF = PPW > WPW ? PPW : WPW;
    static_assert(PPW <= 12 && PPS <= 3, "halo border masks: 5 bits x 6 pieces x 2 registers; one piece per quarter 1-3");
    // Per-lane source offsets (see conv3x3_sp_kernel).  halo piece pi = LDS rows 16 pi .. + 15, lane l -> row 16 pi + (l >> 2),
    // chunk position l & 3; weight piece pw = LDS rows 16 pw .. + 15 of the tile, LDS row (64 w + 16 t + m) = output channel
    // 64 w + 16 (m >> 2) + 4 t + (m & 3): a lane's accumulators of a 64-channel half are then 16 consecutive channels
    unsigned doff[NOFF], pm[2] = {0u, 0u};
#pragma unroll
    for (int i = 0; i < NOFF; ++i) {
        unsigned wv = 0, pv = 0;
        if (i < WPW) {
            const int r = (ridx * WPW + i) * 16 + (lane >> 2);
            const int wm = r & 15, wt = (r >> 4) & 3;
            const int wperm = (r & ~63) + 16 * (wm >> 2) + 4 * wt + (wm & 3);
            wv = (unsigned)((n0 + wperm) * a.Cin * 2 + (((lane & 3) ^ (2 * ((wm >> 2) & 1))) << 4));
        }
        if (i < PPW) {
            const int pi = ridx + 2 * i < P_PIECES ? ridx + 2 * i : P_PIECES - 1;
            const int row = pi * 16 + (lane >> 2);
            const int hy = row / HPW, hx = row - hy * HPW;
            pv = (unsigned)((hy * a.Win + hx) * a.Cin * 2 + (((lane & 3) ^ (2 * ((hx >> 2) & 1))) << 4));
            const unsigned m = (hy == 0 ? 1u : 0u) | (hy == HPW - 1 ? 2u : 0u) | (hx == 0 ? 4u : 0u) |
                               (hx == HPW - 1 ? 8u : 0u) | (row >= NROWS ? 16u : 0u);
            pm[i / 6] |= m << (5 * (i % 6));
        }
        doff[i] = w_role ? wv : pv;
    }
    const int tap_stride = a.Cout * a.Cin * 2;
    const unsigned x_lead = (unsigned)((a.Win + 1) * a.Cin * 2);
    const unsigned long xp = (unsigned long)a.x - x_lead, wpp = (unsigned long)a.wp;
    const u32x4 xrsrc = {(unsigned)xp, (unsigned)(xp >> 32) & 0xffffu, (unsigned)a.x_bytes + x_lead, 0x00020000u};
    const u32x4 wrsrc = {(unsigned)wpp, (unsigned)(wpp >> 32) & 0xffffu, (unsigned)a.w_bytes, 0x00020000u};

    auto tile_origin = [&](int pt, int& b, int& y0, int& x0) {
        b = pt / tiles_per_img;
        const int rem = pt - b * tiles_per_img;
        const int ty = rem / tiles_x;
        y0 = ty << 4;
        x0 = (rem - ty * tiles_x) << 4;
    };
    auto patch_scalar = [&](int pt, int c, unsigned& soff, unsigned& border) {
        int b, y0, x0;
        tile_origin(pt, b, y0, x0);
        const int sy = UPS ? y0 >> 1 : y0, sx = UPS ? x0 >> 1 : x0;
        soff = (unsigned)((((b * a.Hin + sy) * a.Win + sx) * a.Cin) * 2 + c * 64);
        border = (y0 == 0 ? 1u : 0u) | (y0 + 16 == a.Hout ? 2u : 0u) | (x0 == 0 ? 4u : 0u) | (x0 + 16 == a.Wout ? 8u : 0u) | 16u;
    };
    auto dma_patch = [&](int i, int buf, unsigned soff, unsigned border) {     // halo waves only
        const int pi = ridx + 2 * i < P_PIECES ? ridx + 2 * i : P_PIECES - 1;
        const bool ok = (pm[i / 6] & (border << (5 * (i % 6)))) == 0u;
        lds_dma16(xrsrc, ok ? doff[i] : 0x80000000u, soff, lds0 + (unsigned)(buf * P_BYTES + pi * 1024));
    };
    auto dma_w_piece = [&](int c, int tap, int buf, int i) {                     // weight waves only
        lds_dma16(wrsrc, doff[i], (unsigned)(tap * tap_stride + c * 64),
                  lds0 + (unsigned)(2 * P_BYTES + buf * W_BYTES + (ridx * WPW + i) * 1024));
    };

    f32x4 acc[NI][TPX];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < TPX; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int r16 = lane & 15, q = lane >> 4;
    const int aoff = r16 * 64 + ((q ^ (2 * ((r16 >> 2) & 1))) << 4);
    int boff[3];
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
        const int colx = UPS ? ((r16 + kw - 1) >> 1) + 1 : r16 + kw;
        const int row0 = UPS ? (wave_py >> 1) : wave_py;
        boff[kw] = (row0 * HPW + colx) * 64 + ((q ^ (2 * ((colx >> 2) & 1))) << 4);
    }
    constexpr int NR = UPS ? TPX / 2 + 2 : TPX + 2;     // halo rows a wave needs per filter column
    auto rowidx = [](int j, int kh) { return UPS ? ((j + kh - 1) >> 1) + 1 : j + kh; };
    constexpr int NSLOT = UPS ? 6 : 9;
    static_assert((3 * NR) % NSLOT == 0, "a slice's rows must map to the same slots in every slice");
    // A tiles are double-buffered at QUARTER granularity: quarter qi's tiles sit in slot set qi & 1 and are read from LDS
    // while quarter qi - 1 multiplies (NI / 2 slots, not NI: the registers BN = 128 does not have)
    constexpr int NAS = 2 * QT;
    bf16x8 af[NAS], bm[NSLOT];
    auto load_a = [&](int wb, int i) {                                   // tile i's fragment of weight buffer wb
        af[i % NAS] = *reinterpret_cast<const bf16x8*>(w_lds + wb * W_BYTES + aoff + i * 16 * 64);
    };
    auto load_a_quarter = [&](int wb, int qi) {
#pragma unroll
        for (int i = qi * QT; i < (qi + 1) * QT; ++i) load_a(wb, i);
    };
    auto load_b = [&](int pb, int kh, int kw, int part) {               // the rows tap (kh, kw) uses and (kh - 1, kw) did not,
        const unsigned char* pbuf = patch_lds + pb * P_BYTES;           // every third of them (part 0..2, or all: part < 0)
        int cnt = 0;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            bool used = false, prev = false;
#pragma unroll
            for (int j = 0; j < TPX; ++j) {
                used = used || rowidx(j, kh) == r;
                prev = prev || (kh > 0 && rowidx(j, kh - 1) == r);
            }
            if (used && !prev && (part < 0 || cnt++ % 3 == part))
                bm[(kw * NR + r) % NSLOT] = *reinterpret_cast<const bf16x8*>(pbuf + boff[kw] + r * HPW * 64);
        }
    };
    auto mfma_quarter = [&](int kh, int kw, int qi) {
#pragma unroll
        for (int i = qi * QT; i < (qi + 1) * QT; ++i)
#pragma unroll
            for (int j = 0; j < TPX; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i % NAS], bm[(kw * NR + rowidx(j, kh)) % NSLOT], acc[i][j], 0, 0, 0);
    };

    float* const bias_lds = reinterpret_cast<float*>(dsm + 2 * P_BYTES + 3 * W_BYTES);   // [BN]
    if (tid < BN) bias_lds[tid] = a.bias ? a.bias[n0 + tid] : 0.f;

    // the fused epilogues' per-lane sums do not live across the main loop: a tile's 16 values per half are reduced over the
    // 16 pixel lanes by a transposing butterfly (lane r16 ends up with channel r16's total) and added to ONE register
    auto transpose_reduce16 = [&](const float (&v)[16]) {
        float w8[8], w4[4], w2[2];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const bool hi = (r16 & 8) != 0;
            w8[k] = (hi ? v[k + 8] : v[k]) + __shfl_xor(hi ? v[k] : v[k + 8], 8);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const bool hi = (r16 & 4) != 0;
            w4[k] = (hi ? w8[k + 4] : w8[k]) + __shfl_xor(hi ? w8[k] : w8[k + 4], 4);
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const bool hi = (r16 & 2) != 0;
            w2[k] = (hi ? w4[k + 2] : w4[k]) + __shfl_xor(hi ? w4[k] : w4[k + 2], 2);
        }
        const bool hi = (r16 & 1) != 0;
        return (hi ? w2[1] : w2[0]) + __shfl_xor(hi ? w2[0] : w2[1], 1);
    };
    float cs_acc[NH], st1_acc[NH], st2_acc[NH];            // channel (64 hw + 16 q + r16)'s running totals
#pragma unroll
    for (int hw = 0; hw < NH; ++hw) { cs_acc[hw] = 0.f; st1_acc[hw] = 0.f; st2_acc[hw] = 0.f; }
    int st_b = -1;
    auto stats_flush = [&]() {
#pragma unroll
        for (int hw = 0; hw < NH; ++hw) {
            const float v1 = st1_acc[hw], v2 = st2_acc[hw];
            st1_acc[hw] = 0.f; st2_acc[hw] = 0.f;
            unsigned long long* dst = reinterpret_cast<unsigned long long*>(a.stats) +
                                      ((long)st_b * a.Cout + n0 + 64 * hw + 16 * q + r16) * 2;
            if (!(fabsf(v1) < 1e9f) || !(v2 < 1e9f)) {          // (conv3x3_sp_kernel: a non-finite sum must not come out finite)
                atomicMax(reinterpret_cast<long long*>(dst) + 1, 0x7fffffffffffffffLL);
            } else {
                atomicAdd(dst, (unsigned long long)__double2ll_rn((double)v1 * 4294967296.0));
                atomicAdd(dst + 1, (unsigned long long)__double2ll_rn((double)v2 * 4294967296.0));
            }
        }
    };

    auto epilogue = [&](int pt) {       // bias -> residual -> leaky ReLU (or a given mask) -> bf16 NHWC, then clear the accumulators
        if (KO == 9 && a.B >= 0) {
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < TPX; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            return;
        }
        int b, y0, x0;
        tile_origin(pt, b, y0, x0);
        const float cw = MASKED && a.colsum && a.row_scale ? a.row_scale[b] : 1.f;
        const float cw2 = MASKED && a.y2 ? a.row_scale2[b] : 0.f;
        if (STATS) st_b = b;
#pragma unroll
        for (int hw = 0; hw < NH; ++hw) {
            const int co = n0 + 64 * hw + 16 * q;            // this lane's 16 consecutive output channels of the half
            const bool act = co < a.lrelu_ch;
            if (a.pool_sum) {
                const int Hp = a.Hout >> 1, Wp = a.Wout >> 1;
#pragma unroll
                for (int j = 0; j < TPX; j += 2) {
                    float v[16];
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float t = acc[4 * hw + i][j][r] + acc[4 * hw + i][j + 1][r];
                            const int o = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0xB1, 0xF, 0xF, false);
                            v[4 * i + r] = t + __builtin_bit_cast(float, o);
                        }
                    if ((r16 & 1) == 0) {
                        const int yy = (y0 + wave_py + j) >> 1, xx = (x0 + r16) >> 1;
                        const long o = (((long)b * Hp + yy) * Wp + xx) * a.Cout + co;
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            u32x4 out = {pack_bf16x2(v[8 * h + 0], v[8 * h + 1]), pack_bf16x2(v[8 * h + 2], v[8 * h + 3]),
                                         pack_bf16x2(v[8 * h + 4], v[8 * h + 5]), pack_bf16x2(v[8 * h + 6], v[8 * h + 7])};
                            *reinterpret_cast<u32x4*>(a.y + o + 8 * h) = out;
                        }
                    }
                }
            } else {
            float ps[16];                   // ypool: running 2x2 sums of the bf16-rounded outputs of a row pair
            float tl0[(MASKED || STATS) ? 16 : 1], tl1[STATS ? 16 : 1];     // this tile's sums
#pragma unroll
            for (int k2 = 0; k2 < ((MASKED || STATS) ? 16 : 1); ++k2) tl0[k2] = 0.f;
#pragma unroll
            for (int k2 = 0; k2 < (STATS ? 16 : 1); ++k2) tl1[k2] = 0.f;
#pragma unroll
            for (int j = 0; j < TPX; ++j) {
                const int yy = y0 + wave_py + j, xx = x0 + r16;
                const long o = (((long)b * a.Hout + yy) * a.Wout + xx) * a.Cout + co;
                float v[16];
                if (a.bias) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const f32x4 bq = *reinterpret_cast<const f32x4*>(bias_lds + 64 * hw + 16 * q + 4 * i);
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[4 * i + r] = acc[4 * hw + i][j][r] + bq[r];
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[4 * i + r] = acc[4 * hw + i][j][r];
                }
                if (a.resid) {
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const u32x4 rr = *reinterpret_cast<const u32x4*>(a.resid + o + 8 * h);
#pragma unroll
                        for (int w2 = 0; w2 < 4; ++w2) {
                            v[8 * h + 2 * w2] += bf16_lo(rr[w2]);
                            v[8 * h + 2 * w2 + 1] += bf16_hi(rr[w2]);
                        }
                    }
                }
                if (act) {                  // 0 <= slope <= 1 (checked by the launcher): max(v, slope v) IS the leaky ReLU
#pragma unroll
                    for (int k2 = 0; k2 < 16; ++k2) {
                        const float sv = v[k2] * a.slope;
                        asm("v_max_f32 %0, %1, %2" : "=v"(v[k2]) : "v"(v[k2]), "v"(sv));
                    }
                }
                u32x4 mk[MASKED ? 2 : 1];
                if (MASKED) {
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const u32x4 mm = *reinterpret_cast<const u32x4*>(a.mask_y + o + 8 * h);
                        mk[h] = mm;
#pragma unroll
                        for (int w2 = 0; w2 < 4; ++w2) {
                            v[8 * h + 2 * w2] = bf16_lo(mm[w2]) > 0.f ? v[8 * h + 2 * w2] : v[8 * h + 2 * w2] * a.slope;
                            v[8 * h + 2 * w2 + 1] = bf16_hi(mm[w2]) > 0.f ? v[8 * h + 2 * w2 + 1] : v[8 * h + 2 * w2 + 1] * a.slope;
                        }
                    }
                }
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    u32x4 out = {pack_bf16x2(v[8 * h + 0], v[8 * h + 1]), pack_bf16x2(v[8 * h + 2], v[8 * h + 3]),
                                 pack_bf16x2(v[8 * h + 4], v[8 * h + 5]), pack_bf16x2(v[8 * h + 6], v[8 * h + 7])};
                    *reinterpret_cast<u32x4*>(a.y + o + 8 * h) = out;
                    if (MASKED) {           // column sums of what was stored (the rounded values, as the separate pass took them)
#pragma unroll
                        for (int w2 = 0; w2 < 4; ++w2) {
                            tl0[MASKED ? 8 * h + 2 * w2 : 0] += cw * bf16_lo(out[w2]);
                            tl0[MASKED ? 8 * h + 2 * w2 + 1 : 0] += cw * bf16_hi(out[w2]);
                        }
                        if (a.y2) {         // rgbd_axpy_rows_bf16 of (what was stored, the activation tile): the injection operand
                            u32x4 o2;
#pragma unroll
                            for (int w2 = 0; w2 < 4; ++w2)
                                o2[w2] = pack_bf16x2(bf16_lo(out[w2]) + cw2 * bf16_lo(mk[h][w2]),
                                                     bf16_hi(out[w2]) + cw2 * bf16_hi(mk[h][w2]));
                            *reinterpret_cast<u32x4*>(a.y2 + o + 8 * h) = o2;
                        }
                    }
                    if (STATS) {
#pragma unroll
                        for (int w2 = 0; w2 < 4; ++w2) {
                            const float lo = bf16_lo(out[w2]), hi = bf16_hi(out[w2]);
                            tl0[STATS ? 8 * h + 2 * w2 : 0] += lo;
                            tl1[STATS ? 8 * h + 2 * w2 : 0] += lo * lo;
                            tl0[STATS ? 8 * h + 2 * w2 + 1 : 0] += hi;
                            tl1[STATS ? 8 * h + 2 * w2 + 1 : 0] += hi * hi;
                        }
                    }
                    if (a.ypool) {          // the block's downscale2x (rescale.py:12-13) of what was just stored
#pragma unroll
                        for (int w2 = 0; w2 < 4; ++w2) {
                            const float lo = bf16_lo(out[w2]), hi = bf16_hi(out[w2]);
                            ps[8 * h + 2 * w2] = (j & 1) ? ps[8 * h + 2 * w2] + lo : lo;
                            ps[8 * h + 2 * w2 + 1] = (j & 1) ? ps[8 * h + 2 * w2 + 1] + hi : hi;
                        }
                    }
                }
                if (a.ypool && (j & 1)) {
#pragma unroll
                    for (int k2 = 0; k2 < 16; ++k2) {
                        const int o2 = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, ps[k2]), 0xB1, 0xF, 0xF, false);
                        ps[k2] = 0.25f * (ps[k2] + __builtin_bit_cast(float, o2));
                    }
                    if ((r16 & 1) == 0) {
                        const long op = (((long)b * (a.Hout >> 1) + (yy >> 1)) * (a.Wout >> 1) + (xx >> 1)) * a.Cout + co;
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            u32x4 out = {pack_bf16x2(ps[8 * h + 0], ps[8 * h + 1]), pack_bf16x2(ps[8 * h + 2], ps[8 * h + 3]),
                                         pack_bf16x2(ps[8 * h + 4], ps[8 * h + 5]), pack_bf16x2(ps[8 * h + 6], ps[8 * h + 7])};
                            *reinterpret_cast<u32x4*>(a.ypool + op + 8 * h) = out;
                        }
                    }
                }
            }
            if constexpr (MASKED) cs_acc[hw] += transpose_reduce16(tl0);
            if constexpr (STATS) {
                st1_acc[hw] += transpose_reduce16(tl0);
                st2_acc[hw] += transpose_reduce16(tl1);
            }
            }
            // (cleared in ONE place behind both forms: with a clear in each branch the compiler merges the branch tails
            // through a phi of accumulator ADDRESSES, which keeps those accumulators in scratch memory for the whole kernel)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < TPX; ++j) acc[4 * hw + i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if constexpr (STATS) stats_flush();          // per tile: grid-independent statistics (see conv3x3_sp_kernel)
    };

    // ---- prologue: halo patch of slice 0, weight tiles of steps 0-1 and the first pieces of step 2's (step t multiplies
    //      filter tap (kh, kw) = (t % 3, t / 3), i.e. weight image 3 (t % 3) + t / 3), then quarter 0's A tiles and the halo
    //      rows of step 0
    constexpr int WFIRST = WPW == 4 ? 2 : 1;          // a tile's pieces go out in three gaps: WFIRST, then one and one (or one, none)
    if (w_role) {
#pragma unroll
        for (int i = 0; i < WPW; ++i) dma_w_piece(0, 0, 0, i);
#pragma unroll
        for (int i = 0; i < WPW; ++i) dma_w_piece(0, 3, 1, i);
#pragma unroll
        for (int i = 0; i < WFIRST; ++i) dma_w_piece(0, 6, 2, i);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WFIRST) : "memory");
    } else {
        unsigned soff, border;
        patch_scalar(pt_begin, 0, soff, border);
#pragma unroll
        for (int i = 0; i < PPW; ++i) dma_patch(i, 0, soff, border);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    load_a_quarter(0, 0);
    load_b(0, 0, 0, -1);

    // One K step t: quarters 0-2 multiply while the NEXT quarter's A tiles of weight tile t are read; the barrier B_t sits
    // between quarters 2 and 3: by then every read of tile t has been issued and retired (lgkmcnt(0)), so its buffer may take
    // tile t + 3, and tile t + 1 (needed from quarter 3 on: quarter 0's tiles of step t + 1) has landed in every weight wave
    // (vmcnt(WPW): all but the youngest tile, t + 2); at t = 8 the halo waves have retired the next patch.  Quarter 3 runs
    // behind B_t with the reads of step t + 1: its first A tiles and its new halo rows.  DMA issue points (a piece holds its
    // wave for ~60 cycles; the other workgroup's MFMAs cover it): behind quarter 3 the first pieces of tile t + 3, behind
    // quarters 0 and 1 of the next step the rest; the next patch's pieces in the same gaps of steps 0-3.
    int c = 0, pt = pt_begin;
    bool w_waited = false;             // weight waves: tile 1 of this slice was already waited for (in front of an epilogue)
    unsigned long long stamp_sum[4] = {0, 0, 0, 0};     // KO 10: cycles in quarters 0-2 / waits + barrier / quarter 3 + DMA issue / epilogue
    for (int g = 0; g < g_total; ++g) {
        const int c_next = c + 1 == nc ? 0 : c + 1;
        const bool last = g + 1 >= g_total;
        const int pt_next = (c_next == 0 && !last) ? pt + 1 : pt;
        unsigned nsoff, nborder;
        patch_scalar(pt_next, c_next, nsoff, nborder);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int kh = t % 3, kw = t / 3;
            const int tn = (t + 1) % 9, khn = tn % 3, kwn = tn / 3;
            unsigned long long ts0 = 0, ts1 = 0, ts2 = 0;
            if (KO == 10) ts0 = __builtin_amdgcn_s_memtime();
#pragma unroll
            for (int qi = 0; qi < 4; ++qi) {
                if (qi == 3) {
                    if (KO == 10) ts1 = __builtin_amdgcn_s_memtime();
                    if (w_role) {
                        if (!(t == 0 && w_waited)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WPW) : "memory");
                    } else if (t == 8) {
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    RGBD_PP_BARRIER();
                    if (KO == 10) ts2 = __builtin_amdgcn_s_memtime();
                    if (KO != 3 && KO != 6) {
                        load_a_quarter(tn % 3, 0);
                        load_b(t == 8 ? (g + 1) & 1 : g & 1, khn, kwn, -1);
                    }
                } else if (KO != 3 && KO != 6) {
                    load_a_quarter(t % 3, qi + 1);
                }
                mfma_quarter(kh, kw, qi);
#pragma unroll
                for (int k_ = 0; k_ < QT * TPX; ++k_) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (qi != 2) {
                    const int gap = qi == 3 ? 0 : qi + 1;                 // 0: behind B_t, 1 / 2: the next two gaps of the same tile
                    if (w_role && KO != 1 && KO != 6) {
                        // gap 0 opens tile t + 3 (into tile t's buffer); gaps 1, 2 of THIS step finish tile t + 2
                        const int tt = gap == 0 ? t + 3 : t + 2;
                        const int cw_ = tt >= 9 ? c_next : c, tapw = 3 * ((tt % 9) % 3) + (tt % 9) / 3;
                        if (gap == 0) {
#pragma unroll
                            for (int k_ = 0; k_ < WFIRST; ++k_) dma_w_piece(cw_, tapw, tt % 3, k_);
                        } else if (WFIRST + gap - 1 < WPW) {
                            dma_w_piece(cw_, tapw, tt % 3, WFIRST + gap - 1);
                        }
                    } else if (!w_role && t < PSTEPS && KO != 2 && KO != 6) {
                        const int pi_ = qi == 3 ? 2 : qi;                 // gaps behind quarters 0, 1, 3 of steps 0 .. PSTEPS-1
                        if (pi_ < PPS && t * PPS + pi_ < PPW) dma_patch(t * PPS + pi_, (g + 1) & 1, nsoff, nborder);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (KO == 10) {
                const unsigned long long ts3 = __builtin_amdgcn_s_memtime();
                stamp_sum[0] += ts1 - ts0; stamp_sum[1] += ts2 - ts1; stamp_sum[2] += ts3 - ts2;
            }
        }
        w_waited = false;
        if (c_next == 0) {             // last slice of this pixel tile: write it out.  The weight waves first retire tile 1 of
                                       // the next slice (all but the pieces of tile 2 issued so far): behind the epilogue's
                                       // stores a counted wait would also wait for those
            if (w_role) {
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WFIRST) : "memory");
                w_waited = true;
            }
            unsigned long long te0 = 0;
            if (KO == 10) te0 = __builtin_amdgcn_s_memtime();
            epilogue(pt);
            if (KO == 10) stamp_sum[3] += __builtin_amdgcn_s_memtime() - te0;
        }
        c = c_next;
        pt = pt_next;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // no DMA may land in LDS after the workgroup has gone
#ifdef RGBD_DEBUG_BUILD
    if (KO == 10 && a.partial && lane == 0) {
        unsigned* o = reinterpret_cast<unsigned*>(a.partial) + 8 * 1024 + (blockIdx.x * 4 + wid) * 4;
#pragma unroll
        for (int k_ = 0; k_ < 4; ++k_) o[k_] = (unsigned)stamp_sum[k_];
    }
#endif
#ifdef RGBD_DEBUG_BUILD
    if (a.partial && tid == 0) {        // census (scripts/dw_census.py): where and when this workgroup ran
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
        unsigned* o = reinterpret_cast<unsigned*>(a.partial) + 8 * blockIdx.x;
        o[0] = hw; o[1] = xcc; o[2] = (unsigned)census_t0; o[3] = (unsigned)(census_t0 >> 32);
        o[4] = (unsigned)t1; o[5] = (unsigned)(t1 >> 32); o[6] = (unsigned)g_total; o[7] = blockIdx.x;
    }
#endif
    if (MASKED && a.colsum) {
        __syncthreads();
        float* const red = reinterpret_cast<float*>(dsm);              // [4 waves][BN]
#pragma unroll
        for (int hw = 0; hw < NH; ++hw) red[wid * BN + 64 * hw + 16 * q + r16] = cs_acc[hw];
        __syncthreads();
        if (tid < BN) {
            const float t = (red[tid] + red[BN + tid]) + (red[2 * BN + tid] + red[3 * BN + tid]);
            atomicAdd(a.colsum + n0 + tid, t);
        }
    }
}

#endif  // RGBD_DEBUG_BUILD (conv3x3_dw_kernel)

// ------------------------------------------------------------------------------------------------ wgrad
struct WgradArgs {
    const unsigned short* x;
    const unsigned short* dy;
    float* dwp;
    int B, H, W, Cin, Cout;
    int PH, PW, lgPW;      // patch of output pixels (powers of two)
    int npx, npy;          // patches per image along x / y
    int total_patches, patches_per_wg;
    int x_bytes, y_bytes;
    int ups;               // x is (B,H/2,W/2,Cin) and is read through the nearest-2x upsampling (rescale.py:4-5)
    int ko;                // debug library, timing only (wrong results): 1 no LDS-DMA in the loop, 2 no fragment reads in the loop
};

__device__ __forceinline__ s16x4 lds_tr16(const unsigned char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (s16x4 __attribute__((address_space(3)))*)(const_cast<unsigned char*>(p)));
}

template <int NT, bool FAST>  // filter taps: 9 (3x3, pad 1) or 1 (1x1); FAST: 8x16 patches (images >= 16x16)
__device__ __forceinline__ void conv_wgrad_body(const WgradArgs& a, const int split_idx, const int ci_tile,
                                                const int co_tile) {
    // 8 waves, one workgroup per CU.  Wave w owns the 32x32 sub-tile (w>>1 & 1, w & 1) of the 64x64 (co, ci) tile for
    // one half of the filter taps (waves 0-3: taps 0..4, waves 4-7: taps 5..8; waves w and w+4 share a SIMD, so every
    // SIMD carries all nine taps).  LDS is double buffered: patch p+1 is written while patch p is multiplied, patch
    // p+2 is in flight in registers; one barrier per 8x16-pixel patch.
    constexpr int HALO = NT == 9 ? 1 : 0;
    constexpr int KW = NT == 9 ? 3 : 1;
    constexpr int TG0 = NT == 9 ? 5 : 1;               // taps of wave group 0
    constexpr int MAX_X_ROWS = (8 + 2 * HALO) * (16 + 2 * HALO);
    constexpr int X_BYTES = MAX_X_ROWS * 128, Y_BYTES = 128 * 128;
    constexpr int XP = (MAX_X_ROWS * 8 + 511) / 512;   // 16-byte pieces of the X halo patch per thread (3 / 2)
    constexpr int YP = 2;                              // pieces of the dY patch per thread (128 rows)
    extern __shared__ __attribute__((aligned(16))) unsigned char wsm[];   // [2][X_BYTES + Y_BYTES]

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int ci0 = ci_tile * 64, co0 = co_tile * 64;
    const int wc = (wid >> 1) & 1, wi = wid & 1;       // wave -> (co half, ci half) of the 64x64 tile
    const int tg = __builtin_amdgcn_readfirstlane(wid) >> 2;   // tap group (wave-uniform by construction)
    const int HPW = a.PW + 2 * HALO, HPH = a.PH + 2 * HALO;
    const int npix = a.PH * a.PW;
    const int xrows = HPH * HPW;
    const auto xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(a.x), 0, a.x_bytes, 0x00020000);
    const auto yrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(a.dy), 0, a.y_bytes, 0x00020000);

    f32x16 acc[TG0];
#pragma unroll
    for (int t = 0; t < TG0; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // transposed-read lane roles (cdna_hip_programming.md T10): within each 16-lane group, lane 4q+p supplies
    // the address of block row q (a pixel), columns 4p..4p+3 (channels); it receives column (lane & 15).
    const int g = lane >> 4, qq = (lane & 15) >> 2, pp = lane & 3;
    const int m_base = 16 * (g & 1), k_base = 8 * (g >> 1);
    const int a_col_bytes = (wc * 32 + m_base + 4 * pp) * 2;
    const int b_col_bytes = (wi * 32 + m_base + 4 * pp) * 2;

    // ---- staging plan: patch-independent parts of the source offsets / LDS destinations of this thread's pieces.
    //      Pieces outside the halo patch or outside the image load through an out-of-range buffer offset (zeros).
    int xrel[XP], xdst[XP], yrel[YP], ydst[YP], xch[XP];
    short xhy[XP], xhx[XP];
#pragma unroll
    for (int i = 0; i < XP; ++i) {
        const int pc = tid + 512 * i;
        const int row = pc >> 3, chunk = pc & 7;
        const int hy = row / HPW, hx = row - hy * HPW;
        const bool in = row < xrows;
        xhy[i] = in ? (short)(hy - HALO) : (short)-30000;
        xhx[i] = (short)(hx - HALO);
        xrel[i] = (((hy - HALO) * a.W + (hx - HALO)) * a.Cin + ci0 + chunk * 8) * 2;
        xch[i] = (ci0 + chunk * 8) * 2;
        // the 64-byte half of a row is swapped by bit 1 of the halo COLUMN (halo width is even, so row parity ==
        // column parity): four consecutive pixels x 64 B then cover all 64 banks once for ds_read_b64_tr_b16, and
        // the swizzle of a tap-shifted read depends only on the lane and the horizontal tap
        xdst[i] = in ? row * 128 + ((chunk ^ (((hx >> 1) & 1) << 2)) << 4) : -1;
    }
#pragma unroll
    for (int i = 0; i < YP; ++i) {
        const int pc = tid + 512 * i;
        const int row = pc >> 3, chunk = pc & 7;
        const int py = row >> a.lgPW, px = row & (a.PW - 1);
        const bool in = row < npix;
        yrel[i] = in ? ((py * a.W + px) * a.Cout + co0 + chunk * 8) * 2 : (int)0x80000000;
        ydst[i] = in ? X_BYTES + row * 128 + ((chunk ^ (((px >> 1) & 1) << 2)) << 4) : -1;
    }
    u32x4 Xr[XP], Yr[YP];
    const int per_img = a.npx * a.npy;
    auto load_patch = [&](int patch) {
        const int b = patch / per_img;
        const int rem = patch - b * per_img;
        const int pyi = rem / a.npx;
        const int y0 = pyi * a.PH, x0 = (rem - pyi * a.npx) * a.PW;
        const int xbase = ((b * a.H + y0) * a.W + x0) * a.Cin * 2;
        const int ybase = ((b * a.H + y0) * a.W + x0) * a.Cout * 2;
#pragma unroll
        for (int i = 0; i < XP; ++i) {
            const int yy = y0 + xhy[i], xx = x0 + xhx[i];
            const bool ok = (unsigned)yy < (unsigned)a.H && (unsigned)xx < (unsigned)a.W;
            // upsampled operand: the halo is cut out of the VIRTUAL (H,W) image, pixel (yy,xx) of which is source pixel
            // (yy/2, xx/2) -- the 4x larger tensor never exists (the address select is wave-uniform, the load is not
            // conditional)
            const int src = a.ups ? (((b * (a.H >> 1) + (yy >> 1)) * (a.W >> 1) + (xx >> 1)) * a.Cin) * 2 + xch[i]
                                  : xbase + xrel[i];
            Xr[i] = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, ok ? (unsigned)src : 0x80000000u, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < YP; ++i)
            Yr[i] = __builtin_amdgcn_raw_buffer_load_b128(yrsrc, (unsigned)yrel[i] + (unsigned)ybase, 0, 0);
    };
    auto store_patch = [&](int buf) {
        unsigned char* base = wsm + buf * (X_BYTES + Y_BYTES);
#pragma unroll
        for (int i = 0; i < XP; ++i)
            if (xdst[i] >= 0) *reinterpret_cast<u32x4*>(base + xdst[i]) = Xr[i];
#pragma unroll
        for (int i = 0; i < YP; ++i)
            if (ydst[i] >= 0) *reinterpret_cast<u32x4*>(base + ydst[i]) = Yr[i];
    };

    // fast-path lane constants: pixel column of this lane's two transposed reads and their swizzled byte offsets
    int fa[2], fb[3][2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int px = k_base + qq + 4 * h;                                   // 0..15 within the patch row
        fa[h] = px * 128 + (a_col_bytes ^ (((px >> 1) & 1) << 6));
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const int hx = px + kw;                                           // halo column of tap kw
            fb[kw][h] = hx * 128 + (b_col_bytes ^ (((hx >> 1) & 1) << 6));
        }
    }

    const int p_begin = split_idx * a.patches_per_wg;
    const int p_end = min(a.total_patches, p_begin + a.patches_per_wg);
    if (p_begin < p_end) {
        load_patch(p_begin);
        store_patch(0);
        load_patch(min(p_begin + 1, p_end - 1));
    }
    __syncthreads();
    for (int patch = p_begin; patch < p_end; ++patch) {
        const int cur = (patch - p_begin) & 1;
        const unsigned char* xs = wsm + cur * (X_BYTES + Y_BYTES);
        const unsigned char* ys = xs + X_BYTES;
        if (FAST) {
            // PW == 16: the 16 pixels of a K step are one patch row, so every LDS address is
            // (per-lane constant of the horizontal tap) + (compile-time row offset): no address VALU in the loop.
            // The tap group is wave-uniform; each group gets its own fully unrolled body (compile-time taps).
            auto body = [&](auto tgc) {
                constexpr int TG = decltype(tgc)::value;
#pragma unroll
                for (int kr = 0; kr < 8; ++kr) {                       // patch row = K step
                    const s16x4 a0 = lds_tr16(ys + fa[0] + kr * 16 * 128);
                    const s16x4 a1 = lds_tr16(ys + fa[1] + kr * 16 * 128);
                    const bf16x8 af = __builtin_bit_cast(bf16x8, __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
                    for (int tt = 0; tt < TG0; ++tt) {
                        constexpr int dummy = 0;
                        (void)dummy;
                        const int t = TG * TG0 + tt;
                        if (t < NT) {
                            const int kh = t / KW, kw = t - kh * KW;
                            const int ro = (kr + kh) * (16 + 2 * HALO) * 128;
                            const s16x4 b0 = lds_tr16(xs + fb[kw][0] + ro);
                            const s16x4 b1 = lds_tr16(xs + fb[kw][1] + ro);
                            const bf16x8 bfr = __builtin_bit_cast(bf16x8, __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7));
                            acc[tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bfr, acc[tt], 0, 0, 0);
                        }
                    }
                }
            };
            if (tg == 0) body(std::integral_constant<int, 0>{});
            else         body(std::integral_constant<int, 1>{});
        } else {
        for (int ks = 0; ks < npix; ks += 16) {
            const int pix0 = ks + k_base + qq, pix1 = pix0 + 4;
            const int px0 = pix0 & (a.PW - 1), px1 = pix1 & (a.PW - 1);
            const s16x4 a0 = lds_tr16(ys + pix0 * 128 + (a_col_bytes ^ (((px0 >> 1) & 1) << 6)));
            const s16x4 a1 = lds_tr16(ys + pix1 * 128 + (a_col_bytes ^ (((px1 >> 1) & 1) << 6)));
            const bf16x8 af = __builtin_bit_cast(bf16x8, __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7));
            const int h0 = (pix0 >> a.lgPW) * HPW + px0;   // halo row of tap (0,0)
            const int h1 = (pix1 >> a.lgPW) * HPW + px1;
#pragma unroll
            for (int tt = 0; tt < TG0; ++tt) {
                const int t = tg * TG0 + tt;              // wave-uniform tap index
                if (t < NT) {
                    const int kw = t % KW;
                    const int toff = (t / KW) * HPW + kw;
                    const int r0 = h0 + toff, r1 = h1 + toff;
                    const s16x4 b0 = lds_tr16(xs + r0 * 128 + (b_col_bytes ^ ((((px0 + kw) >> 1) & 1) << 6)));
                    const s16x4 b1 = lds_tr16(xs + r1 * 128 + (b_col_bytes ^ ((((px1 + kw) >> 1) & 1) << 6)));
                    const bf16x8 bfr = __builtin_bit_cast(bf16x8, __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7));
                    acc[tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bfr, acc[tt], 0, 0, 0);
                }
            }
        }
        }
        // patch+1 (registers) -> the other LDS buffer (last read one barrier ago), patch+2 -> registers
        store_patch(cur ^ 1);
        load_patch(min(patch + 2, p_end - 1));
        __syncthreads();
    }
    // ---- each workgroup stores its partial (tap, co, ci) tile into its own slab with plain stores (fp32 atomics
    //      run at ~1.3 TB/s chip-wide, plain stores at ~6 TB/s: MI355X_MICROARCH.md "Global float atomics");
    //      wgrad_reduce_kernel sums the slabs.  D[row = co][col = ci]: one register = two 128-byte row segments.
    const int col = lane & 31, rhalf = lane >> 5;
    float* slab = a.dwp + (long)split_idx * NT * a.Cout * a.Cin;
#pragma unroll
    for (int tt = 0; tt < TG0; ++tt) {
        const int t = tg * TG0 + tt;
        if (t < NT) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * rhalf;
                slab[((long)t * a.Cout + co0 + wc * 32 + row) * a.Cin + ci0 + wi * 32 + col] = acc[tt][r];
            }
        }
    }
}

// ---- 3x3 weight gradient on 8x16-pixel patches (images >= 8x16), the hot variant: every wave runs ALL nine taps.
// The same 64x64 (co, ci) tile per workgroup, the same LDS images and transposed reads as conv_wgrad_body, but
//   * waves 0-3 / 4-7 split the patch ROWS (rows 0-3 / 4-7) instead of the taps, each wave accumulating nine 32x32 tap
//     tiles (144 accumulator registers).  The X fragment of halo row r at horizontal tap kw serves vertical taps 0, 1, 2
//     of output rows r, r-1, r-2: kept in a three-row register window, one new halo row (3 fragments) per K step instead
//     of one fragment per tap -- 44 transposed reads per 36 MFMAs where the tap-split body needs 96 per 40, which had the
//     LDS pipe ~80% busy.  The two row halves meet once, at the end, through LDS.
//   * operands arrive by LDS-DMA (see conv3x3_sp_kernel) into three buffers: patch p+2 is requested when the barrier of
//     patch p has certified that buffer (p+2) % 3 is free, and retired two patches later by a counted vmcnt; no
//     staging registers, no ds_write pass, halo borders by out-of-range offsets chosen from per-lane border bits.
__device__ __forceinline__ void conv_wgrad9_body(const WgradArgs& a, const int split_idx, const int ci_tile, const int co_tile) {
    constexpr int HPW = 18, XROWS = 10 * HPW;
    constexpr int XPIECES = (XROWS * 128 + 1023) / 1024;     // 23
    constexpr int YPIECES = 16;
    constexpr int X_BYTES = XPIECES * 1024, BUF = (XPIECES + YPIECES) * 1024;
    constexpr int XPW = (XPIECES + 7) / 8, YPW = YPIECES / 8;   // pieces per wave per patch: 3 + 2
    extern __shared__ __attribute__((aligned(16))) unsigned char wsm[];   // [3][BUF]; at the end [2 halves] exchange
    const unsigned lds0 = (unsigned)(size_t)wsm;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ci0 = ci_tile * 64, co0 = co_tile * 64;
    const int wc = (wid >> 1) & 1, wi = wid & 1, rh = wid >> 2;   // (co half, ci half) of the tile, row half of the patch

    // transposed-read lane roles (cdna_hip_programming.md T10): see conv_wgrad_body
    const int g = lane >> 4, qq = (lane & 15) >> 2, pp = lane & 3;
    const int m_base = 16 * (g & 1), k_base = 8 * (g >> 1);
    const int a_col_bytes = (wc * 32 + m_base + 4 * pp) * 2;
    const int b_col_bytes = (wi * 32 + m_base + 4 * pp) * 2;
    int fa[2], fb[3][2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int px = k_base + qq + 4 * h;                                   // 0..15 within the patch row
        fa[h] = (rh * 4 * 16 + px) * 128 + (a_col_bytes ^ (((px >> 1) & 1) << 6));
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const int hx = px + kw;                                           // halo column of tap kw
            fb[kw][h] = (rh * 4 * HPW + hx) * 128 + (b_col_bytes ^ (((hx >> 1) & 1) << 6));
        }
    }

    // ---- DMA plan (per-lane source offsets, constant over the kernel).  X piece i of wave w = halo rows 8 pi .. + 7
    //      (pi = w + 8 i, clamped: duplicates carry identical bytes), dY piece j = pixel rows 8 (w + 8 j) .. + 7; lane l
    //      fills chunk position (l & 7) of row (l >> 3) with the source chunk the image's swizzle puts there.  Offsets
    //      are relative to the patch's top-left HALO pixel, against a descriptor based one halo row + one pixel before
    //      the tensor (never dereferenced: those lanes are masked by the border bits).
    const int Ws = a.ups ? a.W >> 1 : a.W, Hs = a.ups ? a.H >> 1 : a.H;
    unsigned xoff[XPW], yoff[YPW], xmask = 0u;
#pragma unroll
    for (int i = 0; i < XPW; ++i) {
        const int pi = wid + 8 * i < XPIECES ? wid + 8 * i : XPIECES - 1;
        const int row = pi * 8 + (lane >> 3);
        const int hy = row / HPW, hx = row - hy * HPW;
        const int chunk = (lane & 7) ^ (((hx >> 1) & 1) << 2);
        const int sy = a.ups ? ((hy - 1) >> 1) + 1 : hy, sx = a.ups ? ((hx - 1) >> 1) + 1 : hx;
        xoff[i] = (unsigned)((sy * Ws + sx) * a.Cin * 2 + chunk * 16);
        const unsigned m = (hy == 0 ? 1u : 0u) | (hy == 9 ? 2u : 0u) | (hx == 0 ? 4u : 0u) | (hx == HPW - 1 ? 8u : 0u) |
                           (row >= XROWS ? 16u : 0u);
        xmask |= m << (5 * i);
    }
#pragma unroll
    for (int j = 0; j < YPW; ++j) {
        const int row = (wid + 8 * j) * 8 + (lane >> 3);
        const int py = row >> 4, px = row & 15;
        yoff[j] = (unsigned)((py * a.W + px) * a.Cout * 2 + (((lane & 7) ^ (((px >> 1) & 1) << 2)) << 4));
    }
    const unsigned x_lead = (unsigned)((Ws + 1) * a.Cin * 2);
    const unsigned long xp = (unsigned long)a.x - x_lead, yp = (unsigned long)a.dy;
    const u32x4 xrsrc = {(unsigned)xp, (unsigned)(xp >> 32) & 0xffffu, (unsigned)a.x_bytes + x_lead, 0x00020000u};
    const u32x4 yrsrc = {(unsigned)yp, (unsigned)(yp >> 32) & 0xffffu, (unsigned)a.y_bytes, 0x00020000u};
    const int per_img = a.npx * a.npy;
    // scalar part of a patch's sources (computed once per patch, used by five DMAs spread over the next patch's rows)
    struct PatchSrc { unsigned xs_off, ys_off, border, base; };
    auto patch_src = [&](int patch, int buf) {
        const int b = patch / per_img;
        const int rem = patch - b * per_img;
        const int pyi = rem / a.npx;
        const int y0 = pyi * 8, x0 = (rem - pyi * a.npx) * 16;
        const int sy0 = a.ups ? y0 >> 1 : y0, sx0 = a.ups ? x0 >> 1 : x0;
        PatchSrc ps;
        ps.xs_off = (unsigned)((((b * Hs + sy0) * Ws + sx0) * a.Cin + ci0) * 2);
        ps.ys_off = (unsigned)((((b * a.H + y0) * a.W + x0) * a.Cout + co0) * 2);
        ps.border = (y0 == 0 ? 1u : 0u) | (y0 + 8 == a.H ? 2u : 0u) | (x0 == 0 ? 4u : 0u) | (x0 + 16 == a.W ? 8u : 0u) | 16u;
        ps.base = lds0 + (unsigned)(buf * BUF);
        return ps;
    };
#ifdef RGBD_DEBUG_BUILD
    const int ko = a.ko;
#else
    constexpr int ko = 0;
#endif
    auto issue_piece = [&](const PatchSrc& ps, int idx) {     // idx 0 .. XPW-1: X pieces, XPW .. XPW+YPW-1: dY pieces
        if (ko & 1) return;
        if (idx < XPW) {
            const int pi = wid + 8 * idx < XPIECES ? wid + 8 * idx : XPIECES - 1;
            const bool ok = (xmask & (ps.border << (5 * idx))) == 0u;
            lds_dma16(xrsrc, ok ? xoff[idx] : 0x80000000u, ps.xs_off, ps.base + (unsigned)(pi * 1024));
        } else {
            const int j = idx - XPW;
            lds_dma16(yrsrc, yoff[j], ps.ys_off, ps.base + (unsigned)(X_BYTES + (wid + 8 * j) * 1024));
        }
    };
    constexpr int NPIECE = XPW + YPW;                          // 5 per wave per patch

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const int p_begin = split_idx * a.patches_per_wg;
    const int p_end = min(a.total_patches, p_begin + a.patches_per_wg);
    const int np = p_end - p_begin;
    bool in_loop = false;
    auto frag = [&](const unsigned char* p0, const unsigned char* p1) {
        if ((ko & 2) && in_loop) {                  // knock-out: a cheap register-only stand-in for the fragment
            const bf16x8 z = __builtin_bit_cast(bf16x8, u32x4{(unsigned)(size_t)p0, (unsigned)(size_t)p1, 0x3f803f80u, 0x3f803f80u});
            return z;
        }
        const s16x4 v0 = lds_tr16(p0), v1 = lds_tr16(p1);
        return __builtin_bit_cast(bf16x8, __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7));
    };
    auto clampp = [&](int k) { return p_begin + (k < np ? k : np - 1); };     // patches past the end: duplicates of the last
    // Schedule.  The barrier of patch k sits between its rows 2 and 3 (B_k): every LDS read of patch k has been issued
    // by then (the window runs one row ahead), so B_k frees buffer k % 3 for patch k + 3 and publishes patch k + 1, whose
    // first fragments are read under the MFMAs of row 3.  The five DMAs of patch k + 3 go out one per row: two in row 3
    // of patch k, three in rows 0-2 of patch k + 1; at B_k a wave's five youngest pieces are therefore those of patch
    // k + 2 and vmcnt(5) retires patch k + 1.
    bf16x8 af[2], brow[3][3];                              // dY fragment of row j (two sets), X window [halo row % 3][kw]
    PatchSrc pend = patch_src(p_begin, 0);
    if (np > 0) {
#pragma unroll
        for (int i = 0; i < NPIECE; ++i) issue_piece(pend, i);
        pend = patch_src(clampp(1), 1);
#pragma unroll
        for (int i = 0; i < NPIECE; ++i) issue_piece(pend, i);
        pend = patch_src(clampp(2), 2);
        issue_piece(pend, 0);
        issue_piece(pend, 1);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPIECE + 2) : "memory");
    }
    __syncthreads();
    if (np > 0) {
        af[0] = frag(wsm + X_BYTES + fa[0], wsm + X_BYTES + fa[1]);
#pragma unroll
        for (int l = 0; l < 3; ++l)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) brow[l][kw] = frag(wsm + fb[kw][0] + l * HPW * 128, wsm + fb[kw][1] + l * HPW * 128);
    }
    int bcur = 0, bnext = 1;                               // buffer of patch k / k + 1 (rotating: no k % 3 in the loop)
    in_loop = true;
#pragma unroll 1
    for (int k = 0; k < np; ++k) {
        const unsigned char* xs = wsm + bcur * BUF;
        const unsigned char* ys = xs + X_BYTES;
        const unsigned char* xn = wsm + bnext * BUF;
        const unsigned char* yn = xn + X_BYTES;
#pragma unroll
        for (int j = 0; j < 4; ++j) {                      // this wave's patch rows 4 rh + j; halo rows j .. j + 2 of its half
            if (j == 3) {
                // ---- B_k
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPIECE) : "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                RGBD_PP_BARRIER();
                pend = patch_src(clampp(k + 3), bcur);
            }
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
#pragma unroll
                for (int kw = 0; kw < 3; ++kw)
                    acc[kh * 3 + kw] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[j & 1], brow[(j + kh) % 3][kw],
                                                                               acc[kh * 3 + kw], 0, 0, 0);
                if (j < 3) {
                    if (kh == 0) {                         // halo row j + 3 replaces row j; dY row j + 1; one DMA
#pragma unroll
                        for (int kw = 0; kw < 3; ++kw)
                            brow[j % 3][kw] = frag(xs + fb[kw][0] + (j + 3) * HPW * 128, xs + fb[kw][1] + (j + 3) * HPW * 128);
                        af[(j + 1) & 1] = frag(ys + fa[0] + (j + 1) * 16 * 128, ys + fa[1] + (j + 1) * 16 * 128);
                        __builtin_amdgcn_sched_barrier(0);
                        issue_piece(pend, 2 + j);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                } else {                                   // row 3: the next patch's first fragments, two DMAs
                    if (kh == 0) af[0] = frag(yn + fa[0], yn + fa[1]);
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw)
                        brow[kh][kw] = frag(xn + fb[kw][0] + kh * HPW * 128, xn + fb[kw][1] + kh * HPW * 128);
                    if (kh < 2) {
                        __builtin_amdgcn_sched_barrier(0);
                        issue_piece(pend, kh);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
        }
        const int bfree = bcur;
        bcur = bnext;
        bnext = 3 - bfree - bnext;
    }
    // ---- the two row halves meet through LDS: the upper half hands over taps 0-4 and takes 5-8, then every wave stores
    //      its taps into the workgroup's slab (plain stores; wgrad_reduce sums the slabs)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // duplicates of the last patch: nothing may land after this point
    __syncthreads();
    float* const ex = reinterpret_cast<float*>(wsm);
    const int sub = wc * 2 + wi;
    constexpr int GIVE0 = 5;                                // taps [0, 5) end up in the lower-half waves
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const bool give = rh == 1 ? t < GIVE0 : t >= GIVE0;
        if (give) {
#pragma unroll
            for (int r = 0; r < 16; ++r) ex[((t * 16 + r) * 4 + sub) * 64 + lane] = acc[t][r];
        }
    }
    __syncthreads();
    const int col = lane & 31, rhalf = lane >> 5;
    float* slab = a.dwp + (long)split_idx * 9 * a.Cout * a.Cin;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const bool keep = rh == 0 ? t < GIVE0 : t >= GIVE0;
        if (keep) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * rhalf;
                slab[((long)t * a.Cout + co0 + wc * 32 + row) * a.Cin + ci0 + wi * 32 + col] =
                    acc[t][r] + ex[((t * 16 + r) * 4 + sub) * 64 + lane];
            }
        }
    }
}
constexpr int WGRAD9_LDS = 9 * 16 * 4 * 64 * 4;            // the final exchange (147456 B) exceeds the three buffers

#ifdef RGBD_DEBUG_BUILD
__global__ __launch_bounds__(512, 2) void conv_wgrad_tapsplit_kernel(WgradArgs a) {    // A/B reference (variant 3)
    conv_wgrad_body<9, true>(a, blockIdx.x, blockIdx.y, blockIdx.z);
}
#endif
template <int NT, bool FAST>
__global__ __launch_bounds__(512, 2) void conv_wgrad_kernel(WgradArgs a) {
    if constexpr (NT == 9 && FAST) conv_wgrad9_body(a, blockIdx.x, blockIdx.y, blockIdx.z);
    else conv_wgrad_body<NT, FAST>(a, blockIdx.x, blockIdx.y, blockIdx.z);
}

// Several weight gradients in ONE launch.  Every workgroup costs one (taps x 64 x 64) fp32 slab of reduction traffic
// (written here, read by the reduction), so a launch per layer with one workgroup per CU moves 256 slabs = 38 MB per
// layer whatever the layer's size; with all weight gradients of a backward pass in one launch the chip's 256 workgroups
// are dealt out over the problems in proportion to their work and the pass moves 38 MB in total.  Problems are sorted by
// work per workgroup (largest first), so the dispatcher back-fills CUs with the small ones.
constexpr int WGRAD_MULTI_PROBLEMS = 24;
struct WgradMultiLaunch {
    WgradArgs p[WGRAD_MULTI_PROBLEMS];
    int wg_begin[WGRAD_MULTI_PROBLEMS + 1];
    int n;
};
template <int NT, bool FAST>
__global__ __launch_bounds__(512, 2) void conv_wgrad_multi_kernel(WgradMultiLaunch m) {
    int i = 0;
    while (i + 1 < m.n && (int)blockIdx.x >= m.wg_begin[i + 1]) ++i;          // block-uniform scan
    const WgradArgs& a = m.p[i];
    int local = (int)blockIdx.x - m.wg_begin[i];
    // XCD-aware order WITHIN a problem: hardware deals consecutive blockIdx round-robin over the 8 XCDs (each with its own
    // L2), so the (co, ci) tiles of one patch range -- which read the same x slices (per ci tile) and dy slices (per co
    // tile) -- would sit on different XCDs and each fetch its own copy.  When the problem's workgroup range is aligned
    // to 8 (the planner rounds to that), XCD k takes the k-th contiguous eighth of the problem's (split, tile) list: whole
    // tile grids run on one XCD at the same time and share their slices through its L2, and every XCD still gets an
    // eighth of every problem (a chip-wide contiguous renumbering was 1.5x SLOWER: problems differ in patches per workgroup).
    {
        const int n_i = m.wg_begin[i + 1] - m.wg_begin[i];
        if (((m.wg_begin[i] | n_i) & 7) == 0) local = (local & 7) * (n_i >> 3) + (local >> 3);
    }
    const int tiles_ci = a.Cin >> 6, tiles = tiles_ci * (a.Cout >> 6);
    const int split = local / tiles, tile = local - split * tiles;
    if constexpr (NT == 9 && FAST) conv_wgrad9_body(a, split, tile % tiles_ci, tile / tiles_ci);
    else conv_wgrad_body<NT, FAST>(a, split, tile % tiles_ci, tile / tiles_ci);
}

// Slab reduction + layout change in one launch.  1x1 weights: a block owns 32 float4 of the packed (co, ci) tile and its
// 256 threads are 32 quads x 8 slab groups.  3x3 weights: a block owns (one co, 32 ci, all nine taps) = 72 float4 of
// the packed (tap, co, ci) tile as 72 quads x 3 slab groups, so that after the sum the 288 values are one contiguous run
// of the master layout dw[co][ci][tap] -- they turn through LDS and leave as 72 float4 (the tap-strided scalar
// read-add-write this replaces touched every 128-byte line of dw from nine different blocks).  Every thread sums its
// group's slabs with four independent loads in flight; the owner writes scale * sum (plain stores or read-add-write: dw
// is the optimizer's accumulating gradient buffer).
__host__ __device__ __forceinline__ long wgrad_reduce_blocks(int taps, int cout, int cin) {
    return taps == 9 ? (long)cout * (cin >> 5) : ((long)cout * cin / 4 + 31) / 32;
}
__device__ __forceinline__ f32x4 wgrad_reduce_slabs(const float* __restrict__ slabs, long total, long e4, int s_begin, int s_end) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    int s2 = s_begin;
    for (; s2 + 4 <= s_end; s2 += 4) {
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(slabs + (long)(s2 + 0) * total + e4);
        const f32x4 v1 = *reinterpret_cast<const f32x4*>(slabs + (long)(s2 + 1) * total + e4);
        const f32x4 v2 = *reinterpret_cast<const f32x4*>(slabs + (long)(s2 + 2) * total + e4);
        const f32x4 v3 = *reinterpret_cast<const f32x4*>(slabs + (long)(s2 + 3) * total + e4);
        acc += (v0 + v1) + (v2 + v3);
    }
    for (; s2 < s_end; ++s2) acc += *reinterpret_cast<const f32x4*>(slabs + (long)s2 * total + e4);
    return acc;
}
__device__ __forceinline__ void wgrad_reduce_block(const float* __restrict__ slabs, int nslab, float* __restrict__ dw,
                                                   int taps, int cout, int cin, float scale, int accumulate, long block) {
    __shared__ f32x4 part[8][32];                                  // 1x1: [group][quad]; 3x3: [2][72] partials, then [288] floats
    const long total = (long)taps * cout * cin;
    if (taps == 9) {
        const int chunks = cin >> 5;
        const int co = (int)(block / chunks), ci0 = (int)(block - (long)co * chunks) << 5;
        const int quad = threadIdx.x % 72, grp = threadIdx.x / 72;  // threads 216..255 idle
        const int tap = quad >> 3, c4 = (quad & 7) << 2;
        f32x4* part9 = &part[0][0];
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        if (grp < 3) {
            const int per_group = (nslab + 2) / 3;
            const int s_begin = grp * per_group;
            acc = wgrad_reduce_slabs(slabs, total, ((long)tap * cout + co) * cin + ci0 + c4, s_begin, min(nslab, s_begin + per_group));
            if (grp > 0) part9[(grp - 1) * 72 + quad] = acc;
        }
        __syncthreads();
        if (grp == 0) acc += part9[quad] + part9[72 + quad];
        __syncthreads();
        float* turn = reinterpret_cast<float*>(part9);             // [32 ci][9 taps]
        if (grp == 0) {
#pragma unroll
            for (int k = 0; k < 4; ++k) turn[(c4 + k) * 9 + tap] = acc[k] * scale;
        }
        __syncthreads();
        if (grp == 0) {
            f32x4 v = part9[quad];
            f32x4* o = reinterpret_cast<f32x4*>(dw + ((long)co * cin + ci0) * 9) + quad;
            if (accumulate) v += *o;
            *o = v;
        }
        return;
    }
    const int quad = threadIdx.x & 31, grp = threadIdx.x >> 5;
    const long e4 = (block * 32 + quad) * 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (e4 < total) {
        const int per_group = (nslab + 7) >> 3;
        const int s_begin = grp * per_group;
        acc = wgrad_reduce_slabs(slabs, total, e4, s_begin, min(nslab, s_begin + per_group));
    }
    part[grp][quad] = acc;
    __syncthreads();
    if (grp == 0 && e4 < total) {
#pragma unroll
        for (int g = 1; g < 8; ++g) acc += part[g][quad];
        // taps == 1: packed (co, ci) is the master layout
        f32x4* o = reinterpret_cast<f32x4*>(dw + e4);
        f32x4 v = acc * scale;
        if (accumulate) v += *o;
        *o = v;
    }
}

__global__ __launch_bounds__(256) void wgrad_reduce_finish_kernel(const float* __restrict__ slabs, int nslab,
                                                                  float* __restrict__ dw, int taps, int cout, int cin,
                                                                  float scale, int accumulate) {
    wgrad_reduce_block(slabs, nslab, dw, taps, cout, cin, scale, accumulate, blockIdx.x);
}

// The same reduction for up to WGRAD_MULTI_MAX weight gradients in ONE launch (descriptors by value in the kernel
// arguments): a backward pass whose weight gradients were deferred (functional.deferred_wgrads) pays one reduction launch
// instead of one per layer.
constexpr int WGRAD_MULTI_MAX = 32;
struct WgradMultiArgs {
    rgbd_wgrad_reduce_desc d[WGRAD_MULTI_MAX];
    int block_begin[WGRAD_MULTI_MAX + 1];
    int n;
};
__global__ __launch_bounds__(256) void wgrad_reduce_multi_kernel(WgradMultiArgs m) {
    int i = 0;
    while (i + 1 < m.n && (int)blockIdx.x >= m.block_begin[i + 1]) ++i;     // block-uniform scan, <= 32 entries
    const rgbd_wgrad_reduce_desc& d = m.d[i];
    wgrad_reduce_block(d.workspace, d.nsplit, d.dw, d.taps, d.cout, d.cin, d.scale, d.accumulate,
                       (long)blockIdx.x - m.block_begin[i]);
}

// compute units of the CURRENT device (a process may drive several): looked up once per device, immutable afterwards
// `budget` > 0: the caller's compute-unit budget for THIS launch (the `cus` argument of the conv entry points, the `cus`
// field of rgbd_conv3x3_desc): min(budget, the device's).  The library keeps no budget of its own.
int device_cus(bool for_wgrad = false, int budget = 0) {
#ifdef RGBD_DEBUG_BUILD
    // experiment hook (debug library only): persistent grids sized for FEWER compute units than the chip has
    static const int forced = getenv("RGBD_DEBUG_CUS") ? atoi(getenv("RGBD_DEBUG_CUS")) : 0;
    static const int forced_w = getenv("RGBD_DEBUG_CUS_WGRAD") ? atoi(getenv("RGBD_DEBUG_CUS_WGRAD")) : 0;
    if (for_wgrad && forced_w > 0) return forced_w;
    if (forced > 0) return forced;
#endif
    static int cus[64] = {0};
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64) {
        if (cus[dev] == 0) {
            hipDeviceProp_t prop;
            cus[dev] = hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        }
        n = cus[dev];
    }
    return budget > 0 && budget < n ? budget : n;
}
bool reserve_lds(const void* fn, int bytes) { return rgbd_reserve_lds(fn, bytes); }      // (common.h: once per kernel AND device)
#ifdef RGBD_DEBUG_BUILD
bool g_force_gather = false;   // test hook: route every shape through the generic gather kernel
int g_conv_variant = 0;        // test / tuning hook: 1 = the register-staged conv3x3_patch_kernel instead of the ping-pong one
#else                          // the shipped library has neither switch nor the kernels they select: no process-wide state
constexpr bool g_force_gather = false;
constexpr int g_conv_variant = 0;
#endif

int ilog2(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return l;
}

}  // namespace

#ifdef RGBD_DEBUG_BUILD
namespace { unsigned* g_dw_census = nullptr; }
// census of the dual-workgroup kernel's last launch: 8 words per workgroup (HW_ID, XCC_ID, start and end of s_memrealtime,
// slices, block index); on = 1 allocates the buffer and switches the census on, host_out != null copies n workgroups out
extern "C" int rgbd_debug_dw_census(int on, unsigned* host_out, int n) {
    if (on && !g_dw_census) RGBD_REQUIRE(hipMalloc(&g_dw_census, 8 * 4096 * sizeof(unsigned)) == hipSuccess, "census: hipMalloc");
    if (!on && g_dw_census) { (void)hipFree(g_dw_census); g_dw_census = nullptr; }
    if (host_out && g_dw_census)
        RGBD_REQUIRE(hipMemcpy(host_out, g_dw_census, (size_t)8 * n * sizeof(unsigned), hipMemcpyDeviceToHost) == hipSuccess, "census: copy");
    return 0;
}
extern "C" int rgbd_debug_force_gather_kernel(int on) {
    g_force_gather = on != 0;
    return 0;
}
extern "C" int rgbd_debug_conv_variant(int v) {
    g_conv_variant = v;
    return 0;
}
#endif
namespace {
thread_local const char* g_last_conv_kernel = "";     // per calling thread: profiling label of its last conv launch
}
extern "C" const char* rgbd_last_conv_kernel(void) { return g_last_conv_kernel; }
namespace {
// Split-K plan of the gather kernel: 1 (no split) when the unsplit launch already fills the chip.
struct FpropPlan {
    bool patch;      // 3x3 pad-1 halo-patch kernel
    int ksplit;
    bool small;      // conv3x3_small_kernel: 4x4 / 8x8 images, one workgroup per (128 pixels, 64 co, 64 ci), ksplit = Cin / 64
};
FpropPlan plan_fprop(int B, int Hout, int Wout, int Cin, int Cout, int KH, int KW, int pad, bool allow_small = false) {
    FpropPlan p;
    p.small = false;
    const long M = (long)B * Hout * Wout;
    if (allow_small && KH == 3 && KW == 3 && pad == 1 && Hout == Wout && (Hout == 4 || Hout == 8) && M % 128 == 0 &&
        Cin <= 512 && !g_force_gather && g_conv_variant != 4) {
        p.patch = false;
        p.small = true;
        p.ksplit = Cin / 64;       // (1 for Cin = 64: the finish kernel still applies the epilogue)
        return p;
    }
    const int bn = Cout % 128 == 0 ? 128 : 64;
    const long tiles = ((M + 127) / 128) * (Cout / bn);
    const int nk = KH * KW * (Cin / 64);
    const bool eligible = KH == 3 && KW == 3 && pad == 1 && Hout % 16 == 0 && Wout % 16 == 0 && !g_force_gather;
    const long patch_wgs = (long)B * (Hout / 16) * (Wout / 16) * (Cout / bn);
    // measured (scripts/time_small_conv.py, B=32, 256 channels): a workgroup's K loop is issue/latency bound at
    // ~0.9 us per step whatever the tile count, so split until ~128 workgroups exist, at most 6 ways (beyond that
    // the fp32 partial traffic costs more than the shorter loops save); 16x16 images stay on the halo-patch kernel
    p.patch = eligible && patch_wgs >= 64;
    p.ksplit = 1;
    if (!p.patch && tiles < 128) {
        long s = (128 + tiles - 1) / tiles;
        if (s > 6) s = 6;
        if (s > nk / 2) s = nk / 2;
        if (s > 1) p.ksplit = (int)s;
    }
#ifdef RGBD_DEBUG_BUILD
    static const int ksplit_env = [] {                          // tuning aid, read ONCE: 0 = patch kernel where eligible, n = force n
        const char* e = getenv("RGBD_DEBUG_KSPLIT");
        return e ? atoi(e) : -1;
    }();
    if (ksplit_env >= 0) {
        const int v = ksplit_env;
        if (v == 0) { p.patch = eligible; p.ksplit = 1; }
        else if (v > 0) { p.patch = false; p.ksplit = v > nk / 2 ? (nk / 2 > 0 ? nk / 2 : 1) : v; }
    }
#endif
    return p;
}
}  // namespace

extern "C" int64_t rgbd_conv2d_fprop_workspace(int B, int Hin, int Win, int Cin, int Cout, int KH, int KW, int pad,
                                               int upsample) {
    if (B <= 0 || Hin <= 0 || Win <= 0 || Cin % 64 || Cout % 64 || KH <= 0 || KW <= 0 || pad < 0) return -1;
    const int Hout = (upsample ? 2 * Hin : Hin) + 2 * pad - KH + 1, Wout = (upsample ? 2 * Win : Win) + 2 * pad - KW + 1;
    if (Hout <= 0 || Wout <= 0) return -1;
    const FpropPlan p = plan_fprop(B, Hout, Wout, Cin, Cout, KH, KW, pad, !upsample);
    return p.ksplit > 1 || p.small ? (int64_t)p.ksplit * B * Hout * Wout * Cout * (int64_t)sizeof(float) : 0;
}

namespace {
// One launch of the pipelined 3x3 kernel's MXFP8 form; the LDS reservation is made once per instantiation.
template <int BN, bool UPS, int EPI, bool MX = true, bool EMIT = false>
int launch_sp_mx(const ConvArgs& a, unsigned grid, hipStream_t st) {
    constexpr int pieces = UPS ? 13 : 41, spieces = UPS ? 2 : 6;
    constexpr int lds = 2 * pieces * 1024 + 3 * BN * 128 + BN * 4 + (MX ? 2 * spieces * 256 + 3 * BN * 4 : 0);
    RGBD_REQUIRE(reserve_lds((const void*)&conv3x3_sp_kernel<BN, UPS, 0, EPI, MX, EMIT>, lds),
                 "rgbd_conv3x3: cannot reserve %d B of LDS", lds);
    conv3x3_sp_kernel<BN, UPS, 0, EPI, MX, EMIT><<<grid, 512, lds, st>>>(a);
    RGBD_CHECK_LAUNCH("conv3x3_sp_kernel");
    return 0;
}
template <int EPI>
int launch_sp_mx_any(const ConvArgs& a, bool wide, unsigned grid, hipStream_t st) {
    if (EPI == 1) return wide ? launch_sp_mx<128, false, 1>(a, grid, st) : launch_sp_mx<64, false, 1>(a, grid, st);
    if (a.ups) return wide ? launch_sp_mx<128, true, EPI == 1 ? 0 : EPI>(a, grid, st)
                           : launch_sp_mx<64, true, EPI == 1 ? 0 : EPI>(a, grid, st);
    return wide ? launch_sp_mx<128, false, EPI>(a, grid, st) : launch_sp_mx<64, false, EPI>(a, grid, st);
}
// the EMITting instantiations: plain and masked epilogues, no folded upsample, either operand type
template <bool MX>
int launch_sp_emit(const ConvArgs& a, bool wide, bool masked, unsigned grid, hipStream_t st) {
    if (masked) return wide ? launch_sp_mx<128, false, 1, MX, true>(a, grid, st) : launch_sp_mx<64, false, 1, MX, true>(a, grid, st);
    return wide ? launch_sp_mx<128, false, 0, MX, true>(a, grid, st) : launch_sp_mx<64, false, 0, MX, true>(a, grid, st);
}
}  // namespace

static int conv_fprop_impl(const void* x, const void* wp, const float* bias, const void* residual,
                           void* y, int B, int Hin, int Win, int Cin, int Cout, int KH, int KW, int pad,
                           int upsample, int lrelu_channels, float slope, void* workspace, void* stream, int cus,
                           int pool_sum, void* y_pooled = nullptr, const void* mask_y = nullptr, float* colsum = nullptr,
                           const float* row_scale = nullptr, long long* stats = nullptr, void* y2 = nullptr,
                           const float* row_scale2 = nullptr, const void* x_scales = nullptr,
                           const void* w_scales = nullptr, void* y_q = nullptr, void* y_s = nullptr, void* yp_q = nullptr,
                           void* yp_s = nullptr) {
    RGBD_REQUIRE(x && wp && y, "rgbd_conv2d_fprop_bf16: null pointer");
    RGBD_REQUIRE(B > 0 && Hin > 0 && Win > 0 && KH > 0 && KW > 0 && pad >= 0, "rgbd_conv2d_fprop_bf16: bad shape");
    RGBD_REQUIRE(Cin % 64 == 0 && Cout % 64 == 0,
                 "rgbd_conv2d_fprop_bf16: Cin and Cout must be multiples of 64 (Cin=%d Cout=%d)", Cin, Cout);
    RGBD_REQUIRE(lrelu_channels % 16 == 0 && lrelu_channels >= 0 && lrelu_channels <= Cout,
                 "rgbd_conv2d_fprop_bf16: lrelu_channels must be a multiple of 16 in [0, Cout]");
    RGBD_REQUIRE(lrelu_channels == 0 || (slope >= 0.f && slope <= 1.f),
                 "rgbd_conv2d_fprop_bf16: the leaky-ReLU slope must lie in [0, 1] (the epilogue computes max(v, slope v))");
    ConvArgs a;
    a.x = (const unsigned short*)x; a.wp = (const unsigned short*)wp; a.bias = bias;
    a.resid = (const unsigned short*)residual; a.y = (unsigned short*)y;
    a.B = B; a.Hin = Hin; a.Win = Win; a.Cin = Cin; a.Cout = Cout; a.KH = KH; a.KW = KW; a.pad = pad;
    a.ups = upsample ? 1 : 0;
    const int Hup = upsample ? 2 * Hin : Hin, Wup = upsample ? 2 * Win : Win;
    a.Hout = Hup + 2 * pad - KH + 1;
    a.Wout = Wup + 2 * pad - KW + 1;
    RGBD_REQUIRE(a.Hout > 0 && a.Wout > 0, "rgbd_conv2d_fprop_bf16: empty output");
    RGBD_REQUIRE((long)B * Hin * Win * Cin < 0x3fffffffL && (long)KH * KW * Cout * Cin < 0x3fffffffL,
                 "rgbd_conv2d_fprop_bf16: tensors must be smaller than 2 GiB (32-bit buffer offsets)");
    const bool mx = x_scales != nullptr;
    a.x_bytes = (int)((long)B * Hin * Win * Cin * (mx ? 1 : 2));
    a.w_bytes = (int)((long)KH * KW * Cout * Cin * (mx ? 1 : 2));
    a.xs = (const unsigned char*)x_scales; a.ws = (const unsigned char*)w_scales;
    a.xs_bytes = (int)((long)B * Hin * Win * (Cin / 32));
    a.ws_bytes = (int)((long)KH * KW * Cout * (Cin / 32));
    a.lrelu_ch = lrelu_channels; a.slope = slope;
    a.M = (long)B * a.Hout * a.Wout;
    const long mtiles = (a.M + 127) / 128;
    hipStream_t st = (hipStream_t)stream;
    FpropPlan plan = plan_fprop(B, a.Hout, a.Wout, Cin, Cout, KH, KW, pad,
                                !upsample && workspace && !pool_sum && !y_pooled);
    if (!workspace && plan.ksplit > 1) {       // no scratch from the caller: unsplit (the halo-patch kernel if it applies)
        plan.ksplit = 1;
        plan.patch = KH == 3 && KW == 3 && pad == 1 && a.Hout % 16 == 0 && a.Wout % 16 == 0 && !g_force_gather;
    }
    a.pool_sum = pool_sum ? 1 : 0;
    a.ypool = (unsigned short*)y_pooled;
    a.mask_y = (const unsigned short*)mask_y; a.colsum = colsum; a.row_scale = row_scale;
    a.stats = stats;
    a.y2 = (unsigned short*)y2; a.row_scale2 = row_scale2;
    a.yq = (unsigned char*)y_q; a.ys = (unsigned char*)y_s; a.ypq = (unsigned char*)yp_q; a.yps = (unsigned char*)yp_s;
    if (y_q || yp_q) {
        RGBD_REQUIRE(KH == 3 && KW == 3 && pad == 1 && a.Hout % 16 == 0 && a.Wout % 16 == 0 && !pool_sum && !g_force_gather &&
                     g_conv_variant == 0 && !upsample && !stats && Cout % 32 == 0 && (!y_q || y_s) && (!yp_q || (yp_s && y_pooled)),
                     "rgbd_conv3x3_ex: MXFP8 copies of the outputs need a 3x3 pad-1 conv on output images that are multiples of "
                     "16x16 (the pipelined kernel's epilogue), scale buffers, and y_pooled for the pooled copy");
        plan.patch = true;
        plan.ksplit = 1;
        plan.small = false;
    }
    if (stats) {
        RGBD_REQUIRE(KH == 3 && KW == 3 && pad == 1 && a.Hout % 16 == 0 && a.Wout % 16 == 0 && !pool_sum && !y_pooled &&
                     !mask_y && !g_force_gather && g_conv_variant != 1,
                     "rgbd_conv2d_fprop_stats_bf16: needs a 3x3 pad-1 conv on output images that are multiples of 16x16");
        plan.patch = true;
        plan.ksplit = 1;
        plan.small = false;
    }
    if (mask_y) {
        RGBD_REQUIRE(KH == 3 && KW == 3 && pad == 1 && a.Hout % 16 == 0 && a.Wout % 16 == 0 && !upsample && !pool_sum &&
                     !y_pooled && !bias && lrelu_channels == 0 && !g_force_gather && g_conv_variant != 1,
                     "rgbd_conv3x3_actgrad_bf16: needs a plain 3x3 pad-1 conv on images that are multiples of 16x16");
        plan.patch = true;
        plan.ksplit = 1;
        plan.small = false;
    }
    if (y_pooled) {
        RGBD_REQUIRE(KH == 3 && KW == 3 && pad == 1 && a.Hout % 16 == 0 && a.Wout % 16 == 0 && !pool_sum && !g_force_gather,
                     "rgbd_conv2d_fprop_bf16: y_pooled needs a 3x3 pad-1 conv on images that are multiples of 16x16");
        plan.patch = true;
        plan.ksplit = 1;
    }
    if (pool_sum) {
        RGBD_REQUIRE(KH == 3 && KW == 3 && pad == 1 && a.Hout % 16 == 0 && a.Wout % 16 == 0 && !bias && !residual &&
                     lrelu_channels == 0 && !g_force_gather,
                     "rgbd_conv2d_dgrad_bf16: sum_pool2 needs a 3x3 pad-1 conv on images that are multiples of 16x16");
        plan.patch = true;
        plan.ksplit = 1;
    }
    if (mx) {
        RGBD_REQUIRE(w_scales, "rgbd_conv3x3_mxfp8: null pointer");
        RGBD_REQUIRE(KH == 3 && KW == 3 && pad == 1 && a.Hout % 16 == 0 && a.Wout % 16 == 0 && Cin % 128 == 0,
                     "rgbd_conv3x3_mxfp8: needs a 3x3 pad-1 conv, output images that are multiples of 16x16 and Cin a "
                     "multiple of 128 (Cin=%d, output %dx%d)", Cin, a.Hout, a.Wout);
        plan.patch = true;
        plan.ksplit = 1;
        plan.small = false;
    }
    a.ksplit = plan.ksplit;
    a.partial = plan.ksplit > 1 || plan.small ? (float*)workspace : nullptr;
    if (plan.small) {
        const int S = a.Hout;
        const int lds = (128 / (S * S)) * (S + 2) * (S + 2) * 128 + 9 * 64 * 128;
        RGBD_REQUIRE(reserve_lds(S == 8 ? (const void*)&conv3x3_small_kernel<8> : (const void*)&conv3x3_small_kernel<4>, lds),
                     "rgbd_conv2d_fprop_bf16: cannot reserve %d B of LDS", lds);
        const unsigned grid = (unsigned)((a.M / 128) * (Cout / 64) * (Cin / 64));
        if (S == 8) conv3x3_small_kernel<8><<<grid, 256, lds, st>>>(a);
        else        conv3x3_small_kernel<4><<<grid, 256, lds, st>>>(a);
        RGBD_CHECK_LAUNCH("conv3x3_small_kernel");
        g_last_conv_kernel = S == 8 ? "conv3x3_small_kernel<8>" : "conv3x3_small_kernel<4>";
        const long quads = a.M * Cout / 4;
        conv_splitk_finish_kernel<<<(unsigned)((quads + 255) / 256 < 2048 ? (quads + 255) / 256 : 2048), 256, 0, st>>>(
            a.partial, a.ksplit, a.M, Cout, bias, (const unsigned short*)residual, lrelu_channels, slope,
            (unsigned short*)y);
        RGBD_CHECK_LAUNCH("conv_splitk_finish_kernel");
        return 0;
    }
    if (plan.patch) {
        const long ptiles = (long)B * (a.Hout / 16) * (a.Wout / 16);
        RGBD_REQUIRE(ptiles < 0x7fffffffL, "rgbd_conv2d_fprop_bf16: too many tiles");
        // 128 output channels per workgroup unless that leaves more than half of the chip without one (16x16 images)
        const bool wide = Cout % 128 == 0 && (g_conv_variant == 1 || ptiles * (Cout / 128) >= 128) && g_conv_variant != 2;
        const int n_tiles = wide ? Cout / 128 : Cout / 64;
        // persistent workgroups: one per CU (the kernel's LDS footprint allows exactly one), split evenly over the
        // output-channel tiles; each walks a contiguous range of pixel tiles
        const int num_cus = device_cus(false, cus);
        int per_nt = num_cus / n_tiles;
        if (per_nt < 1) per_nt = 1;
        if (per_nt > ptiles) per_nt = (int)ptiles;
        // ... and no more of them than the number of ROUNDS needs: the launch lasts as long as its busiest workgroup, so
        // 320 tiles are two rounds on 256 workgroups and on 160, and the 96 compute units left over serve the other stream
        // (profiles/r05/cu_budget_sweep.txt)
        per_nt = (int)((ptiles + (ptiles + per_nt - 1) / per_nt - 1) / ((ptiles + per_nt - 1) / per_nt));
        a.ptiles = (int)ptiles;
        a.wgs_per_ntile = per_nt;
        const long grid = (long)per_nt * n_tiles;
        if (a.yq || a.ypq) {
            const int rc = mx ? launch_sp_emit<true>(a, wide, mask_y != nullptr, (unsigned)grid, st)
                              : launch_sp_emit<false>(a, wide, mask_y != nullptr, (unsigned)grid, st);
            if (rc) return rc;
            g_last_conv_kernel = mx ? (mask_y ? (wide ? "conv3x3_sp_kernel<128,actgrad,mxfp8>" : "conv3x3_sp_kernel<64,actgrad,mxfp8>")
                                              : (wide ? "conv3x3_sp_kernel<128,mxfp8>" : "conv3x3_sp_kernel<64,mxfp8>"))
                                    : (mask_y ? (wide ? "conv3x3_sp_kernel<128,actgrad>" : "conv3x3_sp_kernel<64,actgrad>")
                                              : (wide ? "conv3x3_sp_kernel<128>" : "conv3x3_sp_kernel<64>"));
            return 0;
        }
        if (mx) {
            const int rc = stats ? launch_sp_mx_any<2>(a, wide, (unsigned)grid, st)
                         : mask_y ? launch_sp_mx_any<1>(a, wide, (unsigned)grid, st)
                                  : launch_sp_mx_any<0>(a, wide, (unsigned)grid, st);
            if (rc) return rc;
            g_last_conv_kernel = stats ? (wide ? "conv3x3_sp_kernel<128,stats,mxfp8>" : "conv3x3_sp_kernel<64,stats,mxfp8>")
                               : mask_y ? (wide ? "conv3x3_sp_kernel<128,actgrad,mxfp8>" : "conv3x3_sp_kernel<64,actgrad,mxfp8>")
                                        : (wide ? "conv3x3_sp_kernel<128,mxfp8>" : "conv3x3_sp_kernel<64,mxfp8>");
            return 0;
        }
        // ---- the dual-workgroup kernel (two 4-wave workgroups per CU): round 5's A/B against the 8-wave kernel, debug library
        //      only (variants 8 / 7: 64- / 128-channel tiles, 6 and 21-30: its knock-outs and stamps; scripts/ab_conv_dw.py)
#ifdef RGBD_DEBUG_BUILD
        const bool dw_debug = g_conv_variant == 6 || g_conv_variant == 7 || g_conv_variant == 8 || (g_conv_variant >= 21 && g_conv_variant <= 30);
        if (dw_debug) {
            bool wide2 = false;
            if (g_conv_variant == 7 || g_conv_variant == 6 || g_conv_variant >= 21) wide2 = Cout % 128 == 0;
            const int n_tiles2 = wide2 ? Cout / 128 : Cout / 64;
            int per_nt2 = 2 * num_cus / n_tiles2;
            if (per_nt2 < 1) per_nt2 = 1;
            if (per_nt2 > ptiles) per_nt2 = (int)ptiles;
            a.wgs_per_ntile = per_nt2;
            a.partial = nullptr;
#ifdef RGBD_DEBUG_BUILD
            a.partial = (float*)g_dw_census;
#endif
            const unsigned grid2 = (unsigned)(per_nt2 * n_tiles2);
            const int bn2 = wide2 ? 128 : 64;
            const int lds_dw = 2 * (a.ups ? 7 : 21) * 1024 + 3 * bn2 * 64 + bn2 * 4;
            const int epi = stats ? 2 : mask_y ? 1 : 0;
#define RGBD_DW_CASE(BNv, UPSv, EPIv)                                                                                      \
    do {                                                                                                                   \
        RGBD_REQUIRE(reserve_lds((const void*)&conv3x3_dw_kernel<BNv, UPSv, EPIv>, lds_dw),                                \
                     "rgbd_conv3x3: cannot reserve %d B of LDS", lds_dw);                                                  \
        conv3x3_dw_kernel<BNv, UPSv, EPIv><<<grid2, 256, lds_dw, st>>>(a);                                                 \
    } while (0)
#ifdef RGBD_DEBUG_BUILD
            if ((g_conv_variant == 6 || (g_conv_variant >= 21 && g_conv_variant <= 30)) && wide2 && !a.ups && epi == 0) {   // timing knock-outs, stamps
                const int ko = g_conv_variant == 6 ? 9 : g_conv_variant - 20;
                const void* fk = ko == 9 ? (const void*)&conv3x3_dw_kernel<128, false, 0, 9>
                               : ko == 1 ? (const void*)&conv3x3_dw_kernel<128, false, 0, 1>
                               : ko == 2 ? (const void*)&conv3x3_dw_kernel<128, false, 0, 2>
                               : ko == 3 ? (const void*)&conv3x3_dw_kernel<128, false, 0, 3>
                               : ko == 10 ? (const void*)&conv3x3_dw_kernel<128, false, 0, 10>
                                          : (const void*)&conv3x3_dw_kernel<128, false, 0, 6>;
                RGBD_REQUIRE(hipFuncSetAttribute(fk, hipFuncAttributeMaxDynamicSharedMemorySize, lds_dw) == hipSuccess, "lds");
                if (ko == 9) conv3x3_dw_kernel<128, false, 0, 9><<<grid2, 256, lds_dw, st>>>(a);
                else if (ko == 1) conv3x3_dw_kernel<128, false, 0, 1><<<grid2, 256, lds_dw, st>>>(a);
                else if (ko == 2) conv3x3_dw_kernel<128, false, 0, 2><<<grid2, 256, lds_dw, st>>>(a);
                else if (ko == 3) conv3x3_dw_kernel<128, false, 0, 3><<<grid2, 256, lds_dw, st>>>(a);
                else if (ko == 10) conv3x3_dw_kernel<128, false, 0, 10><<<grid2, 256, lds_dw, st>>>(a);
                else conv3x3_dw_kernel<128, false, 0, 6><<<grid2, 256, lds_dw, st>>>(a);
                RGBD_CHECK_LAUNCH("conv3x3_dw_kernel<KO>");
                return 0;
            }
            if (wide2) {           // 128-channel tiles: the A/B of scripts/ab_conv_dw.py only (plain epilogue)
                RGBD_REQUIRE(epi == 0, "rgbd_debug_conv_variant(7): the 128-channel dual-workgroup form exists with the plain epilogue only");
                if (a.ups) RGBD_DW_CASE(128, true, 0); else RGBD_DW_CASE(128, false, 0);
                RGBD_CHECK_LAUNCH("conv3x3_dw_kernel");
                g_last_conv_kernel = "conv3x3_dw_kernel<128>";
                return 0;
            }
#endif
            if (a.ups) { if (epi == 2) RGBD_DW_CASE(64, true, 2); else RGBD_DW_CASE(64, true, 0); }
            else if (epi == 2) RGBD_DW_CASE(64, false, 2);
            else if (epi == 1) RGBD_DW_CASE(64, false, 1);
            else RGBD_DW_CASE(64, false, 0);
#undef RGBD_DW_CASE
            RGBD_CHECK_LAUNCH("conv3x3_dw_kernel");
            g_last_conv_kernel = stats ? "conv3x3_dw_kernel<64,stats>" : mask_y ? "conv3x3_dw_kernel<64,actgrad>" : "conv3x3_dw_kernel<64>";
            return 0;
        }
#endif
        if (g_conv_variant != 1) {
            const int pieces = a.ups ? 13 : 41, bn = wide ? 128 : 64;
            const int lds_sp = 2 * pieces * 1024 + 3 * bn * 128 + bn * 4;
            const int vi = (wide ? 2 : 0) + a.ups;
            const void* fsp = vi == 3 ? (const void*)&conv3x3_sp_kernel<128, true> : vi == 2 ? (const void*)&conv3x3_sp_kernel<128, false>
                            : vi == 1 ? (const void*)&conv3x3_sp_kernel<64, true> : (const void*)&conv3x3_sp_kernel<64, false>;
            RGBD_REQUIRE(reserve_lds(fsp, lds_sp),
                             "rgbd_conv2d_fprop_bf16: cannot reserve %d B of LDS", lds_sp);
#ifdef RGBD_DEBUG_BUILD
            if ((g_conv_variant == 41 || g_conv_variant == 42 || g_conv_variant == 32) && vi == 2 && !stats && !mask_y) {   // DMA roles by wave age
                a.partial = (float*)g_dw_census;
                if (g_conv_variant == 41) {
                    RGBD_REQUIRE(reserve_lds((const void*)&conv3x3_sp_kernel<128, false, 41>, lds_sp), "lds");
                    conv3x3_sp_kernel<128, false, 41><<<(unsigned)grid, 512, lds_sp, st>>>(a);
                } else if (g_conv_variant == 42) {
                    RGBD_REQUIRE(reserve_lds((const void*)&conv3x3_sp_kernel<128, false, 42>, lds_sp), "lds");
                    conv3x3_sp_kernel<128, false, 42><<<(unsigned)grid, 512, lds_sp, st>>>(a);
                } else {
                    RGBD_REQUIRE(reserve_lds((const void*)&conv3x3_sp_kernel<128, false, 31>, lds_sp), "lds");
                    conv3x3_sp_kernel<128, false, 31><<<(unsigned)grid, 512, lds_sp, st>>>(a);
                }
                RGBD_CHECK_LAUNCH("conv3x3_sp_kernel<roles>");
                return 0;
            }
            if (g_conv_variant == 31 && vi == 2 && !stats && !mask_y) {       // in-kernel stamps of the 8-wave kernel (scripts/dw_census.py)
                a.partial = (float*)g_dw_census;
                RGBD_REQUIRE(reserve_lds((const void*)&conv3x3_sp_kernel<128, false, 30>, lds_sp), "lds");
                conv3x3_sp_kernel<128, false, 30><<<(unsigned)grid, 512, lds_sp, st>>>(a);
                RGBD_CHECK_LAUNCH("conv3x3_sp_kernel<stamps>");
                return 0;
            }
            if (g_conv_variant >= 17 && g_conv_variant <= 19 && vi == 2) {     // epilogue / start-skew experiments
                const int ko = g_conv_variant - 10;
                const void* fk = ko == 7 ? (const void*)&conv3x3_sp_kernel<128, false, 7>
                               : ko == 8 ? (const void*)&conv3x3_sp_kernel<128, false, 8>
                                         : (const void*)&conv3x3_sp_kernel<128, false, 9>;
                RGBD_REQUIRE(hipFuncSetAttribute(fk, hipFuncAttributeMaxDynamicSharedMemorySize, lds_sp) == hipSuccess, "lds");
                if (ko == 7) conv3x3_sp_kernel<128, false, 7><<<(unsigned)grid, 512, lds_sp, st>>>(a);
                else if (ko == 8) conv3x3_sp_kernel<128, false, 8><<<(unsigned)grid, 512, lds_sp, st>>>(a);
                else conv3x3_sp_kernel<128, false, 9><<<(unsigned)grid, 512, lds_sp, st>>>(a);
                RGBD_CHECK_LAUNCH("conv3x3_sp_kernel<KO>");
                return 0;
            }
            if (g_conv_variant >= 11 && g_conv_variant <= 16 && vi == 2) {     // timing knock-outs (scripts/ab_conv.py)
                const int ko = g_conv_variant - 10;
                const void* fk = ko == 1 ? (const void*)&conv3x3_sp_kernel<128, false, 1>
                               : ko == 2 ? (const void*)&conv3x3_sp_kernel<128, false, 2>
                               : ko == 3 ? (const void*)&conv3x3_sp_kernel<128, false, 3>
                               : ko == 4 ? (const void*)&conv3x3_sp_kernel<128, false, 4>
                                         : (const void*)&conv3x3_sp_kernel<128, false, 6>;
                RGBD_REQUIRE(hipFuncSetAttribute(fk, hipFuncAttributeMaxDynamicSharedMemorySize, lds_sp) == hipSuccess, "lds");
                if (ko == 1) conv3x3_sp_kernel<128, false, 1><<<(unsigned)grid, 512, lds_sp, st>>>(a);
                else if (ko == 2) conv3x3_sp_kernel<128, false, 2><<<(unsigned)grid, 512, lds_sp, st>>>(a);
                else if (ko == 3) conv3x3_sp_kernel<128, false, 3><<<(unsigned)grid, 512, lds_sp, st>>>(a);
                else if (ko == 4) conv3x3_sp_kernel<128, false, 4><<<(unsigned)grid, 512, lds_sp, st>>>(a);
                else conv3x3_sp_kernel<128, false, 6><<<(unsigned)grid, 512, lds_sp, st>>>(a);
                RGBD_CHECK_LAUNCH("conv3x3_sp_kernel<KO>");
                return 0;
            }
#endif
            if (stats) {
                const void* fs = vi == 3 ? (const void*)&conv3x3_sp_kernel<128, true, 0, 2>
                               : vi == 2 ? (const void*)&conv3x3_sp_kernel<128, false, 0, 2>
                               : vi == 1 ? (const void*)&conv3x3_sp_kernel<64, true, 0, 2>
                                         : (const void*)&conv3x3_sp_kernel<64, false, 0, 2>;
                RGBD_REQUIRE(reserve_lds(fs, lds_sp),
                                 "rgbd_conv2d_fprop_stats_bf16: cannot reserve %d B of LDS", lds_sp);
                if (vi == 3)      conv3x3_sp_kernel<128, true, 0, 2><<<(unsigned)grid, 512, lds_sp, st>>>(a);
                else if (vi == 2) conv3x3_sp_kernel<128, false, 0, 2><<<(unsigned)grid, 512, lds_sp, st>>>(a);
                else if (vi == 1) conv3x3_sp_kernel<64, true, 0, 2><<<(unsigned)grid, 512, lds_sp, st>>>(a);
                else              conv3x3_sp_kernel<64, false, 0, 2><<<(unsigned)grid, 512, lds_sp, st>>>(a);
                RGBD_CHECK_LAUNCH("conv3x3_sp_kernel<stats>");
                g_last_conv_kernel = wide ? "conv3x3_sp_kernel<128,stats>" : "conv3x3_sp_kernel<64,stats>";
                return 0;
            }
            if (mask_y) {
                const void* fm = wide ? (const void*)&conv3x3_sp_kernel<128, false, 0, 1>
                                      : (const void*)&conv3x3_sp_kernel<64, false, 0, 1>;
                RGBD_REQUIRE(reserve_lds(fm, lds_sp),
                                 "rgbd_conv3x3_actgrad_bf16: cannot reserve %d B of LDS", lds_sp);
                if (wide) conv3x3_sp_kernel<128, false, 0, 1><<<(unsigned)grid, 512, lds_sp, st>>>(a);
                else      conv3x3_sp_kernel<64, false, 0, 1><<<(unsigned)grid, 512, lds_sp, st>>>(a);
                RGBD_CHECK_LAUNCH("conv3x3_sp_kernel<masked>");
                g_last_conv_kernel = wide ? "conv3x3_sp_kernel<128,actgrad>" : "conv3x3_sp_kernel<64,actgrad>";
                return 0;
            }
            if (vi == 3)      conv3x3_sp_kernel<128, true><<<(unsigned)grid, 512, lds_sp, st>>>(a);
            else if (vi == 2) conv3x3_sp_kernel<128, false><<<(unsigned)grid, 512, lds_sp, st>>>(a);
            else if (vi == 1) conv3x3_sp_kernel<64, true><<<(unsigned)grid, 512, lds_sp, st>>>(a);
            else              conv3x3_sp_kernel<64, false><<<(unsigned)grid, 512, lds_sp, st>>>(a);
            RGBD_CHECK_LAUNCH("conv3x3_sp_kernel");
            g_last_conv_kernel = wide ? "conv3x3_sp_kernel<128>" : "conv3x3_sp_kernel<64>";
            return 0;
        }
#ifdef RGBD_DEBUG_BUILD
        const int lds = 2 * 324 * 128 + 3 * (wide ? 128 : 64) * 128 + (wide ? 128 : 64) * 4;
        const void* fn = wide ? (a.ups ? (const void*)&conv3x3_patch_kernel<128, true>
                                       : (const void*)&conv3x3_patch_kernel<128, false>)
                              : (a.ups ? (const void*)&conv3x3_patch_kernel<64, true>
                                       : (const void*)&conv3x3_patch_kernel<64, false>);
        RGBD_REQUIRE(reserve_lds(fn, lds), "rgbd_conv2d_fprop_bf16: cannot reserve %d B of LDS", lds);
        if (wide) {
            if (a.ups) conv3x3_patch_kernel<128, true><<<(unsigned)grid, 512, lds, st>>>(a);
            else       conv3x3_patch_kernel<128, false><<<(unsigned)grid, 512, lds, st>>>(a);
        } else {
            if (a.ups) conv3x3_patch_kernel<64, true><<<(unsigned)grid, 512, lds, st>>>(a);
            else       conv3x3_patch_kernel<64, false><<<(unsigned)grid, 512, lds, st>>>(a);
        }
        RGBD_CHECK_LAUNCH("conv3x3_patch_kernel");
        g_last_conv_kernel = wide ? "conv3x3_patch_kernel<128>" : "conv3x3_patch_kernel<64>";
        return 0;
#else
        rgbd_set_error("rgbd_conv2d_fprop_bf16: unreachable");
        return -1;
#endif
    }
    if (Cout % 128 == 0) {
        const long grid = mtiles * (Cout / 128) * a.ksplit;
        RGBD_REQUIRE(grid < 0x7fffffffL, "rgbd_conv2d_fprop_bf16: grid too large");
        conv_fprop_kernel<128><<<(unsigned)grid, 256, 0, st>>>(a);
    } else {
        const long grid = mtiles * (Cout / 64) * a.ksplit;
        RGBD_REQUIRE(grid < 0x7fffffffL, "rgbd_conv2d_fprop_bf16: grid too large");
        conv_fprop_kernel<64><<<(unsigned)grid, 256, 0, st>>>(a);
    }
    RGBD_CHECK_LAUNCH("conv_fprop_kernel");
    g_last_conv_kernel = Cout % 128 == 0 ? "conv_fprop_kernel<128>" : "conv_fprop_kernel<64>";
    if (a.ksplit > 1) {
        const long quads = a.M * Cout / 4;
        conv_splitk_finish_kernel<<<(unsigned)((quads + 255) / 256 < 2048 ? (quads + 255) / 256 : 2048), 256, 0, st>>>(
            a.partial, a.ksplit, a.M, Cout, bias, (const unsigned short*)residual, lrelu_channels, slope,
            (unsigned short*)y);
        RGBD_CHECK_LAUNCH("conv_splitk_finish_kernel");
    }
    return 0;
}

extern "C" int rgbd_conv2d_fprop_bf16(const void* x, const void* wp, const float* bias, const void* residual,
                                      void* y, void* y_pooled, int B, int Hin, int Win, int Cin, int Cout, int KH, int KW,
                                      int pad, int upsample, int lrelu_channels, float slope, void* workspace,
                                      int cus, void* stream) {
    return conv_fprop_impl(x, wp, bias, residual, y, B, Hin, Win, Cin, Cout, KH, KW, pad, upsample, lrelu_channels, slope,
                           workspace, stream, cus, 0, y_pooled);
}

namespace {
struct WgradPlan {
    int PH, PW, lgPW, npx, npy, total_patches, patches_per_wg, nsplit;
};
WgradPlan plan_wgrad(int B, int H, int W, int Cin, int Cout) {
    WgradPlan p;
    p.PW = W < 16 ? W : 16;
    p.PH = H < 8 ? H : 8;
    p.lgPW = ilog2(p.PW);
    p.npx = W / p.PW;
    p.npy = H / p.PH;
    p.total_patches = B * p.npx * p.npy;
    const int tiles = (Cin / 64) * (Cout / 64);
    // one 8-wave workgroup per CU (LDS double-buffered inside the kernel); every workgroup costs one
    // (taps x 64 x 64) fp32 slab of reduction traffic, so do not over-split
    int nsplit = 256 / tiles;
    if (nsplit < 1) nsplit = 1;
    // small layers: a slab costs 147 KB of traffic (written, then read by the reduction) -- as much as ~18 patches of
    // operands -- so give every workgroup at least 8 patches even if that leaves CUs idle
    if (nsplit > p.total_patches / 8) nsplit = p.total_patches / 8;
    if (nsplit < 1) nsplit = 1;
    p.patches_per_wg = (p.total_patches + nsplit - 1) / nsplit;
    p.nsplit = (p.total_patches + p.patches_per_wg - 1) / p.patches_per_wg;
    return p;
}
}  // namespace

extern "C" int rgbd_conv2d_dgrad_bf16(const void* dy, const void* wp_dgrad, const void* residual, void* dx, int B, int H,
                                      int W, int Cin, int Cout, int K, int pad, int sum_pool2, void* workspace,
                                      int cus, void* stream) {
    RGBD_REQUIRE(K >= 1 && pad >= 0 && pad <= K - 1, "rgbd_conv2d_dgrad_bf16: need 0 <= pad <= K-1 (K=%d pad=%d)", K, pad);
    RGBD_REQUIRE(!(residual && sum_pool2), "rgbd_conv2d_dgrad_bf16: residual and sum_pool2 are exclusive");
    // dx = correlation of dy with the flipped, transposed kernel at padding K-1-pad
    return conv_fprop_impl(dy, wp_dgrad, nullptr, residual, dx, B, H, W, Cout, Cin, K, K, K - 1 - pad, 0, 0, 0.2f,
                           workspace, stream, cus, sum_pool2);
}

extern "C" int rgbd_conv3x3_actgrad_supported(int B, int H, int W, int Cin, int Cout) {
    return B > 0 && H > 0 && W > 0 && H % 16 == 0 && W % 16 == 0 && Cin > 0 && Cout > 0 && Cin % 64 == 0 && Cout % 64 == 0 &&
           (long)B * H * W * (Cin > Cout ? Cin : Cout) < 0x3fffffffL && !g_force_gather && g_conv_variant != 1;
}

extern "C" int rgbd_conv3x3_actgrad_bf16(const void* x, const void* wp, const void* residual, const void* act_y, float slope,
                                         float* colsum, const float* row_scale, void* y, void* y2, const float* row_scale2,
                                         int B, int H, int W, int Cin, int Cout, int cus, void* stream) {
    RGBD_REQUIRE(act_y, "rgbd_conv3x3_actgrad_bf16: null pointer");
    RGBD_REQUIRE(colsum || !row_scale, "rgbd_conv3x3_actgrad_bf16: row_scale without colsum");
    RGBD_REQUIRE(!y2 == !row_scale2, "rgbd_conv3x3_actgrad_bf16: y2 and row_scale2 come together");
    return conv_fprop_impl(x, wp, nullptr, residual, y, B, H, W, Cin, Cout, 3, 3, 1, 0, 0, slope, nullptr, stream, cus, 0, nullptr,
                           act_y, colsum, row_scale, nullptr, y2, row_scale2);
}

extern "C" int rgbd_conv2d_fprop_stats_bf16(const void* x, const void* wp, const float* bias, void* y, int64_t* stats, int B,
                                            int Hin, int Win, int Cin, int Cout, int upsample, int lrelu_channels, float slope,
                                            int cus, void* stream) {
    RGBD_REQUIRE(stats, "rgbd_conv2d_fprop_stats_bf16: null pointer");
    return conv_fprop_impl(x, wp, bias, nullptr, y, B, Hin, Win, Cin, Cout, 3, 3, 1, upsample, lrelu_channels, slope, nullptr,
                           stream, cus, 0, nullptr, nullptr, nullptr, nullptr, (long long*)stats);
}

// ---- MXFP8 forms (BASELINE configuration 5): the same launches on e4m3 operands with E8M0 block scales (csrc/mxfp8.hip)
extern "C" int rgbd_conv3x3_mxfp8_supported(int B, int Hout, int Wout, int Cin, int Cout) {
    return B > 0 && Hout > 0 && Wout > 0 && Hout % 16 == 0 && Wout % 16 == 0 && Cin > 0 && Cout > 0 && Cin % 128 == 0 &&
           Cout % 64 == 0 && (long)B * Hout * Wout * (Cin > Cout ? Cin : Cout) < 0x3fffffffL && !g_force_gather;
}

extern "C" int rgbd_conv2d_fprop_mxfp8(const void* xq, const void* xs, const void* wq, const void* ws, const float* bias,
                                       const void* residual, void* y, void* y_pooled, int B, int Hin, int Win, int Cin,
                                       int Cout, int upsample, int lrelu_channels, float slope, int cus, void* stream) {
    RGBD_REQUIRE(xs && ws, "rgbd_conv2d_fprop_mxfp8: null pointer");
    return conv_fprop_impl(xq, wq, bias, residual, y, B, Hin, Win, Cin, Cout, 3, 3, 1, upsample, lrelu_channels, slope, nullptr,
                           stream, cus, 0, y_pooled, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, xs, ws);
}

extern "C" int rgbd_conv2d_dgrad_mxfp8(const void* dyq, const void* dys, const void* wdq, const void* wds,
                                       const void* residual, void* dx, int B, int H, int W, int Cin, int Cout, int sum_pool2,
                                       int cus, void* stream) {
    RGBD_REQUIRE(dys && wds, "rgbd_conv2d_dgrad_mxfp8: null pointer");
    RGBD_REQUIRE(!(residual && sum_pool2), "rgbd_conv2d_dgrad_mxfp8: residual and sum_pool2 are exclusive");
    return conv_fprop_impl(dyq, wdq, nullptr, residual, dx, B, H, W, Cout, Cin, 3, 3, 1, 0, 0, 0.2f, nullptr, stream, cus, sum_pool2,
                           nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, dys, wds);
}

extern "C" int rgbd_conv3x3_actgrad_mxfp8(const void* xq, const void* xs, const void* wq, const void* ws, const void* residual,
                                          const void* act_y, float slope, float* colsum, const float* row_scale, void* y,
                                          void* y2, const float* row_scale2, int B, int H, int W, int Cin, int Cout,
                                          int cus, void* stream) {
    RGBD_REQUIRE(act_y && xs && ws, "rgbd_conv3x3_actgrad_mxfp8: null pointer");
    RGBD_REQUIRE(colsum || !row_scale, "rgbd_conv3x3_actgrad_mxfp8: row_scale without colsum");
    RGBD_REQUIRE(!y2 == !row_scale2, "rgbd_conv3x3_actgrad_mxfp8: y2 and row_scale2 come together");
    return conv_fprop_impl(xq, wq, nullptr, residual, y, B, H, W, Cin, Cout, 3, 3, 1, 0, 0, slope, nullptr, stream, cus, 0, nullptr,
                           act_y, colsum, row_scale, nullptr, y2, row_scale2, xs, ws);
}

extern "C" int rgbd_conv2d_fprop_stats_mxfp8(const void* xq, const void* xs, const void* wq, const void* ws, const float* bias,
                                             void* y, int64_t* stats, int B, int Hin, int Win, int Cin, int Cout, int upsample,
                                             int lrelu_channels, float slope, int cus, void* stream) {
    RGBD_REQUIRE(stats && xs && ws, "rgbd_conv2d_fprop_stats_mxfp8: null pointer");
    return conv_fprop_impl(xq, wq, bias, nullptr, y, B, Hin, Win, Cin, Cout, 3, 3, 1, upsample, lrelu_channels, slope, nullptr,
                           stream, cus, 0, nullptr, nullptr, nullptr, nullptr, (long long*)stats, nullptr, nullptr, xs, ws);
}

// Every option of the pipelined 3x3 launch behind one descriptor (the entry points above are its common special cases)
extern "C" int rgbd_conv3x3_ex(const rgbd_conv3x3_desc* d, void* stream) {
    RGBD_REQUIRE(d, "rgbd_conv3x3_ex: null descriptor");
    RGBD_REQUIRE(!d->x_scales == !d->w_scales, "rgbd_conv3x3_ex: x_scales and w_scales come together");
    RGBD_REQUIRE(!(d->act_y && d->stats), "rgbd_conv3x3_ex: act_y and stats are exclusive");
    RGBD_REQUIRE(d->colsum || !d->row_scale, "rgbd_conv3x3_ex: row_scale without colsum");
    RGBD_REQUIRE(!d->y2 == !d->row_scale2, "rgbd_conv3x3_ex: y2 and row_scale2 come together");
    return conv_fprop_impl(d->x, d->w, d->bias, d->residual, d->y, d->B, d->Hin, d->Win, d->Cin, d->Cout, 3, 3, 1, d->upsample,
                           d->lrelu_channels, d->slope, nullptr, stream, d->cus, d->pool_sum, d->y_pooled, d->act_y, d->colsum,
                           d->row_scale, (long long*)d->stats, d->y2, d->row_scale2, d->x_scales, d->w_scales, d->y_q, d->y_s,
                           d->yp_q, d->yp_s);
}

extern "C" int64_t rgbd_conv2d_wgrad_workspace(int B, int H, int W, int Cin, int Cout, int K) {
    if (B <= 0 || H < 4 || W < 4 || Cin % 64 || Cout % 64 || (K != 1 && K != 3)) return -1;
    const WgradPlan p = plan_wgrad(B, H, W, Cin, Cout);
    return (int64_t)p.nsplit * K * K * Cout * Cin * (int64_t)sizeof(float);
}

static int wgrad_partial_impl(const void* x, const void* dy, void* workspace, int B, int H, int W, int Cin, int Cout, int K,
                              int upsample, void* stream, int* nsplit_out) {
    RGBD_REQUIRE(x && dy && workspace, "rgbd_conv2d_wgrad_bf16: null pointer");
    RGBD_REQUIRE(K == 1 || K == 3, "rgbd_conv2d_wgrad_bf16: K must be 1 or 3 (K=%d)", K);
    RGBD_REQUIRE(Cin % 64 == 0 && Cout % 64 == 0,
                 "rgbd_conv2d_wgrad_bf16: Cin and Cout must be multiples of 64 (Cin=%d Cout=%d)", Cin, Cout);
    RGBD_REQUIRE(B > 0 && H >= 4 && W >= 4 && (H & (H - 1)) == 0 && (W & (W - 1)) == 0,
                 "rgbd_conv2d_wgrad_bf16: H and W must be powers of two >= 4 (H=%d W=%d)", H, W);
    const WgradPlan p = plan_wgrad(B, H, W, Cin, Cout);
    WgradArgs a;
    a.x = (const unsigned short*)x; a.dy = (const unsigned short*)dy; a.dwp = (float*)workspace;
    a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout;
    a.PW = p.PW; a.PH = p.PH; a.lgPW = p.lgPW; a.npx = p.npx; a.npy = p.npy;
    a.total_patches = p.total_patches; a.patches_per_wg = p.patches_per_wg;
    RGBD_REQUIRE((long)B * H * W * Cin < 0x3fffffffL && (long)B * H * W * Cout < 0x3fffffffL,
                 "rgbd_conv2d_wgrad_bf16: tensors must be smaller than 2 GiB (32-bit buffer offsets)");
    a.x_bytes = (int)((long)B * H * W * Cin * 2 / (upsample ? 4 : 1));
    a.y_bytes = (int)((long)B * H * W * Cout * 2);
    a.ups = upsample ? 1 : 0;
    a.ko = g_conv_variant >= 51 && g_conv_variant <= 53 ? g_conv_variant - 50 : 0;
    dim3 grid(p.nsplit, Cin / 64, Cout / 64);
    hipStream_t st = (hipStream_t)stream;
    {
        const bool fast = p.PW == 16 && p.PH == 8;
        const int lds = K == 3 ? (fast ? WGRAD9_LDS : 2 * (180 * 128 + 128 * 128)) : 2 * (128 * 128 + 128 * 128);
        const void* fn = K == 3 ? (fast ? (const void*)&conv_wgrad_kernel<9, true> : (const void*)&conv_wgrad_kernel<9, false>)
                                : (fast ? (const void*)&conv_wgrad_kernel<1, true> : (const void*)&conv_wgrad_kernel<1, false>);
        RGBD_REQUIRE(reserve_lds(fn, lds), "rgbd_conv2d_wgrad_bf16: cannot reserve %d B of LDS", lds);
#ifdef RGBD_DEBUG_BUILD
        if (K == 3 && fast && g_conv_variant == 3) {
            const int lds_old = 2 * (180 * 128 + 128 * 128);
            RGBD_REQUIRE(hipFuncSetAttribute((const void*)&conv_wgrad_tapsplit_kernel,
                                             hipFuncAttributeMaxDynamicSharedMemorySize, lds_old) == hipSuccess, "lds");
            conv_wgrad_tapsplit_kernel<<<grid, 512, lds_old, st>>>(a);
        } else
#endif
        if (K == 3) {
            if (fast) conv_wgrad_kernel<9, true><<<grid, 512, lds, st>>>(a);
            else      conv_wgrad_kernel<9, false><<<grid, 512, lds, st>>>(a);
        } else {
            if (fast) conv_wgrad_kernel<1, true><<<grid, 512, lds, st>>>(a);
            else      conv_wgrad_kernel<1, false><<<grid, 512, lds, st>>>(a);
        }
    }
    RGBD_CHECK_LAUNCH("conv_wgrad_kernel");
    *nsplit_out = p.nsplit;
    return 0;
}

extern "C" int rgbd_conv2d_wgrad_bf16(const void* x, const void* dy, void* workspace, float* dw, int B, int H, int W,
                                      int Cin, int Cout, int K, float scale, int accumulate, int upsample, void* stream) {
    RGBD_REQUIRE(dw, "rgbd_conv2d_wgrad_bf16: null pointer");
    int nsplit = 0;
    const int rc = wgrad_partial_impl(x, dy, workspace, B, H, W, Cin, Cout, K, upsample, stream, &nsplit);
    if (rc != 0) return rc;
    wgrad_reduce_finish_kernel<<<(unsigned)wgrad_reduce_blocks(K * K, Cout, Cin), 256, 0, (hipStream_t)stream>>>(
        (const float*)workspace, nsplit, dw, K * K, Cout, Cin, scale, accumulate);
    RGBD_CHECK_LAUNCH("wgrad_reduce_finish_kernel");
    return 0;
}

extern "C" int rgbd_conv2d_wgrad_partial_bf16(const void* x, const void* dy, void* workspace, int B, int H, int W,
                                              int Cin, int Cout, int K, int upsample, void* stream) {
    int nsplit = 0;
    return wgrad_partial_impl(x, dy, workspace, B, H, W, Cin, Cout, K, upsample, stream, &nsplit);
}

extern "C" int rgbd_wgrad_reduce_multi(const rgbd_wgrad_reduce_desc* descs, int n, void* stream) {
    RGBD_REQUIRE(descs && n > 0, "rgbd_wgrad_reduce_multi: no descriptors");
    for (int base = 0; base < n; base += WGRAD_MULTI_MAX) {
        WgradMultiArgs m;
        m.n = n - base < WGRAD_MULTI_MAX ? n - base : WGRAD_MULTI_MAX;
        long blocks = 0;
        for (int i = 0; i < m.n; ++i) {
            const rgbd_wgrad_reduce_desc& d = descs[base + i];
            RGBD_REQUIRE(d.workspace && d.dw && d.nsplit > 0 && (d.taps == 1 || d.taps == 9) && d.cout > 0 && d.cin > 0 &&
                         d.cin % 64 == 0 && d.cout % 64 == 0, "rgbd_wgrad_reduce_multi: bad descriptor %d", base + i);
            m.d[i] = d;
            m.block_begin[i] = (int)blocks;
            blocks += wgrad_reduce_blocks(d.taps, d.cout, d.cin);
        }
        RGBD_REQUIRE(blocks < 0x7fffffffL, "rgbd_wgrad_reduce_multi: grid too large");
        m.block_begin[m.n] = (int)blocks;
        wgrad_reduce_multi_kernel<<<(unsigned)blocks, 256, 0, (hipStream_t)stream>>>(m);
        RGBD_CHECK_LAUNCH("wgrad_reduce_multi_kernel");
    }
    return 0;
}

// ---- several weight gradients per launch (see conv_wgrad_multi_kernel)
namespace {
bool multi_eligible(const rgbd_wgrad_problem& q) {
    return q.x && q.dy && q.K == 3 && q.Cin > 0 && q.Cout > 0 && q.Cin % 64 == 0 && q.Cout % 64 == 0 && q.B > 0 && q.H >= 8 &&
           q.W >= 16 && (q.H & (q.H - 1)) == 0 && (q.W & (q.W - 1)) == 0 && (long)q.B * q.H * q.W * q.Cin < 0x3fffffffL &&
           (long)q.B * q.H * q.W * q.Cout < 0x3fffffffL;
}
// small images (4x4, 8x8, 8x... below the 8x16 patch): generic body, every problem planned on its own (plan_wgrad); one
// launch for all of them replaces a dozen latency-bound ones
bool multi_eligible_small(const rgbd_wgrad_problem& q) {
    return q.x && q.dy && q.K == 3 && q.Cin > 0 && q.Cout > 0 && q.Cin % 64 == 0 && q.Cout % 64 == 0 && q.B > 0 && q.H >= 4 &&
           q.W >= 4 && !(q.H >= 8 && q.W >= 16) && (q.H & (q.H - 1)) == 0 && (q.W & (q.W - 1)) == 0 &&
           (long)q.B * q.H * q.W * q.Cin < 0x3fffffffL && (long)q.B * q.H * q.W * q.Cout < 0x3fffffffL;
}
}  // namespace


extern "C" int rgbd_conv2d_wgrad_multi_plan(rgbd_wgrad_problem* probs, int n, int total_workgroups) {
    RGBD_REQUIRE(probs && n > 0 && n <= WGRAD_MULTI_PROBLEMS, "rgbd_conv2d_wgrad_multi_plan: 1..%d problems", WGRAD_MULTI_PROBLEMS);
    if (multi_eligible_small(probs[0])) {
        for (int i = 0; i < n; ++i) {
            RGBD_REQUIRE(multi_eligible_small(probs[i]), "rgbd_conv2d_wgrad_multi_plan: problem %d: small-image (below 8x16) "
                         "and large-image problems go into separate launches", i);
            probs[i].nsplit = plan_wgrad(probs[i].B, probs[i].H, probs[i].W, probs[i].Cin, probs[i].Cout).nsplit;
        }
        return 0;
    }
    if (total_workgroups <= 0) total_workgroups = device_cus(true);
    double units = 0.0;
    for (int i = 0; i < n; ++i) {
        RGBD_REQUIRE(multi_eligible(probs[i]), "rgbd_conv2d_wgrad_multi_plan: problem %d is not a 3x3 conv on a power-of-two "
                     "image of at least 8x16 with channels in multiples of 64", i);
        units += (double)probs[i].B * (probs[i].H / 8) * (probs[i].W / 16) * (probs[i].Cin / 64) * (probs[i].Cout / 64);
    }
    for (int i = 0; i < n; ++i) {
        rgbd_wgrad_problem& q = probs[i];
        const int patches = q.B * (q.H / 8) * (q.W / 16);
        const int tiles = (q.Cin / 64) * (q.Cout / 64);
        int nsplit = (int)((double)total_workgroups * patches / units + 0.5);      // = share of workgroups / tiles
        if (nsplit < 1) nsplit = 1;
        if (nsplit > patches) nsplit = patches;
        // workgroups per problem in multiples of 8 where the patch count allows it (conv_wgrad_multi_kernel: one eighth of
        // every problem per XCD): nsplit becomes a multiple of 8 / gcd(tiles, 8)
        const int step = tiles >= 8 ? 1 : 8 / tiles;
        if (step > 1 && patches % step == 0) {
            int cand = (nsplit + step / 2) / step * step;
            if (cand < step) cand = step;
            while (cand > step && patches % cand != 0 && (patches + cand - 1) / cand * (cand - 1) >= patches) cand -= step;
            if (cand <= patches && (patches + (patches + cand - 1) / cand - 1) / ((patches + cand - 1) / cand) == cand) nsplit = cand;
        }
        const int per_wg = (patches + nsplit - 1) / nsplit;
        q.nsplit = (patches + per_wg - 1) / per_wg;
    }
    return 0;
}

extern "C" int rgbd_conv2d_wgrad_partial_multi_bf16(const rgbd_wgrad_problem* probs, int n, void* stream) {
    RGBD_REQUIRE(probs && n > 0 && n <= WGRAD_MULTI_PROBLEMS, "rgbd_conv2d_wgrad_partial_multi_bf16: 1..%d problems",
                 WGRAD_MULTI_PROBLEMS);
    int order[WGRAD_MULTI_PROBLEMS], per_wg[WGRAD_MULTI_PROBLEMS];
    if (multi_eligible_small(probs[0])) {
        WgradMultiLaunch m;
        m.n = n;
        long wgs = 0;
        for (int k = 0; k < n; ++k) {
            const rgbd_wgrad_problem& q = probs[k];
            RGBD_REQUIRE(multi_eligible_small(q) && q.workspace, "rgbd_conv2d_wgrad_partial_multi_bf16: bad small problem %d", k);
            const WgradPlan p = plan_wgrad(q.B, q.H, q.W, q.Cin, q.Cout);
            RGBD_REQUIRE(p.nsplit == q.nsplit, "rgbd_conv2d_wgrad_partial_multi_bf16: nsplit of problem %d is not from the plan", k);
            WgradArgs& a = m.p[k];
            a.x = (const unsigned short*)q.x; a.dy = (const unsigned short*)q.dy; a.dwp = (float*)q.workspace;
            a.B = q.B; a.H = q.H; a.W = q.W; a.Cin = q.Cin; a.Cout = q.Cout;
            a.PW = p.PW; a.PH = p.PH; a.lgPW = p.lgPW; a.npx = p.npx; a.npy = p.npy;
            a.total_patches = p.total_patches; a.patches_per_wg = p.patches_per_wg;
            a.x_bytes = (int)((long)q.B * q.H * q.W * q.Cin * 2 / (q.upsample ? 4 : 1));
            a.y_bytes = (int)((long)q.B * q.H * q.W * q.Cout * 2);
            a.ups = q.upsample ? 1 : 0;
            a.ko = g_conv_variant >= 51 && g_conv_variant <= 53 ? g_conv_variant - 50 : 0;
            m.wg_begin[k] = (int)wgs;
            wgs += (long)q.nsplit * (q.Cin / 64) * (q.Cout / 64);
        }
        m.wg_begin[n] = (int)wgs;
        const int lds_small = 2 * (180 * 128 + 128 * 128);
        RGBD_REQUIRE(reserve_lds((const void*)&conv_wgrad_multi_kernel<9, false>, lds_small),
                     "rgbd_conv2d_wgrad_partial_multi_bf16: cannot reserve %d B of LDS", lds_small);
        conv_wgrad_multi_kernel<9, false><<<(unsigned)wgs, 512, lds_small, (hipStream_t)stream>>>(m);
        RGBD_CHECK_LAUNCH("conv_wgrad_multi_kernel<small>");
        return 0;
    }
    for (int i = 0; i < n; ++i) {
        const rgbd_wgrad_problem& q = probs[i];
        RGBD_REQUIRE(multi_eligible(q) && q.workspace && q.nsplit > 0, "rgbd_conv2d_wgrad_partial_multi_bf16: bad problem %d", i);
        const int patches = q.B * (q.H / 8) * (q.W / 16);
        per_wg[i] = (patches + q.nsplit - 1) / q.nsplit;
        RGBD_REQUIRE((patches + per_wg[i] - 1) / per_wg[i] == q.nsplit,
                     "rgbd_conv2d_wgrad_partial_multi_bf16: nsplit of problem %d is not from the plan", i);
        order[i] = i;
    }
    for (int i = 1; i < n; ++i)                       // largest work per workgroup first
        for (int j = i; j > 0 && per_wg[order[j]] > per_wg[order[j - 1]]; --j) {
            const int t = order[j]; order[j] = order[j - 1]; order[j - 1] = t;
        }
    WgradMultiLaunch m;
    m.n = n;
    long wgs = 0;
    for (int k = 0; k < n; ++k) {
        const rgbd_wgrad_problem& q = probs[order[k]];
        WgradArgs& a = m.p[k];
        a.x = (const unsigned short*)q.x; a.dy = (const unsigned short*)q.dy; a.dwp = (float*)q.workspace;
        a.B = q.B; a.H = q.H; a.W = q.W; a.Cin = q.Cin; a.Cout = q.Cout;
        a.PW = 16; a.PH = 8; a.lgPW = 4; a.npx = q.W / 16; a.npy = q.H / 8;
        a.total_patches = q.B * a.npx * a.npy;
        a.patches_per_wg = per_wg[order[k]];
        a.x_bytes = (int)((long)q.B * q.H * q.W * q.Cin * 2 / (q.upsample ? 4 : 1));
        a.y_bytes = (int)((long)q.B * q.H * q.W * q.Cout * 2);
        a.ups = q.upsample ? 1 : 0;
        a.ko = g_conv_variant >= 51 && g_conv_variant <= 53 ? g_conv_variant - 50 : 0;
        m.wg_begin[k] = (int)wgs;
        wgs += (long)q.nsplit * (q.Cin / 64) * (q.Cout / 64);
    }
    m.wg_begin[n] = (int)wgs;
    RGBD_REQUIRE(wgs < 0x7fffffffL, "rgbd_conv2d_wgrad_partial_multi_bf16: grid too large");
    const int lds = WGRAD9_LDS;
    RGBD_REQUIRE(reserve_lds((const void*)&conv_wgrad_multi_kernel<9, true>, lds),
                 "rgbd_conv2d_wgrad_partial_multi_bf16: cannot reserve %d B of LDS", lds);
    conv_wgrad_multi_kernel<9, true><<<(unsigned)wgs, 512, lds, (hipStream_t)stream>>>(m);
    RGBD_CHECK_LAUNCH("conv_wgrad_multi_kernel");
    return 0;
}
