// Small fused ops of the training step (RGBDUpdater.update_core, updater.py:274-448 of the reference) that sit between
// the big kernels: each replaces a run of 4-15 elementwise / reduction launches of the reference's Chainer graph with
// one launch.  All HBM- or latency-bound and tiny next to the conv engine; what they buy is launch count on the step's
// dependent chain.
#include "common.h"

namespace {

// ------------------------------------------------------------------------------------------------ real batch
// SerialIterator + TransformDataset(x / 127.5 - 1) (train_rgbd.py:308-310) + downsize_real (common/utils/pggan.py:6-50)
// from the uint8 data set resident in HBM:  out[b,c,y,x] = blend of s x s block means of data[idx[b]] / 127.5 - 1.
//   even stage: mean over the s x s block (s = H / S);  fade-in stage: (1 - alpha) * (mean over the 2s x 2s block that
//   contains the pixel = nearest-upsampled lower resolution) + alpha * (mean over the s x s block).
__global__ __launch_bounds__(256) void real_batch_kernel(const unsigned char* __restrict__ data,
                                                         const long* __restrict__ idx, float* __restrict__ out, int B,
                                                         int C, int H, int W, int S, int fade,
                                                         const float* __restrict__ alpha_ptr, float alpha_host) {
    const int s = H / S;
    const long total = (long)B * C * S * S;
    const float alpha = fade ? (alpha_ptr ? alpha_ptr[0] : alpha_host) : 1.f;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int x = (int)(e % S);
        long r = e / S;
        const int y = (int)(r % S);
        r /= S;
        const int c = (int)(r % C);
        const int b = (int)(r / C);
        const unsigned char* plane = data + ((idx[b] * C + c) * (long)H) * W;
        float hi = 0.f;
        for (int dy = 0; dy < s; ++dy)
            for (int dx = 0; dx < s; ++dx)
                hi += __fdiv_rn((float)plane[(long)(y * s + dy) * W + x * s + dx], 127.5f) - 1.f;
        hi *= 1.f / (float)(s * s);
        float v = hi;
        if (fade) {
            const int s2 = 2 * s, y2 = (y >> 1) * s2, x2 = (x >> 1) * s2;
            float lo = 0.f;
            for (int dy = 0; dy < s2; ++dy)
                for (int dx = 0; dx < s2; ++dx)
                    lo += __fdiv_rn((float)plane[(long)(y2 + dy) * W + x2 + dx], 127.5f) - 1.f;
            lo *= 1.f / (float)(s2 * s2);
            v = (1.f - alpha) * lo + alpha * hi;
        }
        out[e] = v;
    }
}

// ------------------------------------------------------------------------------------------------ zero several buffers
constexpr int ZERO_MULTI_MAX = 8;
struct ZeroMultiArgs {
    float* p[ZERO_MULTI_MAX];
    long n[ZERO_MULTI_MAX];
    int count;
};
__global__ __launch_bounds__(256) void zero_multi_kernel(ZeroMultiArgs a) {
    for (int k = 0; k < a.count; ++k) {
        f32x4* p4 = reinterpret_cast<f32x4*>(a.p[k]);
        const long n4 = a.n[k] >> 2;                       // buffers are 16-byte aligned, sizes multiples of 4 floats
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256)
            p4[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (long i = (n4 << 2) + (long)blockIdx.x * 256 + threadIdx.x; i < a.n[k]; i += (long)gridDim.x * 256)
            a.p[k][i] = 0.f;
    }
}

// ------------------------------------------------------------------------------------------------ latent normalisation
// make_hidden (net.py:333-343): z / sqrt(sum_c z^2 / ch + 1e-8), one wave per row; the row is written `copies` times,
// copy k at row index m + k * M (updater.py:300 repeats the same latents for the second view of every pair).
__global__ __launch_bounds__(256) void hidden_normalize_kernel(const float* __restrict__ z, float* __restrict__ out,
                                                               int M, int C, float ch, int copies) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= M) return;
    float acc = 0.f;
    for (int c = lane; c < C; c += 64) {
        const float v = z[(long)row * C + c];
        acc += v * v;
    }
    acc = wave_sum(acc);
    const float inv = 1.f / sqrtf(acc / ch + 1e-8f);
    for (int c = lane; c < C; c += 64) {
        const float v = z[(long)row * C + c] * inv;
        for (int k = 0; k < copies; ++k) out[((long)k * M + row) * C + c] = v;
    }
}

// make_hidden with the draw inside (net.py:333-343: xp.random.normal, then the normalisation above): Philox4x32-10 keyed by a
// seed, counter = (launch number, row, column quad), Box-Muller on the four words.  The launch number lives in device memory
// (state[2]) and is bumped by the LAST block of every launch (ticket in state[3]), so a step replayed from a captured graph
// draws fresh latents on every replay -- what torch.randn's graph-safe generator did for the kernel this replaces (the last
// torch arithmetic kernel on the generator's input side; this library is built without packed-fp32 instructions, DESIGN.md
// section 3).  One wave per row; the row (C <= 1024 values) is held in registers between the two passes.
__device__ __forceinline__ void philox4x32_10(unsigned k0, unsigned k1, unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned (&out)[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
__global__ __launch_bounds__(256) void hidden_draw_kernel(unsigned* __restrict__ state, float* __restrict__ out, int M, int C,
                                                          float ch, int copies) {
    const unsigned launch = *reinterpret_cast<volatile unsigned*>(state + 2);
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row < M) {
        float v[16];                                   // quads lane, lane + 64, ... of this row: up to 4 quads = 16 values
        float acc = 0.f;
        const int quads = C >> 2;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int qd = lane + 64 * i;
            unsigned w[4];
            philox4x32_10(state[0], state[1], launch, (unsigned)row, (unsigned)qd, 0x52474244u, w);
#pragma unroll
            for (int h = 0; h < 2; ++h) {              // Box-Muller: (0,1] x [0,1) -> two N(0,1) values
                const float u1 = ((float)(w[2 * h] >> 8) + 1.0f) * (1.0f / 16777216.0f);
                const float u2 = (float)(w[2 * h + 1] >> 8) * (1.0f / 16777216.0f);
                const float rad = sqrtf(-2.0f * logf(u1));
                float sn, cs;
                sincosf(6.283185307179586f * u2, &sn, &cs);
                v[4 * i + 2 * h] = qd < quads ? rad * cs : 0.f;
                v[4 * i + 2 * h + 1] = qd < quads ? rad * sn : 0.f;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) acc += v[4 * i + k] * v[4 * i + k];
        }
        acc = wave_sum(acc);
        const float inv = 1.f / sqrtf(acc / ch + 1e-8f);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int qd = lane + 64 * i;
            if (qd < quads) {
                const f32x4 o = {v[4 * i] * inv, v[4 * i + 1] * inv, v[4 * i + 2] * inv, v[4 * i + 3] * inv};
                for (int k = 0; k < copies; ++k) *reinterpret_cast<f32x4*>(out + ((long)k * M + row) * C + 4 * qd) = o;
            }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned t = atomicAdd(state + 3, 1u);
        if (t == gridDim.x - 1) {                      // every other block has read `launch` and is done: the next launch number
            state[3] = 0u;
            state[2] = launch + 1u;
        }
    }
}

// ------------------------------------------------------------------------------------------------ R1 penalty
// updater.py:416-418 + loss_functions.py:7-8: loss = coef * (1/B) * sum_b (sqrt(sum g_b^2))^2 over g (B, n) fp32.
constexpr int R1_CHUNKS = 16;
__global__ __launch_bounds__(256) void r1_partial_kernel(const float* __restrict__ g, long n, float* __restrict__ part) {
    const int b = blockIdx.y, chunk = blockIdx.x;
    const long n4 = n >> 2;
    const f32x4* g4 = reinterpret_cast<const f32x4*>(g + (long)b * n);
    const long per = (n4 + R1_CHUNKS - 1) / R1_CHUNKS;
    const long i0 = chunk * per, i1 = i0 + per < n4 ? i0 + per : n4;
    float acc = 0.f;
    for (long i = i0 + threadIdx.x; i < i1; i += 256) {
        const f32x4 v = g4[i];
        acc += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
    }
    if (chunk == R1_CHUNKS - 1)
        for (long i = (n4 << 2) + threadIdx.x; i < n; i += 256) acc += g[(long)b * n + i] * g[(long)b * n + i];
    acc = wave_sum(acc);
    __shared__ float red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) part[b * R1_CHUNKS + chunk] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(64) void r1_final_kernel(const float* __restrict__ part, int B, float coef,
                                                      float* __restrict__ loss) {
    float acc = 0.f;
    for (int b = threadIdx.x; b < B; b += 64) {
        float s = 0.f;
        for (int k = 0; k < R1_CHUNKS; ++k) s += part[b * R1_CHUNKS + k];
        const float nrm = sqrtf(s);                        // grad_l2 = F.sqrt(F.sum(g ** 2)); loss_l2 squares it again
        acc += nrm * nrm;
    }
    acc = wave_sum(acc);
    if (threadIdx.x == 0) loss[0] = coef * (acc / (float)B);
}
// out = (gl[0] * k) * x : gradient of the penalty w.r.t. g (k = 2 * coef / B), gl = d objective / d loss on the device
__global__ __launch_bounds__(256) void scale_by_scalar_kernel(const float* __restrict__ x, const float* __restrict__ gl,
                                                              float k, float* __restrict__ out, long n) {
    const float s = (gl ? gl[0] : 1.f) * k;
    const long n4 = n >> 2;
    const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
    f32x4* o4 = reinterpret_cast<f32x4*>(out);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) o4[i] = x4[i] * s;
    for (long i = (n4 << 2) + (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) out[i] = x[i] * s;
}

// out = a + s[row] * x over (rows, row_len) fp32 (s NULL: s = 1; out may alias a): the merge of D's two gradient buffers at
// the join, and the operand  dout + s_b * x_real  of the adversarial injection at the image planes (functional._ToPlanes).
// (torch's add / mul kernels did this in rounds 1-2; the library's own kernels are built without packed-fp32
// instructions)
__global__ __launch_bounds__(256) void axpy_rows_f32_kernel(const float* a, const float* __restrict__ x,      // `out` may BE `a`
                                                            const float* __restrict__ s, float* out,             // (no restrict)
                                                            long rows, long row_len) {
    const long n4 = rows * row_len >> 2, r4 = row_len >> 2;
    const f32x4* a4 = reinterpret_cast<const f32x4*>(a);
    const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
    f32x4* o4 = reinterpret_cast<f32x4*>(out);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const float sv = s ? s[i / r4] : 1.f;
        const f32x4 av = a4[i], xv = x4[i];
        o4[i] = f32x4{av[0] + sv * xv[0], av[1] + sv * xv[1], av[2] + sv * xv[2], av[3] + sv * xv[3]};
    }
}

// ------------------------------------------------------------------------------------------------ image gradient junction
// The generator's output gradient before the 3-D consistency loss adds its part (updater.py:334,363-365,387):
//   out[b, k, p] = ratio[b] * gx[b, k, p]  for k < 3  (adversarial image gradient, rescaled per sample -- updater.py of
//   this engine: one pass through D(x_fake) seeded with the discriminator's loss),  out[b, 3, p] = 0.
__global__ __launch_bounds__(256) void image_grad_init_kernel(const float* __restrict__ gx, const float* __restrict__ ratio,
                                                              float* __restrict__ out, int B, int KP_in, int KP_out,
                                                              int HW) {
    const long total = (long)B * KP_out * HW;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int p = (int)(e % HW);
        const long r = e / HW;
        const int k = (int)(r % KP_out);
        const int b = (int)(r / KP_out);
        out[e] = k < KP_in ? gx[((long)b * KP_in + k) * HW + p] * (ratio ? ratio[b] : 1.f) : 0.f;
    }
}

// ------------------------------------------------------------------------------------------------ constant input
// SynthesisBlock 0 (net.py:130-153): h = lrelu(W + b0) broadcast over the batch, W (C, HW) fp32 -> (B, HW, C) bf16.
__global__ __launch_bounds__(256) void const_input_fwd_kernel(const float* __restrict__ w, const float* __restrict__ bias,
                                                              unsigned short* __restrict__ out, int B, int HW, int C,
                                                              float slope) {
    const long total = (long)B * HW * C;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int c = (int)(e % C);
        const int p = (int)((e / C) % HW);
        const float v = w[(long)c * HW + p] + bias[c];
        out[e] = f32_to_bf16_bits(v > 0.f ? v : v * slope);
    }
}
// dW[c,p] += m(c,p) * sum_b dh[b,p,c];  db[c] += sum_p of that.  One block per position p, one thread per channel (the
// reads of a sample row are coalesced); the HW partial bias sums meet through fp32 atomics (16 adders per address).
__global__ __launch_bounds__(256) void const_input_bwd_kernel(const unsigned short* __restrict__ dh,
                                                              const float* __restrict__ w, const float* __restrict__ bias,
                                                              float* __restrict__ dw, float* __restrict__ db, int B,
                                                              int HW, int C, float slope) {
    const int p = blockIdx.x;
    for (int c = blockIdx.y * 256 + threadIdx.x; c < C; c += gridDim.y * 256) {
        float s = 0.f;
        for (int b = 0; b < B; ++b) s += bf16_bits_to_f32(dh[((long)b * HW + p) * C + c]);
        const float g = (w[(long)c * HW + p] + bias[c]) > 0.f ? s : s * slope;
        if (dw) dw[(long)c * HW + p] += g;
        if (db) atomicAdd(db + c, g);
    }
}

// ------------------------------------------------------------------------------------------------ dense tail layout
// 4x4-valid conv of the discriminator's base block as a linear layer over (ci, kh, kw) (net.py:363-365,372-377): the
// activation (B, HW, C) bf16 NHWC is brought into the weight's own (C, HW) order as fp32 rows, and back (adjoint).
__global__ __launch_bounds__(256) void nhwc_to_rows_kernel(const unsigned short* __restrict__ h, float* __restrict__ out,
                                                           int B, int HW, int C) {
    const long total = (long)B * HW * C;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int p = (int)(e % HW);
        const long r = e / HW;
        const int c = (int)(r % C);
        const int b = (int)(r / C);
        out[e] = bf16_bits_to_f32(h[((long)b * HW + p) * C + c]);
    }
}
__global__ __launch_bounds__(256) void rows_to_nhwc_kernel(const float* __restrict__ rows, unsigned short* __restrict__ h,
                                                           int B, int HW, int C) {
    const long total = (long)B * HW * C;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int c = (int)(e % C);
        const long r = e / C;
        const int p = (int)(r % HW);
        const int b = (int)(r / HW);
        h[e] = f32_to_bf16_bits(rows[((long)b * C + c) * HW + p]);
    }
}

// ------------------------------------------------------------------------------------------------ channel-wise L2 normalise
// DCGANBlock (net.py:621-648): F.normalize over channels, y = x / (||x||_2 + 1e-5), on NHWC bf16 (fp32 norm).
// LPP lanes own one pixel (8 channels each); the norm meets through wave shuffles.
template <int LPP>
__global__ __launch_bounds__(256) void l2norm_fwd_kernel(const unsigned short* __restrict__ x,
                                                         unsigned short* __restrict__ y, long npix, float eps) {
    constexpr int PPB = 256 / LPP;
    const int sub = threadIdx.x % LPP;
    for (long pix = (long)blockIdx.x * PPB + threadIdx.x / LPP; pix < npix; pix += (long)gridDim.x * PPB) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(x + (pix * LPP + sub) * 8);
        float f[8], ss = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) { f[2 * k] = bf16_lo(v[k]); f[2 * k + 1] = bf16_hi(v[k]); }
#pragma unroll
        for (int k = 0; k < 8; ++k) ss += f[k] * f[k];
#pragma unroll
        for (int off = LPP / 2; off > 0; off >>= 1) ss += __shfl_xor(ss, off, 64);
        const float inv = 1.f / (sqrtf(ss) + eps);
        u32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = pack_bf16x2(f[2 * k] * inv, f[2 * k + 1] * inv);
        *reinterpret_cast<u32x4*>(y + (pix * LPP + sub) * 8) = o;
    }
}
// dx = dy / d - x * (x . dy) / (n * d^2),  n = ||x||, d = n + eps
template <int LPP>
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const unsigned short* __restrict__ x,
                                                         const unsigned short* __restrict__ dy,
                                                         unsigned short* __restrict__ dx, long npix, float eps) {
    constexpr int PPB = 256 / LPP;
    const int sub = threadIdx.x % LPP;
    for (long pix = (long)blockIdx.x * PPB + threadIdx.x / LPP; pix < npix; pix += (long)gridDim.x * PPB) {
        const long o8 = (pix * LPP + sub) * 8;
        const u32x4 v = *reinterpret_cast<const u32x4*>(x + o8), g = *reinterpret_cast<const u32x4*>(dy + o8);
        float f[8], gg[8], ss = 0.f, dot = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            f[2 * k] = bf16_lo(v[k]); f[2 * k + 1] = bf16_hi(v[k]);
            gg[2 * k] = bf16_lo(g[k]); gg[2 * k + 1] = bf16_hi(g[k]);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) { ss += f[k] * f[k]; dot += f[k] * gg[k]; }
#pragma unroll
        for (int off = LPP / 2; off > 0; off >>= 1) { ss += __shfl_xor(ss, off, 64); dot += __shfl_xor(dot, off, 64); }
        const float n = sqrtf(ss), d = n + eps;
        const float a = 1.f / d, c = n > 0.f ? dot / (n * d * d) : 0.f;
        u32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = pack_bf16x2(gg[2 * k] * a - f[2 * k] * c, gg[2 * k + 1] * a - f[2 * k + 1] * c);
        *reinterpret_cast<u32x4*>(dx + o8) = o;
    }
}

// ------------------------------------------------------------------------------------------------ 3x3 binomial blur
// rescale.py:20-25 (blur = depthwise [1 2 1] x [1 2 1] / 16, zero padding 1) on NHWC bf16, used when enable_blur is set:
//   mode 0: y = blur(x)                                   (after downscale2x, net.py:422-423; self-adjoint)
//   mode 1: y = blur(upscale2x(x)), x (B,H/2,W/2,C)       (net.py:140-141; the upsampled tensor is never written)
//   mode 2: y = sum_{2x2}(blur(x)), y (B,H/2,W/2,C)       (adjoint of mode 1)
__global__ __launch_bounds__(256) void blur3x3_kernel(const unsigned short* __restrict__ x, unsigned short* __restrict__ y,
                                                      int B, int H, int W, int C, int mode) {
    // H, W: size of the image the blur acts on; mode 1 reads (H/2, W/2), mode 2 writes (H/2, W/2)
    const int cvec = C >> 3;
    const int Ho = mode == 2 ? H >> 1 : H, Wo = mode == 2 ? W >> 1 : W;
    const long total = (long)B * Ho * Wo * cvec;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int cv = (int)(e % cvec);
        long r = e / cvec;
        const int xo = (int)(r % Wo);
        r /= Wo;
        const int yo = (int)(r % Ho);
        const int b = (int)(r / Ho);
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const int reps = mode == 2 ? 2 : 1;
        for (int sy = 0; sy < reps; ++sy)
            for (int sx = 0; sx < reps; ++sx) {
                const int cy = mode == 2 ? 2 * yo + sy : yo, cx = mode == 2 ? 2 * xo + sx : xo;
#pragma unroll
                for (int i = -1; i <= 1; ++i)
#pragma unroll
                    for (int j = -1; j <= 1; ++j) {
                        const int yy = cy + i, xx = cx + j;
                        if ((unsigned)yy >= (unsigned)H || (unsigned)xx >= (unsigned)W) continue;
                        const float wgt = (float)((2 - (i < 0 ? -i : i)) * (2 - (j < 0 ? -j : j))) * (1.f / 16.f);
                        const long src = mode == 1 ? (((long)b * (H >> 1) + (yy >> 1)) * (W >> 1) + (xx >> 1))
                                                   : (((long)b * H + yy) * W + xx);
                        const u32x4 v = *reinterpret_cast<const u32x4*>(x + (src * cvec + cv) * 8);
#pragma unroll
                        for (int k = 0; k < 4; ++k) { acc[2 * k] += wgt * bf16_lo(v[k]); acc[2 * k + 1] += wgt * bf16_hi(v[k]); }
                    }
            }
        u32x4 o = {pack_bf16x2(acc[0], acc[1]), pack_bf16x2(acc[2], acc[3]), pack_bf16x2(acc[4], acc[5]),
                   pack_bf16x2(acc[6], acc[7])};
        *reinterpret_cast<u32x4*>(y + e * 8) = o;
    }
}

// ------------------------------------------------------------------------------------------------ progressive fade-in
// Odd stages blend two resolutions (net.py:283-290 generator, :490-497 discriminator) with alpha = stage - floor(stage),
// read from a device float when given (a captured graph then follows the schedule) else from the host value.
//   fade_planes:  out[b,c,y,x] = (1-a) * lo[b,c,y/2,x/2] + a * hi[b,c,y,x]            NCHW fp32 planes (generator)
//   its backward: dhi = a * dout;  dlo[b,c,y,x] = (1-a) * sum_{2x2} dout
//   lerp_bf16:    out = (1-a) * p + a * q   on bf16 tensors (discriminator features);  split: (1-a) g, a g
//   pool2_planes: out[b,c,y,x] = 0.25 * sum_{2x2} x   (downscale2x of the image, net.py:491) and its adjoint
__device__ __forceinline__ float fade_alpha(const float* dev, float host) { return dev ? dev[0] : host; }

__global__ __launch_bounds__(256) void fade_planes_fwd_kernel(const float* __restrict__ lo, const float* __restrict__ hi,
                                                              float* __restrict__ out, long planes, int H, int W,
                                                              const float* __restrict__ adev, float ahost) {
    const float a = fade_alpha(adev, ahost);
    const long total = planes * H * W;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int x = (int)(e % W);
        const long r = e / W;
        const int y = (int)(r % H);
        const long pl = r / H;
        out[e] = (1.f - a) * lo[(pl * (H >> 1) + (y >> 1)) * (W >> 1) + (x >> 1)] + a * hi[e];
    }
}
__global__ __launch_bounds__(256) void fade_planes_bwd_kernel(const float* __restrict__ dout, float* __restrict__ dlo,
                                                              float* __restrict__ dhi, long planes, int H, int W,
                                                              const float* __restrict__ adev, float ahost) {
    const float a = fade_alpha(adev, ahost);
    const int h2 = H >> 1, w2 = W >> 1;
    const long total = planes * h2 * w2;          // one thread per low-resolution pixel: its 2x2 block of dout
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int x = (int)(e % w2);
        const long r = e / w2;
        const int y = (int)(r % h2);
        const long pl = r / h2;
        const long o = (pl * H + 2 * y) * W + 2 * x;
        const float g00 = dout[o], g01 = dout[o + 1], g10 = dout[o + W], g11 = dout[o + W + 1];
        if (dlo) dlo[e] = (1.f - a) * ((g00 + g01) + (g10 + g11));
        if (dhi) { dhi[o] = a * g00; dhi[o + 1] = a * g01; dhi[o + W] = a * g10; dhi[o + W + 1] = a * g11; }
    }
}
// mode 0: out = (1-a) p + a q;  mode 1: out = (1-a) p, out2 = a p   (8 bf16 per thread)
__global__ __launch_bounds__(256) void lerp_bf16_kernel(const unsigned short* __restrict__ p, const unsigned short* __restrict__ q,
                                                        unsigned short* __restrict__ out, unsigned short* __restrict__ out2,
                                                        long nvec, int mode, const float* __restrict__ adev, float ahost) {
    const float a = fade_alpha(adev, ahost);
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < nvec; e += (long)gridDim.x * 256) {
        const u32x4 pv = *reinterpret_cast<const u32x4*>(p + e * 8);
        u32x4 o, o2;
        if (mode == 0) {
            const u32x4 qv = *reinterpret_cast<const u32x4*>(q + e * 8);
#pragma unroll
            for (int k = 0; k < 4; ++k)
                o[k] = pack_bf16x2((1.f - a) * bf16_lo(pv[k]) + a * bf16_lo(qv[k]), (1.f - a) * bf16_hi(pv[k]) + a * bf16_hi(qv[k]));
            *reinterpret_cast<u32x4*>(out + e * 8) = o;
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                o[k] = pack_bf16x2((1.f - a) * bf16_lo(pv[k]), (1.f - a) * bf16_hi(pv[k]));
                o2[k] = pack_bf16x2(a * bf16_lo(pv[k]), a * bf16_hi(pv[k]));
            }
            *reinterpret_cast<u32x4*>(out + e * 8) = o;
            *reinterpret_cast<u32x4*>(out2 + e * 8) = o2;
        }
    }
}
// adjoint == 0: out (planes,H/2,W/2) = 0.25 * 2x2 sums of x (planes,H,W);  adjoint != 0: out (planes,H,W) = 0.25 * x[y/2,x/2]
__global__ __launch_bounds__(256) void pool2_planes_kernel(const float* __restrict__ x, float* __restrict__ out, long planes,
                                                           int H, int W, int adjoint) {
    const int h2 = H >> 1, w2 = W >> 1;
    const long total = planes * h2 * w2;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int xx = (int)(e % w2);
        const long r = e / w2;
        const int yy = (int)(r % h2);
        const long pl = r / h2;
        const long o = (pl * H + 2 * yy) * W + 2 * xx;
        if (!adjoint) {
            out[e] = 0.25f * ((x[o] + x[o + 1]) + (x[o + W] + x[o + W + 1]));
        } else {
            const float v = 0.25f * x[e];
            out[o] = v; out[o + 1] = v; out[o + W] = v; out[o + W + 1] = v;
        }
    }
}

inline unsigned grid_for(long n, long cap = 4096) {
    const long b = (n + 255) / 256;
    return (unsigned)(b < 1 ? 1 : (b < cap ? b : cap));
}

}  // namespace

extern "C" int rgbd_real_batch_u8(const uint8_t* data, const int64_t* idx, float* out, int B, int C, int H, int W, int S,
                                  int fade, const float* alpha_device, float alpha, void* stream) {
    RGBD_REQUIRE(data && idx && out, "rgbd_real_batch_u8: null pointer");
    RGBD_REQUIRE(B > 0 && C > 0 && H == W && S > 0 && H % S == 0 && (!fade || (S % 2 == 0 && H % S == 0)),
                 "rgbd_real_batch_u8: bad shape H=%d W=%d S=%d fade=%d", H, W, S, fade);
    real_batch_kernel<<<grid_for((long)B * C * S * S), 256, 0, (hipStream_t)stream>>>(
        data, (const long*)idx, out, B, C, H, W, S, fade ? 1 : 0, alpha_device, alpha);
    RGBD_CHECK_LAUNCH("real_batch_kernel");
    return 0;
}

extern "C" int rgbd_zero_multi_f32(float* const* ptrs, const int64_t* counts, int n, void* stream) {
    RGBD_REQUIRE(ptrs && counts && n > 0 && n <= ZERO_MULTI_MAX, "rgbd_zero_multi_f32: 1..%d buffers", ZERO_MULTI_MAX);
    ZeroMultiArgs a;
    long most = 0;
    for (int k = 0; k < n; ++k) {
        RGBD_REQUIRE(ptrs[k] && counts[k] > 0 && ((uintptr_t)ptrs[k] & 15) == 0, "rgbd_zero_multi_f32: buffer %d", k);
        a.p[k] = ptrs[k];
        a.n[k] = counts[k];
        if (counts[k] > most) most = counts[k];
    }
    a.count = n;
    zero_multi_kernel<<<grid_for(most / 4, 2048), 256, 0, (hipStream_t)stream>>>(a);
    RGBD_CHECK_LAUNCH("zero_multi_kernel");
    return 0;
}

extern "C" int rgbd_hidden_normalize(const float* z, float* out, int M, int C, float ch, int copies, void* stream) {
    RGBD_REQUIRE(z && out && M > 0 && C > 0 && ch > 0.f && copies >= 1, "rgbd_hidden_normalize: bad arguments");
    hidden_normalize_kernel<<<(M + 3) / 4, 256, 0, (hipStream_t)stream>>>(z, out, M, C, ch, copies);
    RGBD_CHECK_LAUNCH("hidden_normalize_kernel");
    return 0;
}

extern "C" int rgbd_hidden_draw(uint32_t* state, float* out, int M, int C, float ch, int copies, void* stream) {
    RGBD_REQUIRE(state && out && M > 0 && C > 0 && C % 4 == 0 && C <= 1024 && ch > 0.f && copies >= 1 && ((uintptr_t)out & 15) == 0,
                 "rgbd_hidden_draw: bad arguments (C must be a multiple of 4, at most 1024; C=%d)", C);
    hidden_draw_kernel<<<(M + 3) / 4, 256, 0, (hipStream_t)stream>>>(state, out, M, C, ch, copies);
    RGBD_CHECK_LAUNCH("hidden_draw_kernel");
    return 0;
}

extern "C" int rgbd_r1_penalty_fwd(const float* g, int B, int64_t n, float coef, float* workspace, float* loss,
                                   void* stream) {
    RGBD_REQUIRE(g && workspace && loss && B > 0 && n > 0 && ((uintptr_t)g & 15) == 0 && (n & 3) == 0,
                 "rgbd_r1_penalty_fwd: bad arguments (n must be a multiple of 4)");
    r1_partial_kernel<<<dim3(R1_CHUNKS, B), 256, 0, (hipStream_t)stream>>>(g, n, workspace);
    RGBD_CHECK_LAUNCH("r1_partial_kernel");
    r1_final_kernel<<<1, 64, 0, (hipStream_t)stream>>>(workspace, B, coef, loss);
    RGBD_CHECK_LAUNCH("r1_final_kernel");
    return 0;
}

extern "C" int rgbd_scale_by_scalar_f32(const float* x, const float* scalar_device, float k, float* out, int64_t n,
                                        void* stream) {
    RGBD_REQUIRE(x && out && n > 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)out & 15) == 0,
                 "rgbd_scale_by_scalar_f32: bad arguments");
    scale_by_scalar_kernel<<<grid_for(n / 4), 256, 0, (hipStream_t)stream>>>(x, scalar_device, k, out, n);
    RGBD_CHECK_LAUNCH("scale_by_scalar_kernel");
    return 0;
}

extern "C" int rgbd_axpy_rows_f32(const float* a, const float* x, const float* s, float* out, int64_t rows, int64_t row_len,
                                  void* stream) {
    RGBD_REQUIRE(a && x && out && rows > 0 && row_len > 0 && row_len % 4 == 0, "rgbd_axpy_rows_f32: bad arguments");
    RGBD_REQUIRE((((uintptr_t)a | (uintptr_t)x | (uintptr_t)out) & 15) == 0, "rgbd_axpy_rows_f32: buffers must be 16-byte aligned");
    axpy_rows_f32_kernel<<<grid_for(rows * row_len / 4), 256, 0, (hipStream_t)stream>>>(a, x, s, out, rows, row_len);
    RGBD_CHECK_LAUNCH("axpy_rows_f32_kernel");
    return 0;
}

extern "C" int rgbd_image_grad_init(const float* gx, const float* ratio, float* out, int B, int KP_in, int KP_out, int HW,
                                    void* stream) {
    RGBD_REQUIRE(gx && out && B > 0 && KP_in > 0 && KP_out >= KP_in && HW > 0, "rgbd_image_grad_init: bad arguments");
    image_grad_init_kernel<<<grid_for((long)B * KP_out * HW), 256, 0, (hipStream_t)stream>>>(gx, ratio, out, B, KP_in,
                                                                                           KP_out, HW);
    RGBD_CHECK_LAUNCH("image_grad_init_kernel");
    return 0;
}

extern "C" int rgbd_const_input_fwd(const float* w, const float* bias, void* out, int B, int HW, int C, float slope,
                                    void* stream) {
    RGBD_REQUIRE(w && bias && out && B > 0 && HW > 0 && C > 0, "rgbd_const_input_fwd: bad arguments");
    const_input_fwd_kernel<<<grid_for((long)B * HW * C), 256, 0, (hipStream_t)stream>>>(w, bias, (unsigned short*)out, B,
                                                                                      HW, C, slope);
    RGBD_CHECK_LAUNCH("const_input_fwd_kernel");
    return 0;
}

extern "C" int rgbd_const_input_bwd(const void* dh, const float* w, const float* bias, float* dw, float* db, int B, int HW,
                                    int C, float slope, void* stream) {
    RGBD_REQUIRE(dh && w && bias && B > 0 && HW > 0 && C > 0, "rgbd_const_input_bwd: bad arguments");
    const_input_bwd_kernel<<<dim3(HW, (C + 255) / 256), 256, 0, (hipStream_t)stream>>>((const unsigned short*)dh, w, bias, dw,
                                                                                       db, B, HW, C, slope);
    RGBD_CHECK_LAUNCH("const_input_bwd_kernel");
    return 0;
}

extern "C" int rgbd_nhwc_to_rows_f32(const void* h, float* rows, int B, int HW, int C, void* stream) {
    RGBD_REQUIRE(h && rows && B > 0 && HW > 0 && C > 0, "rgbd_nhwc_to_rows_f32: bad arguments");
    nhwc_to_rows_kernel<<<grid_for((long)B * HW * C), 256, 0, (hipStream_t)stream>>>((const unsigned short*)h, rows, B, HW, C);
    RGBD_CHECK_LAUNCH("nhwc_to_rows_kernel");
    return 0;
}

extern "C" int rgbd_rows_to_nhwc_bf16(const float* rows, void* h, int B, int HW, int C, void* stream) {
    RGBD_REQUIRE(h && rows && B > 0 && HW > 0 && C > 0, "rgbd_rows_to_nhwc_bf16: bad arguments");
    rows_to_nhwc_kernel<<<grid_for((long)B * HW * C), 256, 0, (hipStream_t)stream>>>(rows, (unsigned short*)h, B, HW, C);
    RGBD_CHECK_LAUNCH("rows_to_nhwc_kernel");
    return 0;
}

extern "C" int rgbd_l2norm_fwd(const void* x, void* y, int64_t npix, int C, float eps, void* stream) {
    RGBD_REQUIRE(x && y && npix > 0 && (C == 128 || C == 256 || C == 512), "rgbd_l2norm_fwd: C must be 128, 256 or 512 (C=%d)", C);
    hipStream_t st = (hipStream_t)stream;
    const int lpp = C / 8;
    const unsigned blocks = grid_for(npix * lpp);
    if (lpp == 16)      l2norm_fwd_kernel<16><<<blocks, 256, 0, st>>>((const unsigned short*)x, (unsigned short*)y, npix, eps);
    else if (lpp == 32) l2norm_fwd_kernel<32><<<blocks, 256, 0, st>>>((const unsigned short*)x, (unsigned short*)y, npix, eps);
    else                l2norm_fwd_kernel<64><<<blocks, 256, 0, st>>>((const unsigned short*)x, (unsigned short*)y, npix, eps);
    RGBD_CHECK_LAUNCH("l2norm_fwd_kernel");
    return 0;
}

extern "C" int rgbd_l2norm_bwd(const void* x, const void* dy, void* dx, int64_t npix, int C, float eps, void* stream) {
    RGBD_REQUIRE(x && dy && dx && npix > 0 && (C == 128 || C == 256 || C == 512),
                 "rgbd_l2norm_bwd: C must be 128, 256 or 512 (C=%d)", C);
    hipStream_t st = (hipStream_t)stream;
    const int lpp = C / 8;
    const unsigned blocks = grid_for(npix * lpp);
    const unsigned short *xs = (const unsigned short*)x, *gs = (const unsigned short*)dy;
    if (lpp == 16)      l2norm_bwd_kernel<16><<<blocks, 256, 0, st>>>(xs, gs, (unsigned short*)dx, npix, eps);
    else if (lpp == 32) l2norm_bwd_kernel<32><<<blocks, 256, 0, st>>>(xs, gs, (unsigned short*)dx, npix, eps);
    else                l2norm_bwd_kernel<64><<<blocks, 256, 0, st>>>(xs, gs, (unsigned short*)dx, npix, eps);
    RGBD_CHECK_LAUNCH("l2norm_bwd_kernel");
    return 0;
}

extern "C" int rgbd_blur3x3_bf16(const void* x, void* y, int B, int H, int W, int C, int mode, void* stream) {
    RGBD_REQUIRE(x && y && B > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0 && mode >= 0 && mode <= 2 &&
                 (mode == 0 || (H % 2 == 0 && W % 2 == 0)), "rgbd_blur3x3_bf16: bad arguments (H=%d W=%d C=%d mode=%d)", H, W, C, mode);
    const long outs = (long)B * (mode == 2 ? H / 2 : H) * (mode == 2 ? W / 2 : W) * (C / 8);
    blur3x3_kernel<<<grid_for(outs, 8192), 256, 0, (hipStream_t)stream>>>((const unsigned short*)x, (unsigned short*)y, B, H,
                                                                        W, C, mode);
    RGBD_CHECK_LAUNCH("blur3x3_kernel");
    return 0;
}

extern "C" int rgbd_fade_planes_fwd(const float* lo, const float* hi, float* out, int64_t planes, int H, int W,
                                    const float* alpha_device, float alpha, void* stream) {
    RGBD_REQUIRE(lo && hi && out && planes > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0, "rgbd_fade_planes_fwd: bad arguments");
    fade_planes_fwd_kernel<<<grid_for(planes * H * W), 256, 0, (hipStream_t)stream>>>(lo, hi, out, planes, H, W, alpha_device, alpha);
    RGBD_CHECK_LAUNCH("fade_planes_fwd_kernel");
    return 0;
}

extern "C" int rgbd_fade_planes_bwd(const float* dout, float* dlo, float* dhi, int64_t planes, int H, int W,
                                    const float* alpha_device, float alpha, void* stream) {
    RGBD_REQUIRE(dout && (dlo || dhi) && planes > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0, "rgbd_fade_planes_bwd: bad arguments");
    fade_planes_bwd_kernel<<<grid_for(planes * (H / 2) * (W / 2)), 256, 0, (hipStream_t)stream>>>(dout, dlo, dhi, planes, H, W,
                                                                                               alpha_device, alpha);
    RGBD_CHECK_LAUNCH("fade_planes_bwd_kernel");
    return 0;
}

extern "C" int rgbd_lerp_bf16(const void* p, const void* q, void* out, void* out2, int64_t n, int mode,
                              const float* alpha_device, float alpha, void* stream) {
    RGBD_REQUIRE(p && out && n > 0 && n % 8 == 0 && ((mode == 0 && q) || (mode == 1 && out2)), "rgbd_lerp_bf16: bad arguments");
    lerp_bf16_kernel<<<grid_for(n / 8), 256, 0, (hipStream_t)stream>>>((const unsigned short*)p, (const unsigned short*)q,
                                                                      (unsigned short*)out, (unsigned short*)out2, n / 8, mode,
                                                                      alpha_device, alpha);
    RGBD_CHECK_LAUNCH("lerp_bf16_kernel");
    return 0;
}

extern "C" int rgbd_pool2_planes(const float* x, float* out, int64_t planes, int H, int W, int adjoint, void* stream) {
    RGBD_REQUIRE(x && out && planes > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0, "rgbd_pool2_planes: bad arguments");
    pool2_planes_kernel<<<grid_for(planes * (H / 2) * (W / 2)), 256, 0, (hipStream_t)stream>>>(x, out, planes, H, W, adjoint);
    RGBD_CHECK_LAUNCH("pool2_planes_kernel");
    return 0;
}
