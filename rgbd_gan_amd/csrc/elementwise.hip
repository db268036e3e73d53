// HBM-bound helper kernels: weight packing, AdaIN, clipped Adam.  All vectorised 16 B per lane where the
// layout allows; bounded by the ~6.3 TB/s achievable HBM rate (MI355X_MICROARCH.md), not by MFMA.
#include "common.h"

#include <mutex>

static thread_local char g_err[512] = "";

void rgbd_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

__global__ __launch_bounds__(256) void zero_kernel(unsigned int* __restrict__ p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = 0u;
}

hipError_t rgbd_zero_async(void* ptr, size_t bytes, hipStream_t stream) {
    const size_t n = bytes / 4;
    if (n == 0) return hipSuccess;
    const unsigned blocks = (unsigned)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    zero_kernel<<<blocks, 256, 0, stream>>>((unsigned int*)ptr, n);
    return hipGetLastError();
}

extern "C" const char* rgbd_last_error(void) { return g_err; }
extern "C" int rgbd_abi_version(void) { return RGBD_ABI_VERSION; }

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel, device) and again only when a LARGER size is asked for.
// Called from whichever thread launches (the host's, autograd's workers): the table is guarded by a mutex -- the one-time
// module state section 8(b) allows; it never influences what a launch computes.
bool rgbd_reserve_lds(const void* fn, int bytes) {
    struct Key { const void* fn; int dev; int bytes; };
    static Key done[512];
    static int ndone = 0;
    static std::mutex mu;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    std::lock_guard<std::mutex> lock(mu);
    Key* slot = nullptr;
    for (int i = 0; i < ndone; ++i)
        if (done[i].fn == fn && done[i].dev == dev) { slot = &done[i]; break; }
    if (slot && slot->bytes >= bytes) return true;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return false;
    if (slot) slot->bytes = bytes;
    else if (ndone < 512) done[ndone++] = Key{fn, dev, bytes};
    return true;
}

extern "C" int rgbd_zero_f32(float* p, int64_t n, void* stream) {
    RGBD_REQUIRE(p && n > 0, "rgbd_zero_f32: bad arguments");
    if (rgbd_zero_async(p, (size_t)n * sizeof(float), (hipStream_t)stream) != hipSuccess) {
        rgbd_set_error("rgbd_zero_f32: launch failed");
        return -2;
    }
    return 0;
}

namespace {
struct FinitePtrs { const float* p[8]; };
// bit i of *mask is set when *p[i] is NaN or +-Inf (mask is OR-ed: sticky until the host clears it)
__global__ void nonfinite_mask_kernel(FinitePtrs a, int n, int* __restrict__ mask) {
    const int i = threadIdx.x;
    if (i < n && a.p[i]) {
        const unsigned bits = __float_as_uint(*a.p[i]);
        if ((bits & 0x7f800000u) == 0x7f800000u) atomicOr(mask, 1 << i);
    }
}
}  // namespace

// The reference asserts `not xp.isnan(loss.data)` three times per step (updater.py:336,360,439), each a host synchronisation.
// Here the losses stay on the device: this launch folds their finiteness into one sticky int32 that the host reads a step
// later through a pinned copy (RGBDUpdater._nan_watch), so a diverging run stops within two steps and no step waits.
extern "C" int rgbd_nonfinite_mask_f32(const float* const* scalars_host, int n, int32_t* mask, void* stream) {
    RGBD_REQUIRE(scalars_host && mask && n > 0 && n <= 8, "rgbd_nonfinite_mask_f32: up to 8 device scalars");
    FinitePtrs a;
    for (int i = 0; i < 8; ++i) a.p[i] = i < n ? scalars_host[i] : nullptr;
    nonfinite_mask_kernel<<<1, 64, 0, (hipStream_t)stream>>>(a, n, mask);
    RGBD_CHECK_LAUNCH("nonfinite_mask_kernel");
    return 0;
}

namespace {

// ------------------------------------------------------------------------------------------------ weights
__global__ __launch_bounds__(256) void pack_weights_kernel(const float* __restrict__ w, int cout, int cin, int kh,
                                                           int kw, float scale, unsigned short* __restrict__ wf,
                                                           unsigned short* __restrict__ wd) {
    const long total = (long)cout * cin * kh * kw;
    const int taps = kh * kw;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        // e indexes the packed forward image [tap][co][ci] so stores are coalesced
        const int ci = (int)(e % cin);
        const long r = e / cin;
        const int co = (int)(r % cout);
        const int tap = (int)(r / cout);
        const float v = w[((long)co * cin + ci) * taps + tap] * scale;
        const unsigned short h = f32_to_bf16_bits(v);
        if (wf) wf[e] = h;
        if (wd) wd[((long)(taps - 1 - tap) * cin + ci) * cout + co] = h;
    }
}

// ------------------------------------------------------------------------------------------------ AdaIN
// x is (B, HW, C) bf16.  A block owns one batch item, one 64-channel group and a strip of pixels:
// thread t handles the 8-channel chunk (t & 7) of pixels (t >> 3), (t >> 3) + 32, ...
// The strip sums leave as PLAIN stores into sums[strip][b][c][2] and the apply kernels add the strips in index order:
// no atomics, so the statistics -- and with them every activation downstream -- are bit-reproducible run to run
// (fp32 atomics in arrival order flipped bf16 roundings: ~1e-3 relative run-to-run noise on the generator output).
constexpr int ADAIN_STRIP = 1024;

// sum over strips of the (first, second) statistics of this thread's 8 channels, (sample, channel) pair index sidx.
// nstrips > 0: float pairs, sums[strip][B][C][2] (adain_reduce_kernel).  nstrips < 0: the statistics came out of the
// producing convolution's epilogue (rgbd_conv2d_fprop_stats_bf16): ONE pair of 64-bit integers per (sample, channel) in
// units of 2^nstrips, `sums` pointing at int64 [B][C][2].
__device__ __forceinline__ void adain_strip_sum(const float* __restrict__ sums, long sidx, long strip_stride, int nstrips,
                                                float (&s1)[8], float (&s2)[8]) {
#pragma unroll
    for (int k = 0; k < 8; ++k) { s1[k] = 0.f; s2[k] = 0.f; }
    if (nstrips < 0) {
        const double unit = __builtin_ldexp(1.0, nstrips);
        const long long* q8 = reinterpret_cast<const long long*>(sums) + 2 * sidx;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const long long m2 = q8[2 * k + 1];
            const bool poisoned = m2 < 0 || m2 >= (1LL << 62);     // see stats_flush (conv.hip): a non-finite or overflowing sum
            s1[k] = poisoned ? __builtin_nanf("") : (float)((double)q8[2 * k] * unit);
            s2[k] = poisoned ? __builtin_nanf("") : (float)((double)m2 * unit);
        }
        return;
    }
    const float* part = sums + 2 * sidx;
    // four strips' loads in flight at a time (a 128x128 image has 16 strips: one dependent L2 round trip per strip
    // was several microseconds of prologue in every block); the adds stay in strip order
    for (int s = 0; s < nstrips; s += 4) {
        f32x4 q[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float* p = part + (long)(s + u < nstrips ? s + u : s) * strip_stride;
#pragma unroll
            for (int v = 0; v < 4; ++v) q[u][v] = *reinterpret_cast<const f32x4*>(p + 4 * v);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (s + u >= nstrips) break;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                s1[2 * v] += q[u][v][0]; s2[2 * v] += q[u][v][1];
                s1[2 * v + 1] += q[u][v][2]; s2[2 * v + 1] += q[u][v][3];
            }
        }
    }
}

template <bool WITH_DY>
__global__ __launch_bounds__(256) void adain_reduce_kernel(const unsigned short* __restrict__ x,
                                                           const unsigned short* __restrict__ dy,
                                                           const float* __restrict__ mean,
                                                           const float* __restrict__ rstd, float* __restrict__ sums,
                                                           int HW, int C) {
    const int b = blockIdx.z, cg = blockIdx.y, strip = blockIdx.x;
    const int chunk = threadIdx.x & 7, lane_p = threadIdx.x >> 3;
    const int c0 = cg * 64 + chunk * 8;
    const int p_begin = strip * ADAIN_STRIP;
    const int p_end = min(HW, p_begin + ADAIN_STRIP);
    float s0[8], s1[8], mu[8], rs[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        s0[k] = 0.f; s1[k] = 0.f;
        if (WITH_DY) { mu[k] = mean[(long)b * C + c0 + k]; rs[k] = rstd[(long)b * C + c0 + k]; }
    }
    // NB passes' loads issued before any is used: 8 independent 16-byte loads in flight per lane either way
    constexpr int NB = WITH_DY ? 4 : 8;
    for (int p0 = p_begin + lane_p; p0 < p_end; p0 += 32 * NB) {
        u32x4 xq[NB], gq[WITH_DY ? NB : 1];
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int p = p0 + 32 * u < p_end ? p0 + 32 * u : p0;
            const long off = ((long)b * HW + p) * C + c0;
            xq[u] = *reinterpret_cast<const u32x4*>(x + off);
            if constexpr (WITH_DY) gq[u] = *reinterpret_cast<const u32x4*>(dy + off);
        }
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            if (p0 + 32 * u >= p_end) break;
            const u32x4 xv = xq[u];
            u32x4 gv = {0u, 0u, 0u, 0u};
            if constexpr (WITH_DY) gv = gq[u];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float xl = bf16_lo(xv[k]), xh = bf16_hi(xv[k]);
                if (WITH_DY) {
                    const float gl = bf16_lo(gv[k]), gh = bf16_hi(gv[k]);
                    s0[2 * k] += gl; s0[2 * k + 1] += gh;
                    s1[2 * k] += gl * ((xl - mu[2 * k]) * rs[2 * k]);
                    s1[2 * k + 1] += gh * ((xh - mu[2 * k + 1]) * rs[2 * k + 1]);
                } else {
                    s0[2 * k] += xl; s0[2 * k + 1] += xh;
                    s1[2 * k] += xl * xl; s1[2 * k + 1] += xh * xh;
                }
            }
        }
    }
    __shared__ float red[2][32][65];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        red[0][lane_p][chunk * 8 + k] = s0[k];
        red[1][lane_p][chunk * 8 + k] = s1[k];
    }
    __syncthreads();
    if (threadIdx.x < 128) {
        const int which = threadIdx.x >> 6, c = threadIdx.x & 63;
        float acc = 0.f;
#pragma unroll 8
        for (int r = 0; r < 32; ++r) acc += red[which][r][c];
        sums[(((long)strip * gridDim.z + b) * C + cg * 64 + c) * 2 + which] = acc;
    }
}

// y = (x - mean) * rstd * scale + shift, mean / rstd derived from the (sum x, sum x^2) strip partials (adain.py:62-63:
// biased variance, (var + eps)^-1/2).  A block owns a strip of rows of ONE sample and one 64-channel group, so the
// per-(b,c) constants are formed once per thread; the block of the first strip also stores mean / rstd for backward.
__global__ __launch_bounds__(256) void adain_apply_kernel(const unsigned short* __restrict__ x,
                                                          const float* __restrict__ scale,
                                                          const float* __restrict__ shift,
                                                          const float* __restrict__ sums,
                                                          float* __restrict__ mean, float* __restrict__ rstd,
                                                          unsigned short* __restrict__ y, int HW, int C,
                                                          int ld, float inv_hw, float eps, int rows_per_block,
                                                          int nstrips, unsigned char* __restrict__ yq,
                                                          unsigned char* __restrict__ ys, int c_live) {
    const int cg = blockIdx.y, b = blockIdx.z;
    const int chunk = threadIdx.x & 7, lane_p = threadIdx.x >> 3;
    const int c0 = cg * 64 + chunk * 8;
    const long sidx = (long)b * C + c0;
    const long aidx = (long)b * ld + c0;            // scale / shift rows are ld floats apart
    // the first batch of x is requested BEFORE the statistics (a chain of dependent L2 round trips) are formed, and
    // every later batch while the previous one is computed
    const int r_begin = blockIdx.x * rows_per_block;
    const int r_end = min(HW, r_begin + rows_per_block);
    const long base = (long)b * HW;
    u32x4 xn[4];
    auto request = [&](int r0) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int r = (r0 + 32 * u) < r_end ? (r0 + 32 * u) : r0;
            xn[u] = *reinterpret_cast<const u32x4*>(x + (base + r) * C + c0);
        }
    };
    if (r_begin + lane_p < r_end) request(r_begin + lane_p);
    float s1[8], s2[8];
    adain_strip_sum(sums, sidx, (long)gridDim.z * C * 2, nstrips, s1, s2);
    // channels >= c_live are zero padding (a 32-channel block of the DeepVoxels generator on the engine's 64-channel
    // granularity): no scale / shift exists for them -- the [scale | shift] window is 2 c_live wide -- and they stay zero
    const bool live = c0 < c_live;
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    const f32x4 g0 = live ? *reinterpret_cast<const f32x4*>(scale + aidx) : z4, g1 = live ? *reinterpret_cast<const f32x4*>(scale + aidx + 4) : z4;
    const f32x4 h0 = live ? *reinterpret_cast<const f32x4*>(shift + aidx) : z4, h1 = live ? *reinterpret_cast<const f32x4*>(shift + aidx + 4) : z4;
    const float gg[8] = {g0[0], g0[1], g0[2], g0[3], g1[0], g1[1], g1[2], g1[3]};
    const float hh[8] = {h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
    float m[8], a[8];
    {
        float rs[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            m[k] = s1[k] * inv_hw;
            rs[k] = rsqrtf(fmaxf(s2[k] * inv_hw - m[k] * m[k], 0.f) + eps);
            a[k] = rs[k] * gg[k];
        }
        if (blockIdx.x == 0 && lane_p == 0) {       // the statistics for backward
            *reinterpret_cast<f32x4*>(mean + sidx) = f32x4{m[0], m[1], m[2], m[3]};
            *reinterpret_cast<f32x4*>(mean + sidx + 4) = f32x4{m[4], m[5], m[6], m[7]};
            *reinterpret_cast<f32x4*>(rstd + sidx) = f32x4{rs[0], rs[1], rs[2], rs[3]};
            *reinterpret_cast<f32x4*>(rstd + sidx + 4) = f32x4{rs[4], rs[5], rs[6], rs[7]};
        }
    }
    for (int r0 = r_begin + lane_p; r0 < r_end; r0 += 128) {
        u32x4 xv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) xv[u] = xn[u];
        if (r0 + 128 < r_end) request(r0 + 128);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int r = r0 + 32 * u;
            if (r < r_end) {
                u32x4 out;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float v0 = (bf16_lo(xv[u][k]) - m[2 * k]) * a[2 * k] + hh[2 * k];
                    const float v1 = (bf16_hi(xv[u][k]) - m[2 * k + 1]) * a[2 * k + 1] + hh[2 * k + 1];
                    out[k] = pack_bf16x2(v0, v1);
                }
                *reinterpret_cast<u32x4*>(y + (base + r) * C + c0) = out;
                if (yq) {               // conv_dtype mxfp8: the next convolution's operand leaves with the bf16 tensor
                    const long e = (base + r) * C + c0;
                    mx8_emit8(out, yq + e, ys + (e >> 5), (chunk & 3) == 0);
                }
            }
        }
    }
}

// dx = rstd * scale * (dy - sum(dy)/HW - xhat * sum(dy*xhat)/HW);  sums = (sum dy, sum dy*xhat) per (b,c)
// A block owns a strip of rows of ONE sample and one 64-channel group: the per-(b,c) constants (mean, rstd, scale, the
// two sums) are read once per thread, and 8 independent 16-byte loads per lane are in flight (lane layout and 4-pass
// batching of lrelu_bwd_colsum_kernel below).
// MASK: x is the OUTPUT of the leaky ReLU that feeds this AdaIN (net.py:150-153,157-160: conv -> bias -> lrelu -> style),
// so the activation gradient is applied in the same pass, dz = dx * (x > 0 ? 1 : slope), and the bias gradient
// bias_grad[c] += sum dz rides along: one pass over HBM instead of the AdaIN backward plus a separate
// activation-gradient pass.
template <bool MASK, int NT = 256>   // NT 1024 with the bias gradient on large images: fewer blocks, fewer same-address atomics
__global__ __launch_bounds__(NT) void adain_bwd_apply_kernel(const unsigned short* __restrict__ x,
                                                              const unsigned short* __restrict__ dy,
                                                              const float* __restrict__ scale,
                                                              const float* __restrict__ mean,
                                                              const float* __restrict__ rstd,
                                                              const float* __restrict__ sums,
                                                              unsigned short* __restrict__ dx,
                                                              float* __restrict__ dscale, float* __restrict__ dshift,
                                                              int HW, int C, float inv_hw, int ld, int rows_per_block,
                                                              float slope, float* __restrict__ bias_grad, int nstrips,
                                                              unsigned char* __restrict__ dxq,
                                                              unsigned char* __restrict__ dxs, int c_live) {
    const int cg = blockIdx.y, b = blockIdx.z;
    const int chunk = threadIdx.x & 7, lane_p = threadIdx.x >> 3;
    const int c0 = cg * 64 + chunk * 8;
    const long sidx = (long)b * C + c0;
    const long aidx = (long)b * ld + c0;
    // first batch of x / dy requested before the per-(b,c) constants, later ones while the previous batch is computed
    const int r_begin = blockIdx.x * rows_per_block;
    const int r_end = min(HW, r_begin + rows_per_block);
    const long base = (long)b * HW;
    u32x4 xn[4], gn[4];
    auto request = [&](int r0) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int r = (r0 + (NT / 8) * u) < r_end ? (r0 + (NT / 8) * u) : r0;
            const long off = (base + r) * C + c0;
            xn[u] = *reinterpret_cast<const u32x4*>(x + off);
            gn[u] = *reinterpret_cast<const u32x4*>(dy + off);
        }
    };
    if (r_begin + lane_p < r_end) request(r_begin + lane_p);
    float s1[8], s2[8];                                 // sum dy, sum dy * xhat
    adain_strip_sum(sums, sidx, (long)gridDim.z * C * 2, nstrips, s1, s2);
    const bool live = c0 < c_live;              // (padding channels: scale 0, so dx = 0, and no d scale / d shift entries)
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    const f32x4 g0 = live ? *reinterpret_cast<const f32x4*>(scale + aidx) : z4, g1 = live ? *reinterpret_cast<const f32x4*>(scale + aidx + 4) : z4;
    const f32x4 m0 = *reinterpret_cast<const f32x4*>(mean + sidx), m1 = *reinterpret_cast<const f32x4*>(mean + sidx + 4);
    const f32x4 r0v = *reinterpret_cast<const f32x4*>(rstd + sidx), r1v = *reinterpret_cast<const f32x4*>(rstd + sidx + 4);
    const float gg[8] = {g0[0], g0[1], g0[2], g0[3], g1[0], g1[1], g1[2], g1[3]};
    const float mm[8] = {m0[0], m0[1], m0[2], m0[3], m1[0], m1[1], m1[2], m1[3]};
    const float rr[8] = {r0v[0], r0v[1], r0v[2], r0v[3], r1v[0], r1v[1], r1v[2], r1v[3]};
    if (blockIdx.x == 0 && lane_p == 0 && live) {      // d shift = sum dy, d scale = sum dy * xhat (adain.py:76-77)
        *reinterpret_cast<f32x4*>(dshift + aidx) = f32x4{s1[0], s1[1], s1[2], s1[3]};
        *reinterpret_cast<f32x4*>(dshift + aidx + 4) = f32x4{s1[4], s1[5], s1[6], s1[7]};
        *reinterpret_cast<f32x4*>(dscale + aidx) = f32x4{s2[0], s2[1], s2[2], s2[3]};
        *reinterpret_cast<f32x4*>(dscale + aidx + 4) = f32x4{s2[4], s2[5], s2[6], s2[7]};
    }
    float bs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int r0 = r_begin + lane_p; r0 < r_end; r0 += 4 * (NT / 8)) {
        u32x4 xv[4], gv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { xv[u] = xn[u]; gv[u] = gn[u]; }
        if (r0 + 4 * (NT / 8) < r_end) request(r0 + 4 * (NT / 8));
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int r = r0 + (NT / 8) * u;
            if (r < r_end) {
                u32x4 out;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float v[2];
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int c = 2 * k + h;
                        const float xr = h ? bf16_hi(xv[u][k]) : bf16_lo(xv[u][k]);
                        const float xh = (xr - mm[c]) * rr[c];
                        const float g = h ? bf16_hi(gv[u][k]) : bf16_lo(gv[u][k]);
                        v[h] = rr[c] * gg[c] * (g - s1[c] * inv_hw - xh * s2[c] * inv_hw);
                        if (MASK) v[h] = xr > 0.f ? v[h] : v[h] * slope;
                    }
                    out[k] = pack_bf16x2(v[0], v[1]);
                    if (MASK) {
                        bs[2 * k] += bf16_lo(out[k]);
                        bs[2 * k + 1] += bf16_hi(out[k]);
                    }
                }
                *reinterpret_cast<u32x4*>(dx + (base + r) * C + c0) = out;
                if (dxq) {
                    const long e = (base + r) * C + c0;
                    mx8_emit8(out, dxq + e, dxs + (e >> 5), (chunk & 3) == 0);
                }
            }
        }
    }
    if (MASK && bias_grad) {
        __shared__ float red[NT / 8][65];
#pragma unroll
        for (int k = 0; k < 8; ++k) red[lane_p][chunk * 8 + k] = bs[k];
        __syncthreads();
        if (threadIdx.x < 64) {
            float acc = 0.f;
#pragma unroll 8
            for (int r = 0; r < NT / 8; ++r) acc += red[r][threadIdx.x];
            atomicAdd(bias_grad + cg * 64 + threadIdx.x, acc);
        }
    }
}

// ------------------------------------------------------------------------------------------------ leaky ReLU grad
// dz = dy * (y > 0 ? 1 : slope) on the first act_channels channels of an NHWC bf16 tensor (rest pass through).
// y is the *output* of the activation (sign-preserving), so no pre-activation tensor is kept.
__global__ __launch_bounds__(256) void lrelu_bwd_kernel(const unsigned short* __restrict__ dy,
                                                        const unsigned short* __restrict__ y,
                                                        unsigned short* __restrict__ dz, long nvec, int C,
                                                        int act_channels, float slope) {
    const int cvec = C >> 3;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < nvec; e += (long)gridDim.x * 256) {
        const int c0 = (int)(e % cvec) * 8;
        const u32x4 g = *reinterpret_cast<const u32x4*>(dy + e * 8);
        u32x4 out = g;
        if (c0 < act_channels) {
            const u32x4 yy = *reinterpret_cast<const u32x4*>(y + e * 8);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float g0 = bf16_lo(g[k]), g1 = bf16_hi(g[k]);
                const float r0 = bf16_lo(yy[k]) > 0.f ? g0 : g0 * slope;
                const float r1 = bf16_hi(yy[k]) > 0.f ? g1 : g1 * slope;
                out[k] = pack_bf16x2(r0, r1);
            }
        }
        *reinterpret_cast<u32x4*>(dz + e * 8) = out;
    }
}

// Same, fused with the bias gradient: bias_grad[c] += sum_m dz[m][c] (fp32 atomics into the caller's buffer).
// A block owns a strip of rows and one 64-channel group so the column sums reduce on chip; one pass over HBM.
template <int NT>   // threads: 256 (small tensors) or 1024 (large: a quarter of the blocks, a quarter of the atomics)
__global__ __launch_bounds__(NT) void lrelu_bwd_colsum_kernel(const unsigned short* __restrict__ dy,
                                                               const unsigned short* __restrict__ y,
                                                               unsigned short* __restrict__ dz, long M, int C,
                                                               int act_channels, float slope, int rows_per_block,
                                                               float* __restrict__ bias_grad,
                                                               const float* __restrict__ row_scale,
                                                               long rows_per_sample) {
    // bias_grad[c] += sum_r w(r) dz[r][c], w(r) = row_scale[r / rows_per_sample] (1 when row_scale is null).
    // 8 lanes cover one row's 64-channel group (128 B), NT/8 rows per pass, FOUR passes' loads issued before any is
    // used: 8 independent 16-byte loads in flight per lane keep HBM busy with a handful of waves per CU.
    const int cg = blockIdx.y;
    const int chunk = threadIdx.x & 7, lane_p = threadIdx.x >> 3;
    const int c0 = cg * 64 + chunk * 8;
    const bool act = c0 < act_channels;
    const long r_begin = (long)blockIdx.x * rows_per_block;
    const long r_end = min(M, r_begin + rows_per_block);
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (long r0 = r_begin + lane_p; r0 < r_end; r0 += 4 * (NT / 8)) {
        u32x4 g[4], yy[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long r = r0 + (NT / 8) * u;
            const long off = (r < r_end ? r : r0) * C + c0;
            g[u] = *reinterpret_cast<const u32x4*>(dy + off);
            yy[u] = act ? *reinterpret_cast<const u32x4*>(y + off) : u32x4{0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long r = r0 + (NT / 8) * u;
            if (r < r_end) {
                u32x4 out = g[u];
                if (act) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float g0 = bf16_lo(g[u][k]), g1 = bf16_hi(g[u][k]);
                        const float v0 = bf16_lo(yy[u][k]) > 0.f ? g0 : g0 * slope;
                        const float v1 = bf16_hi(yy[u][k]) > 0.f ? g1 : g1 * slope;
                        out[k] = pack_bf16x2(v0, v1);
                    }
                }
                *reinterpret_cast<u32x4*>(dz + r * C + c0) = out;
                const float wr = row_scale ? row_scale[r / rows_per_sample] : 1.f;
#pragma unroll
                for (int k = 0; k < 4; ++k) { s[2 * k] += wr * bf16_lo(out[k]); s[2 * k + 1] += wr * bf16_hi(out[k]); }
            }
        }
    }
    __shared__ float red[NT / 8][65];
#pragma unroll
    for (int k = 0; k < 8; ++k) red[lane_p][chunk * 8 + k] = s[k];
    __syncthreads();
    if (threadIdx.x < 64) {
        float acc = 0.f;
#pragma unroll 8
        for (int r = 0; r < NT / 8; ++r) acc += red[r][threadIdx.x];
        atomicAdd(bias_grad + cg * 64 + threadIdx.x, acc);
    }
}

template <int NT>
__global__ __launch_bounds__(NT) void unpool_lrelu_bwd_kernel(const unsigned short* __restrict__ dp,
                                                               const unsigned short* __restrict__ y,
                                                               unsigned short* __restrict__ dz, long M, int H, int W,
                                                               int C, float slope, int rows_per_block,
                                                               float* __restrict__ bias_grad,
                                                               float* __restrict__ bias_grad2,
                                                               const float* __restrict__ row_scale,
                                                               unsigned char* __restrict__ dzq,
                                                               unsigned char* __restrict__ dzs) {
    // same lane layout and 4-pass load batching as lrelu_bwd_colsum_kernel; dp is read at the pooled position
    const int cg = blockIdx.y;
    const int chunk = threadIdx.x & 7, lane_p = threadIdx.x >> 3;
    const int c0 = cg * 64 + chunk * 8;
    const long r_begin = (long)blockIdx.x * rows_per_block;
    const long r_end = min(M, r_begin + rows_per_block);
    const int Wp = W >> 1, Hp = H >> 1;
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (long r0 = r_begin + lane_p; r0 < r_end; r0 += 4 * (NT / 8)) {
        u32x4 g[4], yy[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long r = (r0 + (NT / 8) * u) < r_end ? (r0 + (NT / 8) * u) : r0;
            const int w = (int)(r % W);
            const long t = r / W;
            const int h = (int)(t % H);
            const long b = t / H;
            const long rp = (b * Hp + (h >> 1)) * Wp + (w >> 1);
            g[u] = *reinterpret_cast<const u32x4*>(dp + rp * C + c0);
            yy[u] = y ? *reinterpret_cast<const u32x4*>(y + r * C + c0) : u32x4{0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long r = r0 + (NT / 8) * u;
            if (r < r_end) {
                u32x4 out;
                const float wr = row_scale ? row_scale[r / ((long)H * W)] : 1.f;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float g0 = bf16_lo(g[u][k]) * 0.25f, g1 = bf16_hi(g[u][k]) * 0.25f;
                    if (y) {
                        g0 = bf16_lo(yy[u][k]) > 0.f ? g0 : g0 * slope;
                        g1 = bf16_hi(yy[u][k]) > 0.f ? g1 : g1 * slope;
                    }
                    out[k] = pack_bf16x2(g0, g1);
                    s[2 * k] += wr * bf16_lo(out[k]);
                    s[2 * k + 1] += wr * bf16_hi(out[k]);
                }
                *reinterpret_cast<u32x4*>(dz + r * C + c0) = out;
                if (dzq) {
                    const long e = r * C + c0;
                    mx8_emit8(out, dzq + e, dzs + (e >> 5), (chunk & 3) == 0);
                }
            }
        }
    }
    if (bias_grad) {
        __shared__ float red[NT / 8][65];
#pragma unroll
        for (int k = 0; k < 8; ++k) red[lane_p][chunk * 8 + k] = s[k];
        __syncthreads();
        if (threadIdx.x < 64) {
            float acc = 0.f;
#pragma unroll 8
            for (int r = 0; r < NT / 8; ++r) acc += red[r][threadIdx.x];
            atomicAdd(bias_grad + cg * 64 + threadIdx.x, acc);
            if (bias_grad2) atomicAdd(bias_grad2 + cg * 64 + threadIdx.x, acc);
        }
    }
}

// pool: out[b,hp,wp,c] = scale * sum_{2x2} x[b,h,w,c] * (y ? lrelu'(y[b,h,w,c]) : 1)     (scale 0.25: average, 1: sum)
__global__ __launch_bounds__(256) void pool2_masked_kernel(const unsigned short* __restrict__ x,
                                                           const unsigned short* __restrict__ y,
                                                           unsigned short* __restrict__ out, long nvec_out, int H, int W,
                                                           int C, float slope, float scale) {
    const int cvec = C >> 3;
    const int Wp = W >> 1, Hp = H >> 1;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < nvec_out; e += (long)gridDim.x * 256) {
        const int cv = (int)(e % cvec);
        const long rp = e / cvec;
        const int wp = (int)(rp % Wp);
        const long t = rp / Wp;
        const int hp = (int)(t % Hp);
        const long b = t / Hp;
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                const long r = (b * H + 2 * hp + dy) * W + 2 * wp + dx;
                const u32x4 v = *reinterpret_cast<const u32x4*>(x + r * C + cv * 8);
                u32x4 yy = {0u, 0u, 0u, 0u};
                if (y) yy = *reinterpret_cast<const u32x4*>(y + r * C + cv * 8);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float v0 = bf16_lo(v[k]), v1 = bf16_hi(v[k]);
                    if (y) {
                        v0 = bf16_lo(yy[k]) > 0.f ? v0 : v0 * slope;
                        v1 = bf16_hi(yy[k]) > 0.f ? v1 : v1 * slope;
                    }
                    acc[2 * k] += v0;
                    acc[2 * k + 1] += v1;
                }
            }
        u32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = pack_bf16x2(acc[2 * k] * scale, acc[2 * k + 1] * scale);
        *reinterpret_cast<u32x4*>(out + e * 8) = o;
    }
}

// column sums of an (M, C) bf16 matrix -> out[C] fp32 (atomics; out zeroed by the caller): bias gradients.
template <int NT>
__global__ __launch_bounds__(NT) void colsum_kernel(const unsigned short* __restrict__ x, float* __restrict__ out,
                                                     long M, int C, int rows_per_block,
                                                     const float* __restrict__ row_scale, long rows_per_sample) {
    // out[c] += sum_r w(r) * x[r][c], w(r) = row_scale[r / rows_per_sample] (1 when row_scale is null)
    const int cg = blockIdx.y;
    const int chunk = threadIdx.x & 7, lane_p = threadIdx.x >> 3;
    const int c0 = cg * 64 + chunk * 8;
    const long r_begin = (long)blockIdx.x * rows_per_block;
    const long r_end = min(M, r_begin + rows_per_block);
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (long r0 = r_begin + lane_p; r0 < r_end; r0 += 4 * (NT / 8)) {
        u32x4 v[4];
        float wgt[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long r = r0 + (NT / 8) * u;
            const long rc = r < r_end ? r : r0;
            v[u] = *reinterpret_cast<const u32x4*>(x + rc * C + c0);
            wgt[u] = r < r_end ? (row_scale ? row_scale[rc / rows_per_sample] : 1.f) : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                s[2 * k] += wgt[u] * bf16_lo(v[u][k]);
                s[2 * k + 1] += wgt[u] * bf16_hi(v[u][k]);
            }
    }
    __shared__ float red[NT / 8][65];
#pragma unroll
    for (int k = 0; k < 8; ++k) red[lane_p][chunk * 8 + k] = s[k];
    __syncthreads();
    if (threadIdx.x < 64) {
        float acc = 0.f;
#pragma unroll 8
        for (int r = 0; r < NT / 8; ++r) acc += red[r][threadIdx.x];
        atomicAdd(out + cg * 64 + threadIdx.x, acc);
    }
}

// out[r][c] = a[r][c] + s[r / rows_per_sample] * x[r][c]   (bf16 in / out, fp32 arithmetic)
__global__ __launch_bounds__(256) void axpy_rows_kernel(const unsigned short* __restrict__ a,
                                                        const unsigned short* __restrict__ x,
                                                        const float* __restrict__ s, unsigned short* __restrict__ out,
                                                        long nvec, long vec_per_sample) {
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < nvec; e += (long)gridDim.x * 256) {
        const float sc = s[e / vec_per_sample];
        const u32x4 av = *reinterpret_cast<const u32x4*>(a + e * 8);
        const u32x4 xv = *reinterpret_cast<const u32x4*>(x + e * 8);
        u32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            o[k] = pack_bf16x2(bf16_lo(av[k]) + sc * bf16_lo(xv[k]), bf16_hi(av[k]) + sc * bf16_hi(xv[k]));
        *reinterpret_cast<u32x4*>(out + e * 8) = o;
    }
}

// ------------------------------------------------------------------------------------------------ RGB <-> features
// from_rgb: y[b,h,w,co] = act( sum_c w[co][c] * x[b,c,h,w] + bias[co] ), x NCHW fp32 (KP planes), y NHWC bf16.
// One thread = one pixel x 8 output channels (16-byte store).  HBM-bound: 4*KP B in, 2*C B out per pixel.
template <int KP>
__global__ __launch_bounds__(256) void from_planes_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                          const float* __restrict__ bias,
                                                          unsigned short* __restrict__ y, int B, int HW, int C,
                                                          float wscale, int act, float slope) {
    // A thread keeps ONE octet of output channels for its whole walk (the grid stride is a multiple of C/8), so its
    // 8 x KP weights and 8 biases sit in registers; NU pixels' plane values are requested before any is used (four, or
    // two with four planes: 64 registers, so that a block still fits on a SIMD next to the weight-gradient kernel's two
    // 224-register waves -- at 78 the launch waited for that kernel to finish, 375 us instead of 27).
    constexpr int NU = KP == 4 ? 2 : 4;
    const int cvec = C >> 3;
    const long nvec = (long)B * HW * cvec;
    const long e0 = (long)blockIdx.x * 256 + threadIdx.x;
    const long stride = (long)gridDim.x * 256;
    const int cv = (int)(e0 % cvec);
    float wr[8][KP], br[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        br[j] = bias ? bias[cv * 8 + j] : 0.f;
#pragma unroll
        for (int k = 0; k < KP; ++k) wr[j][k] = w[(cv * 8 + j) * KP + k] * wscale;
    }
    for (long e = e0; e < nvec; e += NU * stride) {
        float xin[NU][KP];
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const long eu = e + u * stride;
            const long pix = (eu < nvec ? eu : e) / cvec;
            const int b = (int)(pix / HW);
            const int p = (int)(pix - (long)b * HW);
#pragma unroll
            for (int k = 0; k < KP; ++k) xin[u][k] = x[((long)b * KP + k) * HW + p];
        }
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const long eu = e + u * stride;
            if (eu < nvec) {
                float r[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float acc = br[j];
#pragma unroll
                    for (int k = 0; k < KP; ++k) acc += wr[j][k] * xin[u][k];
                    r[j] = act ? (acc > 0.f ? acc : acc * slope) : acc;
                }
                u32x4 out = {pack_bf16x2(r[0], r[1]), pack_bf16x2(r[2], r[3]), pack_bf16x2(r[4], r[5]),
                             pack_bf16x2(r[6], r[7])};
                *reinterpret_cast<u32x4*>(y + eu * 8) = out;
            }
        }
    }
}

// to_planes: out[b,k,h,w] = sum_c w[k][c] * h[b,h,w,c] * wscale + bias[k]; h NHWC bf16, out NCHW fp32 (KP planes).
// 8 lanes cooperate on one pixel (16 B of channels each, shuffle reduce); used for the generator's `outs` 1x1
// conv (KP = 4) and for the input gradient of from_rgb (KP = 3, weights transposed by the caller).
template <int KP>
__global__ __launch_bounds__(256) void to_planes_kernel(const unsigned short* __restrict__ h,
                                                        const float* __restrict__ w, const float* __restrict__ bias,
                                                        float* __restrict__ out, int B, int HW, int C, float wscale) {
    const int sub = threadIdx.x & 7;
    const long npix = (long)B * HW;
    if (C == 64) {
        // the common case (64 feature channels): a lane's 8 x KP weights are loop invariants -- in registers; two pixels'
        // loads in flight per lane
        float wr[KP][8];
#pragma unroll
        for (int k = 0; k < KP; ++k)
#pragma unroll
            for (int j = 0; j < 8; ++j) wr[k][j] = w[k * 64 + sub * 8 + j] * wscale;
        float bk[KP];
#pragma unroll
        for (int k = 0; k < KP; ++k) bk[k] = bias ? bias[k] : 0.f;
        const long step = (long)gridDim.x * 32;
        for (long pix0 = (long)blockIdx.x * 32 + (threadIdx.x >> 3); pix0 < npix; pix0 += 2 * step) {
            u32x4 v[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const long pix = pix0 + u * step < npix ? pix0 + u * step : pix0;
                v[u] = *reinterpret_cast<const u32x4*>(h + pix * 64 + sub * 8);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const long pix = pix0 + u * step;
                float f[8], acc[KP];
#pragma unroll
                for (int q = 0; q < 4; ++q) { f[2 * q] = bf16_lo(v[u][q]); f[2 * q + 1] = bf16_hi(v[u][q]); }
#pragma unroll
                for (int k = 0; k < KP; ++k) {
                    acc[k] = 0.f;
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[k] += wr[k][j] * f[j];
                    acc[k] += __shfl_xor(acc[k], 1, 64);
                    acc[k] += __shfl_xor(acc[k], 2, 64);
                    acc[k] += __shfl_xor(acc[k], 4, 64);
                }
                if (sub == 0 && pix < npix) {
                    const int b = (int)(pix / HW);
                    const int p = (int)(pix - (long)b * HW);
#pragma unroll
                    for (int k = 0; k < KP; ++k) out[((long)b * KP + k) * HW + p] = acc[k] + bk[k];
                }
            }
        }
        return;
    }
    for (long pix = (long)blockIdx.x * 32 + (threadIdx.x >> 3); pix < npix; pix += (long)gridDim.x * 32) {
        float acc[KP];
#pragma unroll
        for (int k = 0; k < KP; ++k) acc[k] = 0.f;
        for (int c0 = sub * 8; c0 < C; c0 += 64) {
            const u32x4 v = *reinterpret_cast<const u32x4*>(h + pix * C + c0);
            float f[8];
#pragma unroll
            for (int q = 0; q < 4; ++q) { f[2 * q] = bf16_lo(v[q]); f[2 * q + 1] = bf16_hi(v[q]); }
#pragma unroll
            for (int k = 0; k < KP; ++k)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[k] += w[k * C + c0 + j] * f[j];
        }
#pragma unroll
        for (int k = 0; k < KP; ++k) {
            acc[k] += __shfl_xor(acc[k], 1, 64);
            acc[k] += __shfl_xor(acc[k], 2, 64);
            acc[k] += __shfl_xor(acc[k], 4, 64);
        }
        if (sub == 0) {
            const int b = (int)(pix / HW);
            const int p = (int)(pix - (long)b * HW);
#pragma unroll
            for (int k = 0; k < KP; ++k) out[((long)b * KP + k) * HW + p] = acc[k] * wscale + (bias ? bias[k] : 0.f);
        }
    }
}

// planes_outer: o[k][c] += sum_pix p[b,k,pix] * t[b,pix,c] (fp32 atomics; o zeroed by the caller), plus
// tsum[c] += sum_pix t[b,pix,c] when tsum != NULL and psum[k] += sum_pix p[b,k,pix] when psum != NULL (NOT zeroed: it is
// the bias gradient buffer of to_rgb).  Weight (and bias) gradients of from_rgb / to_rgb.
template <int KP>
__global__ __launch_bounds__(256) void planes_outer_kernel(const unsigned short* __restrict__ t,
                                                           const float* __restrict__ p, float* __restrict__ o,
                                                           float* __restrict__ tsum, float* __restrict__ psum, int B,
                                                           int HW, int C, int rows_per_block) {
    const int cg = blockIdx.y, b = blockIdx.z;
    const int chunk = threadIdx.x & 7, lane_p = threadIdx.x >> 3;
    const int c0 = cg * 64 + chunk * 8;
    const int r_begin = blockIdx.x * rows_per_block;
    const int r_end = min(HW, r_begin + rows_per_block);
    float acc[KP + 1][8];
    float ps[KP];
#pragma unroll
    for (int k = 0; k < KP; ++k) ps[k] = 0.f;
#pragma unroll
    for (int k = 0; k <= KP; ++k)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[k][j] = 0.f;
    for (int r0 = r_begin + lane_p; r0 < r_end; r0 += 128) {            // four rows' loads in flight per lane
        u32x4 v[4];
        float pv[4][KP];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int r = r0 + 32 * u < r_end ? r0 + 32 * u : r0;
            v[u] = *reinterpret_cast<const u32x4*>(t + ((long)b * HW + r) * C + c0);
#pragma unroll
            for (int k = 0; k < KP; ++k) pv[u][k] = r0 + 32 * u < r_end ? p[((long)b * KP + k) * HW + r] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (r0 + 32 * u >= r_end) break;
            float f[8];
#pragma unroll
            for (int q = 0; q < 4; ++q) { f[2 * q] = bf16_lo(v[u][q]); f[2 * q + 1] = bf16_hi(v[u][q]); }
#pragma unroll
            for (int k = 0; k < KP; ++k)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[k][j] += pv[u][k] * f[j];
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[KP][j] += f[j];
#pragma unroll
            for (int k = 0; k < KP; ++k) ps[k] += pv[u][k];
        }
    }
    __shared__ float red[32][65];
    if (psum && cg == 0) {               // plane sums: the rows of this block once (chunk 0's lanes hold every row once)
        if (chunk == 0) {
#pragma unroll
            for (int k = 0; k < KP; ++k) red[lane_p][k] = ps[k];
        }
        __syncthreads();
        if (threadIdx.x < KP) {
            float a2 = 0.f;
            for (int r = 0; r < 32; ++r) a2 += red[r][threadIdx.x];
            atomicAdd(psum + threadIdx.x, a2);
        }
    }
    for (int k = 0; k <= KP; ++k) {
        if (k == KP && !tsum) break;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 8; ++j) red[lane_p][chunk * 8 + j] = acc[k][j];
        __syncthreads();
        if (threadIdx.x < 64) {
            float a2 = 0.f;
#pragma unroll 8
            for (int r = 0; r < 32; ++r) a2 += red[r][threadIdx.x];
            if (k < KP) atomicAdd(o + (long)k * C + cg * 64 + threadIdx.x, a2);
            else atomicAdd(tsum + cg * 64 + threadIdx.x, a2);
        }
    }
}

// ------------------------------------------------------------------------------------------------ small linears
// Equalized-LR linear layers with a small batch (M <= 64 rows): mapping MLP (net.py:58-62), pose-conditioned style
// (net.py:220-224), StyleBlock affines (net.py:96-101).  y = act(c * x W^T + b), fp32 like the reference.  These are
// launch-latency bound (2 MFLOP each): one launch per layer forward, two backward, instead of ~10 library / elementwise
// launches.  x (M,K), W (N,K), y (M,N) row-major.
constexpr int LIN_MAXM = 64;
typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));   // a quad of floats on a 4-byte boundary

// block = 256 threads -> 8 output columns for all rows: thread = (row m = tid & 63, column pair tid >> 6).
// Few, long phases (K in chunks of 128): the kernel's time is load latency, so it uses N/8 blocks instead of N/32.
// Forward / input-gradient of the small linears on the fp32 matrix cores (v_mfma_f32_16x16x4_f32): a wave owns a
// 16 x 16 output tile, its lane (r = lane & 15, q = lane >> 4) streams four consecutive K values of row r of each
// operand per step straight from global memory (no LDS, no barriers; the operands are L2-resident and the whole
// problem is a few MFLOP, so the kernels are one dependent-load latency long).  The four waves of a block take the
// four 16-row slices of the <= 64 rows and share the other operand's tile through L1.
template <bool VEC>
__global__ __launch_bounds__(256) void linear_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ bias, float* __restrict__ y, int M,
                                                         int K, int N, float c, int act, float slope,
                                                         const float* __restrict__ mask_y, float* __restrict__ partial) {
    // blockIdx.y > 0 / gridDim.y > 1: split-K (the discriminator's dense tail has K = 4096 and N = 256: 16 column blocks
    // each walking 32 dependent-load chunks took ~100 us); every K slice leaves raw fp32 partial sums in
    // partial[slice][M][N] and linear_splitk_finish_kernel applies scale, bias and activation
    // Waves a small M leaves without a row tile take a share of K instead (M <= 16: four K parts, M <= 32: two): these
    // launches are chains of dependent load round trips, and the parts meet through LDS at the end.
    __shared__ f32x4 kred[3][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int mtiles = (M + 15) >> 4;
    const int kparts = gridDim.y == 1 ? (mtiles == 1 ? 4 : mtiles == 2 ? 2 : 1) : 1;
    const int mt = kparts == 4 ? 0 : kparts == 2 ? (wave & 1) : wave, kp = kparts == 4 ? wave : kparts == 2 ? (wave >> 1) : 0;
    const int m0 = mt * 16;
    if (kparts == 1 && m0 >= M) return;
    const int r = lane & 15, q = lane >> 4;
    const int n0 = blockIdx.x * 16;
    const int kslice = (K + (int)gridDim.y - 1) / (int)gridDim.y;
    int kbeg = (kslice * (int)blockIdx.y + 127) / 128 * 128 < K ? ((kslice * (int)blockIdx.y + 127) / 128 * 128) : K;
    int kend = gridDim.y == 1 ? K : ((kslice * ((int)blockIdx.y + 1) + 127) / 128 * 128 < K
                                     ? (kslice * ((int)blockIdx.y + 1) + 127) / 128 * 128 : K);
    if (kparts > 1) {
        const int chunks = (K + 127) >> 7;
        kbeg = min(K, (chunks * kp / kparts) << 7);
        kend = min(K, (chunks * (kp + 1) / kparts) << 7);
    }
    const bool mok = m0 + r < M, nok = n0 + r < N;
    const float* xr = x + (long)(mok ? m0 + r : 0) * K;
    const float* wr = w + (long)(nok ? n0 + r : 0) * K;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    f32x4 acc = zero;
    // NS K-steps of 16 per chunk: 16 independent loads in flight; the ragged-K variant takes half chunks (8 loads) to
    // stay under 64 registers: at 92 its blocks could not start beside the other stream's weight-gradient kernel
    constexpr int NS = VEC ? 8 : 4;
    for (int k0 = kbeg; k0 < kend; k0 += 16 * NS) {
        f32x4 a[NS], b[NS];
#pragma unroll
        for (int s2 = 0; s2 < NS; ++s2) {
            const int kb = k0 + 16 * s2 + 4 * q;
            if (VEC) {
                const int kc = kb < kend ? kb : 0;
                a[s2] = *reinterpret_cast<const f32x4*>(xr + kc);
                b[s2] = *reinterpret_cast<const f32x4*>(wr + kc);
                if (kb >= kend || !mok) a[s2] = zero;
                if (kb >= kend || !nok) b[s2] = zero;
            } else {
                // K not a multiple of 4 (the pose style: K = 265): rows start on 4-byte boundaries only.  Whole quads are
                // still fetched with ONE 16-byte load each (global_load_dwordx4 needs dword alignment only), the
                // ragged last quad element by element: 64 scalar loads per chunk per lane made this launch 190 us next
                // to the other stream's convolutions.
                if (kb + 3 < kend) {
                    a[s2] = *reinterpret_cast<const f32x4_u*>(xr + kb);
                    b[s2] = *reinterpret_cast<const f32x4_u*>(wr + kb);
                    if (!mok) a[s2] = zero;
                    if (!nok) b[s2] = zero;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const bool ok = kb + j < kend;
                        const float av = xr[ok ? kb + j : 0], bv = wr[ok ? kb + j : 0];
                        a[s2][j] = ok && mok ? av : 0.f;
                        b[s2][j] = ok && nok ? bv : 0.f;
                    }
                }
            }
        }
#pragma unroll
        for (int s2 = 0; s2 < NS; ++s2)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s2][j], b[s2][j], acc, 0, 0, 0);
    }
    if (kparts > 1) {                                  // block-uniform
        // kparts 4: waves 1-3 -> slots 0-2, wave 0 adds all three; kparts 2: waves 2,3 -> slots 0,1, waves 0,1 add theirs
        if (kp > 0) kred[kparts == 4 ? wave - 1 : wave - 2][lane] = acc;
        __syncthreads();
        if (kp > 0) return;
        if (kparts == 4) acc += (kred[0][lane] + kred[1][lane]) + kred[2][lane];
        else acc += kred[mt][lane];
    }
    const int n = n0 + r;
    if (n < N) {
        if (gridDim.y > 1) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int m = m0 + 4 * q + t;
                if (m < M) partial[((long)blockIdx.y * M + m) * N + n] = acc[t];
            }
            return;
        }
        const float bv = bias ? bias[n] : 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int m = m0 + 4 * q + t;
            if (m < M) {
                float v = acc[t] * c + bv;
                if (act) v = v > 0.f ? v : v * slope;
                if (mask_y) v = mask_y[(long)m * N + n] > 0.f ? v : v * slope;   // times lrelu'(.) of a GIVEN activation output
                y[(long)m * N + n] = v;
            }
        }
    }
}

__global__ __launch_bounds__(256) void linear_splitk_finish_kernel(const float* __restrict__ partial, int S, int M, int N,
                                                                   const float* __restrict__ bias, float c, int act,
                                                                   float slope, const float* __restrict__ mask_y,
                                                                   float* __restrict__ y) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= M * N) return;
    float v = 0.f;
    for (int s2 = 0; s2 < S; ++s2) v += partial[(long)s2 * M * N + e];
    v = v * c + (bias ? bias[e % N] : 0.f);
    if (act) v = v > 0.f ? v : v * slope;
    if (mask_y) v = mask_y[e] > 0.f ? v : v * slope;
    y[e] = v;
}

template <bool VEC>
__global__ __launch_bounds__(256) void linear_dgrad_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                           const float* __restrict__ w, float* __restrict__ dx, int M,
                                                           int K, int N, float c, int act, float slope, int accumulate) {
    // as linear_fwd_kernel: waves without a row tile take a share of the reduction (here over N)
    __shared__ f32x4 nred[3][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int mtiles = (M + 15) >> 4;
    const int nparts = mtiles == 1 ? 4 : mtiles == 2 ? 2 : 1;
    const int mt = nparts == 4 ? 0 : nparts == 2 ? (wave & 1) : wave, np = nparts == 4 ? wave : nparts == 2 ? (wave >> 1) : 0;
    const int m0 = mt * 16;
    if (nparts == 1 && m0 >= M) return;
    const int r = lane & 15, q = lane >> 4;
    const int k0 = blockIdx.x * 16;
    const bool mok = m0 + r < M, kok = k0 + r < K;
    const long zrow = (long)(mok ? m0 + r : 0) * N;
    const float* wc = w + (kok ? k0 + r : 0);
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    f32x4 acc = zero;
    const int nchunks = (N + 63) >> 6;
    const int nbeg = (nchunks * np / nparts) << 6, nend = min(N, (nchunks * (np + 1) / nparts) << 6);
    for (int n0 = nbeg; n0 < nend; n0 += 64) {     // 4 N-steps of 16 per chunk
        f32x4 a[4], yy[4], b[4];
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) {
            const int nb = n0 + 16 * s2 + 4 * q;
            if (VEC) {
                const int nc = nb < N ? nb : 0;
                a[s2] = *reinterpret_cast<const f32x4*>(dy + zrow + nc);
                yy[s2] = act ? *reinterpret_cast<const f32x4*>(y + zrow + nc) : f32x4{1.f, 1.f, 1.f, 1.f};
                if (nb >= N || !mok) a[s2] = zero;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const bool ok = nb + j < N;
                    const float av = dy[zrow + (ok ? nb + j : 0)];
                    yy[s2][j] = act ? y[zrow + (ok ? nb + j : 0)] : 1.f;
                    a[s2][j] = ok && mok ? av : 0.f;
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool ok = nb + j < N;
                const float bv = wc[(long)(ok ? nb + j : 0) * K];
                b[s2][j] = ok && kok ? bv : 0.f;
            }
        }
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float g = yy[s2][j] > 0.f ? a[s2][j] : a[s2][j] * slope;
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(g, b[s2][j], acc, 0, 0, 0);
            }
    }
    if (nparts > 1) {                                  // block-uniform
        if (np > 0) nred[nparts == 4 ? wave - 1 : wave - 2][lane] = acc;
        __syncthreads();
        if (np > 0) return;
        if (nparts == 4) acc += (nred[0][lane] + nred[1][lane]) + nred[2][lane];
        else acc += nred[mt][lane];
    }
    const int k = k0 + r;
    if (k < K) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int m = m0 + 4 * q + t;
            if (m < M) {
                const long o = (long)m * K + k;
                const float v = acc[t] * c;
                dx[o] = accumulate ? dx[o] + v : v;
            }
        }
    }
}

__global__ __launch_bounds__(256) void linear_wgrad_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                           const float* __restrict__ x, float* __restrict__ dw,
                                                           float* __restrict__ db, int M, int K, int N, float c, int act,
                                                           float slope) {
    __shared__ float zs[LIN_MAXM][17];
    __shared__ float xs[LIN_MAXM][65];
    const int n0 = blockIdx.y * 16, k0 = blockIdx.x * 64;
    for (int e = threadIdx.x; e < M * 16; e += 256) {
        const int m = e >> 4, nn = e & 15;
        float g = 0.f;
        if (n0 + nn < N) {
            g = dy[(long)m * N + n0 + nn];
            if (act && !(y[(long)m * N + n0 + nn] > 0.f)) g *= slope;
        }
        zs[m][nn] = g;
    }
    for (int e = threadIdx.x; e < M * 64; e += 256) {
        const int m = e >> 6, kk = e & 63;
        xs[m][kk] = (k0 + kk < K) ? x[(long)m * K + k0 + kk] : 0.f;
    }
    __syncthreads();
    const int nl = threadIdx.x >> 4, kq = (threadIdx.x & 15) * 4;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;
    for (int m = 0; m < M; ++m) {
        const float g = zs[m][nl];
        bsum += g;
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] += g * xs[m][kq + j];
    }
    if (n0 + nl < N) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (k0 + kq + j < K) dw[(long)(n0 + nl) * K + k0 + kq + j] += acc[j] * c;
        if (db && blockIdx.x == 0 && kq == 0) db[n0 + nl] += bsum;
    }
}

// ------------------------------------------------------------------------------------------------ Adam
constexpr int NORM_BLOCKS = 1024;

__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, long n, float gscale,
                                                    float* __restrict__ partial) {
    float acc = 0.f;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
        const float v = g[e] * gscale;
        acc += v * v;
    }
    acc = wave_sum(acc);
    __shared__ float red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// ws[NORM_BLOCKS] = norm, ws[NORM_BLOCKS + 1] = rate, ws[NORM_BLOCKS + 2] = sqrt(1-b2^t)/(1-b1^t); *step += 1
__global__ __launch_bounds__(256) void norm_final_kernel(float* __restrict__ ws, int nblocks, float clip,
                                                         float beta1, float beta2, int* __restrict__ step,
                                                         float* __restrict__ norm_out) {
    double acc = 0.0;
    for (int k = threadIdx.x; k < nblocks; k += 256) acc += (double)ws[k];
    __shared__ double red[256];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float norm = (float)sqrt(red[0]);
        const float rate = clip / norm;   // chainer GradientClipping: scale only when rate < 1
        ws[NORM_BLOCKS] = norm;
        ws[NORM_BLOCKS + 1] = rate < 1.f ? rate : 1.f;
        const int t = step[0] + 1;        // chainer Adam: t counts updates, incremented before use
        step[0] = t;
        const double fix1 = 1.0 - pow((double)beta1, (double)t);
        const double fix2 = 1.0 - pow((double)beta2, (double)t);
        ws[NORM_BLOCKS + 2] = (float)(sqrt(fix2) / fix1);
        if (norm_out) norm_out[0] = norm;
    }
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, long begin,
                                                   long end, float alpha, float beta1, float beta2, float eps,
                                                   float gscale, const float* __restrict__ ws) {
    const float rate = ws[NORM_BLOCKS + 1] * gscale;
    const float alpha_t = alpha * ws[NORM_BLOCKS + 2];
    for (long e = begin + (long)blockIdx.x * 256 + threadIdx.x; e < end; e += (long)gridDim.x * 256) {
        const float grad = g[e] * rate;
        float mm = m[e], vv = v[e];
        mm += (1.f - beta1) * (grad - mm);
        vv += (1.f - beta2) * (grad * grad - vv);
        m[e] = mm;
        v[e] = vv;
        p[e] -= alpha_t * mm / (sqrtf(vv) + eps);
    }
}

}  // namespace

extern "C" int rgbd_pack_weights(const float* w, int cout, int cin, int kh, int kw, float scale, void* w_fprop,
                                 void* w_dgrad, void* stream) {
    RGBD_REQUIRE(w && (w_fprop || w_dgrad), "rgbd_pack_weights: null pointer");
    RGBD_REQUIRE(cout > 0 && cin > 0 && kh > 0 && kw > 0, "rgbd_pack_weights: bad shape");
    const long total = (long)cout * cin * kh * kw;
    const int blocks = (int)min((long)2048, (total + 255) / 256);
    pack_weights_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(w, cout, cin, kh, kw, scale,
                                                                 (unsigned short*)w_fprop, (unsigned short*)w_dgrad);
    RGBD_CHECK_LAUNCH("pack_weights_kernel");
    return 0;
}

namespace {
constexpr int PACK_TCO = 8, PACK_TCI = 64;          // tile of the tiled path: 8 x 64 (co, ci) x up to 9 taps = 18 KB of LDS
// Element (co, ci, tap) of the (cout, cin, taps) weight a descriptor describes -> its index in D.w, or -1 for a padding zero.
// fold == 0: D.w IS that tensor.  Otherwise D.w is a reference-shaped MASTER parameter and (cout, cin, taps) its rearrangement
// for the 2-D conv engine (rgbd_fold_weight_f32's modes; the DeepVoxels generator): the fold happens in this read, the folded fp32
// copy (a launch and 8 bytes per element of traffic on every rebuild of the images) is never made.
__device__ __forceinline__ long pack_src_index(const rgbd_pack_desc& D, int co, int ci, int tap) {
    if (D.fold == 0) return ((long)co * D.cin + ci) * D.taps + tap;
    const int mode = (D.fold & 3) - 1, Co = (D.fold >> 2) & 0x7fff, Ci = (D.fold >> 17) & 0x7fff;
    if (mode == 0) {                                 // (Co,Ci,3,3,3) -> (cout, 3 * Cip, 3 x 3): channel kd * Cip + c carries depth tap kd
        const int Cip = D.cin / 3, kd = ci / Cip, c = ci - kd * Cip;
        return (co < Co && c < Ci) ? (((long)co * Ci + c) * 3 + kd) * 9 + tap : -1;
    }
    if (mode == 1) {                                 // (Co,Ci,4,4) -> (cout, 16 * Cip, 1 x 1): channel (ky * 4 + kx) * Cip + c
        const int Cip = D.cin / 16, t = ci / Cip, c = ci - t * Cip;
        return (co < Co && c < Ci) ? ((long)co * Ci + c) * 16 + t : -1;
    }
    return (co < Co && ci < Ci) ? ((long)co * Ci + ci) * D.taps + tap : -1;       // zero padding of both channel counts only
}
__global__ __launch_bounds__(256) void pack_weights_multi_kernel(const rgbd_pack_desc* __restrict__ descs, int n) {
    extern __shared__ float ptile[];                // [PACK_TCO][PACK_TCI * taps], as the master stores it
    int d = 0;
    for (int i = 1; i < n; ++i)
        if (descs[i].block_begin <= (int)blockIdx.x) d = i;
    const rgbd_pack_desc D = descs[d];
    const int nblk = (d + 1 < n ? descs[d + 1].block_begin : (int)gridDim.x) - D.block_begin;
    const long total = (long)D.cout * D.cin * D.taps;
    unsigned short* wf = (unsigned short*)D.w_fprop;
    unsigned short* wd = (unsigned short*)D.w_dgrad;
    if (D.fold != 0) {
        const int mode = (D.fold & 3) - 1, Co = (D.fold >> 2) & 0x7fff, Ci = (D.fold >> 17) & 0x7fff;
        const int G = mode == 0 ? 3 : mode == 1 ? 16 : 1;               // the folded cin is G groups of Cip channels
        const int Cip = D.cin / G, T = D.taps;
        const int Tm = mode == 0 ? 27 : mode == 1 ? 16 : T;            // master elements per (co, ci): contiguous in the master
        // master channels per tile: chosen so that the tile fits the LDS of the unfolded path (8 x (64 x 9 + 1) floats = 18 KB) -- a launch
        // reserves ONE size for all its workgroups, and at the 55 KB of an 8 x 64 x 27 tile two workgroups per CU instead of eight
        // slowed every launch down, the unfolded layers' included (default step: 29 -> 50 us per call, profiles/r06/time_pack.txt)
        const int CT = mode == 0 ? 16 : mode == 1 ? 32 : PACK_TCI;
        if (Cip % CT == 0 && D.cout % PACK_TCO == 0 && Tm <= (mode == 2 ? 9 : 27)) {
            // Tiled over the MASTER: 8 co x CT ci x all Tm taps -- per co one contiguous run of up to CT * Tm floats, read
            // coalesced ONCE (the element-wise gather below touches every 64- / 108-byte group of the master from G tiles, four
            // bytes at a time) -- then every group g writes its (tap, co, g Cip + ci) / (tap, g Cip + ci, co) slice of the images
            // as the unfolded path does.  Pitch per ci odd (16 -> 17): the ci-fastest reads stay conflict-free.
            const int Tp = Tm | 1, rowm = CT * Tp + 1;
            const int tiles_ci = Cip / CT, tiles = tiles_ci * (D.cout / PACK_TCO);
            for (int tile = (int)blockIdx.x - D.block_begin; tile < tiles; tile += nblk) {
                const int co0 = (tile / tiles_ci) * PACK_TCO, c0 = (tile % tiles_ci) * CT;
                for (int e = threadIdx.x; e < PACK_TCO * CT * Tm; e += 256) {
                    const int r = e / (CT * Tm), cc = e - r * (CT * Tm), c = cc / Tm, j = cc - c * Tm;
                    const bool ok = co0 + r < Co && c0 + c < Ci;
                    ptile[r * rowm + c * Tp + j] = ok ? D.w[((long)(co0 + r) * Ci + c0) * Tm + cc] * D.scale : 0.f;
                }
                __syncthreads();
                for (int e = threadIdx.x; e < G * T * PACK_TCO * CT; e += 256) {
                    const int gt = e / (PACK_TCO * CT), g = gt / T, tap = gt - g * T, j = mode == 0 ? g * 9 + tap : mode == 1 ? g : tap;
                    const int l = e - gt * (PACK_TCO * CT);
                    if (wf) {                                                               // (tap, co, ci), ci fastest
                        const int ci = l % CT, r = l / CT;
                        wf[((long)tap * D.cout + co0 + r) * D.cin + g * Cip + c0 + ci] = f32_to_bf16_bits(ptile[r * rowm + ci * Tp + j]);
                    }
                    if (wd) {                                                               // (tap, ci, co), co fastest
                        const int r = l % PACK_TCO, ci = l / PACK_TCO;
                        wd[((long)(T - 1 - tap) * D.cin + g * Cip + c0 + ci) * D.cout + co0 + r] = f32_to_bf16_bits(ptile[r * rowm + ci * Tp + j]);
                    }
                }
                __syncthreads();
            }
            return;
        }
    } else if (D.taps <= 9 && D.cin % PACK_TCI == 0 && D.cout % PACK_TCO == 0) {
        // Tiled: the master's rows (one co: cin x taps contiguous) are read coalesced into LDS, the fprop image leaves in
        // 128-byte runs along ci and the dgrad image in 16-byte runs along co -- the element-wise form below gathers the
        // master with a stride of `taps` floats and scatters the dgrad image 2 bytes at a time.
        const int T = D.taps, row = PACK_TCI * T + 1;     // odd row pitch: the co-fastest reads below stay conflict-free
        const int tiles_ci = D.cin / PACK_TCI, tiles = tiles_ci * (D.cout / PACK_TCO);
        for (int tile = (int)blockIdx.x - D.block_begin; tile < tiles; tile += nblk) {
            const int co0 = (tile / tiles_ci) * PACK_TCO, ci0 = (tile % tiles_ci) * PACK_TCI;
            for (int e = threadIdx.x; e < PACK_TCO * PACK_TCI * T; e += 256) {
                const int r = e / (PACK_TCI * T), c = e - r * (PACK_TCI * T);
                ptile[r * row + c] = D.w[((long)(co0 + r) * D.cin + ci0) * T + c] * D.scale;
            }
            __syncthreads();
            if (wf)
                for (int e = threadIdx.x; e < T * PACK_TCO * PACK_TCI; e += 256) {          // (tap, co, ci), ci fastest
                    const int ci = e % PACK_TCI, r = e / PACK_TCI % PACK_TCO, tap = e / (PACK_TCI * PACK_TCO);
                    wf[((long)tap * D.cout + co0 + r) * D.cin + ci0 + ci] = f32_to_bf16_bits(ptile[r * row + ci * T + tap]);
                }
            if (wd)
                for (int e = threadIdx.x; e < T * PACK_TCO * PACK_TCI; e += 256) {          // (tap, ci, co), co fastest
                    const int r = e % PACK_TCO, ci = e / PACK_TCO % PACK_TCI, tap = e / (PACK_TCO * PACK_TCI);
                    wd[((long)(T - 1 - tap) * D.cin + ci0 + ci) * D.cout + co0 + r] = f32_to_bf16_bits(ptile[r * row + ci * T + tap]);
                }
            __syncthreads();
        }
        return;
    }
    for (long e = (long)((int)blockIdx.x - D.block_begin) * 256 + threadIdx.x; e < total; e += (long)nblk * 256) {
        const int ci = (int)(e % D.cin);
        const long r = e / D.cin;
        const int co = (int)(r % D.cout);
        const int tap = (int)(r / D.cout);
        const long m = pack_src_index(D, co, ci, tap);
        const unsigned short h = f32_to_bf16_bits(m >= 0 ? D.w[m] * D.scale : 0.f);
        if (wf) wf[e] = h;
        if (wd) wd[((long)(D.taps - 1 - tap) * D.cin + ci) * D.cout + co] = h;
    }
}
}  // namespace

extern "C" int rgbd_pack_weights_multi(const rgbd_pack_desc* descs_device, int n, int total_blocks, void* stream) {
    RGBD_REQUIRE(descs_device && n > 0 && total_blocks > 0, "rgbd_pack_weights_multi: bad arguments");
    constexpr int lds = PACK_TCO * (PACK_TCI * 9 + 1) * (int)sizeof(float);       // every tile shape of the kernel fits this
    RGBD_REQUIRE(rgbd_reserve_lds((const void*)&pack_weights_multi_kernel, lds),
                 "rgbd_pack_weights_multi: cannot reserve %d B of LDS", lds);
    pack_weights_multi_kernel<<<total_blocks, 256, lds, (hipStream_t)stream>>>(descs_device, n);
    RGBD_CHECK_LAUNCH("pack_weights_multi_kernel");
    return 0;
}

extern "C" int64_t rgbd_adain_workspace(int B, int HW, int C) {
    if (B <= 0 || HW <= 0 || C <= 0) return -1;
    return (int64_t)ceil_div(HW, ADAIN_STRIP) * B * C * 2;
}

extern "C" int rgbd_adain_fwd(const void* x, const float* scale, const float* shift, void* y, float* sums,
                              float* mean, float* rstd, int B, int HW, int C, int c_live, int ld, float eps, void* y_q, void* y_s,
                              void* stream) {
    RGBD_REQUIRE(!y_q || (y_s && C % 32 == 0), "rgbd_adain_fwd: the MXFP8 copy needs its scale buffer");
    RGBD_REQUIRE(x && scale && shift && y && sums && mean && rstd, "rgbd_adain_fwd: null pointer");
    RGBD_REQUIRE(ld >= C, "rgbd_adain_fwd: ld must be >= C");
    RGBD_REQUIRE(B > 0 && HW > 0 && C > 0 && C % 64 == 0, "rgbd_adain_fwd: C must be a multiple of 64 (C=%d)", C);
    RGBD_REQUIRE(c_live > 0 && c_live <= C && c_live % 8 == 0, "rgbd_adain_fwd: c_live must be a multiple of 8 in (0, C] (c_live=%d)", c_live);
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(ceil_div(HW, ADAIN_STRIP), C / 64, B);
    adain_reduce_kernel<false><<<grid, 256, 0, st>>>((const unsigned short*)x, nullptr, nullptr, nullptr, sums, HW, C);
    RGBD_CHECK_LAUNCH("adain_reduce_kernel");
    const int rows = HW <= 4096 ? 256 : 512;
    dim3 agrid(ceil_div(HW, rows), C / 64, B);
    adain_apply_kernel<<<agrid, 256, 0, st>>>((const unsigned short*)x, scale, shift, sums, mean, rstd,
                                              (unsigned short*)y, HW, C, ld, 1.f / (float)HW, eps, rows, (int)grid.x,
                                              (unsigned char*)y_q, (unsigned char*)y_s, c_live);
    RGBD_CHECK_LAUNCH("adain_apply_kernel");
    return 0;
}

extern "C" int rgbd_adain_apply_fixed(const void* x, const float* scale, const float* shift, void* y, const int64_t* stats,
                                      float* mean, float* rstd, int B, int HW, int C, int ld, float eps, void* y_q, void* y_s,
                                      void* stream) {
    RGBD_REQUIRE(!y_q || y_s, "rgbd_adain_apply_fixed: the MXFP8 copy needs its scale buffer");
    RGBD_REQUIRE(x && scale && shift && y && stats && mean && rstd, "rgbd_adain_apply_fixed: null pointer");
    RGBD_REQUIRE(ld >= C, "rgbd_adain_apply_fixed: ld must be >= C");
    RGBD_REQUIRE(B > 0 && HW > 0 && C > 0 && C % 64 == 0, "rgbd_adain_apply_fixed: C must be a multiple of 64 (C=%d)", C);
    const int rows = HW <= 4096 ? 256 : 512;
    dim3 agrid(ceil_div(HW, rows), C / 64, B);
    adain_apply_kernel<<<agrid, 256, 0, (hipStream_t)stream>>>((const unsigned short*)x, scale, shift, (const float*)stats,
                                                                mean, rstd, (unsigned short*)y, HW, C, ld, 1.f / (float)HW,
                                                                eps, rows, -32, (unsigned char*)y_q, (unsigned char*)y_s, C);
    RGBD_CHECK_LAUNCH("adain_apply_kernel");
    return 0;
}

extern "C" int rgbd_adain_bwd(const void* x, const void* dy, const float* scale, const float* mean,
                              const float* rstd, void* dx, float* dscale, float* dshift, float* sums, int B,
                              int HW, int C, int c_live, int ld, float lrelu_slope, float* bias_grad, void* dx_q, void* dx_s,
                              void* stream) {
    RGBD_REQUIRE(c_live > 0 && c_live <= C && c_live % 8 == 0, "rgbd_adain_bwd: c_live must be a multiple of 8 in (0, C] (c_live=%d)", c_live);
    RGBD_REQUIRE(!dx_q || dx_s, "rgbd_adain_bwd: the MXFP8 copy needs its scale buffer");
    unsigned char* const qq = (unsigned char*)dx_q;
    unsigned char* const qs = (unsigned char*)dx_s;
    RGBD_REQUIRE(x && dy && scale && mean && rstd && dx && dscale && dshift && sums, "rgbd_adain_bwd: null pointer");
    RGBD_REQUIRE(ld >= C, "rgbd_adain_bwd: ld must be >= C");
    RGBD_REQUIRE(B > 0 && HW > 0 && C > 0 && C % 64 == 0, "rgbd_adain_bwd: C must be a multiple of 64 (C=%d)", C);
    RGBD_REQUIRE(!bias_grad || lrelu_slope > 0.f, "rgbd_adain_bwd: bias_grad comes with the activation gradient (lrelu_slope > 0)");
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(ceil_div(HW, ADAIN_STRIP), C / 64, B);
    adain_reduce_kernel<true><<<grid, 256, 0, st>>>((const unsigned short*)x, (const unsigned short*)dy, mean, rstd,
                                                    sums, HW, C);
    RGBD_CHECK_LAUNCH("adain_reduce_kernel<dy>");
    const int rows = HW <= 4096 ? 256 : 512;
    dim3 agrid(ceil_div(HW, rows), C / 64, B);
    if (lrelu_slope > 0.f && bias_grad && HW >= 4096) {
        const int big_rows = 2048;
        dim3 bgrid(ceil_div(HW, big_rows), C / 64, B);
        adain_bwd_apply_kernel<true, 1024><<<bgrid, 1024, 0, st>>>((const unsigned short*)x, (const unsigned short*)dy, scale,
                                                                   mean, rstd, sums, (unsigned short*)dx, dscale, dshift, HW,
                                                                   C, 1.f / (float)HW, ld, big_rows, lrelu_slope, bias_grad,
                                                                   (int)grid.x, qq, qs, c_live);
    } else if (lrelu_slope > 0.f)
        adain_bwd_apply_kernel<true><<<agrid, 256, 0, st>>>((const unsigned short*)x, (const unsigned short*)dy, scale,
                                                            mean, rstd, sums, (unsigned short*)dx, dscale, dshift, HW, C,
                                                            1.f / (float)HW, ld, rows, lrelu_slope, bias_grad, (int)grid.x, qq, qs, c_live);
    else
        adain_bwd_apply_kernel<false><<<agrid, 256, 0, st>>>((const unsigned short*)x, (const unsigned short*)dy, scale,
                                                             mean, rstd, sums, (unsigned short*)dx, dscale, dshift, HW,
                                                             C, 1.f / (float)HW, ld, rows, 0.f, nullptr, (int)grid.x, qq, qs, c_live);
    RGBD_CHECK_LAUNCH("adain_bwd_apply_kernel");
    return 0;
}

// Strip size of the passes that carry a column sum (one atomic per channel per block at the end: with 512-row strips the
// 128x128 layers ended in 1024 x 64 same-address atomics, ~15 us of serialised tail).  Large tensors: 1024-thread
// blocks on 2048-row strips; small ones: 256 threads on strips short enough to give every CU a block.
struct ColsumPlan { int threads, rows; };
static ColsumPlan plan_colsum(long M) {
#ifdef RGBD_DEBUG_BUILD
    static const int dbg_rows = [] { const char* e = getenv("RGBD_DEBUG_COLSUM_ROWS"); return e ? atoi(e) : 0; }();
    if (dbg_rows > 0) return ColsumPlan{dbg_rows >= 1024 ? 1024 : 256, dbg_rows};
#endif
    if (M >= 131072) return ColsumPlan{1024, 2048};
    if (M >= 32768) return ColsumPlan{256, 256};
    return ColsumPlan{256, 128};
}

extern "C" int rgbd_lrelu_bwd(const void* dy, const void* y, void* dz, int64_t M, int C, int act_channels,
                              float slope, float* bias_grad, const float* row_scale, int64_t rows_per_sample,
                              void* stream) {
    RGBD_REQUIRE(dy && y && dz, "rgbd_lrelu_bwd: null pointer");
    RGBD_REQUIRE(M > 0 && C > 0 && C % 8 == 0 && act_channels % 8 == 0, "rgbd_lrelu_bwd: C must be a multiple of 8");
    hipStream_t st = (hipStream_t)stream;
    if (bias_grad) {
        RGBD_REQUIRE(C % 64 == 0, "rgbd_lrelu_bwd: the fused bias gradient needs C %% 64 == 0 (C=%d)", C);
        RGBD_REQUIRE(!row_scale || rows_per_sample > 0, "rgbd_lrelu_bwd: rows_per_sample must be positive with row_scale");
        const ColsumPlan cp = plan_colsum(M);
        dim3 grid(ceil_div(M, cp.rows), C / 64);
        if (cp.threads == 1024)
            lrelu_bwd_colsum_kernel<1024><<<grid, 1024, 0, st>>>((const unsigned short*)dy, (const unsigned short*)y,
                                                                 (unsigned short*)dz, M, C, act_channels, slope, cp.rows,
                                                                 bias_grad, row_scale, row_scale ? rows_per_sample : 1);
        else
            lrelu_bwd_colsum_kernel<256><<<grid, 256, 0, st>>>((const unsigned short*)dy, (const unsigned short*)y,
                                                               (unsigned short*)dz, M, C, act_channels, slope, cp.rows,
                                                               bias_grad, row_scale, row_scale ? rows_per_sample : 1);
        RGBD_CHECK_LAUNCH("lrelu_bwd_colsum_kernel");
        return 0;
    }
    const long nvec = M * C / 8;
    const int blocks = (int)min((long)4096, (nvec + 255) / 256);
    lrelu_bwd_kernel<<<blocks, 256, 0, st>>>((const unsigned short*)dy, (const unsigned short*)y,
                                             (unsigned short*)dz, nvec, C, act_channels, slope);
    RGBD_CHECK_LAUNCH("lrelu_bwd_kernel");
    return 0;
}

extern "C" int rgbd_colsum_bf16(const void* x, float* out, int64_t M, int C, int accumulate, const float* row_scale,
                                int64_t rows_per_sample, void* stream) {
    RGBD_REQUIRE(x && out, "rgbd_colsum_bf16: null pointer");
    RGBD_REQUIRE(M > 0 && C > 0 && C % 64 == 0, "rgbd_colsum_bf16: C must be a multiple of 64 (C=%d)", C);
    RGBD_REQUIRE(!row_scale || rows_per_sample > 0, "rgbd_colsum_bf16: rows_per_sample must be positive with row_scale");
    hipStream_t st = (hipStream_t)stream;
    if (!accumulate && rgbd_zero_async(out, (size_t)C * sizeof(float), st) != hipSuccess) {
        rgbd_set_error("rgbd_colsum_bf16: memset failed");
        return -2;
    }
    const ColsumPlan cp = plan_colsum(M);
    dim3 grid(ceil_div(M, cp.rows), C / 64);
    if (cp.threads == 1024)
        colsum_kernel<1024><<<grid, 1024, 0, st>>>((const unsigned short*)x, out, M, C, cp.rows, row_scale,
                                                   row_scale ? rows_per_sample : 1);
    else
        colsum_kernel<256><<<grid, 256, 0, st>>>((const unsigned short*)x, out, M, C, cp.rows, row_scale,
                                                 row_scale ? rows_per_sample : 1);
    RGBD_CHECK_LAUNCH("colsum_kernel");
    return 0;
}

extern "C" int rgbd_axpy_rows_bf16(const void* a, const void* x, const float* s, void* out, int64_t B,
                                   int64_t elems_per_sample, void* stream) {
    RGBD_REQUIRE(a && x && s && out, "rgbd_axpy_rows_bf16: null pointer");
    RGBD_REQUIRE(B > 0 && elems_per_sample > 0 && elems_per_sample % 8 == 0,
                 "rgbd_axpy_rows_bf16: elements per sample must be a multiple of 8");
    const long nvec = B * elems_per_sample / 8;
    axpy_rows_kernel<<<(int)min((long)4096, (nvec + 255) / 256), 256, 0, (hipStream_t)stream>>>(
        (const unsigned short*)a, (const unsigned short*)x, s, (unsigned short*)out, nvec, elems_per_sample / 8);
    RGBD_CHECK_LAUNCH("axpy_rows_kernel");
    return 0;
}

extern "C" int rgbd_unpool2_lrelu_bwd(const void* dp, const void* y, void* dz, int B, int H, int W, int C, float slope,
                                      float* bias_grad, float* bias_grad2, const float* row_scale, void* dz_q, void* dz_s,
                                      void* stream) {
    RGBD_REQUIRE(!dz_q || dz_s, "rgbd_unpool2_lrelu_bwd: the MXFP8 copy needs its scale buffer");
    RGBD_REQUIRE(dp && dz, "rgbd_unpool2_lrelu_bwd: null pointer");
    RGBD_REQUIRE(B > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && C % 64 == 0,
                 "rgbd_unpool2_lrelu_bwd: H, W must be even and C a multiple of 64 (H=%d W=%d C=%d)", H, W, C);
    const long M = (long)B * H * W;
    const ColsumPlan cp = plan_colsum(M);
    dim3 grid(ceil_div(M, cp.rows), C / 64);
    if (cp.threads == 1024)
        unpool_lrelu_bwd_kernel<1024><<<grid, 1024, 0, (hipStream_t)stream>>>(
            (const unsigned short*)dp, (const unsigned short*)y, (unsigned short*)dz, M, H, W, C, slope, cp.rows, bias_grad,
            bias_grad ? bias_grad2 : nullptr, row_scale, (unsigned char*)dz_q, (unsigned char*)dz_s);
    else
        unpool_lrelu_bwd_kernel<256><<<grid, 256, 0, (hipStream_t)stream>>>(
            (const unsigned short*)dp, (const unsigned short*)y, (unsigned short*)dz, M, H, W, C, slope, cp.rows, bias_grad,
            bias_grad ? bias_grad2 : nullptr, row_scale, (unsigned char*)dz_q, (unsigned char*)dz_s);
    RGBD_CHECK_LAUNCH("unpool_lrelu_bwd_kernel");
    return 0;
}

extern "C" int rgbd_pool2_masked(const void* x, const void* y, void* out, int B, int H, int W, int C, float slope,
                                 void* stream) {
    RGBD_REQUIRE(x && out, "rgbd_pool2_masked: null pointer");
    RGBD_REQUIRE(B > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && C % 8 == 0,
                 "rgbd_pool2_masked: H, W must be even and C a multiple of 8 (H=%d W=%d C=%d)", H, W, C);
    const long nvec = (long)B * (H / 2) * (W / 2) * C / 8;
    const int blocks = (int)min((long)4096, (nvec + 255) / 256);
    pool2_masked_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>((const unsigned short*)x, (const unsigned short*)y,
                                                                 (unsigned short*)out, nvec, H, W, C, slope, 0.25f);
    RGBD_CHECK_LAUNCH("pool2_masked_kernel");
    return 0;
}

extern "C" int rgbd_pool2_sum_bf16(const void* x, void* out, int B, int H, int W, int C, void* stream) {
    RGBD_REQUIRE(x && out, "rgbd_pool2_sum_bf16: null pointer");
    RGBD_REQUIRE(B > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && C % 8 == 0, "rgbd_pool2_sum_bf16: bad shape");
    const long nvec = (long)B * (H / 2) * (W / 2) * C / 8;
    const int blocks = (int)min((long)4096, (nvec + 255) / 256);
    pool2_masked_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>((const unsigned short*)x, nullptr, (unsigned short*)out, nvec,
                                                                 H, W, C, 0.f, 1.0f);
    RGBD_CHECK_LAUNCH("pool2_masked_kernel");
    return 0;
}

extern "C" int rgbd_from_planes(const float* x, const float* w, const float* bias, void* y, int B, int HW, int KP,
                                int C, float wscale, int act, float slope, void* stream) {
    RGBD_REQUIRE(x && w && y, "rgbd_from_planes: null pointer");
    RGBD_REQUIRE((KP == 3 || KP == 4) && C % 8 == 0 && B > 0 && HW > 0, "rgbd_from_planes: bad shape KP=%d C=%d", KP, C);
    const long nvec = (long)B * HW * C / 8;
    const int blocks = (int)min((long)8192, (nvec + 1023) / 1024);       // four items per thread (two rounds with four planes)
    hipStream_t st = (hipStream_t)stream;
    if (KP == 3) from_planes_kernel<3><<<blocks, 256, 0, st>>>(x, w, bias, (unsigned short*)y, B, HW, C, wscale, act, slope);
    else         from_planes_kernel<4><<<blocks, 256, 0, st>>>(x, w, bias, (unsigned short*)y, B, HW, C, wscale, act, slope);
    RGBD_CHECK_LAUNCH("from_planes_kernel");
    return 0;
}

extern "C" int rgbd_to_planes(const void* h, const float* w, const float* bias, float* out, int B, int HW, int KP,
                              int C, float wscale, void* stream) {
    RGBD_REQUIRE(h && w && out, "rgbd_to_planes: null pointer");
    RGBD_REQUIRE((KP == 3 || KP == 4) && C % 64 == 0 && B > 0 && HW > 0, "rgbd_to_planes: bad shape KP=%d C=%d", KP, C);
    const long npix = (long)B * HW;
    const int blocks = (int)min((long)8192, (npix + 31) / 32);
    hipStream_t st = (hipStream_t)stream;
    if (KP == 3) to_planes_kernel<3><<<blocks, 256, 0, st>>>((const unsigned short*)h, w, bias, out, B, HW, C, wscale);
    else         to_planes_kernel<4><<<blocks, 256, 0, st>>>((const unsigned short*)h, w, bias, out, B, HW, C, wscale);
    RGBD_CHECK_LAUNCH("to_planes_kernel");
    return 0;
}

extern "C" int rgbd_planes_outer(const void* t, const float* p, float* o, float* tsum, float* psum, int B, int HW, int KP,
                                 int C, void* stream) {
    RGBD_REQUIRE(t && p && o, "rgbd_planes_outer: null pointer");
    RGBD_REQUIRE((KP == 3 || KP == 4) && C % 64 == 0 && B > 0 && HW > 0, "rgbd_planes_outer: bad shape KP=%d C=%d", KP, C);
    hipStream_t st = (hipStream_t)stream;
    if (rgbd_zero_async(o, (size_t)KP * C * sizeof(float), st) != hipSuccess ||
        (tsum && rgbd_zero_async(tsum, (size_t)C * sizeof(float), st) != hipSuccess)) {
        rgbd_set_error("rgbd_planes_outer: memset failed");
        return -2;
    }
    // rows per block: every block ends with (KP + 1) * 64 atomics on the same (KP + 1) * C addresses, so few, long blocks
    // (512-row blocks: 327 K atomics, ~1000 deep per address, were 30 of the launch's 38 us) -- but at least ~256 of them
    int rows = 512;
    while (rows < 4096 && (long)ceil_div(HW, 2 * rows) * (C / 64) * B >= 256) rows *= 2;
    dim3 grid(ceil_div(HW, rows), C / 64, B);
    if (KP == 3) planes_outer_kernel<3><<<grid, 256, 0, st>>>((const unsigned short*)t, p, o, tsum, psum, B, HW, C, rows);
    else         planes_outer_kernel<4><<<grid, 256, 0, st>>>((const unsigned short*)t, p, o, tsum, psum, B, HW, C, rows);
    RGBD_CHECK_LAUNCH("planes_outer_kernel");
    return 0;
}

namespace {
constexpr int LIN_SPLITK_MAX = 16;
// K slices of the forward linear: 1 below K = 1024; else enough 256-deep slices to put ~256 blocks on the chip
int linear_ksplit(int K, int N) {
    if (K < 1024) return 1;
    int s = 256 / ((N + 15) / 16);
    if (s > K / 256) s = K / 256;
    if (s > LIN_SPLITK_MAX) s = LIN_SPLITK_MAX;
    return s < 1 ? 1 : s;
}
int linear_fwd_launch(const float* x, const float* w, const float* bias, const float* mask_y, float* y, int M, int K, int N,
                      float c, int act, float slope, float* workspace, hipStream_t st) {
    const int ks = workspace ? linear_ksplit(K, N) : 1;
    const dim3 grid((N + 15) / 16, ks);
    if ((K & 3) == 0) linear_fwd_kernel<true><<<grid, 256, 0, st>>>(x, w, bias, y, M, K, N, c, act, slope, mask_y, workspace);
    else              linear_fwd_kernel<false><<<grid, 256, 0, st>>>(x, w, bias, y, M, K, N, c, act, slope, mask_y, workspace);
    RGBD_CHECK_LAUNCH("linear_fwd_kernel");
    if (ks > 1) {
        linear_splitk_finish_kernel<<<(M * N + 255) / 256, 256, 0, st>>>(workspace, ks, M, N, bias, c, act, slope, mask_y, y);
        RGBD_CHECK_LAUNCH("linear_splitk_finish_kernel");
    }
    return 0;
}
}  // namespace

extern "C" int64_t rgbd_linear_fwd_workspace(int M, int K, int N) {
    if (M <= 0 || K <= 0 || N <= 0) return -1;
    const int ks = linear_ksplit(K, N);
    return ks > 1 ? (int64_t)ks * M * N : 0;          /* floats */
}

extern "C" int rgbd_linear_fwd(const float* x, const float* w, const float* bias, float* y, int M, int K, int N,
                               float c, int act, float slope, float* workspace, void* stream) {
    RGBD_REQUIRE(x && w && y, "rgbd_linear_fwd: null pointer");
    RGBD_REQUIRE(M > 0 && M <= LIN_MAXM && K > 0 && N > 0, "rgbd_linear_fwd: needs 0 < M <= %d (M=%d)", LIN_MAXM, M);
    return linear_fwd_launch(x, w, bias, nullptr, y, M, K, N, c, act, slope, workspace, (hipStream_t)stream);
}

extern "C" int rgbd_linear_fwd_masked(const float* x, const float* w, const float* mask_y, float* y, int M, int K, int N,
                                      float c, float slope, float* workspace, void* stream) {
    RGBD_REQUIRE(x && w && y && mask_y, "rgbd_linear_fwd_masked: null pointer");
    RGBD_REQUIRE(M > 0 && M <= LIN_MAXM && K > 0 && N > 0, "rgbd_linear_fwd_masked: needs 0 < M <= %d (M=%d)", LIN_MAXM, M);
    return linear_fwd_launch(x, w, nullptr, mask_y, y, M, K, N, c, 0, slope, workspace, (hipStream_t)stream);
}

extern "C" int rgbd_linear_bwd(const float* dy, const float* y, const float* x, const float* w, float* dx, float* dw,
                               float* db, int M, int K, int N, float c, int act, float slope, int accumulate_dx,
                               void* stream) {
    RGBD_REQUIRE(dy && w && (x || !dw), "rgbd_linear_bwd: null pointer (x is needed for dw)");
    RGBD_REQUIRE(!act || y, "rgbd_linear_bwd: the activation output y is needed for its gradient");
    RGBD_REQUIRE(M > 0 && M <= LIN_MAXM && K > 0 && N > 0, "rgbd_linear_bwd: needs 0 < M <= %d (M=%d)", LIN_MAXM, M);
    hipStream_t st = (hipStream_t)stream;
    if (dx) {
        if ((N & 3) == 0) linear_dgrad_kernel<true><<<(K + 15) / 16, 256, 0, st>>>(dy, y, w, dx, M, K, N, c, act, slope, accumulate_dx);
        else              linear_dgrad_kernel<false><<<(K + 15) / 16, 256, 0, st>>>(dy, y, w, dx, M, K, N, c, act, slope, accumulate_dx);
        RGBD_CHECK_LAUNCH("linear_dgrad_kernel");
    }
    if (dw) {
        linear_wgrad_kernel<<<dim3((K + 63) / 64, (N + 15) / 16), 256, 0, st>>>(dy, y, x, dw, db, M, K, N, c, act, slope);
        RGBD_CHECK_LAUNCH("linear_wgrad_kernel");
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------- fused MLP chain
namespace {
// The mapping network (net.py:22-62: 8 x [equalized linear C -> C, leaky ReLU] on 2B <= 64 latent rows) as ONE launch per pass
// instead of one per layer (forward) / two per layer (backward).  The chain is a dependent sequence of tiny GEMMs: per layer a
// launch costs a kernel boundary plus a dependent global-load round trip (~10 us), eight of them sit at the head of the
// generator's forward and sixteen at the tail of its backward, on the step's critical stream.
// One 8-wave workgroup per 16 rows walks ALL layers: the 16 x C activations live in LDS (double-buffered), every wave owns
// C/128 column tiles of 16 and streams its 16 x 256 slices of W straight from global memory (L2-resident: the whole network is
// 2 MB at C = 256) into registers in the fp32 MFMA's B-operand layout -- the slice of the NEXT unit is requested before the
// current one is multiplied, across layer boundaries too (weights do not depend on activations), so after the first slice
// the kernel runs at the CU's fp32 matrix rate: 16 x C x C MACs per layer = 3.4 us at C = 256.  No cross-workgroup hand-off:
// rows are independent.  fp32 throughout (v_mfma_f32_16x16x4_f32), the summation order over k differs from the per-layer
// kernels' (quads of k interleaved the same way, halves of K accumulated in sequence).
constexpr int MLP_MAX_LAYERS = 8;
struct MlpArgs {
    const float* w[MLP_MAX_LAYERS];     // (C,C) row-major [n][k] master weights
    const float* b[MLP_MAX_LAYERS];     // (C) or null (forward only)
    const float* x;                     // forward: (M,C) input;  backward: (M,C) gradient of the last layer's output
    const float* acts_in;               // backward: (L,M,C) the layers' outputs
    float* acts;                        // forward: (L,M,C) out
    float* dz;                          // backward: (L,M,C) out: dz[l] = d loss / d (pre-activation of layer l)
    float* dx;                          // backward: (M,C) out
    int L, M;
    float c, slope;
};

template <int C, bool BWD>
__global__ __launch_bounds__(512) void mlp_chain_kernel(MlpArgs a) {
    constexpr int LD = C + 4;
    constexpr int KH = C / 256;                     // K halves of 256 per column tile
    constexpr int CTW = C / 16 / 8;                 // column tiles per wave
    constexpr int UPL = CTW * KH;                   // units (16 columns x 256 k) per wave per layer: 2 (C = 256), 8 (C = 512)
    static_assert(C % 256 == 0 && UPL % 2 == 0, "mlp_chain_kernel: C must be a multiple of 256");
    __shared__ float xs[2][16][LD];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 15, q = lane >> 4;
    const int m0 = blockIdx.x * 16;
    const int L = a.L, M = a.M;
    // ---- the chain's input into LDS (rows beyond M: zeros)
    for (int i = threadIdx.x; i < 16 * (C / 4); i += 512) {
        const int row = i / (C / 4), c4 = (i % (C / 4)) * 4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (m0 + row < M) {
            v = *reinterpret_cast<const f32x4*>(a.x + (long)(m0 + row) * C + c4);
            if (BWD) {                              // dz[L-1] = dy * lrelu'(y[L-1])
                const f32x4 y = *reinterpret_cast<const f32x4*>(a.acts_in + ((long)(L - 1) * M + m0 + row) * C + c4);
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = y[j] > 0.f ? v[j] : v[j] * a.slope;
                *reinterpret_cast<f32x4*>(a.dz + ((long)(L - 1) * M + m0 + row) * C + c4) = v;
            }
        }
        *reinterpret_cast<f32x4*>(&xs[0][row][c4]) = v;
    }
    __syncthreads();

    auto load_unit = [&](int u, f32x4 (&b)[16]) {
        const int lay = BWD ? L - 1 - u / UPL : u / UPL, within = u % UPL;
        const int ct = wave + 8 * (within / KH), kh = within % KH;
        const float* w = a.w[lay];
#pragma unroll
        for (int s2 = 0; s2 < 16; ++s2) {
            const int kk = 256 * kh + 16 * s2 + 4 * q;          // reduction index of this lane's quad
            if (!BWD) {
                b[s2] = *reinterpret_cast<const f32x4*>(w + (long)(16 * ct + r) * C + kk);        // W[n][k..k+3]
            } else {
                // W[n..n+3][k]: a wave-uniform row base (scalar registers) + ONE per-lane offset for all 64 loads of a unit
                const int lane_off = 4 * q * C + r;
#pragma unroll
                for (int j = 0; j < 4; ++j) b[s2][j] = (w + (long)(256 * kh + 16 * s2 + j) * C + 16 * ct)[lane_off];
            }
        }
    };
    int cur = 0;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    auto compute_unit = [&](int u, const f32x4 (&b)[16]) {
        const int lay = BWD ? L - 1 - u / UPL : u / UPL, within = u % UPL;
        const int ct = wave + 8 * (within / KH), kh = within % KH;
        if (kh == 0) acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s2 = 0; s2 < 16; ++s2) {
            const f32x4 av = *reinterpret_cast<const f32x4*>(&xs[cur][r][256 * kh + 16 * s2 + 4 * q]);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], b[s2][j], acc, 0, 0, 0);
        }
        if (kh != KH - 1) return;
        const int col = 16 * ct + r;
        const float bv = (!BWD && a.b[lay]) ? a.b[lay][col] : 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int row = 4 * q + t, m = m0 + row;
            float v = acc[t] * a.c + bv;
            if (!BWD) {
                v = v > 0.f ? v : v * a.slope;
                if (m < M) a.acts[((long)lay * M + m) * C + col] = v;
            } else if (lay > 0) {                   // gradient w.r.t. layer lay-1's output -> w.r.t. its pre-activation
                const float y = m < M ? a.acts_in[((long)(lay - 1) * M + m) * C + col] : 0.f;
                v = y > 0.f ? v : v * a.slope;
                if (m < M) a.dz[((long)(lay - 1) * M + m) * C + col] = v;
            } else if (m < M) {
                a.dx[(long)m * C + col] = v;
            }
            xs[cur ^ 1][row][col] = v;
        }
    };

    const int total = L * UPL;
    f32x4 bA[16], bB[16];
    load_unit(0, bA);
    for (int u = 0; u < total; u += 2) {
        load_unit(u + 1, bB);
        compute_unit(u, bA);
        if (u + 2 < total) load_unit(u + 2, bA);    // the next layer's first slice is on its way before the barrier below
        compute_unit(u + 1, bB);
        if ((u + 2) % UPL == 0) {                   // layer done (uniform: every wave has UPL units per layer)
            __syncthreads();
            cur ^= 1;
        }
    }
}

// dW[l][n][k] += c * sum_m dz[l][m][n] * in_l[m][k],  db[l][n] += sum_m dz[l][m][n]  for ALL layers in one launch
// (in_0 = the chain's input, in_l = acts[l-1]).  M <= a few dozen rows: a thread owns four k of one n.
struct MlpWgradArgs {
    float* dw[MLP_MAX_LAYERS];
    float* db[MLP_MAX_LAYERS];
    const float* x;
    const float* acts;
    const float* dz;
    int L, M;
    float c;
};
template <int C>
__global__ __launch_bounds__(256) void mlp_wgrad_kernel(MlpWgradArgs a) {
    const int lay = blockIdx.y, M = a.M;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6), kq = threadIdx.x & 63;
    const float* in = lay == 0 ? a.x : a.acts + (long)(lay - 1) * M * C;
    const float* dz = a.dz + (long)lay * M * C + n;
    float* dw = a.dw[lay];
    if (!dw) return;
    float bsum = 0.f;
    for (int kk = kq * 4; kk < C; kk += 256) {
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        bsum = 0.f;
        for (int m = 0; m < M; ++m) {
            const float g = dz[(long)m * C];
            const f32x4 xv = *reinterpret_cast<const f32x4*>(in + (long)m * C + kk);
            s += g * xv;
            bsum += g;
        }
        f32x4* o = reinterpret_cast<f32x4*>(dw + (long)n * C + kk);
        *o = *o + a.c * s;
    }
    if (kq == 0 && a.db[lay]) a.db[lay][n] += bsum;
}
}  // namespace

extern "C" int rgbd_mlp_fwd(const float* x, const float* const* w_host, const float* const* b_host, int L, int M, int C,
                            float c, float slope, float* acts, void* stream) {
    RGBD_REQUIRE(x && w_host && acts, "rgbd_mlp_fwd: null pointer");
    RGBD_REQUIRE(L > 0 && L <= MLP_MAX_LAYERS && M > 0 && (C == 256 || C == 512),
                 "rgbd_mlp_fwd: needs 1 <= L <= %d layers of C = 256 or 512 (L=%d C=%d)", MLP_MAX_LAYERS, L, C);
    MlpArgs a{};
    for (int l = 0; l < L; ++l) {
        RGBD_REQUIRE(w_host[l], "rgbd_mlp_fwd: null weight pointer");
        a.w[l] = w_host[l];
        a.b[l] = b_host ? b_host[l] : nullptr;
    }
    a.x = x; a.acts = acts; a.L = L; a.M = M; a.c = c; a.slope = slope;
    const unsigned grid = (unsigned)((M + 15) / 16);
    if (C == 256) mlp_chain_kernel<256, false><<<grid, 512, 0, (hipStream_t)stream>>>(a);
    else          mlp_chain_kernel<512, false><<<grid, 512, 0, (hipStream_t)stream>>>(a);
    RGBD_CHECK_LAUNCH("mlp_chain_kernel");
    return 0;
}

extern "C" int rgbd_mlp_bwd(const float* dy, const float* x, const float* acts, const float* const* w_host,
                            float* const* dw_host, float* const* db_host, int L, int M, int C, float c, float slope,
                            float* dz, float* dx, void* stream) {
    RGBD_REQUIRE(dy && x && acts && w_host && dz && dx, "rgbd_mlp_bwd: null pointer");
    RGBD_REQUIRE(L > 0 && L <= MLP_MAX_LAYERS && M > 0 && (C == 256 || C == 512),
                 "rgbd_mlp_bwd: needs 1 <= L <= %d layers of C = 256 or 512 (L=%d C=%d)", MLP_MAX_LAYERS, L, C);
    hipStream_t st = (hipStream_t)stream;
    MlpArgs a{};
    for (int l = 0; l < L; ++l) {
        RGBD_REQUIRE(w_host[l], "rgbd_mlp_bwd: null weight pointer");
        a.w[l] = w_host[l];
    }
    a.x = dy; a.acts_in = acts; a.dz = dz; a.dx = dx; a.L = L; a.M = M; a.c = c; a.slope = slope;
    const unsigned grid = (unsigned)((M + 15) / 16);
    if (C == 256) mlp_chain_kernel<256, true><<<grid, 512, 0, st>>>(a);
    else          mlp_chain_kernel<512, true><<<grid, 512, 0, st>>>(a);
    RGBD_CHECK_LAUNCH("mlp_chain_kernel");
    if (dw_host) {
        MlpWgradArgs g{};
        for (int l = 0; l < L; ++l) {
            g.dw[l] = dw_host[l];
            g.db[l] = db_host ? db_host[l] : nullptr;
        }
        g.x = x; g.acts = acts; g.dz = dz; g.L = L; g.M = M; g.c = c;
        if (C == 256) mlp_wgrad_kernel<256><<<dim3(C / 4, L), 256, 0, st>>>(g);
        else          mlp_wgrad_kernel<512><<<dim3(C / 4, L), 256, 0, st>>>(g);
        RGBD_CHECK_LAUNCH("mlp_wgrad_kernel");
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------- small pointwise ops
namespace {
__global__ __launch_bounds__(256) void pixelnorm_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                        float* __restrict__ out, int C, float eps) {
    // one block per row; dy == nullptr: forward
    __shared__ float red[2][4];
    const float* xr = x + (long)blockIdx.x * C;
    float sq = 0.f, dot = 0.f;
    for (int c = threadIdx.x; c < C; c += 256) {
        const float v = xr[c];
        sq += v * v;
        if (dy) dot += v * dy[(long)blockIdx.x * C + c];
    }
    sq = wave_sum(sq);
    dot = wave_sum(dot);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = sq; red[1][threadIdx.x >> 6] = dot; }
    __syncthreads();
    sq = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    dot = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    const float r = 1.0f / sqrtf(sq / (float)C + eps);
    const float k = dy ? r * dot / (float)C : 0.f;           // mean(dy * y)
    for (int c = threadIdx.x; c < C; c += 256) {
        const long i = (long)blockIdx.x * C + c;
        out[i] = dy ? r * (dy[i] - xr[c] * r * k) : xr[c] * r;
    }
}

__global__ __launch_bounds__(256) void depth_head_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                         const float* __restrict__ dy, float* __restrict__ out,
                                                         long HW, long total) {
    // total = B*4*HW; dy == nullptr: forward
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int ch = (int)((i / HW) & 3);
        if (ch != 3) {
            out[i] = dy ? dy[i] : x[i];
        } else if (!dy) {
            const float v = x[i];
            const float sp = fmaxf(v, 0.f) + log1pf(expf(-fabsf(v)));
            out[i] = 1.0f / (sp + 1e-4f);
        } else {
            const float v = x[i], yy = y[i];
            const float sg = 1.0f / (1.0f + expf(-v));
            out[i] = -dy[i] * yy * yy * sg;
        }
    }
}

// The logit heads of the non-saturating GAN loss (loss_functions.py:15-28) in one single-block launch:
//   losses[0] = mean softplus(-y)   (generator's loss on fakes / discriminator's on reals)
//   losses[1] = mean softplus(+y)   (discriminator's loss on fakes)
//   seed_neg  = d losses[0] / dy = -sigmoid(-y) / n,   seed_pos = d losses[1] / dy = sigmoid(y) / n
//   ratio     = seed_neg / seed_pos = -exp(-y)
// Derivatives are taken at max(y, -60): below that seed_neg is -1/n to fp32 precision and seed_pos < 1e-26/n, while
// their ratio stays finite.  F.softplus as chainer computes it: max(x, 0) + log1p(exp(-|x|)).
__global__ __launch_bounds__(256) void gan_logit_heads_kernel(const float* __restrict__ y, int n,
                                                              float* __restrict__ losses, float* __restrict__ seed_neg,
                                                              float* __restrict__ seed_pos, float* __restrict__ ratio) {
    __shared__ float red[2][256];
    float sn = 0.f, sp = 0.f;
    const float inv_n = 1.0f / (float)n;
    for (int i = threadIdx.x; i < n; i += 256) {
        const float v = y[i];
        const float l = log1pf(expf(-fabsf(v)));
        sn += fmaxf(-v, 0.f) + l;
        sp += fmaxf(v, 0.f) + l;
        const float vc = fmaxf(v, -60.f);
        const float e = expf(-vc);                       // <= e^60, finite in fp32
        seed_neg[i] = -(e / (1.0f + e)) * inv_n;         // -sigmoid(-vc) / n
        seed_pos[i] = (1.0f / (1.0f + e)) * inv_n;       //  sigmoid(vc) / n
        ratio[i] = -e;
    }
    red[0][threadIdx.x] = sn;
    red[1][threadIdx.x] = sp;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            red[0][threadIdx.x] += red[0][threadIdx.x + s];
            red[1][threadIdx.x] += red[1][threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        losses[0] = red[0][0] * inv_n;
        losses[1] = red[1][0] * inv_n;
    }
}

// loss = mean_i softplus(t_i) * sigmoid(t_i)^gamma with t = sign * y (loss_functions.py:15-31: gamma = 0 the plain
// non-saturating terms, gamma > 0 the focal generator loss), dy_i = d loss / d y_i -- one block, tree-ordered sum
__global__ __launch_bounds__(256) void softplus_mean_kernel(const float* __restrict__ y, int n, float sign, float gamma,
                                                            float* __restrict__ loss, float* __restrict__ dy) {
    __shared__ float red[256];
    float acc = 0.f;
    const float inv_n = 1.0f / (float)n;
    for (int i = threadIdx.x; i < n; i += 256) {
        const float t = sign * y[i];
        const float sp = fmaxf(t, 0.f) + log1pf(expf(-fabsf(t)));                   // softplus(t)
        const float e = expf(-fabsf(t));
        const float sg = t >= 0.f ? 1.0f / (1.0f + e) : e / (1.0f + e);             // sigmoid(t)
        float f = sp, df = sg;
        if (gamma != 0.f) {
            const float pw = powf(sg, gamma);
            f = sp * pw;
            df = sg * pw + gamma * sp * pw * (1.0f - sg);                           // d/dt [softplus(t) sigmoid(t)^gamma]
        }
        acc += f;
        dy[i] = sign * df * inv_n;
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss[0] = red[0] * inv_n;
}

__global__ __launch_bounds__(256) void ema_kernel(float* __restrict__ dst, const float* __restrict__ src, long n,
                                                  float tau) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        float d = dst[i];
        d *= (1.0f - tau);                 // copy_param.py:30-31: two statements, two roundings
        d += tau * src[i];
        dst[i] = d;
    }
}
}  // namespace

extern "C" int rgbd_pixelnorm_fwd(const float* x, float* y, int M, int C, float eps, void* stream) {
    RGBD_REQUIRE(x && y && M > 0 && C > 0, "rgbd_pixelnorm_fwd: bad arguments");
    pixelnorm_kernel<<<M, 256, 0, (hipStream_t)stream>>>(x, nullptr, y, C, eps);
    RGBD_CHECK_LAUNCH("pixelnorm_kernel");
    return 0;
}

extern "C" int rgbd_pixelnorm_bwd(const float* x, const float* dy, float* dx, int M, int C, float eps, void* stream) {
    RGBD_REQUIRE(x && dy && dx && M > 0 && C > 0, "rgbd_pixelnorm_bwd: bad arguments");
    pixelnorm_kernel<<<M, 256, 0, (hipStream_t)stream>>>(x, dy, dx, C, eps);
    RGBD_CHECK_LAUNCH("pixelnorm_kernel");
    return 0;
}

extern "C" int rgbd_depth_head_fwd(const float* x, float* y, int B, int HW, void* stream) {
    RGBD_REQUIRE(x && y && B > 0 && HW > 0, "rgbd_depth_head_fwd: bad arguments");
    const long total = (long)B * 4 * HW;
    depth_head_kernel<<<(int)min((long)2048, (total + 255) / 256), 256, 0, (hipStream_t)stream>>>(x, nullptr, nullptr, y,
                                                                                                 HW, total);
    RGBD_CHECK_LAUNCH("depth_head_kernel");
    return 0;
}

extern "C" int rgbd_depth_head_bwd(const float* x, const float* y, const float* dy, float* dx, int B, int HW,
                                   void* stream) {
    RGBD_REQUIRE(x && y && dy && dx && B > 0 && HW > 0, "rgbd_depth_head_bwd: bad arguments");
    const long total = (long)B * 4 * HW;
    depth_head_kernel<<<(int)min((long)2048, (total + 255) / 256), 256, 0, (hipStream_t)stream>>>(x, y, dy, dx, HW, total);
    RGBD_CHECK_LAUNCH("depth_head_kernel");
    return 0;
}

extern "C" int rgbd_gan_logit_heads(const float* y, int n, float* losses, float* seed_neg, float* seed_pos, float* ratio,
                                    void* stream) {
    RGBD_REQUIRE(y && losses && seed_neg && seed_pos && ratio && n > 0, "rgbd_gan_logit_heads: bad arguments");
    gan_logit_heads_kernel<<<1, 256, 0, (hipStream_t)stream>>>(y, n, losses, seed_neg, seed_pos, ratio);
    RGBD_CHECK_LAUNCH("gan_logit_heads_kernel");
    return 0;
}

extern "C" int rgbd_softplus_mean(const float* y, int n, float sign, float gamma, float* loss, float* dy, void* stream) {
    RGBD_REQUIRE(y && loss && dy && n > 0, "rgbd_softplus_mean: bad arguments");
    RGBD_REQUIRE((sign == 1.f || sign == -1.f) && gamma >= 0.f, "rgbd_softplus_mean: sign must be +-1, gamma >= 0");
    softplus_mean_kernel<<<1, 256, 0, (hipStream_t)stream>>>(y, n, sign, gamma, loss, dy);
    RGBD_CHECK_LAUNCH("softplus_mean_kernel");
    return 0;
}

extern "C" int rgbd_ema_update(float* dst, const float* src, int64_t n, float tau, void* stream) {
    RGBD_REQUIRE(dst && src && n > 0, "rgbd_ema_update: bad arguments");
    ema_kernel<<<(int)min((long)2048, (long)((n + 255) / 256)), 256, 0, (hipStream_t)stream>>>(dst, src, n, tau);
    RGBD_CHECK_LAUNCH("ema_kernel");
    return 0;
}

extern "C" int rgbd_adam_clip_multi(float* p, float* g, float* m, float* v, int64_t n, int nseg,
                                    const int64_t* seg_begin, const float* seg_alpha, float beta1, float beta2,
                                    float eps, float clip, float grad_scale, int32_t* step, float* workspace,
                                    float* norm_out, void* stream) {
    RGBD_REQUIRE(p && g && m && v && workspace && seg_begin && seg_alpha && step, "rgbd_adam_clip_multi: null pointer");
    RGBD_REQUIRE(n > 0 && nseg > 0, "rgbd_adam_clip_multi: empty");
    RGBD_REQUIRE(seg_begin[0] == 0 && seg_begin[nseg] == n, "rgbd_adam_clip_multi: segments must cover [0,n)");
    hipStream_t st = (hipStream_t)stream;
    const int nb = (int)min((long)NORM_BLOCKS, (long)((n + 255) / 256));
    sumsq_kernel<<<nb, 256, 0, st>>>(g, n, grad_scale, workspace);
    RGBD_CHECK_LAUNCH("sumsq_kernel");
    norm_final_kernel<<<1, 256, 0, st>>>(workspace, nb, clip, beta1, beta2, step, norm_out);
    RGBD_CHECK_LAUNCH("norm_final_kernel");
    for (int s = 0; s < nseg; ++s) {
        const long b = seg_begin[s], e = seg_begin[s + 1];
        RGBD_REQUIRE(e >= b, "rgbd_adam_clip_multi: segments must be ascending");
        if (e == b) continue;
        const int blocks = (int)min((long)2048, (e - b + 255) / 256);
        adam_kernel<<<blocks, 256, 0, st>>>(p, g, m, v, b, e, seg_alpha[s], beta1, beta2, eps, grad_scale, workspace);
        RGBD_CHECK_LAUNCH("adam_kernel");
    }
    return 0;
}
