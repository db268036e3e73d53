// Fused 3D-consistency loss: depth warp + bilinear sampling + out-of-frame / occlusion masks + L1 terms.
//
// Replaces common/loss_functions.py:63-146,171-228 of the reference (about 100 small CuPy kernels and
// 8 scatter-adds per call) with one forward and one backward kernel.  HBM-bound and tiny (about 1.5 MB per
// view pair), so the goal is launch count and exact index math, not MFMA.
//
// THIS FILE IS COMPILED WITH -ffp-contract=off: the projection and interpolation arithmetic must be
// evaluated unfused, left to right, to be bit-exact against oracle/warp_loss.py:forward_np
// ("Bit-exactness definition").
#include "common.h"

namespace {

struct PixelWarp {
    float zp0, zp1, zp2;      // projected z * (x, y, 1)
    float den, u, v;          // u = row coordinate, v = column coordinate (reference swaps x/y)
    float u0f, u1f, v0f, v1f;
    float w1, w2, w3, w4;     // masked interpolation weights
    int u0m, v0m, v1m;        // masked taps (u1m == u0m: loss_functions.py:219)
    bool mask;
};

// coef: A(9) c(3); sign = -1 for warp (zp = A(z p) - c), +1 for inv_warp (zp = A'(z p) + c').
__device__ __forceinline__ PixelWarp project_pixel(const float* __restrict__ cf, float sign, float z,
                                                   int i, int j, int S) {
    PixelWarp w;
    const float p0 = (float)j, p1 = (float)i;
    const float a0 = z * p0, a1 = z * p1, a2 = z * 1.0f;
    float s0 = (cf[0] * a0 + cf[1] * a1) + cf[2] * a2;
    float s1 = (cf[3] * a0 + cf[4] * a1) + cf[5] * a2;
    float s2 = (cf[6] * a0 + cf[7] * a1) + cf[8] * a2;
    if (sign < 0.f) { s0 = s0 - cf[9]; s1 = s1 - cf[10]; s2 = s2 - cf[11]; }
    else            { s0 = s0 + cf[9]; s1 = s1 + cf[10]; s2 = s2 + cf[11]; }
    w.zp0 = s0; w.zp1 = s1; w.zp2 = s2;
    w.den = fminf(fmaxf(s2, 1e-4f), 10000.f);
    const float x = s0 / w.den;
    const float y = s1 / w.den;
    w.u = y; w.v = x;
    const int u0 = (int)w.u;   // truncation toward zero (saturating for out-of-range; such pixels are masked)
    const int v0 = (int)w.v;
    const int u1 = (int)((unsigned)u0 + 1u);
    const int v1 = (int)((unsigned)v0 + 1u);
    w.u0f = (float)u0; w.u1f = (float)u1; w.v0f = (float)v0; w.v1f = (float)v1;
    w.w1 = (w.u1f - w.u) * (w.v1f - w.v);
    w.w2 = (w.u - w.u0f) * (w.v1f - w.v);
    w.w3 = (w.u1f - w.u) * (w.v - w.v0f);
    w.w4 = (w.u - w.u0f) * (w.v - w.v0f);
    const float lim = (float)(S - 1);
    w.mask = (w.u >= 0.f) && (w.u < lim) && (w.v >= 0.f) && (w.v < lim) && (s2 > 1e-4f);
    const float mf = w.mask ? 1.f : 0.f;
    w.u0m = w.mask ? u0 : 0;
    w.v0m = w.mask ? v0 : 0;
    w.v1m = w.mask ? v1 : 0;
    w.w1 *= mf; w.w2 *= mf; w.w3 *= mf; w.w4 *= mf;
    return w;
}

// grid: (ceil(b*hw/256), 2). blockIdx.y = direction (0: sample img_rot at warp(img); 1: the inverse).
__global__ __launch_bounds__(256) void warp_loss_fwd_kernel(
    const float* __restrict__ img, const float* __restrict__ img_rot, const float* __restrict__ coef,
    int b, int S, int flags, float max_depth, float min_depth, float hinge_min,
    float* __restrict__ partials, float* __restrict__ dbg_zp, float* __restrict__ dbg_warped,
    int32_t* __restrict__ dbg_idx) {
    const int dir = blockIdx.y;
    const int hw = S * S;
    const long n = (long)blockIdx.x * 256 + threadIdx.x;
    const long N = (long)b * hw;
    const float* own = dir == 0 ? img : img_rot;   // image whose depth is projected / whose RGB is the target
    const float* src = dir == 0 ? img_rot : img;   // image that is sampled
    float l_rgb = 0.f, l_d = 0.f, l_h = 0.f;
    if (n < N) {
        const int bi = (int)(n / hw);
        const int pix = (int)(n - (long)bi * hw);
        const int i = pix / S, j = pix - i * S;
        const float* cf = coef + bi * 24 + dir * 12;
        const float* ob = own + (long)bi * 4 * hw;
        const float* sb = src + (long)bi * 4 * hw;
        const float z = ob[3 * hw + pix];
        {   // depth-range hinge (updater.py:357-359): the two directions' "own" images cover every pixel of the batch once
            const float hv = fmaxf(hinge_min - z, 0.f);
            l_h = hv * hv;
        }
        const PixelWarp w = project_pixel(cf, dir == 0 ? -1.f : 1.f, z, i, j, S);
        const int o00 = w.u0m * S + w.v0m;
        const int o01 = w.u0m * S + w.v1m;
        float warped[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float a = sb[c * hw + o00];
            const float d = sb[c * hw + o01];
            // taps in the reference's order: (u0,v0), (u1,v0), (u0,v1), (u1,v1) with u1 == u0
            warped[c] = ((w.w1 * a + w.w2 * a) + w.w3 * d) + w.w4 * d;
        }
        const float mf = w.mask ? 1.f : 0.f;
        float target[4];
#pragma unroll
        for (int c = 0; c < 3; ++c) target[c] = ob[c * hw + pix] * mf;
        target[3] = w.zp2 * mf;
        bool vis = true;
        if (flags & RGBD_WARP_OCCLUSION) vis = vis && (warped[3] > w.zp2);
        if (flags & RGBD_WARP_MAX_DEPTH) vis = vis && (z < max_depth);
        if (flags & RGBD_WARP_MIN_DEPTH) vis = vis && (z > min_depth);
        const float vf = vis ? 1.f : 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) l_rgb += fabsf(warped[c] * vf - target[c] * vf);
        l_d = fabsf(warped[3] * vf - target[3] * vf);
        if (dbg_zp) {
            float* o = dbg_zp + ((long)dir * N + n) * 3;
            o[0] = w.zp0; o[1] = w.zp1; o[2] = w.zp2;
        }
        if (dbg_warped) {
            float* o = dbg_warped + ((long)dir * N + n) * 4;
            o[0] = warped[0]; o[1] = warped[1]; o[2] = warped[2]; o[3] = warped[3];
        }
        if (dbg_idx) {
            int32_t* o = dbg_idx + ((long)dir * N + n) * 4;
            o[0] = w.u0m; o[1] = w.v0m; o[2] = w.v1m; o[3] = w.mask ? 1 : 0;
        }
    }
    // deterministic block reduction
    __shared__ float red[3][4];
    l_rgb = wave_sum(l_rgb);
    l_d = wave_sum(l_d);
    l_h = wave_sum(l_h);
    const int wid = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[0][wid] = l_rgb; red[1][wid] = l_d; red[2][wid] = l_h; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float* o = partials + ((long)blockIdx.x * 2 + dir) * 3;
        o[0] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        o[1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
        o[2] = (red[2][0] + red[2][1]) + (red[2][2] + red[2][3]);
    }
}

// one block: loss = mae_rgb(fwd) + mae_rgb(inv) + lambda * (mae_d(fwd) + mae_d(inv)) [+ hinge_lambda * mean hinge]
__global__ __launch_bounds__(256) void warp_loss_final_kernel(const float* __restrict__ partials, int nblocks,
                                                             float inv_n, float inv_n_rgb, float lambda_geo, float hinge_lambda,
                                                             float* __restrict__ loss) {
    float acc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};  // rgb0, d0, h0, rgb1, d1, h1
    for (int k = threadIdx.x; k < nblocks; k += 256) {
#pragma unroll
        for (int q = 0; q < 6; ++q) acc[q] += partials[(long)k * 6 + q];
    }
    __shared__ float red[6][4];
#pragma unroll
    for (int q = 0; q < 6; ++q) acc[q] = wave_sum(acc[q]);
    const int wid = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int q = 0; q < 6; ++q) red[q][wid] = acc[q];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float t[6];
#pragma unroll
        for (int q = 0; q < 6; ++q) t[q] = (red[q][0] + red[q][1]) + (red[q][2] + red[q][3]);
        const float rgb = t[0] * inv_n_rgb + t[3] * inv_n_rgb;      // inv_n_rgb = inv_n / (C - 1): mean over N x (C - 1) elements
        const float dep = t[1] * inv_n * lambda_geo + t[4] * inv_n * lambda_geo;
        float total = rgb + dep;
        if (hinge_lambda != 0.f) total = total + ((t[2] + t[5]) * (0.5f * inv_n)) * hinge_lambda;
        loss[0] = total;
    }
}

// The scatter of the backward pass, ORDER-INDEPENDENT: a tap's contribution is  g_c * w  with  g_c in {-k_c, 0, +k_c}
// (k_c = the constant L1 seed of channel c) and an interpolation weight w in [0,1], so what is accumulated is the
// signed weight in fixed point (2^-40 steps: exact for every fp32 weight >= 2^-17) with 64-bit INTEGER atomics --
// integer addition is associative, the sums are the same bits whatever the interleaving of the waves -- and the finish
// kernel multiplies by k_c once.  (Rounds 1-2 scattered with fp32 atomics: different from run to run, which hid the real
// fault of those rounds behind "atomic noise" -- packed-fp32 arithmetic going wrong next to another queue's MFMA waves,
// DESIGN.md section 3; the library is built without those instructions now.)  The own-pixel terms (target colours, projected
// depth, interpolation weights -> depth, depth hinge) belong to exactly one thread each: plain read-modify-writes.
#define RGBD_WARP_FIX_BITS 40
__device__ __forceinline__ void scatter_fix(long long* p, float signed_w) {
    const long long q = __double2ll_rn((double)signed_w * 1099511627776.0);   // 2^40; the product is exact
    if (q != 0) atomicAdd(reinterpret_cast<unsigned long long*>(p), (unsigned long long)q);
}

__global__ __launch_bounds__(256) void warp_loss_bwd_kernel(
    const float* __restrict__ img, const float* __restrict__ img_rot, const float* __restrict__ coef,
    int b, int S, int flags, float lambda_geo, float max_depth, float min_depth, float hinge_lambda, float hinge_min,
    const float* __restrict__ grad_loss, float grad_scale, float* __restrict__ gimg, float* __restrict__ gimg_rot,
    long long* __restrict__ acc) {
    const int dir = blockIdx.y;
    const int hw = S * S;
    const long n = (long)blockIdx.x * 256 + threadIdx.x;
    const long N = (long)b * hw;
    if (n >= N) return;
    const float* own = dir == 0 ? img : img_rot;
    const float* src = dir == 0 ? img_rot : img;
    float* gown = dir == 0 ? gimg : gimg_rot;
    const int bi = (int)(n / hw);
    const int pix = (int)(n - (long)bi * hw);
    const int i = pix / S, j = pix - i * S;
    const float* cf = coef + bi * 24 + dir * 12;
    const float* ob = own + (long)bi * 4 * hw;
    const float* sb = src + (long)bi * 4 * hw;
    float* gob = gown + (long)bi * 4 * hw;
    long long* asb = acc + ((long)(dir == 0 ? b : 0) + bi) * 4 * hw;    // acc: (2b,4,hw), images first, then rotated images
    const float z = ob[3 * hw + pix];
    const float go = (grad_loss ? grad_loss[0] : 1.f) * grad_scale;
    float gz_own = 0.f;
    if (hinge_lambda != 0.f && z < hinge_min)      // d/dz of hinge_lambda * mean_{2N} relu(hinge_min - z)^2
        gz_own = go * hinge_lambda * (0.5f / (float)N) * (-2.f * (hinge_min - z));
    const PixelWarp w = project_pixel(cf, dir == 0 ? -1.f : 1.f, z, i, j, S);
    bool vis = w.mask;   // masked pixels have zero weights, zero targets and constant taps: no gradient
    const int o00 = w.u0m * S + w.v0m;
    const int o01 = w.u0m * S + w.v1m;
    float a[4], d[4], warped[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        a[c] = sb[c * hw + o00];
        d[c] = sb[c * hw + o01];
        warped[c] = ((w.w1 * a[c] + w.w2 * a[c]) + w.w3 * d[c]) + w.w4 * d[c];
    }
    if (flags & RGBD_WARP_OCCLUSION) vis = vis && (warped[3] > w.zp2);
    if (flags & RGBD_WARP_MAX_DEPTH) vis = vis && (z < max_depth);
    if (flags & RGBD_WARP_MIN_DEPTH) vis = vis && (z > min_depth);
    if (!vis) {
        if (gz_own != 0.f) gob[3 * hw + pix] += gz_own;
        return;
    }
    const float inv_n = 1.f / (float)N;
    const float k_rgb = go * inv_n / 3.f;
    const float k_d = go * inv_n * lambda_geo;
    float sg[4], g[4];   // sign of the L1 term's derivative, and the derivative
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float diff = warped[c] - ob[c * hw + pix];
        sg[c] = diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f);
        g[c] = sg[c] * k_rgb;
    }
    {
        const float diff = warped[3] - w.zp2;
        sg[3] = diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f);
        g[3] = sg[3] * k_d;
    }
    // (1) gathered values -> scatter into the sampled image (two distinct taps; rows u0 and "u1" coincide)
    const float wl = w.w1 + w.w2, wr = w.w3 + w.w4;
    float gw_a = 0.f, gw_d = 0.f;  // sum_c g[c] * tap value
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        scatter_fix(asb + c * hw + o00, sg[c] * wl);
        scatter_fix(asb + c * hw + o01, sg[c] * wr);
        gw_a += g[c] * a[c];
        gw_d += g[c] * d[c];
    }
    // (2) targets: own RGB and projected depth
#pragma unroll
    for (int c = 0; c < 3; ++c)
        if (sg[c] != 0.f) gob[c * hw + pix] -= g[c];
    float gzp2 = -g[3];
    // (3) interpolation weights -> (u, v) -> zp -> own depth.  dL/dw1 = dL/dw2 = gw_a, dL/dw3 = dL/dw4 = gw_d.
    const float du1 = w.u1f - w.u, du0 = w.u - w.u0f, dv1 = w.v1f - w.v, dv0 = w.v - w.v0f;
    const float gu = (gw_a * (-dv1) + gw_a * dv1) + (gw_d * (-dv0) + gw_d * dv0);   // cancels exactly
    const float gv = (gw_a * (-du1) + gw_a * (-du0)) + (gw_d * du1 + gw_d * du0);
    const float gzp1 = gu / w.den;
    const float gzp0 = gv / w.den;
    const float gden = -(gu * w.u + gv * w.v) / w.den;
    if (w.zp2 >= 1e-4f && w.zp2 <= 10000.f) gzp2 += gden;
    const float p0 = (float)j, p1 = (float)i;
    const float gz = gzp0 * (cf[0] * p0 + cf[1] * p1 + cf[2]) + gzp1 * (cf[3] * p0 + cf[4] * p1 + cf[5]) +
                     gzp2 * (cf[6] * p0 + cf[7] * p1 + cf[8]);
    gob[3 * hw + pix] += gz_own + gz;
}

// grad[(image, c, pixel)] += k_c * acc * 2^-40 : the scattered sums, converted once.  Four pixels per thread.
__global__ __launch_bounds__(256) void warp_loss_bwd_finish_kernel(
    const long long* __restrict__ acc, int b, int hw, float lambda_geo, const float* __restrict__ grad_loss,
    float grad_scale, float* __restrict__ gimg, float* __restrict__ gimg_rot) {
    const long N = (long)b * hw;
    const long e = ((long)blockIdx.x * 256 + threadIdx.x) * 4;        // element of the (2b,4,hw) accumulator
    if (e >= 8 * N) return;
    const long img_i = e / (4 * (long)hw);
    const int c = (int)((e / hw) & 3);
    const float go = (grad_loss ? grad_loss[0] : 1.f) * grad_scale;
    const float inv_n = 1.f / (float)N;
    const float k = c < 3 ? go * inv_n / 3.f : go * inv_n * lambda_geo;
    float* g = (img_i < b ? gimg : gimg_rot - 4 * N) + e;
    const long long* a = acc + e;
    const long long q0 = a[0], q1 = a[1], q2 = a[2], q3 = a[3];
    if ((q0 | q1 | q2 | q3) == 0) return;
    f32x4 v = *reinterpret_cast<f32x4*>(g);
    v[0] += k * ((float)((double)q0 * 9.094947017729282e-13));
    v[1] += k * ((float)((double)q1 * 9.094947017729282e-13));
    v[2] += k * ((float)((double)q2 * 9.094947017729282e-13));
    v[3] += k * ((float)((double)q3 * 9.094947017729282e-13));
    *reinterpret_cast<f32x4*>(g) = v;
}

// ---- any number of channels (the last one is the depth), L1 or L2 criterion: loss_functions.py:137-145 with norm="l2" and
//      the 257-channel feature maps of updater.py:345-354.  Same projection / taps / masks as above; a thread loops over the
//      channels of its pixel.  The backward's scatter is ORDER-INDEPENDENT like the RGB-D kernels': what is accumulated per
//      tap is q * w with q = sign(warped - target) (L1) or warped - target itself (L2: the seed 2 k (warped - target) is
//      not a per-channel constant, but k is), as a 64-bit integer in units of 2^-40 (L1) / 2^-32 (L2: differences up to 2^31,
//      resolution 2.3e-10) by integer atomics into a workspace, and multiplied by the channel's constant once, in the finish
//      kernel.  Off the training step of every shipped config (`rotate_feature`).
__global__ __launch_bounds__(256) void warp_loss_nc_fwd_kernel(
    const float* __restrict__ img, const float* __restrict__ img_rot, const float* __restrict__ coef,
    int b, int C, int S, int flags, int l2, float max_depth, float min_depth, float* __restrict__ partials) {
    const int dir = blockIdx.y;
    const int hw = S * S;
    const long n = (long)blockIdx.x * 256 + threadIdx.x;
    const long N = (long)b * hw;
    const float* own = dir == 0 ? img : img_rot;
    const float* src = dir == 0 ? img_rot : img;
    float l_rgb = 0.f, l_d = 0.f;
    if (n < N) {
        const int bi = (int)(n / hw);
        const int pix = (int)(n - (long)bi * hw);
        const int i = pix / S, j = pix - i * S;
        const float* cf = coef + bi * 24 + dir * 12;
        const float* ob = own + (long)bi * C * hw;
        const float* sb = src + (long)bi * C * hw;
        const float z = ob[(long)(C - 1) * hw + pix];
        const PixelWarp w = project_pixel(cf, dir == 0 ? -1.f : 1.f, z, i, j, S);
        const int o00 = w.u0m * S + w.v0m, o01 = w.u0m * S + w.v1m;
        const float mf = w.mask ? 1.f : 0.f;
        auto sample = [&](int c) {
            const float a = sb[(long)c * hw + o00], d = sb[(long)c * hw + o01];
            return ((w.w1 * a + w.w2 * a) + w.w3 * d) + w.w4 * d;
        };
        const float wd = sample(C - 1);
        bool vis = true;
        if (flags & RGBD_WARP_OCCLUSION) vis = vis && (wd > w.zp2);
        if (flags & RGBD_WARP_MAX_DEPTH) vis = vis && (z < max_depth);
        if (flags & RGBD_WARP_MIN_DEPTH) vis = vis && (z > min_depth);
        const float vf = vis ? 1.f : 0.f;
        for (int c = 0; c < C - 1; ++c) {
            const float diff = sample(c) * vf - (ob[(long)c * hw + pix] * mf) * vf;
            l_rgb += l2 ? diff * diff : fabsf(diff);
        }
        const float dd = wd * vf - (w.zp2 * mf) * vf;
        l_d = l2 ? dd * dd : fabsf(dd);
    }
    __shared__ float red[2][4];
    l_rgb = wave_sum(l_rgb);
    l_d = wave_sum(l_d);
    const int wid = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[0][wid] = l_rgb; red[1][wid] = l_d; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float* o = partials + ((long)blockIdx.x * 2 + dir) * 3;
        o[0] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        o[1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
        o[2] = 0.f;
    }
}

__global__ __launch_bounds__(256) void warp_loss_nc_bwd_kernel(
    const float* __restrict__ img, const float* __restrict__ img_rot, const float* __restrict__ coef,
    int b, int C, int S, int flags, int l2, float lambda_geo, float max_depth, float min_depth,
    const float* __restrict__ grad_loss, float grad_scale, float* __restrict__ gimg, float* __restrict__ gimg_rot,
    long long* __restrict__ acc) {
    const int dir = blockIdx.y;
    const int hw = S * S;
    const long n = (long)blockIdx.x * 256 + threadIdx.x;
    const long N = (long)b * hw;
    if (n >= N) return;
    const float* own = dir == 0 ? img : img_rot;
    const float* src = dir == 0 ? img_rot : img;
    float* gown = dir == 0 ? gimg : gimg_rot;
    const int bi = (int)(n / hw);
    const int pix = (int)(n - (long)bi * hw);
    const int i = pix / S, j = pix - i * S;
    const float* cf = coef + bi * 24 + dir * 12;
    const float* ob = own + (long)bi * C * hw;
    const float* sb = src + (long)bi * C * hw;
    float* gob = gown + (long)bi * C * hw;
    long long* asb = acc + ((long)(dir == 0 ? b : 0) + bi) * C * hw;      // acc: (2b,C,hw), images first, then rotated images
    const float z = ob[(long)(C - 1) * hw + pix];
    const PixelWarp w = project_pixel(cf, dir == 0 ? -1.f : 1.f, z, i, j, S);
    if (!w.mask) return;
    const int o00 = w.u0m * S + w.v0m, o01 = w.u0m * S + w.v1m;
    const float wl = w.w1 + w.w2, wr = w.w3 + w.w4;
    const float a3 = sb[(long)(C - 1) * hw + o00], d3 = sb[(long)(C - 1) * hw + o01];
    const float wd = ((w.w1 * a3 + w.w2 * a3) + w.w3 * d3) + w.w4 * d3;
    bool vis = true;
    if (flags & RGBD_WARP_OCCLUSION) vis = vis && (wd > w.zp2);
    if (flags & RGBD_WARP_MAX_DEPTH) vis = vis && (z < max_depth);
    if (flags & RGBD_WARP_MIN_DEPTH) vis = vis && (z > min_depth);
    if (!vis) return;
    const float go = (grad_loss ? grad_loss[0] : 1.f) * grad_scale;
    const float k_rgb = go / ((float)N * (float)(C - 1));
    const float k_d = go * lambda_geo / (float)N;
    // q: the per-pixel factor of the seed (sign or difference); seed = (l2 ? 2 : 1) * k * q
    auto factor = [&](float diff) { return l2 ? diff : (diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f)); };
    const double unit = l2 ? 4294967296.0 : 1099511627776.0;            // 2^32 / 2^40
    const float kk = l2 ? 2.f : 1.f;
    auto scatter = [&](long long* p, float v) {
        const long long q = __double2ll_rn((double)v * unit);
        if (q != 0) atomicAdd(reinterpret_cast<unsigned long long*>(p), (unsigned long long)q);
    };
    float gw_a = 0.f, gw_d = 0.f;
    for (int c = 0; c < C - 1; ++c) {
        const float a = sb[(long)c * hw + o00], d = sb[(long)c * hw + o01];
        const float wv = ((w.w1 * a + w.w2 * a) + w.w3 * d) + w.w4 * d;
        const float q = factor(wv - ob[(long)c * hw + pix]);
        const float g = kk * k_rgb * q;
        if (q != 0.f) {
            scatter(asb + (long)c * hw + o00, q * wl);
            scatter(asb + (long)c * hw + o01, q * wr);
            gob[(long)c * hw + pix] -= g;                                // own pixel: this thread alone
        }
        gw_a += g * a;
        gw_d += g * d;
    }
    const float q3 = factor(wd - w.zp2);
    const float g3 = kk * k_d * q3;
    scatter(asb + (long)(C - 1) * hw + o00, q3 * wl);
    scatter(asb + (long)(C - 1) * hw + o01, q3 * wr);
    gw_a += g3 * a3;
    gw_d += g3 * d3;
    float gzp2 = -g3;
    const float du1 = w.u1f - w.u, du0 = w.u - w.u0f, dv1 = w.v1f - w.v, dv0 = w.v - w.v0f;
    const float gu = (gw_a * (-dv1) + gw_a * dv1) + (gw_d * (-dv0) + gw_d * dv0);
    const float gv = (gw_a * (-du1) + gw_a * (-du0)) + (gw_d * du1 + gw_d * du0);
    const float gzp1 = gu / w.den, gzp0 = gv / w.den;
    const float gden = -(gu * w.u + gv * w.v) / w.den;
    if (w.zp2 >= 1e-4f && w.zp2 <= 10000.f) gzp2 += gden;
    const float p0 = (float)j, p1 = (float)i;
    const float gz = gzp0 * (cf[0] * p0 + cf[1] * p1 + cf[2]) + gzp1 * (cf[3] * p0 + cf[4] * p1 + cf[5]) +
                     gzp2 * (cf[6] * p0 + cf[7] * p1 + cf[8]);
    gob[(long)(C - 1) * hw + pix] += gz;
}

// grad[(image, c, pixel)] += (l2 ? 2 : 1) * k_c * acc * unit: the scattered sums, converted once
__global__ __launch_bounds__(256) void warp_loss_nc_bwd_finish_kernel(
    const long long* __restrict__ acc, int b, int C, int hw, int l2, float lambda_geo, const float* __restrict__ grad_loss,
    float grad_scale, float* __restrict__ gimg, float* __restrict__ gimg_rot) {
    const long N = (long)b * hw;
    const long total = 2 * N * C;
    const float go = (grad_loss ? grad_loss[0] : 1.f) * grad_scale;
    const float kk = l2 ? 2.f : 1.f;
    const double unit = l2 ? 2.3283064365386963e-10 : 9.094947017729282e-13;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const long long q = acc[e];
        if (q == 0) continue;
        const int c = (int)((e / hw) % C);
        const float k = c < C - 1 ? go / ((float)N * (float)(C - 1)) : go * lambda_geo / (float)N;
        float* g = e < N * C ? gimg + e : gimg_rot + (e - N * C);
        *g += kk * k * (float)((double)q * unit);
    }
}

}  // namespace

extern "C" int rgbd_warp_loss_fwd(const float* img, const float* img_rot, const float* coef, int b, int S,
                                  int flags, float lambda_geometric, float max_depth, float min_depth,
                                  float hinge_lambda, float hinge_min,
                                  float* partials, float* loss, float* dbg_zp, float* dbg_warped,
                                  int32_t* dbg_idx, void* stream) {
    RGBD_REQUIRE(img && img_rot && coef && partials && loss, "rgbd_warp_loss_fwd: null pointer");
    RGBD_REQUIRE(b > 0 && S >= 2, "rgbd_warp_loss_fwd: bad shape b=%d S=%d", b, S);
    const long N = (long)b * S * S;
    const int nblocks = ceil_div(N, 256);
    hipStream_t st = (hipStream_t)stream;
    warp_loss_fwd_kernel<<<dim3(nblocks, 2), 256, 0, st>>>(img, img_rot, coef, b, S, flags, max_depth, min_depth,
                                                           hinge_min, partials, dbg_zp, dbg_warped, dbg_idx);
    RGBD_CHECK_LAUNCH("warp_loss_fwd_kernel");
    warp_loss_final_kernel<<<1, 256, 0, st>>>(partials, nblocks, 1.f / (float)N, (1.f / (float)N) / 3.f, lambda_geometric,
                                              hinge_lambda, loss);
    RGBD_CHECK_LAUNCH("warp_loss_final_kernel");
    return 0;
}

extern "C" int64_t rgbd_warp_loss_bwd_workspace(int b, int S) {
    return b > 0 && S > 0 ? (int64_t)2 * b * 4 * S * S * (int64_t)sizeof(long long) : 0;
}

extern "C" int rgbd_warp_loss_bwd(const float* img, const float* img_rot, const float* coef, int b, int S,
                                  int flags, float lambda_geometric, float max_depth, float min_depth,
                                  float hinge_lambda, float hinge_min,
                                  const float* grad_loss, float grad_scale, float* grad_img, float* grad_img_rot,
                                  int accumulate, void* workspace, void* stream) {
    RGBD_REQUIRE(img && img_rot && coef && grad_img && grad_img_rot && workspace, "rgbd_warp_loss_bwd: null pointer");
    RGBD_REQUIRE(b > 0 && S >= 2 && (S * S) % 4 == 0, "rgbd_warp_loss_bwd: bad shape b=%d S=%d", b, S);
    RGBD_REQUIRE(((uintptr_t)workspace & 15) == 0 && ((uintptr_t)grad_img & 15) == 0 && ((uintptr_t)grad_img_rot & 15) == 0,
                 "rgbd_warp_loss_bwd: workspace and gradient buffers must be 16-byte aligned");
    const long N = (long)b * S * S;
    hipStream_t st = (hipStream_t)stream;
    long long* acc = static_cast<long long*>(workspace);
    bool ok = rgbd_zero_async(acc, (size_t)rgbd_warp_loss_bwd_workspace(b, S), st) == hipSuccess;
    if (!accumulate)
        ok = ok && rgbd_zero_async(grad_img, N * 4 * sizeof(float), st) == hipSuccess &&
             rgbd_zero_async(grad_img_rot, N * 4 * sizeof(float), st) == hipSuccess;
    if (!ok) {
        rgbd_set_error("rgbd_warp_loss_bwd: clearing the accumulators failed");
        return -2;
    }
    warp_loss_bwd_kernel<<<dim3(ceil_div(N, 256), 2), 256, 0, st>>>(img, img_rot, coef, b, S, flags, lambda_geometric,
                                                                    max_depth, min_depth, hinge_lambda, hinge_min,
                                                                    grad_loss, grad_scale, grad_img, grad_img_rot, acc);
    RGBD_CHECK_LAUNCH("warp_loss_bwd_kernel");
    warp_loss_bwd_finish_kernel<<<ceil_div(2 * N, 256), 256, 0, st>>>(acc, b, S * S, lambda_geometric, grad_loss,
                                                                      grad_scale, grad_img, grad_img_rot);
    RGBD_CHECK_LAUNCH("warp_loss_bwd_finish_kernel");
    return 0;
}

extern "C" int rgbd_warp_loss_nc_fwd(const float* img, const float* img_rot, const float* coef, int b, int C, int S, int flags,
                                     int norm_l2, float lambda_geometric, float max_depth, float min_depth,
                                     float* partials, float* loss, void* stream) {
    RGBD_REQUIRE(img && img_rot && coef && partials && loss, "rgbd_warp_loss_nc_fwd: null pointer");
    RGBD_REQUIRE(b > 0 && S >= 2 && C >= 2, "rgbd_warp_loss_nc_fwd: bad shape b=%d C=%d S=%d", b, C, S);
    const long N = (long)b * S * S;
    const int nblocks = ceil_div(N, 256);
    hipStream_t st = (hipStream_t)stream;
    warp_loss_nc_fwd_kernel<<<dim3(nblocks, 2), 256, 0, st>>>(img, img_rot, coef, b, C, S, flags, norm_l2 ? 1 : 0, max_depth,
                                                              min_depth, partials);
    RGBD_CHECK_LAUNCH("warp_loss_nc_fwd_kernel");
    warp_loss_final_kernel<<<1, 256, 0, st>>>(partials, nblocks, 1.f / (float)N, 1.f / ((float)N * (float)(C - 1)),
                                              lambda_geometric, 0.f, loss);
    RGBD_CHECK_LAUNCH("warp_loss_final_kernel");
    return 0;
}

extern "C" int64_t rgbd_warp_loss_nc_bwd_workspace(int b, int C, int S) {
    return b > 0 && C > 0 && S > 0 ? (int64_t)2 * b * C * S * S * (int64_t)sizeof(long long) : 0;
}

extern "C" int rgbd_warp_loss_nc_bwd(const float* img, const float* img_rot, const float* coef, int b, int C, int S, int flags,
                                     int norm_l2, float lambda_geometric, float max_depth, float min_depth,
                                     const float* grad_loss, float grad_scale, float* grad_img, float* grad_img_rot,
                                     int accumulate, void* workspace, void* stream) {
    RGBD_REQUIRE(img && img_rot && coef && grad_img && grad_img_rot && workspace, "rgbd_warp_loss_nc_bwd: null pointer");
    RGBD_REQUIRE(b > 0 && S >= 2 && C >= 2, "rgbd_warp_loss_nc_bwd: bad shape b=%d C=%d S=%d", b, C, S);
    RGBD_REQUIRE(((uintptr_t)workspace & 7) == 0, "rgbd_warp_loss_nc_bwd: the workspace must be 8-byte aligned");
    const long N = (long)b * S * S;
    hipStream_t st = (hipStream_t)stream;
    long long* acc = static_cast<long long*>(workspace);
    bool ok = rgbd_zero_async(acc, (size_t)rgbd_warp_loss_nc_bwd_workspace(b, C, S), st) == hipSuccess;
    if (!accumulate)
        ok = ok && rgbd_zero_async(grad_img, (size_t)N * C * sizeof(float), st) == hipSuccess &&
             rgbd_zero_async(grad_img_rot, (size_t)N * C * sizeof(float), st) == hipSuccess;
    if (!ok) {
        rgbd_set_error("rgbd_warp_loss_nc_bwd: clearing the accumulators failed");
        return -2;
    }
    warp_loss_nc_bwd_kernel<<<dim3(ceil_div(N, 256), 2), 256, 0, st>>>(img, img_rot, coef, b, C, S, flags, norm_l2 ? 1 : 0,
                                                                       lambda_geometric, max_depth, min_depth, grad_loss,
                                                                       grad_scale, grad_img, grad_img_rot, acc);
    RGBD_CHECK_LAUNCH("warp_loss_nc_bwd_kernel");
    const long total = 2 * N * C;
    warp_loss_nc_bwd_finish_kernel<<<(unsigned)(ceil_div(total, 256) < 4096 ? ceil_div(total, 256) : 4096), 256, 0, st>>>(
        acc, b, C, S * S, norm_l2 ? 1 : 0, lambda_geometric, grad_loss, grad_scale, grad_img, grad_img_rot);
    RGBD_CHECK_LAUNCH("warp_loss_nc_bwd_finish_kernel");
    return 0;
}
