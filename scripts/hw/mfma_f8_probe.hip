// Hardware probe for the block-scaled fp8 MFMA of gfx950 (v_mfma_scale_f32_16x16x128_f8f6f4, e4m3 operands):
//   1. which (row, k) does byte j of lane l's A / B operand hold, and which element does accumulator register r hold;
//   2. which (row, 32-wide k block) does lane l's E8M0 scale byte apply to, and what do the opsel immediates select;
//   3. what v_cvt_pk_fp8_f32 does at the edges (round to nearest even, saturation, NaN, subnormals);
//   4. how fast the instruction issues next to v_mfma_f32_16x16x32_bf16 (same wave structure).
// The guides give no operand map for this instruction ("check the map with exact integer data before relying on it").
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 scripts/hw/mfma_f8_probe.hip -o /tmp/f8probe && /tmp/f8probe
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

// ---- host-side e4m3fn (OCP) encode / decode: RNE, saturating to +-448, NaN -> 0x7f
static float e4m3_decode(unsigned char v) {
    const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
    float r;
    if (e == 15 && m == 7) return NAN;
    if (e == 0) r = ldexpf((float)m, -9);
    else r = ldexpf(1.f + m / 8.f, e - 7);
    return s ? -r : r;
}
static unsigned char e4m3_encode(float f) {
    if (isnan(f)) return 0x7f;
    const unsigned char s = signbit(f) ? 0x80 : 0;
    float a = fabsf(f);
    if (a >= 448.f) return s | 0x7e;
    // candidates: nearest of the 127 finite magnitudes (monotone in the code)
    int lo = 0, hi = 0x7e;
    while (hi - lo > 1) {
        const int mid = (lo + hi) / 2;
        if (e4m3_decode((unsigned char)mid) <= a) lo = mid; else hi = mid;
    }
    const float dl = a - e4m3_decode((unsigned char)lo), dh = e4m3_decode((unsigned char)hi) - a;
    int pick = dl < dh ? lo : (dh < dl ? hi : ((lo & 1) ? hi : lo));
    return s | (unsigned char)pick;
}

// ---- 1 + 2: one MFMA, operands gathered through host-provided maps
// a_k[l*32 + j] = k index of byte j of lane l's A operand (row = l & 15); likewise b_k (col = l & 15)
// sa_blk[l] = which k block's scale lane l supplies for A (row l & 15); byte position `pos` of the scale dword holds it,
// the other three bytes hold a poison scale (so a wrong opsel shows)
template <int OPA, int OPB>
__global__ void mfma_probe(const unsigned char* A, const unsigned char* B, const unsigned char* SA, const unsigned char* SB,
                           const int* a_k, const int* b_k, const int* sa_blk, const int* sb_blk, float* D) {
    const int l = threadIdx.x;
    unsigned char ab[32], bb[32];
    for (int j = 0; j < 32; ++j) {
        ab[j] = A[(l & 15) * 128 + a_k[l * 32 + j]];
        bb[j] = B[b_k[l * 32 + j] * 16 + (l & 15)];
    }
    i32x8 av, bv;
    memcpy(&av, ab, 32);
    memcpy(&bv, bb, 32);
    unsigned sa = 0x85858585u, sb = 0x85858585u;             // poison: 2^6
    sa = (sa & ~(0xffu << (8 * OPA))) | ((unsigned)SA[(l & 15) * 4 + sa_blk[l]] << (8 * OPA));
    sb = (sb & ~(0xffu << (8 * OPB))) | ((unsigned)SB[(l & 15) * 4 + sb_blk[l]] << (8 * OPB));
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, c, 0, 0, OPA, (int)sa, OPB, (int)sb);
    for (int r = 0; r < 4; ++r) D[l * 4 + r] = c[r];
}

// ---- 3: conversions
__global__ void cvt_probe(const float* x, unsigned char* out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const int p = __builtin_amdgcn_cvt_pk_fp8_f32(x[i], 0.f, 0, false);
        out[i] = (unsigned char)(p & 0xff);
    }
}

// ---- 4: issue rate.  Every wave: NACC independent accumulators, ITERS rounds, operands in registers.
template <int KIND>   // 0: bf16 16x16x32, 1: f8 scaled 16x16x128 (unit scales), 2: f8 scaled with varying scale registers
__global__ __launch_bounds__(512) void rate_kernel(float* out, int iters, int seed) {
    const int l = threadIdx.x & 63;
    i32x8 a, b;
    for (int j = 0; j < 8; ++j) {
        a[j] = (int)(0x38383838u ^ (unsigned)((l * 7 + j * 13 + seed) & 0x07070707));     // ~1.0 .. 1.9, exact small values
        b[j] = (int)(0x30303030u ^ (unsigned)((l * 5 + j * 3 + seed) & 0x07070707));
    }
    bf16x8 ah, bh;
    for (int j = 0; j < 8; ++j) { ah[j] = (short)(0x3f80 + ((l + j) & 7)); bh[j] = (short)(0x3e80 + ((l * 3 + j) & 7)); }
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    int s0 = 0x7f7f7f7f, s1 = KIND == 2 ? 0x7e7f807f + (l & 1) : 0x7f7f7f7f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (KIND == 0) {
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc[i], 0, 0, 0);
            } else {
                acc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc[i], 0, 0, 0, s0, 0, s1);
            }
        }
    }
    float t = 0.f;
    for (int i = 0; i < 16; ++i) t += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = t;
}

template <typename T>
static T* dev(const std::vector<T>& h) {
    T* p;
    hipMalloc(&p, h.size() * sizeof(T));
    hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice);
    return p;
}

template <int OPA, int OPB>
static double run_layout(const std::vector<int>& ak, const std::vector<int>& bk, const std::vector<int>& sablk,
                         const std::vector<int>& sbblk, bool unit_scales, int dmap, const char* name) {
    std::vector<unsigned char> A(16 * 128), B(128 * 16), SA(16 * 4), SB(16 * 4);
    srand(7);
    for (auto& v : A) v = e4m3_encode((float)((rand() % 17) - 8) * 0.25f);         // exact in e4m3
    for (auto& v : B) v = e4m3_encode((float)((rand() % 13) - 6) * 0.5f);
    for (int i = 0; i < 64; ++i) {
        SA[i] = unit_scales ? 127 : (unsigned char)(124 + (i * 5) % 7);
        SB[i] = unit_scales ? 127 : (unsigned char)(125 + (i * 3) % 5);
    }
    std::vector<float> D(256), ref(256, 0.f);
    unsigned char *dA = dev(A), *dB = dev(B), *dSA = dev(SA), *dSB = dev(SB);
    int *dak = dev(ak), *dbk = dev(bk), *dsa = dev(sablk), *dsb = dev(sbblk);
    float* dD;
    hipMalloc(&dD, 256 * 4);
    mfma_probe<OPA, OPB><<<1, 64>>>(dA, dB, dSA, dSB, dak, dbk, dsa, dsb, dD);
    hipMemcpy(D.data(), dD, 256 * 4, hipMemcpyDeviceToHost);
    for (int m = 0; m < 16; ++m)
        for (int n = 0; n < 16; ++n) {
            double s = 0;
            for (int k = 0; k < 128; ++k)
                s += (double)e4m3_decode(A[m * 128 + k]) * ldexp(1.0, SA[m * 4 + k / 32] - 127) *
                     (double)e4m3_decode(B[k * 16 + n]) * ldexp(1.0, SB[n * 4 + k / 32] - 127);
            ref[m * 16 + n] = (float)s;
        }
    double worst = 0;
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 4; ++r) {
            // dmap 0: col = l & 15, row = 4 (l >> 4) + r  (the bf16 16x16 map, A = rows);  dmap 1: transposed
            const int row = dmap == 0 ? 4 * (l >> 4) + r : (l & 15), col = dmap == 0 ? (l & 15) : 4 * (l >> 4) + r;
            worst = fmax(worst, fabs((double)D[l * 4 + r] - ref[row * 16 + col]));
        }
    printf("%-58s max |D - ref| = %g %s\n", name, worst, worst == 0 ? "EXACT" : "");
    return worst;
}

int main() {
    // ---------------- operand maps
    std::vector<int> contiguous(64 * 32), split16(64 * 32), blk(64), blk0(64, 0);
    for (int l = 0; l < 64; ++l) {
        for (int j = 0; j < 32; ++j) {
            contiguous[l * 32 + j] = 32 * (l >> 4) + j;                                          // H1: k = 32 (l>>4) + j
            split16[l * 32 + j] = j < 16 ? 16 * (l >> 4) + j : 64 + 16 * (l >> 4) + (j - 16);    // H2: two 64-wide halves
        }
        blk[l] = l >> 4;
    }
    printf("== 1. operand layout (unit scales)\n");
    run_layout<0, 0>(contiguous, contiguous, blk, blk, true, 0, "H1 k = 32 (lane>>4) + j, D as bf16 16x16 map");
    run_layout<0, 0>(contiguous, contiguous, blk, blk, true, 1, "H1, D transposed");
    run_layout<0, 0>(split16, split16, blk, blk, true, 0, "H2 k = {16 q + j, 64 + 16 q + j}");
    printf("   (any consistent k permutation of A and B gives the same sums: H2 EXACT only says the sum is over all k)\n");
    run_layout<0, 0>(contiguous, split16, blk, blk, true, 0, "A by H1, B by H2 (must FAIL)");
    printf("== 2. scales.  The instruction's own K order is 'H2' (byte j of lane l: k = 16 (l>>4) + j, then 64 + 16 (l>>4) + j - 16):\n"
           "      lane l supplies the scale of (row l & 15, K block l >> 4) of THAT order; operands placed by H2 below\n");
    run_layout<0, 0>(split16, split16, blk, blk, false, 0, "opsel 0/0, scale in byte 0");
    run_layout<1, 2>(split16, split16, blk, blk, false, 0, "opsel 1/2, scale in byte 1 / 2");
    run_layout<3, 3>(split16, split16, blk, blk, false, 0, "opsel 3/3, scale in byte 3");
    run_layout<0, 0>(split16, split16, blk0, blk0, false, 0, "every lane supplies block 0's scale (must FAIL)");
    run_layout<0, 0>(contiguous, contiguous, blk, blk, false, 0, "operands by H1 with these block scales (must FAIL)");

    // ---------------- 3. conversions
    printf("== 3. v_cvt_pk_fp8_f32 against round-to-nearest-even, saturating\n");
    {
        std::vector<float> x;
        for (int c = 0; c < 0x7f; ++c) {            // every finite magnitude, the midpoints and a bit either side
            const float a = e4m3_decode((unsigned char)c), b = c + 1 < 0x7f ? e4m3_decode((unsigned char)(c + 1)) : 480.f;
            x.push_back(a); x.push_back(-a);
            x.push_back(0.5f * (a + b)); x.push_back(-0.5f * (a + b));
            x.push_back(nextafterf(0.5f * (a + b), 0.f)); x.push_back(nextafterf(0.5f * (a + b), 1e9f));
        }
        const float extra[] = {448.f, 449.f, 463.9f, 464.f, 464.1f, 480.f, 511.f, 512.f, 1000.f, 1e30f, INFINITY, -INFINITY, NAN,
                               0.f, -0.f, 1e-10f, 0.0009765625f, 0.00097656256f, 0.001953125f};
        for (float e : extra) x.push_back(e);
        float* dx = dev(x);
        unsigned char* dout;
        hipMalloc(&dout, x.size());
        cvt_probe<<<(int)(x.size() + 255) / 256, 256>>>(dx, dout, (int)x.size());
        std::vector<unsigned char> o(x.size());
        hipMemcpy(o.data(), dout, x.size(), hipMemcpyDeviceToHost);
        int bad = 0;
        for (size_t i = 0; i < x.size(); ++i) {
            const unsigned char want = e4m3_encode(x[i]);
            const bool same = o[i] == want || (isnan(x[i]) && (o[i] & 0x7f) == 0x7f);
            if (!same && bad < 24) printf("   x = %.9g (%a): hardware 0x%02x (%g), RNE-saturating 0x%02x (%g)\n", x[i], x[i], o[i],
                                          e4m3_decode(o[i]), want, e4m3_decode(want));
            bad += !same;
        }
        printf("   %d of %zu values differ from RNE + saturation at 448\n", bad, x.size());
    }

    // ---------------- 4. issue rate
    printf("== 4. issue rate, 256 workgroups x 512 threads (2 waves per SIMD) and x 256 threads (1 per SIMD)\n");
    {
        float* dout;
        hipMalloc(&dout, 1024 * 512 * 4);
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        const int iters = 2000;
        for (int threads = 512; threads >= 256; threads -= 256)
            for (int kind = 0; kind < 3; ++kind) {
                float ms = 0;
                for (int rep = 0; rep < 3; ++rep) {
                    hipEventRecord(e0);
                    if (kind == 0) rate_kernel<0><<<256, threads>>>(dout, iters, rep);
                    if (kind == 1) rate_kernel<1><<<256, threads>>>(dout, iters, rep);
                    if (kind == 2) rate_kernel<2><<<256, threads>>>(dout, iters, rep);
                    hipEventRecord(e1);
                    hipEventSynchronize(e1);
                    hipEventElapsedTime(&ms, e0, e1);
                }
                const double mfmas = 256.0 * (threads / 64) * iters * 16;
                const double flop = mfmas * 2.0 * 16 * 16 * (kind == 0 ? 32 : 128);
                printf("   %s, %d threads: %.3f ms, %.0f TFLOP/s, %.1f ns per MFMA per SIMD\n",
                       kind == 0 ? "bf16 16x16x32        " : kind == 1 ? "f8 16x16x128 unit    " : "f8 16x16x128 varying ",
                       threads, ms, flop / ms * 1e-9, ms * 1e6 / (iters * 16.0 * (threads / 256)));
            }
    }
    return 0;
}
