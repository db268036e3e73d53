// Hardware probe: does a SALU instruction that reads an SGPR pair just written by a VALU compare see all 64 lanes' bits
// when ANOTHER wave on the same SIMD is issuing MFMAs?  ("two streams": the warp-loss backward next
// to the convolution kernels took wrong branches in lanes 48-63.)
//
//   hipcc --offload-arch=gfx950 -O3 scripts/hw/sgpr_forward_probe.hip -o /tmp/sgpr_probe
//   /tmp/sgpr_probe <variant> <seconds>      # next to a process that keeps MFMA waves resident (atomic_share_stress.py
//                                            # --role k_gather), and alone as the control
// variant 0: v_cmp -> s_and_saveexec_b64 (what hipcc emits for a divergent `if`), stores under the mask
//         1: the same with `s_nop 3` between the compare and the SALU read
//         2: v_cmp -> s_mov_b64 (plain SALU copy of the mask), the copy is stored by lane 0
//         3: as 2 with `s_nop 3`
//         4: C++ `if` with a store in it (compiler-generated code), for reference
// Every launch is compared with the first one on the device; mismatching lanes are histogrammed by lane & 63.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int STEPS = 16;

template <int VAR>
__global__ __launch_bounds__(256) void victim(const float* __restrict__ in, float* __restrict__ out, unsigned long long* __restrict__ masks, int n) {
    const int tid = blockIdx.x * 256 + threadIdx.x;
    if (tid >= n) return;
    float x = in[tid];
    float* o = out + (size_t)tid * STEPS;
#pragma unroll
    for (int k = 0; k < STEPS; ++k) {
        const float c = -0.9f + 0.11f * k;
        x = x * 1.0001f + 0.003f;                       // a VALU dependency chain in front of every compare
        float* p = o + k;
        if (VAR == 0 || VAR == 1) {
            unsigned long long saved;
            if (VAR == 0)
                asm volatile("v_cmp_gt_f32 vcc, %1, %2\n\ts_and_saveexec_b64 %0, vcc\n\tglobal_store_dword %3, %1, off\n\ts_mov_b64 exec, %0"
                             : "=&s"(saved) : "v"(x), "v"(c), "v"(p) : "vcc", "memory");
            else
                asm volatile("v_cmp_gt_f32 vcc, %1, %2\n\ts_nop 3\n\ts_and_saveexec_b64 %0, vcc\n\tglobal_store_dword %3, %1, off\n\ts_mov_b64 exec, %0"
                             : "=&s"(saved) : "v"(x), "v"(c), "v"(p) : "vcc", "memory");
        } else if (VAR == 2 || VAR == 3) {
            unsigned long long m;
            if (VAR == 2)
                asm volatile("v_cmp_gt_f32 vcc, %1, %2\n\ts_mov_b64 %0, vcc" : "=s"(m) : "v"(x), "v"(c) : "vcc");
            else
                asm volatile("v_cmp_gt_f32 vcc, %1, %2\n\ts_nop 3\n\ts_mov_b64 %0, vcc" : "=s"(m) : "v"(x), "v"(c) : "vcc");
            if ((threadIdx.x & 63) == 0) masks[(size_t)(tid >> 6) * STEPS + k] = m;
            *p = x;
        } else {
            if (x > c) *p = x;
        }
    }
}

__global__ void compare(const float* a, const float* b, size_t n, const unsigned long long* ma, const unsigned long long* mb,
                        size_t nm, unsigned* lane_hist, unsigned* total) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        if (__float_as_uint(a[i]) != __float_as_uint(b[i])) {
            atomicAdd(&lane_hist[(i / STEPS) & 63], 1u);
            atomicAdd(total, 1u);
        }
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nm; i += (size_t)gridDim.x * blockDim.x) {
        unsigned long long d = ma[i] ^ mb[i];
        if (d) {
            atomicAdd(total, 1u);
            for (int l = 0; l < 64; ++l) if ((d >> l) & 1) atomicAdd(&lane_hist[l], 1u);
        }
    }
}

int main(int argc, char** argv) {
    const int var = argc > 1 ? atoi(argv[1]) : 0;
    const double seconds = argc > 2 ? atof(argv[2]) : 10.0;
    const int n = 256 * 1024;
    std::vector<float> h(n);
    srand(1);
    for (auto& v : h) v = 2.f * rand() / RAND_MAX - 1.f;
    float *in, *out, *ref;
    unsigned long long *m, *mref;
    unsigned *hist, *total;
    const size_t no = (size_t)n * STEPS, nm = (size_t)(n / 64) * STEPS;
    CHECK(hipMalloc(&in, n * 4)); CHECK(hipMalloc(&out, no * 4)); CHECK(hipMalloc(&ref, no * 4));
    CHECK(hipMalloc(&m, nm * 8)); CHECK(hipMalloc(&mref, nm * 8));
    CHECK(hipMalloc(&hist, 64 * 4)); CHECK(hipMalloc(&total, 4));
    CHECK(hipMemcpy(in, h.data(), n * 4, hipMemcpyHostToDevice));
    CHECK(hipMemset(hist, 0, 64 * 4)); CHECK(hipMemset(total, 0, 4));
    auto launch = [&](float* o, unsigned long long* mm) {
        CHECK(hipMemsetAsync(o, 0xff, no * 4, 0));
        CHECK(hipMemsetAsync(mm, 0, nm * 8, 0));
        switch (var) {
            case 0: victim<0><<<n / 256, 256>>>(in, o, mm, n); break;
            case 1: victim<1><<<n / 256, 256>>>(in, o, mm, n); break;
            case 2: victim<2><<<n / 256, 256>>>(in, o, mm, n); break;
            case 3: victim<3><<<n / 256, 256>>>(in, o, mm, n); break;
            default: victim<4><<<n / 256, 256>>>(in, o, mm, n); break;
        }
    };
    launch(ref, mref);
    CHECK(hipDeviceSynchronize());
    long reps = 0, bad = 0;
    unsigned last = 0;
    const auto t0 = std::chrono::steady_clock::now();
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
        for (int r = 0; r < 50; ++r) {
            launch(out, m);
            compare<<<256, 256>>>(out, ref, no, m, mref, nm, hist, total);
            ++reps;
        }
        unsigned t;
        CHECK(hipMemcpy(&t, total, 4, hipMemcpyDeviceToHost));
        if (t != last) { ++bad; last = t; }
    }
    unsigned hh[64], t;
    CHECK(hipMemcpy(hh, hist, 256, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(&t, total, 4, hipMemcpyDeviceToHost));
    printf("variant %d: %ld launches, %u mismatching values/masks, in %ld of %ld batches of 50; by lane quarter: [0-15] %u [16-31] %u [32-47] %u [48-63] %u\n",
           var, reps, t, bad, reps / 50,
           [&] { unsigned s = 0; for (int i = 0; i < 16; ++i) s += hh[i]; return s; }(),
           [&] { unsigned s = 0; for (int i = 16; i < 32; ++i) s += hh[i]; return s; }(),
           [&] { unsigned s = 0; for (int i = 32; i < 48; ++i) s += hh[i]; return s; }(),
           [&] { unsigned s = 0; for (int i = 48; i < 64; ++i) s += hh[i]; return s; }());
    return 0;
}
