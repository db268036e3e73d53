// Hardware probe: what does `buffer_load_dwordx4 ... offen lds` write for out-of-range lanes, and where does lane l land?
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 scripts/hw/lds_dma_probe.hip -o /tmp/probe && /tmp/probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void dma16(u32x4 rsrc, unsigned voff, unsigned soff, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                 :: "s"(lds_dst), "v"(voff), "s"(rsrc), "s"(soff) : "memory", "m0");
}
__global__ void k(const float* x, float* out, int nbytes) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];
    float* f = reinterpret_cast<float*>(dsm);
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) f[i] = -7.f;      // poison
    __syncthreads();
    const unsigned long p = (unsigned long)x;
    u32x4 r = {(unsigned)p, (unsigned)(p >> 32) & 0xffffu, (unsigned)nbytes, 0x00020000u};
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    unsigned voff = (unsigned)(63 - lane) * 16;          // reversed source order: lane l reads chunk 63 - l
    if ((lane & 3) == 1) voff = 0x80000000u;             // out of range
    if ((lane & 3) == 2) voff = (unsigned)nbytes - 8;    // straddles the end
    dma16(r, voff, wid * 1024, (unsigned)(size_t)dsm + wid * 2048 + 1024);   // soffset picks the wave's source KiB
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) out[i] = f[i];
}
int main() {
    const int n = 2048;                                   // floats of source = 8 KiB
    std::vector<float> h(n);
    for (int i = 0; i < n; ++i) h[i] = (float)i;
    float *dx, *dout;
    hipMalloc(&dx, n * 4); hipMalloc(&dout, 2048 * 4);
    hipMemcpy(dx, h.data(), n * 4, hipMemcpyHostToDevice);
    k<<<1, 128, 8192>>>(dx, dout, n * 4);
    std::vector<float> o(2048);
    hipMemcpy(o.data(), dout, 2048 * 4, hipMemcpyDeviceToHost);
    for (int w = 0; w < 2; ++w) {
        printf("wave %d: first KiB (must stay -7): %g %g ; DMA KiB by lane (4 floats each):\n", w, o[w * 512], o[w * 512 + 255]);
        for (int l = 0; l < 8; ++l)
            printf("  lane %d: %g %g %g %g\n", l, o[w * 512 + 256 + 4 * l], o[w * 512 + 256 + 4 * l + 1], o[w * 512 + 256 + 4 * l + 2],
                   o[w * 512 + 256 + 4 * l + 3]);
    }
    return 0;
}
