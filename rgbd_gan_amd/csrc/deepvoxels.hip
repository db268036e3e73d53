// DeepVoxels frustum path (config 4 of the reference): projection index math, trilinear resampling of the voxel grid
// into the camera frustum, accumulative occlusion compositing.  All HBM-bound (4.2 MB grid -> 29.4 MB frustum volume
// per sample); fp32 like the reference.
//
// Replaces deepvoxel/projection.py:48-105 (compute_proj_idcs, a Python loop over the batch with boolean
// compaction), deepvoxel/deepvoxel.py:388-428 (interpolate_trilinear: 8 advanced-index gathers + scatter_add) and
// deepvoxel/deepvoxel.py:574-587,886-889,903-904 (AccumulativeOcclusionNet + compositing + depth rescale).
//
// COMPILED WITH -ffp-contract=off: the index math is specified unfused, left to right, in float32
// (oracle/deepvoxels.py), and must be bit-exact against it.
#include "common.h"

namespace {

struct FrustumArgs {
    int W, H, D, G;          // image width/height, frustum depth, grid side
    float voxel, near_plane, fx, fy, cx, cy;
};

// voxel coordinates of frustum element n for camera matrix C (row-major 4x4); returns the in-grid mask
__device__ __forceinline__ bool frustum_point(const FrustumArgs& f, const float* __restrict__ C, int n, float v[3]) {
    const int wh = f.W * f.H;
    const int zi = n / wh;
    float zc = (float)zi;
    const int tmp = n - (int)((zc * (float)f.W) * (float)f.H);
    float yc = (float)tmp / (float)f.W;                 // true division: fractional row coordinate (projection.py:69)
    float xc = (float)(tmp % f.W);
    zc = zc * f.voxel;
    zc = zc + f.near_plane;
    xc = (xc - f.cx) / f.fx;
    yc = (yc - f.cy) / f.fy;
    xc = xc * zc;
    yc = yc * zc;
    bool in = true;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float g = ((C[4 * k + 0] * xc + C[4 * k + 1] * yc) + C[4 * k + 2] * zc) + C[4 * k + 3];
        v[k] = g / f.voxel + (float)(f.G / 2);
        in = in && (v[k] >= 0.f) && (v[k] < (float)f.G);
    }
    return in;
}

// pass 1: number of in-grid elements per 256-element block (order-preserving compaction needs a scan)
__global__ __launch_bounds__(256) void proj_count_kernel(FrustumArgs f, const float* __restrict__ cams, int N,
                                                         int* __restrict__ block_counts) {
    const int b = blockIdx.y;
    const int n = blockIdx.x * 256 + threadIdx.x;
    float v[3];
    const bool in = n < N && frustum_point(f, cams + b * 16, n, v);
    const unsigned long long m = __ballot(in);
    __shared__ int wc[4];
    if ((threadIdx.x & 63) == 0) wc[threadIdx.x >> 6] = __popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) block_counts[b * gridDim.x + blockIdx.x] = wc[0] + wc[1] + wc[2] + wc[3];
}

// pass 2: exclusive scan of the block counts of one sample (one block per sample), total -> counts[b]
__global__ __launch_bounds__(1024) void proj_scan_kernel(int* __restrict__ block_counts, int nblocks,
                                                         int* __restrict__ counts) {
    const int b = blockIdx.x;
    int* bc = block_counts + b * nblocks;
    __shared__ int part[1024];
    const int per = (nblocks + 1023) / 1024;
    const int lo = threadIdx.x * per, hi = min(nblocks, lo + per);
    int s = 0;
    for (int i = lo; i < hi; ++i) s += bc[i];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {          // Hillis-Steele inclusive scan
        const int t = threadIdx.x >= off ? part[threadIdx.x - off] : 0;
        __syncthreads();
        part[threadIdx.x] += t;
        __syncthreads();
    }
    int run = part[threadIdx.x] - s;
    for (int i = lo; i < hi; ++i) { const int c = bc[i]; bc[i] = run; run += c; }
    if (threadIdx.x == 1023) counts[b] = part[1023];
}

// pass 3: recompute and write (n, v) at block offset + rank inside the block
__global__ __launch_bounds__(256) void proj_write_kernel(FrustumArgs f, const float* __restrict__ cams, int N,
                                                         const int* __restrict__ block_offsets,
                                                         int* __restrict__ idx, float* __restrict__ coords) {
    const int b = blockIdx.y;
    const int n = blockIdx.x * 256 + threadIdx.x;
    float v[3];
    const bool in = n < N && frustum_point(f, cams + b * 16, n, v);
    const unsigned long long m = __ballot(in);
    __shared__ int wc[4];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (lane == 0) wc[wid] = __popcll(m);
    __syncthreads();
    int base = block_offsets[b * gridDim.x + blockIdx.x];
    for (int w2 = 0; w2 < wid; ++w2) base += wc[w2];
    if (in) {
        const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
        idx[(long)b * N + pos] = n;
#pragma unroll
        for (int k = 0; k < 3; ++k) coords[((long)b * 3 + k) * N + pos] = v[k];
    }
}

// ------------------------------------------------------------------------------------------------ trilinear
struct Corners {
    int o[8];
    float w[8];
};

__device__ __forceinline__ Corners trilinear_corners_at(float xi, float yi, float zi, int G);
__device__ __forceinline__ Corners trilinear_corners(const float* __restrict__ coords, long base, int N, int pos,
                                                     int G) {
    // deepvoxel.py:394-410: x = v[2], y = v[1], z = v[0]; grid indexed [x][y][z]
    return trilinear_corners_at(coords[base + 2l * N + pos], coords[base + 1l * N + pos], coords[base + pos], G);
}
__device__ __forceinline__ Corners trilinear_corners_at(float xi, float yi, float zi, int G) {
    const int x0 = (int)xi, y0 = (int)yi, z0 = (int)zi;
    const int x1 = min(max(x0 + 1, 0), G - 1), y1 = min(max(y0 + 1, 0), G - 1), z1 = min(max(z0 + 1, 0), G - 1);
    const float x = xi - (float)x0, y = yi - (float)y0, z = zi - (float)z0;
    Corners c;
    // same order as the reference's eight terms; each weight is a left-to-right product
    c.o[0] = (x0 * G + y0) * G + z0; c.w[0] = ((1.f - x) * (1.f - y)) * (1.f - z);
    c.o[1] = (x1 * G + y0) * G + z0; c.w[1] = (x * (1.f - y)) * (1.f - z);
    c.o[2] = (x0 * G + y1) * G + z0; c.w[2] = ((1.f - x) * y) * (1.f - z);
    c.o[3] = (x0 * G + y0) * G + z1; c.w[3] = ((1.f - x) * (1.f - y)) * z;
    c.o[4] = (x1 * G + y0) * G + z1; c.w[4] = (x * (1.f - y)) * z;
    c.o[5] = (x0 * G + y1) * G + z1; c.w[5] = ((1.f - x) * y) * z;
    c.o[6] = (x1 * G + y1) * G + z0; c.w[6] = (x * y) * (1.f - z);
    c.o[7] = (x1 * G + y1) * G + z1; c.w[7] = (x * y) * z;
    return c;
}

// out (B,F,N) must be zero-filled; thread = one compacted frustum element, loop over features
__global__ __launch_bounds__(256) void trilinear_fwd_kernel(const float* __restrict__ grid, const int* __restrict__ idx,
                                                            const float* __restrict__ coords,
                                                            const int* __restrict__ counts, float* __restrict__ out,
                                                            int F, int G, int N) {
    const int b = blockIdx.y;
    const int pos = blockIdx.x * 256 + threadIdx.x;
    if (pos >= counts[b]) return;
    const Corners c = trilinear_corners(coords, (long)b * 3 * N, N, pos, G);
    const int n = idx[(long)b * N + pos];
    const long g3 = (long)G * G * G;
    for (int f = 0; f < F; ++f) {
        const float* g = grid + ((long)b * F + f) * g3;
        float acc = g[c.o[0]] * c.w[0];
#pragma unroll
        for (int k = 1; k < 8; ++k) acc = acc + g[c.o[k]] * c.w[k];
        out[((long)b * F + f) * N + n] = acc;
    }
}

// Backward of the resampling: dgrid[b][f][corner] += w * dout[b][f][n].  587 M fp32 atomics per step at the shipped
// sizes; issued feature-major into a (B, G^3, F) scratch so that the 32 features of one sample's corner are ONE
// 128-byte line per atomic instruction (scattered 4-byte atomics into the (B,F,G^3) layout ran at 45 G atomics/s:
// 13 ms per step).  A block takes 64 compacted samples: their dout values are read sample-major (coalesced along n),
// turned in LDS, and every sample's eight corners are computed once.
constexpr int TRI_S = 64;          // samples per block
__global__ __launch_bounds__(256) void trilinear_bwd_scatter_kernel(const float* __restrict__ dout,
                                                                    const int* __restrict__ idx,
                                                                    const float* __restrict__ coords,
                                                                    const int* __restrict__ counts,
                                                                    float* __restrict__ ws, int F, int G, int N) {
    __shared__ float tile[TRI_S][33];
    __shared__ int co[TRI_S][8];
    __shared__ float cw[TRI_S][8];
    const int b = blockIdx.y;
    const int p0 = blockIdx.x * TRI_S;
    const int cnt = counts[b];
    if (p0 >= cnt) return;
    const int tid = threadIdx.x;
    {
        const int p = tid & (TRI_S - 1), fq = tid >> 6;           // 4 feature phases
        const int pos = p0 + p;
        const bool live = pos < cnt;
        const int n = live ? idx[(long)b * N + pos] : 0;
        for (int f = fq; f < F; f += 4) tile[p][f] = live ? dout[((long)b * F + f) * N + n] : 0.f;
        if (fq == 0) {
            if (live) {
                const Corners c = trilinear_corners(coords, (long)b * 3 * N, N, pos, G);
#pragma unroll
                for (int k = 0; k < 8; ++k) { co[p][k] = c.o[k]; cw[p][k] = c.w[k]; }
            } else {
#pragma unroll
                for (int k = 0; k < 8; ++k) { co[p][k] = 0; cw[p][k] = 0.f; }
            }
        }
    }
    __syncthreads();
    // 32 feature lanes x 8 runs of 8 CONSECUTIVE samples: consecutive compacted samples are neighbouring pixels of one
    // frustum row, whose corner voxels change every 2-3 pixels, so a lane adds up the contributions to the same voxel (per
    // corner slot) in a register and issues one atomic per run of equal voxels instead of one per sample
    const int f = tid & 31, e0 = (tid >> 5) * 8;
    if (f >= F) return;
    const long g3 = (long)G * G * G;
    float* base = ws + (long)b * g3 * F + f;
    const int e1 = min(e0 + 8, cnt - p0);
    if (e0 >= e1) return;
    float go[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) go[i] = e0 + i < e1 ? tile[e0 + i][f] : 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        int cur = co[e0][k];
        float acc = go[0] * cw[e0][k];
#pragma unroll
        for (int i = 1; i < 8; ++i) {
            if (e0 + i < e1) {
                const int v = co[e0 + i][k];
                if (v != cur) {
                    atomicAdd(base + (long)cur * F, acc);
                    cur = v;
                    acc = 0.f;
                }
                acc += go[i] * cw[e0 + i][k];
            }
        }
        atomicAdd(base + (long)cur * F, acc);
    }
}

// The same backward over BRICKS of the frustum (16 x 8 pixels x 2 depth slices = 256 samples per workgroup) with the brick's
// 2048 (sample, corner) contributions SORTED by voxel before anything is added: one line atomic per DISTINCT voxel of the
// brick (~100) instead of one per run of equal voxels along a pixel row (~800) -- neighbouring rows and depth slices hit the same
// voxels, which the row-wise list kernel above cannot see (its counter traffic was 3.0x the algorithmic bytes, the memory-side
// atomic rate its bound).  The voxel coordinates are recomputed from the camera (frustum_point: the projection kernels' own
// arithmetic, bit for bit), so the brick needs no compacted list.  LDS: the brick's dout tile [256][F] (stride 33), the corner
// weights [256][8], 2048 sort words = (relative voxel position << 11 | sample << 3 | corner), sorted by a counting sort; then 8
// groups of 32 feature lanes walk an equal share of the sorted words each and flush a register sum whenever the voxel changes.  Round 5's LDS
// HASH-TABLE form of the same idea lost to the LDS float atomics' serialisation (profiles/r05/trilinear_brick_experiment.txt);
// here nothing is added in LDS.
constexpr int TB_X = 16, TB_Y = 8, TB_D = 2;
constexpr int TB_S = TB_X * TB_Y * TB_D;            // 256 samples
constexpr int TB_BINS = 4096;                       // 16^3 voxel positions relative to the brick's lowest corner voxel
__global__ __launch_bounds__(256) void trilinear_bwd_brick_kernel(FrustumArgs fa, const float* __restrict__ cams,
                                                                  const float* __restrict__ dout, float* __restrict__ ws,
                                                                  int F, int N, int nbricks, int ko) {
    // (ko: timing knock-outs of the debug library, 0 in the shipped one: 1 = no dout loads, 2 = stop behind the loads,
    //  3 = stop behind the sort, 4 = fold without the atomics)
    // A brick is at most 16 x 8 pixels x 2 slices of ONE camera's frustum: its physical diagonal is < 12 voxels at the far plane,
    // so the corner voxels of all its samples lie within 16 voxels of the lowest one along every axis -- 4 bits per axis.  The
    // sort by voxel is therefore a COUNTING sort in LDS (a histogram over 4096 relative positions, an exclusive scan that also
    // lists the occupied positions, a scatter) instead of a comparison sort (the 66-pass bitonic network of this kernel's first
    // form took as long as the row-wise kernel's atomics: 554 vs 648 us, scripts/time_trilinear_bwd.py).
    __shared__ __attribute__((aligned(16))) float tile[TB_S][36];     // row stride 144 B: 16-byte reads of four features
    __shared__ float cw[TB_S][8];
    __shared__ unsigned bins[TB_BINS];
    __shared__ unsigned short sorted_[TB_S * 8];      // (sample << 3 | corner), grouped by voxel
    __shared__ unsigned short bstart[TB_S * 8 + 1];   // occupied voxel j: its words are sorted_[bstart[j] .. bstart[j + 1])
    __shared__ unsigned short bkey[TB_S * 8];         //                   its relative position (x << 8 | y << 4 | z)
    __shared__ int box[4][3];
    __shared__ unsigned wsum[4];
    const int tid = threadIdx.x;
    // PERSISTENT workgroups (two per CU): a workgroup walks brick PAIRS p = blockIdx, blockIdx + grid, ...; bricks 2p and 2p + 1 are
    // x neighbours, which share every 128-byte line of dout (a brick row is 16 floats), so the second one finds its half in this CU's
    // cache.  One launch per brick cost ~80 us per launch before a brick did anything (8960 workgroups of 74 KB LDS each, 40 % of them
    // bricks that lie outside the grid and exit at once: profiles/r06/trilinear_bwd_bricks.txt).
    const int bricks_x = fa.W / TB_X, bricks_y = fa.H / TB_Y, bricks_d = (fa.D + TB_D - 1) / TB_D;
    const int lane = tid & 63, wv = tid >> 6;
    // (the pair's two bricks on two workgroups of ONE XCD at the same time instead of back to back on one workgroup: 336 against
    //  322 us, profiles/r06/trilinear_bwd_bricks.txt)
    for (int brick = 2 * (int)blockIdx.x; brick < nbricks; brick += (brick & 1) ? 2 * (int)gridDim.x - 1 : 1) {
    __syncthreads();                                // the previous brick's fold has finished with the LDS images
    const int bxi = brick % bricks_x, byi = (brick / bricks_x) % bricks_y, bdi = (brick / (bricks_x * bricks_y)) % bricks_d;
    const int b = brick / (bricks_x * bricks_y * bricks_d);
    const int sx = tid & (TB_X - 1), sy = (tid >> 4) & (TB_Y - 1), sd = tid >> 7;
    const int d = bdi * TB_D + sd;
    const int n = (d * fa.H + (byi * TB_Y + sy)) * fa.W + bxi * TB_X + sx;
    float v[3];
    const bool live = d < fa.D && frustum_point(fa, cams + b * 16, n, v);
    if (!__syncthreads_or(live)) continue;          // the whole brick lies outside the grid (frustum corners)
    if (ko == 8) { if (v[0] == 1.2345f) ws[0] = 1.f; return; }
    // the sample's dout values: requested now, parked in LDS behind the sort (their latency covers it)
    float dv[32];
#pragma unroll
    for (int f = 0; f < 32; ++f) dv[f] = (live && f < F && ko != 1) ? dout[((long)b * F + f) * N + n] : 0.f;
    for (int i = tid; i < TB_BINS; i += 256) bins[i] = 0u;
    // corners (deepvoxel.py:394-410: x = v[2], y = v[1], z = v[0]; same arithmetic as trilinear_corners_at)
    int cx0 = 0x7fffffff, cy0 = 0x7fffffff, cz0 = 0x7fffffff, cx1 = 0, cy1 = 0, cz1 = 0;
    if (live) {
        const float xi = v[2], yi = v[1], zi = v[0];
        cx0 = (int)xi; cy0 = (int)yi; cz0 = (int)zi;
        cx1 = min(max(cx0 + 1, 0), fa.G - 1); cy1 = min(max(cy0 + 1, 0), fa.G - 1); cz1 = min(max(cz0 + 1, 0), fa.G - 1);
        const float x = xi - (float)cx0, y = yi - (float)cy0, z = zi - (float)cz0;
        cw[tid][0] = ((1.f - x) * (1.f - y)) * (1.f - z);
        cw[tid][1] = (x * (1.f - y)) * (1.f - z);
        cw[tid][2] = ((1.f - x) * y) * (1.f - z);
        cw[tid][3] = ((1.f - x) * (1.f - y)) * z;
        cw[tid][4] = (x * (1.f - y)) * z;
        cw[tid][5] = ((1.f - x) * y) * z;
        cw[tid][6] = (x * y) * (1.f - z);
        cw[tid][7] = (x * y) * z;
    }
    {   // the brick's lowest corner voxel: wave minima by shuffles, the four waves' through LDS (no LDS atomics)
        int mx = cx0, my = cy0, mz = cz0;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            mx = min(mx, __shfl_xor(mx, off)); my = min(my, __shfl_xor(my, off)); mz = min(mz, __shfl_xor(mz, off));
        }
        if (lane == 0) { box[wv][0] = mx; box[wv][1] = my; box[wv][2] = mz; }
    }
    __syncthreads();
    if (ko == 9) { if (box[0][0] == 12345 || dv[tid & 31] == 1.2345f) ws[0] = 1.f; return; }
    const int bx = min(min(box[0][0], box[1][0]), min(box[2][0], box[3][0]));
    const int by = min(min(box[0][1], box[1][1]), min(box[2][1], box[3][1]));
    const int bz = min(min(box[0][2], box[1][2]), min(box[2][2], box[3][2]));
    unsigned key[8];
    bool fits = true;
    if (live) {
        const int lx0 = cx0 - bx, lx1 = cx1 - bx, ly0 = cy0 - by, ly1 = cy1 - by, lz0 = cz0 - bz, lz1 = cz1 - bz;
        fits = (lx1 | ly1 | lz1 | lx0 | ly0 | lz0) < 16;                // (x1 >= x0 >= box: all non-negative)
        // corner order of trilinear_corners_at: (x0y0z0, x1y0z0, x0y1z0, x0y0z1, x1y0z1, x0y1z1, x1y1z0, x1y1z1)
        key[0] = (lx0 << 8) | (ly0 << 4) | lz0; key[1] = (lx1 << 8) | (ly0 << 4) | lz0;
        key[2] = (lx0 << 8) | (ly1 << 4) | lz0; key[3] = (lx0 << 8) | (ly0 << 4) | lz1;
        key[4] = (lx1 << 8) | (ly0 << 4) | lz1; key[5] = (lx0 << 8) | (ly1 << 4) | lz1;
        key[6] = (lx1 << 8) | (ly1 << 4) | lz0; key[7] = (lx1 << 8) | (ly1 << 4) | lz1;
    }
    if (__syncthreads_or(!fits)) {
        // never at the shipped geometry (see above); a brick that does not fit falls back to one atomic per contribution
        if (live) {
            const int G = fa.G;
            const int o[8] = {(cx0 * G + cy0) * G + cz0, (cx1 * G + cy0) * G + cz0, (cx0 * G + cy1) * G + cz0, (cx0 * G + cy0) * G + cz1,
                              (cx1 * G + cy0) * G + cz1, (cx0 * G + cy1) * G + cz1, (cx1 * G + cy1) * G + cz0, (cx1 * G + cy1) * G + cz1};
            float* base = ws + (long)b * G * G * G * F;
            for (int f = 0; f < F; ++f)
#pragma unroll
                for (int k = 0; k < 8; ++k) atomicAdd(base + (long)o[k] * F + f, dv[f] * cw[tid][k]);
        }
        continue;
    }
    if (live) {
#pragma unroll
        for (int k = 0; k < 8; ++k) atomicAdd(&bins[key[k]], 1u);
    }
#pragma unroll
    for (int f = 0; f < 32; ++f) tile[tid][f] = dv[f];
    __syncthreads();
    if (ko == 2) { if (tile[tid][tid & 31] == 1.2345f) ws[0] = 1.f; return; }
    // exclusive scan of the 4096 counts (and of the number of occupied positions, in the upper half of the same word): 16 positions
    // per thread, a wave scan of the thread totals, the four wave totals through LDS
    unsigned c16[16], mine = 0u;
#pragma unroll
    for (int i = 0; i < 16; ++i) { c16[i] = bins[tid * 16 + i]; mine += c16[i] + (c16[i] ? 0x10000u : 0u); }
    unsigned inc = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned t = __shfl_up(inc, off);
        if (lane >= off) inc += t;
    }
    if (lane == 63) wsum[wv] = inc;
    __syncthreads();
    unsigned run = inc - mine;
    for (int w2 = 0; w2 < wv; ++w2) run += wsum[w2];
    const unsigned total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    const int nwords = (int)(total & 0xffffu), nocc = (int)(total >> 16);
    {
        unsigned words = run & 0xffffu, occ = run >> 16;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            bins[tid * 16 + i] = words;
            if (c16[i]) { bstart[occ] = (unsigned short)words; bkey[occ] = (unsigned short)(tid * 16 + i); ++occ; }
            words += c16[i];
        }
        if (tid == 255) bstart[nocc] = (unsigned short)nwords;
    }
    __syncthreads();
    if (live) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const unsigned pos = atomicAdd(&bins[key[k]], 1u);
            sorted_[pos] = (unsigned short)((tid << 3) | k);
        }
    }
    __syncthreads();
    if (ko == 3) { if (sorted_[tid] == 0xbeefu) ws[0] = 1.f; return; }
    // fold: 32 groups of 8 lanes, a lane owns FOUR features (one 16-byte LDS read per word); a group takes every 32nd occupied
    // voxel and adds up that voxel's words (no comparisons: the voxel's range is known), four words in flight at a time; then
    // four atomics per lane = the voxel's 128-byte line.  (With 32 feature lanes per group -- one feature per lane -- the groups'
    // serial chains of dependent LDS reads were four times as long and the fold took 260 of the kernel's 450 us.)
    const int fq = (tid & 7) * 4, g = tid >> 3;
    const int G = fa.G;
    float* wsb = ws + (long)b * G * G * G * F;
    const int iters = (nocc + 31) >> 5;              // uniform: the hand-over below is a wave-wide exchange
    for (int it = 0; it < iters; ++it) {
        const int j = g + 32 * it;
        const bool valid = j < nocc;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        int vo = -1;
        if (valid) {
            const int s0 = bstart[j], s1 = bstart[j + 1];
            const unsigned k2 = bkey[j];
            for (int i = s0; i < s1; i += 4) {
                unsigned e[4];
                f32x4 t[4];
                float c[4];
#pragma unroll
                for (int q2 = 0; q2 < 4; ++q2) e[q2] = sorted_[min(i + q2, s1 - 1)];
#pragma unroll
                for (int q2 = 0; q2 < 4; ++q2) { t[q2] = *reinterpret_cast<const f32x4*>(&tile[e[q2] >> 3][fq]); c[q2] = cw[0][e[q2]]; }
#pragma unroll
                for (int q2 = 0; q2 < 4; ++q2) acc += (i + q2 < s1 ? c[q2] : 0.f) * t[q2];
            }
            vo = ((bx + (int)(k2 >> 8)) * G + by + (int)((k2 >> 4) & 15u)) * G + bz + (int)(k2 & 15u);
        }
        // hand-over inside the wave: its 8 groups hold 8 voxels x 32 features as (group, feature quad) x 4; the atomics want whole
        // lines -- lane L of instruction c adds feature L & 31 of the voxel of group 2c + (L >> 5): 2 full 128-byte lines per
        // instruction instead of 8 quarter lines (four scalar atomics per lane touched every line four times: 80 us per launch)
#pragma unroll
        for (int c2 = 0; c2 < 4; ++c2) {
            const int f = lane & 31;
            const int src = 8 * (2 * c2 + (lane >> 5)) + (f >> 2);
            const float x0 = __shfl(acc[0], src), x1 = __shfl(acc[1], src), x2 = __shfl(acc[2], src), x3 = __shfl(acc[3], src);
            const int vsrc = __shfl(vo, src);
            const float val = (f & 2) ? ((f & 1) ? x3 : x2) : ((f & 1) ? x1 : x0);
            if (vsrc >= 0 && f < F) {
                if (ko != 4) atomicAdd(wsb + (long)vsrc * F + f, val);
                else if (val == 1.2345f) wsb[0] = val;
            }
        }
    }
    }   // bricks of this workgroup
}

// Forward from a FEATURE-MINOR grid (B, G^3, F) -- the layout the voxel generator's NHWC conv stack produces: the 32
// features of a corner are one 128-byte line, so a sample costs 8 line reads instead of 256 scattered 4-byte gathers.  A
// block takes 64 compacted samples: 32 feature lanes x 8 sample groups accumulate (same term order as
// trilinear_fwd_kernel: bit-identical values), the tile turns in LDS and leaves sample-major.
__global__ __launch_bounds__(256) void trilinear_fwd_fm_kernel(const float* __restrict__ grid, const int* __restrict__ idx,
                                                               const float* __restrict__ coords,
                                                               const int* __restrict__ counts, float* __restrict__ out, int F,
                                                               int G, int N) {
    __shared__ float tile[TRI_S][33];
    __shared__ int co[TRI_S][8], nn[TRI_S];
    __shared__ float cw[TRI_S][8];
    const int b = blockIdx.y;
    const int p0 = blockIdx.x * TRI_S;
    const int cnt = counts[b];
    if (p0 >= cnt) return;
    const int tid = threadIdx.x;
    if (tid < TRI_S) {
        const int pos = p0 + tid;
        if (pos < cnt) {
            const Corners c = trilinear_corners(coords, (long)b * 3 * N, N, pos, G);
#pragma unroll
            for (int k = 0; k < 8; ++k) { co[tid][k] = c.o[k]; cw[tid][k] = c.w[k]; }
            nn[tid] = idx[(long)b * N + pos];
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) { co[tid][k] = 0; cw[tid][k] = 0.f; }
            nn[tid] = -1;
        }
    }
    __syncthreads();
    const long g3 = (long)G * G * G;
    {
        const int f = tid & 31, e0 = tid >> 5;
        if (f < F) {
            const float* g = grid + (long)b * g3 * F + f;
            for (int e = e0; e < TRI_S; e += 8) {
                float acc = g[(long)co[e][0] * F] * cw[e][0];
#pragma unroll
                for (int k = 1; k < 8; ++k) acc = acc + g[(long)co[e][k] * F] * cw[e][k];
                tile[e][f] = acc;
            }
        }
    }
    __syncthreads();
    const int p = tid & (TRI_S - 1), fq = tid >> 6;
    const int n = nn[p];
    if (n >= 0)
        for (int f = fq; f < F; f += 4) out[((long)b * F + f) * N + n] = tile[p][f];
}

// The feature-minor forward WITHOUT the compacted list and WITHOUT a zero fill of its 293 MB output: a workgroup takes 64
// consecutive frustum elements (a pixel row of a depth slice), recomputes their voxel coordinates from the camera
// (frustum_point: bit for bit what the projection kernels compute), gathers as trilinear_fwd_fm_kernel does (same term order: the
// same bits) and writes EVERY element of its runs -- zeros where the element lies outside the grid.  The list form cleared the
// whole output first (a 293 MB fill per call) and then wrote the 55 % of it that lies inside.
constexpr int TF_S = 64;     // elements per workgroup (with 256 the gather was a long chain per workgroup at three workgroups per CU: 165 us against 121)
__global__ __launch_bounds__(256) void trilinear_fwd_frustum_kernel(FrustumArgs fa, const float* __restrict__ cams,
                                                                    const float* __restrict__ grid, float* __restrict__ out,
                                                                    int F, int N) {
    __shared__ float tile[TF_S][33];
    __shared__ int co[TF_S][8];
    __shared__ float cw[TF_S][8];
    __shared__ int lv[TF_S];
    const int b = blockIdx.y;
    const int n0 = blockIdx.x * TF_S;
    const int tid = threadIdx.x;
    if (tid < TF_S) {
        const int n = n0 + tid;
        float v[3];
        const bool live = n < N && frustum_point(fa, cams + b * 16, n, v);
        if (live) {
            const Corners c = trilinear_corners_at(v[2], v[1], v[0], fa.G);
#pragma unroll
            for (int k = 0; k < 8; ++k) { co[tid][k] = c.o[k]; cw[tid][k] = c.w[k]; }
        }
        lv[tid] = live ? 1 : 0;
    }
    __syncthreads();
    const int G = fa.G;
    const long g3 = (long)G * G * G;
    {
        const int f = tid & 31, e0 = tid >> 5;
        if (f < F) {
            const float* g = grid + (long)b * g3 * F + f;
            for (int e = e0; e < TF_S; e += 8) {
                float acc = 0.f;
                if (lv[e]) {
                    acc = g[(long)co[e][0] * F] * cw[e][0];
#pragma unroll
                    for (int k = 1; k < 8; ++k) acc = acc + g[(long)co[e][k] * F] * cw[e][k];
                }
                tile[e][f] = acc;
            }
        }
    }
    __syncthreads();
    const int p = tid & (TF_S - 1), fq = tid >> 6;
    const int n = n0 + p;
    if (n < N)
        for (int f = fq; f < F; f += 4) out[((long)b * F + f) * N + n] = tile[p][f];
}

// (B, V, F) -> (B, F, V) through a 32 x 32 LDS tile
__global__ __launch_bounds__(256) void transpose_vf_kernel(const float* __restrict__ in, float* __restrict__ out, long V,
                                                           int F) {
    __shared__ float t[32][33];
    const int b = blockIdx.z;
    const long v0 = (long)blockIdx.x * 32;
    const int f0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;        // 8 rows per pass
    for (int r = ty; r < 32; r += 8)
        t[r][tx] = (v0 + r < V && f0 + tx < F) ? in[((long)b * V + v0 + r) * F + f0 + tx] : 0.f;
    __syncthreads();
    for (int r = ty; r < 32; r += 8)
        if (f0 + r < F && v0 + tx < V) out[((long)b * F + f0 + r) * V + v0 + tx] = t[tx][r];
}

// ------------------------------------------------------------------------------------------------ occlusion
constexpr int OCC_NF = 4;       // occnet_nf (deepvoxel.py:835)
constexpr int OCC_MAXF = 32;    // grid features

struct OccArgs {
    int B, F, D, HW;            // vol is (B, F, D, HW)
    float c1, c2, threshold;    // equalized-LR scales sqrt(2/(F+1)), sqrt(2/nf)
};

__device__ __forceinline__ float depth_coord(int d, int D) { return (float)(d - D / 2) / (float)D; }

// s[b,d,p] = sigmoid( W2 . lrelu(W1 . c1*[coord, vol] + b1) * c2 + b2 - threshold )
__global__ __launch_bounds__(256) void occ_score_kernel(OccArgs a, const float* __restrict__ vol,
                                                        const float* __restrict__ W1, const float* __restrict__ b1,
                                                        const float* __restrict__ W2, const float* __restrict__ b2,
                                                        float* __restrict__ s) {
    const long vox = (long)a.D * a.HW;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)a.B * vox) return;
    const int b = (int)(i / vox);
    const long r = i - (long)b * vox;
    const int d = (int)(r / a.HW);
    float h[OCC_NF];
    const float xc = depth_coord(d, a.D) * a.c1;
#pragma unroll
    for (int j = 0; j < OCC_NF; ++j) h[j] = b1[j] + W1[j * (a.F + 1)] * xc;
    for (int f = 0; f < a.F; ++f) {
        const float x = vol[((long)b * a.F + f) * vox + r] * a.c1;
#pragma unroll
        for (int j = 0; j < OCC_NF; ++j) h[j] += W1[j * (a.F + 1) + 1 + f] * x;
    }
    float pre = b2[0];
#pragma unroll
    for (int j = 0; j < OCC_NF; ++j) {
        const float hj = h[j] > 0.f ? h[j] : 0.2f * h[j];
        pre += W2[j] * (hj * a.c2);
    }
    s[i] = 1.f / (1.f + __expf(-(pre - a.threshold)));
}

// per ray: w[d] = clip(cumsum s, 0, 1)[d] - clip(...)[d-1]; depth = sum w * coord (rescaled)
__global__ __launch_bounds__(256) void occ_scan_kernel(OccArgs a, const float* __restrict__ s, float* __restrict__ w,
                                                       float* __restrict__ depth, float depth_scale, float near_plane) {
    const long ray = (long)blockIdx.x * 256 + threadIdx.x;
    if (ray >= (long)a.B * a.HW) return;
    const int b = (int)(ray / a.HW);
    const int p = (int)(ray - (long)b * a.HW);
    const long base = (long)b * a.D * a.HW + p;
    float run = 0.f, prev = 0.f, dacc = 0.f;
    for (int d = 0; d < a.D; ++d) {
        run += s[base + (long)d * a.HW];
        const float c = fminf(fmaxf(run, 0.f), 1.f);
        const float wd = c - prev;
        prev = c;
        w[base + (long)d * a.HW] = wd;
        dacc += depth_coord(d, a.D) * wd;
    }
    depth[ray] = ((dacc + 0.5f) * depth_scale) + near_plane;
}

// feat[b,f,p] = sum_d w[b,d,p] * vol[b,f,d,p]
__global__ __launch_bounds__(256) void occ_compose_kernel(OccArgs a, const float* __restrict__ vol,
                                                          const float* __restrict__ w, float* __restrict__ feat) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)a.B * a.F * a.HW) return;
    const int p = (int)(i % a.HW);
    const long bf = i / a.HW;
    const int b = (int)(bf / a.F);
    float acc = 0.f;
    for (int d = 0; d < a.D; ++d)
        acc += w[((long)b * a.D + d) * a.HW + p] * vol[(bf * a.D + d) * a.HW + p];
    feat[i] = acc;
}

// The three forward kernels above as ONE pass over the volume for F = 32 (the shipped grid width): a thread owns a ray,
// holds its 32 feature sums in registers and walks the depth axis with the next depth's 32 loads in flight while it scores
// and composes the current one -- the volume (335 MB at B = 20, D = 32, 64x64 rays) is read once instead of twice, and the
// score / weight planes are written once instead of written and read back.  Same expressions in the same order as
// occ_score / occ_scan / occ_compose: bit-identical outputs (tests/test_deepvoxels.py).
template <int F>
__global__ __launch_bounds__(256, 3) void occ_fwd_fused_kernel(OccArgs a, const float* __restrict__ vol,
                                                            const float* __restrict__ W1, const float* __restrict__ b1,
                                                            const float* __restrict__ W2, const float* __restrict__ b2,
                                                            float* __restrict__ s, float* __restrict__ w,
                                                            float* __restrict__ feat, float* __restrict__ depth,
                                                            float depth_scale, float near_plane) {
    // the first layer's weights, transposed to [input][4 hidden units], in LDS: one broadcast ds_read_b128 per input feature
    // and depth step (as 132 + 9 scalars they did not fit the scalar file and were spilled to vector lanes)
    constexpr int F1 = F + 1;
    __shared__ f32x4 w1t[F1];
    static_assert(OCC_NF == 4, "w1t packs the four hidden units of an input into one float4");
    if (threadIdx.x < F1) {
        const f32x4 t = {W1[threadIdx.x], W1[F1 + threadIdx.x], W1[2 * F1 + threadIdx.x], W1[3 * F1 + threadIdx.x]};
        w1t[threadIdx.x] = t;
    }
    __syncthreads();
    long ray = (long)blockIdx.x * 256 + threadIdx.x;
    const bool live = ray < (long)a.B * a.HW;
    if (!live) return;
    const int b = (int)(ray / a.HW);
    const int p = (int)(ray - (long)b * a.HW);
    const long vox = (long)a.D * a.HW;
    // volume reads: (per-thread byte offset of the ray) + (scalar offset of the feature / depth plane) against one buffer
    // descriptor -- one address register instead of 32 pointer pairs (the launcher checks that the volume is below 4 GB)
    const auto vrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(vol), 0, (int)((long)a.B * F * vox * 4), 0x00020000);
    const unsigned voff = (unsigned)(((long)b * F * vox + p) * 4);
    const long base = (long)b * vox + p;
    float acc[F], xa[F], xb[F];
#pragma unroll
    for (int f = 0; f < F; ++f) acc[f] = 0.f;
    float run = 0.f, prev = 0.f, dacc = 0.f;
    auto load = [&](float (&x)[F], int d) {
#pragma unroll
        for (int f = 0; f < F; ++f)
            x[f] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(vrsrc, voff, (unsigned)(((long)f * vox + (long)d * a.HW) * 4), 0));
    };
    auto step = [&](const float (&x)[F], int d) {
        float h[OCC_NF];
        const float xc = depth_coord(d, a.D) * a.c1;
        unsigned opaque = 0;            // the weights are re-read from LDS every step (hoisted out of the depth loop they
        asm volatile("" : "+v"(opaque));                                   // occupied 132 registers: one wave per SIMD)
        const f32x4* wl = reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(w1t) + opaque);
#pragma unroll
        for (int j = 0; j < OCC_NF; ++j) h[j] = b1[j] + wl[0][j] * xc;
#pragma unroll
        for (int f = 0; f < F; ++f) {
            const float xs = x[f] * a.c1;
            const f32x4 wf = wl[1 + f];
#pragma unroll
            for (int j = 0; j < OCC_NF; ++j) h[j] += wf[j] * xs;
        }
        float pre = b2[0];
#pragma unroll
        for (int j = 0; j < OCC_NF; ++j) {
            const float hj = h[j] > 0.f ? h[j] : 0.2f * h[j];
            pre += W2[j] * (hj * a.c2);
        }
        const float sv = 1.f / (1.f + __expf(-(pre - a.threshold)));
        run += sv;
        const float c = fminf(fmaxf(run, 0.f), 1.f);
        const float wd = c - prev;
        prev = c;
        s[base + (long)d * a.HW] = sv;
        w[base + (long)d * a.HW] = wd;
        dacc += depth_coord(d, a.D) * wd;
#pragma unroll
        for (int f = 0; f < F; ++f) acc[f] += wd * x[f];
    };
    load(xa, 0);
    for (int d = 0; d < a.D; d += 2) {          // D is even (checked by the launcher)
        load(xb, d + 1);
        step(xa, d);
        if (d + 2 < a.D) load(xa, d + 2);
        step(xb, d + 1);
    }
#pragma unroll
    for (int f = 0; f < F; ++f) feat[((long)b * F + f) * a.HW + p] = acc[f];
    depth[ray] = ((dacc + 0.5f) * depth_scale) + near_plane;
}

// backward 1: dw[b,d,p] = sum_f dfeat[b,f,p] * vol[b,f,d,p] + ddepth'[b,p] * coord(d)
__global__ __launch_bounds__(256) void occ_bwd_dw_kernel(OccArgs a, const float* __restrict__ vol,
                                                         const float* __restrict__ dfeat,
                                                         const float* __restrict__ ddepth, float depth_scale,
                                                         float* __restrict__ dw) {
    const long vox = (long)a.D * a.HW;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)a.B * vox) return;
    const int b = (int)(i / vox);
    const long r = i - (long)b * vox;
    const int d = (int)(r / a.HW);
    const int p = (int)(r - (long)d * a.HW);
    float acc = ddepth[(long)b * a.HW + p] * depth_scale * depth_coord(d, a.D);
    for (int f = 0; f < a.F; ++f) acc += dfeat[((long)b * a.F + f) * a.HW + p] * vol[((long)b * a.F + f) * vox + r];
    dw[i] = acc;
}

// backward 2 (per ray): dw -> ds through diff, clip (closed interval passes) and cumsum
__global__ __launch_bounds__(256) void occ_bwd_scan_kernel(OccArgs a, const float* __restrict__ s,
                                                           const float* __restrict__ dw, float* __restrict__ ds) {
    const long ray = (long)blockIdx.x * 256 + threadIdx.x;
    if (ray >= (long)a.B * a.HW) return;
    const int b = (int)(ray / a.HW);
    const int p = (int)(ray - (long)b * a.HW);
    const long base = (long)b * a.D * a.HW + p;
    // forward running sums are needed for the clip mask: recompute them into ds first
    float run = 0.f;
    for (int d = 0; d < a.D; ++d) {
        run += s[base + (long)d * a.HW];
        ds[base + (long)d * a.HW] = run;
    }
    float next_dw = 0.f, acc = 0.f;
    for (int d = a.D - 1; d >= 0; --d) {
        const float cur = dw[base + (long)d * a.HW];
        const float dc = cur - next_dw;                 // c[d] enters w[d] (+) and w[d+1] (-)
        next_dw = cur;
        const float cs = ds[base + (long)d * a.HW];
        if (cs >= 0.f && cs <= 1.f) acc += dc;          // clip gradient
        ds[base + (long)d * a.HW] = acc;                // reverse cumulative sum
    }
}

// backward 3: through the per-voxel MLP; dvol = w * dfeat (compositing) + MLP path, and the 141 parameter gradients
// (dW1 nf x (F+1), db1 nf, dW2 nf, db2 1 -> packed in dparams).
// A block is 64 voxels x 4 feature groups: wave g owns features 8g .. 8g+7 of the block's 64 consecutive voxels (one
// coalesced 256-byte run per feature plane), so a thread carries 32 + 13 gradient accumulators instead of 141 -- the
// one-thread-per-voxel version sat at 255 registers, two waves per SIMD, and took 790 us for 1.2 GB of traffic.  The hidden
// pre-activations need all 32 features: the four partial dot products meet in LDS (double buffered, one barrier per
// 64-voxel chunk).  Blocks are persistent; each leaves its 141 sums in a row of `partial` with plain stores and
// occ_bwd_params_kernel adds the rows in index order: no atomics, the parameter gradients are bit-reproducible.
constexpr int OCC_FG = 8;
constexpr int OCC_NPARAM_MAX = OCC_NF * (OCC_MAXF + 1) + 2 * OCC_NF + 1;
__global__ __launch_bounds__(256) void occ_bwd_mlp_kernel(OccArgs a, const float* __restrict__ vol,
                                                          const float* __restrict__ W1, const float* __restrict__ b1,
                                                          const float* __restrict__ W2, const float* __restrict__ s,
                                                          const float* __restrict__ ds, const float* __restrict__ w,
                                                          const float* __restrict__ dfeat, float* __restrict__ dvol,
                                                          float* __restrict__ partial) {
    __shared__ float hp[2][4][OCC_NF][64];
    __shared__ float red[OCC_NPARAM_MAX];
    const int v = threadIdx.x & 63;
    const int fg = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), f0 = fg * OCC_FG;   // wave-uniform: W1's slice sits in SGPRs
    const long vox = (long)a.D * a.HW;
    const long total = (long)a.B * vox;
    const int F1 = a.F + 1, nW1 = OCC_NF * F1;
    float w1[OCC_NF][OCC_FG], w10[OCC_NF], bb[OCC_NF], w2[OCC_NF];
    float g[OCC_NF][OCC_FG], g0[OCC_NF], gb1[OCC_NF], gW2[OCC_NF], gb2 = 0.f;
#pragma unroll
    for (int j = 0; j < OCC_NF; ++j) {
        w10[j] = W1[j * F1]; bb[j] = b1[j]; w2[j] = W2[j] * a.c2;
        g0[j] = 0.f; gb1[j] = 0.f; gW2[j] = 0.f;
#pragma unroll
        for (int k = 0; k < OCC_FG; ++k) {
            w1[j][k] = f0 + k < a.F ? W1[j * F1 + 1 + f0 + k] : 0.f;
            g[j][k] = 0.f;
        }
    }
    const long nchunks = (total + 63) / 64;
    int buf = 0;
    for (long c = blockIdx.x; c < nchunks; c += gridDim.x, buf ^= 1) {      // block-uniform trip count
        const long i = c * 64 + v;
        const bool live = i < total;
        const long ii = live ? i : total - 1;
        const int b = (int)(ii / vox);
        const long r = ii - (long)b * vox;
        const int d = (int)(r / a.HW);
        const int p = (int)(r - (long)d * a.HW);
        float x[OCC_FG], df[OCC_FG];
#pragma unroll
        for (int k = 0; k < OCC_FG; ++k) {
            const bool ok = f0 + k < a.F;
            const long plane = (long)b * a.F + (ok ? f0 + k : 0);
            x[k] = ok ? vol[plane * vox + r] * a.c1 : 0.f;
            df[k] = ok ? dfeat[plane * a.HW + p] : 0.f;
        }
        const float sv = s[ii], dsv = ds[ii], wd = w[ii];
#pragma unroll
        for (int j = 0; j < OCC_NF; ++j) {
            float t = 0.f;
#pragma unroll
            for (int k = 0; k < OCC_FG; ++k) t += w1[j][k] * x[k];
            hp[buf][fg][j][v] = t;
        }
        __syncthreads();
        const float xc = depth_coord(d, a.D) * a.c1;
        const float dpre2 = live ? dsv * sv * (1.f - sv) : 0.f;
        float dpre1[OCC_NF];
#pragma unroll
        for (int j = 0; j < OCC_NF; ++j) {
            const float h = (bb[j] + w10[j] * xc) + ((hp[buf][0][j][v] + hp[buf][1][j][v]) + (hp[buf][2][j][v] + hp[buf][3][j][v]));
            const float ha = h > 0.f ? h : 0.2f * h;
            dpre1[j] = w2[j] * dpre2 * (h > 0.f ? 1.f : 0.2f);
            g0[j] += dpre1[j] * xc;              // (these four are read from wave 0 only)
            gb1[j] += dpre1[j];
            gW2[j] += dpre2 * a.c2 * ha;
        }
        gb2 += dpre2;
#pragma unroll
        for (int k = 0; k < OCC_FG; ++k) {
            float dx = 0.f;
#pragma unroll
            for (int j = 0; j < OCC_NF; ++j) {
                dx += w1[j][k] * dpre1[j];
                g[j][k] += dpre1[j] * x[k];
            }
            if (live && f0 + k < a.F) dvol[((long)b * a.F + f0 + k) * vox + r] = wd * df[k] + dx * a.c1;
        }
    }
    // every parameter has ONE owner wave: plain stores into the block's LDS row, then the block's row of `partial`
#pragma unroll
    for (int j = 0; j < OCC_NF; ++j) {
#pragma unroll
        for (int k = 0; k < OCC_FG; ++k) {
            const float t = wave_sum(g[j][k]);
            if (v == 0 && f0 + k < a.F) red[j * F1 + 1 + f0 + k] = t;
        }
        const float t0 = wave_sum(g0[j]), t1 = wave_sum(gb1[j]), t2 = wave_sum(gW2[j]);
        if (v == 0 && fg == 0) { red[j * F1] = t0; red[nW1 + j] = t1; red[nW1 + OCC_NF + j] = t2; }
    }
    {
        const float t3 = wave_sum(gb2);
        if (v == 0 && fg == 0) red[nW1 + 2 * OCC_NF] = t3;
    }
    __syncthreads();
    const int nparams = nW1 + 2 * OCC_NF + 1;
    for (int t = threadIdx.x; t < nparams; t += 256) partial[(long)blockIdx.x * nparams + t] = red[t];
}

// The same pass with FOUR consecutive voxels per thread (HW % 4 == 0): a wave reads and writes 1-KB runs of a feature plane with
// 16-byte accesses instead of 256-byte runs with 4-byte ones, and a workgroup meets at its barrier once per 256 voxels instead of once
// per 64 -- the one-voxel form moved its 620 MB at 1.9 TB/s (section 3 of DESIGN.md, round 6).  Same expressions per voxel; the
// parameter-gradient sums are formed in another (still fixed) order.
__global__ __launch_bounds__(256) void occ_bwd_mlp4_kernel(OccArgs a, const float* __restrict__ vol,
                                                           const float* __restrict__ W1, const float* __restrict__ b1,
                                                           const float* __restrict__ W2, const float* __restrict__ s,
                                                           const float* __restrict__ ds, const float* __restrict__ w,
                                                           const float* __restrict__ dfeat, float* __restrict__ dvol,
                                                           float* __restrict__ partial) {
    __shared__ f32x4 hp[2][4][OCC_NF][64];
    __shared__ float red[OCC_NPARAM_MAX];
    const int v = threadIdx.x & 63;
    const int fg = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), f0 = fg * OCC_FG;
    const long vox = (long)a.D * a.HW;
    const long total = (long)a.B * vox;
    const int F1 = a.F + 1, nW1 = OCC_NF * F1;
    float w1[OCC_NF][OCC_FG], w10[OCC_NF], bb[OCC_NF], w2[OCC_NF];
    float g[OCC_NF][OCC_FG], g0[OCC_NF], gb1[OCC_NF], gW2[OCC_NF], gb2 = 0.f;
#pragma unroll
    for (int j = 0; j < OCC_NF; ++j) {
        w10[j] = W1[j * F1]; bb[j] = b1[j]; w2[j] = W2[j] * a.c2;
        g0[j] = 0.f; gb1[j] = 0.f; gW2[j] = 0.f;
#pragma unroll
        for (int k = 0; k < OCC_FG; ++k) {
            w1[j][k] = f0 + k < a.F ? W1[j * F1 + 1 + f0 + k] : 0.f;
            g[j][k] = 0.f;
        }
    }
    const long nchunks = (total + 255) / 256;
    int buf = 0;
    for (long c = blockIdx.x; c < nchunks; c += gridDim.x, buf ^= 1) {      // block-uniform trip count
        const long i = c * 256 + 4 * v;                                     // first of this thread's four voxels
        const bool live = i < total;                                        // (total % 4 == 0: all four or none)
        const long ii = live ? i : total - 4;
        const int b = (int)(ii / vox);
        const long r = ii - (long)b * vox;
        const int d = (int)(r / a.HW);
        const int p = (int)(r - (long)d * a.HW);
        f32x4 x[OCC_FG], df[OCC_FG];
#pragma unroll
        for (int k = 0; k < OCC_FG; ++k) {
            const bool ok = f0 + k < a.F;
            const long plane = (long)b * a.F + (ok ? f0 + k : 0);
            x[k] = *reinterpret_cast<const f32x4*>(vol + plane * vox + r) * (ok ? a.c1 : 0.f);
            df[k] = *reinterpret_cast<const f32x4*>(dfeat + plane * a.HW + p) * (ok ? 1.f : 0.f);
        }
        const f32x4 sv = *reinterpret_cast<const f32x4*>(s + ii), dsv = *reinterpret_cast<const f32x4*>(ds + ii),
                    wd = *reinterpret_cast<const f32x4*>(w + ii);
#pragma unroll
        for (int j = 0; j < OCC_NF; ++j) {
            f32x4 t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < OCC_FG; ++k) t += w1[j][k] * x[k];
            hp[buf][fg][j][v] = t;
        }
        __syncthreads();
        const float xc = depth_coord(d, a.D) * a.c1;
        f32x4 dpre1[OCC_NF];
        f32x4 dpre2;
#pragma unroll
        for (int e = 0; e < 4; ++e) dpre2[e] = live ? dsv[e] * sv[e] * (1.f - sv[e]) : 0.f;
#pragma unroll
        for (int j = 0; j < OCC_NF; ++j) {
            const f32x4 hsum = (hp[buf][0][j][v] + hp[buf][1][j][v]) + (hp[buf][2][j][v] + hp[buf][3][j][v]);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float h = (bb[j] + w10[j] * xc) + hsum[e];
                const float ha = h > 0.f ? h : 0.2f * h;
                const float d1 = w2[j] * dpre2[e] * (h > 0.f ? 1.f : 0.2f);
                dpre1[j][e] = d1;
                g0[j] += d1 * xc;                  // (these four are read from wave 0 only)
                gb1[j] += d1;
                gW2[j] += dpre2[e] * a.c2 * ha;
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) gb2 += dpre2[e];
#pragma unroll
        for (int k = 0; k < OCC_FG; ++k) {
            f32x4 dx = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < OCC_NF; ++j) {
                dx += w1[j][k] * dpre1[j];
#pragma unroll
                for (int e = 0; e < 4; ++e) g[j][k] += dpre1[j][e] * x[k][e];
            }
            if (live && f0 + k < a.F)
                *reinterpret_cast<f32x4*>(dvol + ((long)b * a.F + f0 + k) * vox + r) = wd * df[k] + dx * a.c1;
        }
    }
#pragma unroll
    for (int j = 0; j < OCC_NF; ++j) {
#pragma unroll
        for (int k = 0; k < OCC_FG; ++k) {
            const float t = wave_sum(g[j][k]);
            if (v == 0 && f0 + k < a.F) red[j * F1 + 1 + f0 + k] = t;
        }
        const float t0 = wave_sum(g0[j]), t1 = wave_sum(gb1[j]), t2 = wave_sum(gW2[j]);
        if (v == 0 && fg == 0) { red[j * F1] = t0; red[nW1 + j] = t1; red[nW1 + OCC_NF + j] = t2; }
    }
    {
        const float t3 = wave_sum(gb2);
        if (v == 0 && fg == 0) red[nW1 + 2 * OCC_NF] = t3;
    }
    __syncthreads();
    const int nparams = nW1 + 2 * OCC_NF + 1;
    for (int t = threadIdx.x; t < nparams; t += 256) partial[(long)blockIdx.x * nparams + t] = red[t];
}

// dparams[t] = sum over the rows of `partial` (one per block of occ_bwd_mlp_kernel), in row order
__global__ __launch_bounds__(256) void occ_bwd_params_kernel(const float* __restrict__ partial, int nrows, int nparams,
                                                             float* __restrict__ dparams) {
    __shared__ float red[4];
    const int t = blockIdx.x;
    float acc = 0.f;
    for (int r = threadIdx.x; r < nrows; r += 256) acc += partial[(long)r * nparams + t];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) dparams[t] = (red[0] + red[1]) + (red[2] + red[3]);
}

#ifdef RGBD_DEBUG_BUILD
bool g_occ_unfused = false;    // test hook (debug library): the three-kernel forward for F = 32 too (bit-identity A/B)
#else
constexpr bool g_occ_unfused = false;
#endif
}  // namespace

#ifdef RGBD_DEBUG_BUILD
extern "C" void rgbd_debug_occ_unfused(int on) { g_occ_unfused = on != 0; }
#endif

extern "C" int rgbd_proj_idcs(const float* cam2world, int B, int W, int H, int D, int G, float voxel_size,
                              float near_plane, float fx, float fy, float cx, float cy, int32_t* idx, float* coords,
                              int32_t* counts, int32_t* workspace, void* stream) {
    RGBD_REQUIRE(cam2world && idx && coords && counts && workspace, "rgbd_proj_idcs: null pointer");
    RGBD_REQUIRE(B > 0 && W > 0 && H > 0 && D > 0 && G > 0, "rgbd_proj_idcs: bad shape");
    const long N = (long)W * H * D;
    RGBD_REQUIRE(N < (1l << 24), "rgbd_proj_idcs: frustum must have fewer than 2^24 elements (float-exact indices)");
    FrustumArgs f{W, H, D, G, voxel_size, near_plane, fx, fy, cx, cy};
    const int nblocks = (int)((N + 255) / 256);
    hipStream_t st = (hipStream_t)stream;
    proj_count_kernel<<<dim3(nblocks, B), 256, 0, st>>>(f, cam2world, (int)N, workspace);
    RGBD_CHECK_LAUNCH("proj_count_kernel");
    proj_scan_kernel<<<B, 1024, 0, st>>>(workspace, nblocks, counts);
    RGBD_CHECK_LAUNCH("proj_scan_kernel");
    proj_write_kernel<<<dim3(nblocks, B), 256, 0, st>>>(f, cam2world, (int)N, workspace, idx, coords);
    RGBD_CHECK_LAUNCH("proj_write_kernel");
    return 0;
}

extern "C" int rgbd_trilinear_fwd(const float* grid, const int32_t* idx, const float* coords, const int32_t* counts,
                                  float* out, int B, int F, int G, int N, void* stream) {
    RGBD_REQUIRE(grid && idx && coords && counts && out, "rgbd_trilinear_fwd: null pointer");
    hipStream_t st = (hipStream_t)stream;
    if (rgbd_zero_async(out, (size_t)B * F * N * sizeof(float), st) != hipSuccess) {
        rgbd_set_error("rgbd_trilinear_fwd: zero fill failed");
        return -2;
    }
    trilinear_fwd_kernel<<<dim3((N + 255) / 256, B), 256, 0, st>>>(grid, idx, coords, counts, out, F, G, N);
    RGBD_CHECK_LAUNCH("trilinear_fwd_kernel");
    return 0;
}

extern "C" int rgbd_trilinear_fwd_fm(const float* grid_fm, const int32_t* idx, const float* coords, const int32_t* counts,
                                     float* out, int B, int F, int G, int N, void* stream) {
    RGBD_REQUIRE(grid_fm && idx && coords && counts && out, "rgbd_trilinear_fwd_fm: null pointer");
    RGBD_REQUIRE(B > 0 && F > 0 && F <= 32 && G > 0 && N > 0, "rgbd_trilinear_fwd_fm: needs 0 < F <= 32 (F=%d)", F);
    hipStream_t st = (hipStream_t)stream;
    if (rgbd_zero_async(out, (size_t)B * F * N * sizeof(float), st) != hipSuccess) {
        rgbd_set_error("rgbd_trilinear_fwd_fm: zero fill failed");
        return -2;
    }
    trilinear_fwd_fm_kernel<<<dim3((N + TRI_S - 1) / TRI_S, B), 256, 0, st>>>(grid_fm, idx, coords, counts, out, F, G, N);
    RGBD_CHECK_LAUNCH("trilinear_fwd_fm_kernel");
    return 0;
}

extern "C" int rgbd_trilinear_bwd_fm(const float* dout, const int32_t* idx, const float* coords, const int32_t* counts,
                                     float* dgrid_fm, int B, int F, int G, int N, void* stream) {
    RGBD_REQUIRE(dout && idx && coords && counts && dgrid_fm, "rgbd_trilinear_bwd_fm: null pointer");
    RGBD_REQUIRE(B > 0 && F > 0 && F <= 32 && G > 0 && N > 0, "rgbd_trilinear_bwd_fm: needs 0 < F <= 32 (F=%d)", F);
    hipStream_t st = (hipStream_t)stream;
    if (rgbd_zero_async(dgrid_fm, (size_t)B * G * G * G * F * sizeof(float), st) != hipSuccess) {
        rgbd_set_error("rgbd_trilinear_bwd_fm: zero fill failed");
        return -2;
    }
    trilinear_bwd_scatter_kernel<<<dim3((N + TRI_S - 1) / TRI_S, B), 256, 0, st>>>(dout, idx, coords, counts, dgrid_fm, F, G, N);
    RGBD_CHECK_LAUNCH("trilinear_bwd_scatter_kernel");
    return 0;
}

extern "C" int rgbd_trilinear_fwd_frustum(const float* grid_fm, const float* cam2world, int B, int F, int W, int H, int D, int G,
                                          float voxel_size, float near_plane, float fx, float fy, float cx, float cy, float* out,
                                          void* stream) {
    RGBD_REQUIRE(grid_fm && cam2world && out, "rgbd_trilinear_fwd_frustum: null pointer");
    RGBD_REQUIRE(B > 0 && F > 0 && F <= 32 && G > 0 && W > 0 && H > 0 && D > 0 && (long)W * H * D < (1l << 24),
                 "rgbd_trilinear_fwd_frustum: needs 0 < F <= 32 and fewer than 2^24 frustum elements (F=%d)", F);
    const int N = W * H * D;
    FrustumArgs f{W, H, D, G, voxel_size, near_plane, fx, fy, cx, cy};
    trilinear_fwd_frustum_kernel<<<dim3((N + TF_S - 1) / TF_S, B), 256, 0, (hipStream_t)stream>>>(f, cam2world, grid_fm, out, F, N);
    RGBD_CHECK_LAUNCH("trilinear_fwd_frustum_kernel");
    return 0;
}

extern "C" int rgbd_trilinear_bwd_frustum_supported(int W, int H, int D, int G, int F) {
    return W > 0 && H > 0 && D > 0 && W % TB_X == 0 && H % TB_Y == 0 && F > 0 && F <= 32 && G > 0 &&
           (long)G * G * G <= (1l << 20) && (long)W * H * D < (1l << 24);
}

extern "C" int rgbd_trilinear_bwd_frustum(const float* dout, const float* cam2world, int B, int F, int W, int H, int D,
                                          int G, float voxel_size, float near_plane, float fx, float fy, float cx, float cy,
                                          float* dgrid_fm, void* stream) {
    RGBD_REQUIRE(dout && cam2world && dgrid_fm, "rgbd_trilinear_bwd_frustum: null pointer");
    RGBD_REQUIRE(B > 0 && rgbd_trilinear_bwd_frustum_supported(W, H, D, G, F),
                 "rgbd_trilinear_bwd_frustum: needs W %% 16 == 0, H %% 8 == 0, F <= 32, G^3 <= 2^20 (W=%d H=%d D=%d G=%d F=%d)",
                 W, H, D, G, F);
    hipStream_t st = (hipStream_t)stream;
    if (rgbd_zero_async(dgrid_fm, (size_t)B * G * G * G * F * sizeof(float), st) != hipSuccess) {
        rgbd_set_error("rgbd_trilinear_bwd_frustum: zero fill failed");
        return -2;
    }
    FrustumArgs f{W, H, D, G, voxel_size, near_plane, fx, fy, cx, cy};
#ifdef RGBD_DEBUG_BUILD
    static const int ko = getenv("RGBD_DEBUG_TRIBRICK_KO") ? atoi(getenv("RGBD_DEBUG_TRIBRICK_KO")) : 0;
#else
    constexpr int ko = 0;
#endif
    const int nbricks = (W / TB_X) * (H / TB_Y) * ((D + TB_D - 1) / TB_D) * B;        // (W / 16 is even or the pairs straddle rows: harmless)
    int dev = 0, cus = 256;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
        cus = prop.multiProcessorCount;
    const int npairs = (nbricks + 1) / 2;
    const int grid = npairs < 2 * cus ? npairs : 2 * cus;
    trilinear_bwd_brick_kernel<<<(unsigned)grid, 256, 0, st>>>(f, cam2world, dout, dgrid_fm, F, W * H * D, nbricks, ko);
    RGBD_CHECK_LAUNCH("trilinear_bwd_brick_kernel");
    return 0;
}

extern "C" int rgbd_trilinear_bwd(const float* dout, const int32_t* idx, const float* coords, const int32_t* counts,
                                  float* dgrid, float* workspace, int B, int F, int G, int N, void* stream) {
    RGBD_REQUIRE(dout && idx && coords && counts && dgrid && workspace, "rgbd_trilinear_bwd: null pointer");
    RGBD_REQUIRE(B > 0 && F > 0 && F <= 32 && G > 0 && N > 0, "rgbd_trilinear_bwd: needs 0 < F <= 32 (F=%d)", F);
    hipStream_t st = (hipStream_t)stream;
    const long g3 = (long)G * G * G;
    if (rgbd_zero_async(workspace, (size_t)B * g3 * F * sizeof(float), st) != hipSuccess) {
        rgbd_set_error("rgbd_trilinear_bwd: zero fill failed");
        return -2;
    }
    trilinear_bwd_scatter_kernel<<<dim3((N + TRI_S - 1) / TRI_S, B), 256, 0, st>>>(dout, idx, coords, counts, workspace,
                                                                                   F, G, N);
    RGBD_CHECK_LAUNCH("trilinear_bwd_scatter_kernel");
    transpose_vf_kernel<<<dim3((unsigned)((g3 + 31) / 32), (F + 31) / 32, B), 256, 0, st>>>(workspace, dgrid, g3, F);
    RGBD_CHECK_LAUNCH("transpose_vf_kernel");
    return 0;
}

extern "C" int rgbd_occlusion_accum_fwd(const float* vol, const float* W1, const float* b1, const float* W2,
                                        const float* b2, float threshold, float voxel_size, float near_plane,
                                        float* s, float* w, float* feat, float* depth, int B, int F, int D, int HW,
                                        void* stream) {
    RGBD_REQUIRE(vol && W1 && b1 && W2 && b2 && s && w && feat && depth, "rgbd_occlusion_accum_fwd: null pointer");
    RGBD_REQUIRE(F > 0 && F <= OCC_MAXF && B > 0 && D > 0 && HW > 0, "rgbd_occlusion_accum_fwd: bad shape");
    OccArgs a{B, F, D, HW, sqrtf(2.f / (float)(F + 1)), sqrtf(2.f / (float)OCC_NF), threshold};
    hipStream_t st = (hipStream_t)stream;
    if (F == 32 && (D & 1) == 0 && (long)B * F * D * HW * 4 < 0x7fffffffL && !g_occ_unfused) {
        occ_fwd_fused_kernel<32><<<(unsigned)(((long)B * HW + 255) / 256), 256, 0, st>>>(a, vol, W1, b1, W2, b2, s, w, feat, depth,
                                                                                       (float)D * voxel_size, near_plane);
        RGBD_CHECK_LAUNCH("occ_fwd_fused_kernel");
        return 0;
    }
    const long nv = (long)B * D * HW;
    occ_score_kernel<<<(unsigned)((nv + 255) / 256), 256, 0, st>>>(a, vol, W1, b1, W2, b2, s);
    RGBD_CHECK_LAUNCH("occ_score_kernel");
    occ_scan_kernel<<<(unsigned)(((long)B * HW + 255) / 256), 256, 0, st>>>(a, s, w, depth, (float)D * voxel_size,
                                                                           near_plane);
    RGBD_CHECK_LAUNCH("occ_scan_kernel");
    occ_compose_kernel<<<(unsigned)(((long)B * F * HW + 255) / 256), 256, 0, st>>>(a, vol, w, feat);
    RGBD_CHECK_LAUNCH("occ_compose_kernel");
    return 0;
}

extern "C" int rgbd_occlusion_accum_bwd(const float* vol, const float* W1, const float* b1, const float* W2,
                                        const float* s, const float* w, const float* dfeat, const float* ddepth,
                                        float voxel_size, float* dw_ws, float* ds_ws, float* dvol, float* dparams,
                                        int B, int F, int D, int HW, void* stream) {
    RGBD_REQUIRE(vol && W1 && b1 && W2 && s && w && dfeat && ddepth && dw_ws && ds_ws && dvol && dparams,
                 "rgbd_occlusion_accum_bwd: null pointer");
    RGBD_REQUIRE(F > 0 && F <= OCC_MAXF && B > 0 && D > 0 && HW > 0, "rgbd_occlusion_accum_bwd: bad shape");
    OccArgs a{B, F, D, HW, sqrtf(2.f / (float)(F + 1)), sqrtf(2.f / (float)OCC_NF), 0.f};
    hipStream_t st = (hipStream_t)stream;
    const int nparams = OCC_NF * (F + 1) + OCC_NF + OCC_NF + 1;
    if (rgbd_zero_async(dparams, (size_t)((nparams + 3) / 4 * 4) * sizeof(float), st) != hipSuccess) {
        rgbd_set_error("rgbd_occlusion_accum_bwd: zero fill failed");
        return -2;
    }
    const long nv = (long)B * D * HW;
    occ_bwd_dw_kernel<<<(unsigned)((nv + 255) / 256), 256, 0, st>>>(a, vol, dfeat, ddepth, (float)D * voxel_size, dw_ws);
    RGBD_CHECK_LAUNCH("occ_bwd_dw_kernel");
    occ_bwd_scan_kernel<<<(unsigned)(((long)B * HW + 255) / 256), 256, 0, st>>>(a, s, dw_ws, ds_ws);
    RGBD_CHECK_LAUNCH("occ_bwd_scan_kernel");
    // dw_ws has been consumed by the scan: its first rows now take the blocks' parameter-gradient sums
    const long nchunks = (nv + 63) / 64;
    long nblocks = nchunks < 1024 ? nchunks : 1024;
    if (nblocks * nparams > nv) nblocks = nv / nparams;          // (tiny volumes: as many rows as dw_ws holds)
    RGBD_REQUIRE(nblocks >= 1, "rgbd_occlusion_accum_bwd: volume smaller than the parameter-gradient row (%ld < %d)", nv, nparams);
    if (HW % 4 == 0 && nv >= 1024 * 256 && ((uintptr_t)vol & 15) == 0 && ((uintptr_t)dvol & 15) == 0 && ((uintptr_t)s & 15) == 0 &&
        ((uintptr_t)ds_ws & 15) == 0 && ((uintptr_t)w & 15) == 0 && ((uintptr_t)dfeat & 15) == 0) {
        // four voxels per thread: 1-KB runs per feature plane, a quarter of the barriers (the step's volumes; small ones keep the
        // one-voxel form below, whose row count `nblocks` was sized for 64-voxel chunks)
        occ_bwd_mlp4_kernel<<<(unsigned)nblocks, 256, 0, st>>>(a, vol, W1, b1, W2, s, ds_ws, w, dfeat, dvol, dw_ws);
        RGBD_CHECK_LAUNCH("occ_bwd_mlp4_kernel");
    } else {
        occ_bwd_mlp_kernel<<<(unsigned)nblocks, 256, 0, st>>>(a, vol, W1, b1, W2, s, ds_ws, w, dfeat, dvol, dw_ws);
        RGBD_CHECK_LAUNCH("occ_bwd_mlp_kernel");
    }
    occ_bwd_params_kernel<<<nparams, 256, 0, st>>>(dw_ws, (int)nblocks, nparams, dparams);
    RGBD_CHECK_LAUNCH("occ_bwd_params_kernel");
    return 0;
}


// ------------------------------------------------------------------------------------------------ layout folds
// The DeepVoxels networks reach the 2-D conv engine through three data rearrangements (deepvoxels_generator.py): each is
// ONE launch forward and ONE backward here (gathers, no atomics) instead of pad + slice + cat chains of framework copies.
namespace {
// fold_depth_taps: x (B,D0,H,W,C) bf16 [read through a 2x depth repeat when up] -> y (B*D,H,W,3C):
//   y[b,d,h,w,k*C+c] = xs[b,d+k-1,h,w,c] (zero outside 0 <= d+k-1 < D), xs[d] = x[up ? d >> 1 : d].  16 bytes per thread.
__global__ __launch_bounds__(256) void fold_depth_fwd_kernel(const u32x4* __restrict__ x, u32x4* __restrict__ y, int B, int D0,
                                                             long HW, int C8, int up) {
    const int D = up ? 2 * D0 : D0;
    const long total = (long)B * D * HW * 3 * C8;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int c8 = (int)(e % C8);
        long r = e / C8;
        const int k = (int)(r % 3); r /= 3;
        const long p = r % HW; r /= HW;
        const int d = (int)(r % D);
        const int b = (int)(r / D);
        const int ds = d + k - 1;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (ds >= 0 && ds < D) v = x[(((long)b * D0 + (up ? ds >> 1 : ds)) * HW + p) * C8 + c8];
        y[e] = v;
    }
}
// adjoint: dx[b,d0,h,w,c] = sum over repeated depths ds of d0 and taps k of dy[b, ds - k + 1, h, w, k*C + c]
__global__ __launch_bounds__(256) void fold_depth_bwd_kernel(const u32x4* __restrict__ dy, u32x4* __restrict__ dx, int B, int D0,
                                                             long HW, int C8, int up) {
    const int D = up ? 2 * D0 : D0;
    const long total = (long)B * D0 * HW * C8;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int c8 = (int)(e % C8);
        long r = e / C8;
        const long p = r % HW; r /= HW;
        const int d0 = (int)(r % D0);
        const int b = (int)(r / D0);
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int rep = 0; rep < (up ? 2 : 1); ++rep) {
            const int ds = up ? 2 * d0 + rep : d0;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int d = ds - k + 1;
                if (d < 0 || d >= D) continue;
                const u32x4 v = dy[((((long)b * D + d) * HW + p) * 3 + k) * C8 + c8];
#pragma unroll
                for (int j = 0; j < 4; ++j) { acc[2 * j] += bf16_lo(v[j]); acc[2 * j + 1] += bf16_hi(v[j]); }
            }
        }
        dx[e] = u32x4{pack_bf16x2(acc[0], acc[1]), pack_bf16x2(acc[2], acc[3]), pack_bf16x2(acc[4], acc[5]),
                      pack_bf16x2(acc[6], acc[7])};
    }
}
// fold_4x4s2: x (B,H,W,C) -> y (B,H/2,W/2,16C): y[b,i,j,(ky*4+kx)*C+c] = x[b,2i+ky-1,2j+kx-1,c] (zero outside)
__global__ __launch_bounds__(256) void fold_4x4s2_fwd_kernel(const u32x4* __restrict__ x, u32x4* __restrict__ y, int B, int H, int W,
                                                             int C8) {
    const int Ho = H >> 1, Wo = W >> 1;
    const long total = (long)B * Ho * Wo * 16 * C8;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int c8 = (int)(e % C8);
        long r = e / C8;
        const int t = (int)(r % 16); r /= 16;
        const int j = (int)(r % Wo); r /= Wo;
        const int i = (int)(r % Ho);
        const int b = (int)(r / Ho);
        const int yy = 2 * i + (t >> 2) - 1, xx = 2 * j + (t & 3) - 1;
        u32x4 v = {0u, 0u, 0u, 0u};
        if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) v = x[(((long)b * H + yy) * W + xx) * C8 + c8];
        y[e] = v;
    }
}
// adjoint: dx[b,y,x,c] = sum over the (ky,kx) with 2i+ky-1 = y, 2j+kx-1 = x  (two ky of the right parity, two kx)
__global__ __launch_bounds__(256) void fold_4x4s2_bwd_kernel(const u32x4* __restrict__ dy, u32x4* __restrict__ dx, int B, int H,
                                                             int W, int C8) {
    const int Ho = H >> 1, Wo = W >> 1;
    const long total = (long)B * H * W * C8;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int c8 = (int)(e % C8);
        long r = e / C8;
        const int xx = (int)(r % W); r /= W;
        const int yy = (int)(r % H);
        const int b = (int)(r / H);
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int a2 = 0; a2 < 2; ++a2) {
            const int ky = ((yy + 1) & 1) + 2 * a2;          // ky = y + 1 (mod 2)
            const int i = (yy + 1 - ky) >> 1;
            if (i < 0 || i >= Ho || yy + 1 - ky < 0) continue;
#pragma unroll
            for (int b2 = 0; b2 < 2; ++b2) {
                const int kx = ((xx + 1) & 1) + 2 * b2;
                const int j = (xx + 1 - kx) >> 1;
                if (j < 0 || j >= Wo || xx + 1 - kx < 0) continue;
                const u32x4 v = dy[((((long)b * Ho + i) * Wo + j) * 16 + ky * 4 + kx) * C8 + c8];
#pragma unroll
                for (int q = 0; q < 4; ++q) { acc[2 * q] += bf16_lo(v[q]); acc[2 * q + 1] += bf16_hi(v[q]); }
            }
        }
        dx[e] = u32x4{pack_bf16x2(acc[0], acc[1]), pack_bf16x2(acc[2], acc[3]), pack_bf16x2(acc[4], acc[5]),
                      pack_bf16x2(acc[6], acc[7])};
    }
}
// pad_last: rows of C0 elements -> rows of C1 >= C0 elements (zero tail), or the adjoint slice (C1 < C0); 2- or 4-byte elements
template <typename T>
__global__ __launch_bounds__(256) void pad_last_kernel(const T* __restrict__ x, T* __restrict__ y, long rows, int C0, int C1) {
    const long total = rows * C1;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int c = (int)(e % C1);
        const long r = e / C1;
        y[e] = c < C0 ? x[r * C0 + c] : (T)0;
    }
}
int fold_blocks(long total) { return (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096); }
}  // namespace

extern "C" int rgbd_fold_depth_taps_bf16(const void* x, void* y, int B, int D0, int H, int W, int C, int upsample_depth,
                                         int adjoint, void* stream) {
    RGBD_REQUIRE(x && y && B > 0 && D0 > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, "rgbd_fold_depth_taps_bf16: bad arguments");
    const long HW = (long)H * W;
    const int D = upsample_depth ? 2 * D0 : D0;
    if (adjoint) {      // x = dy (B*D,H,W,3C), y = dx (B,D0,H,W,C)
        fold_depth_bwd_kernel<<<fold_blocks((long)B * D0 * HW * (C / 8)), 256, 0, (hipStream_t)stream>>>(
            (const u32x4*)x, (u32x4*)y, B, D0, HW, C / 8, upsample_depth ? 1 : 0);
    } else {
        fold_depth_fwd_kernel<<<fold_blocks((long)B * D * HW * 3 * (C / 8)), 256, 0, (hipStream_t)stream>>>(
            (const u32x4*)x, (u32x4*)y, B, D0, HW, C / 8, upsample_depth ? 1 : 0);
    }
    RGBD_CHECK_LAUNCH("fold_depth_kernel");
    return 0;
}

extern "C" int rgbd_fold_4x4s2_bf16(const void* x, void* y, int B, int H, int W, int C, int adjoint, void* stream) {
    RGBD_REQUIRE(x && y && B > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && C > 0 && C % 8 == 0,
                 "rgbd_fold_4x4s2_bf16: bad arguments");
    if (adjoint) fold_4x4s2_bwd_kernel<<<fold_blocks((long)B * H * W * (C / 8)), 256, 0, (hipStream_t)stream>>>(
                     (const u32x4*)x, (u32x4*)y, B, H, W, C / 8);
    else fold_4x4s2_fwd_kernel<<<fold_blocks((long)B * (H / 2) * (W / 2) * 16 * (C / 8)), 256, 0, (hipStream_t)stream>>>(
             (const u32x4*)x, (u32x4*)y, B, H, W, C / 8);
    RGBD_CHECK_LAUNCH("fold_4x4s2_kernel");
    return 0;
}

extern "C" int rgbd_pad_last(const void* x, void* y, int64_t rows, int C0, int C1, int elem_bytes, void* stream) {
    RGBD_REQUIRE(x && y && rows > 0 && C0 > 0 && C1 > 0 && (elem_bytes == 2 || elem_bytes == 4), "rgbd_pad_last: bad arguments");
    const int blocks = fold_blocks(rows * C1);
    if (elem_bytes == 2) pad_last_kernel<unsigned short><<<blocks, 256, 0, (hipStream_t)stream>>>(
                             (const unsigned short*)x, (unsigned short*)y, rows, C0, C1);
    else pad_last_kernel<unsigned int><<<blocks, 256, 0, (hipStream_t)stream>>>((const unsigned int*)x, (unsigned int*)y, rows, C0, C1);
    RGBD_CHECK_LAUNCH("pad_last_kernel");
    return 0;
}


// ------------------------------------------------------------------------------------------------ weight folds
// Master parameter (reference shape) -> the (Cop, Cfp, KH, KW) fp32 weight the 2-D conv engine packs, and the adjoint
// (gradient of the folded weight -> gradient of the master).  The maps are injective, so both directions are gathers.
//   mode 0: (Co,Ci,3,3,3) -> (Cop, 3*Cip, 3, 3), input channel kd*Cip + ci   (3x3x3 conv over depth slices)
//   mode 1: (Co,Ci,4,4)   -> (Cop, 16*Cip, 1, 1), input channel (ky*4+kx)*Cip + ci   (4x4 stride-2 conv as a 1x1)
//   mode 2: (Co,Ci,K,K)   -> (Cop, Cip, K, K)   (channel padding only)
namespace {
struct FoldW { int mode, Co, Ci, K, Cop, Cip; };
__device__ __forceinline__ long fold_w_master_index(const FoldW& f, long e) {      // folded element -> master element or -1
    if (f.mode == 0) {
        const int kw = (int)(e % 3), kh = (int)(e / 3 % 3);
        const int cf = (int)(e / 9 % (3 * f.Cip)), co = (int)(e / 9 / (3 * f.Cip));
        const int kd = cf / f.Cip, ci = cf - kd * f.Cip;
        return (co < f.Co && ci < f.Ci) ? ((((long)co * f.Ci + ci) * 3 + kd) * 3 + kh) * 3 + kw : -1;
    }
    if (f.mode == 1) {
        const int cf = (int)(e % (16 * f.Cip)), co = (int)(e / (16 * f.Cip));
        const int t = cf / f.Cip, ci = cf - t * f.Cip;
        return (co < f.Co && ci < f.Ci) ? (((long)co * f.Ci + ci) * 4 + (t >> 2)) * 4 + (t & 3) : -1;
    }
    const int kk = f.K * f.K;
    const int k = (int)(e % kk), ci = (int)(e / kk % f.Cip), co = (int)(e / kk / f.Cip);
    return (co < f.Co && ci < f.Ci) ? ((long)co * f.Ci + ci) * kk + k : -1;
}
__device__ __forceinline__ long fold_w_folded_index(const FoldW& f, long m) {      // master element -> folded element
    if (f.mode == 0) {
        const int kw = (int)(m % 3), kh = (int)(m / 3 % 3), kd = (int)(m / 9 % 3);
        const int ci = (int)(m / 27 % f.Ci), co = (int)(m / 27 / f.Ci);
        return (((long)co * 3 * f.Cip + kd * f.Cip + ci) * 3 + kh) * 3 + kw;
    }
    if (f.mode == 1) {
        const int kx = (int)(m % 4), ky = (int)(m / 4 % 4), ci = (int)(m / 16 % f.Ci), co = (int)(m / 16 / f.Ci);
        return (long)co * 16 * f.Cip + (ky * 4 + kx) * f.Cip + ci;
    }
    const int kk = f.K * f.K;
    const int k = (int)(m % kk), ci = (int)(m / kk % f.Ci), co = (int)(m / kk / f.Ci);
    return ((long)co * f.Cip + ci) * kk + k;
}
__global__ __launch_bounds__(256) void fold_weight_kernel(const float* __restrict__ src, float* __restrict__ dst, FoldW f,
                                                          long n_folded, long n_master, int adjoint) {
    const long total = adjoint ? n_master : n_folded;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        if (adjoint) {
            const float v = src[fold_w_folded_index(f, e)];
            dst[e] = adjoint == 2 ? dst[e] + v : v;       // 2: accumulate into the master's gradient buffer
        } else {
            const long m = fold_w_master_index(f, e);
            dst[e] = m >= 0 ? src[m] : 0.f;
        }
    }
}
}  // namespace

// Up to FOLD_MULTI_MAX folds (or adjoints) in ONE launch: the DeepVoxels generator has 18 derived layers, whose folds are a few
// microseconds each and were a launch each, on every rebuild of the weight images and again (adjoint) behind every backward
// pass.  Descriptors by value in the kernel arguments; blocks are dealt out by element count.
namespace {
constexpr int FOLD_MULTI_MAX = 32;
struct FoldMultiArgs {
    const float* src[FOLD_MULTI_MAX];
    float* dst[FOLD_MULTI_MAX];
    FoldW f[FOLD_MULTI_MAX];
    long n_folded[FOLD_MULTI_MAX], n_master[FOLD_MULTI_MAX];
    int adjoint[FOLD_MULTI_MAX];
    int block_begin[FOLD_MULTI_MAX + 1];
    int n;
};
__global__ __launch_bounds__(256) void fold_weight_multi_kernel(FoldMultiArgs m) {
    int i = 0;
    while (i + 1 < m.n && (int)blockIdx.x >= m.block_begin[i + 1]) ++i;          // block-uniform scan
    const FoldW f = m.f[i];
    const int adjoint = m.adjoint[i];
    const long total = adjoint ? m.n_master[i] : m.n_folded[i];
    const float* __restrict__ src = m.src[i];
    float* __restrict__ dst = m.dst[i];
    const int nb = m.block_begin[i + 1] - m.block_begin[i];
    for (long e = (long)((int)blockIdx.x - m.block_begin[i]) * 256 + threadIdx.x; e < total; e += (long)nb * 256) {
        if (adjoint) {
            const float v = src[fold_w_folded_index(f, e)];
            dst[e] = adjoint == 2 ? dst[e] + v : v;
        } else {
            const long mi = fold_w_master_index(f, e);
            dst[e] = mi >= 0 ? src[mi] : 0.f;
        }
    }
}
}  // namespace

extern "C" int rgbd_fold_weight_multi_f32(const rgbd_fold_desc* descs, int n, void* stream) {
    RGBD_REQUIRE(descs && n > 0, "rgbd_fold_weight_multi_f32: no descriptors");
    for (int g0 = 0; g0 < n; g0 += FOLD_MULTI_MAX) {
        FoldMultiArgs m;
        m.n = n - g0 < FOLD_MULTI_MAX ? n - g0 : FOLD_MULTI_MAX;
        int blocks = 0;
        for (int i = 0; i < m.n; ++i) {
            const rgbd_fold_desc& d = descs[g0 + i];
            RGBD_REQUIRE(d.src && d.dst && d.mode >= 0 && d.mode <= 2 && d.Co > 0 && d.Ci > 0 && d.Cop >= d.Co && d.Cip >= d.Ci &&
                         d.K > 0 && (d.mode != 0 || d.K == 3) && (d.mode != 1 || d.K == 4) && d.adjoint >= 0 && d.adjoint <= 2,
                         "rgbd_fold_weight_multi_f32: bad descriptor %d", g0 + i);
            m.src[i] = d.src; m.dst[i] = d.dst; m.adjoint[i] = d.adjoint;
            m.f[i] = FoldW{d.mode, d.Co, d.Ci, d.K, d.Cop, d.Cip};
            m.n_master[i] = (long)d.Co * d.Ci * (d.mode == 0 ? 27 : d.K * d.K);
            m.n_folded[i] = d.mode == 0 ? (long)d.Cop * 3 * d.Cip * 9 : d.mode == 1 ? (long)d.Cop * 16 * d.Cip
                                                                                     : (long)d.Cop * d.Cip * d.K * d.K;
            const long total = d.adjoint ? m.n_master[i] : m.n_folded[i];
            long nb = (total + 1023) / 1024;              // four elements per thread
            if (nb > 256) nb = 256;
            m.block_begin[i] = blocks;
            blocks += (int)nb;
        }
        m.block_begin[m.n] = blocks;
        fold_weight_multi_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(m);
        RGBD_CHECK_LAUNCH("fold_weight_multi_kernel");
    }
    return 0;
}

extern "C" int rgbd_fold_weight_f32(const float* src, float* dst, int mode, int Co, int Ci, int K, int Cop, int Cip, int adjoint,
                                    void* stream) {
    RGBD_REQUIRE(src && dst && mode >= 0 && mode <= 2 && Co > 0 && Ci > 0 && Cop >= Co && Cip >= Ci && K > 0,
                 "rgbd_fold_weight_f32: bad arguments");
    RGBD_REQUIRE(mode != 0 || K == 3, "rgbd_fold_weight_f32: mode 0 folds 3x3x3 kernels");
    RGBD_REQUIRE(mode != 1 || K == 4, "rgbd_fold_weight_f32: mode 1 folds 4x4 kernels");
    const FoldW f = {mode, Co, Ci, K, Cop, Cip};
    const long n_master = (long)Co * Ci * (mode == 0 ? 27 : K * K);
    const long n_folded = mode == 0 ? (long)Cop * 3 * Cip * 9 : mode == 1 ? (long)Cop * 16 * Cip : (long)Cop * Cip * K * K;
    fold_weight_kernel<<<fold_blocks(adjoint ? n_master : n_folded), 256, 0, (hipStream_t)stream>>>(src, dst, f, n_folded, n_master,
                                                                                                     adjoint);
    RGBD_CHECK_LAUNCH("fold_weight_kernel");
    return 0;
}
