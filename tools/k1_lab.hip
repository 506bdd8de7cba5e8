// tools/k1_lab.hip -- development harness (not part of the product): times reference copy kernels and
// variants of the project+label kernel (rows per wave, resident blocks per CU) back to back in one process on one GPU.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -o tools/k1_lab tools/k1_lab.hip
#include "../lidar_object_detection_amd/csrc/lpf_kernels.hip.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// ---- platform reference points: plain streaming kernels of the same byte shape ----
template <int ROWS>
__global__ __launch_bounds__(256) void ref_copy16(const float4 *__restrict__ in, float4 *__restrict__ out, int n)
{
    const int base = blockIdx.x * 256 * ROWS + threadIdx.x;
    float4 v[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) v[r] = in[min(base + r * 256, n - 1)];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) if (base + r * 256 < n) out[base + r * 256] = v[r];
}
template <int ROWS>
__global__ __launch_bounds__(256) void ref_copy_16_8_4(const float4 *__restrict__ in, int2 *__restrict__ o1, uint32_t *__restrict__ o2, int n)
{
    const int base = blockIdx.x * 256 * ROWS + threadIdx.x;
    float4 v[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) v[r] = in[min(base + r * 256, n - 1)];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) if (base + r * 256 < n) {
        o1[base + r * 256] = make_int2(__float_as_int(v[r].x), __float_as_int(v[r].y));
        o2[base + r * 256] = __float_as_uint(v[r].z) ^ __float_as_uint(v[r].w);
    }
}
template <int ROWS>
__global__ __launch_bounds__(256) void ref_read16(const float4 *__restrict__ in, uint32_t *__restrict__ o2, int n)
{
    const int base = blockIdx.x * 256 * ROWS + threadIdx.x;
    float4 v[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) v[r] = in[min(base + r * 256, n - 1)];
    float acc = 0.f;
#pragma unroll
    for (int r = 0; r < ROWS; ++r) acc += v[r].x + v[r].y + v[r].z + v[r].w;
    if (acc == 1.2345e-30f) o2[base] = 1;
}

struct Variant { const char *name; void (*launch)(const LpfParams &, int nblk, hipStream_t); int rows; };

template <int ROWS, unsigned FL>
static void launch_t(const LpfParams &P, int nblk, hipStream_t s)
{
    hipLaunchKernelGGL((lpf_k1_project_t<ROWS, FL>), dim3(nblk * (P.seg_pts / (LPF_BLOCK * ROWS))), dim3(LPF_BLOCK), 0, s, P);
}

// The same tiles handed out through an atomic work counter to a resident grid -- the skeleton of a work-queue kernel
// that could take tail work items behind the tiles.  Measured at 16 M points against 84 us for the
// plain grid launch: 211 us with one tile per pop (a single address absorbs an atomic every ~13 ns), 108 / 111 / 138 us with
// 4 / 8 / 16 tiles per pop.  usage: k1_lab N NBUF ITERS [resident blocks] [tiles per pop]
template <int ROWS, unsigned FL, typename LT = uint32_t>
__global__ __launch_bounds__(LPF_BLOCK) void lpf_k1_persist_t(const LpfParams P, unsigned *queue, int ntiles, int grab)
{
    __shared__ unsigned s_cnt[LPF_TAB_ROWS];
    __shared__ int s_next;
    for (;;) {
        if (threadIdx.x == 0) s_next = (int)atomicAdd(queue, (unsigned)grab);     // `grab` consecutive tiles per pop
        __syncthreads();
        const int t0 = s_next;
        if (t0 >= ntiles) break;                            // block-uniform; every block reaches it
        for (int t = t0; t < min(t0 + grab, ntiles); ++t) {
            lpf_k1_tile<ROWS, FL, LT>(P, t, s_cnt);
            __syncthreads();
        }
    }
}

// work-queue form of the same kernel: a resident grid of `g_persist_blocks` blocks pops tiles from an atomic counter
static unsigned *g_queue = nullptr;
static int g_persist_blocks = 256 * 7, g_grab = 1;
template <int ROWS, unsigned FL>
static void launch_persist_t(const LpfParams &P, int nblk, hipStream_t s)
{
    const int ntiles = nblk * (P.seg_pts / (LPF_BLOCK * ROWS));
    (void)hipMemsetAsync(g_queue, 0, 4, s);
    hipLaunchKernelGGL((lpf_k1_persist_t<ROWS, FL>), dim3(g_persist_blocks), dim3(LPF_BLOCK), 0, s, P, g_queue, ntiles, g_grab);
}

int main(int argc, char **argv)
{
    const int N = argc > 1 ? atoi(argv[1]) : 2000000;
    const int NBUF = argc > 2 ? atoi(argv[2]) : 12;
    const int ITERS = argc > 3 ? atoi(argv[3]) : 240;
    const int W = 1408, H = 376;
    std::vector<float> h((size_t)N * 4);
    unsigned long long st = 12345;
    auto rnd = [&]() { st = st * 6364136223846793005ull + 1442695040888963407ull; return (double)(st >> 11) / 9007199254740992.0; };
    for (int i = 0; i < N; ++i) { h[4 * i] = (float)(rnd() * 160 - 80); h[4 * i + 1] = (float)(rnd() * 160 - 80); h[4 * i + 2] = (float)(rnd() * 6 - 3); h[4 * i + 3] = (float)rnd(); }
    std::vector<uint32_t> limg((size_t)W * H);
    for (size_t i = 0; i < limg.size(); ++i) limg[i] = (rnd() < 0.23) ? (1u << (int)(rnd() * 8)) : 0u;

    std::vector<float4 *> pts(NBUF); std::vector<int2 *> uv(NBUF); std::vector<uint32_t *> lab(NBUF);
    for (int b = 0; b < NBUF; ++b) {
        CK(hipMalloc(&pts[b], (size_t)N * 16)); CK(hipMalloc(&uv[b], (size_t)N * 8)); CK(hipMalloc(&lab[b], (size_t)N * 4));
        CK(hipMemcpy(pts[b], h.data(), (size_t)N * 16, hipMemcpyHostToDevice));
    }
    uint32_t *d_limg; CK(hipMalloc(&d_limg, limg.size() * 4)); CK(hipMemcpy(d_limg, limg.data(), limg.size() * 4, hipMemcpyHostToDevice));
    const size_t maxseg = 65536;
    unsigned long long *vbal, *mbal; uint4 *tab; LpfFrame *d_fr;
    CK(hipMalloc(&vbal, ((size_t)N / 64 + maxseg * 64) * 8)); CK(hipMalloc(&mbal, ((size_t)N / 64 + maxseg * 64) * 8));
    CK(hipMalloc(&tab, LPF_TAB_GROUPS * maxseg * 16)); CK(hipMemset(tab, 0, LPF_TAB_GROUPS * maxseg * 16)); CK(hipMalloc(&d_fr, sizeof(LpfFrame)));

    LpfParams P; memset(&P, 0, sizeof P);
    const double T[12] = {0.04304, -0.99905, -0.00691, 0.2631, -0.08838, 0.00308, -0.99608, -0.1031, 0.99516, 0.04348, -0.08816, -0.8295};
    const double K[9] = {552.554261, 0, 682.049453, 0, 552.554261, 238.769549, 0, 0, 1};
    memcpy(P.T, T, sizeof T); memcpy(P.K, K, sizeof K);
    P.dmin = 0; P.dmax = 30; P.W = W; P.H = H; P.F = 1; P.M = 8; P.frames = d_fr; P.label_img = d_limg;
    P.vbal = vbal; P.mbal = mbal; P.seg_tab = tab;
    {   // the other two counter levels K1's tiles add into
        uint4 *grp, *frm;
        CK(hipMalloc(&grp, LPF_TAB_GROUPS * (maxseg / LPF_GROUP_SEGS + 1) * 16)); CK(hipMemset(grp, 0, LPF_TAB_GROUPS * (maxseg / LPF_GROUP_SEGS + 1) * 16));
        CK(hipMalloc(&frm, LPF_FRM_SHARDS * LPF_TAB_GROUPS * 16)); CK(hipMemset(frm, 0, LPF_FRM_SHARDS * LPF_TAB_GROUPS * 16));
        P.grp_tab = grp; P.frm_tab = frm; P.ngrp_cap = (int)(maxseg / LPF_GROUP_SEGS + 1);
        float4 *ml; CK(hipMalloc(&ml, (size_t)N * 16)); P.mlist = ml;
    }

    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));

    const unsigned BEST = LPF_F_X4 | LPF_F_NTLOAD | LPF_F_NTSTORE;
    const Variant vars[] = {
        {"r4  x4 nt", launch_t<4, BEST>, 4},
        {"r2  x4 nt", launch_t<2, BEST>, 2},
        {"r4  x4 nt work-queue", launch_persist_t<4, BEST>, 4},
    };
    CK(hipMalloc(&g_queue, 4));
    if (argc > 4) g_persist_blocks = atoi(argv[4]);
    if (argc > 5) g_grab = atoi(argv[5]);
    {   // reference kernels
        float4 *o4; CK(hipMalloc(&o4, (size_t)N * 16));
        auto timeit = [&](const char *name, auto fn, double bytes) {
            for (int it = 0; it < 24; ++it) fn(it % NBUF);
            CK(hipStreamSynchronize(s)); CK(hipEventRecord(e0, s));
            for (int it = 0; it < ITERS; ++it) fn(it % NBUF);
            CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            const double us = 1e3 * ms / ITERS;
            printf("%-34s %10.2f us  %8.1f GB/s (actual bytes)\n", name, us, bytes / us / 1e3);
        };
        const int nb4 = (N + 1023) / 1024, nb8 = (N + 2047) / 2048, nb1 = (N + 255) / 256;
        timeit("ref read16 r4", [&](int b) { hipLaunchKernelGGL(ref_read16<4>, dim3(nb4), dim3(256), 0, s, pts[b], lab[b], N); }, 16.0 * N);
        timeit("ref read16 r8", [&](int b) { hipLaunchKernelGGL(ref_read16<8>, dim3(nb8), dim3(256), 0, s, pts[b], lab[b], N); }, 16.0 * N);
        timeit("ref copy16->16 r1", [&](int b) { hipLaunchKernelGGL(ref_copy16<1>, dim3(nb1), dim3(256), 0, s, pts[b], o4, N); }, 32.0 * N);
        timeit("ref copy16->16 r4", [&](int b) { hipLaunchKernelGGL(ref_copy16<4>, dim3(nb4), dim3(256), 0, s, pts[b], o4, N); }, 32.0 * N);
        timeit("ref copy16->16 r8", [&](int b) { hipLaunchKernelGGL(ref_copy16<8>, dim3(nb8), dim3(256), 0, s, pts[b], o4, N); }, 32.0 * N);
        timeit("ref copy16->8+4 r1", [&](int b) { hipLaunchKernelGGL(ref_copy_16_8_4<1>, dim3(nb1), dim3(256), 0, s, pts[b], uv[b], lab[b], N); }, 28.0 * N);
        timeit("ref copy16->8+4 r4", [&](int b) { hipLaunchKernelGGL(ref_copy_16_8_4<4>, dim3(nb4), dim3(256), 0, s, pts[b], uv[b], lab[b], N); }, 28.0 * N);
        timeit("ref copy16->8+4 r8", [&](int b) { hipLaunchKernelGGL(ref_copy_16_8_4<8>, dim3(nb8), dim3(256), 0, s, pts[b], uv[b], lab[b], N); }, 28.0 * N);
        CK(hipFree(o4));
    }
    {   // ---- occupancy sweep: how many resident blocks per CU does the kernel need to hold its bandwidth?  A dynamic-LDS
        //      pad caps the blocks a CU admits (160 KB / pad); tail kernels running beside K1 take wave slots the same way.
        const int nseg = (N + LPF_SEG_QUANTUM - 1) / LPF_SEG_QUANTUM;
        LpfFrame fr; memset(&fr, 0, sizeof fr); fr.N = N; fr.nseg = nseg;
        P.seg_pts = LPF_SEG_QUANTUM; P.nseg_total = nseg; P.nseg_cap = nseg; P.frame0 = fr; P.F = 1;
        printf("%-10s %6s %10s %10s\n", "rows", "blk/CU", "us/launch", "GB/s(28B)");
        auto sweep = [&](const char *name, auto kern, int rows) {
            for (int occ : {8, 6, 5, 4, 3, 2}) {
                const size_t pad = (size_t)(160 * 1024 / occ) - 1024;
                (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pad);
                const dim3 g(nseg * (LPF_SEG_QUANTUM / (LPF_BLOCK * rows)));
                for (int it = 0; it < 12; ++it) { P.pts = pts[it % NBUF]; P.uv = uv[it % NBUF]; P.label_bits = lab[it % NBUF]; hipLaunchKernelGGL(kern, g, dim3(LPF_BLOCK), pad, s, P); }
                CK(hipStreamSynchronize(s)); CK(hipEventRecord(e0, s));
                for (int it = 0; it < ITERS; ++it) { P.pts = pts[it % NBUF]; P.uv = uv[it % NBUF]; P.label_bits = lab[it % NBUF]; hipLaunchKernelGGL(kern, g, dim3(LPF_BLOCK), pad, s, P); }
                CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                const double us = 1e3 * ms / ITERS;
                printf("%-10s %6d %10.2f %10.1f\n", name, occ, us, 28.0 * N / us / 1e3);
            }
        };
        sweep("rows 2", lpf_k1_project_t<2, BEST, uint8_t>, 2);
        sweep("rows 4", lpf_k1_project_t<4, BEST, uint8_t>, 4);
        sweep("rows 8", lpf_k1_project_t<8, BEST, uint8_t>, 8);
        sweep("rows 16", lpf_k1_project_t<16, BEST, uint8_t>, 16);
    }
    const int segmul[] = {1};
    printf("%-34s %8s %8s %10s %10s\n", "variant", "seg_pts", "blocks", "us/launch", "GB/s(28B)");
    for (const Variant &v : vars) {
        for (int sm : segmul) {
            const int seg_pts = LPF_SEG_QUANTUM * sm;
            const int nseg = (N + seg_pts - 1) / seg_pts;
            if ((size_t)nseg > maxseg) continue;
            LpfFrame fr; memset(&fr, 0, sizeof fr); fr.N = N; fr.nseg = nseg;
            CK(hipMemcpy(d_fr, &fr, sizeof fr, hipMemcpyHostToDevice));
            P.seg_pts = seg_pts; P.nseg_total = nseg; P.nseg_cap = nseg; P.frame0 = fr;
            for (int it = 0; it < 24; ++it) { P.pts = pts[it % NBUF]; P.uv = uv[it % NBUF]; P.label_bits = lab[it % NBUF]; v.launch(P, nseg, s); }
            CK(hipStreamSynchronize(s));
            CK(hipEventRecord(e0, s));
            for (int it = 0; it < ITERS; ++it) { P.pts = pts[it % NBUF]; P.uv = uv[it % NBUF]; P.label_bits = lab[it % NBUF]; v.launch(P, nseg, s); }
            CK(hipEventRecord(e1, s));
            CK(hipStreamSynchronize(s));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            const double us = 1e3 * ms / ITERS;
            printf("%-34s %8d %8d %10.2f %10.1f\n", v.name, seg_pts, nseg, us, 28.0 * N / us / 1e3);
        }
    }
    return 0;
}
