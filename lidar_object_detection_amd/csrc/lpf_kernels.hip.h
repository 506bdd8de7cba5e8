// lpf_kernels.hip.h -- gfx950 (MI355X, wave64) kernels of the LiDAR projection +
// instance point-filter path.  Included by lpf_api.hip only.
//
// Kernel map (reference statements: /root/reference/Coding_testes, see include/lpf.h)
//   lpf_pack16 / lpf_pack_erode / lpf_erode_packed
//                      masks -> uint32 label image (bit m = mask m); 3x3-cross erosion
//                      on an LDS-staged tile of packed bits                             (V3:82-97, V3:222)
//   lpf_k1_project     float4 stream: 4x4 transform, cam2image, clip, label gather,
//                      per-row wave ballots + per-segment counters                      (V3:565-569, 584, 225)
//   lpf_k2_lists       ballots -> stable valid / per-instance index lists (wave prefix),
//                      masked points x boxes slab test -> integer counters              (V3:585, 228, 187-202, 370)
//   lpf_k3_finalize    first-strict-max box scan + per-frame summary                    (V3:353-379)
//
// Arithmetic: everything the reference computes in float64 is float64 here, with the
// summation order NumPy/OpenBLAS uses (see oracle/lpf_oracle.c); the file is compiled
// with -ffp-contract=off so only the fma() calls written below fuse.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define LPF_BLOCK 256            // 4 waves of 64
#define LPF_SEG_QUANTUM 4096     // points per segment = 64 ballot rows = one K2 wave; every K1 tile size divides it
#define LPF_MAX_MASKS_DEV 32     // = LPF_MAX_MASKS of include/lpf.h
#define LPF_TAB_ROWS 36          // counters per segment: 0 valid, 1 masked, 2+m instance m (34 used)
#define LPF_TAB_GROUPS 9         // stored as uint4 groups: counter c lives in group c>>2, component c&3

struct LpfFrame {                // one per frame, device + host copy
    long long pt_off;            // first point of the frame in the concatenated arrays
    long long inst_base;         // first entry of the frame in inst_idx
    int N;                       // points in the frame
    int seg_off;                 // first segment of the frame
    int nseg;                    // segments of the frame
    int box_off;                 // first box of the frame
    int B;                       // boxes of the frame
    int pad;                     // frame index (set by the host)
    long long cand_off;          // first word of the frame's candidate-box grid
    int cand_words;              // 64-bit words per grid cell = ceil(B / 64)
    int pad2;
};

struct LpfParams {
    double T[12];                // rows 0..2 of TrVeloToRect
    double K[9];                 // camera.K[:3,:3]
    double dmin, dmax;
    int W, H;
    int F, M;
    int seg_pts;                 // points per segment (= LPF_SEG_QUANTUM)
    int nseg_total;
    int nseg_cap;                // pitch (segments) of one seg_tab group
    int oriented;
    long long inst_cap;
    LpfFrame frame0;             // the frame table by value when F == 1 (no dependent load)
    const LpfFrame *frames;      // [F]
    const LpfFrame *segs;        // [nseg_total] the owning frame's record per segment (pad = frame id)
    const float4 *pts;
    const void *label_img;       // [F][H][W] label image (uint8 / uint16 / uint32 elements, see LT) or null
    const double *boxp;          // [Btot][16] exact box parameters
    const float *boxq;           // [Btot][8]  conservative float AABB {lo xyz, hi xyz}
    const unsigned long long *cand;   // per frame [cells][cand_words]: boxes whose accepted region can project into the cell
    int cell_w, cell_shift;      // cells per image row, log2(cell size in pixels)
    // outputs (nullable)
    int2 *uv;
    uint32_t *label_bits;
    double *depth, *uf, *vf;
    long long *valid_idx;
    int2 *uv_valid;              // compact (u, v) / labels of the valid points, in valid_idx order (needs uv / label_bits)
    uint32_t *label_valid;
    long long *inst_idx;
    int32_t *count_out;
    void *summary;               // lpf_frame_summary[F]
    // scratch
    unsigned long long *vbal, *mbal;   // one 64-bit ballot per 64 points
    uint4 *seg_tab;              // [LPF_TAB_GROUPS][nseg_cap] per-segment counters, 4 per uint4;
                                 // K1 tiles add into it, K3 leaves it zeroed
    uint4 *seg_pre;              // [LPF_TAB_GROUPS][nseg_cap] written by the scan (see lpf_scan_segments)
    unsigned *frame_tot;         // [F][LPF_TAB_ROWS] totals per frame
    unsigned *cnt;               // [M*Btot] inside counts (self-cleaned by K3)
    float4 *mlist;               // [Ntot] per K1 wave (64*ROWS points), at the wave's first slot: {x, y, z, label
                                 // bits} of its masked points in point order (K2 never gathers from the cloud)
    int tile_pts;                // points per K1 tile of this launch (4 waves)
    int inline_scan;             // 1: no scan kernel ran -- lpf_k2_block derives its prefixes from seg_tab, lpf_k3_finalize the
                                 // totals (and cleans seg_tab); only when every frame has <= 64 segments
};

__device__ __forceinline__ int lpf_lane() { return threadIdx.x & 63; }
__device__ __forceinline__ int lpf_wave() { return threadIdx.x >> 6; }

// Blocks are dealt round-robin to the 8 XCDs: give each XCD a contiguous run of
// segments so one frame's label image stays in one XCD's L2 (speed only).
__device__ __forceinline__ int lpf_xcd_remap(int b, int n)
{
    const int q = n >> 3, r = n & 7, x = b & 7, i = b >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}

__device__ __forceinline__ int lpf_find_frame(const LpfFrame *frames, int F, int sid)
{
    int lo = 0, hi = F;                      // last f with seg_off[f] <= sid
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (frames[mid].seg_off <= sid) lo = mid; else hi = mid;
    }
    return lo;
}

// int32 pixel convention of the ABI: saturate, NaN -> INT32_MIN.  v_cvt_i32_f64 saturates
// out-of-range inputs by itself (and gives 0 for NaN); r is already integral (rint).
__device__ __forceinline__ int32_t lpf_sat_i32(double r)
{
    int v;
    asm("v_cvt_i32_f64 %0, %1" : "=v"(v) : "v"(r));
    return (r != r) ? INT32_MIN : v;
}

// biased exponent field of a double (0 for zero/subnormal, 2047 for inf/nan)
__device__ __forceinline__ unsigned lpf_expo(double x) { return ((unsigned)__double2hiint(x) >> 20) & 0x7ffu; }

// (qx/ad, qy/ad), both correctly rounded.  The compiler's IEEE f64 division is
// div_scale -> rcp -> two Newton steps -> mul -> residual fma -> div_fmas -> div_fixup;
// when all operands are mid-range (2^-300 <= |x| < 2^301: biased exponent in [723, 1323]) the
// scale/fixup steps are identities, so the same arithmetic with ONE shared reciprocal gives the
// same bits for both quotients at about half the instructions.  Anything else (zeros,
// subnormals, inf, nan, huge ratios) takes the plain '/' operator.
__device__ __forceinline__ void lpf_div2(double qx, double qy, double ad, double &uf, double &vf)
{
    const unsigned ea = lpf_expo(ad), ex = lpf_expo(qx), ey = lpf_expo(qy);
    const unsigned lo = min(ea, min(ex, ey)), hi = max(ea, max(ex, ey));       // v_min3_u32 / v_max3_u32
    if (lo >= 723u && hi <= 1323u) {
        double r = __builtin_amdgcn_rcp(ad);
        double e = fma(-ad, r, 1.0); r = fma(r, e, r);
        e = fma(-ad, r, 1.0);        r = fma(r, e, r);
        double q = qx * r; double s = fma(-ad, q, qx); uf = fma(s, r, q);
        q = qy * r;        s = fma(-ad, q, qy);        vf = fma(s, r, q);
    } else {
        uf = qx / ad;
        vf = qy / ad;
    }
}

// One point through K1 + K2 of the reference (V3:565-568): rows of T and K as k-ordered fma
// chains (= OpenBLAS dgemm on these shapes), depth 0 -> -1e-6, x/|z| and y/|z|.
__device__ __forceinline__ void lpf_project_point(const LpfParams &P, float fx, float fy, float fz,
                                                  double &uf, double &vf, double &d)
{
    const double x = (double)fx, y = (double)fy, z = (double)fz;
    double cx = P.T[0] * x; cx = fma(P.T[1], y, cx); cx = fma(P.T[2],  z, cx); cx = cx + P.T[3];
    double cy = P.T[4] * x; cy = fma(P.T[5], y, cy); cy = fma(P.T[6],  z, cy); cy = cy + P.T[7];
    double cz = P.T[8] * x; cz = fma(P.T[9], y, cz); cz = fma(P.T[10], z, cz); cz = cz + P.T[11];
    double qx = P.K[0] * cx; qx = fma(P.K[1], cy, qx); qx = fma(P.K[2], cz, qx);
    double qy = P.K[3] * cx; qy = fma(P.K[4], cy, qy); qy = fma(P.K[5], cz, qy);
    d = P.K[6] * cx; d = fma(P.K[7], cy, d); d = fma(P.K[8], cz, d);
    if (d == 0.0) d = -1e-6;
    lpf_div2(qx, qy, fabs(d), uf, vf);
}

// ------------------------------------------------------------------------------------
// K1: one block = one tile of 256*ROWS consecutive points of one frame (a segment of 4096 points is 4 or 8
// tiles); a wave owns ROWS consecutive rows of 64 points.  All float4 loads of a lane are issued before the
// first use, the label gathers are issued as soon as a row's pixel is known, and only then do the ballots /
// label stores consume them.
// Algorithmic HBM bytes per point: 16 (xyzI) + 8 (u,v) + 4 (label) = 28, + 0.25 (ballots).
// ------------------------------------------------------------------------------------
// FL bits: production flags first, the LAB_* ones only exist for tools/k1_lab.hip ablations.
#define LPF_F_X4 1u          // keep the reflectance lane alive: 16-byte loads instead of 12
#define LPF_F_NTLOAD 2u      // nontemporal point loads
#define LPF_F_NTSTORE 4u     // nontemporal output stores
#define LPF_F_LAB_NOMATH 8u
#define LPF_F_LAB_NOGATHER 16u
#define LPF_F_LAB_NOSTORE 32u

#define LPF_F_LAB_NOBAL 64u
#define LPF_F_LAB_NOTAB 128u

template <int ROWS, unsigned FL, typename LT>
__device__ __forceinline__ void lpf_k1_tile(const LpfParams &P, const int blk, unsigned *s_cnt)
{
    // one block = one tile of 256*ROWS points; seg_pts / tile tiles share a K2 segment
    constexpr int TILE = LPF_BLOCK * ROWS;
    static_assert(LPF_SEG_QUANTUM % TILE == 0, "tiles must divide segments");
    const int tid = threadIdx.x, lane = lpf_lane(), wave = lpf_wave();
    const int tiles_per_seg = P.seg_pts / TILE;
    const int lb = lpf_xcd_remap(blk, P.nseg_total * tiles_per_seg);
    const int sid = lb / tiles_per_seg;
    LpfFrame fr = P.frame0;
    if (P.F > 1) fr = P.segs[__builtin_amdgcn_readfirstlane(sid)];        // wave-uniform: scalar loads, no search
    const int f = fr.pad;
    const int seg_start = (sid - fr.seg_off) * P.seg_pts;
    const int seg_end = min(seg_start + P.seg_pts, fr.N);
    const int c = seg_start + (lb - sid * tiles_per_seg) * TILE;       // first point of the tile
    if (c >= seg_end) return;                                            // padding tile of a short segment
    const float4 *__restrict__ pts = P.pts + fr.pt_off;
    const LT *__restrict__ limg =
        (P.label_img && P.M > 0) ? static_cast<const LT *>(P.label_img) + (size_t)f * (size_t)P.W * (size_t)P.H : nullptr;
    const int rows_per_seg = P.seg_pts >> 6;
    if (tid < LPF_TAB_ROWS) s_cnt[tid] = 0;
    __syncthreads();
    unsigned nvalid_w = 0, nmask_w = 0;

    // a wave owns ROWS consecutive rows of 64 points: its ballots form one contiguous run
    {
        const int wbase = c + wave * (ROWS * 64) + lane;
        float4 p[ROWS];
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            // clamp instead of branching: the loads issue back to back, waits are counted
            const float4 *src = pts + min(wbase + r * 64, seg_end - 1);
            if (FL & LPF_F_NTLOAD) {
                typedef float f4v __attribute__((ext_vector_type(4)));
                const f4v t = __builtin_nontemporal_load(reinterpret_cast<const f4v *>(src));
                p[r] = make_float4(t.x, t.y, t.z, t.w);
            } else {
                p[r] = *src;
            }
        }
        uint32_t lab[ROWS];
        bool valid[ROWS];
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const int idx = wbase + r * 64;
            const bool live = idx < seg_end;
            if (FL & LPF_F_X4) asm volatile("" ::"v"(p[r].w));
            double uf, vf, d, ru, rv;
            if (FL & LPF_F_LAB_NOMATH) {
                uf = (double)p[r].x; vf = (double)p[r].y; d = (double)p[r].z;
                ru = uf + 700.0; rv = vf + 100.0;
            } else {
                lpf_project_point(P, p[r].x, p[r].y, p[r].z, uf, vf, d);
                ru = rint(uf); rv = rint(vf);                   // np.round: half to even
            }
            // K3: clip.  ru, rv are integral: 0 <= ru < W  <=>  (unsigned)sat_i32(ru) < W  (saturation and
            // NaN -> INT32_MIN both land outside), so the image test runs on the integers (u, v)
            const int ui = lpf_sat_i32(ru), vi = lpf_sat_i32(rv);
            const bool ok = live && ((unsigned)ui < (unsigned)P.W) && ((unsigned)vi < (unsigned)P.H) &&
                            (d > P.dmin) && (d < P.dmax);
            valid[r] = ok;
            // K4: label gather (2.1 MB image, L2 resident); consumed after the loop
            lab[r] = 0;
            if (!(FL & LPF_F_LAB_NOGATHER)) {
                if (ok && limg) lab[r] = (uint32_t)limg[vi * P.W + ui];
            }
            if (live && !(FL & LPF_F_LAB_NOSTORE)) {
                const long long g = fr.pt_off + idx;
                if (P.uv) {
                    if (FL & LPF_F_NTSTORE) {
                        typedef int i2v __attribute__((ext_vector_type(2)));
                        i2v t; t.x = ui; t.y = vi;
                        __builtin_nontemporal_store(t, reinterpret_cast<i2v *>(P.uv + g));
                    } else {
                        P.uv[g] = make_int2(ui, vi);
                    }
                }
                if (P.depth) P.depth[g] = d;
                if (P.uf) P.uf[g] = uf;
                if (P.vf) P.vf[g] = vf;
            }
        }
        unsigned long long myv = 0, mym = 0;                // lane r keeps the ballots of row r
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const int idx = wbase + r * 64;
            uint32_t l = lab[r];
            if (idx < seg_end && P.label_bits && !(FL & LPF_F_LAB_NOSTORE)) {
                if (FL & LPF_F_NTSTORE) __builtin_nontemporal_store(l, P.label_bits + fr.pt_off + idx);
                else P.label_bits[fr.pt_off + idx] = l;
            }
            const unsigned long long vb = __ballot(valid[r]);
            const unsigned long long mb = __ballot(l != 0);
            if (lane == r) { myv = vb; mym = mb; }
            // the wave's masked points {x, y, z, label}, compacted in point order at the wave's own
            // first slots: K2 reads them back coalesced instead of gathering from the cloud
            if (l && P.mlist && !(FL & LPF_F_LAB_NOSTORE))
                P.mlist[fr.pt_off + (wbase - lane) + nmask_w + __popcll(mb & ((1ull << lane) - 1ull))] =
                    make_float4(p[r].x, p[r].y, p[r].z, __uint_as_float(l));
            nvalid_w += __popcll(vb);
            nmask_w += __popcll(mb);
            while (l) {                                     // rare: per-instance counts
                const int m = __ffs(l) - 1;
                l &= l - 1;
                atomicAdd(&s_cnt[2 + m], 1u);
            }
        }
        if (lane < ROWS && !(FL & LPF_F_LAB_NOBAL)) {       // one contiguous 8*ROWS-byte store per array
            const size_t row = (size_t)sid * rows_per_seg + ((c - seg_start) >> 6) + wave * ROWS + lane;
            P.vbal[row] = myv;
            P.mbal[row] = mym;
        }
    }
    if (lane == 0) { atomicAdd(&s_cnt[0], nvalid_w); atomicAdd(&s_cnt[1], nmask_w); }
    __syncthreads();
    if (!(FL & LPF_F_LAB_NOTAB)) {
        // integer adds into the segment's counters (order-independent, so still deterministic);
        // two 32-bit counters per 64-bit atomic: neither half can carry (each sum < 2^32)
        if (tid < (2 + P.M + 1) >> 1) {
            const unsigned long long v = (unsigned long long)s_cnt[2 * tid] | ((unsigned long long)s_cnt[2 * tid + 1] << 32);
            if (v)
                atomicAdd(reinterpret_cast<unsigned long long *>(P.seg_tab + (size_t)(tid >> 1) * P.nseg_cap + sid) + (tid & 1), v);
        }
    }
}

template <int ROWS, unsigned FL, typename LT = uint32_t>   // LT: label-image element (uint8 for M <= 8, uint16 for M <= 16)
__global__ __launch_bounds__(LPF_BLOCK) void lpf_k1_project_t(const LpfParams P)
{
    __shared__ unsigned s_cnt[LPF_TAB_ROWS];
    lpf_k1_tile<ROWS, FL, LT>(P, (int)blockIdx.x, s_cnt);
}

#define LPF_K1_FLAGS (LPF_F_X4 | LPF_F_NTLOAD | LPF_F_NTSTORE)

// ------------------------------------------------------------------------------------
// K6 helpers: membership of one point in one box, from precomputed box parameters
//   oriented: boxp = { c0[3], (v[3], vv) x 3, -, -, -, exact_ok }   (V3:187-202)
//   aabb    : boxp = { lo[3], hi[3] }                               (V3:158-162)
// The reference tests t = d / vv with 0 <= t <= 1.  For a positive normal vv and a d that
// is zero or not tiny, round-to-nearest division gives  t >= 0 <=> d >= 0  and
// t <= 1 <=> d <= vv  (the next double above vv is >= vv*(1+2^-53), which rounds above 1),
// so the quotient is only formed for degenerate boxes / denormal-range d.
// ------------------------------------------------------------------------------------
__device__ __forceinline__ bool lpf_oriented_inside(double px, double py, double pz, const double *__restrict__ b)
{
    const double rx = px - b[0], ry = py - b[1], rz = pz - b[2];
    const bool exact_ok = b[15] != 0.0;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const double v0 = b[3 + 4 * a], v1 = b[4 + 4 * a], v2 = b[5 + 4 * a], vv = b[6 + 4 * a];
        double d = v1 * ry; d = fma(v0, rx, d); d = fma(v2, rz, d);   // dgemv_t tail order
        bool in;
        if (exact_ok && !(fabs(d) < 1e-250 && d != 0.0)) {
            in = (d >= 0.0) && (d <= vv);
        } else {
            const double t = d / vv;
            in = (t >= 0.0) && (t <= 1.0);
        }
        if (!in) return false;
    }
    return true;
}

__device__ __forceinline__ bool lpf_aabb_inside(double px, double py, double pz, const double *__restrict__ b)
{
    return (px >= b[0]) && (px <= b[3]) && (py >= b[1]) && (py <= b[4]) && (pz >= b[2]) && (pz <= b[5]);
}

// ------------------------------------------------------------------------------------
// SCAN: one block per frame.  Turns the per-segment counters K1 accumulated into
//   seg_pre[g][seg] : exclusive prefix over the frame's earlier segments; for instance
//                     counters the frame-level list offset inst_off[m] is already added, so the
//                     value is the list position where the segment's first entry of mask m goes
//   frame_tot[f][c] : totals of the frame
// and leaves seg_tab zeroed for the next call (it reads every entry anyway).
// ------------------------------------------------------------------------------------
// block-wide exclusive offsets of per-thread sums (4 components); also returns the totals
__device__ __forceinline__ void lpf_block_excl4(const unsigned sum[4], unsigned excl[4], unsigned all[4],
                                                unsigned (*s_wsum)[4], int lane, int wave)
{
    unsigned inc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        unsigned x = sum[j];
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned t = __shfl_up(x, o);
            if (lane >= o) x += t;
        }
        inc[j] = x;
        if (lane == 63) s_wsum[wave][j] = x;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        unsigned wo = 0, a = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) { const unsigned t = s_wsum[w][j]; if (w < wave) wo += t; a += t; }
        excl[j] = wo + inc[j] - sum[j];
        all[j] = a;
    }
    __syncthreads();
}

template <int NG>   // NG > 0: at most NG groups and 1024 segments -> everything stays in registers
__global__ __launch_bounds__(LPF_BLOCK) void lpf_scan_segments_t(const LpfParams P)
{
    // thread t owns the contiguous run of R = ceil(nseg/256) segments starting at t*R.
    __shared__ unsigned s_toff[NG > 0 ? 1 : LPF_TAB_GROUPS][LPF_BLOCK][4];
    __shared__ unsigned s_wsum[4][4], s_tot[LPF_TAB_ROWS], s_off[LPF_TAB_ROWS];
    const int f = blockIdx.x, tid = threadIdx.x, lane = lpf_lane(), wave = lpf_wave();
    const LpfFrame fr = (P.F > 1) ? P.frames[f] : P.frame0;
    const int seg_lo = fr.seg_off, seg_hi = fr.seg_off + fr.nseg;
    const int ngroups = (2 + P.M + 3) >> 2;
    const int R = (NG > 0) ? 4 : (fr.nseg + LPF_BLOCK - 1) / LPF_BLOCK;
    const int my_lo = min(seg_lo + tid * R, seg_hi), my_hi = min(my_lo + R, seg_hi);

    if (NG > 0) {
        // ---- one memory round trip: all runs of all groups are loaded before the first use ----
        constexpr int G = NG > 0 ? NG : 1;
        uint4 q[G][4];
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int k = 0; k < 4; ++k)
                q[g][k] = (g < ngroups && my_lo + k < my_hi) ? P.seg_tab[(size_t)g * P.nseg_cap + my_lo + k] : make_uint4(0u, 0u, 0u, 0u);
        unsigned excl[G][4];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const unsigned sum[4] = {q[g][0].x + q[g][1].x + q[g][2].x + q[g][3].x, q[g][0].y + q[g][1].y + q[g][2].y + q[g][3].y,
                                     q[g][0].z + q[g][1].z + q[g][2].z + q[g][3].z, q[g][0].w + q[g][1].w + q[g][2].w + q[g][3].w};
            unsigned all[4];
            lpf_block_excl4(sum, excl[g], all, s_wsum, lane, wave);
            if (tid == 0) { s_tot[4 * g] = all[0]; s_tot[4 * g + 1] = all[1]; s_tot[4 * g + 2] = all[2]; s_tot[4 * g + 3] = all[3]; }
        }
        __syncthreads();
        if (tid < LPF_TAB_ROWS) {
            const bool used = tid < 4 * ngroups;
            unsigned off = 0;                              // inst_off[m] for the instance counters
            if (used) for (int c = 2; c < tid; ++c) off += s_tot[c];
            s_off[tid] = off;
            P.frame_tot[(size_t)f * LPF_TAB_ROWS + tid] = used ? s_tot[tid] : 0u;
        }
        __syncthreads();
#pragma unroll
        for (int g = 0; g < G; ++g) {
            if (g >= ngroups) break;
            unsigned run[4] = {s_off[4 * g] + excl[g][0], s_off[4 * g + 1] + excl[g][1], s_off[4 * g + 2] + excl[g][2], s_off[4 * g + 3] + excl[g][3]};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (my_lo + k < my_hi) {
                    P.seg_pre[(size_t)g * P.nseg_cap + my_lo + k] = make_uint4(run[0], run[1], run[2], run[3]);
                    P.seg_tab[(size_t)g * P.nseg_cap + my_lo + k] = make_uint4(0u, 0u, 0u, 0u);   // self-clean
                }
                run[0] += q[g][k].x; run[1] += q[g][k].y; run[2] += q[g][k].z; run[3] += q[g][k].w;
            }
        }
        return;
    }
    // ---- general shape: phase 1 sums each run, phase 2 re-reads the runs (L2-hot) ------------
    for (int g = 0; g < ngroups; ++g) {
        const uint4 *__restrict__ row = P.seg_tab + (size_t)g * P.nseg_cap;
        unsigned sum[4] = {0, 0, 0, 0};
#pragma unroll 8
        for (int sg = my_lo; sg < my_hi; ++sg) {
            const uint4 v = row[sg];
            sum[0] += v.x; sum[1] += v.y; sum[2] += v.z; sum[3] += v.w;
        }
        unsigned excl[4], all[4];
        lpf_block_excl4(sum, excl, all, s_wsum, lane, wave);
#pragma unroll
        for (int j = 0; j < 4; ++j) { s_toff[NG > 0 ? 0 : g][tid][j] = excl[j]; if (tid == 0) s_tot[4 * g + j] = all[j]; }
    }
    __syncthreads();
    if (tid < LPF_TAB_ROWS) {
        const bool used = tid < 4 * ngroups;
        unsigned off = 0;
        if (used) for (int c = 2; c < tid; ++c) off += s_tot[c];
        s_off[tid] = off;
        P.frame_tot[(size_t)f * LPF_TAB_ROWS + tid] = used ? s_tot[tid] : 0u;
    }
    __syncthreads();
    for (int g = 0; g < ngroups; ++g) {
        uint4 *__restrict__ row = P.seg_tab + (size_t)g * P.nseg_cap;
        uint4 *__restrict__ pre = P.seg_pre + (size_t)g * P.nseg_cap;
        unsigned run[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) run[j] = s_off[4 * g + j] + s_toff[NG > 0 ? 0 : g][tid][j];
#pragma unroll 8
        for (int sg = my_lo; sg < my_hi; ++sg) {
            const uint4 v = row[sg];
            pre[sg] = make_uint4(run[0], run[1], run[2], run[3]);
            row[sg] = make_uint4(0u, 0u, 0u, 0u);           // self-clean for the next call
            run[0] += v.x; run[1] += v.y; run[2] += v.z; run[3] += v.w;
        }
    }
}

// ------------------------------------------------------------------------------------
// K2: one WAVE = one segment of LPF_SEG_QUANTUM (4096) points = 64 ballot rows, one per lane; four
// independent waves per block, no block barriers.  Memory round trips on a wave's critical
// path: {segment record, ballots, prefixes} -> {labels + xyz of the masked points, box
// bounds} -> stores / atomics.
// ------------------------------------------------------------------------------------
#define LPF_K2_ROWS (LPF_SEG_QUANTUM / 64)
#define LPF_K2_WAVES 4
#define LPF_K2_LDSCNT 256         // LDS inside-counters: M * B up to this many (else one global atomic per hit)

__device__ __forceinline__ unsigned lpf_rl(unsigned v, int l) { return (unsigned)__builtin_amdgcn_readlane((int)v, l); }
__device__ __forceinline__ float lpf_rlf(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
__device__ __forceinline__ unsigned long long lpf_rl64(unsigned long long v, int l)
{
    return (unsigned long long)lpf_rl((unsigned)v, l) | ((unsigned long long)lpf_rl((unsigned)(v >> 32), l) << 32);
}
// per-lane source lane (ds_bpermute): every lane is active at the call sites
__device__ __forceinline__ unsigned long long lpf_rl64_var(unsigned long long v, int l)
{
    return (unsigned long long)(unsigned)__shfl((int)(unsigned)v, l) | ((unsigned long long)(unsigned)__shfl((int)(unsigned)(v >> 32), l) << 32);
}

// Set bits of 64 ballots (lane r holds row r, rowbase = bits set in earlier rows) -> ascending
// list in LDS.  Every lane walks its own row: work is O(set bits), not O(rows x 64).
__device__ __forceinline__ void lpf_bits_to_list(unsigned long long rowbits, unsigned rowbase, int lane, unsigned short *lst)
{
    unsigned pos = rowbase;
    const unsigned short tag = (unsigned short)(lane * 64);
    while (rowbits) {
        const int b = __ffsll((long long)rowbits) - 1;
        rowbits &= rowbits - 1ull;
        lst[pos++] = (unsigned short)(tag + b);
    }
}

#define LPF_F2_LAB_NOVALID 1u
#define LPF_F2_LAB_NOLIST 2u      // stop after valid_idx
#define LPF_F2_LAB_NOBOX 4u
#define LPF_F2_LAB_NOINST 8u
#define LPF_F2_LAB_NOEXACT 16u
#define LPF_F2_LAB_NOCAND 32u
#define LPF_F2_LAB_NOPROJ 64u

template <unsigned FL2>
__global__ __launch_bounds__(LPF_BLOCK) void lpf_k2_lists_t(const LpfParams P)
{
#ifdef LPF_LAB_SMALL_LIDX
    __shared__ unsigned short s_lidx[LPF_K2_WAVES][1024];   // LAB ONLY: sparse synthetic data
#else
    __shared__ unsigned short s_lidx[LPF_K2_WAVES][LPF_SEG_QUANTUM];   // valid, then masked points (segment-relative)
#endif
    __shared__ float4 s_pt[LPF_K2_WAVES][64];                          // xyz of the current 64 masked points
    __shared__ float4 s_bq[LPF_K2_WAVES][2 * 64];                      // {lo, hi} of the current <= 64 boxes
    __shared__ unsigned s_q[LPF_K2_WAVES][128];                        // (point, box) pairs that passed the float bounds
    __shared__ unsigned s_cnt[LPF_K2_WAVES][LPF_K2_LDSCNT];            // this wave's inside counts [M][B] when they fit
    const int lane = lpf_lane(), wave = lpf_wave();
    const int sid = blockIdx.x * LPF_K2_WAVES + wave;
    if (sid >= P.nseg_total) return;
    const unsigned long long lt = (1ull << lane) - 1ull;
    // ---- round trip 1: everything that only depends on sid -------------------------------
    LpfFrame fr = P.frame0;
    if (P.F > 1) fr = P.segs[__builtin_amdgcn_readfirstlane(sid)];       // wave-uniform: scalar loads, one branch
    const int ngroups = (2 + P.M + 3) >> 2;
    unsigned long long vb = 0, mb = 0;
    uint4 pre4 = make_uint4(0u, 0u, 0u, 0u);
    vb = P.vbal[(size_t)sid * LPF_K2_ROWS + lane];
    mb = P.mbal[(size_t)sid * LPF_K2_ROWS + lane];
    if (lane < ngroups) pre4 = P.seg_pre[(size_t)lane * P.nseg_cap + sid];

    const int seg_start = (sid - fr.seg_off) * LPF_SEG_QUANTUM;
    const int seg_end = min(seg_start + LPF_SEG_QUANTUM, fr.N);
    const int nrows = (seg_end - seg_start + 63) >> 6;
    if (lane >= nrows) { vb = 0; mb = 0; }                 // rows K1 never wrote
    const unsigned cv = __popcll(vb), cm = __popcll(mb);
    unsigned iv = cv, im = cm;
#pragma unroll
    for (int o = 1; o < LPF_K2_ROWS; o <<= 1) {            // inclusive scan over the 64 row counts
        const unsigned tv = __shfl_up(iv, o), tm = __shfl_up(im, o);
        if (lane >= o) { iv += tv; im += tm; }
    }
    const unsigned vbase = iv - cv, mbase = im - cm;
    const unsigned nv = lpf_rl(iv, LPF_K2_ROWS - 1), L = lpf_rl(im, LPF_K2_ROWS - 1);
    const long long run_v = (long long)lpf_rl(pre4.x, 0);

    // ---- valid_idx: ascending by construction (rows in order, lanes in order); staged in LDS so
    //      the HBM writes are whole 512-byte runs instead of a few bytes per row ---------------
    unsigned short *lst = s_lidx[wave];
    if (P.valid_idx && nv && !(FL2 & LPF_F2_LAB_NOVALID)) {
        lpf_bits_to_list(vb, vbase, lane, lst);
        __builtin_amdgcn_wave_barrier();
        long long *__restrict__ dst = P.valid_idx + fr.pt_off + run_v;
        for (unsigned e = lane; e < nv; e += 64) dst[e] = (long long)(seg_start + (int)lst[e]);
        if (P.uv_valid || P.label_valid) {                 // compact copies for callers that only read the valid points
            const long long o = fr.pt_off + run_v, g0 = fr.pt_off + seg_start;
            for (unsigned e = lane; e < nv; e += 64) {
                const long long g = g0 + (int)lst[e];
                if (P.uv_valid) P.uv_valid[o + e] = P.uv[g];
                if (P.label_valid) P.label_valid[o + e] = P.label_bits[g];
            }
        }
        __builtin_amdgcn_wave_barrier();                   // the list is reused for the masked points
    }
    const int B = fr.B;
    const bool do_inst = (P.inst_idx != nullptr) && !(FL2 & LPF_F2_LAB_NOINST);
    const bool do_box = (B > 0) && (P.M > 0) && !(FL2 & LPF_F2_LAB_NOBOX);
    if (L == 0 || !(do_inst || do_box) || (FL2 & LPF_F2_LAB_NOLIST)) return;

    // ---- masked points of the segment -> this wave's LDS list, same stable order -----------
    lpf_bits_to_list(mb, mbase, lane, lst);
    __builtin_amdgcn_wave_barrier();                       // same wave, in-order LDS queue: reads below see the writes

    // lane m keeps the next list position of mask m (the scan already added inst_off[m])
    unsigned posreg;
    {
        const int c = 2 + lane, g = min(c >> 2, LPF_TAB_GROUPS - 1);
        const unsigned x = __shfl(pre4.x, g), y = __shfl(pre4.y, g), z = __shfl(pre4.z, g), w = __shfl(pre4.w, g);
        posreg = ((c & 3) == 0) ? x : ((c & 3) == 1) ? y : ((c & 3) == 2) ? z : w;
    }
    const float4 *__restrict__ boxq = reinterpret_cast<const float4 *>(P.boxq) + (size_t)fr.box_off * 2;
    const double *__restrict__ boxp = P.boxp + (size_t)fr.box_off * 16;
    unsigned *__restrict__ cnt = P.cnt + (size_t)P.M * fr.box_off;
    // On a real scan the masked points of a segment sit mostly inside the same one or two boxes: one global
    // atomic per hit piles thousands of adds onto a handful of L2 addresses (26 us of a 65 us kernel on sample
    // frame 100).  Counts are gathered per wave in LDS (when M x B fits) and flushed once at the end.
    const bool lds_cnt = do_box && (P.M * B <= LPF_K2_LDSCNT);
    if (lds_cnt) {
        for (int i = lane; i < P.M * B; i += 64) s_cnt[wave][i] = 0u;
        __builtin_amdgcn_wave_barrier();
    }
    // K1 left each of its waves' masked points {x, y, z, label} compacted at the wave's first slot, in
    // the same order as the set bits of the masked ballots: entry e of the segment, found in
    // row r, is entry e - mbase[first row of r's K1 wave] of that wave.  No gather from the cloud,
    // the label array or (u, v).
    const float4 *__restrict__ mseg = P.mlist + fr.pt_off + seg_start;
    const int rows_per_wave = P.tile_pts >> 8;              // K1 tile = 4 waves of tile_pts/4 points

    for (unsigned e0 = 0; e0 < L; e0 += 64) {
        // ---- round trip 2: the tile lists (coalesced) and the box bounds ---------------------
        const unsigned e = e0 + lane;
        const bool act = e < L;
        float4 pq = make_float4(0.f, 0.f, 0.f, 0.f);
        const unsigned li = lst[act ? e : 0];                // segment-relative point index
        const int first_row = ((int)(li >> 6) / rows_per_wave) * rows_per_wave;
        const unsigned wb = (unsigned)__shfl((int)mbase, first_row);            // all lanes take part
        if (act) pq = mseg[first_row * 64 + (int)(e - wb)];
        const unsigned idx = (unsigned)seg_start + li;
        const unsigned lab = __float_as_uint(pq.w);
        float4 blo = make_float4(0.f, 0.f, 0.f, 0.f), bhi = blo;
        if (do_box && lane < B) { blo = boxq[2 * lane]; bhi = boxq[2 * lane + 1]; }

        // ---- K5: split by instance; ballot order == ascending point index --------------------
        if (do_inst) {
            for (int m = 0; m < P.M; ++m) {
                const bool hit = (lab >> m) & 1u;
                const unsigned long long bal = __ballot(hit);
                if (!bal) continue;
                const long long base = (long long)lpf_rl(posreg, m);
                if (hit) {
                    const long long w = base + __popcll(bal & lt);
                    if (w < P.inst_cap) P.inst_idx[fr.inst_base + w] = (long long)idx;
                }
                if (lane == m) posreg += (unsigned)__popcll(bal);
            }
        }
        // ---- K6: lane = masked point.  The frame's candidate grid (built with the boxes) lists, per
        //      32x32-pixel cell, the boxes whose accepted region can project there; a point only
        //      meets those.  Candidates pass a conservative float AABB of the region first, the
        //      survivors are queued and take the reference's f64 test a whole wave at a time. ------
        if (do_box) {
            __builtin_amdgcn_wave_barrier();
            s_pt[wave][lane] = make_float4(pq.x, pq.y, pq.z, __uint_as_float(lab));   // .w carries the label bits
            s_bq[wave][2 * lane] = blo; s_bq[wave][2 * lane + 1] = bhi;               // bounds of boxes 0..63
            __builtin_amdgcn_wave_barrier();
            unsigned *qq = s_q[wave];
            int qn = 0;                                     // wave-uniform queue length
            auto exact = [&](int count) {
                if (lane < count && !(FL2 & LPF_F2_LAB_NOEXACT)) {
                    const unsigned ent = qq[lane];
                    const int e = (int)(ent & 63u), b = (int)(ent >> 6);
                    const float4 x = s_pt[wave][e];
                    const double *bp = boxp + (size_t)b * 16;
                    const bool in = P.oriented ? lpf_oriented_inside((double)x.x, (double)x.y, (double)x.z, bp)
                                               : lpf_aabb_inside((double)x.x, (double)x.y, (double)x.z, bp);
                    if (in) {
                        unsigned l = __float_as_uint(x.w);
                        while (l) {
                            const int m = __ffs(l) - 1;
                            l &= l - 1;
                            if (lds_cnt) atomicAdd(&s_cnt[wave][m * B + b], 1u);
                            else atomicAdd(&cnt[m * B + b], 1u);
                        }
                    }
                }
            };
            int cell = 0;
            if (act && !(FL2 & LPF_F2_LAB_NOPROJ)) {        // same arithmetic as K1 => the same pixel; valid => in range
                double uf, vf, d;
                lpf_project_point(P, pq.x, pq.y, pq.z, uf, vf, d);
                cell = ((int)rint(vf) >> P.cell_shift) * P.cell_w + ((int)rint(uf) >> P.cell_shift);
            }
            const unsigned long long *__restrict__ cg = P.cand + fr.cand_off + (size_t)cell * fr.cand_words;
            for (int w = 0; w < fr.cand_words; ++w) {
                unsigned long long mset = (act && !(FL2 & LPF_F2_LAB_NOCAND)) ? cg[w] : 0ull;
                while (__any(mset != 0ull)) {
                    const bool has = mset != 0ull;
                    const int b = has ? (w << 6) + __ffsll((long long)mset) - 1 : 0;
                    mset &= mset - 1ull;
                    bool near = false;
                    if (has) {
                        float4 lo, hi;
                        if (b < 64) { lo = s_bq[wave][2 * b]; hi = s_bq[wave][2 * b + 1]; }
                        else { lo = boxq[2 * b]; hi = boxq[2 * b + 1]; }
                        near = pq.x >= lo.x && pq.x <= hi.x && pq.y >= lo.y && pq.y <= hi.y && pq.z >= lo.z && pq.z <= hi.z;
                    }
                    const unsigned long long bal = __ballot(near);
                    if (!bal) continue;
                    if (near) qq[qn + __popcll(bal & lt)] = (unsigned)lane | ((unsigned)b << 6);
                    qn += __popcll(bal);
                    if (qn >= 64) {
                        __builtin_amdgcn_wave_barrier();
                        exact(64);
                        const unsigned t = qq[64 + lane];
                        __builtin_amdgcn_wave_barrier();
                        qq[lane] = t;
                        qn -= 64;
                        __builtin_amdgcn_wave_barrier();
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
            exact(qn);
            __builtin_amdgcn_wave_barrier();
        }
    }
    if (lds_cnt) {
        __builtin_amdgcn_wave_barrier();
        for (int i = lane; i < P.M * B; i += 64) {
            const unsigned v = s_cnt[wave][i];
            if (v) atomicAdd(&cnt[i], v);
        }
    }
}

#define lpf_k2_lists lpf_k2_lists_t<0u>

// Small frames (<= 64 segments, lane = segment) need no scan kernel: a wave derives, from the frame's rows of
// seg_tab, for counter c = lane: bef = sum over the frame's segments before segment k, tot = sum over all of them.
__device__ __forceinline__ void lpf_wave_frame_counts(const LpfParams &P, const LpfFrame &fr, int k, int lane, unsigned &bef, unsigned &tot)
{
    const int ngroups = (2 + P.M + 3) >> 2;
    bef = 0u; tot = 0u;
    for (int g = 0; g < ngroups; ++g) {
        uint4 t = make_uint4(0u, 0u, 0u, 0u);
        if (lane < fr.nseg) t = P.seg_tab[(size_t)g * P.nseg_cap + fr.seg_off + lane];
        unsigned x[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {              // inclusive scan over the segments
                const unsigned u = __shfl_up(x[j], o);
                if (lane >= o) x[j] += u;
            }
            const unsigned b = (k > 0) ? lpf_rl(x[j], k - 1) : 0u, a = lpf_rl(x[j], 63);
            if (lane == 4 * g + j) { bef = b; tot = a; }
        }
    }
}

// ------------------------------------------------------------------------------------
// K2, one BLOCK per segment: same results as lpf_k2_lists_t, for launches that leave most of the chip idle
// (a single frame, a few real frames).  There a wave has its SIMD to itself and runs at one instruction every
// ~6 cycles, and real scans are dense in places -- consecutive points are neighbours in space, so a segment
// lying on cars holds over a thousand valid and hundreds of masked points: the per-segment wave of the
// throughput form becomes a 40 us serial program (measured on sample frame 100: list building 20 k cycles,
// 8 chunks x 9 k cycles of instance split + candidate walk + exact test).  Here the NW (4 or 8) waves of a block share
// one segment: rows are split NW ways for the lists (written row-parallel: a row's valid points go out as
// one contiguous run, no LDS staging), chunks of 64 masked points are dealt round-robin to the waves, and the
// order-dependent part -- where in the instance lists a chunk's points go -- comes from a per-(chunk, mask)
// count table in LDS, so nothing is serial across chunks.
// ------------------------------------------------------------------------------------
#define LPF_K2B_LDSB 32           // boxes whose exact parameters the block keeps in LDS

template <int NW>   // waves per segment: 4 or 8
__global__ __launch_bounds__(NW * 64) void lpf_k2_block(const LpfParams P)
{
    __shared__ unsigned short s_list[LPF_SEG_QUANTUM];                 // masked points, segment-relative, stable order
    __shared__ unsigned short s_cc[LPF_K2_ROWS][LPF_MAX_MASKS_DEV];    // [chunk][mask] -> entries of that mask in the chunk
    __shared__ float4 s_pt[NW][64];                          // xyz + label of a wave's current chunk
    __shared__ unsigned s_q[NW][128];                        // (point, box) pairs that passed the float bounds
    __shared__ float4 s_bq[2 * 64];                                    // {lo, hi} float bounds of boxes 0..63
    __shared__ double s_bp[LPF_K2B_LDSB * 16];                         // exact parameters of boxes 0..LPF_K2B_LDSB-1
    __shared__ unsigned s_cnt[LPF_K2_LDSCNT];                          // inside counts [M][B] when they fit
    const int tid = threadIdx.x, lane = lpf_lane(), wave = lpf_wave();
    const int sid = blockIdx.x;
    const unsigned long long lt = (1ull << lane) - 1ull;
    // ---- round trip 1 (every wave, redundantly: it is 1 KB and the scan is 60 instructions) -------------
    LpfFrame fr = P.frame0;
    if (P.F > 1) fr = P.segs[sid];
    const int ngroups = (2 + P.M + 3) >> 2;
    unsigned long long vb = P.vbal[(size_t)sid * LPF_K2_ROWS + lane];
    unsigned long long mb = P.mbal[(size_t)sid * LPF_K2_ROWS + lane];
    uint4 pre4 = make_uint4(0u, 0u, 0u, 0u);
    if (lane < ngroups && !P.inline_scan) pre4 = P.seg_pre[(size_t)lane * P.nseg_cap + sid];
    const int seg_start = (sid - fr.seg_off) * LPF_SEG_QUANTUM;
    const int seg_end = min(seg_start + LPF_SEG_QUANTUM, fr.N);
    const int nrows = (seg_end - seg_start + 63) >> 6;
    if (lane >= nrows) { vb = 0; mb = 0; }
    const unsigned cv = __popcll(vb), cm = __popcll(mb);
    unsigned iv = cv, im = cm;
#pragma unroll
    for (int o = 1; o < LPF_K2_ROWS; o <<= 1) {
        const unsigned tv = __shfl_up(iv, o), tm = __shfl_up(im, o);
        if (lane >= o) { iv += tv; im += tm; }
    }
    const unsigned vbase = iv - cv, mbase = im - cm;
    const unsigned L = lpf_rl(im, LPF_K2_ROWS - 1);
    long long run_v = (long long)lpf_rl(pre4.x, 0);
    unsigned segpos_inline = 0u;                            // lane m: list position of mask m at the start of the segment
    if (P.inline_scan) {
        unsigned bef, tot;
        lpf_wave_frame_counts(P, fr, sid - fr.seg_off, lane, bef, tot);
        run_v = (long long)lpf_rl(bef, 0);
        unsigned off = (lane >= 2 && lane < 2 + P.M) ? tot : 0u;     // inst_off[m] = totals of the masks before m
        const unsigned own = off;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned u = __shfl_up(off, o);
            if (lane >= o) off += u;
        }
        segpos_inline = (unsigned)__shfl((int)(off - own + bef), (lane + 2) & 63);
    }
    const int B = fr.B;
    const bool do_inst = P.inst_idx != nullptr;
    const bool do_box = (B > 0) && (P.M > 0);
    const bool masked_part = (L != 0) && (do_inst || do_box);   // block-uniform
    const float4 *__restrict__ boxq = reinterpret_cast<const float4 *>(P.boxq) + (size_t)fr.box_off * 2;
    const double *__restrict__ boxp = P.boxp + (size_t)fr.box_off * 16;
    unsigned *__restrict__ cnt = P.cnt + (size_t)P.M * fr.box_off;
    const bool lds_cnt = do_box && (P.M * B <= LPF_K2_LDSCNT);

    // box data of the frame -> LDS, loads issued before the list work (one round trip for the block)
    if (masked_part && do_box) {
        if (tid < 2 * min(B, 64)) s_bq[tid] = boxq[tid];
        for (int i = tid; i < min(B, LPF_K2B_LDSB) * 16; i += NW * 64) s_bp[i] = boxp[i];
        if (lds_cnt) for (int i = tid; i < P.M * B; i += NW * 64) s_cnt[i] = 0u;
    }
    // ---- lists, row-parallel: wave w owns 64/NW consecutive rows; lane = point of the row ---------------
    {
        long long *__restrict__ dst = P.valid_idx ? P.valid_idx + fr.pt_off + run_v : nullptr;
        for (int r = wave * (64 / NW); r < (wave + 1) * (64 / NW); ++r) {
            const unsigned long long rv = lpf_rl64(vb, r), rm = lpf_rl64(mb, r);     // wave-uniform
            if (dst && ((rv >> lane) & 1ull)) {                                      // contiguous run of popc(rv) entries
                const long long pos = lpf_rl(vbase, r) + __popcll(rv & lt), g = fr.pt_off + seg_start + r * 64 + lane;
                dst[pos] = (long long)(seg_start + r * 64 + lane);
                if (P.uv_valid) P.uv_valid[fr.pt_off + run_v + pos] = P.uv[g];
                if (P.label_valid) P.label_valid[fr.pt_off + run_v + pos] = P.label_bits[g];
            }
            if (masked_part && ((rm >> lane) & 1ull))
                s_list[lpf_rl(mbase, r) + __popcll(rm & lt)] = (unsigned short)(r * 64 + lane);
        }
    }
    if (!masked_part) return;                               // block-uniform: no barrier is skipped by part of a block
    __syncthreads();

    const float4 *__restrict__ mseg = P.mlist + fr.pt_off + seg_start;     // K1's hand-off lists (see lpf_k2_lists_t)
    const int rows_per_wave = P.tile_pts >> 8;
    const int rpw_shift = (rows_per_wave == 4) ? 2 : (rows_per_wave == 2) ? 1 : 0;
    const int nchunks = (int)((L + 63u) >> 6);
    auto load_chunk = [&](int c, unsigned &li) -> float4 {  // entry c*64+lane of the segment's masked points (clamped, branch-free)
        const unsigned e = (unsigned)c * 64u + lane;
        const bool act = e < L;
        li = s_list[act ? e : 0];
        const int first_row = (int)((li >> 6) >> rpw_shift) << rpw_shift;
        const unsigned wb = (unsigned)__shfl((int)mbase, first_row);
        return mseg[act ? first_row * 64 + (int)(e - wb) : 0];
    };
    // ---- pass A: per-(chunk, mask) counts; chunks dealt round-robin, the first two kept in registers --------
    float4 keep_p[2];
    unsigned keep_li[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) { keep_p[k] = make_float4(0.f, 0.f, 0.f, 0.f); keep_li[k] = 0u; }
    if (do_inst) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int c = wave + NW * k;
            if (c < nchunks) keep_p[k] = load_chunk(c, keep_li[k]);
        }
        for (int c = wave, k = 0; c < nchunks; c += NW, ++k) {
            unsigned li;
            const float4 p = (k == 0) ? keep_p[0] : (k == 1) ? keep_p[1] : load_chunk(c, li);
            const unsigned lab = ((unsigned)c * 64u + lane < L) ? __float_as_uint(p.w) : 0u;
            unsigned mine = 0u;
            for (int m = 0; m < P.M; ++m) {
                const unsigned n = (unsigned)__popcll(__ballot((lab >> m) & 1u));
                if (lane == m) mine = n;
            }
            if (lane < LPF_MAX_MASKS_DEV) s_cc[c][lane] = (unsigned short)mine;
        }
    }
    __syncthreads();

    // ---- pass B: instance lists + box counts of the wave's chunks ------------------------------------------
    unsigned *qq = s_q[wave];
    auto exact = [&](int count) {
        if (lane < count) {
            const unsigned ent = qq[lane];
            const int e = (int)(ent & 63u), b = (int)(ent >> 6);
            const float4 x = s_pt[wave][e];
            const double px = (double)x.x, py = (double)x.y, pz = (double)x.z;
            bool in;
            if (b < LPF_K2B_LDSB) {
                const double *bp = s_bp + b * 16;
                in = P.oriented ? lpf_oriented_inside(px, py, pz, bp) : lpf_aabb_inside(px, py, pz, bp);
            } else {
                const double *bp = boxp + (size_t)b * 16;
                in = P.oriented ? lpf_oriented_inside(px, py, pz, bp) : lpf_aabb_inside(px, py, pz, bp);
            }
            if (in) {
                unsigned l = __float_as_uint(x.w);
                while (l) {
                    const int m = __ffs(l) - 1;
                    l &= l - 1;
                    if (lds_cnt) atomicAdd(&s_cnt[m * B + b], 1u);
                    else atomicAdd(&cnt[m * B + b], 1u);
                }
            }
        }
    };
    // lane m: list position of mask m at the start of the segment (the scan already added inst_off[m])
    unsigned segpos;
    {
        const int c = 2 + lane, g = min(c >> 2, LPF_TAB_GROUPS - 1);
        const unsigned x = __shfl(pre4.x, g), y = __shfl(pre4.y, g), z = __shfl(pre4.z, g), w = __shfl(pre4.w, g);
        segpos = ((c & 3) == 0) ? x : ((c & 3) == 1) ? y : ((c & 3) == 2) ? z : w;
        if (P.inline_scan) segpos = segpos_inline;
    }
    unsigned before = 0u;                                   // lane m: entries of mask m in chunks [0, c) -- advanced incrementally
    int counted = 0;
    for (int c = wave, k = 0; c < nchunks; c += NW, ++k) {
        unsigned li = (k == 0) ? keep_li[0] : keep_li[1];
        const bool kept = do_inst && k < 2;
        const float4 p = kept ? ((k == 0) ? keep_p[0] : keep_p[1]) : load_chunk(c, li);
        const int nact = (int)min(64u, L - (unsigned)c * 64u);
        const bool act = lane < nact;
        const unsigned idx = (unsigned)seg_start + li;
        const unsigned lab = act ? __float_as_uint(p.w) : 0u;
        if (do_inst) {
            if (lane < LPF_MAX_MASKS_DEV)
                for (; counted < c; ++counted) before += s_cc[counted][lane];
            counted = c;
            const unsigned posreg = segpos + before;
            for (int m = 0; m < P.M; ++m) {
                const bool hit = (lab >> m) & 1u;
                const unsigned long long bal = __ballot(hit);
                if (!bal) continue;
                const long long base = (long long)lpf_rl(posreg, m);
                if (hit) {
                    const long long w = base + __popcll(bal & lt);
                    if (w < P.inst_cap) P.inst_idx[fr.inst_base + w] = (long long)idx;
                }
            }
        }
        if (!do_box) continue;
        __builtin_amdgcn_wave_barrier();
        s_pt[wave][lane] = make_float4(p.x, p.y, p.z, __uint_as_float(lab));   // .w carries the label bits
        __builtin_amdgcn_wave_barrier();
        int qn = 0;
        int cell = 0;
        if (act) {                                          // same arithmetic as K1 => the same pixel; valid => in range
            double uf, vf, d;
            lpf_project_point(P, p.x, p.y, p.z, uf, vf, d);
            cell = ((int)rint(vf) >> P.cell_shift) * P.cell_w + ((int)rint(uf) >> P.cell_shift);
        }
        const unsigned long long *__restrict__ cg = P.cand + fr.cand_off + (size_t)cell * fr.cand_words;
        for (int w = 0; w < fr.cand_words; ++w) {
            unsigned long long mset = act ? cg[w] : 0ull;
            while (__any(mset != 0ull)) {
                const bool has = mset != 0ull;
                const int b = has ? (w << 6) + __ffsll((long long)mset) - 1 : 0;
                mset &= mset - 1ull;
                bool near = false;
                if (has) {
                    float4 lo, hi;
                    if (b < 64) { lo = s_bq[2 * b]; hi = s_bq[2 * b + 1]; }
                    else { lo = boxq[2 * b]; hi = boxq[2 * b + 1]; }
                    near = p.x >= lo.x && p.x <= hi.x && p.y >= lo.y && p.y <= hi.y && p.z >= lo.z && p.z <= hi.z;
                }
                const unsigned long long bal = __ballot(near);
                if (!bal) continue;
                if (near) qq[qn + __popcll(bal & lt)] = (unsigned)lane | ((unsigned)b << 6);
                qn += __popcll(bal);
                if (qn >= 64) {
                    __builtin_amdgcn_wave_barrier();
                    exact(64);
                    const unsigned t = qq[64 + lane];
                    __builtin_amdgcn_wave_barrier();
                    qq[lane] = t;
                    qn -= 64;
                    __builtin_amdgcn_wave_barrier();
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        exact(qn);
        __builtin_amdgcn_wave_barrier();
    }
    if (lds_cnt) {
        __syncthreads();
        for (int i = tid; i < P.M * B; i += NW * 64) {
            const unsigned v = s_cnt[i];
            if (v) atomicAdd(&cnt[i], v);
        }
    }
}

// ------------------------------------------------------------------------------------
// K3: one block (4 waves) per frame.  Layout of lpf_frame_summary (include/lpf.h), in
// int64 words: [0] n_valid  [1] n_labelled  [2..33] inst_count  [34..66] inst_off
// [67..98] best_cnt, then int32: best_box[32], inst_overflow, reserved  => 928 bytes
// ------------------------------------------------------------------------------------
#define LPF_SUMMARY_BYTES 928

__global__ __launch_bounds__(LPF_BLOCK) void lpf_k3_finalize(const LpfParams P)
{
    const int f = blockIdx.x, tid = threadIdx.x, lane = lpf_lane(), wave = lpf_wave();
    const LpfFrame fr = (P.F > 1) ? P.frames[f] : P.frame0;
    const int B = fr.B, M = P.M;
    char *base = P.summary ? (char *)P.summary + (size_t)f * LPF_SUMMARY_BYTES : nullptr;
    long long *w = (long long *)base;
    int32_t *bb = base ? (int32_t *)(base + 99 * 8) : nullptr;
    __shared__ unsigned s_tot[LPF_TAB_ROWS];
    const unsigned *__restrict__ tot = P.frame_tot + (size_t)f * LPF_TAB_ROWS;
    if (P.inline_scan) {                                    // block-uniform: the frame's totals, then seg_tab is handed back clean
        if (tid < 64) {
            unsigned bef, all;
            lpf_wave_frame_counts(P, fr, 0, lane, bef, all);
            if (lane < LPF_TAB_ROWS) s_tot[lane] = (lane < 2 + M) ? all : 0u;
        }
        __syncthreads();
        tot = s_tot;
        const int ngroups = (2 + M + 3) >> 2;
        for (int i = tid; i < ngroups * fr.nseg; i += LPF_BLOCK)
            P.seg_tab[(size_t)(i / fr.nseg) * P.nseg_cap + fr.seg_off + (i % fr.nseg)] = make_uint4(0u, 0u, 0u, 0u);
    }

    // first strict maximum over the boxes, starting from 0: one wave per mask, lanes over boxes
    unsigned *__restrict__ cnt = P.cnt + (size_t)M * fr.box_off;
    for (int m = wave; m < M; m += 4) {
        unsigned best = 0;
        int best_idx = 0x7fffffff;
        for (int b = lane; b < B; b += 64) {
            const unsigned c = cnt[m * B + b];
            if (c > best) { best = c; best_idx = b; }      // ascending b per lane: keeps the first
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned ob = __shfl_down(best, o);
            const int oi = __shfl_down(best_idx, o);
            if (ob > best || (ob == best && oi < best_idx)) { best = ob; best_idx = oi; }
        }
        if (lane == 0 && base) {
            w[67 + m] = (long long)best;
            bb[m] = best ? best_idx : -1;
        }
    }
    if (base && tid < 64) {                                // wave 0: counts, offsets, flags
        const long long c = (lane < M) ? (long long)tot[2 + lane] : 0;
        long long incl = c;
#pragma unroll
        for (int o = 1; o < 32; o <<= 1) {
            const long long t = __shfl_up(incl, o);
            if (lane >= o) incl += t;
        }
        const long long total = __shfl(incl, 31);
        if (lane < 32) {
            w[2 + lane] = c;
            w[35 + lane] = incl;
            if (lane >= M) { w[67 + lane] = 0; bb[lane] = -1; }
        }
        if (lane == 0) {
            w[0] = (long long)tot[0]; w[1] = (long long)tot[1]; w[34] = 0;
            bb[32] = (total > P.inst_cap && P.inst_idx) ? 1 : 0;
            bb[33] = 0;
        }
    }
    __syncthreads();
    // hand the counters over and leave the scratch zeroed for the next call
    int32_t *out = P.count_out ? P.count_out + (size_t)M * fr.box_off : nullptr;
    for (int i = tid; i < M * B; i += LPF_BLOCK) {
        if (out) out[i] = (int32_t)cnt[i];
        cnt[i] = 0;
    }
}

// ------------------------------------------------------------------------------------
// Standalone K6: inside[b][i] for k points x B boxes -- the drop-in for
// oriented_point_in_bbox / point_in_bbox (V3:143-208), which return the per-point mask.
// Box parameters are staged in LDS 32 boxes at a time; one thread per point.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(LPF_BLOCK) void lpf_points_in_boxes_kernel(const float *__restrict__ pts, long long k, int stride,
                                                                        const double *__restrict__ boxp, int B, int oriented,
                                                                        uint8_t *__restrict__ inside)
{
    __shared__ double s_box[32][16];
    const long long i = (long long)blockIdx.x * LPF_BLOCK + threadIdx.x;
    double px = 0, py = 0, pz = 0;
    if (i < k) { px = (double)pts[i * stride]; py = (double)pts[i * stride + 1]; pz = (double)pts[i * stride + 2]; }
    for (int b0 = 0; b0 < B; b0 += 32) {
        const int nb = min(32, B - b0);
        __syncthreads();
        for (int j = threadIdx.x; j < nb * 16; j += LPF_BLOCK) s_box[j >> 4][j & 15] = boxp[(size_t)b0 * 16 + j];
        __syncthreads();
        if (i < k)
            for (int b = 0; b < nb; ++b)
                inside[(size_t)(b0 + b) * k + i] =
                    (uint8_t)(oriented ? lpf_oriented_inside(px, py, pz, s_box[b]) : lpf_aabb_inside(px, py, pz, s_box[b]));
    }
}

// ------------------------------------------------------------------------------------
// Last-writer depth image (seg_with_pointcloud.py:160-170): D[v][u] = depth of the valid
// point with the LARGEST index projecting to (u, v); the reference's per-mask maps are where(mask, D, 0).
// Pass 1 takes the per-pixel maximum of (index + 1) with integer atomics (deterministic), pass 2 lets
// the winner write its depth.  Both stream the cloud once with the K1 arithmetic.
// ------------------------------------------------------------------------------------
template <int PASS>
__global__ __launch_bounds__(LPF_BLOCK) void lpf_depth_image_kernel(const LpfParams P, int n, unsigned *__restrict__ win,
                                                                    double *__restrict__ D)
{
    const int i = blockIdx.x * LPF_BLOCK + threadIdx.x;
    if (i >= n) return;
    const float4 p = P.pts[i];
    double uf, vf, d;
    lpf_project_point(P, p.x, p.y, p.z, uf, vf, d);
    const int ui = lpf_sat_i32(rint(uf)), vi = lpf_sat_i32(rint(vf));
    if (((unsigned)ui < (unsigned)P.W) && ((unsigned)vi < (unsigned)P.H) && (d > P.dmin) && (d < P.dmax)) {
        const size_t pix = (size_t)vi * P.W + ui;
        if (PASS == 0) atomicMax(&win[pix], (unsigned)i + 1u);
        else if (win[pix] == (unsigned)i + 1u) D[pix] = d;
    }
}

// ------------------------------------------------------------------------------------
// Box preparation: for every annotated box, from its 8 corners in the cam-0 frame,
//   visible[b]      filter_visible_bboxes (V3:121-140): >= 2 corners with depth > 0.1 inside the image,
//                   corners projected WITHOUT R_rect (reference quirk, kept)
//   corners_velo[b] transform_bboxes_to_velodyne (V3:41-52): (inv(TrVeloToCam) . [c 1])[:3]
//   bbox2d[b]       V4:157-168: min/max of the integer (u, v) over the corners with depth > 0
//                   ({umin, vmin, umax, vmax} as doubles; front[b] = number of such corners)
// One thread per corner, 8 lanes per box; float64 with NumPy's dgemm order (k-ordered fma chains).
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(LPF_BLOCK) void lpf_box_prep_kernel(const double *__restrict__ corners_cam, int nbox,
                                                                 const double *__restrict__ Tcv /*[16] cam->velo*/,
                                                                 const double *__restrict__ K /*[9]*/, int W, int H,
                                                                 uint8_t *__restrict__ visible, double *__restrict__ corners_velo,
                                                                 double *__restrict__ bbox2d, int *__restrict__ front)
{
    const int t = blockIdx.x * LPF_BLOCK + threadIdx.x;
    const int b = t >> 3, k = t & 7;
    const bool live = b < nbox;
    double x = 0, y = 0, z = 0;
    if (live) { const double *c = corners_cam + ((size_t)b * 8 + k) * 3; x = c[0]; y = c[1]; z = c[2]; }
    // cam2image on the raw cam-0 corners
    double qx = K[0] * x; qx = fma(K[1], y, qx); qx = fma(K[2], z, qx);
    double qy = K[3] * x; qy = fma(K[4], y, qy); qy = fma(K[5], z, qy);
    double d  = K[6] * x; d  = fma(K[7], y, d);  d  = fma(K[8], z, d);
    if (d == 0.0) d = -1e-6;
    const double ad = fabs(d);
    const double ru = rint(qx / ad), rv = rint(qy / ad);
    const bool in_img = (ru >= 0.0) && (ru < (double)W) && (rv >= 0.0) && (rv < (double)H);
    const bool vis = live && (d > 0.1) && in_img;
    const bool fr = live && (d > 0.0);
    // 8-lane group reductions (lanes of one box are contiguous and 8-aligned inside the wave)
    const unsigned long long grp = 0xFFull << (lpf_lane() & 56);
    const int nvis = __popcll(__ballot(vis) & grp), nfront = __popcll(__ballot(fr) & grp);
    double umin = fr ? ru : 1e300, umax = fr ? ru : -1e300, vmin = fr ? rv : 1e300, vmax = fr ? rv : -1e300;
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) {
        umin = fmin(umin, __shfl_xor(umin, o)); umax = fmax(umax, __shfl_xor(umax, o));
        vmin = fmin(vmin, __shfl_xor(vmin, o)); vmax = fmax(vmax, __shfl_xor(vmax, o));
    }
    if (live) {
        double *o = corners_velo + ((size_t)b * 8 + k) * 3;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            double a = Tcv[4 * i] * x; a = fma(Tcv[4 * i + 1], y, a); a = fma(Tcv[4 * i + 2], z, a); a = fma(Tcv[4 * i + 3], 1.0, a);
            o[i] = a;
        }
        if (k == 0) {
            visible[b] = (uint8_t)(nvis >= 2);
            front[b] = nfront;
            bbox2d[4 * b] = umin; bbox2d[4 * b + 1] = vmin; bbox2d[4 * b + 2] = umax; bbox2d[4 * b + 3] = vmax;
        }
    }
}

// ------------------------------------------------------------------------------------
// K8: masks -> label image.
//   MODE 0: uint8, nonzero.  MODE 1: float, astype(uint8) != 0.  MODE 2: float, (x*255) -> u8 == 255.
//   MODE 3: float, x > 0.5 (NaN is not a member).
// ------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned lpf_f32_to_u8(float v)
{
    int t;
    if (!(v > -2147483904.0f && v < 2147483648.0f)) t = INT32_MIN; else t = (int)v;
    return (unsigned)t & 0xFFu;
}

template <typename T, int MODE>
__device__ __forceinline__ bool lpf_member(T v)
{
    if (MODE == 0) return v != 0;
    if (MODE == 1) return lpf_f32_to_u8((float)v) != 0u;
    if (MODE == 3) return (float)v > 0.5f;
    return lpf_f32_to_u8((float)v * 255.0f) == 255u;
}

// Streaming pack, 16 pixels per lane: uint8 masks are read 16 bytes per lane per mask
// (float masks 4 x 16 bytes), the packed labels leave as four 16-byte stores.
// Requires hw % 16 == 0 and 16-byte aligned mask planes (checked on the host).
template <typename T, int MODE, typename LT>
__global__ __launch_bounds__(LPF_BLOCK) void lpf_pack16(const T *__restrict__ masks, LT *__restrict__ label,
                                                        int M, long long hw, long long total16)
{
    const long long g = (long long)blockIdx.x * LPF_BLOCK + threadIdx.x;    // group of 16 pixels, over all frames
    if (g >= total16) return;
    const long long per_frame = hw >> 4;
    const long long f = g / per_frame, o = (g - f * per_frame) << 4;
    const T *__restrict__ mf = masks + (size_t)f * M * hw + o;
    uint32_t bits[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) bits[i] = 0;
    if (sizeof(T) == 1) {
        for (int m0 = 0; m0 < M; m0 += 8) {                 // eight independent 16-byte loads in flight
            uint4 q[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) q[j] = *reinterpret_cast<const uint4 *>(mf + (size_t)min(m0 + j, M - 1) * hw);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (m0 + j < M) {
                    const unsigned wv[4] = {q[j].x, q[j].y, q[j].z, q[j].w};
#pragma unroll
                    for (int i = 0; i < 16; ++i)
                        bits[i] |= (((wv[i >> 2] >> (8 * (i & 3))) & 0xFFu) != 0u ? 1u : 0u) << (m0 + j);
                }
            }
        }
    } else {
        for (int m0 = 0; m0 < M; m0 += 2) {
            float4 q[2][4];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    q[j][k] = *reinterpret_cast<const float4 *>(mf + (size_t)min(m0 + j, M - 1) * hw + 4 * k);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (m0 + j < M) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        bits[4 * k + 0] |= (lpf_member<float, MODE>(q[j][k].x) ? 1u : 0u) << (m0 + j);
                        bits[4 * k + 1] |= (lpf_member<float, MODE>(q[j][k].y) ? 1u : 0u) << (m0 + j);
                        bits[4 * k + 2] |= (lpf_member<float, MODE>(q[j][k].z) ? 1u : 0u) << (m0 + j);
                        bits[4 * k + 3] |= (lpf_member<float, MODE>(q[j][k].w) ? 1u : 0u) << (m0 + j);
                    }
                }
            }
        }
    }
    LT *dst = label + (size_t)f * hw + o;                   // 16 pixels: 16 / 32 / 64 contiguous bytes per lane
    if (sizeof(LT) == 1) {
        uint4 v;
        v.x = bits[0] | (bits[1] << 8) | (bits[2] << 16) | (bits[3] << 24);
        v.y = bits[4] | (bits[5] << 8) | (bits[6] << 16) | (bits[7] << 24);
        v.z = bits[8] | (bits[9] << 8) | (bits[10] << 16) | (bits[11] << 24);
        v.w = bits[12] | (bits[13] << 8) | (bits[14] << 16) | (bits[15] << 24);
        *reinterpret_cast<uint4 *>(dst) = v;
    } else if (sizeof(LT) == 2) {
#pragma unroll
        for (int k = 0; k < 2; ++k)
            reinterpret_cast<uint4 *>(dst)[k] = make_uint4(bits[8 * k] | (bits[8 * k + 1] << 16), bits[8 * k + 2] | (bits[8 * k + 3] << 16),
                                                            bits[8 * k + 4] | (bits[8 * k + 5] << 16), bits[8 * k + 6] | (bits[8 * k + 7] << 16));
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k)
            reinterpret_cast<uint4 *>(dst)[k] = make_uint4(bits[4 * k], bits[4 * k + 1], bits[4 * k + 2], bits[4 * k + 3]);
    }
}

// General-shape pack with optional fused first erosion: 64x16 output tile per block,
// (64+2)x(16+2) LDS tile of packed membership bits; erosion = AND of the plus-shaped
// neighbourhood, pixels outside the image read as all-ones (OpenCV erode border).
#define LPF_TW 64
#define LPF_TH 16

template <typename T, int MODE, typename LT>
__global__ __launch_bounds__(LPF_BLOCK) void lpf_pack_erode(const T *__restrict__ masks, LT *__restrict__ label,
                                                            int M, int H, int W, int erode)
{
    __shared__ uint32_t s_tile[LPF_TH + 2][LPF_TW + 2 + 1];
    const int f = blockIdx.z;
    const int x0 = blockIdx.x * LPF_TW, y0 = blockIdx.y * LPF_TH;
    const size_t hw = (size_t)H * W;
    const T *__restrict__ mf = masks + (size_t)f * M * hw;
    for (int p = threadIdx.x; p < (LPF_TH + 2) * (LPF_TW + 2); p += LPF_BLOCK) {
        const int ty = p / (LPF_TW + 2), tx = p - ty * (LPF_TW + 2);
        const int y = y0 + ty - 1, x = x0 + tx - 1;
        uint32_t bits = 0xFFFFFFFFu;
        if (y >= 0 && y < H && x >= 0 && x < W) {
            bits = 0;
            const size_t o = (size_t)y * W + x;
            for (int m = 0; m < M; ++m)
                if (lpf_member<T, MODE>(mf[m * hw + o])) bits |= 1u << m;
        }
        s_tile[ty][tx] = bits;
    }
    __syncthreads();
    const int tx = threadIdx.x & (LPF_TW - 1);
    for (int ty = threadIdx.x >> 6; ty < LPF_TH; ty += LPF_BLOCK / LPF_TW) {
        const int y = y0 + ty, x = x0 + tx;
        if (y < H && x < W) {
            uint32_t v = s_tile[ty + 1][tx + 1];
            if (erode) v &= s_tile[ty][tx + 1] & s_tile[ty + 2][tx + 1] & s_tile[ty + 1][tx] & s_tile[ty + 1][tx + 2];
            label[(size_t)f * hw + (size_t)y * W + x] = (LT)v;
        }
    }
}

// erosion iterations on the packed image (all 32 masks per AND), LDS-staged tile
template <typename LT>
__global__ __launch_bounds__(LPF_BLOCK) void lpf_erode_packed(const LT *__restrict__ in, LT *__restrict__ out,
                                                              int H, int W)
{
    __shared__ uint32_t s_tile[LPF_TH + 2][LPF_TW + 2 + 1];
    const int f = blockIdx.z;
    const int x0 = blockIdx.x * LPF_TW, y0 = blockIdx.y * LPF_TH;
    const size_t hw = (size_t)H * W;
    for (int p = threadIdx.x; p < (LPF_TH + 2) * (LPF_TW + 2); p += LPF_BLOCK) {
        const int ty = p / (LPF_TW + 2), tx = p - ty * (LPF_TW + 2);
        const int y = y0 + ty - 1, x = x0 + tx - 1;
        s_tile[ty][tx] = (y >= 0 && y < H && x >= 0 && x < W) ? (uint32_t)in[(size_t)f * hw + (size_t)y * W + x] : 0xFFFFFFFFu;
    }
    __syncthreads();
    const int tx = threadIdx.x & (LPF_TW - 1);
    for (int ty = threadIdx.x >> 6; ty < LPF_TH; ty += LPF_BLOCK / LPF_TW) {
        const int y = y0 + ty, x = x0 + tx;
        if (y < H && x < W)
            out[(size_t)f * hw + (size_t)y * W + x] = (LT)(s_tile[ty + 1][tx + 1] & s_tile[ty][tx + 1] & s_tile[ty + 2][tx + 1] &
                                                            s_tile[ty + 1][tx] & s_tile[ty + 1][tx + 2]);
    }
}
