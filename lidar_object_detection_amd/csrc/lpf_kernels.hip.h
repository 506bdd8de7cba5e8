// lpf_kernels.hip.h -- gfx950 (MI355X, wave64) kernels of the LiDAR projection +
// instance point-filter path.  Included by lpf_api.hip only.
//
// Kernel map (reference statements: /root/reference/Coding_testes, see include/lpf.h)
//   lpf_pack_erode_*   masks -> uint32 label image, 3x3-cross erosion in an LDS tile   (V3:82-97, V3:222)
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
#define LPF_CHUNK_ROWS 4         // K1: float4 loads in flight per lane
#define LPF_CHUNK (LPF_BLOCK * LPF_CHUNK_ROWS)   // 1024 points
#define LPF_K2_BATCH 4096        // K2: points per LDS batch (64 ballot rows)
#define LPF_K2_ROWS (LPF_K2_BATCH / 64)

struct LpfFrame {                // one per frame, device + host copy
    long long pt_off;            // first point of the frame in the concatenated arrays
    long long inst_base;         // first entry of the frame in inst_idx
    int N;                       // points in the frame
    int seg_off;                 // first segment of the frame
    int nseg;                    // segments of the frame
    int box_off;                 // first box of the frame
    int B;                       // boxes of the frame
    int pad;
};

struct LpfParams {
    double T[12];                // rows 0..2 of TrVeloToRect
    double K[9];                 // camera.K[:3,:3]
    double dmin, dmax;
    int W, H;
    int F, M;
    int seg_pts;                 // points per segment (multiple of LPF_CHUNK)
    int nseg_total;
    int nseg_cap;                // row pitch of seg_inst
    int oriented;
    long long inst_cap;
    const LpfFrame *frames;
    const float4 *pts;
    const uint32_t *label_img;   // [F][H][W] or null
    const double *boxp;          // [Btot][16]
    // outputs (nullable)
    int2 *uv;
    uint32_t *label_bits;
    double *depth, *uf, *vf;
    long long *valid_idx;
    long long *inst_idx;
    int32_t *count_out;
    void *summary;               // lpf_frame_summary[F]
    // scratch
    unsigned long long *vbal, *mbal;   // one 64-bit ballot per 64 points
    uint2 *seg_cnt;              // per segment {n_valid, n_masked}
    unsigned *seg_inst;          // [32][nseg_cap] per-segment per-instance counts
    unsigned *inst_total;        // [F][32]
    unsigned *cnt;               // [M*Btot] inside counts (self-cleaned by K3)
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

__device__ __forceinline__ int32_t lpf_sat_i32(double r)
{
    if (r != r) return INT32_MIN;
    if (r >= 2147483647.0) return INT32_MAX;
    if (r <= -2147483648.0) return INT32_MIN;
    return (int32_t)r;
}

// ------------------------------------------------------------------------------------
// K1: one block = one segment of one frame, streamed in chunks of 1024 points.
// Algorithmic HBM bytes per point: 16 (xyzI) + 8 (u,v) + 4 (label) = 28, + 0.25 (ballots).
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(LPF_BLOCK) void lpf_k1_project(const LpfParams P)
{
    __shared__ unsigned s_nvalid, s_nmask, s_inst[32];
    const int tid = threadIdx.x, lane = lpf_lane(), wave = lpf_wave();
    const int sid = lpf_xcd_remap(blockIdx.x, P.nseg_total);
    const int f = lpf_find_frame(P.frames, P.F, sid);
    const LpfFrame fr = P.frames[f];
    const int seg_start = (sid - fr.seg_off) * P.seg_pts;
    const int seg_end = min(seg_start + P.seg_pts, fr.N);
    const float4 *__restrict__ pts = P.pts + fr.pt_off;
    const uint32_t *__restrict__ limg =
        (P.label_img && P.M > 0) ? P.label_img + (size_t)f * (size_t)P.W * (size_t)P.H : nullptr;
    const int rows_per_seg = P.seg_pts >> 6;
    const double Wd = (double)P.W, Hd = (double)P.H;

    if (tid < 32) s_inst[tid] = 0;
    if (tid == 0) { s_nvalid = 0; s_nmask = 0; }
    __syncthreads();
    unsigned nvalid_w = 0, nmask_w = 0;

    for (int c = seg_start; c < seg_end; c += LPF_CHUNK) {
        float4 p[LPF_CHUNK_ROWS];
#pragma unroll
        for (int r = 0; r < LPF_CHUNK_ROWS; ++r) {
            // clamp instead of branching: all four loads issue back to back, one wait
            const int idx = c + r * LPF_BLOCK + tid;
            p[r] = pts[min(idx, seg_end - 1)];
        }
#pragma unroll
        for (int r = 0; r < LPF_CHUNK_ROWS; ++r) {
            const int idx = c + r * LPF_BLOCK + tid;
            const bool live = idx < seg_end;
            const double x = (double)p[r].x, y = (double)p[r].y, z = (double)p[r].z;
            // K1: rows of T as k-ordered fma chains (= OpenBLAS dgemm on this shape)
            double cx = P.T[0] * x; cx = fma(P.T[1], y, cx); cx = fma(P.T[2],  z, cx); cx = cx + P.T[3];
            double cy = P.T[4] * x; cy = fma(P.T[5], y, cy); cy = fma(P.T[6],  z, cy); cy = cy + P.T[7];
            double cz = P.T[8] * x; cz = fma(P.T[9], y, cz); cz = fma(P.T[10], z, cz); cz = cz + P.T[11];
            // K2: cam2image
            double qx = P.K[0] * cx; qx = fma(P.K[1], cy, qx); qx = fma(P.K[2], cz, qx);
            double qy = P.K[3] * cx; qy = fma(P.K[4], cy, qy); qy = fma(P.K[5], cz, qy);
            double d  = P.K[6] * cx; d  = fma(P.K[7], cy, d);  d  = fma(P.K[8], cz, d);
            if (d == 0.0) d = -1e-6;
            const double ad = fabs(d);
            const double uf = qx / ad, vf = qy / ad;
            const double ru = rint(uf), rv = rint(vf);          // np.round: half to even
            // K3: clip
            const bool valid = live && (ru >= 0.0) && (ru < Wd) && (rv >= 0.0) && (rv < Hd) &&
                               (d > P.dmin) && (d < P.dmax);
            // K4: label gather (2.1 MB image, L2 resident)
            uint32_t lab = 0;
            if (valid && limg) lab = limg[(int)rv * P.W + (int)ru];
            if (live) {
                const long long g = fr.pt_off + idx;
                if (P.uv) P.uv[g] = make_int2(lpf_sat_i32(ru), lpf_sat_i32(rv));
                if (P.label_bits) P.label_bits[g] = lab;
                if (P.depth) P.depth[g] = d;
                if (P.uf) P.uf[g] = uf;
                if (P.vf) P.vf[g] = vf;
            }
            const unsigned long long vb = __ballot(valid);
            const unsigned long long mb = __ballot(lab != 0);
            if (lane == 0) {
                const size_t row = (size_t)sid * rows_per_seg + ((c - seg_start) >> 6) + r * 4 + wave;
                P.vbal[row] = vb;
                P.mbal[row] = mb;
            }
            nvalid_w += __popcll(vb);
            nmask_w += __popcll(mb);
            while (lab) {                                   // rare: per-instance counts
                const int m = __ffs(lab) - 1;
                lab &= lab - 1;
                atomicAdd(&s_inst[m], 1u);
            }
        }
    }
    if (lane == 0) { atomicAdd(&s_nvalid, nvalid_w); atomicAdd(&s_nmask, nmask_w); }
    __syncthreads();
    if (tid == 0) P.seg_cnt[sid] = make_uint2(s_nvalid, s_nmask);
    if (tid < 32) {
        const unsigned cnt = s_inst[tid];
        P.seg_inst[(size_t)tid * P.nseg_cap + sid] = cnt;
        if (cnt) atomicAdd(&P.inst_total[f * 32 + tid], cnt);
    }
}

// ------------------------------------------------------------------------------------
// K6 helpers: membership of one point in one box, from precomputed box parameters
//   oriented: boxp = { c0[3], (v[3], vv) x 3 }      (V3:187-202)
//   aabb    : boxp = { lo[3], hi[3] }                (V3:158-162)
// ------------------------------------------------------------------------------------
__device__ __forceinline__ bool lpf_oriented_inside(double px, double py, double pz, const double *__restrict__ b)
{
    const double rx = px - b[0], ry = py - b[1], rz = pz - b[2];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const double v0 = b[3 + 4 * a], v1 = b[4 + 4 * a], v2 = b[5 + 4 * a], vv = b[6 + 4 * a];
        double d = v1 * ry; d = fma(v0, rx, d); d = fma(v2, rz, d);   // dgemv_t tail order
        const double t = d / vv;
        if (!(t >= 0.0 && t <= 1.0)) return false;
    }
    return true;
}

__device__ __forceinline__ bool lpf_aabb_inside(double px, double py, double pz, const double *__restrict__ b)
{
    return (px >= b[0]) && (px <= b[3]) && (py >= b[1]) && (py <= b[4]) && (pz >= b[2]) && (pz <= b[5]);
}

// ------------------------------------------------------------------------------------
// K2: same segmentation as K1.  Reads 2 bits per point of ballots, writes the lists.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(LPF_BLOCK) void lpf_k2_lists(const LpfParams P)
{
    __shared__ unsigned long long s_vbal[LPF_K2_ROWS], s_mbal[LPF_K2_ROWS];
    __shared__ unsigned s_vbase[LPF_K2_ROWS], s_mbase[LPF_K2_ROWS];
    __shared__ unsigned s_vtot, s_mtot;
    __shared__ unsigned s_red[2][4];
    __shared__ long long s_instpos[32];     // next write position of instance m (frame-relative)
    __shared__ unsigned s_lidx[LPF_K2_BATCH], s_llab[LPF_K2_BATCH];

    const int tid = threadIdx.x, lane = lpf_lane(), wave = lpf_wave();
    const int sid = blockIdx.x;
    const int f = lpf_find_frame(P.frames, P.F, sid);
    const LpfFrame fr = P.frames[f];
    const int seg_start = (sid - fr.seg_off) * P.seg_pts;
    const int seg_end = min(seg_start + P.seg_pts, fr.N);
    const int rows_per_seg = P.seg_pts >> 6;
    const unsigned long long lt = (1ull << lane) - 1ull;
    const uint2 mine = P.seg_cnt[sid];
    if (mine.x == 0) return;                               // no valid point => no masked point either

    // (1) exclusive prefix of {n_valid, n_masked} over the frame's earlier segments
    unsigned pv = 0;
    for (int s = fr.seg_off + tid; s < sid; s += LPF_BLOCK) pv += P.seg_cnt[s].x;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) pv += __shfl_down(pv, o);
    if (lane == 0) s_red[0][wave] = pv;

    // (2) per-instance: list offset inside the frame + count in earlier segments
    const bool do_inst = (mine.y > 0) && (P.inst_idx != nullptr);
    if (do_inst) {
        for (int m = wave; m < P.M; m += 4) {
            unsigned acc = 0;
            const unsigned *__restrict__ col = P.seg_inst + (size_t)m * P.nseg_cap;
            for (int s = fr.seg_off + lane; s < sid; s += 64) acc += col[s];
            unsigned before = (lane < m) ? P.inst_total[f * 32 + lane] : 0u;   // inst_off[m]
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { acc += __shfl_down(acc, o); before += __shfl_down(before, o); }
            if (lane == 0) s_instpos[m] = (long long)before + (long long)acc;
        }
    }
    __syncthreads();
    long long run_v = (long long)s_red[0][0] + s_red[0][1] + s_red[0][2] + s_red[0][3];

    const float4 *__restrict__ pts = P.pts + fr.pt_off;
    const int B = fr.B;
    const bool do_box = (mine.y > 0) && (B > 0) && (P.M > 0);

    for (int b0 = seg_start; b0 < seg_end; b0 += LPF_K2_BATCH) {
        const int nrows = min(LPF_K2_ROWS, (seg_end - b0 + 63) >> 6);
        if (tid < LPF_K2_ROWS) {                           // == wave 0
            const size_t row = (size_t)sid * rows_per_seg + ((b0 - seg_start) >> 6) + tid;
            const unsigned long long vb = (tid < nrows) ? P.vbal[row] : 0ull;
            const unsigned long long mb = (tid < nrows) ? P.mbal[row] : 0ull;
            s_vbal[tid] = vb; s_mbal[tid] = mb;
            unsigned cv = __popcll(vb), cm = __popcll(mb), iv = cv, im = cm;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {             // inclusive wave scan over the 64 rows
                const unsigned tv = __shfl_up(iv, o), tm = __shfl_up(im, o);
                if (lane >= o) { iv += tv; im += tm; }
            }
            s_vbase[tid] = iv - cv; s_mbase[tid] = im - cm;
            if (tid == 63) { s_vtot = iv; s_mtot = im; }
        }
        __syncthreads();
        const unsigned L = s_mtot;
        // valid_idx: ascending by construction (rows in order, lanes in order)
        if (P.valid_idx) {
            for (int row = wave; row < nrows; row += 4) {
                const unsigned long long bal = s_vbal[row];
                if ((bal >> lane) & 1ull)
                    P.valid_idx[fr.pt_off + run_v + s_vbase[row] + __popcll(bal & lt)] =
                        (long long)(b0 + row * 64 + lane);
            }
        }
        if (L > 0 && (do_inst || do_box)) {
            // masked points of this batch -> LDS list, same stable order
            for (int row = wave; row < nrows; row += 4) {
                const unsigned long long bal = s_mbal[row];
                if ((bal >> lane) & 1ull) {
                    const unsigned pos = s_mbase[row] + __popcll(bal & lt);
                    const unsigned idx = (unsigned)(b0 + row * 64 + lane);
                    s_lidx[pos] = idx;
                    s_llab[pos] = P.label_bits[fr.pt_off + idx];
                }
            }
            __syncthreads();
            if (do_inst) {                                 // K5: split by instance, one wave per mask
                for (int m = wave; m < P.M; m += 4) {
                    long long pos = s_instpos[m];
                    for (unsigned e0 = 0; e0 < L; e0 += 64) {
                        const unsigned e = e0 + lane;
                        const bool hit = (e < L) && ((s_llab[e] >> m) & 1u);
                        const unsigned long long bal = __ballot(hit);
                        if (hit) {
                            const long long w = pos + __popcll(bal & lt);
                            if (w < P.inst_cap) P.inst_idx[fr.inst_base + w] = (long long)s_lidx[e];
                        }
                        pos += __popcll(bal);
                    }
                    if (lane == 0) s_instpos[m] = pos;
                }
            }
            if (do_box) {                                  // K6: dense over masked points, boxes uniform
                const double *__restrict__ boxp = P.boxp + (size_t)fr.box_off * 16;
                unsigned *__restrict__ cnt = P.cnt + (size_t)P.M * fr.box_off;
                for (unsigned e = tid; e < L; e += LPF_BLOCK) {
                    const float4 q = pts[s_lidx[e]];
                    const unsigned lab = s_llab[e];
                    const double px = (double)q.x, py = (double)q.y, pz = (double)q.z;
                    for (int b = 0; b < B; ++b) {
                        const double *bp = boxp + (size_t)b * 16;
                        const bool in = P.oriented ? lpf_oriented_inside(px, py, pz, bp)
                                                   : lpf_aabb_inside(px, py, pz, bp);
                        if (in) {
                            unsigned l = lab;
                            while (l) {
                                const int m = __ffs(l) - 1;
                                l &= l - 1;
                                atomicAdd(&cnt[m * B + b], 1u);
                            }
                        }
                    }
                }
            }
        }
        run_v += s_vtot;
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------
// K3: one block per frame.  Layout of lpf_frame_summary (include/lpf.h), in int64 words:
//   [0] n_valid  [1] n_labelled  [2..33] inst_count  [34..66] inst_off  [67..98] best_cnt
//   then int32: best_box[32], inst_overflow, reserved   => 99*8 + 34*4 = 928 bytes
// ------------------------------------------------------------------------------------
#define LPF_SUMMARY_BYTES 928

__global__ __launch_bounds__(64) void lpf_k3_finalize(const LpfParams P)
{
    const int f = blockIdx.x, lane = threadIdx.x;
    const LpfFrame fr = P.frames[f];
    unsigned long long nv = 0, nm = 0;
    for (int s = fr.seg_off + lane; s < fr.seg_off + fr.nseg; s += 64) {
        const uint2 c = P.seg_cnt[s];
        nv += c.x; nm += c.y;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { nv += __shfl_down(nv, o); nm += __shfl_down(nm, o); }

    const int B = fr.B, M = P.M;
    unsigned *__restrict__ cnt = P.cnt + (size_t)M * fr.box_off;
    long long icount = 0, best = 0;
    int best_idx = -1;
    if (lane < 32) {
        icount = (lane < M) ? (long long)P.inst_total[f * 32 + lane] : 0;
        if (lane < M) {
            for (int b = 0; b < B; ++b) {                  // first strict maximum, starting from 0
                const long long c = (long long)cnt[lane * B + b];
                if (c > best) { best = c; best_idx = b; }
            }
        }
    }
    long long incl = icount;                               // inclusive scan over the 32 masks
#pragma unroll
    for (int o = 1; o < 32; o <<= 1) {
        const long long t = __shfl_up(incl, o);
        if (lane >= o) incl += t;
    }
    const long long total = __shfl(incl, 31);
    if (P.summary) {
        char *base = (char *)P.summary + (size_t)f * LPF_SUMMARY_BYTES;
        long long *w = (long long *)base;
        int32_t *bb = (int32_t *)(base + 99 * 8);
        if (lane == 0) {
            w[0] = (long long)nv; w[1] = (long long)nm; w[34] = 0;
            bb[32] = (total > P.inst_cap && P.inst_idx) ? 1 : 0;
            bb[33] = 0;
        }
        if (lane < 32) {
            w[2 + lane] = icount;
            w[35 + lane] = incl;
            w[67 + lane] = best;
            bb[lane] = best_idx;
        }
    }
    __syncthreads();
    // hand the counters over and leave the scratch zeroed for the next call
    int32_t *out = P.count_out ? P.count_out + (size_t)M * fr.box_off : nullptr;
    for (int i = lane; i < M * B; i += 64) {
        if (out) out[i] = (int32_t)cnt[i];
        cnt[i] = 0;
    }
    if (lane < 32) P.inst_total[f * 32 + lane] = 0;
}

// ------------------------------------------------------------------------------------
// K8: masks -> label image.  64x16 output tile per block, (64+2)x(16+2) LDS tile of
// packed membership bits; erosion = AND of the plus-shaped neighbourhood, pixels outside
// the image read as all-ones (OpenCV erode border).
//   MODE 0: uint8, nonzero.  MODE 1: float, astype(uint8) != 0.  MODE 2: float, (x*255) -> u8 == 255.
// ------------------------------------------------------------------------------------
#define LPF_TW 64
#define LPF_TH 16

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
    return lpf_f32_to_u8((float)v * 255.0f) == 255u;
}

template <typename T, int MODE>
__global__ __launch_bounds__(LPF_BLOCK) void lpf_pack_erode(const T *__restrict__ masks, uint32_t *__restrict__ label,
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
            label[(size_t)f * hw + (size_t)y * W + x] = v;
        }
    }
}

// further erosion iterations on the packed image (all 32 masks per AND)
__global__ __launch_bounds__(LPF_BLOCK) void lpf_erode_packed(const uint32_t *__restrict__ in, uint32_t *__restrict__ out,
                                                              int H, int W)
{
    __shared__ uint32_t s_tile[LPF_TH + 2][LPF_TW + 2 + 1];
    const int f = blockIdx.z;
    const int x0 = blockIdx.x * LPF_TW, y0 = blockIdx.y * LPF_TH;
    const size_t hw = (size_t)H * W;
    for (int p = threadIdx.x; p < (LPF_TH + 2) * (LPF_TW + 2); p += LPF_BLOCK) {
        const int ty = p / (LPF_TW + 2), tx = p - ty * (LPF_TW + 2);
        const int y = y0 + ty - 1, x = x0 + tx - 1;
        s_tile[ty][tx] = (y >= 0 && y < H && x >= 0 && x < W) ? in[(size_t)f * hw + (size_t)y * W + x] : 0xFFFFFFFFu;
    }
    __syncthreads();
    const int tx = threadIdx.x & (LPF_TW - 1);
    for (int ty = threadIdx.x >> 6; ty < LPF_TH; ty += LPF_BLOCK / LPF_TW) {
        const int y = y0 + ty, x = x0 + tx;
        if (y < H && x < W)
            out[(size_t)f * hw + (size_t)y * W + x] = s_tile[ty + 1][tx + 1] & s_tile[ty][tx + 1] & s_tile[ty + 2][tx + 1] &
                                                       s_tile[ty + 1][tx] & s_tile[ty + 1][tx + 2];
    }
}
