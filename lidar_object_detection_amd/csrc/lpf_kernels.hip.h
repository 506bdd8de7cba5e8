// lpf_kernels.hip.h -- gfx950 (MI355X, wave64) kernels of the LiDAR projection +
// instance point-filter path.  Included by lpf_api.hip only.
//
// Kernel map (reference statements: /root/reference/Coding_testes, see include/lpf.h)
//   lpf_pack16 / lpf_pack_erode / lpf_erode_packed
//                      masks -> label image (bit m = mask m); 3x3-cross erosion
//                      on an LDS-staged tile of packed bits                             (V3:82-97, V3:222)
//   lpf_k1_project     float4 stream: 4x4 transform, cam2image, clip, label gather,
//                      per-row wave ballots + three levels of counters                  (V3:565-569, 584, 225)
//   lpf_tail / lpf_tail_wide   one launch for the two roles after K1:
//     lists            ballots -> stable valid / per-instance index lists (wave prefix)  (V3:585, 228)
//     box count        masked points x candidate boxes of one 64-box word, slab test -> integer counters   (V3:187-202, 370)
//   lpf_finalize       first-strict-max box scan + per-frame summary; hands the counters back zeroed       (V3:353-379)
//   box job            per-frame box preparation (filter_visible_bboxes, transform_bboxes_to_velodyne) and the tables the
//                      box count reads: slab parameters, float bounds, ground grids       (V3:556-562, 121-140, 41-52)
//   lpf_step_t         software-pipelined modes: all of the above as roles of ONE launch per run
//
// Arithmetic: everything the reference computes in float64 is float64 here, with the
// summation order NumPy/OpenBLAS uses (see oracle/lpf_oracle.c); the file is compiled
// with -ffp-contract=off so only the fma() calls written below fuse.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define LPF_BLOCK 256            // 4 waves of 64
#define LPF_SEG_QUANTUM 4096     // points per segment of a big launch = 64 ballot rows = one list wave (small launches: LPF_SEG_SMALL)
#define LPF_SEG_SMALL 1024       // ... of a small one: 16 rows, four times the waves for the same points; every K1 tile size divides both
#define LPF_GROUP_SEGS 64        // segments per group (second level of the counters: one lane per segment / per group)
#define LPF_FRM_SHARDS 8         // the frame-level counters are kept in 8 copies (segment index mod 8) so no address is hot
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
    long long cand_off;          // first word of the frame's candidate-box grids
    int cand_words;              // grids of the frame: one per 64 boxes = ceil(B / 64)
    int grp_off;                 // first group (of LPF_GROUP_SEGS segments) of the frame
    int shift;                   // pt_off & 63: the frame's ballot rows start at the 64-point boundary at or below its first point, so that
                                 // every wave's loads and stores are whole 64-byte lines whatever the frame's place in the batch (a batch of
                                 // real scans: frames of ~116 k points, none a multiple of 64 -- every 512-byte store of a wave straddled
                                 // a ninth line, every label store a fifth: +9 % HBM requests, PMC).  Row r, lane l is frame point
                                 // 64 r + l - shift; the first `shift` lanes of row 0 are dead.
    int pad4;
};

struct LpfParams {
    double T[12];                // rows 0..2 of TrVeloToRect
    double K[9];                 // camera.K[:3,:3]
    double dmin, dmax;
    int W, H;
    int F, M;
    int seg_pts;                 // points per segment: LPF_SEG_QUANTUM or LPF_SEG_SMALL
    int nseg_total;
    int nseg_cap;                // pitch (segments) of one seg_tab group
    int ngrp_cap;                // pitch (groups) of one grp_tab group
    int oriented;
    long long inst_cap;
    LpfFrame frame0;             // the frame table by value when F == 1 (no dependent load)
    const LpfFrame *frames;      // [F]
    const LpfFrame *segs;        // [nseg_total] the owning frame's record per segment (pad = frame id)
    const int2 *blks;            // [nblk] list blocks of the tail: {first segment, frame << 3 | segments (0..4)}
    const int4 *cblks;           // [ncblk] box-count blocks: {first segment, frame, candidate word, segments (0..4)}
    int nblk, ncblk;
    int lists_small;             // narrow tail blocks build their lists with the 16-row wave (small launches of dense real scans)
    int csplit;                  // count blocks per (group of segments, word): they share the group's chunks of 64 masked points (1, or 4 in small
                                 // software-pipelined launches, whose longest chain is a count block on a car)
    const float4 *pts;
    const void *label_img;       // [F][H][W] label image (uint8 / uint16 / uint32 elements, see LT) or null
    const int4 *rects;           // [F][M] {x0, y0, x1, y1} (half open) of the masks (LpfDirectRect tiles only): a mask is read as zero outside
    const uint32_t *rect_grid;   // [F][rg_cells] per cell of LPF_RG_CELL x LPF_RG_CELL pixels: the masks whose rectangle meets the cell (lpf_rect_grid_block)
    int rg_cw, rg_cells;         // cells per row of the grid, cells per frame
    const double *boxp;          // [Btot][16] exact box parameters
    const float *boxq;           // [Btot][8]  conservative float AABB {lo xyz, hi xyz}
    const unsigned long long *cand;   // per (frame, 64-box word) a ground grid (LPF_GRID_WORDS words): see lpf_box_frame_block
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
                                 // K1 tiles add into it, the tail's finalizing block leaves it zeroed
    uint4 *grp_tab;              // [LPF_TAB_GROUPS][ngrp_cap] the same per group of LPF_GROUP_SEGS segments
    uint4 *frm_tab;              // [F][LPF_FRM_SHARDS][LPF_TAB_GROUPS] ... and per frame (sum the shards)
    uint4 *seg_pre;              // [LPF_TAB_GROUPS][nseg_cap] written by lpf_scan_segments (frames of more than 64 groups only)
    unsigned *cnt;               // [M*Btot] inside counts (self-cleaned by the finalizing block)
    float4 *mlist;               // [Ntot] per K1 wave (64*ROWS points), at the wave's first slot: {x, y, z, label
                                 // bits} of its masked points in point order (nothing gathers from the cloud later)
    int count_boxes;             // 1: the tail launch carries the box-count blocks
    int count_lazy;              // 1: a count block first looks whether its segments hold a masked point at all and leaves if not -- before it
                                 // stages its word's boxes (10 KB).  Large launches of real scans: most groups of segments hold none (the
                                 // cars are a few per cent of a scan), and a frame with 300 boxes has five blocks per group.  Small
                                 // launches keep both loads in one round trip: there the longest block is what counts.
    int tile_pts;                // points per K1 tile of this launch (4 waves)
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

// The frame record of a block: by value from the kernel arguments when the launch has one frame (no dependent load), else
// from the table, and then made wave-uniform by hand.  Left to itself the compiler merges the two into one choice between
// the two pointers, and the fields arrive through flat vector loads, one at each first use (and live in vector registers:
// lpf_tail_t 70 -> 58 VGPRs with this).
__device__ __forceinline__ int lpf_uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ long long lpf_uni64(long long v)
{
    return (long long)(((unsigned long long)(unsigned)lpf_uni((int)((unsigned long long)v >> 32)) << 32) | (unsigned)lpf_uni((int)v));
}
__device__ __forceinline__ LpfFrame lpf_frame_record(const LpfFrame &by_value, const LpfFrame *table, const bool many, const int idx)
{
    LpfFrame fr = by_value;
    if (many) {
        const LpfFrame t = table[lpf_uni(idx)];
        fr.pt_off = lpf_uni64(t.pt_off); fr.inst_base = lpf_uni64(t.inst_base); fr.N = lpf_uni(t.N); fr.seg_off = lpf_uni(t.seg_off);
        fr.nseg = lpf_uni(t.nseg); fr.box_off = lpf_uni(t.box_off); fr.B = lpf_uni(t.B); fr.pad = lpf_uni(t.pad);
        fr.cand_off = lpf_uni64(t.cand_off); fr.cand_words = lpf_uni(t.cand_words); fr.grp_off = lpf_uni(t.grp_off); fr.shift = lpf_uni(t.shift);
    }
    return fr;
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

// The same with the 21 constants read from memory (LDS: every lane reads the same word, a broadcast) instead of from
// the kernel arguments: the tail kernel cannot afford the 42 SGPRs next to everything else it keeps.
__device__ __forceinline__ void lpf_project_point_mem(const double *__restrict__ TK, float fx, float fy, float fz,
                                                      double &uf, double &vf, double &d)
{
    const double x = (double)fx, y = (double)fy, z = (double)fz;
    double cx = TK[0] * x; cx = fma(TK[1], y, cx); cx = fma(TK[2],  z, cx); cx = cx + TK[3];
    double cy = TK[4] * x; cy = fma(TK[5], y, cy); cy = fma(TK[6],  z, cy); cy = cy + TK[7];
    double cz = TK[8] * x; cz = fma(TK[9], y, cz); cz = fma(TK[10], z, cz); cz = cz + TK[11];
    double qx = TK[12] * cx; qx = fma(TK[13], cy, qx); qx = fma(TK[14], cz, qx);
    double qy = TK[15] * cx; qy = fma(TK[16], cy, qy); qy = fma(TK[17], cz, qy);
    d = TK[18] * cx; d = fma(TK[19], cy, d); d = fma(TK[20], cz, d);
    if (d == 0.0) d = -1e-6;
    lpf_div2(qx, qy, fabs(d), uf, vf);
}

// ------------------------------------------------------------------------------------
// Mask membership rules (V3:222-235 / cvs_erosion.py: what the reference's binarisation makes of a mask value) and the two
// sources K1 takes a point's label bits from: the packed label image (one gather per point), or -- small launches whose
// masks need no erosion -- the caller's masks themselves, M gathers per VALID point issued together.  A real frame has
// ~20 k valid points against 530 k pixels x M masks: packing first costs a 4.7 us launch that reads 2.6 MB to serve them.
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

template <typename T, int MODE> struct LpfDirect { };        // tag: label bits straight from masks of element T under rule MODE
template <typename LT> struct LpfIsDirect { static constexpr bool value = false; };
template <typename T, int MODE> struct LpfIsDirect<LpfDirect<T, MODE> > { static constexpr bool value = true; };

// ... and with the masks' rectangles (lpf_set_mask_rects: a detector hands out every mask with its 2D box): a valid point is tested
// against the frame's <= 32 rectangles (wave-uniform scalar loads) and reads mask m only inside rectangle m -- launches of ANY size then
// run no pack and write no label image.  A real frame's five masks are 2.6 MB of which a few per cent lie inside the boxes, and a
// fifth of its points are valid but only a few per cent of them fall into a rectangle: 146 real frames per step moved 743 MB per launch
// (PMC) of which 88 were the pack's and ~40 the tiles' label look-ups, against 474 of strict point traffic.
template <typename T, int MODE> struct LpfDirectRect { };
template <typename LT> struct LpfIsDirectRect { static constexpr bool value = false; };
template <typename T, int MODE> struct LpfIsDirectRect<LpfDirectRect<T, MODE> > { static constexpr bool value = true; };
template <typename T, int MODE> struct LpfIsDirect<LpfDirectRect<T, MODE> > { static constexpr bool value = true; };     // (no pack beside such tiles)

template <typename LT>
struct LpfLabelSrc {                                         // packed label image [F][H][W] of LT
    typedef LT elem;
    static __device__ __forceinline__ uint32_t get(const elem *__restrict__ img, const LpfParams &, const int pix, const int4 *, int, int) { return (uint32_t)img[pix]; }
    static __device__ __forceinline__ size_t frame_stride(const LpfParams &P) { return (size_t)P.W * (size_t)P.H; }
};
template <typename T, int MODE>
struct LpfLabelSrc<LpfDirect<T, MODE> > {                    // masks [F][M][H][W] of T
    typedef T elem;
    // rc: the frame's rectangles (lpf_set_mask_rects) or null -- a mask counts inside its rectangle only, pixel for pixel, as in every
    // other form that takes the hint (here all M bytes are read anyway: a small launch has no candidate grid, which would be one more
    // kernel in front of the tiles -- in order 18.8 vs 23.0 us for a single real frame)
    static __device__ __forceinline__ uint32_t get(const elem *__restrict__ msk, const LpfParams &P, const int pix, const int4 *__restrict__ rc,
                                                   const int ui, const int vi)
    {
        const size_t hw = (size_t)P.W * (size_t)P.H;
        uint32_t l = 0;
        for (int m0 = 0; m0 < P.M; m0 += 8) {                // eight loads in flight, then their tests (M <= 8: one round trip)
            T v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = msk[(size_t)min(m0 + j, P.M - 1) * hw + pix];
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (m0 + j < P.M && lpf_member<T, MODE>(v[j])) l |= 1u << (m0 + j);
        }
        if (rc && l) {                                      // (divergent: only lanes that lie in some mask look at its rectangle)
            uint32_t rest = l;
            while (rest) {
                const int m = __ffs(rest) - 1;
                rest &= rest - 1u;
                const int4 q = rc[m];
                if (!(ui >= q.x && ui < q.z && vi >= q.y && vi < q.w)) l &= ~(1u << m);
            }
        }
        return l;
    }
    static __device__ __forceinline__ size_t frame_stride(const LpfParams &P) { return (size_t)P.M * (size_t)P.W * (size_t)P.H; }
};

template <typename T, int MODE>
struct LpfLabelSrc<LpfDirectRect<T, MODE> > {                // masks [F][M][H][W] of T, read inside their rectangles only (lpf_k1_tile)
    typedef T elem;
    static constexpr int mode = MODE;
    static __device__ __forceinline__ uint32_t get(const elem *__restrict__, const LpfParams &, const int, const int4 *, int, int) { return 0u; }      // (not used)
    static __device__ __forceinline__ size_t frame_stride(const LpfParams &P) { return (size_t)P.M * (size_t)P.W * (size_t)P.H; }
};

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

#define LPF_RG_SHIFT 4            // cells of the candidate grid of LpfDirectRect tiles: 16 x 16 pixels
#define LPF_RG_CELL (1 << LPF_RG_SHIFT)
__device__ __forceinline__ unsigned lpf_wave_or(unsigned x);

template <int ROWS, unsigned FL, typename LT>
__device__ __forceinline__ void lpf_k1_tile(const LpfParams &P, const int blk, unsigned *s_cnt)
{
    // one block = one tile of 256*ROWS points; seg_pts / tile tiles share a segment (= one list wave of the tail)
    constexpr int TILE = LPF_BLOCK * ROWS;
    static_assert(LPF_SEG_QUANTUM % TILE == 0 && (LPF_SEG_SMALL % TILE == 0 || ROWS >= 8), "tiles must divide segments (8-row and larger tiles: large geometry only)");
    const int tid = threadIdx.x, lane = lpf_lane(), wave = lpf_wave();
    const int tiles_per_seg = P.seg_pts / TILE;
    const int lb = lpf_xcd_remap(blk, P.nseg_total * tiles_per_seg);
    const int sid = lb / tiles_per_seg;
    LpfFrame fr = P.frame0;
    if (P.F > 1) fr = P.segs[__builtin_amdgcn_readfirstlane(sid)];        // wave-uniform: scalar loads, no search
    const int f = fr.pad;
    // (from here on "point" indices are positions in the frame's rows: frame point + fr.shift, see LpfFrame)
    const int sh = fr.shift;
    const long long pbase = fr.pt_off - sh;                              // the 64-point boundary the frame's rows start at
    const int seg_start = (sid - fr.seg_off) * P.seg_pts;
    const int seg_end = min(seg_start + P.seg_pts, fr.N + sh);
    const int c = seg_start + (lb - sid * tiles_per_seg) * TILE;       // first point of the tile
    if (c >= seg_end) return;                                            // padding tile of a short segment
    const float4 *__restrict__ pts = P.pts + pbase;
    typedef LpfLabelSrc<LT> Src;
    const typename Src::elem *__restrict__ limg =
        (P.label_img && P.M > 0) ? static_cast<const typename Src::elem *>(P.label_img) + (size_t)f * Src::frame_stride(P) : nullptr;
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
        constexpr bool RECT = LpfIsDirectRect<LT>::value;
        const int4 *__restrict__ rc_f = ((RECT || LpfIsDirect<LT>::value) && P.rects) ? P.rects + (size_t)f * P.M : nullptr;
        const uint32_t *__restrict__ rgrid = (RECT && limg && P.rect_grid) ? P.rect_grid + (size_t)f * P.rg_cells : nullptr;
        const size_t mhw = (size_t)P.W * (size_t)P.H;
        unsigned long long myv = 0, mym = 0;                // lane r keeps the ballots of row r
        // what happens to a row once its label bits are known: label store, the two ballots, the hand-off entries of its masked
        // points, the per-instance counts.  Tiles that gather from a label image do this in a second loop, after every row's gather
        // has been issued; RECT tiles know a row's bits at once and keep nothing per row (no lab[] / valid[]: registers).
        auto consume = [&](const int r, uint32_t l, const bool okr) {
            const int idx = wbase + r * 64;
            if (idx >= sh && idx < seg_end && P.label_bits && !(FL & LPF_F_LAB_NOSTORE)) {
                if (FL & LPF_F_NTSTORE) __builtin_nontemporal_store(l, P.label_bits + pbase + idx);
                else P.label_bits[pbase + idx] = l;
            }
            const unsigned long long vb = __ballot(okr);
            const unsigned long long mb = __ballot(l != 0);
            if (lane == r) { myv = vb; mym = mb; }
            // the wave's masked points {x, y, z, label}, compacted in point order at the wave's own
            // first slots: the tail reads them back in runs instead of gathering from the cloud
            // (behind the dead lanes of the frame's first wave: the slots below the frame's first point are the previous frame's)
            if (l && P.mlist && !(FL & LPF_F_LAB_NOSTORE))
                P.mlist[pbase + (wbase - lane) + ((wbase - lane) == 0 ? sh : 0) + nmask_w + __popcll(mb & ((1ull << lane) - 1ull))] =
                    make_float4(p[r].x, p[r].y, p[r].z, __uint_as_float(l));
            nvalid_w += __popcll(vb);
            nmask_w += __popcll(mb);
            while (l) {                                     // rare: per-instance counts
                const int m = __ffs(l) - 1;
                l &= l - 1;
                atomicAdd(&s_cnt[2 + m], 1u);
            }
        };
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const int idx = wbase + r * 64;
            const bool live = idx >= sh && idx < seg_end;
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
            if constexpr (RECT) {
                // candidates: the masks whose rectangle meets the point's cell of the frame's coarse grid -- one 4-byte gather from a
                // table of a few KB (L1 / L2 resident), issued for every row before any is consumed, like the label-image gather
                if (ok && rgrid) lab[r] = rgrid[(vi >> LPF_RG_SHIFT) * P.rg_cw + (ui >> LPF_RG_SHIFT)];
            } else
            if (!(FL & LPF_F_LAB_NOGATHER)) {
                if (ok && limg) lab[r] = Src::get(limg, P, vi * P.W + ui, rc_f, ui, vi);
            }
            if (live && !(FL & LPF_F_LAB_NOSTORE)) {
                const long long g = pbase + idx;
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
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            uint32_t l = lab[r];
            if constexpr (RECT) {
                // Candidates -> label bits: only a row that has any (a few per cent of a real scan's rows: the cells around the cars)
                // takes the exact test -- the pixel again (not kept per row: registers), rectangle m from a scalar load, and the
                // bytes of mask m for the lanes inside it: a scalar base and a 32-bit offset per lane.  A mask counts inside its
                // rectangle only, pixel for pixel.
                if (__any(l != 0u)) {                        // (wave-uniform)
                    double uf, vf, d;
                    lpf_project_point(P, p[r].x, p[r].y, p[r].z, uf, vf, d);
                    const int ui = lpf_sat_i32(rint(uf)), vi = lpf_sat_i32(rint(vf));
                    const unsigned pix = (unsigned)(vi * P.W + ui);
                    uint32_t any = lpf_wave_or(l), l2 = 0u;
                    while (any) {                           // (wave-uniform: the masks that are a candidate of some lane)
                        const int m = __ffs(any) - 1;
                        any &= any - 1u;
                        const int4 q = rc_f[m];             // scalar load
                        const bool in = ((l >> m) & 1u) && ui >= q.x && ui < q.z && vi >= q.y && vi < q.w;
                        if (__any(in)) {
                            const typename Src::elem *__restrict__ mk = limg + (size_t)m * mhw;
                            if (in && lpf_member<typename Src::elem, Src::mode>(mk[pix])) l2 |= 1u << m;
                        }
                    }
                    l = l2;
                }
            }
            consume(r, l, valid[r]);
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
            if (v) {
                // three levels -- segment, group of 64 segments, frame (8 shards) -- so that a list wave finds its
                // segment's place in the frame's lists with two wave sums instead of a scan over the frame
                // (the record's fields for this are read again here rather than kept in registers through the whole tile: the 8-row
                //  tiles of the step kernel have none to spare)
                int seg_off = P.frame0.seg_off, grp_off = P.frame0.grp_off, fid = P.frame0.pad;
                if (P.F > 1) { const LpfFrame *e = P.segs + lpf_uni(sid); seg_off = e->seg_off; grp_off = e->grp_off; fid = e->pad; }
                const int k = sid - seg_off, g = tid >> 1, h = tid & 1;
                atomicAdd(reinterpret_cast<unsigned long long *>(P.seg_tab + (size_t)g * P.nseg_cap + sid) + h, v);
                atomicAdd(reinterpret_cast<unsigned long long *>(P.grp_tab + (size_t)g * P.ngrp_cap + grp_off + (k >> 6)) + h, v);
                atomicAdd(reinterpret_cast<unsigned long long *>(P.frm_tab + ((size_t)fid * LPF_FRM_SHARDS + (k & (LPF_FRM_SHARDS - 1))) * LPF_TAB_GROUPS + g) + h, v);
            }
        }
    }
}

template <int ROWS, unsigned FL, typename LT = uint32_t>   // LT: label-image element (uint8 for M <= 8, uint16 for M <= 16), or LpfDirect<T, MODE>
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
// The reference tests t = d / vv with 0 <= t <= 1.  For 1e-100 <= vv <= 1e100 (exact_ok, set by the host) and a d that
// is zero or at least 1e-200 in magnitude, round-to-nearest division gives
//   t >= 0 <=> d >= 0 : |d| / vv >= 1e-300 is a normal number, so the quotient keeps d's sign (a smaller |d| / vv could
//                       underflow to -0.0, which the reference counts as >= 0);  d = +-0 gives t = +-0, >= 0 both ways;
//   t <= 1 <=> d <= vv: the next double above vv is vv + ulp(vv) with ulp(vv) / vv > 2^-53 strictly (equality would need
//                       vv to be the power of two ABOVE its binade), so that quotient rounds to at least 1 + 2^-52;
// NaN fails every comparison in both forms.  Everything else (degenerate boxes, tiny d) forms the quotient.
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
        if (exact_ok && !(fabs(d) < 1e-200 && d != 0.0)) {
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
// Cross-lane sums by DPP (no LDS, no ds_bpermute): quad swaps, half-row and row mirrors, then the two
// row broadcasts -- six v_add_u32 with a DPP modifier; lanes 48..63 end up holding the wave total.
// ------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned lpf_rl(unsigned v, int l) { return (unsigned)__builtin_amdgcn_readlane((int)v, l); }
__device__ __forceinline__ float lpf_rlf(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
__device__ __forceinline__ unsigned long long lpf_rl64(unsigned long long v, int l)
{
    return (unsigned long long)lpf_rl((unsigned)v, l) | ((unsigned long long)lpf_rl((unsigned)(v >> 32), l) << 32);
}
#define LPF_DPP_ADD(x, ctrl, rows) ((x) + (unsigned)__builtin_amdgcn_update_dpp(0, (int)(x), (ctrl), (rows), 0xf, false))
__device__ __forceinline__ unsigned lpf_wave_sum(unsigned x)           // sum over the 64 lanes, wave-uniform
{
    x = LPF_DPP_ADD(x, 0xB1, 0xf);       // quad_perm [1,0,3,2]
    x = LPF_DPP_ADD(x, 0x4E, 0xf);       // quad_perm [2,3,0,1]
    x = LPF_DPP_ADD(x, 0x141, 0xf);      // row_half_mirror
    x = LPF_DPP_ADD(x, 0x140, 0xf);      // row_mirror: every lane of a row of 16 holds the row's sum
    x = LPF_DPP_ADD(x, 0x142, 0xa);      // row_bcast:15 into rows 1 and 3
    x = LPF_DPP_ADD(x, 0x143, 0xc);      // row_bcast:31 into rows 2 and 3
    return lpf_rl(x, 63);
}
__device__ __forceinline__ unsigned lpf_sum8(unsigned x)               // sum over lanes 0..7, wave-uniform
{
    x = LPF_DPP_ADD(x, 0xB1, 0xf);
    x = LPF_DPP_ADD(x, 0x4E, 0xf);
    x = LPF_DPP_ADD(x, 0x141, 0xf);
    return lpf_rl(x, 0);
}
__device__ __forceinline__ unsigned lpf_wave_or(unsigned x)            // OR over the 64 lanes, wave-uniform
{
#define LPF_DPP_OR(x, ctrl, rows) ((x) | (unsigned)__builtin_amdgcn_update_dpp(0, (int)(x), (ctrl), (rows), 0xf, false))
    x = LPF_DPP_OR(x, 0xB1, 0xf); x = LPF_DPP_OR(x, 0x4E, 0xf); x = LPF_DPP_OR(x, 0x141, 0xf); x = LPF_DPP_OR(x, 0x140, 0xf);
    x = LPF_DPP_OR(x, 0x142, 0xa); x = LPF_DPP_OR(x, 0x143, 0xc);
#undef LPF_DPP_OR
    return lpf_rl(x, 63);
}

// ------------------------------------------------------------------------------------
// SCAN (only for frames of more than 64 x 64 segments, whose prefixes one wave cannot derive from the group
// table): one block per frame; turns the per-segment counters K1 accumulated into
//   seg_pre[g][seg] : exclusive prefix over the frame's earlier segments; for instance counters the
//                     frame-level list offset inst_off[m] is already added.
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

__global__ __launch_bounds__(LPF_BLOCK) void lpf_scan_segments(const LpfParams P)
{
    // thread t owns the contiguous run of R = ceil(nseg/256) segments starting at t*R:
    // phase 1 sums each run, phase 2 re-reads the runs (L2-hot) and writes the prefixes
    __shared__ unsigned s_toff[LPF_TAB_GROUPS][LPF_BLOCK][4];
    __shared__ unsigned s_wsum[4][4], s_tot[LPF_TAB_ROWS], s_off[LPF_TAB_ROWS];
    const int f = blockIdx.x, tid = threadIdx.x, lane = lpf_lane(), wave = lpf_wave();
    const LpfFrame fr = lpf_frame_record(P.frame0, P.frames, P.F > 1, f);
    const int seg_lo = fr.seg_off, seg_hi = fr.seg_off + fr.nseg;
    const int ngroups = (2 + P.M + 3) >> 2;
    const int R = (fr.nseg + LPF_BLOCK - 1) / LPF_BLOCK;
    const int my_lo = min(seg_lo + tid * R, seg_hi), my_hi = min(my_lo + R, seg_hi);
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
        for (int j = 0; j < 4; ++j) { s_toff[g][tid][j] = excl[j]; if (tid == 0) s_tot[4 * g + j] = all[j]; }
    }
    __syncthreads();
    if (tid < LPF_TAB_ROWS) {
        const bool used = tid < 4 * ngroups;
        unsigned off = 0;                                  // inst_off[m] for the instance counters
        if (used) for (int c = 2; c < tid; ++c) off += s_tot[c];
        s_off[tid] = off;
    }
    __syncthreads();
    for (int g = 0; g < ngroups; ++g) {
        const uint4 *__restrict__ row = P.seg_tab + (size_t)g * P.nseg_cap;
        uint4 *__restrict__ pre = P.seg_pre + (size_t)g * P.nseg_cap;
        unsigned run[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) run[j] = s_off[4 * g + j] + s_toff[g][tid][j];
#pragma unroll 8
        for (int sg = my_lo; sg < my_hi; ++sg) {
            const uint4 v = row[sg];
            pre[sg] = make_uint4(run[0], run[1], run[2], run[3]);
            run[0] += v.x; run[1] += v.y; run[2] += v.z; run[3] += v.w;
        }
    }
}

// ------------------------------------------------------------------------------------
// LISTS: one WAVE = one segment (seg_pts points = seg_pts/64 ballot rows, one per lane); four independent
// waves per block, no block barriers.  From the ballots K1 left it writes the stable index lists:
//   valid_idx (+ the compact uv_valid / label_valid)                                    (V3:585, 590-592)
//   inst_idx, one list per mask                                                       (V3:225-228)
// Where a segment's entries start in its frame's lists is derived by the wave itself from the three
// levels of counters K1's tiles added into (segment, group of 64 segments, frame): two masked wave sums
// per counter, no scan kernel in between (PRE = true: frames with more than 64 groups, see lpf_scan_segments).
// Memory round trips on a wave's critical path: {segment record, ballots, counters} -> {hand-off
// entries of the masked points} -> stores.
// ------------------------------------------------------------------------------------
#define LPF_LISTS_WAVES 4
#define LPF_LIST_CAP 1024            // masked entries staged in LDS per pass (2 KB per wave)


// Set bits of the ballots of rows (lane r holds row r; rowbase = bits set in earlier rows of the pass) ->
// ascending list in LDS.  Every lane walks its own row: work is O(set bits), not O(rows x 64).
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

// lane c: entries of counter c (0 valid, 1 masked, 2+m instance m) in the frame before segment sid (bef) and in the whole frame
// (tot; PRE: bef of the instance counters already includes inst_off[m], tot is not produced).  Without the scan kernel the
// wave sums its group's segments and the frame's groups, a lane each.  STEP = 2 (one round trip up to six masks) is what the
// software-pipelined step kernel uses (98.1 vs 98.6 us per step); in the stand-alone tail kernels the same code measured
// worse on small launches (45.2 vs 41.3 us for a batch of 20 real frames, even when only present, not executed), so they
// take the groups one at a time.
template <bool PRE, int STEP>               // STEP: counter groups whose loads are in flight together (1 or 2)
__device__ __forceinline__ void lpf_list_prefix(const LpfParams &P, const LpfFrame &fr, const int sid, unsigned &bef, unsigned &tot)
{
    const int lane = lpf_lane();
    const int ngroups = (2 + P.M + 3) >> 2;
    const int k = sid - fr.seg_off;
    bef = 0; tot = 0;
    if (PRE) {
        uint4 q = make_uint4(0u, 0u, 0u, 0u);
        if (lane < ngroups) q = P.seg_pre[(size_t)lane * P.nseg_cap + sid];
        const int g = min(lane >> 2, LPF_TAB_GROUPS - 1);
        const unsigned x = __shfl(q.x, g), y = __shfl(q.y, g), z = __shfl(q.z, g), w = __shfl(q.w, g);
        bef = ((lane & 3) == 0) ? x : ((lane & 3) == 1) ? y : ((lane & 3) == 2) ? z : w;   // instance counters: inst_off[m] included
    } else {
        const int gi = k >> 6, j = k & 63;                 // group of 64 segments, place in it (gi < 64: host guarantees)
        for (int g0 = 0; g0 < ngroups; g0 += STEP) {
            uint4 a[STEP], b[STEP], t[STEP];
#pragma unroll
            for (int u = 0; u < STEP; ++u) {
                const int g = g0 + u;
                a[u] = make_uint4(0u, 0u, 0u, 0u); b[u] = a[u]; t[u] = a[u];
                if (g < ngroups) {
                    if (lane < j) a[u] = P.seg_tab[(size_t)g * P.nseg_cap + fr.seg_off + (gi << 6) + lane];
                    if (lane < gi) b[u] = P.grp_tab[(size_t)g * P.ngrp_cap + fr.grp_off + lane];
                    if (lane < LPF_FRM_SHARDS) t[u] = P.frm_tab[((size_t)fr.pad * LPF_FRM_SHARDS + lane) * LPF_TAB_GROUPS + g];
                }
            }
#pragma unroll
            for (int u = 0; u < STEP; ++u) {
                const int g = g0 + u;
                if (g >= ngroups) break;
                const unsigned px = lpf_wave_sum(a[u].x + b[u].x), py = lpf_wave_sum(a[u].y + b[u].y), pz = lpf_wave_sum(a[u].z + b[u].z), pw = lpf_wave_sum(a[u].w + b[u].w);
                const unsigned tx = lpf_sum8(t[u].x), ty = lpf_sum8(t[u].y), tz = lpf_sum8(t[u].z), tw = lpf_sum8(t[u].w);
                if ((lane >> 2) == g) {
                    bef = ((lane & 3) == 0) ? px : ((lane & 3) == 1) ? py : ((lane & 3) == 2) ? pz : pw;
                    tot = ((lane & 3) == 0) ? tx : ((lane & 3) == 1) ? ty : ((lane & 3) == 2) ? tz : tw;
                }
            }
        }
    }
}

// lane m: where the next entry of mask m goes in the frame's concatenated instance lists
template <bool PRE>
__device__ __forceinline__ unsigned lpf_list_posreg(const LpfParams &P, const unsigned bef, const unsigned tot)
{
    const int lane = lpf_lane();
    if (PRE) return (unsigned)__shfl((int)bef, (lane + 2) & 63);
    unsigned off = (lane >= 2 && lane < 2 + P.M) ? tot : 0u;      // inst_off[m] = totals of the masks before m
    const unsigned own = off;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned u = __shfl_up(off, o);
        if (lane >= o) off += u;
    }
    return (unsigned)__shfl((int)(off - own + bef), (lane + 2) & 63);
}

template <bool PRE, int STEP>
__device__ __forceinline__ void lpf_lists_wave(const LpfParams &P, const LpfFrame &fr, const int sid, unsigned short *lst)
{
    const int lane = lpf_lane();
    const unsigned long long lt = (1ull << lane) - 1ull;
    const int rps = P.seg_pts >> 6;                        // ballot rows per segment: 16 or 64
    const int k = sid - fr.seg_off;                        // segment of its frame
    unsigned long long vb = 0, mb = 0;
    if (lane < rps) {
        vb = P.vbal[(size_t)sid * rps + lane];
        mb = P.mbal[(size_t)sid * rps + lane];
    }
    unsigned bef, tot;
    lpf_list_prefix<PRE, STEP>(P, fr, sid, bef, tot);
    const int sh = fr.shift;                               // (row positions are frame points + shift: LpfFrame)
    const int seg_start = k * P.seg_pts;
    const int seg_end = min(seg_start + P.seg_pts, fr.N + sh);
    const int nrows = (seg_end - seg_start + 63) >> 6;
    if (lane >= nrows) { vb = 0; mb = 0; }                 // rows K1 never wrote
    const unsigned cv = __popcll(vb), cm = __popcll(mb);
    unsigned iv = cv, im = cm;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {                     // inclusive scan over the row counts
        const unsigned tv = __shfl_up(iv, o), tm = __shfl_up(im, o);
        if (lane >= o) { iv += tv; im += tm; }
    }
    const unsigned vbase = iv - cv, mbase = im - cm;
    const unsigned nv = lpf_rl(iv, 63), L = lpf_rl(im, 63);
    const long long run_v = (long long)lpf_rl(bef, 0);     // valid points of the frame before this segment

    // ---- valid_idx: ascending by construction.  Sparse segment: every lane walks the set bits of its own row into the
    //      wave's LDS list (as many steps as the fullest row has bits), which then leaves as whole 512-byte runs -- 8-byte
    //      stores scattered over 64 lines cost four times their bytes in HBM writes (measured: 22.7 MB for 7.3 MB of
    //      lists); dense one: row after row, a row's entries are one contiguous run already. ---------------------------
    if (P.valid_idx && nv) {
        long long *__restrict__ dst = P.valid_idx + fr.pt_off + run_v;
        const long long o = fr.pt_off + run_v, g0 = fr.pt_off - sh + seg_start;
        // run_v comes from counters in memory: whatever they hold, a store never leaves the frame's own N slots (with sound
        // counters run_v + nv <= N always; see DESIGN.md section 9 for the fault this guard is the answer to)
        const long long room = (long long)fr.N - run_v;
        if (nv > (unsigned)LPF_LIST_CAP || nv > 6u * (unsigned)nrows) {
            for (int r = 0; r < nrows; ++r) {
                const unsigned long long rv = lpf_rl64(vb, r);                      // wave-uniform
                if (rv == 0ull) continue;                   // (a real scan: rows are full or empty -- half of them hold no valid point)
                const long long pos = lpf_rl(vbase, r) + __popcll(rv & lt);
                if (((rv >> lane) & 1ull) && pos < room) {
                    dst[pos] = (long long)(seg_start + r * 64 + lane - sh);
                    if (P.uv_valid) P.uv_valid[o + pos] = P.uv[g0 + r * 64 + lane];
                    if (P.label_valid) P.label_valid[o + pos] = P.label_bits[g0 + r * 64 + lane];
                }
            }
        } else {
            lpf_bits_to_list(vb, vbase, lane, lst);
            __builtin_amdgcn_wave_barrier();
            for (unsigned e = lane; e < nv && (long long)e < room; e += 64) {
                const int pt = seg_start + (int)lst[e] - sh;
                dst[e] = (long long)pt;
                if (P.uv_valid) P.uv_valid[o + e] = P.uv[fr.pt_off + pt];
                if (P.label_valid) P.label_valid[o + e] = P.label_bits[fr.pt_off + pt];
            }
            __builtin_amdgcn_wave_barrier();               // the list is reused for the masked points
        }
    }
    if (L == 0 || P.inst_idx == nullptr) return;

    unsigned posreg = lpf_list_posreg<PRE>(P, bef, tot);
    // K1 left each of its waves' masked points {x, y, z, label} compacted at the wave's first slot, in
    // the same order as the set bits of the masked ballots: entry e of the segment, found in
    // row r, is entry e - mbase[first row of r's K1 wave] of that wave.
    const float4 *__restrict__ mseg = P.mlist + (fr.pt_off - sh) + seg_start;
    const int rows_per_wave = P.tile_pts >> 8;             // K1 tile = 4 waves of tile_pts/4 points
    const int rpw_shift = (rows_per_wave == 8) ? 3 : (rows_per_wave == 4) ? 2 : (rows_per_wave == 2) ? 1 : 0;
    const int dead0 = (seg_start == 0) ? sh : 0;           // the frame's first K1 wave keeps its entries behind its dead lanes

    for (int r0 = 0; r0 < nrows;) {                        // passes of at most LPF_LIST_CAP entries (one, except on very dense segments)
        const unsigned start = lpf_rl(mbase, r0);
        const unsigned long long fit = __ballot(lane >= r0 && lane < nrows && (im - start) <= (unsigned)LPF_LIST_CAP);
        const int r1 = r0 + max(__popcll(fit), 1);         // rows [r0, r1): im is non-decreasing, so the fitting rows are a run
        const unsigned cnt = lpf_rl(im, r1 - 1) - start;
        if (cnt) {
            if (cnt > 6u * (unsigned)(r1 - r0)) {           // dense: row after row, lane = point
                for (int r = r0; r < r1; ++r) {
                    const unsigned long long rm = lpf_rl64(mb, r);
                    if ((rm >> lane) & 1ull) lst[lpf_rl(mbase, r) - start + __popcll(rm & lt)] = (unsigned short)(r * 64 + lane);
                }
            } else if (lane >= r0 && lane < r1) {
                lpf_bits_to_list(mb, mbase - start, lane, lst);
            }
            __builtin_amdgcn_wave_barrier();               // same wave, in-order LDS queue: reads below see the writes
            // four chunks of 64 entries at a time: their label reads (one dependent round trip each) are issued together,
            // then split one after the other -- chunk by chunk, a segment lying on a car (hundreds of masked points) cost
            // the wave a memory latency per chunk, the longest chain of a small launch's tail (8.4 us measured)
            for (unsigned e0 = 0; e0 < cnt; e0 += 256) {
                unsigned lab4[4], li4[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    lab4[q] = 0u; li4[q] = 0u;
                    if (e0 + 64u * q >= cnt) continue;     // (wave-uniform: most segments of a sparse launch hold one chunk)
                    const unsigned e = e0 + 64u * q + lane;
                    const bool act = e < cnt;
                    const unsigned li = lst[act ? e : 0];  // segment-relative point index
                    const int first_row = (int)((li >> 6) >> rpw_shift) << rpw_shift;
                    const unsigned wb = (unsigned)__shfl((int)mbase, first_row);        // all lanes take part
                    li4[q] = li;                           // (only the label bits of the hand-off entry are needed here)
                    if (act) lab4[q] = __float_as_uint(mseg[first_row * 64 + (first_row == 0 ? dead0 : 0) + (int)(start + e - wb)].w);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {              // split by instance; ballot order == ascending point index
                    if (e0 + 64u * q >= cnt) continue;
                    const unsigned lab = lab4[q];
                    const long long idx = (long long)(seg_start + (int)li4[q] - sh);
                    unsigned any = lpf_wave_or(lab);
                    while (any) {
                        const int m = __ffs(any) - 1;
                        any &= any - 1u;
                        const bool hit = (lab >> m) & 1u;
                        const unsigned long long bal = __ballot(hit);
                        const long long base = (long long)lpf_rl(posreg, m);
                        if (hit) {
                            const long long w = base + __popcll(bal & lt);
                            if (w < P.inst_cap) P.inst_idx[fr.inst_base + w] = idx;
                        }
                        if (lane == m) posreg += (unsigned)__popcll(bal);
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();               // the list is rewritten by the next pass
        }
        r0 = r1;
    }
}

// The same for a segment of a SMALL launch (16 rows of 64 points), where nothing but latency counts: no LDS list, no
// chunks -- the rows are walked one after the other with lane = point.  The label of every masked point is read from the
// hand-off slots as soon as the ballots are in (its slot follows from the row prefixes alone), so those reads and the
// counter reads are one round trip; then valid_idx row by row, then each row with masked points is split by instance
// (rows ascending, lanes ascending = point order).  A frame's slowest list wave: 8.4 -> 4.6 us.
template <bool PRE>
__device__ __forceinline__ void lpf_lists_wave_small(const LpfParams &P, const LpfFrame &fr, const int sid)
{
    constexpr int RPS = LPF_SEG_SMALL >> 6;
    const int lane = lpf_lane();
    const unsigned long long lt = (1ull << lane) - 1ull;
    const int k = sid - fr.seg_off;
    unsigned long long vb = 0, mb = 0;
    if (lane < RPS) {
        vb = P.vbal[(size_t)sid * RPS + lane];
        mb = P.mbal[(size_t)sid * RPS + lane];
    }
    const int sh = fr.shift;                               // (row positions are frame points + shift: LpfFrame)
    const int seg_start = k * LPF_SEG_SMALL;
    const int seg_end = min(seg_start + LPF_SEG_SMALL, fr.N + sh);
    const int nrows = (seg_end - seg_start + 63) >> 6;
    if (lane >= nrows) { vb = 0; mb = 0; }                 // rows K1 never wrote
    const unsigned cv = __popcll(vb), cm = __popcll(mb);
    unsigned iv = cv, im = cm;
#pragma unroll
    for (int o = 1; o < RPS; o <<= 1) {                    // inclusive scan over the row counts (lanes >= 16 hold zeros)
        const unsigned tv = __shfl_up(iv, o), tm = __shfl_up(im, o);
        if (lane >= o) { iv += tv; im += tm; }
    }
    const unsigned vbase = iv - cv, mbase = im - cm;
    const unsigned nv = lpf_rl(iv, RPS - 1), L = lpf_rl(im, RPS - 1);
    const bool want_inst = L != 0 && P.inst_idx != nullptr;
    unsigned labr[RPS];
    if (want_inst) {
        // K1 left each of its waves' masked points compacted at the wave's first slot, in ballot order: the masked point of
        // (row r, lane l) is entry mbase[r] + popc(bits below l) of the segment, entry - mbase[first row of r's K1 wave] of that wave
        const float4 *__restrict__ mseg = P.mlist + (fr.pt_off - sh) + seg_start;
        const int rows_per_wave = P.tile_pts >> 8;
        const int rpw_shift = (rows_per_wave == 8) ? 3 : (rows_per_wave == 4) ? 2 : (rows_per_wave == 2) ? 1 : 0;
#pragma unroll
        for (int r = 0; r < RPS; ++r) {
            const unsigned long long rm = lpf_rl64(mb, r);                              // wave-uniform
            labr[r] = 0u;
            if (rm) {
                const int first_row = (r >> rpw_shift) << rpw_shift;
                const unsigned at = lpf_rl(mbase, r) - lpf_rl(mbase, first_row) + __popcll(rm & lt) +
                                    ((first_row == 0 && seg_start == 0) ? (unsigned)sh : 0u);     // (behind the dead lanes of the frame's first K1 wave)
                if ((rm >> lane) & 1ull) labr[r] = __float_as_uint(mseg[first_row * 64 + (int)at].w);
            }
        }
    }
    unsigned bef, tot;
    lpf_list_prefix<PRE, 1>(P, fr, sid, bef, tot);
    const long long run_v = (long long)lpf_rl(bef, 0);     // valid points of the frame before this segment
    if (P.valid_idx && nv) {
        long long *__restrict__ dst = P.valid_idx + fr.pt_off + run_v;
        const long long o = fr.pt_off + run_v, g0 = fr.pt_off - sh + seg_start;
        // run_v comes from counters in memory: whatever they hold, a store never leaves the frame's own N slots (with sound
        // counters run_v + nv <= N always; see DESIGN.md section 9 for the fault this guard is the answer to)
        const long long room = (long long)fr.N - run_v;
        for (int r = 0; r < nrows; ++r) {
            const unsigned long long rv = lpf_rl64(vb, r);                              // wave-uniform
            if (rv == 0ull) continue;
            const long long pos = lpf_rl(vbase, r) + __popcll(rv & lt);
            if (((rv >> lane) & 1ull) && pos < room) {
                dst[pos] = (long long)(seg_start + r * 64 + lane - sh);
                if (P.uv_valid) P.uv_valid[o + pos] = P.uv[g0 + r * 64 + lane];
                if (P.label_valid) P.label_valid[o + pos] = P.label_bits[g0 + r * 64 + lane];
            }
        }
    }
    if (!want_inst) return;
    unsigned posreg = lpf_list_posreg<PRE>(P, bef, tot);
#pragma unroll
    for (int r = 0; r < RPS; ++r) {
        const unsigned lab = labr[r];
        unsigned any = lpf_wave_or(lab);
        const long long idx = (long long)(seg_start + r * 64 + lane - sh);
        while (any) {
            const int m = __ffs(any) - 1;
            any &= any - 1u;
            const bool hit = (lab >> m) & 1u;
            const unsigned long long bal = __ballot(hit);
            const long long base = (long long)lpf_rl(posreg, m);
            if (hit) {
                const long long w = base + __popcll(bal & lt);
                if (w < P.inst_cap) P.inst_idx[fr.inst_base + w] = idx;
            }
            if (lane == m) posreg += (unsigned)__popcll(bal);
        }
    }
}

// ------------------------------------------------------------------------------------
// BOX COUNT: count_mb[m][b] = number of points of mask m inside box b (V3:344-376: np.sum(oriented_point_in_bbox(...))).
// One WAVE = one segment, like the lists -- and independent of them: the wave compacts its segment's masked points
// itself (entry e -> ballot row by a binary search over the row prefixes -> K1 wave slot -> hand-off entry), so the
// two run side by side in one launch.  The frame's ground grids (built with the boxes, one per 64 of them) list, per cell of the
// (x, y) plane, the boxes whose bounds reach into it; a point only meets those of its cell.  Candidates pass a conservative float
// AABB of the region first, the survivors are queued per wave and take the reference's f64 test a whole wave at a
// time.  Hits are counted in the block's LDS counters and flushed once per block.
// ------------------------------------------------------------------------------------
#define LPF_BC_WORD 64            // boxes of a box-count block: ONE 64-box word of its frame (one ground grid).  A frame with more boxes gets a
                                  // block per word (and 4 segments): each keeps its 64 boxes' float bounds (24 bytes each), exact parameters
                                  // (128 bytes each) and inside counters (16 bits each, M x 64) in LDS -- whatever the frame's box count, the
                                  // candidate loop and the exact tests never read box data from memory.  (With the frame's first 64 / 32 boxes
                                  // staged and the rest read from memory, a chunk of 64 masked points of a 314-box frame took ~30 us.)

// row prefixes of a segment's masked ballots: lane r -> entries in rows 0..r (im) and before row r (mbase); returns the total
// The candidate structure: per (frame, word of 64 boxes) a GROUND grid -- LPF_GRID x LPF_GRID cells over the (x, y) extent of the
// word's boxes in the velodyne frame, a 64-bit box set per cell, and behind the cells the grid's domain {x0, y0, 1/cell_x, 1/cell_y}
// as four floats.  Seen from above, annotated objects hardly overlap (seen from the camera, everything along a ray does: a street
// of 300 boxes put 50 candidates on a point).  A box is entered in the cells [cell(lo), cell(hi)] of its float bounds, a point
// looks into cell(p): the SAME monotone function on both sides, so lo <= p <= hi (the float bounds test every candidate takes
// first, itself a superset of the exact test) implies the box is in the point's cell.
#define LPF_GRID 32              // (cells of a metre at least: see the domain in lpf_box_frame_block)
#define LPF_GRID_CELLS (LPF_GRID * LPF_GRID)
#define LPF_GRID_WORDS (LPF_GRID_CELLS + 2)
__device__ __forceinline__ int lpf_ground_cell(const float v, const float v0, const float inv)
{
    float t = (v - v0) * inv;                               // monotone in v (inv >= 0); NaN (inf * 0) -> cell 0 below
    t = fminf(fmaxf(t, 0.f), (float)(LPF_GRID - 1));
    return (int)t;
}

__device__ __forceinline__ unsigned lpf_count_rows(const LpfParams &P, const LpfFrame &fr, const int sid, unsigned &im, unsigned &mbase)
{
    const int lane = lpf_lane();
    const int rps = P.seg_pts >> 6;
    unsigned long long mb = 0;
    if (lane < rps) mb = P.mbal[(size_t)sid * rps + lane];
    const int seg_start = (sid - fr.seg_off) * P.seg_pts;
    const int nrows = (min(seg_start + P.seg_pts, fr.N + fr.shift) - seg_start + 63) >> 6;       // (row positions: frame points + shift)
    if (lane >= nrows) mb = 0;                             // rows K1 never wrote
    const unsigned cm = __popcll(mb);
    im = cm;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {                     // inclusive scan over the row counts
        const unsigned tm = __shfl_up(im, o);
        if (lane >= o) im += tm;
    }
    mbase = im - cm;
    return lpf_rl(im, 63);
}

// one chunk of 64 masked points (entries e0 .. e0+63 of segment sid, L in all) against candidate word wd of the frame's grid:
// boxes 64 wd .. 64 wd + 63, whose bounds / exact parameters / counters the block holds in LDS under their index in the word
__device__ __forceinline__ void lpf_count_chunk(const LpfParams &P, const LpfFrame &fr, const int sid, const unsigned im, const unsigned mbase,
                                                const unsigned L, const unsigned e0, float4 *s_pt, unsigned *qq, unsigned *s_cnt,
                                                const float *s_bq, const double *s_bp, const float *s_dom, const int wd)
{
    const int lane = lpf_lane();
    const unsigned long long lt = (1ull << lane) - 1ull;
    const float4 *__restrict__ mseg = P.mlist + (fr.pt_off - fr.shift) + (size_t)(sid - fr.seg_off) * P.seg_pts;
    const int rows_per_wave = P.tile_pts >> 8;
    const int rpw_shift = (rows_per_wave == 8) ? 3 : (rows_per_wave == 4) ? 2 : (rows_per_wave == 2) ? 1 : 0;
    auto exact = [&](int count) {
        bool in = false;
        int j = 0;
        unsigned lb = 0u;
        if (lane < count) {
            const unsigned ent = qq[lane];
            const int e = (int)(ent & 63u);                                         // point of the chunk,
            j = (int)(ent >> 6);                                                    // box of the word
            const float4 x = s_pt[e];
            const double px = (double)x.x, py = (double)x.y, pz = (double)x.z;
            const double *bp = s_bp + j * 16;
            in = P.oriented ? lpf_oriented_inside(px, py, pz, bp) : lpf_aabb_inside(px, py, pz, bp);
            lb = __float_as_uint(x.w);
        }
        // Counting.  The pairs of a round are mostly ONE box and ONE mask -- the k-th candidate of 64 neighbouring points on the same
        // car -- so 64 lanes adding 1 to the same LDS word is the common case and the slow one (a same-address atomic per lane, one
        // after the other: a parked car annotated 17 times cost a chunk 17 such rounds, ~4 us).  Such a round adds its popcount once.
        const unsigned long long inb = __ballot(in);
        if (inb == 0ull) return;
        const int src = __ffsll((long long)inb) - 1;
        const int j0 = __builtin_amdgcn_readlane(j, src);
        const unsigned l0 = (unsigned)__builtin_amdgcn_readlane((int)lb, src);
        if (__all(!in || (j == j0 && lb == l0))) {
            if (lane == src) {
                const unsigned n = (unsigned)__popcll(inb);
                unsigned l = l0;
                while (l) {
                    const int m = __ffs(l) - 1;
                    l &= l - 1;
                    const int ci = m * LPF_BC_WORD + j0;
                    atomicAdd(&s_cnt[ci >> 1], (ci & 1) ? (n << 16) : n);
                }
            }
        } else if (in) {
            unsigned l = lb;
            while (l) {
                const int m = __ffs(l) - 1;
                l &= l - 1;
                const int ci = m * LPF_BC_WORD + j;
                atomicAdd(&s_cnt[ci >> 1], (ci & 1) ? 0x10000u : 1u);
            }
        }
    };
    const unsigned e = e0 + lane;
    const bool act = e < L;
    // row of entry e: the first row whose inclusive prefix exceeds e (im is non-decreasing over the lanes)
    int row = 0;
#pragma unroll
    for (int st = 32; st > 0; st >>= 1) {
        const unsigned v = (unsigned)__shfl((int)im, row + st - 1);
        if (v <= e) row += st;
    }
    row = min(row, 63);
    const int first_row = (row >> rpw_shift) << rpw_shift;
    const unsigned wb = (unsigned)__shfl((int)mbase, first_row);
    float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
    if (act) p = mseg[first_row * 64 + ((first_row == 0 && sid == fr.seg_off) ? fr.shift : 0) + (int)(e - wb)];    // (behind the dead lanes of the frame's first K1 wave)
    __builtin_amdgcn_wave_barrier();
    s_pt[lane] = p;                                         // .w carries the label bits (0 for idle lanes)
    __builtin_amdgcn_wave_barrier();
    int qn = 0;                                             // wave-uniform queue length
    const int cell = lpf_ground_cell(p.y, s_dom[1], s_dom[3]) * LPF_GRID + lpf_ground_cell(p.x, s_dom[0], s_dom[2]);
    unsigned long long mset = act ? P.cand[fr.cand_off + (size_t)wd * LPF_GRID_WORDS + cell] : 0ull;
    // pairs that pass the float bounds are queued; a full queue takes the exact test a whole wave at a time
    auto enqueue = [&](const bool near, const int j) {
        const unsigned long long bal = __ballot(near);
        if (!bal) return;
        if (near) qq[qn + __popcll(bal & lt)] = (unsigned)lane | ((unsigned)j << 6);
        qn += __popcll(bal);
        if (qn >= 64) {
            __builtin_amdgcn_wave_barrier();
            exact(64);
            const unsigned rest = qq[64 + lane];
            __builtin_amdgcn_wave_barrier();
            qq[lane] = rest;
            qn -= 64;
            __builtin_amdgcn_wave_barrier();
        }
    };
    // TWO candidates of every lane per round, their bounds read together: the loop is a chain of LDS round trips (bounds -> compare
    // -> ballot -> queue), and a cell in a street of parked cars holds dozens of candidates (KITTI-360 annotates a car once per
    // timestamp: 17 boxes on one parked car)
    while (__any(mset != 0ull)) {
        const bool has1 = mset != 0ull;
        const int j1 = has1 ? __ffsll((long long)mset) - 1 : 0;
        mset &= mset - 1ull;
        const bool has2 = mset != 0ull;
        const int j2 = has2 ? __ffsll((long long)mset) - 1 : 0;
        mset &= mset - 1ull;
        const float *s1 = s_bq + 6 * j1, *s2 = s_bq + 6 * j2;
        const float a0 = s1[0], a1 = s1[1], a2 = s1[2], a3 = s1[3], a4 = s1[4], a5 = s1[5];
        const float b0 = s2[0], b1 = s2[1], b2 = s2[2], b3 = s2[3], b4 = s2[4], b5 = s2[5];
        const bool near1 = has1 && p.x >= a0 && p.x <= a3 && p.y >= a1 && p.y <= a4 && p.z >= a2 && p.z <= a5;
        const bool near2 = has2 && p.x >= b0 && p.x <= b3 && p.y >= b1 && p.y <= b4 && p.z >= b2 && p.z <= b5;
        enqueue(near1, j1);
        enqueue(near2, j2);
    }
    __builtin_amdgcn_wave_barrier();
    exact(qn);
    __builtin_amdgcn_wave_barrier();
}

// a wave takes a whole segment (big sparse launches: a chunk or so per segment)
__device__ __forceinline__ void lpf_boxcount_wave(const LpfParams &P, const LpfFrame &fr, const int sid, float4 *s_pt, unsigned *qq,
                                                  unsigned *s_cnt, const float *s_bq, const double *s_bp, const float *s_dom, const int wd)
{
    unsigned im, mbase;
    const unsigned L = lpf_count_rows(P, fr, sid, im, mbase);
    for (unsigned e0 = 0; e0 < L; e0 += 64) lpf_count_chunk(P, fr, sid, im, mbase, L, e0, s_pt, qq, s_cnt, s_bq, s_bp, s_dom, wd);
}

// Box data of word wd of the frame (boxes 64 wd ..) and the domain of its ground grid -> LDS, the block's counters zeroed; and, after the
// counting, the counters flushed into the frame's [M][B] counts.  nthr threads take part.
__device__ __forceinline__ void lpf_count_stage(const LpfParams &P, const LpfFrame &fr, const int wd, const int tid, const int nthr,
                                                unsigned *s_cnt, float *s_bq, double *s_bp, float *s_dom)
{
    const int b0 = wd * LPF_BC_WORD, nb = min(fr.B - b0, LPF_BC_WORD);
    const float *__restrict__ bqf = P.boxq + ((size_t)fr.box_off + b0) * 8;          // 8 floats per box: {lo xyz, -, hi xyz, -}
    const double *__restrict__ boxp = P.boxp + ((size_t)fr.box_off + b0) * 16;
    for (int i = tid; i < (P.M * LPF_BC_WORD + 1) >> 1; i += nthr) s_cnt[i] = 0u;
    for (int i = tid; i < nb * 6; i += nthr) { const int bx = i / 6, j = i - 6 * bx; s_bq[i] = bqf[8 * bx + (j < 3 ? j : j + 1)]; }
    for (int i = tid; i < nb * 16; i += nthr) s_bp[i] = boxp[i];
    if (tid < 4) s_dom[tid] = reinterpret_cast<const float *>(P.cand + fr.cand_off + (size_t)wd * LPF_GRID_WORDS + LPF_GRID_CELLS)[tid];
}
__device__ __forceinline__ void lpf_count_flush(const LpfParams &P, const LpfFrame &fr, const int wd, const int tid, const int nthr, const unsigned *s_cnt)
{
    const int b0 = wd * LPF_BC_WORD, nb = min(fr.B - b0, LPF_BC_WORD);
    unsigned *__restrict__ cnt = P.cnt + (size_t)P.M * fr.box_off;
    for (int i = tid; i < P.M * LPF_BC_WORD; i += nthr) {
        const int m = i / LPF_BC_WORD, j = i - m * LPF_BC_WORD;
        const unsigned v = (s_cnt[i >> 1] >> (16 * (i & 1))) & 0xFFFFu;
        if (v && j < nb) atomicAdd(&cnt[m * fr.B + b0 + j], v);
    }
}

// ------------------------------------------------------------------------------------
// FINALIZE (one block per frame): first strict maximum over the boxes + per-frame summary.
// Layout of lpf_frame_summary (include/lpf.h), in int64 words: [0] n_valid  [1] n_labelled  [2..33] inst_count
// [34..66] inst_off  [67..98] best_cnt, then int32: best_box[32], inst_overflow, reserved  => 928 bytes
// ------------------------------------------------------------------------------------
#define LPF_SUMMARY_BYTES 928

#define LPF_FIN_STAGE 2048        // inside counts staged in LDS (M x B up to this many: one memory round trip for everything)

__device__ __forceinline__ void lpf_finalize_frame(const LpfParams &P, const LpfFrame &fr, const int f, unsigned *s_tot, unsigned *s_c)
{
    const int tid = threadIdx.x, lane = lpf_lane(), wave = lpf_wave();
    const int B = fr.B, M = P.M, MB = M * B;
    char *base = P.summary ? (char *)P.summary + (size_t)f * LPF_SUMMARY_BYTES : nullptr;
    long long *w = (long long *)base;
    int32_t *bb = base ? (int32_t *)(base + 99 * 8) : nullptr;
    unsigned *__restrict__ cnt = P.cnt + (size_t)M * fr.box_off;
    int32_t *out = P.count_out ? P.count_out + (size_t)M * fr.box_off : nullptr;
    // The inside counts go through LDS in passes of gm whole masks (gm x B <= LPF_FIN_STAGE; nearly always ONE pass with all of
    // them -- a crowded street scene of 300 boxes x 20 masks takes four); only B > LPF_FIN_STAGE reads them from memory.
    const bool staged = B <= LPF_FIN_STAGE;
    const int gm = !staged ? 0 : (B > 0 ? min(M, LPF_FIN_STAGE / B) : M);
    // ONE round trip: the frame's totals (sum of the 8 shards K1's tiles added into) and the inside counts (first pass) are
    // loaded together; the counts go to the caller and to LDS, their scratch is handed back zeroed
    const int ngroups = (2 + M + 3) >> 2;
    unsigned a = 0, cv[LPF_FIN_STAGE / LPF_BLOCK];
    if (tid < 4 * ngroups)
        for (int sh = 0; sh < LPF_FRM_SHARDS; ++sh)
            a += reinterpret_cast<const unsigned *>(P.frm_tab + ((size_t)f * LPF_FRM_SHARDS + sh) * LPF_TAB_GROUPS)[tid];
    auto stage_load = [&](int m0) {                         // counts of masks m0 .. m0 + gm - 1 -> registers (independent loads)
        const int n = (min(M, m0 + gm) - m0) * B;
#pragma unroll
        for (int k = 0; k < LPF_FIN_STAGE / LPF_BLOCK; ++k) cv[k] = (tid + k * LPF_BLOCK < n) ? cnt[m0 * B + tid + k * LPF_BLOCK] : 0u;
    };
    auto stage_store = [&](int m0) {                        // -> LDS, the caller's counts; the scratch zeroed
        const int n = (min(M, m0 + gm) - m0) * B;
#pragma unroll
        for (int k = 0; k < LPF_FIN_STAGE / LPF_BLOCK; ++k) {
            const int i = tid + k * LPF_BLOCK;
            if (i < n) { s_c[i] = cv[k]; if (out) out[m0 * B + i] = (int32_t)cv[k]; cnt[m0 * B + i] = 0u; }
        }
    };
    if (staged) stage_load(0);
    if (tid < LPF_TAB_ROWS) s_tot[tid] = (tid < 2 + M) ? a : 0u;
    if (staged) stage_store(0);
    __syncthreads();
    const unsigned *__restrict__ tot = s_tot;
    {   // ... and the three counter levels are handed back zeroed for the next run (this is their last reader)
        const uint4 z = make_uint4(0u, 0u, 0u, 0u);
        const int ngrp = (fr.nseg + LPF_GROUP_SEGS - 1) / LPF_GROUP_SEGS;
        for (int i = tid; i < ngroups * fr.nseg; i += LPF_BLOCK) P.seg_tab[(size_t)(i / fr.nseg) * P.nseg_cap + fr.seg_off + (i % fr.nseg)] = z;
        for (int i = tid; i < ngroups * ngrp; i += LPF_BLOCK) P.grp_tab[(size_t)(i / ngrp) * P.ngrp_cap + fr.grp_off + (i % ngrp)] = z;
        for (int i = tid; i < LPF_FRM_SHARDS * LPF_TAB_GROUPS; i += LPF_BLOCK) P.frm_tab[(size_t)f * LPF_FRM_SHARDS * LPF_TAB_GROUPS + i] = z;
    }
    // first strict maximum over the boxes, starting from 0: one wave per mask, lanes over boxes
    auto maxima = [&](int m0, int m1) {
        for (int m = m0 + wave; m < m1; m += 4) {
            unsigned best = 0;
            int best_idx = 0x7fffffff;
            for (int b = lane; b < B; b += 64) {
                unsigned c = s_c[((m - m0) * B + b) & (LPF_FIN_STAGE - 1)];       // (LDS; memory for B > LPF_FIN_STAGE, kept apart:
                asm volatile("" : "+v"(c));                                     //  see lpf_count_chunk)
                if (!staged) c = cnt[m * B + b];
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
    };
    if (!staged) maxima(0, M);
    else {
        maxima(0, min(M, gm));
        for (int m0 = gm; m0 < M; m0 += gm) {               // (block-uniform; rarely entered)
            stage_load(m0);
            __syncthreads();                                // the previous pass's maxima have read s_c
            stage_store(m0);
            __syncthreads();
            maxima(m0, min(M, m0 + gm));
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
    if (!staged) {                                          // many masks x many boxes: from memory, after every wave has read them
        __syncthreads();
        for (int i = tid; i < MB; i += LPF_BLOCK) {
            if (out) out[i] = (int32_t)cnt[i];
            cnt[i] = 0;
        }
    }
}

// ------------------------------------------------------------------------------------
// TAIL: lists and box counts in ONE launch.  Block table entry = up to four consecutive segments of one frame (a wave
// each); blocks [0, nblk) build the lists of their segments, blocks [nblk, 2 nblk) -- present when boxes are to be
// counted -- count their segments' masked points into the boxes.  Neither half needs anything from the other, and
// nothing waits inside the kernel.  lpf_finalize then writes the per-frame summaries.
// ------------------------------------------------------------------------------------
// Tail block tables.  List blocks: {first segment, frame << 3 | segments (0..4)} of block i.  Box-count blocks: {first segment,
// frame, candidate word, part << 3 | segments} -- a group of four segments gets a count block per 64-box word of its frame (and,
// P.csplit > 1, per part: the parts share the group's chunks).
// A launch of ONE frame needs no table (and so no upload when its size changes from run to run): list block i takes segments
// 4 i .. 4 i + 3, count block i is (group, word, part) = (i / (words csplit), i / csplit % words, i % csplit).
__device__ __forceinline__ int2 lpf_tail_entry(const LpfParams &P, const int i)
{
    if (P.F > 1) return P.blks[i];
    const int first = i * LPF_LISTS_WAVES;
    return make_int2(first, max(0, min(LPF_LISTS_WAVES, P.frame0.nseg - first)));
}
__device__ __forceinline__ void lpf_count_entry(const LpfParams &P, const int i, int &first, int &f, int &nw, int &wd, int &part)
{
    if (P.F > 1) {
        const int4 e = P.cblks[i];
        first = e.x; f = e.y; wd = e.z; nw = e.w & 7; part = e.w >> 3;
        return;
    }
    const int words = max(1, P.frame0.cand_words), per = words * P.csplit;
    const int g = i / per, rem = i - g * per;
    wd = rem / P.csplit; part = rem - wd * P.csplit; f = 0; first = g * LPF_LISTS_WAVES;
    nw = max(0, min(LPF_LISTS_WAVES, P.frame0.nseg - first));
}

struct LpfTailListsLds { unsigned short lidx[LPF_LISTS_WAVES][LPF_LIST_CAP]; };            // masked entries of a pass
struct LpfTailCountLds {
    float4 pt[LPF_LISTS_WAVES][64];           // xyz + label of a wave's current chunk
    unsigned q[LPF_LISTS_WAVES][128];         // (point, box) pairs that passed the float bounds
    unsigned cnt[LPF_MAX_MASKS_DEV * LPF_BC_WORD / 2];     // the block's inside counts [M][64], 16 bits each
    float bq[6 * LPF_BC_WORD];                // {lo xyz, hi xyz} float bounds of the word's boxes
    double bp[LPF_BC_WORD * 16];              // their exact parameters
    float dom[4];                             // the word's ground grid: {x0, y0, 1 / cell_x, 1 / cell_y}
    unsigned im[LPF_LISTS_WAVES][64], mbase[LPF_LISTS_WAVES][64], L[LPF_LISTS_WAVES];      // row prefixes of the block's four segments
};

#define LPF_TAIL_LDS (sizeof(LpfTailCountLds) > sizeof(LpfTailListsLds) ? sizeof(LpfTailCountLds) : sizeof(LpfTailListsLds))
#define LPF_STEP_LDS (LPF_TAIL_LDS > sizeof(LpfBoxJobLds) ? LPF_TAIL_LDS : sizeof(LpfBoxJobLds))

// one tail block: the first P.ncblk count boxes (when there are any: the longer chain goes first), the next P.nblk build lists
template <bool PRE, int STEP>
__device__ __forceinline__ void lpf_tail_block(const LpfParams &P, const int tb, char *s_raw)
{
    LpfTailListsLds &LL = *reinterpret_cast<LpfTailListsLds *>(s_raw);
    LpfTailCountLds &LC = *reinterpret_cast<LpfTailCountLds *>(s_raw);
    const int tid = threadIdx.x, wave = lpf_wave();
    const int ncount = P.count_boxes ? P.ncblk : 0;         // (the list blocks follow; none when no list is wanted)
    if (tb >= ncount) {
        const int2 ent = lpf_tail_entry(P, tb - ncount);    // {first segment, frame << 3 | segments}
        const int f = ent.y >> 3, nw = ent.y & 7;
        const LpfFrame fr = lpf_frame_record(P.frame0, P.frames, P.F > 1, f);
        if (wave < nw && (P.valid_idx || P.inst_idx)) {
            // (1024-point segments -- small launches: the form that walks the 16 rows with lane = point; a dense row of a real scan
            //  costs the other form 64 serial steps)
            if (P.lists_small) lpf_lists_wave_small<PRE>(P, fr, ent.x + wave);
            else lpf_lists_wave<PRE, STEP>(P, fr, ent.x + wave, LL.lidx[wave]);
        }
        return;
    }
    int first, f, nw, wd, part;
    lpf_count_entry(P, tb, first, f, nw, wd, part);
    const LpfFrame fr = lpf_frame_record(P.frame0, P.frames, P.F > 1, f);
    if (wd * LPF_BC_WORD >= fr.B) return;                   // a frame without boxes in a batch that has some: no grid to look into (block-uniform)
    if (!P.count_lazy) lpf_count_stage(P, fr, wd, tid, LPF_BLOCK, LC.cnt, LC.bq, LC.bp, LC.dom);
    {                                                       // the block's (up to) four segments: row prefixes -> LDS
        const int lane = lpf_lane();
        unsigned im = 0, mbase = 0, L = 0;
        if (wave < nw) L = lpf_count_rows(P, fr, first + wave, im, mbase);
        LC.im[wave][lane] = im; LC.mbase[wave][lane] = mbase;
        if (lane == 0) LC.L[wave] = L;
    }
    __syncthreads();
    if (P.count_lazy) {                                     // (block-uniform)
        if ((LC.L[0] | LC.L[1] | LC.L[2] | LC.L[3]) == 0u) return;       // nothing to count: the boxes are never fetched
        lpf_count_stage(P, fr, wd, tid, LPF_BLOCK, LC.cnt, LC.bq, LC.bp, LC.dom);
        __syncthreads();
    }
    // The chunks of 64 masked points of the four segments are shared: chunk c goes to wave (c mod 4 csplit) of the group's csplit
    // blocks.  Consecutive segments of a real scan lie on the same car -- all four heavy or all four empty -- so sharing within one
    // block gains nothing, sharing over four does (small pipelined launches: frame 100 in a stream 16 -> 12 us per frame).
    {
        const int lane = lpf_lane();
        const unsigned L0 = LC.L[0], L1 = LC.L[1], L2 = LC.L[2], L3 = LC.L[3];
        const int c0 = (int)((L0 + 63) >> 6), c1 = (int)((L1 + 63) >> 6), c2 = (int)((L2 + 63) >> 6), c3 = (int)((L3 + 63) >> 6);
        for (int c = part * LPF_LISTS_WAVES + wave; c < c0 + c1 + c2 + c3; c += LPF_LISTS_WAVES * P.csplit) {     // wave-uniform
            int sg = 0, cc = c;
            if (cc >= c0) { cc -= c0; sg = 1; if (cc >= c1) { cc -= c1; sg = 2; if (cc >= c2) { cc -= c2; sg = 3; } } }
            lpf_count_chunk(P, fr, first + sg, LC.im[sg][lane], LC.mbase[sg][lane], LC.L[sg], (unsigned)cc * 64u, LC.pt[wave], LC.q[wave],
                            LC.cnt, LC.bq, LC.bp, LC.dom, wd);
        }
    }
    __syncthreads();
    lpf_count_flush(P, fr, wd, tid, LPF_BLOCK, LC.cnt);
}

// 7 blocks per CU (20 KB LDS): 60 VGPRs since the frame record lives in scalar registers (lpf_frame_record); before that it needed
// 72, and forced to 64 it spilled six registers per thread to scratch -- HBM traffic too.
template <bool PRE>
__global__ __launch_bounds__(LPF_BLOCK, 7) void lpf_tail_t(const LpfParams P)
{
    __shared__ __attribute__((aligned(16))) char s_raw[LPF_TAIL_LDS];
    lpf_tail_block<PRE, 1>(P, (int)blockIdx.x, s_raw);
}

// ------------------------------------------------------------------------------------
// TAIL, wide form (small launches: a frame or a few): the same two roles with 16 waves per block.  A real scan is dense
// in places -- consecutive points are neighbours in space, so a segment lying on a car holds hundreds of masked points
// while its neighbours hold none -- and with a frame's worth of segments the launch is as long as its heaviest wave.
// Here the box-count block's 16 waves SHARE the chunks (64 masked points) of the block's four segments: waves 0..3
// publish their segment's row prefixes in LDS, then chunk c goes to wave c mod 16.  The list blocks use four of the
// waves (a segment each), the others leave at once.  Same results as lpf_tail_t.
// ------------------------------------------------------------------------------------
#define LPF_WIDE_WAVES 16

struct LpfTailWideLds {
    float4 pt[LPF_WIDE_WAVES][64];
    unsigned q[LPF_WIDE_WAVES][128];
    unsigned cnt[LPF_MAX_MASKS_DEV * LPF_BC_WORD / 2];
    float bq[6 * LPF_BC_WORD];
    double bp[LPF_BC_WORD * 16];
    float dom[4];
    unsigned im[LPF_LISTS_WAVES][64], mbase[LPF_LISTS_WAVES][64], L[LPF_LISTS_WAVES];
};

template <bool PRE>
__global__ __launch_bounds__(64 * LPF_WIDE_WAVES) void lpf_tail_wide_t(const LpfParams P)
{
    __shared__ __attribute__((aligned(16))) char s_raw[sizeof(LpfTailWideLds)];
    LpfTailWideLds &LC = *reinterpret_cast<LpfTailWideLds *>(s_raw);
    const int tid = threadIdx.x, lane = lpf_lane(), wave = tid >> 6, tb = (int)blockIdx.x;
    const int ncount = P.count_boxes ? P.ncblk : 0;
    if (tb >= ncount) {
        const int2 ent = lpf_tail_entry(P, tb - ncount);
        const int f = ent.y >> 3, nw = ent.y & 7;
        const LpfFrame fr = lpf_frame_record(P.frame0, P.frames, P.F > 1, f);
        if (wave < nw && (P.valid_idx || P.inst_idx)) lpf_lists_wave_small<PRE>(P, fr, ent.x + wave);   // (this form: small launches only)
        return;
    }
    int first, f, nw, wd, part;
    lpf_count_entry(P, tb, first, f, nw, wd, part);           // (part is always 0 here: the wide form shares over its own 16 waves)
    const LpfFrame fr = lpf_frame_record(P.frame0, P.frames, P.F > 1, f);
    if (wd * LPF_BC_WORD >= fr.B) return;                   // (as in lpf_tail_block)
    if (wave < LPF_LISTS_WAVES) {                           // the block's (up to) four segments: row prefixes -> LDS
        unsigned im = 0, mbase = 0, L = 0;
        if (wave < nw) L = lpf_count_rows(P, fr, first + wave, im, mbase);
        LC.im[wave][lane] = im; LC.mbase[wave][lane] = mbase;
        if (lane == 0) LC.L[wave] = L;
    }
    lpf_count_stage(P, fr, wd, tid, 64 * LPF_WIDE_WAVES, LC.cnt, LC.bq, LC.bp, LC.dom);
    __syncthreads();
    const unsigned L0 = LC.L[0], L1 = LC.L[1], L2 = LC.L[2], L3 = LC.L[3];
    const int c0 = (int)((L0 + 63) >> 6), c1 = (int)((L1 + 63) >> 6), c2 = (int)((L2 + 63) >> 6), c3 = (int)((L3 + 63) >> 6);
    for (int c = wave; c < c0 + c1 + c2 + c3; c += LPF_WIDE_WAVES) {      // wave-uniform: chunk c of the block -> (segment, chunk of it)
        int sg = 0, cc = c;
        if (cc >= c0) { cc -= c0; sg = 1; if (cc >= c1) { cc -= c1; sg = 2; if (cc >= c2) { cc -= c2; sg = 3; } } }
        lpf_count_chunk(P, fr, first + sg, LC.im[sg][lane], LC.mbase[sg][lane], LC.L[sg], (unsigned)cc * 64u, LC.pt[wave], LC.q[wave],
                        LC.cnt, LC.bq, LC.bp, LC.dom, wd);
    }
    __syncthreads();
    lpf_count_flush(P, fr, wd, tid, 64 * LPF_WIDE_WAVES, LC.cnt);
}

__global__ __launch_bounds__(LPF_BLOCK) void lpf_finalize(const LpfParams P)
{
    __shared__ unsigned s_tot[LPF_TAB_ROWS], s_c[LPF_FIN_STAGE];
    const int f = blockIdx.x;
    const LpfFrame fr = lpf_frame_record(P.frame0, P.frames, P.F > 1, f);
    lpf_finalize_frame(P, fr, f, s_tot, s_c);
}

// ------------------------------------------------------------------------------------
// STEP (software-pipelined modes): ONE launch carries the streaming kernel of run i, the tail of run i-1, the
// summaries of run i-2 and the box job of run i (its tables are read by run i's tail, a launch later) -- and, in mode 4,
// the mask pack of run i+1, whose own streaming kernel the next launch carries.  One scratch set and one box set per run
// in flight, so nothing in a launch depends on anything else in it; the launch boundaries order the runs' phases.  The
// tail's ~2000 short, latency-bound blocks are dealt out among the K1 tiles, eight (one per XCD) after every `kper` tiles,
// so they trickle through the chip beside the streaming work instead of standing in front of it or behind it; the pack's
// ~1000 blocks are pure streaming work and come last, where they run while the final tiles drain (dealt among the tiles
// they cost what they take):
//     blocks [0, nfin8)                               summaries of run i-2 (nfin8 = frames, padded to a multiple of 8)
//     then nbox8 box-job blocks                       (frames x 64-box words of the run being queued, padded)
//     then nrg8 rectangle-grid blocks                 (mode 4, masks read inside their rectangles: the candidate grid of the run being queued)
//     then nper periods of (kper K1 tiles, 8 tail blocks)
//     then the remaining K1 tiles
//     then the pack blocks
// In order (no pipelining) the same kernel carries a run's tiles and its box job, nothing else.
// A K1 block's XCD is blockIdx & 7 throughout (every offset is a multiple of 8), which lpf_k1_tile's tile
// mapping relies on (speed only).
// ------------------------------------------------------------------------------------
struct LpfStepLayout {
    int nfin, nfin8;             // summary blocks (frames of run i-2), padded
    int nbox, nbox8;             // box-job blocks (frames x 64-box chunks of the run whose boxes were set since the last launch), padded
    int ntail;                   // tail blocks of run i-1
    int kper, nper;              // K1 tiles per period (multiple of 8), periods that carry tail blocks
    int nk1;                     // K1 tiles of run i
    int nrg, nrg8;               // rectangle-grid blocks (mode 4, LpfDirectRect: of the run whose K1 tiles the NEXT launch carries), padded
    int npack;                   // mask-pack blocks (mode 4: of the run whose K1 tiles the NEXT launch carries), after all K1 tiles
    int rest;                    // K1 block slots after the periods
#ifdef LPF_LAB
    unsigned long long *clk;     // role clock (lpf_lab_role_clock), or null: per role {first start, last end, sum, blocks, longest block}
#endif
};
// Lab builds: how long the blocks of each role of a step launch run, in ticks of the 100 MHz wall clock (tools/role_clock.py).
// Roles: 0 summaries, 1 box job, 2 lists, 3 box counts, 4 mask pack, 5 project+label tiles.
#ifdef LPF_LAB
#define LPF_ROLE_BEGIN const unsigned long long t0_ = wall_clock64();
#define LPF_ROLE_END(role)                                                                                              \
    if (Y.clk) {                                                                                                        \
        __syncthreads();                                                                                                \
        if (threadIdx.x == 0) {                                                                                         \
            const unsigned long long t1_ = wall_clock64();                                                              \
            unsigned long long *k_ = Y.clk + 5 * (role);                                                                \
            atomicMin(k_, t0_); atomicMax(k_ + 1, t1_); atomicAdd(k_ + 2, t1_ - t0_); atomicAdd(k_ + 3, 1ull);          \
            atomicMax(k_ + 4, t1_ - t0_);                                                                               \
        }                                                                                                               \
    }
#else
#define LPF_ROLE_BEGIN
#define LPF_ROLE_END(role)
#endif
struct LpfBoxFrame {             // per frame, host-built from the box counts
    int box_off, B;
    long long cand_off;          // first word of the frame's grid
};
struct LpfBoxJob {               // one box preparation / table set-up (lpf_box_frame_block): a block per frame
    const double *src;           // [Btot][8][3] corners, cam-0 frame (cam0 = 1) or velodyne frame
    const uint8_t *enabled_in;   // velodyne-frame input: [Btot] 0 = dropped earlier by filter_visible_bboxes, or null
    int cam0, filter_visible, oriented, F;
    double Tcv[12];              // rows 0..2 of inv(TrVeloToCam) (cam0 = 1)
    double K[9];                 // camera.K[:3,:3] (the visibility filter of cam-0 boxes)
    int W, H, chunks;            // chunks: blocks per frame = 64-bit words of the frame with the most boxes
    const LpfBoxFrame *bframes;  // [F] (F > 1)
    LpfBoxFrame frame0;          // ... by value for one frame
    double *boxp; float *boxq; unsigned long long *cand;
    double *corners_keep;        // [Btot][8][3] the context's copy of the velodyne-frame corners (a camera change rebuilds from it), or null
    uint8_t *enabled_out;        // [Btot] the context's copy of `visible` (cam0 = 1), or null
    uint8_t *visible; double *corners_out; double *bbox2d; int32_t *front;      // optional outputs of lpf_set_boxes_cam0 (device memory)
};
struct LpfBoxJobLds {
    double c[32][8][3];           // velodyne-frame corners of the pass's 32 boxes, then its intermediate results (8 slots per box)
    unsigned long long grid[LPF_GRID_CELLS];      // the word's ground grid while it is built
    float bnd[64][4];             // {lo x, lo y, hi x, hi y} of the float bounds of the word's 64 boxes (empty: lo > hi)
    float dom[4];                 // the grid's domain
};

__device__ __forceinline__ void lpf_box_frame_block(const LpfBoxJob &J, const int blk, char *s_raw);
struct LpfPackJob {              // uint8 masks [F][M][H][W] -> label image [F][H][W] of the step's LT (lpf_pack16_block)
    const uint8_t *masks;
    void *label;
    long long hw, total16;
    int M, W;
    const int4 *rects;           // [F][M] {x0, y0, x1, y1} (half open): the caller's word that mask m of frame f is zero outside, or null
};
template <typename T, int MODE, typename LT>
__device__ __forceinline__ void lpf_pack16_block(const T *__restrict__ masks, LT *__restrict__ label,
                                                 const int M, const long long hw, const long long total16, const long long blk,
                                                 const int4 *__restrict__ rects, const int W);

// Candidate grid of the masks' rectangles (LpfDirectRect tiles): cell (cy, cx) of frame f = the masks whose rectangle meets the
// LPF_RG_CELL x LPF_RG_CELL pixels of the cell -- a superset of the masks a pixel of the cell can lie in.  A thread per cell, the
// frame's rectangles from wave-uniform loads; a few KB per frame (1408 x 376: 2112 cells), built once per run: by blocks of the run's
// own launch where its tiles come a launch later (lpf_set_pipelined 4), else by a small kernel ahead of them.
struct LpfRectJob {
    const int4 *rects;           // [F][M]
    uint32_t *grid;              // [F][cells]
    int F, M, cw, ch, cells, bpf;    // bpf: blocks per frame = ceil(cells / LPF_BLOCK)
};
__device__ __forceinline__ void lpf_rect_grid_block(const LpfRectJob &G, const int blk)
{
    const int f = blk / G.bpf, i = (blk - f * G.bpf) * LPF_BLOCK + (int)threadIdx.x;
    if (f >= G.F || i >= G.cells) return;
    const int cy = i / G.cw, cx = i - cy * G.cw;
    const int x0 = cx << LPF_RG_SHIFT, y0 = cy << LPF_RG_SHIFT;
    const int4 *__restrict__ rc = G.rects + (size_t)lpf_uni(f) * G.M;
    uint32_t bits = 0u;
    for (int m0 = 0; m0 < G.M; m0 += 8) {                    // eight rectangles' loads in flight (one at a time: a 6 us kernel for nine blocks)
        int4 q[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) q[j] = rc[min(m0 + j, G.M - 1)];
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (m0 + j < G.M && q[j].x < x0 + LPF_RG_CELL && q[j].z > x0 && q[j].y < y0 + LPF_RG_CELL && q[j].w > y0 && q[j].z > q[j].x && q[j].w > q[j].y)
                bits |= 1u << (m0 + j);
    }
    G.grid[(size_t)f * G.cells + i] = bits;
}
__global__ __launch_bounds__(LPF_BLOCK) void lpf_rect_grid_kernel(const LpfRectJob G) { lpf_rect_grid_block(G, (int)blockIdx.x); }

template <int ROWS, unsigned FL, typename LT, bool PRE, bool BOXES>
__global__ __launch_bounds__(LPF_BLOCK, 7) void lpf_step_t(const LpfParams P, const LpfParams Q, const LpfParams R, const LpfStepLayout Y,
                                                           const LpfPackJob J, const LpfBoxJob X, const LpfRectJob G)
{
    __shared__ __attribute__((aligned(16))) char s_raw[LPF_STEP_LDS];
    __shared__ unsigned s_cnt[LPF_TAB_ROWS];
    int b = (int)blockIdx.x;
    LPF_ROLE_BEGIN
    if (b < Y.nfin8) {                                      // ---- summaries of run i-2
        if (b < Y.nfin) {
            const LpfFrame fr = lpf_frame_record(R.frame0, R.frames, R.F > 1, b);
            lpf_finalize_frame(R, fr, b, s_cnt, reinterpret_cast<unsigned *>(s_raw));      // (16.7 KB of role LDS: room for the 4 KB stage)
            LPF_ROLE_END(0)
        }
        return;
    }
    b -= Y.nfin8;
    if (b < Y.nbox8) {                                      // ---- box tables of the run that follows (a block per frame): they are read
        if constexpr (BOXES) {                              //      by its tail, one (mode 2) or two (mode 4) launches from here
            if (b < Y.nbox) { lpf_box_frame_block(X, b, s_raw); LPF_ROLE_END(1) }
        }
        return;
    }
    b -= Y.nbox8;
    if (b < Y.nrg8) {                                       // ---- candidate grid of the masks' rectangles of the run whose tiles come next launch
        if (b < Y.nrg) lpf_rect_grid_block(G, b);
        return;
    }
    b -= Y.nrg8;
    const int plen = Y.kper + 8, periodic = Y.nper * plen;
    int vblk;                                               // K1: virtual block index, (rank on the XCD) << 3 | XCD
    if (b < periodic) {
        const int per = b / plen, pos = b - per * plen;
        if (pos >= Y.kper) {                                // ---- tail of run i-1, then the mask pack (mode 4)
            const int tb = per * 8 + (pos - Y.kper);
            if (tb < Y.ntail) { lpf_tail_block<PRE, 2>(Q, tb, s_raw); LPF_ROLE_END(tb < (Q.count_boxes ? Q.ncblk : 0) ? 3 : 2) }
            return;
        }
        vblk = ((per * (Y.kper >> 3) + (pos >> 3)) << 3) | (pos & 7);
    } else {
        const int r = b - periodic;
        if (r >= Y.rest) {                                  // ---- the mask pack (mode 4): behind the tiles, it fills their ramp-down
            if constexpr (!LpfIsDirect<LT>::value) {        // (tiles that read the masks directly never share a launch with a pack)
                if (r - Y.rest < Y.npack) { lpf_pack16_block<uint8_t, 0, LT>(J.masks, static_cast<LT *>(J.label), J.M, J.hw, J.total16, r - Y.rest, J.rects, J.W); LPF_ROLE_END(4) }
            }
            return;
        }
        vblk = ((Y.nper * (Y.kper >> 3) + (r >> 3)) << 3) | (r & 7);
    }
    // ---- K1 tile of run i: XCD x owns a contiguous run of count_x tiles (lpf_xcd_remap); the grid is padded per XCD
    const int x = vblk & 7, q = Y.nk1 >> 3, rem = Y.nk1 & 7;
    if ((vblk >> 3) >= q + (x < rem ? 1 : 0)) return;
    lpf_k1_tile<ROWS, FL, LT>(P, vblk, s_cnt);
    LPF_ROLE_END(5)
}

// ------------------------------------------------------------------------------------
// cv2.resize(mask_u8, (W, H)), INTER_LINEAR on 8-bit data (V3:222 for masks that do not arrive at camera size), as OpenCV 4.x's
// C++ reference path computes it (resize.cpp: HResizeLinear / VResizeLinear, 11-bit weights; restated and cited in
// oracle/numpy_path.py: cv2_resize_linear_u8 -- pinned by construction only, OpenCV is not in the image).  The weight tables
// {source index, second index, w0, w1} per destination column and row are made on the host as OpenCV makes them; a thread
// produces one destination pixel: two 32-bit horizontal sums, then the vertical fixed-point blend.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(LPF_BLOCK) void lpf_resize_linear_u8_kernel(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst,
                                                                         const int4 *__restrict__ xtab, const int4 *__restrict__ ytab,
                                                                         const int w, const int h, const int W, const int H, const long long total)
{
    const long long i = (long long)blockIdx.x * LPF_BLOCK + threadIdx.x;
    if (i >= total) return;
    const long long hw = (long long)W * H;
    const long long n = i / hw;
    const int rem = (int)(i - n * hw), y = rem / W, x = rem - y * W;
    const int4 cx = xtab[x], cy = ytab[y];
    const uint8_t *__restrict__ s = src + (size_t)n * w * h;
    const int r0 = cy.x * w, r1 = cy.y * w;
    const int S0 = (int)s[r0 + cx.x] * cx.z + (int)s[r0 + cx.y] * cx.w;
    const int S1 = (int)s[r1 + cx.x] * cx.z + (int)s[r1 + cx.y] * cx.w;
    dst[i] = (uint8_t)((((cy.z * (S0 >> 4)) >> 16) + ((cy.w * (S1 >> 4)) >> 16) + 2) >> 2);
}

// ------------------------------------------------------------------------------------
// Host callers whose result buffers are page-locked (lpf_host_alloc / hipHostMalloc: the GPU can write them): the filled parts of the
// compact results -- valid_idx, (u, v) and labels of the valid points, the instance lists, the counts, the summaries -- go to host
// memory by THIS kernel, which reads the lengths from the summaries on the device.  The copy engine needs the lengths on the host
// first: a host wait, then four copies per frame (each ~10 us of latency for a few hundred KB) and a second wait -- most of what a
// frame's host call cost once the kernels take 20 us.  LPF_R2H_BLOCKS blocks per frame, 8-byte stores (every array is 8-byte aligned
// at a frame's offset except the 4-byte labels, which go word by word).
// ------------------------------------------------------------------------------------
#define LPF_R2H_BLOCKS 8
struct LpfToHost {
    long long *valid_idx; int2 *uv_valid; uint32_t *label_valid; long long *inst_idx; int32_t *count_mb; void *summary;   // host (mapped), nullable
    long long inst_cap;
    int n_count;                 // M * Btot
};

__global__ __launch_bounds__(LPF_BLOCK) void lpf_results_to_host(const LpfParams P, const LpfToHost D)
{
    const int f = (int)blockIdx.x / LPF_R2H_BLOCKS, part = (int)blockIdx.x - f * LPF_R2H_BLOCKS;
    if (f >= P.F) return;
    const LpfFrame fr = lpf_frame_record(P.frame0, P.frames, P.F > 1, f);
    const long long *S = reinterpret_cast<const long long *>(static_cast<const char *>(P.summary) + (size_t)f * LPF_SUMMARY_BYTES);   // (layout: lpf_finalize_frame)
    const long long nv = S[0];
    long long tot = S[34 + LPF_MAX_MASKS_DEV];                // inst_off[M_max] = entries of all lists
    if (tot > D.inst_cap) tot = D.inst_cap;
    const long long t0 = (long long)part * LPF_BLOCK + threadIdx.x, step = (long long)LPF_R2H_BLOCKS * LPF_BLOCK;
    const long long a = fr.pt_off;
    if (D.valid_idx) for (long long i = t0; i < nv; i += step) D.valid_idx[a + i] = P.valid_idx[a + i];
    if (D.uv_valid) for (long long i = t0; i < nv; i += step) D.uv_valid[a + i] = P.uv_valid[a + i];
    if (D.label_valid) for (long long i = t0; i < nv; i += step) D.label_valid[a + i] = P.label_valid[a + i];
    if (D.inst_idx) {
        const long long b = (long long)f * D.inst_cap;
        for (long long i = t0; i < tot; i += step) D.inst_idx[b + i] = P.inst_idx[b + i];
    }
    if (part == 0) {
        long long *dst = reinterpret_cast<long long *>(static_cast<char *>(D.summary) + (size_t)f * LPF_SUMMARY_BYTES);
        for (int i = threadIdx.x; i < LPF_SUMMARY_BYTES / 8; i += LPF_BLOCK) dst[i] = S[i];
    }
    if (D.count_mb)                                            // one array for the whole batch: dealt over all blocks
        for (long long i = (long long)blockIdx.x * LPF_BLOCK + threadIdx.x; i < D.n_count; i += (long long)gridDim.x * LPF_BLOCK) D.count_mb[i] = P.count_out[i];
}

// The one case cv2.resize(..., INTER_LINEAR) does not compute linearly: a source of exactly twice the target in BOTH axes is handed to
// INTER_AREA (resize(): `if (interpolation == INTER_LINEAR && is_area_fast && iscale_x == 2 && iscale_y == 2) interpolation =
// INTER_AREA`), whose 8-bit single-channel 2 x 2 path (ResizeAreaFastVec, scalar and SIMD alike) is the rounded mean of the four
// source pixels: (a + b + c + d + 2) >> 2.  A thread produces four destination pixels of a row from two 8-byte runs (W % 4 tails
// pixel by pixel).
__global__ __launch_bounds__(LPF_BLOCK) void lpf_resize_area2_u8_kernel(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst,
                                                                        const int W, const int H, const long long quads)
{
    const long long i = (long long)blockIdx.x * LPF_BLOCK + threadIdx.x;
    if (i >= quads) return;
    const int qw = (W + 3) >> 2;                                  // quads per destination row
    const long long row = i / qw;                                 // plane * H + y
    const int x0 = (int)(i - row * qw) * 4, nx = min(4, W - x0);
    const uint8_t *__restrict__ s0 = src + (size_t)row * 4 * W + 2 * x0;    // source row 2y of the same plane: (plane * 2H + 2y) * 2W
    const uint8_t *__restrict__ s1 = s0 + 2 * W;
    uint8_t *__restrict__ d = dst + (size_t)row * W + x0;
    for (int k = 0; k < nx; ++k)
        d[k] = (uint8_t)(((int)s0[2 * k] + (int)s0[2 * k + 1] + (int)s1[2 * k] + (int)s1[2 * k + 1] + 2) >> 2);
}

// cv2.erode(plane_u8, getStructuringElement(MORPH_ELLIPSE, (3, 3))) on 8-bit VALUES (V3:83-90 on masks that are not at camera size,
// where the erosion comes before the resize, V3:222): the minimum over the plus-shaped neighbourhood, pixels outside the image left
// out (OpenCV's erode border is +infinity).  One iteration, n planes [h][w]; a thread per pixel.
__global__ __launch_bounds__(LPF_BLOCK) void lpf_erode_u8_kernel(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, const int w, const int h,
                                                                 const long long total)
{
    const long long i = (long long)blockIdx.x * LPF_BLOCK + threadIdx.x;
    if (i >= total) return;
    const long long hw = (long long)w * h;
    const long long rem = i % hw;
    const int y = (int)(rem / w), x = (int)(rem - (long long)y * w);
    unsigned v = src[i];
    if (x > 0) v = min(v, (unsigned)src[i - 1]);
    if (x + 1 < w) v = min(v, (unsigned)src[i + 1]);
    if (y > 0) v = min(v, (unsigned)src[i - w]);
    if (y + 1 < h) v = min(v, (unsigned)src[i + w]);
    dst[i] = (uint8_t)v;
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
struct LpfBoxPrep { double Tcv[16]; double K[9]; int W, H; };     // cam -> velo transform, camera.K[:3,:3], image size

__global__ __launch_bounds__(LPF_BLOCK) void lpf_box_prep_kernel(const LpfBoxPrep A, const double *__restrict__ corners_cam, int nbox,
                                                                 uint8_t *__restrict__ visible, double *__restrict__ corners_velo,
                                                                 double *__restrict__ bbox2d, int *__restrict__ front)
{
    const double *Tcv = A.Tcv, *K = A.K;
    const int W = A.W, H = A.H;
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
            if (visible) visible[b] = (uint8_t)(nvis >= 2);
            if (front) front[b] = nfront;
            if (bbox2d) { bbox2d[4 * b] = umin; bbox2d[4 * b + 1] = vmin; bbox2d[4 * b + 2] = umax; bbox2d[4 * b + 3] = vmax; }
        }
    }
}

// ------------------------------------------------------------------------------------
// BOX JOB (per frame, per box change): the reference's per-frame box preparation (V3:556-562) and the tables the counting
// kernels read, in ONE block per frame -- as a kernel of its own (serial mode, graph capture) or as a role of lpf_step_t
// (software-pipelined modes: the boxes of run i are prepared by blocks of run i's launch, two launches before its tail
// counts into them).  From the 8 corners of every box -- cam-0 frame (cam0 = 1: filter_visible_bboxes +
// transform_bboxes_to_velodyne first, exactly lpf_box_prep_kernel's arithmetic) or velodyne frame --
//   boxp[b]  the oracle's slab parameters { c0, (v_a, |v_a|^2) x 3, exact_ok }  (oracle/lpf_oracle.c: orc_oriented_inside;
//            V3:187-197) or { lo, hi } for the axis-aligned test (V3:158-162): the same operations in the same order as
//            the reference's NumPy statements, so the counting kernels decide exactly as it does;
//   boxq[b]  a conservative float AABB of the accepted region;
//   cand     per 64 boxes a ground grid: per cell the bit set of the boxes whose float bounds reach into it (above: lpf_ground_cell).
// boxq and cand only skip hopeless (point, box) pairs: every candidate still takes the exact test.
// Eight lanes share a box (lane = corner, then lane = slab / edge / coordinate), 32 boxes per pass of the block.  The grid is
// built in LDS (the block has the word to itself): zeroed there, every box ORs its bit into the cells of its rectangle with LDS
// atomics, and the finished cells leave as plain coalesced stores -- no memset launch in front, nothing to clean afterwards.
// A box whose `enabled` byte is 0 (filter_visible_bboxes dropped it) gets an empty AABB and no cell.
// ------------------------------------------------------------------------------------
__device__ __forceinline__ double lpf_min8(double x)   // over the 8 lanes that share a box (contiguous, 8-aligned)
{
    x = fmin(x, __shfl_xor(x, 1)); x = fmin(x, __shfl_xor(x, 2)); return fmin(x, __shfl_xor(x, 4));
}
__device__ __forceinline__ double lpf_max8(double x)
{
    x = fmax(x, __shfl_xor(x, 1)); x = fmax(x, __shfl_xor(x, 2)); return fmax(x, __shfl_xor(x, 4));
}

// 1 / x for the work-skipping structures only (the float bounds: margin 1e-5): the hardware
// reciprocal and one Newton step (~2^-50), a fifth of the IEEE division's chain
__device__ __forceinline__ double lpf_rcp_approx(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    return fma(fma(-x, r, 1.0), r, r);
}

__device__ __forceinline__ void lpf_box_frame_block(const LpfBoxJob &J, const int blk, char *s_raw)
{
    // The work of a box is cut into short phases that hand their results on through LDS (lane = corner -> lane = slab ->
    // lane = edge of the region -> lane = coordinate): inside lpf_step_t the role has 72 registers, and the straight-line form
    // (every lane everything: 111) spilled.  A block is one dependent chain with nothing beside it to hide latency, so the
    // chain is kept short: shared-reciprocal division where the reference's quotient is needed (lpf_div2, bit-equal to '/'),
    // approximate reciprocals where only the conservative structures are concerned, no square root -- and a block takes only 64
    // boxes of its frame, ONE ground grid: block blk = (frame blk / chunks, word blk % chunks), chunks = the
    // words of the frame with the most boxes (a frame with fewer leaves its surplus blocks at once).
    LpfBoxJobLds &L = *reinterpret_cast<LpfBoxJobLds *>(s_raw);
    const int tid = threadIdx.x;
    const int f = blk / J.chunks, wd = blk - f * J.chunks;
    LpfBoxFrame bf = J.frame0;
    if (J.F > 1) {
        const LpfBoxFrame t = J.bframes[lpf_uni(f)];
        bf.box_off = lpf_uni(t.box_off); bf.B = lpf_uni(t.B); bf.cand_off = lpf_uni64(t.cand_off);
    }
    const int words = (bf.B + 63) >> 6;
    if (wd >= words) return;
    const int b_lo = wd << 6, B = min(bf.B, b_lo + 64);     // this block's boxes: [b_lo, B)
    // ---- the grid starts empty, and so do the bounds of the slots past the word's last box ----------------------------------------
    unsigned long long *__restrict__ gf = J.cand + bf.cand_off + (size_t)wd * LPF_GRID_WORDS;
    {
        int t0_ = tid;
        asm volatile("" : "+v"(t0_));
        for (int i = t0_; i < LPF_GRID_CELLS; i += LPF_BLOCK) L.grid[i] = 0ull;
        L.bnd[t0_ >> 2][t0_ & 3] = (t0_ & 2) ? -INFINITY : INFINITY;
        __syncthreads();
    }
    for (int b0 = b_lo; b0 < B; b0 += 32) {
        // (everything that depends on the thread index is derived again in every pass, behind an empty asm: hoisted out of
        //  the loop, those loop invariants -- addresses, selectors, constants -- cost more registers than the work itself)
        int t_ = tid;
        asm volatile("" : "+v"(t_));
        const int grp = t_ >> 3, k = t_ & 7;
        const unsigned long long gmask = 0xFFull << (t_ & 56);
        double (*C)[3] = L.c[grp];                          // the box's 8 slots of 3 doubles: corners first, then reused (see below)
        const int b = b0 + grp;
        const bool live = b < B;
        const size_t gb = (size_t)bf.box_off + (size_t)(live ? b : 0);
        // ---- phase 1, lane = corner: to the velodyne frame (and, cam-0 input, the visibility of the box) -----------------
        bool on = true;
        {
            double x = 0.0, y = 0.0, z = 0.0;
            if (live) { const double *cc = J.src + (gb * 8 + k) * 3; x = cc[0]; y = cc[1]; z = cc[2]; }
            if (J.cam0) {
                // cam2image on the raw cam-0 corners, WITHOUT R_rect (reference quirk, V3:129-138)
                double qx = J.K[0] * x; qx = fma(J.K[1], y, qx); qx = fma(J.K[2], z, qx);
                double qy = J.K[3] * x; qy = fma(J.K[4], y, qy); qy = fma(J.K[5], z, qy);
                double d  = J.K[6] * x; d  = fma(J.K[7], y, d);  d  = fma(J.K[8], z, d);
                if (d == 0.0) d = -1e-6;
                double uf, vf;
                lpf_div2(qx, qy, fabs(d), uf, vf);          // both quotients as '/' gives them (see lpf_div2)
                const double ru = rint(uf), rv = rint(vf);
                const bool in_img = (ru >= 0.0) && (ru < (double)J.W) && (rv >= 0.0) && (rv < (double)J.H);
                const bool vis = live && (d > 0.1) && in_img;
                const bool frn = live && (d > 0.0);
                const int nvis = __popcll(__ballot(vis) & gmask), nfront = __popcll(__ballot(frn) & gmask);
                on = !J.filter_visible || nvis >= 2;
                if (J.bbox2d) {                             // V4:157-168: min / max of the integer (u, v) over the corners in front
                    const double umin = lpf_min8(frn ? ru : 1e300), umax = lpf_max8(frn ? ru : -1e300);
                    const double vmin = lpf_min8(frn ? rv : 1e300), vmax = lpf_max8(frn ? rv : -1e300);
                    if (live && k == 0) { J.bbox2d[4 * gb] = umin; J.bbox2d[4 * gb + 1] = vmin; J.bbox2d[4 * gb + 2] = umax; J.bbox2d[4 * gb + 3] = vmax; }
                }
                if (live && k == 0) {
                    if (J.visible) J.visible[gb] = (uint8_t)(nvis >= 2);
                    if (J.front) J.front[gb] = nfront;
                    if (J.enabled_out) J.enabled_out[gb] = (uint8_t)(nvis >= 2);
                }
                // transform_bboxes_to_velodyne (V3:41-52): rows of inv(TrVeloToCam) as k-ordered fma chains
                double a0 = J.Tcv[0] * x; a0 = fma(J.Tcv[1], y, a0); a0 = fma(J.Tcv[2],  z, a0); a0 = fma(J.Tcv[3],  1.0, a0);
                double a1 = J.Tcv[4] * x; a1 = fma(J.Tcv[5], y, a1); a1 = fma(J.Tcv[6],  z, a1); a1 = fma(J.Tcv[7],  1.0, a1);
                double a2 = J.Tcv[8] * x; a2 = fma(J.Tcv[9], y, a2); a2 = fma(J.Tcv[10], z, a2); a2 = fma(J.Tcv[11], 1.0, a2);
                x = a0; y = a1; z = a2;
            } else if (J.enabled_in) {
                on = !live || J.enabled_in[gb] != 0;
            }
            if (live) {
                if (J.corners_keep) { double *o = J.corners_keep + (gb * 8 + k) * 3; o[0] = x; o[1] = y; o[2] = z; }
                if (J.corners_out) { double *o = J.corners_out + (gb * 8 + k) * 3; o[0] = x; o[1] = y; o[2] = z; }
            }
            C[k][0] = x; C[k][1] = y; C[k][2] = z;
        }
        __builtin_amdgcn_wave_barrier();                    // the 8 lanes of a box are lanes of one wave: its LDS queue is in order
        bool bounded = true;
        if (J.oriented) {
            // ---- phase 2, lanes 1..3 = slab a: v_a = c_{1,3,4} - c0, |v_a|^2 -- the exact parameters (V3:187-197) -----------
            bool ok = true;
            if (k >= 1 && k <= 3) {
                const int a = k - 1, oc = (a == 0) ? 1 : (a == 1) ? 3 : 4;
                const double v0 = C[oc][0] - C[0][0], v1 = C[oc][1] - C[0][1], v2 = C[oc][2] - C[0][2];
                double w = v0 * v0; w = fma(v1, v1, w); w = fma(v2, v2, w);
                ok = (w >= 1e-100 && w <= 1e100);           // also false for NaN (see lpf_oriented_inside)
                if (live) {
                    double *o = J.boxp + gb * 16 + 3 + 4 * a;
                    o[0] = on ? v0 : 0.0; o[1] = on ? v1 : 0.0; o[2] = on ? v2 : 0.0; o[3] = on ? w : 0.0;
                }
                C[5 + a][0] = v0; C[5 + a][1] = v1; C[5 + a][2] = v2;           // corners 5..7 are not needed any more
                asm volatile("" ::: "memory");
                C[4][a] = w;                                // (slab 2 has read corner 4: in-order LDS queue of the wave)
            }
            ok = __popcll(__ballot(!ok) & gmask) == 0;
            if (live && k == 0) {
                double *o = J.boxp + gb * 16;
                o[0] = on ? C[0][0] : 0.0; o[1] = on ? C[0][1] : 0.0; o[2] = on ? C[0][2] : 0.0; o[15] = (on && ok) ? 1.0 : 0.0;
            }
            __builtin_amdgcn_wave_barrier();
            // ---- phase 3, lanes 0..2 = edge a of the region { p : 0 <= (p - c0) . v_a <= |v_a|^2 }: with w_a the reciprocal
            //      basis (w_a . v_b = delta_ab), e_a = |v_a|^2 w_a, w_a = (v_b x v_c) / det, (a, b, c) cyclic -------------------
            bool good = true;
            if (k < 3) {
                const int a = k, bb = (a + 1) % 3, cc = (a + 2) % 3;
                const double bx = C[5 + bb][0], by = C[5 + bb][1], bz = C[5 + bb][2];
                const double cx = C[5 + cc][0], cy = C[5 + cc][1], cz = C[5 + cc][2];
                const double nx = by * cz - bz * cy, ny = bz * cx - bx * cz, nz = bx * cy - by * cx;
                const double det = C[5 + a][0] * nx + C[5 + a][1] * ny + C[5 + a][2] * nz;
                const double r = C[4][a] * lpf_rcp_approx(det);
                const double ex = nx * r, ey = ny * r, ez = nz * r;
                // well-conditioned: |det| > 1e-6 |v0| |v1| |v2|, compared as squares (no square root)
                good = (det * det > 1e-12 * (C[4][0] * C[4][1] * C[4][2])) && (ex == ex) && (ey == ey) && (ez == ez);
                C[1 + a][0] = ex; C[1 + a][1] = ey; C[1 + a][2] = ez;           // corners 1..3 have been read
            }
            if (!ok || __popcll(__ballot(!good) & gmask) != 0) bounded = false;
            __builtin_amdgcn_wave_barrier();
        } else {
            // point_in_bbox (V3:158-162) as the oracle restates it: lo = c0; if (w < lo) lo = w over corners 1..7 -- a NaN in
            // corner 0 stays (every comparison with it is false), a NaN elsewhere is skipped: fmin / fmax skip NaNs everywhere.
            // Lane k < 3 takes coordinate k: the region is the box c0' = lo, edges (hi - lo) along the axes.
            const double c0k = C[0][k % 3];
            const double mine = C[k][0], miney = C[k][1], minez = C[k][2];
            const double l0 = lpf_min8(mine), h0 = lpf_max8(mine), l1 = lpf_min8(miney), h1 = lpf_max8(miney), l2 = lpf_min8(minez), h2 = lpf_max8(minez);
            double lo = (k % 3 == 0) ? l0 : (k % 3 == 1) ? l1 : l2, hi = (k % 3 == 0) ? h0 : (k % 3 == 1) ? h1 : h2;
            if (!(c0k == c0k)) { lo = c0k; hi = c0k; }
            if (live && k < 3) { J.boxp[gb * 16 + k] = on ? lo : 0.0; J.boxp[gb * 16 + 3 + k] = on ? hi : 0.0; }
            if (live) { J.boxp[gb * 16 + 6 + k] = 0.0; if (k < 2) J.boxp[gb * 16 + 14 + k] = 0.0; }
            __builtin_amdgcn_wave_barrier();                // every lane has read its corner
            if (k < 3) {
                C[0][k] = lo;                               // origin of the region
                C[1 + k][0] = (k == 0) ? hi - lo : 0.0; C[1 + k][1] = (k == 1) ? hi - lo : 0.0; C[1 + k][2] = (k == 2) ? hi - lo : 0.0;
            }
            if (__popcll(__ballot(k < 3 && (!isfinite(lo) || !isfinite(hi))) & gmask) != 0) bounded = false;
            __builtin_amdgcn_wave_barrier();
        }
        // ---- phase 4, lanes 0..2 = coordinate: conservative float bounds of the region c0 + s0 e0 + s1 e1 + s2 e2, s in [0,1]^3.
        //      The margin (1e-5 relative + 1e-6) is far above the float rounding of the conversion, so no next-float step. ------
        {
            bool fin = true;
            if (k < 3) {
                const double c0k = C[0][k], e0 = C[1][k], e1 = C[2][k], e2 = C[3][k];
                const double lov = c0k + fmin(e0, 0.0) + fmin(e1, 0.0) + fmin(e2, 0.0), hiv = c0k + fmax(e0, 0.0) + fmax(e1, 0.0) + fmax(e2, 0.0);
                fin = isfinite(lov) && isfinite(hiv);
                const double m = 1e-5 * (fabs(lov) + fabs(hiv) + (hiv - lov)) + 1e-6;
                C[4][k] = lov - m; C[5][k] = hiv + m;
            }
            if (__popcll(__ballot(!fin) & gmask) != 0) bounded = false;
            __builtin_amdgcn_wave_barrier();
            if (live) {                                     // {lo xyz, 0, hi xyz, 0}: lane k writes float k
                const int kk = k & 3;
                float q = 0.f;
                if (kk < 3) {
                    if (!on) q = (k < 4) ? INFINITY : -INFINITY;                      // empty: nothing is near
                    else if (!bounded) q = (k < 4) ? -INFINITY : INFINITY;
                    else q = (float)C[k < 4 ? 4 : 5][kk];
                }
                J.boxq[gb * 8 + k] = q;
                if (kk < 2) L.bnd[b - b_lo][(k >> 2) * 2 + kk] = q;
            }
        }
        __builtin_amdgcn_wave_barrier();                    // the slots are rewritten by the next pass
    }
    __syncthreads();
    // ---- the domain: the (x, y) extent of the word's boxes that are on and bounded (one wave, lane = box) ---------------------------
    if (tid < 64) {
        const float lx = L.bnd[tid][0], ly = L.bnd[tid][1], hx = L.bnd[tid][2], hy = L.bnd[tid][3];
        const bool in = lx <= hx && ly <= hy && isfinite(lx) && isfinite(ly) && isfinite(hx) && isfinite(hy);
        float x0 = in ? lx : INFINITY, y0 = in ? ly : INFINITY, x1 = in ? hx : -INFINITY, y1 = in ? hy : -INFINITY;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            x0 = fminf(x0, __shfl_xor(x0, o)); y0 = fminf(y0, __shfl_xor(y0, o));
            x1 = fmaxf(x1, __shfl_xor(x1, o)); y1 = fmaxf(y1, __shfl_xor(y1, o));
        }
        if (tid == 0) {
            const float ex = x1 - x0, ey = y1 - y0;         // (no such box: -inf)
            // cells of extent / LPF_GRID, but not under a metre: a tight group of parked cars would otherwise put every box into
            // hundreds of cells (KITTI-360 annotates a moving car once per timestamp: dozens of boxes on the same few metres)
            float ix = (ex > (float)LPF_GRID) ? (float)LPF_GRID / ex : 1.f, iy = (ey > (float)LPF_GRID) ? (float)LPF_GRID / ey : 1.f;
            if (!(ix > 0.f)) ix = 0.f;                      // (an extent beyond the floats: one cell)
            if (!(iy > 0.f)) iy = 0.f;
            L.dom[0] = isfinite(x0) ? x0 : 0.f; L.dom[1] = isfinite(y0) ? y0 : 0.f; L.dom[2] = ix; L.dom[3] = iy;
        }
    }
    __syncthreads();
    // ---- every box into the cells of its rectangle: four lanes per box (off: lo > hi, no cell; unbounded: every cell) ---------------
    {
        int t_ = tid;
        asm volatile("" : "+v"(t_));
        const int j = t_ >> 2, k = t_ & 3;
        const float lx = L.bnd[j][0], ly = L.bnd[j][1], hx = L.bnd[j][2], hy = L.bnd[j][3];
        if (lx <= hx && ly <= hy) {
            const int x0 = lpf_ground_cell(lx, L.dom[0], L.dom[2]), x1 = lpf_ground_cell(hx, L.dom[0], L.dom[2]);
            const int y0 = lpf_ground_cell(ly, L.dom[1], L.dom[3]), y1 = lpf_ground_cell(hy, L.dom[1], L.dom[3]);
            const int nx = x1 - x0 + 1, nc = nx * (y1 - y0 + 1);
            const unsigned long long bit = 1ull << j;
            for (int i = k; i < nc; i += 4) {
                const int yy = y0 + i / nx, xx = x0 + i - (i / nx) * nx;
                atomicOr(&L.grid[yy * LPF_GRID + xx], bit);
            }
        }
    }
    __syncthreads();
    {
        int t_ = tid;
        asm volatile("" : "+v"(t_));
        for (int i = t_; i < LPF_GRID_CELLS; i += LPF_BLOCK) gf[i] = L.grid[i];
        if (t_ < 4) reinterpret_cast<float *>(gf + LPF_GRID_CELLS)[t_] = L.dom[t_];
    }
}

__global__ __launch_bounds__(LPF_BLOCK) void lpf_box_job_kernel(const LpfBoxJob J)
{
    __shared__ __attribute__((aligned(16))) char s_raw[sizeof(LpfBoxJobLds)];
    lpf_box_frame_block(J, (int)blockIdx.x, s_raw);
}

// ------------------------------------------------------------------------------------
// K8: masks -> label image.
//   MODE 0: uint8, nonzero.  MODE 1: float, astype(uint8) != 0.  MODE 2: float, (x*255) -> u8 == 255.
//   MODE 3: float, x > 0.5 (NaN is not a member).
// ------------------------------------------------------------------------------------
// Streaming pack, 16 pixels per lane: uint8 masks are read 16 bytes per lane per mask
// (float masks 4 x 16 bytes), the packed labels leave as four 16-byte stores.
// Requires hw % 16 == 0 and 16-byte aligned mask planes (checked on the host).
// Mask rectangles (lpf_set_mask_rects): a detector hands out every mask with its 2D box and the mask is zero outside it
// (ultralytics crops the masks to their boxes) -- a real frame's five masks are 2.6 MB of which a few per cent lie inside the boxes.
// With the rectangles a mask is READ AS ZERO outside its rectangle, pixel for pixel: the pack reads a group of 16 pixels of mask m only
// where it meets m's rectangle and keeps the pixels inside; the label image is written in full either way.  uint8 masks under rule 0
// and float masks under rule 1, without erosion (the host passes no rectangles otherwise); tiles that read the masks themselves
// honour the hint the same way (LpfDirectRect), so a result never depends on which of the two forms a launch takes.
// which of the 16 pixels that start at (y, x) -- and run on into row y + 1 when the row ends first (W >= 16) -- lie inside rectangle r
__device__ __forceinline__ unsigned lpf_rect_keep16(const int4 r, const int y, const int x, const int W)
{
    const int n0 = min(16, W - x);                          // pixels of the group on row y
    const int rx0 = max(r.x, 0), rx1 = min(r.z, W);         // (clipped to the image first: INT_MAX as "no limit" must not overflow below)
    unsigned k = 0u;
    if (y >= r.y && y < r.w) {
        const int lo = max(rx0 - x, 0), hi = min(rx1 - x, n0);
        if (hi > lo) k = (0xFFFFu >> (16 - hi)) & (0xFFFFu << lo);
    }
    if (n0 < 16 && y + 1 >= r.y && y + 1 < r.w) {           // (W not a multiple of 16: pixel i >= n0 is column i - n0 of the next row)
        const int lo = rx0 + n0, hi = min(rx1 + n0, 16);
        if (hi > lo) k |= (0xFFFFu >> (16 - hi)) & (0xFFFFu << lo);
    }
    return k & 0xFFFFu;
}

template <typename T, int MODE, typename LT>
__device__ __forceinline__ void lpf_pack16_block(const T *__restrict__ masks, LT *__restrict__ label,
                                                 const int M, const long long hw, const long long total16, const long long blk,
                                                 const int4 *__restrict__ rects, const int W)
{
    const long long g = blk * LPF_BLOCK + threadIdx.x;      // group of 16 pixels, over all frames
    if (g >= total16) return;
    const long long per_frame = hw >> 4;
    const long long f = g / per_frame, o = (g - f * per_frame) << 4;
    const T *__restrict__ mf = masks + (size_t)f * M * hw + o;
    uint32_t bits[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) bits[i] = 0;
    if (sizeof(T) == 1) {
        const int y = rects ? (int)(o / W) : 0, x = rects ? (int)(o - (long long)y * W) : 0;       // the group's first pixel
        for (int m0 = 0; m0 < M; m0 += 8) {                 // eight independent 16-byte loads in flight
            uint4 q[8];
            // which of the group's 16 pixels lie inside mask j's rectangle, 16 bits per mask, two masks per register (the step kernel,
            // of which this is a role, has no registers to spare): a mask is read as zero outside its rectangle, pixel for pixel -- as
            // the tiles that read the masks themselves do (LpfDirectRect) -- and not read at all where the group misses the rectangle
            unsigned kp[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
            if (rects) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const unsigned k = lpf_rect_keep16(rects[(size_t)f * M + min(m0 + j, M - 1)], y, x, W);
                    if (j & 1) kp[j >> 1] = (kp[j >> 1] & 0xFFFFu) | (k << 16); else kp[j >> 1] = (kp[j >> 1] & 0xFFFF0000u) | (k & 0xFFFFu);
                }
                if ((kp[0] | kp[1] | kp[2] | kp[3]) == 0u) continue;          // (most groups of a real frame: nothing to read, nothing to extract)
            }
#pragma unroll
            for (int j = 0; j < 8; ++j)
                q[j] = ((kp[j >> 1] >> (16 * (j & 1))) & 0xFFFFu) ? *reinterpret_cast<const uint4 *>(mf + (size_t)min(m0 + j, M - 1) * hw) : make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const unsigned kj = (kp[j >> 1] >> (16 * (j & 1))) & 0xFFFFu;
                if (m0 + j < M && kj) {
                    unsigned wv[4] = {q[j].x, q[j].y, q[j].z, q[j].w};
                    if (rects && kj != 0xFFFFu) {           // the bytes outside the rectangle read as zero: 4 bits -> 4 byte masks per word
#pragma unroll
                        for (int w = 0; w < 4; ++w) wv[w] &= ((((kj >> (4 * w)) & 0xFu) * 0x00204081u) & 0x01010101u) * 0xFFu;
                    }
#pragma unroll
                    for (int i = 0; i < 16; ++i)
                        bits[i] |= (((wv[i >> 2] >> (8 * (i & 3))) & 0xFFu) != 0u ? 1u : 0u) << (m0 + j);
                }
            }
        }
    } else {
        const int y = rects ? (int)(o / W) : 0, x = rects ? (int)(o - (long long)y * W) : 0;
        for (int m0 = 0; m0 < M; m0 += 2) {
            float4 q[2][4];
            unsigned kk[2] = {0xFFFFu, 0xFFFFu};            // pixels of the group inside the mask's rectangle (as above)
            if (rects) {
#pragma unroll
                for (int j = 0; j < 2; ++j) kk[j] = lpf_rect_keep16(rects[(size_t)f * M + min(m0 + j, M - 1)], y, x, W);
                if ((kk[0] | kk[1]) == 0u) continue;
            }
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    q[j][k] = kk[j] ? *reinterpret_cast<const float4 *>(mf + (size_t)min(m0 + j, M - 1) * hw + 4 * k) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (m0 + j < M && kk[j]) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        bits[4 * k + 0] |= ((lpf_member<float, MODE>(q[j][k].x) && ((kk[j] >> (4 * k + 0)) & 1u)) ? 1u : 0u) << (m0 + j);
                        bits[4 * k + 1] |= ((lpf_member<float, MODE>(q[j][k].y) && ((kk[j] >> (4 * k + 1)) & 1u)) ? 1u : 0u) << (m0 + j);
                        bits[4 * k + 2] |= ((lpf_member<float, MODE>(q[j][k].z) && ((kk[j] >> (4 * k + 2)) & 1u)) ? 1u : 0u) << (m0 + j);
                        bits[4 * k + 3] |= ((lpf_member<float, MODE>(q[j][k].w) && ((kk[j] >> (4 * k + 3)) & 1u)) ? 1u : 0u) << (m0 + j);
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

template <typename T, int MODE, typename LT>
__global__ __launch_bounds__(LPF_BLOCK) void lpf_pack16(const T *__restrict__ masks, LT *__restrict__ label,
                                                        int M, long long hw, long long total16, const int4 *__restrict__ rects, int W)
{
    lpf_pack16_block<T, MODE, LT>(masks, label, M, hw, total16, (long long)blockIdx.x, rects, W);
}

// General-shape pack with optional fused first erosion: 64x16 output tile per block,
// (64+2)x(16+2) LDS tile of packed membership bits; erosion = AND of the plus-shaped
// neighbourhood, pixels outside the image read as all-ones (OpenCV erode border).
#define LPF_TW 64
#define LPF_TH 16

template <typename T, int MODE, typename LT>
__global__ __launch_bounds__(LPF_BLOCK) void lpf_pack_erode(const T *__restrict__ masks, LT *__restrict__ label,
                                                            int M, int H, int W, int erode, const int4 *__restrict__ rects)
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
            for (int m = 0; m < M; ++m) {
                if (rects) {                                // (lpf_set_mask_rects: a mask is read as zero outside its rectangle)
                    const int4 r = rects[(size_t)f * M + m];
                    if (!(x >= r.x && x < r.z && y >= r.y && y < r.w)) continue;
                }
                if (lpf_member<T, MODE>(mf[m * hw + o])) bits |= 1u << m;
            }
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
