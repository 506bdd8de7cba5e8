// lpf_api.hip -- host side of liblpf.so: the C ABI of include/lpf.h over the gfx950
// kernels in lpf_kernels.hip.h.  No CPU compute path exists here: every entry point
// either launches HIP kernels or fails with an error code.
#include "lpf_kernels.hip.h"
#include "../../include/lpf.h"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <mutex>
#include <string>
#include <vector>

// Points per launch up to which a run uses the small geometry: 512-point K1 tiles and 1024-point segments (more,
// shorter blocks and list waves: a single real frame is 107 segments instead of 27).  Measured on single synthetic clouds
// (tools/geometry_probe.py, us per step small / large): 1 M 34.9 / 38.7, 2 M 38.5 / 46.9, 3 M 46.0 / 49.5, 4 M 57.1 / 52.0.
#define LPF_SMALL_LAUNCH (7ll << 19)
// Tail blocks (of four 1024-point segments) up to which the wide form of the tail runs (measured on copies of sample frame 100,
// wide vs narrow: 4 frames = 108 blocks 21.5 vs 26.8 us; 8 frames 24.2 vs 29.2; 12 frames 29.3 vs 33.0; 20 frames = 540 blocks
// 41.0 vs 40.6; 32 frames 55.8 vs 50.6)
#define LPF_WIDE_BELOW 480
#define LPF_FEW_BLOCKS 64           // list blocks of "a frame or two": see lpf_run_batch (count blocks in four parts, 16-row list wave)
// Share of a step launch's K1 tiles (in twentieths) among which the previous run's tail blocks are dealt (see lpf_run_batch)
#ifndef LPF_TAIL_SPREAD_20THS
#define LPF_TAIL_SPREAD_20THS 13
#endif

// K1 tile of a large software-pipelined launch whose tiles read the masks inside their rectangles (LpfDirectRect): 1024 points.
// (2048, what the packed form uses there, needs more registers than the step kernel has -- the exact test of a candidate row computes
//  its pixel again -- and measured slower on real scans even in a form that fitted: 146 frames per step 170 vs 158 us.)
#define LPF_RECT_FUSED_TILE 1024

static_assert(sizeof(lpf_frame_summary) == LPF_SUMMARY_BYTES, "summary layout is shared with lpf_finalize_frame");

namespace {

thread_local std::string g_create_error;

struct DevBuf {                       // grow-only device buffer
    void *p = nullptr;
    size_t cap = 0;
};

}  // namespace

struct lpf_graph {
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    unsigned long long generation = 0;    // lpf_ctx::generation at capture time
};

#define LPF_NSETS 4                   // scratch sets / box sets in rotation (mode 4 keeps four runs in flight)

struct lpf_ctx {
    bool capturing = false;
    // Bumped by everything a captured graph bakes in and a later call may invalidate: buffer regrowth, table
    // uploads, box / camera / stream / pipelining changes.  lpf_graph_launch refuses a graph of another generation.
    unsigned long long generation = 0;
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::string err;

    bool have_camera = false;
    double T[12], K[9], dmin = 0, dmax = 0;
    int W = 0, H = 0;

    // masks -> label images
    int mask_F = 0, mask_M = 0;       // 0 frames = no masks set
    int mask_set = 0;                 // the scratch set whose label image holds them
    DevBuf mask_stage;
    // Masks not packed yet (serial mode, no erosion, host masks in mask_stage or device masks the caller lends: on_device 2):
    // a small launch reads them directly in K1 (LpfDirect), anything else packs them first (ensure_packed).
    struct Lazy { bool valid = false; const void *p = nullptr; bool f32 = false; int mode = 0; const int4 *rects = nullptr; } lazy;

    // Boxes.  A ring of box sets: in the software-pipelined modes the tail that counts into the boxes of run i executes one
    // or two launches after run i was queued, so a lpf_set_boxes* for the NEXT run must not touch the tables run i's tail is
    // going to read -- it takes the next set of the ring instead (no drain, no synchronisation), and the preparation of its
    // tables (LpfBoxJob) rides in that run's launch.  Serial mode stays on one set: the stream orders everything.
    struct BoxSet {
        int F = 0, oriented = 1;
        std::vector<int32_t> box_off;             // F+1
        std::vector<LpfBoxFrame> h_bframes, h_bframes_dev;   // per-frame records: being built / in HBM (F > 1 only)
        std::vector<long long> cand_off;          // [F] first word of frame f's grid
        size_t cand_words = 0;                    // words of the whole grid
        int max_words = 1;                        // 64-bit words per cell of the frame with the most boxes
        DevBuf boxp;                              // [Btot][16] double
        DevBuf boxq;                              // [Btot][8] float conservative AABB
        DevBuf cand;                              // ground grids: LPF_GRID_WORDS words per 64 boxes of a frame
        DevBuf corners;                           // [Btot][8][3] velodyne-frame corners the tables were built from (kept: a camera change rebuilds them)
        DevBuf enabled;                           // [Btot] bytes: 0 = dropped by filter_visible_bboxes (lpf_set_boxes_cam0)
        DevBuf aux;                               // lpf_set_boxes_cam0 for host callers: projected 2D boxes + front counts
        DevBuf bframes;                           // [F] LpfBoxFrame
        DevBuf stage;                             // corners copied from the caller (host memory, or device memory that is not lent)
        bool have_enabled = false;
        bool used = false;                        // a run has been queued with these tables since they were set
        long long last_ref = -1;                  // ... the last such run (lpf_ctx::run_seq)
        bool job_valid = false;                   // tables not built yet: the job rides in the next run's launch (or is launched by it)
        LpfBoxJob job;
    } bx[LPF_NSETS];
    int box_cur = 0;
    bool cand_dirty = false;          // the camera changed since the current set's tables were built: rebuild from its corners

    // Scratch of one in-flight run, including its geometry tables (frame records, per-segment frame records, tail block table):
    // a run whose batch shape differs from the previous one's uploads its own tables -- through the pinned ring below, in stream
    // order, without waiting -- instead of draining the pipeline to rewrite shared ones.  A launch of ONE frame needs no table.
    struct Scratch {
        DevBuf vbal, mbal, seg_tab, grp_tab, frm_tab, seg_pre, cnt, mlist;
        DevBuf label_a, label_b;      // label images [F][H][W] uint32 (b = erosion ping-pong)
        void *label_cur = nullptr;
        int label_bytes = 4;          // element size of the label image: 1 (M <= 8), 2 (M <= 16) or 4
        DevBuf rgrid;                 // candidate grid of the masks' rectangles (LpfDirectRect tiles), [F][cells] uint32
        DevBuf rects;                 // lpf_set_mask_rects from host memory: the copy this set's run reads (its tiles may run a launch later)
        DevBuf tab;                   // [frames | segs | blks]
        std::vector<LpfFrame> tab_frames;         // the frame table `tab` holds (empty: none)
        size_t o_segs = 0, o_blks = 0, o_cblks = 0;
    } sc[LPF_NSETS];
    int parity = 0;
    long long run_seq = 0;            // runs queued so far
    // Software-pipelined modes (lpf_set_pipelined 2 / 4): what earlier runs still owe.  The tail of the last run and the
    // summaries of the one before ride in the next run's launch (lpf_step_t) -- in mode 4 the last run's streaming kernel
    // too (pend_k1) -- or are launched by flush_pending().
    struct Pending {
        bool valid = false;
        LpfParams P;
        bool pre = false;                 // its prefixes come from the scan kernel
        int ntail = 0;                    // tail blocks
        int nk1 = 0, lb = 4;              // (pend_k1) K1 tiles, label element size
        bool small = false;
        bool direct = false;              // its tiles read the lent masks themselves (LpfDirect; small launches)
        int dsel = 0;                     // ... under membership rule 0 (uint8) or 1..3 (float); 4 / 5: inside the masks' rectangles only
                                          // (LpfDirectRect, launches of any size: uint8 rule 0 / float rule 1)
    } pend_k1, pend_tail, pend_fin;
    bool fused = false;               // pipelined: one launch per run (modes 2 / 4)
    // Mode 4: the mask pack rides as well -- the launch of run i carries the pack of run i's masks, the K1 tiles of run i-1
    // (pend_k1), the tail of run i-2 and the summaries of run i-3; four scratch sets.
    bool defer = false;
    // Lent masks of a software-pipelined context that have not been packed: the next run decides -- a small launch reads them
    // directly, a large one in mode 4 lets their pack ride in its launch (uint8, 16-byte aligned planes), anything else packs now.
    struct Ride { bool valid = false; const void *masks = nullptr; bool f32 = false, can_ride = false; int mode = 0, F = 0, M = 0; void *label = nullptr; const int4 *rects = nullptr; } ride;
    // lpf_get_stats: [0] host waits, [1] drains (owed work launched outside a run), [2] uploads through the pinned ring, [3] step
    // launches, [4] box jobs launched as a kernel of their own, [5] box jobs that rode in a step launch, [6] blocking uploads
    long long stats[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    // lpf_set_mask_rects: rectangles for the NEXT lpf_set_masks_* (device pointer: the caller's, or rects_buf), consumed by it
    DevBuf resize_buf;                // lpf_resize_masks_u8: weight tables (+ staging for host callers)
    const int4 *rects_pending = nullptr; int rects_F = 0, rects_M = 0;
    DevBuf lab_clk;                   // LPF_LAB builds (lpf_lab_role_clock): 6 roles x 5 counters, or empty
    int geometry = 0;                 // LPF_LAB builds (lpf_set_geometry): 0 by launch size, 1 small, 2 large, 3 large + scan-kernel prefixes, 4 small + narrow tail

    // Pinned host ring for small uploads (tables, host corners) that must not block: the source of an asynchronous copy has to
    // stay put until the copy has executed, so each upload takes the next piece of the ring; a quarter of the ring is reused only
    // after an event recorded when it was left has completed (one event per 2 MB of uploads, not per upload).
    struct PinRing { char *base = nullptr; size_t cap = 0, head = 0; hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr}; bool rec[4] = {false, false, false, false}; } ring;

    struct PinBack { char *p = nullptr; size_t cap = 0; } pin_back;   // page-locked landing area of small results (lpf_prepare_boxes)

    // host-io staging
    DevBuf pib_box, pib_pts, pib_out, boxprep, dimg, coll;
    DevBuf st_uvv, st_labv;
    DevBuf st_pts, st_uv, st_label, st_depth, st_uf, st_vf, st_valid, st_inst, st_count, st_summary;
    std::vector<LpfFrame> h_frames;   // table being built
    std::vector<char> h_tab;          // [frames | segs | blks] being built

    // optional event bracketing of K1 (lpf_profile_*)
    bool profiling = false;
    std::vector<hipEvent_t> ev;       // pairs: ev[2i] before, ev[2i+1] after
    size_t ev_used = 0;               // pairs recorded since the last reset
};

namespace {

int fail(lpf_ctx *c, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->err = buf; else g_create_error = buf;
    if (c && c->capturing) {              // an error inside a capture ends it: the stream must not stay in capture mode
        c->capturing = false;
        hipGraph_t g = nullptr;
        (void)hipStreamEndCapture(c->stream, &g);
        if (g) (void)hipGraphDestroy(g);
        (void)hipGetLastError();
        c->err += " [the graph capture in progress was abandoned]";
    }
    return code;
}

#define LPF_HIP(c, call)                                                                         \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess)                                                                    \
            return fail((c), LPF_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
                        __FILE__, __LINE__);                                                     \
    } while (0)

hipError_t host_wait(lpf_ctx *c)        // the host blocks until the context's stream is idle (counted: lpf_get_stats)
{
    ++c->stats[0];
    return hipStreamSynchronize(c->stream);
}

// the narrow tail kernel (serial mode)
void launch_tail(hipStream_t st, const LpfParams &P, int ntail, bool pre)
{
    if (pre) hipLaunchKernelGGL((lpf_tail_t<true>), dim3((unsigned)ntail), dim3(LPF_BLOCK), 0, st, P);
    else hipLaunchKernelGGL((lpf_tail_t<false>), dim3((unsigned)ntail), dim3(LPF_BLOCK), 0, st, P);
}

// camera-dependent fields of a box job: filled when the job is launched, so a job that waits for its run always sees the
// camera of that run
void box_job_camera(const lpf_ctx *c, LpfBoxJob &J)
{
    memcpy(J.K, c->K, sizeof J.K);
    J.W = c->W; J.H = c->H;
}

// blocks of a set's box job: frames x 64-box chunks of the frame with the most boxes
int box_job_blocks(const lpf_ctx::BoxSet &B) { return B.F * B.job.chunks; }

// the box job of a set as a kernel of its own, on the context's stream (serial mode, graph capture, drains)
int launch_box_job(lpf_ctx *c, lpf_ctx::BoxSet &B)
{
    if (!B.job_valid) return LPF_OK;
    B.job_valid = false;
    if (B.F == 0 || B.box_off[B.F] == 0) return LPF_OK;
    box_job_camera(c, B.job);
    ++c->stats[4];
    hipLaunchKernelGGL(lpf_box_job_kernel, dim3((unsigned)box_job_blocks(B)), dim3(LPF_BLOCK), 0, c->stream, B.job);
    LPF_HIP(c, hipGetLastError());
    return LPF_OK;
}

// One launch of lpf_step_t: the streaming tiles of run K, the tail blocks of run Q dealt out among them, the summaries of
// run R, the box job X of the run being queued (a block per frame, in front) and -- mode 4 -- the pack of the masks waiting
// in c->ride behind the tiles (label elements of pack_lb bytes; K's when K is there: the host keeps the two equal).  Any of
// the roles may be absent.  `after` is recorded behind the launch.
int launch_step(lpf_ctx *c, const lpf_ctx::Pending &KK, const lpf_ctx::Pending &Q, const lpf_ctx::Pending &R, bool ride, int pack_lb,
                lpf_ctx::BoxSet *XB, hipEvent_t after, const LpfRectJob *RG = nullptr)
{
    static const LpfParams none = {};                      // unused roles get a well-formed struct
    static const LpfBoxJob nojob = {};
    if (KK.valid && KK.direct && KK.dsel >= 4 && Q.valid && Q.pre) {
        // tiles that read the masks inside their rectangles exist without the scan-kernel form of the riding tail (frames of more than
        // 16.7 M points: rare): that tail and the summaries go in a launch of their own, ahead of the tiles
        static const lpf_ctx::Pending nobody = lpf_ctx::Pending();
        int rc_ = launch_step(c, nobody, Q, R, false, pack_lb, nullptr, nullptr);
        if (rc_) return rc_;
        return launch_step(c, KK, nobody, nobody, ride, pack_lb, XB, after, RG);
    }
    const LpfParams &KP = KK.valid ? KK.P : none, &QP = Q.valid ? Q.P : none, &RP = R.valid ? R.P : none;
    const int k_lb = KK.valid ? KK.lb : pack_lb;
    LpfStepLayout Y;
    LpfPackJob J;
    memset(&J, 0, sizeof J);
    Y.nfin = R.valid ? R.P.F : 0;
#ifdef LPF_LAB
    Y.clk = (unsigned long long *)c->lab_clk.p;
#endif
    Y.nfin8 = (Y.nfin + 7) & ~7;
    const bool boxes = XB && XB->job_valid && XB->F > 0 && XB->box_off[XB->F] > 0;
    if (XB) XB->job_valid = false;
    if (boxes) box_job_camera(c, XB->job);
    const LpfBoxJob &XJ = boxes ? XB->job : nojob;
    Y.nbox = boxes ? box_job_blocks(*XB) : 0;
    Y.nbox8 = (Y.nbox + 7) & ~7;
    static const LpfRectJob norect = {};
    const LpfRectJob &GJ = RG ? *RG : norect;
    Y.nrg = RG ? RG->F * RG->bpf : 0;
    Y.nrg8 = (Y.nrg + 7) & ~7;
    Y.ntail = Q.valid ? Q.ntail : 0;
    Y.nk1 = KK.valid ? KK.nk1 : 0;
    Y.npack = 0;
    if (ride) {
        J.masks = (const uint8_t *)c->ride.masks; J.label = c->ride.label; J.M = c->ride.M; J.hw = (long long)c->H * c->W;
        J.rects = c->ride.rects; J.W = c->W;
        J.total16 = (long long)c->ride.F * (J.hw / 16);
        Y.npack = (int)((J.total16 + LPF_BLOCK - 1) / LPF_BLOCK);
    }
    const int nk1_pad = (Y.nk1 + 7) & ~7;
    Y.nper = (Y.ntail + 7) / 8;
    Y.kper = Y.nk1 > 0 ? 8 : 0;                            // (a drain launch without tiles: the tail blocks alone, no empty tile slots)
    if (Y.nper > 0 && Y.nk1 > 0) {                                      // spread the tail blocks over the first two thirds of the tiles (same box,
        // us per step at 40 / 50 / 60 / 70 / 80 / 90 %: 105.6 / 103-105.6 / 99.5-101.6 / 100.3-101.1 / 101.8-102.3 / 102.3-102.5;
        // again with the pack riding, 40 / 50 / 65 / 80 / 95 %: 95.4 / 95.6-95.9 / 91.4-92.0 / 93.2-93.3 / 93.9:
        // early enough that the last tail blocks do not outlive the tiles, late enough not to crowd the start)
        const long long k = ((long long)nk1_pad * LPF_TAIL_SPREAD_20THS / 20 / 8) / Y.nper;
        Y.kper = (int)(k < 1 ? 1 : k) * 8;
    }
    const long long rest = (long long)nk1_pad - (long long)Y.nper * Y.kper;
    Y.rest = (int)(rest > 0 ? rest : 0);
    const long long grid = (long long)Y.nfin8 + Y.nbox8 + Y.nrg8 + (long long)Y.nper * (Y.kper + 8) + Y.rest + Y.npack;
    if (grid > 0) {
        const dim3 gs((unsigned)grid);
        ++c->stats[3];
        if (boxes) ++c->stats[5];
#define LPF_STEP_LAUNCH(RW, LT, PR) do { if (boxes) hipLaunchKernelGGL((lpf_step_t<RW, LPF_K1_FLAGS, LT, PR, true>), gs, dim3(LPF_BLOCK), 0, c->stream, KP, QP, RP, Y, J, XJ, GJ); \
                                         else hipLaunchKernelGGL((lpf_step_t<RW, LPF_K1_FLAGS, LT, PR, false>), gs, dim3(LPF_BLOCK), 0, c->stream, KP, QP, RP, Y, J, XJ, GJ); } while (0)
#define LPF_STEP_LT(RW, PR) do { if (k_lb == 1) LPF_STEP_LAUNCH(RW, uint8_t, PR); else if (k_lb == 2) LPF_STEP_LAUNCH(RW, uint16_t, PR); else LPF_STEP_LAUNCH(RW, uint32_t, PR); } while (0)
        const bool qpre = Q.valid && Q.pre;
        const int rows = KP.tile_pts >> 8;
        if (KK.valid && KK.direct && KK.dsel >= 4) {       // tiles of any size that read the masks inside their rectangles (no pack, no label image)
            typedef LpfDirectRect<uint8_t, 0> R0; typedef LpfDirectRect<float, 1> R1;
#define LPF_STEP_RECT(RW) do { if (KK.dsel == 4) LPF_STEP_LAUNCH(RW, R0, false); else LPF_STEP_LAUNCH(RW, R1, false); } while (0)
            if (rows == 2) LPF_STEP_RECT(2); else LPF_STEP_RECT(4);
#undef LPF_STEP_RECT
        } else
        if (KK.valid && KK.direct) {                       // small launch whose tiles read the lent masks themselves (no pack role beside it)
            typedef LpfDirect<uint8_t, 0> D0; typedef LpfDirect<float, 1> D1; typedef LpfDirect<float, 2> D2; typedef LpfDirect<float, 3> D3;
#define LPF_STEP_DIRECT(D) do { if (qpre) LPF_STEP_LAUNCH(2, D, true); else LPF_STEP_LAUNCH(2, D, false); } while (0)
            if (KK.dsel == 0) LPF_STEP_DIRECT(D0); else if (KK.dsel == 1) LPF_STEP_DIRECT(D1); else if (KK.dsel == 2) LPF_STEP_DIRECT(D2); else LPF_STEP_DIRECT(D3);
#undef LPF_STEP_DIRECT
        } else
        if (rows == 2) { if (qpre) LPF_STEP_LT(2, true); else LPF_STEP_LT(2, false); }
        else if (rows == 8) { if (qpre) LPF_STEP_LT(8, true); else LPF_STEP_LT(8, false); }
        else       { if (qpre) LPF_STEP_LT(4, true); else LPF_STEP_LT(4, false); }
#undef LPF_STEP_LT
#undef LPF_STEP_LAUNCH
        LPF_HIP(c, hipGetLastError());
    }
    if (after) LPF_HIP(c, hipEventRecord(after, c->stream));
    if (KK.valid && KK.pre && KK.P.nseg_total > 0) {       // frames beyond 64 groups: their prefixes are scanned before the tail rides
        hipLaunchKernelGGL(lpf_scan_segments, dim3(KK.P.F), dim3(LPF_BLOCK), 0, c->stream, KK.P);
        LPF_HIP(c, hipGetLastError());
    }
    return LPF_OK;
}

bool anything_owed(const lpf_ctx *c) { return c->pend_k1.valid || c->pend_tail.valid || c->pend_fin.valid; }

// software-pipelined modes: launch what earlier runs still owe -- the same step launches, with fewer roles each time; a box job
// that was waiting for its run goes with the first of them (or alone)
int flush_pending(lpf_ctx *c)
{
    lpf_ctx::BoxSet *XB = &c->bx[c->box_cur];
    if (anything_owed(c)) ++c->stats[1];
    while (anything_owed(c)) {
        const lpf_ctx::Pending K = c->pend_k1, Q = c->pend_tail, R = c->pend_fin;
        int rc_ = launch_step(c, K, Q, R, false, 4, XB, nullptr);
        if (rc_) return rc_;
        c->pend_fin = Q;                   // its tail has just been launched: summaries next
        c->pend_tail = K;
        c->pend_k1.valid = false;
    }
    return launch_box_job(c, *XB);
}

int sync_all(lpf_ctx *c)                // the stream idle, nothing owed: any table / buffer may be rewritten or freed
{
    if (c->capturing)
        return fail(c, LPF_ERR_STATE, "this call needs a synchronisation or (re)allocation, which cannot be captured into a graph: "
                                      "run the same shapes once before lpf_graph_begin");
    { int rc_ = flush_pending(c); if (rc_) return rc_; }
    LPF_HIP(c, host_wait(c));
    for (bool &r : c->ring.rec) r = false;                  // every upload has executed
    return LPF_OK;
}

// grow-only; new memory is zeroed (the self-cleaning counters rely on it)
int reserve(lpf_ctx *c, DevBuf &b, size_t bytes, bool zero = false)
{
    if (bytes <= b.cap && b.p) return LPF_OK;
    if (c->capturing) return fail(c, LPF_ERR_STATE, "allocation inside graph capture: run the same shapes once before lpf_graph_begin");
    if (bytes == 0) bytes = 256;
    if (b.p) {
        int rc_ = sync_all(c);
        if (rc_) return rc_;
        LPF_HIP(c, hipFree(b.p));
        b.p = nullptr; b.cap = 0;
        ++c->generation;                  // a captured graph may hold the freed pointer
    }
    const size_t want = bytes + bytes / 4;        // headroom against regrowth
    if (hipMalloc(&b.p, want) != hipSuccess) {
        b.p = nullptr;
        return fail(c, LPF_ERR_NOMEM, "hipMalloc(%zu) failed", want);
    }
    b.cap = want;
    if (zero) LPF_HIP(c, hipMemsetAsync(b.p, 0, want, c->stream));
    return LPF_OK;
}

void release(DevBuf &b)
{
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr; b.cap = 0;
}

int use_device(lpf_ctx *c)
{
    LPF_HIP(c, hipSetDevice(c->device));
    return LPF_OK;
}

// `bytes` of host memory -> device memory `dst`, in stream order, WITHOUT waiting: the data is copied into the pinned ring
// (which an asynchronous copy may read from whenever it executes) and the copy is queued from there.  Only an upload larger
// than a quarter of the ring takes the blocking route: everything owed is launched, the stream drained, the copy made.
#define LPF_RING_BYTES (8u << 20)
int upload(lpf_ctx *c, void *dst, const void *src, size_t bytes)
{
    if (bytes == 0) return LPF_OK;
    if (c->capturing) return fail(c, LPF_ERR_STATE, "a table upload cannot be captured into a graph: run the same shapes once before lpf_graph_begin");
    lpf_ctx::PinRing &R = c->ring;
    if (!R.base) {
        LPF_HIP(c, hipHostMalloc((void **)&R.base, LPF_RING_BYTES, hipHostMallocDefault));
        R.cap = LPF_RING_BYTES; R.head = 0;
        for (hipEvent_t &e : R.ev) LPF_HIP(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    const size_t q = R.cap / 4, need = (bytes + 255) & ~(size_t)255;
    if (need >= q) {
        int rc_ = sync_all(c);
        if (rc_) return rc_;
        LPF_HIP(c, hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
        ++c->stats[6];
        return LPF_OK;
    }
    // Invariant: head < cap, and head never rests on a quarter's end -- a piece that would fill its quarter exactly starts the
    // next one instead (">=": every `need` and q are multiples of 256, so exact fills are the common case, not the rare one; with
    // ">" the head walked to `cap` after 32768 small uploads and the next copy went past the ring).
    size_t cur = R.head / q;
    if (R.head - cur * q + need >= q) {                     // leave this quarter: it is reusable once what was queued from it has run
        LPF_HIP(c, hipEventRecord(R.ev[cur], c->stream));
        R.rec[cur] = true;
        cur = (cur + 1) & 3;
        R.head = cur * q;
        if (R.rec[cur]) {
            if (hipEventQuery(R.ev[cur]) != hipSuccess) { ++c->stats[0]; LPF_HIP(c, hipEventSynchronize(R.ev[cur])); }
            R.rec[cur] = false;
        }   // a full lap of uploads ago: long done
    }
    char *h = R.base + R.head;
    R.head += need;
    memcpy(h, src, bytes);
    ++c->stats[2];
    LPF_HIP(c, hipMemcpyAsync(dst, h, bytes, hipMemcpyHostToDevice, c->stream));
    return LPF_OK;
}

// The address a kernel may write for a host pointer that lies in page-locked, GPU-mapped memory (hipHostMalloc / lpf_host_alloc,
// hipHostRegister), or NULL for any other memory.  Asked every time: the answer for an address can change between calls.
void *device_alias_of_pinned(void *host)
{
    hipPointerAttribute_t at;
    memset(&at, 0, sizeof at);
    if (hipPointerGetAttributes(&at, host) != hipSuccess) { (void)hipGetLastError(); return nullptr; }    // (plain malloc memory: an error in some runtimes)
    // (hostPointer == host: the answer is about THIS address -- a pointer into the middle of an allocation is only taken when the runtime
    //  says so itself, never by assuming that its device alias lies at the same offset)
    return (at.type == hipMemoryTypeHost && at.hostPointer == host) ? at.devicePointer : nullptr;
}

// box parameters in the oracle's arithmetic (oracle/lpf_oracle.c: orc_oriented_inside) for lpf_points_in_boxes
void box_params(const double *c, int oriented, double *o)
{
    for (int i = 0; i < 16; ++i) o[i] = 0.0;
    if (oriented) {
        static const int other[3] = {1, 3, 4};
        o[0] = c[0]; o[1] = c[1]; o[2] = c[2];
        bool ok = true;
        for (int a = 0; a < 3; ++a) {
            const double *p = c + 3 * other[a];
            const double v0 = p[0] - c[0], v1 = p[1] - c[1], v2 = p[2] - c[2];
            double w = v0 * v0; w = std::fma(v1, v1, w); w = std::fma(v2, v2, w);
            o[3 + 4 * a] = v0; o[4 + 4 * a] = v1; o[5 + 4 * a] = v2; o[6 + 4 * a] = w;
            if (!(w >= 1e-100 && w <= 1e100)) ok = false;      // also false for NaN; the range in which the kernel's
                                                               // division-free slab test is provably the quotient test (see lpf_oriented_inside)
        }
        o[15] = ok ? 1.0 : 0.0;            // 1: "0 <= d <= vv" decides the slab exactly (see kernel comment)
    } else {
        for (int k = 0; k < 3; ++k) {
            double a = c[k], b = c[k];
            for (int j = 1; j < 8; ++j) {
                const double w = c[3 * j + k];
                if (w < a) a = w;
                if (w > b) b = w;
            }
            o[k] = a; o[3 + k] = b;
        }
    }
}

// The box set a lpf_set_boxes* call writes: in the software-pipelined modes the next one of the ring once the current one has
// been used by a run (its tables are still to be read by that run's tail), else the current one.  Then its shapes: per-frame
// records, grid offsets, buffers.  Nothing here waits for the GPU unless a buffer has to grow.
int box_layout(lpf_ctx *c, const int32_t *box_off, int F, int oriented, const char *who, lpf_ctx::BoxSet **out)
{
    if (F < 0 || (F > 0 && !box_off)) return fail(c, LPF_ERR_ARG, "%s: F=%d box_off=%p", who, F, (const void *)box_off);
    if (F > 0 && box_off[0] != 0) return fail(c, LPF_ERR_ARG, "%s: box_off[0] must be 0", who);
    for (int f = 0; f < F; ++f)
        if (box_off[f + 1] < box_off[f]) return fail(c, LPF_ERR_ARG, "%s: box_off not ascending at %d", who, f);
    if (!c->have_camera) return fail(c, LPF_ERR_STATE, "%s: lpf_set_camera must be called first (cam-0 boxes are filtered with its K, W, H)", who);
    int rc;
    lpf_ctx::BoxSet *B = &c->bx[c->box_cur];
    if (c->fused && !c->capturing) {
        if (B->used) {
            B->job_valid = false;                          // (cannot be: a run launches the job of the set it uses)
            c->box_cur = (c->box_cur + 1) % LPF_NSETS;
            B = &c->bx[c->box_cur];
        }
        // the tail of the last run that used this set's old tables must have been LAUNCHED before they are rewritten (in stream
        // order); with one set per run and four sets it always has -- else launch what is owed first (still no host wait)
        const int depth = c->defer ? 2 : 1;
        if (B->last_ref >= 0 && B->last_ref + depth >= c->run_seq && (rc = flush_pending(c))) return rc;
    }
    B->used = false; B->last_ref = -1; B->job_valid = false; B->F = 0;
    B->h_bframes.resize((size_t)F);
    B->cand_off.assign((size_t)F, 0);
    B->max_words = 1;
    size_t total = 0;
    for (int f = 0; f < F; ++f) {
        const int nb = box_off[f + 1] - box_off[f];
        B->h_bframes[f].box_off = box_off[f]; B->h_bframes[f].B = nb; B->h_bframes[f].cand_off = (long long)total;
        B->cand_off[f] = (long long)total;
        total += (size_t)LPF_GRID_WORDS * (size_t)((nb + 63) / 64);      // a ground grid per 64 boxes
        if ((nb + 63) / 64 > B->max_words) B->max_words = (nb + 63) / 64;
    }
    const int Btot = F ? box_off[F] : 0;
    const size_t nb = (size_t)(Btot ? Btot : 1);
    if ((rc = reserve(c, B->boxp, nb * 16 * sizeof(double)))) return rc;
    if ((rc = reserve(c, B->boxq, nb * 8 * sizeof(float)))) return rc;
    if ((rc = reserve(c, B->cand, (total ? total : 1) * 8))) return rc;
    if ((rc = reserve(c, B->corners, nb * 24 * sizeof(double)))) return rc;
    if ((rc = reserve(c, B->enabled, nb))) return rc;
    if (F > 1) {
        if ((rc = reserve(c, B->bframes, (size_t)F * sizeof(LpfBoxFrame)))) return rc;
        if (B->h_bframes_dev.size() != B->h_bframes.size() ||
            memcmp(B->h_bframes_dev.data(), B->h_bframes.data(), (size_t)F * sizeof(LpfBoxFrame)) != 0) {
            if ((rc = upload(c, B->bframes.p, B->h_bframes.data(), (size_t)F * sizeof(LpfBoxFrame)))) return rc;
            B->h_bframes_dev = B->h_bframes;
            ++c->generation;                                  // graphs captured for other box counts index these tables
        }
    }
    B->cand_words = total;
    B->box_off.assign(box_off, box_off + F + 1);
    B->F = F; B->oriented = oriented ? 1 : 0;
    *out = B;
    return LPF_OK;
}

// the job that (re)builds a set's tables from its own copy of the velodyne-frame corners (the camera changed)
void box_rebuild_job(lpf_ctx::BoxSet &B)
{
    LpfBoxJob &J = B.job;
    memset(&J, 0, sizeof J);
    J.src = (const double *)B.corners.p;
    J.enabled_in = B.have_enabled ? (const uint8_t *)B.enabled.p : nullptr;
    J.oriented = B.oriented; J.F = B.F; J.chunks = B.max_words;
    J.bframes = (const LpfBoxFrame *)B.bframes.p;
    if (B.F > 0) J.frame0 = B.h_bframes[0];
    J.boxp = (double *)B.boxp.p; J.boxq = (float *)B.boxq.p; J.cand = (unsigned long long *)B.cand.p;
    B.job_valid = true;
}

// lpf_set_boxes_ex / lpf_set_boxes_cam0: corners (velodyne or cam-0 frame) -> the job that builds the set's tables.  The job is
// launched here when the caller wants results in host memory (then the call waits for them), in serial mode and inside a graph
// capture; a software-pipelined context leaves it to the next lpf_run*, in whose launch it rides.
int set_boxes_impl(lpf_ctx *c, const double *corners, int on_device, const int32_t *box_off, int F, int oriented, bool cam0, const double *Tcv,
                   int filter_visible, uint8_t *visible, double *corners_velo, double *bbox2d, int32_t *front, const char *who)
{
    if (!c) return LPF_ERR_ARG;
    if (use_device(c)) return LPF_ERR_HIP;
    if (on_device < 0 || on_device > 2) return fail(c, LPF_ERR_ARG, "%s: on_device=%d (0 host, 1 device, 2 device and lent)", who, on_device);
    if (F == 0) { lpf_ctx::BoxSet &B0 = c->bx[c->box_cur]; B0.F = 0; B0.box_off.clear(); B0.job_valid = false; return LPF_OK; }
    if (cam0 && !Tcv) return fail(c, LPF_ERR_ARG, "%s: T_cam_to_velo is NULL", who);
    if (box_off && F > 0 && box_off[F] > 0 && !corners) return fail(c, LPF_ERR_ARG, "%s: corners is NULL", who);
    int rc;
    lpf_ctx::BoxSet *B = nullptr;
    if ((rc = box_layout(c, box_off, F, oriented, who, &B))) return rc;
    const int Btot = box_off[F];
    B->have_enabled = cam0 && filter_visible != 0;
    c->cand_dirty = false;                                 // the job reads the camera when it is launched
    if (Btot == 0) return LPF_OK;
    const size_t nb = (size_t)Btot;
    const double *src = corners;
    if (on_device != 2) {                                  // not lent: the context's own copy, made in stream order
        if ((rc = reserve(c, B->stage, nb * 192))) { B->F = 0; return rc; }
        if (on_device) LPF_HIP(c, hipMemcpyAsync(B->stage.p, corners, nb * 192, hipMemcpyDeviceToDevice, c->stream));
        else if ((rc = upload(c, B->stage.p, corners, nb * 192))) { B->F = 0; return rc; }
        src = (const double *)B->stage.p;
    }
    const bool host_out = !on_device && (visible || corners_velo || bbox2d || front);
    if (host_out && (rc = reserve(c, B->aux, nb * 36))) { B->F = 0; return rc; }
    LpfBoxJob &J = B->job;
    memset(&J, 0, sizeof J);
    J.src = src; J.cam0 = cam0 ? 1 : 0; J.filter_visible = filter_visible ? 1 : 0; J.oriented = B->oriented; J.F = F; J.chunks = B->max_words;
    if (cam0) memcpy(J.Tcv, Tcv, sizeof J.Tcv);
    J.bframes = (const LpfBoxFrame *)B->bframes.p; J.frame0 = B->h_bframes[0];
    J.boxp = (double *)B->boxp.p; J.boxq = (float *)B->boxq.p; J.cand = (unsigned long long *)B->cand.p;
    J.corners_keep = (double *)B->corners.p;
    if (cam0) {
        J.enabled_out = (uint8_t *)B->enabled.p;
        if (on_device) { J.visible = visible; J.corners_out = corners_velo; J.bbox2d = bbox2d; J.front = front; }
        else if (host_out) { J.bbox2d = (double *)B->aux.p; J.front = (int32_t *)((char *)B->aux.p + nb * 32); }
    }
    B->job_valid = true;
    // The job waits for the next lpf_run* and rides in the launch of its streaming tiles (every mode: in order too the tables are
    // only read by the tail, one launch later) -- unless the caller wants results in host memory now.
    if (host_out && (rc = launch_box_job(c, *B))) return rc;
    if (host_out) {
        if (visible) LPF_HIP(c, hipMemcpyAsync(visible, B->enabled.p, nb, hipMemcpyDeviceToHost, c->stream));
        if (corners_velo) LPF_HIP(c, hipMemcpyAsync(corners_velo, B->corners.p, nb * 192, hipMemcpyDeviceToHost, c->stream));
        if (bbox2d) LPF_HIP(c, hipMemcpyAsync(bbox2d, B->aux.p, nb * 32, hipMemcpyDeviceToHost, c->stream));
        if (front) LPF_HIP(c, hipMemcpyAsync(front, (char *)B->aux.p + nb * 32, nb * 4, hipMemcpyDeviceToHost, c->stream));
        LPF_HIP(c, host_wait(c));       // the caller's host buffers are filled on return
    }
    return LPF_OK;
}

// masks -> label image with element type LT, on stream ms, into S.label_a (S.label_b = erosion ping-pong)
template <typename T, typename LT>
int pack_typed(lpf_ctx *c, lpf_ctx::Scratch &S, hipStream_t ms, const T *d_masks, int F, int M, int mode, int erode_iters, void **result,
               const int4 *rects = nullptr)
{
    const size_t hw = (size_t)c->H * c->W;
    dim3 grid((c->W + LPF_TW - 1) / LPF_TW, (c->H + LPF_TH - 1) / LPF_TH, F);
    LT *cur = (LT *)S.label_a.p;
    int rc;
    if (M == 0) {
        LPF_HIP(c, hipMemsetAsync(cur, 0, (size_t)F * hw * sizeof(LT), ms));
    } else {
        int left = erode_iters;
        if (hw % 16 == 0 && ((uintptr_t)d_masks & 15) == 0) {
            // streaming pack, 16 pixels per lane; erosion (if any) then runs on the packed image
            const long long total16 = (long long)F * (long long)(hw / 16);
            const unsigned nb = (unsigned)((total16 + LPF_BLOCK - 1) / LPF_BLOCK);
            // (rectangles: only where set_masks_impl accepted them -- uint8 rule 0 / float rule 1, no erosion)
            if (mode == 0)
                hipLaunchKernelGGL((lpf_pack16<T, 0, LT>), dim3(nb), dim3(LPF_BLOCK), 0, ms, d_masks, cur, M, (long long)hw, total16, rects, c->W);
            else if (mode == 1)
                hipLaunchKernelGGL((lpf_pack16<T, 1, LT>), dim3(nb), dim3(LPF_BLOCK), 0, ms, d_masks, cur, M, (long long)hw, total16, rects, c->W);
            else if (mode == 2)
                hipLaunchKernelGGL((lpf_pack16<T, 2, LT>), dim3(nb), dim3(LPF_BLOCK), 0, ms, d_masks, cur, M, (long long)hw, total16, (const int4 *)nullptr, c->W);
            else
                hipLaunchKernelGGL((lpf_pack16<T, 3, LT>), dim3(nb), dim3(LPF_BLOCK), 0, ms, d_masks, cur, M, (long long)hw, total16, (const int4 *)nullptr, c->W);
        } else {
            const int fuse = erode_iters > 0 ? 1 : 0;
            left -= fuse;
            if (mode == 0)
                hipLaunchKernelGGL((lpf_pack_erode<T, 0, LT>), grid, dim3(LPF_BLOCK), 0, ms, d_masks, cur, M, c->H, c->W, fuse, rects);
            else if (mode == 1)
                hipLaunchKernelGGL((lpf_pack_erode<T, 1, LT>), grid, dim3(LPF_BLOCK), 0, ms, d_masks, cur, M, c->H, c->W, fuse, rects);
            else if (mode == 2)
                hipLaunchKernelGGL((lpf_pack_erode<T, 2, LT>), grid, dim3(LPF_BLOCK), 0, ms, d_masks, cur, M, c->H, c->W, fuse, (const int4 *)nullptr);
            else
                hipLaunchKernelGGL((lpf_pack_erode<T, 3, LT>), grid, dim3(LPF_BLOCK), 0, ms, d_masks, cur, M, c->H, c->W, fuse, (const int4 *)nullptr);
        }
        LPF_HIP(c, hipGetLastError());
        if (left > 0) {
            if ((rc = reserve(c, S.label_b, (size_t)F * hw * 4))) return rc;
            LT *other = (LT *)S.label_b.p;
            for (int it = 0; it < left; ++it) {
                hipLaunchKernelGGL((lpf_erode_packed<LT>), grid, dim3(LPF_BLOCK), 0, ms, cur, other, c->H, c->W);
                LPF_HIP(c, hipGetLastError());
                LT *t = cur; cur = other; other = t;
            }
        }
    }
    *result = cur;
    return LPF_OK;
}

template <typename T>
int set_masks_impl(lpf_ctx *c, const T *masks, int F, int M, int mode, int erode_iters, int on_device)
{
    if (!c) return LPF_ERR_ARG;
    if (use_device(c)) return LPF_ERR_HIP;
    if (!c->have_camera) return fail(c, LPF_ERR_STATE, "lpf_set_camera must be called before masks (W, H)");
    if (M > LPF_MAX_MASKS)
        return fail(c, LPF_ERR_ARG, "set_masks: M=%d masks per frame, a run takes at most LPF_MAX_MASKS = %d (one bit each in label_bits): run the frame "
                                    "once per group of %d masks -- everything per point is the same in every pass (pipeline.run_frames does so)",
                    M, LPF_MAX_MASKS, LPF_MAX_MASKS);
    if (F < 0 || M < 0 || erode_iters < 0 || (F > 0 && M > 0 && !masks))
        return fail(c, LPF_ERR_ARG, "set_masks: F=%d M=%d erode_iters=%d masks=%p", F, M, erode_iters, (const void *)masks);
    // software-pipelined modes + device masks: the label images rotate with the scratch sets (the tiles of the previous run may
    // still have to read theirs); anything else uses set 0, with nothing owed
    const bool per_set = c->fused && on_device && !c->capturing;
    int rc;
    if (!per_set && anything_owed(c) && (rc = sync_all(c))) return rc;    // owed tiles read set 0's label image / the staging copy
    lpf_ctx::Scratch &S = c->sc[per_set ? c->parity : 0];
    hipStream_t ms = c->stream;
    c->mask_F = 0; c->mask_M = 0; S.label_cur = nullptr; c->lazy.valid = false;
    c->mask_set = per_set ? c->parity : 0;
    if (F == 0) return LPF_OK;
    const size_t hw = (size_t)c->H * c->W;
    if ((rc = reserve(c, S.label_a, (size_t)F * hw * 4))) return rc;
    const T *d_masks = masks;
    if (M > 0 && !on_device) {
        const size_t bytes = (size_t)F * M * hw * sizeof(T);
        if ((rc = reserve(c, c->mask_stage, bytes))) return rc;
        LPF_HIP(c, hipMemcpyAsync(c->mask_stage.p, masks, bytes, hipMemcpyHostToDevice, c->stream));
        d_masks = (const T *)c->mask_stage.p;
    }
    const int lb = (M <= 8) ? 1 : (M <= 16) ? 2 : 4;
    // rectangles given for these masks (lpf_set_mask_rects): used where uint8 masks are packed as they are; consumed either way
    const int4 *rects = (((sizeof(T) == 1 && mode == 0) || (sizeof(T) == 4 && mode == 1)) && erode_iters == 0 && c->rects_F == F && c->rects_M == M &&
                         c->W >= 16) ? c->rects_pending : nullptr;         // (W >= 16: a group of 16 pixels of the pack spans at most two rows)
    c->rects_pending = nullptr;
    c->ride.valid = false;
    if (c->fused && per_set && M > 0 && erode_iters == 0 && on_device == 2) {
        // software-pipelined modes, lent masks: left to the next lpf_run* (see lpf_ctx::Ride)
        c->ride.valid = true; c->ride.masks = d_masks; c->ride.F = F; c->ride.M = M; c->ride.label = S.label_a.p;
        c->ride.f32 = sizeof(T) == 4; c->ride.mode = mode; c->ride.rects = rects;
        c->ride.can_ride = c->defer && sizeof(T) == 1 && hw % 16 == 0 && ((uintptr_t)d_masks & 15) == 0;
        S.label_bytes = lb;
        S.label_cur = S.label_a.p;
        c->mask_F = F; c->mask_M = M;
        return LPF_OK;
    }
    if (M > 0 && erode_iters == 0 && !c->fused && on_device != 1) {
        // serial mode, nothing to erode, and the masks stay where they are (our staging buffer, or lent by the caller):
        // packing is left to the run -- a small launch does without it
        c->lazy.valid = true; c->lazy.p = d_masks; c->lazy.f32 = sizeof(T) == 4; c->lazy.mode = mode; c->lazy.rects = rects;
        S.label_bytes = lb;
        if (!on_device) LPF_HIP(c, host_wait(c));
        c->mask_F = F; c->mask_M = M;
        return LPF_OK;
    }
    void *cur = nullptr;
    if (lb == 1) rc = pack_typed<T, uint8_t>(c, S, ms, d_masks, F, M, mode, erode_iters, &cur, rects);
    else if (lb == 2) rc = pack_typed<T, uint16_t>(c, S, ms, d_masks, F, M, mode, erode_iters, &cur, rects);
    else rc = pack_typed<T, uint32_t>(c, S, ms, d_masks, F, M, mode, erode_iters, &cur, rects);
    if (rc) return rc;
    S.label_bytes = lb;
    if (!on_device) LPF_HIP(c, host_wait(c));   // the host buffer may be reused by the caller
    S.label_cur = cur;
    c->mask_F = F; c->mask_M = M;
    return LPF_OK;
}

// masks [F][M][H][W] (uint8 under rule 0, or float under rule mode) -> label image of scratch set S, by a launch of their own
int pack_masks_now(lpf_ctx *c, lpf_ctx::Scratch &S, const void *masks, bool f32, int mode, int F, int M, const int4 *rects)
{
    const int lb = S.label_bytes;
    void *cur = nullptr;
    int rc;
    if (f32) {
        const float *m = (const float *)masks;
        if (lb == 1) rc = pack_typed<float, uint8_t>(c, S, c->stream, m, F, M, mode, 0, &cur, rects);
        else if (lb == 2) rc = pack_typed<float, uint16_t>(c, S, c->stream, m, F, M, mode, 0, &cur, rects);
        else rc = pack_typed<float, uint32_t>(c, S, c->stream, m, F, M, mode, 0, &cur, rects);
    } else {
        const uint8_t *m = (const uint8_t *)masks;
        if (lb == 1) rc = pack_typed<uint8_t, uint8_t>(c, S, c->stream, m, F, M, mode, 0, &cur, rects);
        else if (lb == 2) rc = pack_typed<uint8_t, uint16_t>(c, S, c->stream, m, F, M, mode, 0, &cur, rects);
        else rc = pack_typed<uint8_t, uint32_t>(c, S, c->stream, m, F, M, mode, 0, &cur, rects);
    }
    if (rc) return rc;
    S.label_cur = cur;
    return LPF_OK;
}

// lent masks of a pipelined context still unpacked (lpf_ctx::Ride) -> packed now
int pack_ride_now(lpf_ctx *c)
{
    if (!c->ride.valid) return LPF_OK;
    int rc = pack_masks_now(c, c->sc[c->mask_set], c->ride.masks, c->ride.f32, c->ride.mode, c->ride.F, c->ride.M, c->ride.rects);
    if (rc) return rc;
    c->ride.valid = false;
    return LPF_OK;
}

// masks left unpacked by lpf_set_masks_* in serial mode (lpf_ctx::Lazy) -> label image of scratch set 0
int ensure_packed(lpf_ctx *c)
{
    if (!c->lazy.valid) return LPF_OK;
    int rc = pack_masks_now(c, c->sc[0], c->lazy.p, c->lazy.f32, c->lazy.mode, c->mask_F, c->mask_M, c->lazy.rects);
    if (rc) return rc;
    c->lazy.valid = false;
    return LPF_OK;
}

}  // namespace

extern "C" {

int lpf_abi_version(void) { return LPF_ABI_VERSION; }

// The sources this binary was compiled from (see include/lpf.h).  The marker string is what _build.library_id() looks for in
// the file, so a stale library is recognised without loading it.
#ifndef LPF_BUILD_ID_STR
#define LPF_BUILD_ID_STR "unknown"
#endif
static const char lpf_build_marker[] = "LPF_BUILD_ID=" LPF_BUILD_ID_STR;
const char *lpf_build_id(void) { return lpf_build_marker + 13; }

void *lpf_host_alloc(size_t bytes)
{
    void *p = nullptr;
    const hipError_t e = hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault);
    if (e != hipSuccess) { fail(nullptr, LPF_ERR_NOMEM, "lpf_host_alloc(%zu): %s", bytes, hipGetErrorString(e)); return nullptr; }
    return p;
}

void lpf_host_free(void *p) { if (p) (void)hipHostFree(p); }

int lpf_create(lpf_ctx **out, int device_id)
{
    if (!out) return fail(nullptr, LPF_ERR_ARG, "lpf_create: out is NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(nullptr, LPF_ERR_HIP, "no HIP device: %s", e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
    if (device_id < 0 || device_id >= n) return fail(nullptr, LPF_ERR_ARG, "device_id %d out of range [0,%d)", device_id, n);
    lpf_ctx *c = new (std::nothrow) lpf_ctx();
    if (!c) return fail(nullptr, LPF_ERR_NOMEM, "out of host memory");
    c->device = device_id;
    if ((e = hipSetDevice(device_id)) != hipSuccess || (e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess) {
        fail(nullptr, LPF_ERR_HIP, "stream creation failed: %s", hipGetErrorString(e));
        delete c;
        return LPF_ERR_HIP;
    }
    c->own_stream = true;
    *out = c;
    return LPF_OK;
}

void lpf_destroy(lpf_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    for (auto &S : c->sc) {
        DevBuf *sb[] = {&S.vbal, &S.mbal, &S.seg_tab, &S.grp_tab, &S.frm_tab, &S.seg_pre, &S.cnt, &S.mlist, &S.label_a, &S.label_b, &S.tab, &S.rects, &S.rgrid};
        for (DevBuf *b : sb) release(*b);
    }
    for (auto &B : c->bx) {
        DevBuf *bb[] = {&B.boxp, &B.boxq, &B.cand, &B.corners, &B.enabled, &B.aux, &B.bframes, &B.stage};
        for (DevBuf *b : bb) release(*b);
    }
    DevBuf *all[] = {&c->resize_buf, &c->lab_clk, &c->mask_stage, &c->pib_box, &c->pib_pts, &c->pib_out, &c->boxprep, &c->dimg, &c->coll, &c->st_uvv, &c->st_labv, &c->st_pts, &c->st_uv, &c->st_label,
                     &c->st_depth, &c->st_uf, &c->st_vf, &c->st_valid, &c->st_inst, &c->st_count, &c->st_summary};
    for (DevBuf *b : all) release(*b);
    for (hipEvent_t e : c->ev) (void)hipEventDestroy(e);
    for (hipEvent_t e : c->ring.ev) if (e) (void)hipEventDestroy(e);
    if (c->ring.base) (void)hipHostFree(c->ring.base);
    if (c->pin_back.p) (void)hipHostFree(c->pin_back.p);
    if (c->own_stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

const char *lpf_last_error(const lpf_ctx *c) { return c ? c->err.c_str() : g_create_error.c_str(); }

int lpf_set_stream(lpf_ctx *c, void *s)
{
    if (!c) return LPF_ERR_ARG;
    if (use_device(c)) return LPF_ERR_HIP;
    { int rc_ = sync_all(c); if (rc_) return rc_; }
    if (c->own_stream) { (void)hipStreamDestroy(c->stream); c->own_stream = false; }
    c->stream = (hipStream_t)s;          // NULL is the null stream itself (torch's default stream has handle 0)
    ++c->generation;
    return LPF_OK;
}

int lpf_use_own_stream(lpf_ctx *c)
{
    if (!c) return LPF_ERR_ARG;
    if (use_device(c)) return LPF_ERR_HIP;
    { int rc_ = sync_all(c); if (rc_) return rc_; }
    if (c->own_stream) return LPF_OK;
    c->stream = nullptr;
    LPF_HIP(c, hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    c->own_stream = true;
    ++c->generation;
    return LPF_OK;
}

// ---- explicit ordering against the caller's other streams (no device-wide synchronisation) ---------------
int lpf_wait_for_stream(lpf_ctx *c, void *producer)
{
    if (!c) return LPF_ERR_ARG;
    if (use_device(c)) return LPF_ERR_HIP;
    if (c->capturing) return fail(c, LPF_ERR_STATE, "lpf_wait_for_stream inside graph capture");
    if ((hipStream_t)producer == c->stream) return LPF_OK;      // same stream: already ordered
    hipEvent_t e;
    LPF_HIP(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    hipError_t r = hipEventRecord(e, (hipStream_t)producer);
    if (r == hipSuccess) r = hipStreamWaitEvent(c->stream, e, 0);
    (void)hipEventDestroy(e);                                  // released once it has completed
    if (r != hipSuccess) return fail(c, LPF_ERR_HIP, "lpf_wait_for_stream: %s", hipGetErrorString(r));
    return LPF_OK;
}

int lpf_release_to_stream(lpf_ctx *c, void *consumer)
{
    if (!c) return LPF_ERR_ARG;
    if (use_device(c)) return LPF_ERR_HIP;
    if (c->capturing) return fail(c, LPF_ERR_STATE, "lpf_release_to_stream inside graph capture");
    { int rc_ = flush_pending(c); if (rc_) return rc_; }
    if (c->stream == (hipStream_t)consumer) return LPF_OK;
    hipEvent_t e;
    LPF_HIP(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    hipError_t r = hipEventRecord(e, c->stream);
    if (r == hipSuccess) r = hipStreamWaitEvent((hipStream_t)consumer, e, 0);
    (void)hipEventDestroy(e);
    if (r != hipSuccess) return fail(c, LPF_ERR_HIP, "lpf_release_to_stream: %s", hipGetErrorString(r));
    return LPF_OK;
}

int lpf_sync(lpf_ctx *c)
{
    if (!c) return LPF_ERR_ARG;
    if (use_device(c)) return LPF_ERR_HIP;
    return sync_all(c);
}

#ifdef LPF_LAB
// Lab builds only (liblpf_lab.so: tools/ and the forced-geometry fuzz tests): override the launch geometry a run would pick
// by its size.  0 = by launch size, 1 = small with the wide tail, 2 = large, 3 = large with the segment prefixes taken from
// the scan kernel, 4 = small with the narrow tail.  Results do not depend on it.
int lpf_set_geometry(lpf_ctx *c, int mode)
{
    if (!c) return LPF_ERR_ARG;
    if (mode < 0 || mode > 5) return fail(c, LPF_ERR_ARG, "lpf_set_geometry: mode=%d (0 by launch size, 1 small with the wide tail, 2 large, 3 large with scan-kernel prefixes, 4 small with the narrow tail, 5 small with 1024-point tiles)", mode);
    c->geometry = mode;
    ++c->generation;
    return LPF_OK;
}

// Role clock of the step launches (software-pipelined modes): out[6][5] = per role {first block start, last block end, sum of the
// block durations, blocks, longest block} in ticks of the 100 MHz wall clock, accumulated since the last reset.  The first call
// switches the clock on (the step kernel of a lab build then ends every block with a barrier and four atomics).  Synchronises.
int lpf_lab_role_clock(lpf_ctx *c, unsigned long long *out, int reset)
{
    if (!c) return LPF_ERR_ARG;
    if (use_device(c)) return LPF_ERR_HIP;
    int rc = sync_all(c);
    if (rc) return rc;
    unsigned long long init[30];
    for (int r = 0; r < 6; ++r) { init[5 * r] = ~0ull; init[5 * r + 1] = init[5 * r + 2] = init[5 * r + 3] = init[5 * r + 4] = 0ull; }
    if (!c->lab_clk.p) {
        if ((rc = reserve(c, c->lab_clk, sizeof init))) return rc;
        if (hipMemcpy(c->lab_clk.p, init, sizeof init, hipMemcpyHostToDevice) != hipSuccess) return fail(c, LPF_ERR_HIP, "lpf_lab_role_clock: hipMemcpy");
    }
    if (out && hipMemcpy(out, c->lab_clk.p, sizeof init, hipMemcpyDeviceToHost) != hipSuccess) return fail(c, LPF_ERR_HIP, "lpf_lab_role_clock: hipMemcpy");
    if (reset && hipMemcpy(c->lab_clk.p, init, sizeof init, hipMemcpyHostToDevice) != hipSuccess) return fail(c, LPF_ERR_HIP, "lpf_lab_role_clock: hipMemcpy");
    return LPF_OK;
}
#endif

// Rectangles for the masks of the NEXT lpf_set_masks_* call: rects[F][M] = {x0, y0, x1, y1}, half open, pixels; the caller's
// word that mask m of frame f is zero outside its rectangle (a detector's masks are cropped to their boxes).  A hint: where uint8
// masks are packed as they are (no erosion) the pack skips the 16-pixel groups that lie outside; every other form reads the masks in
// full.  Results are those without the hint as long as the caller's word holds.  on_device: 0 host memory (copied now, no wait),
// else device memory read when the masks are packed (it must stay unchanged until then, like lent masks).  NULL clears.
int lpf_set_mask_rects(lpf_ctx *c, const int32_t *rects, int on_device, int F, int M)
{
    if (!c) return LPF_ERR_ARG;
    c->rects_pending = nullptr; c->rects_F = 0; c->rects_M = 0;
    if (!rects || F <= 0 || M <= 0) return LPF_OK;
    if (M > LPF_MAX_MASKS) return fail(c, LPF_ERR_ARG, "lpf_set_mask_rects: M=%d (at most %d)", M, LPF_MAX_MASKS);
    if (use_device(c)) return LPF_ERR_HIP;
    const size_t bytes = (size_t)F * M * sizeof(int4);
    if (on_device) {
        if (((uintptr_t)rects & 15) != 0) return fail(c, LPF_ERR_ARG, "lpf_set_mask_rects: the device array must be 16-byte aligned");
        c->rects_pending = (const int4 *)rects;
    } else {
        if (c->capturing) return LPF_OK;                    // (a copy from host memory is not captured: the hint is dropped)
        // the copy lives in the scratch set of the run these masks are for: in the pipelined modes that run's tiles may read the
        // rectangles a launch after the NEXT run's have been uploaded (into the next set); stream-ordered behind the set's last run
        DevBuf &rb = c->sc[c->fused ? c->parity : 0].rects;
        int rc = reserve(c, rb, bytes);
        if (rc) return rc;
        if ((rc = upload(c, rb.p, rects, bytes))) return rc;
        c->rects_pending = (const int4 *)rb.p;
    }
    c->rects_F = F; c->rects_M = M;
    return LPF_OK;
}

int lpf_set_pipelined(lpf_ctx *c, int on)
{
    if (!c) return LPF_ERR_ARG;
    if (use_device(c)) return LPF_ERR_HIP;
    if (on == 1 || on == 3)
        return fail(c, LPF_ERR_ARG, "lpf_set_pipelined: modes 1 and 3 (tail kernels / mask pack on side streams) were removed in ABI 5 -- "
                                    "they measured slower than 2 and 4 on every workload (DESIGN.md section 8); use 2 or 4");
    if (on != 0 && on != 2 && on != 4) return fail(c, LPF_ERR_ARG, "lpf_set_pipelined: mode %d (0 off, 2 the tail rides in the next run's launch, "
                                                                   "4 = 2 + the mask pack rides as well)", on);
    int rc = sync_all(c);
    if (rc) return rc;
    if ((rc = ensure_packed(c))) return rc;
    if ((rc = pack_ride_now(c))) return rc;                // lent masks that were waiting for a run: packed now, where the next run looks
    c->fused = on != 0;
    c->defer = on == 4;
    c->parity = 0;
    if (c->mask_F && c->mask_set != 0) { c->mask_F = 0; c->mask_M = 0; }     // label images rotate with the sets: set the masks again
    ++c->generation;
    return LPF_OK;
}

int lpf_set_camera(lpf_ctx *c, const double T[16], const double K[9], int W, int H, double dmin, double dmax)
{
    if (!c) return LPF_ERR_ARG;
    if (!T || !K || W <= 0 || H <= 0 || (long long)W * H > (1ll << 30))
        return fail(c, LPF_ERR_ARG, "set_camera: T=%p K=%p W=%d H=%d", (const void *)T, (const void *)K, W, H);
    // pipelined modes: a run still owed projects / counts boxes with the OLD camera (its box job)
    if (!c->capturing && (anything_owed(c) || c->bx[c->box_cur].job_valid)) {
        if (use_device(c)) return LPF_ERR_HIP;
        int rc_ = sync_all(c);
        if (rc_) return rc_;
    }
    if (W != c->W || H != c->H) {            // label images (and the visibility filter of cam-0 boxes) are per W x H: set masks / boxes again
        c->mask_F = 0; c->mask_M = 0; c->lazy.valid = false; c->ride.valid = false;
        for (auto &B : c->bx) { B.F = 0; B.box_off.clear(); B.job_valid = false; }
    }
    memcpy(c->T, T, sizeof c->T);          // row 3 of the 4x4 is never used by the reference either (V3:567 [:, :3])
    memcpy(c->K, K, sizeof c->K);
    c->W = W; c->H = H; c->dmin = dmin; c->dmax = dmax;
    c->have_camera = true;
    c->cand_dirty = true;                    // a cam-0 box job filters with K, W, H: the current set's tables are rebuilt
    ++c->generation;                         // T, K, W, H are kernel arguments of a captured graph
    return LPF_OK;
}

int lpf_set_masks_u8(lpf_ctx *c, const uint8_t *masks, int F, int M, int erode_iters, int on_device)
{
    return set_masks_impl<uint8_t>(c, masks, F, M, 0, erode_iters, on_device);
}

int lpf_set_masks_f32(lpf_ctx *c, const float *masks, int F, int M, int binarize, int erode_iters, int on_device)
{
    if (c && (binarize < 0 || binarize > 2)) return fail(c, LPF_ERR_ARG, "lpf_set_masks_f32: binarize=%d (0, 1 or 2)", binarize);
    return set_masks_impl<float>(c, masks, F, M, binarize + 1, erode_iters, on_device);
}

int lpf_set_label_image(lpf_ctx *c, const uint32_t *label, int F, int M, int on_device)
{
    if (!c) return LPF_ERR_ARG;
    if (use_device(c)) return LPF_ERR_HIP;
    if (!c->have_camera) return fail(c, LPF_ERR_STATE, "lpf_set_camera must be called first");
    if (F < 0 || M < 0 || M > LPF_MAX_MASKS || (F > 0 && !label)) return fail(c, LPF_ERR_ARG, "set_label_image: F=%d M=%d", F, M);
    int rc;
    if ((rc = sync_all(c))) return rc;
    lpf_ctx::Scratch &S = c->sc[c->fused ? c->parity : 0];
    c->mask_F = 0; c->mask_M = 0; S.label_cur = nullptr; c->lazy.valid = false;
    c->mask_set = c->fused ? c->parity : 0;
    if (F == 0) return LPF_OK;
    const size_t bytes = (size_t)F * c->H * c->W * 4;
    if ((rc = reserve(c, S.label_a, bytes))) return rc;
    LPF_HIP(c, hipMemcpyAsync(S.label_a.p, label, bytes, on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, c->stream));
    LPF_HIP(c, host_wait(c));
    S.label_cur = S.label_a.p;
    S.label_bytes = 4;
    c->mask_F = F; c->mask_M = M;
    return LPF_OK;
}

int lpf_get_label_image(lpf_ctx *c, uint32_t *out, int on_device)
{
    if (!c || !out) return LPF_ERR_ARG;
    if (use_device(c)) return LPF_ERR_HIP;
    { int rc_ = sync_all(c); if (rc_) return rc_; }
    { int rc_ = ensure_packed(c); if (rc_) return rc_; }
    { int rc_ = pack_ride_now(c); if (rc_) return rc_; }
    lpf_ctx::Scratch &S = c->sc[c->mask_set];
    if (!S.label_cur || !c->mask_F) return fail(c, LPF_ERR_STATE, "no masks set");
    const size_t npix = (size_t)c->mask_F * c->H * c->W;
    if (S.label_bytes == 4) {
        LPF_HIP(c, hipMemcpyAsync(out, S.label_cur, npix * 4, on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, c->stream));
        LPF_HIP(c, host_wait(c));
        return LPF_OK;
    }
    // narrow label images (M <= 16) are widened on the host: this entry point exists for tests
    std::vector<uint8_t> raw(npix * (size_t)S.label_bytes);
    LPF_HIP(c, hipMemcpyAsync(raw.data(), S.label_cur, raw.size(), hipMemcpyDeviceToHost, c->stream));
    LPF_HIP(c, host_wait(c));
    std::vector<uint32_t> wide(npix);
    for (size_t i = 0; i < npix; ++i)
        wide[i] = S.label_bytes == 1 ? (uint32_t)raw[i] : (uint32_t)reinterpret_cast<const uint16_t *>(raw.data())[i];
    LPF_HIP(c, hipMemcpyAsync(out, wide.data(), npix * 4, on_device ? hipMemcpyHostToDevice : hipMemcpyHostToHost, c->stream));
    LPF_HIP(c, host_wait(c));
    return LPF_OK;
}

int lpf_set_boxes(lpf_ctx *c, const double *corners, const int32_t *box_off, int F, int oriented)
{
    return lpf_set_boxes_ex(c, corners, 0, box_off, F, oriented);
}

int lpf_set_boxes_ex(lpf_ctx *c, const double *corners, int on_device, const int32_t *box_off, int F, int oriented)
{
    return set_boxes_impl(c, corners, on_device, box_off, F, oriented, false, nullptr, 0, nullptr, nullptr, nullptr, nullptr, "set_boxes");
}

int lpf_set_boxes_cam0(lpf_ctx *c, const double *corners_cam0, int on_device, const int32_t *box_off, int F, const double Tcv[16],
                       int filter_visible, int oriented, uint8_t *visible, double *corners_velo, double *bbox2d, int32_t *front)
{
    return set_boxes_impl(c, corners_cam0, on_device, box_off, F, oriented, true, Tcv, filter_visible, visible, corners_velo, bbox2d, front,
                          "set_boxes_cam0");
}

int lpf_run_batch(lpf_ctx *c, const float *pts, const int64_t *frame_off, int F, int pts_on_device, const lpf_outputs *out)
{
    if (!c) return LPF_ERR_ARG;
    if (use_device(c)) return LPF_ERR_HIP;
    if (!c->have_camera) return fail(c, LPF_ERR_STATE, "lpf_set_camera has not been called");
    if (!out || !frame_off || F <= 0) return fail(c, LPF_ERR_ARG, "run: out=%p frame_off=%p F=%d", (const void *)out, (const void *)frame_off, F);
    if (frame_off[0] != 0) return fail(c, LPF_ERR_ARG, "run: frame_off[0] must be 0");
    for (int f = 0; f < F; ++f) {
        const int64_t n = frame_off[f + 1] - frame_off[f];
        if (n < 0 || n > 0x7fffffffll - LPF_SEG_QUANTUM) return fail(c, LPF_ERR_ARG, "run: frame %d has %lld points", f, (long long)n);
    }
    const int64_t Ntot = frame_off[F];
    if (Ntot > 0 && !pts) return fail(c, LPF_ERR_ARG, "run: pts is NULL");
    lpf_ctx::BoxSet &BX = c->bx[c->box_cur];
    if (c->mask_F != 0 && c->mask_F != F) return fail(c, LPF_ERR_STATE, "masks were set for %d frames, run has %d", c->mask_F, F);
    if (BX.F != 0 && BX.F != F) return fail(c, LPF_ERR_STATE, "boxes were set for %d frames, run has %d", BX.F, F);
    if (out->inst_idx && out->inst_cap <= 0) return fail(c, LPF_ERR_ARG, "run: inst_idx given with inst_cap=%lld", (long long)out->inst_cap);
    if ((out->uv_valid || out->label_valid) && !out->valid_idx)
        return fail(c, LPF_ERR_ARG, "run: uv_valid / label_valid need valid_idx as well (they share its order)");
    const int M = c->mask_F ? c->mask_M : 0;
    const int Btot = BX.F ? BX.box_off[F] : 0;
    const bool host_io = !out->on_device;
    int rc;

    // ---- segmentation: segments of 4096 points (1024 for small launches), one list wave each; K1 tiles subdivide them;
    //      groups of 64 segments are the second level of the counters ------------------------------------------------
    const bool small = c->geometry == 1 || c->geometry == 4 || c->geometry == 5 || (c->geometry == 0 && Ntot <= LPF_SMALL_LAUNCH);
    const int64_t seg_pts = small ? LPF_SEG_SMALL : LPF_SEG_QUANTUM;
    c->h_frames.resize(F);
    int nseg_total = 0, ngrp_total = 0, max_ngrp = 0;
    for (int f = 0; f < F; ++f) {
        LpfFrame &fr = c->h_frames[f];
        fr.pt_off = frame_off[f];
        fr.N = (int)(frame_off[f + 1] - frame_off[f]);
        fr.seg_off = nseg_total;
        fr.shift = (int)(frame_off[f] & 63);               // the frame's rows start at the 64-point boundary below its first point (LpfFrame)
        fr.nseg = fr.N ? (int)((fr.N + fr.shift + seg_pts - 1) / seg_pts) : 0;
        nseg_total += fr.nseg;
        fr.box_off = BX.F ? BX.box_off[f] : 0;
        fr.B = BX.F ? BX.box_off[f + 1] - BX.box_off[f] : 0;
        fr.inst_base = (long long)f * out->inst_cap;
        fr.pad = f;
        fr.cand_off = BX.F ? BX.cand_off[f] : 0;
        fr.cand_words = (fr.B + 63) / 64;
        fr.grp_off = ngrp_total;
        fr.pad4 = 0;
        const int ngrp = (fr.nseg + LPF_GROUP_SEGS - 1) / LPF_GROUP_SEGS;
        ngrp_total += ngrp;
        if (ngrp > max_ngrp) max_ngrp = ngrp;
    }
    const int nseg_cap = nseg_total > 0 ? nseg_total : 1, ngrp_cap = ngrp_total > 0 ? ngrp_total : 1;
    // tail blocks: four consecutive segments of one frame each (an empty frame still gets one, to write its summary);
    // the list blocks, then -- when boxes are to be counted -- as many box-count blocks
    const bool count_boxes = M > 0 && Btot > 0;
    // (fused is decided further down; the same condition here)
    int nblk_early = 0;
    for (int f = 0; f < F; ++f) nblk_early += c->h_frames[f].nseg > 0 ? (c->h_frames[f].nseg + LPF_LISTS_WAVES - 1) / LPF_LISTS_WAVES : 1;
    // a frame or two of a real scan (up to 64 list blocks = 262 144 points): the launch is as long as its longest block, so the count
    // blocks are cut in four and the lists use the 16-row wave.  Beyond that both cost more than they save (same box, us per step in a
    // pipelined stream with / without: 20 real frames 29.8 / 25.5 with the cut, 25.2 / 25.5 with the list wave; one 2 M-point
    // synthetic cloud 25.7 / 21.3 and 22.9 / 21.3, in order 42.4 / 38.9 with the list wave).
    const bool few = small && nblk_early <= LPF_FEW_BLOCKS;
    const int csplit = (c->fused && !host_io && pts_on_device && !c->capturing && few) ? 4 : 1;     // count blocks per (group, word): see lpf_tail_block
    int nblk = 0, ncblk = 0;                               // list blocks; box-count blocks: one per group of segments, 64-box word and part
    for (int f = 0; f < F; ++f) {
        const int nb = c->h_frames[f].nseg > 0 ? (c->h_frames[f].nseg + LPF_LISTS_WAVES - 1) / LPF_LISTS_WAVES : 1;
        nblk += nb;
        ncblk += nb * std::max(1, c->h_frames[f].cand_words) * csplit;
    }
    const size_t rows = (size_t)nseg_cap * (size_t)(seg_pts / 64);
    // a list wave sums one group's segments and the frame's groups, a lane each: frames of more than 64 groups
    // (16.7 M points) take their prefixes from the scan kernel instead
    const bool pre_scan = max_ngrp > 64 || c->geometry == 3;

    // software-pipelined device runs rotate through the scratch sets (3 in mode 2, 4 in mode 4); everything else uses set 0 with
    // nothing owed
    const bool fused = c->fused && !host_io && pts_on_device && !c->capturing;   // one launch per run, the tail rides in the next
    if (!fused && anything_owed(c) && (rc = sync_all(c))) return rc;
    lpf_ctx::Scratch &S = c->sc[fused ? c->parity : 0];
    if ((rc = reserve(c, S.vbal, rows * 8))) return rc;
    if ((rc = reserve(c, S.mbal, rows * 8))) return rc;
    if ((rc = reserve(c, S.seg_tab, (size_t)LPF_TAB_GROUPS * nseg_cap * sizeof(uint4), true))) return rc;
    if ((rc = reserve(c, S.grp_tab, (size_t)LPF_TAB_GROUPS * ngrp_cap * sizeof(uint4), true))) return rc;
    if ((rc = reserve(c, S.frm_tab, (size_t)F * LPF_FRM_SHARDS * LPF_TAB_GROUPS * sizeof(uint4), true))) return rc;
    if (pre_scan && (rc = reserve(c, S.seg_pre, (size_t)LPF_TAB_GROUPS * nseg_cap * sizeof(uint4)))) return rc;
    if ((rc = reserve(c, S.cnt, (size_t)(M > 0 ? M : 1) * (Btot > 0 ? Btot : 1) * 4, true))) return rc;

    // ---- geometry tables of the run, in its own scratch set: the frame records, the owning frame's record per segment, the
    //      tail block table.  A single frame needs none of them (record by value, block table computed).  They only change
    //      when the batch shape does, and then travel through the pinned ring: no wait, no drain ---------------------------
    const bool have_tab = S.tab_frames.size() == (size_t)F && memcmp(S.tab_frames.data(), c->h_frames.data(), (size_t)F * sizeof(LpfFrame)) == 0;
    if (F > 1 && !have_tab) {
        const size_t b_frames = (size_t)F * sizeof(LpfFrame), b_segs = (size_t)nseg_total * sizeof(LpfFrame), b_blks = (size_t)nblk * sizeof(int2),
                     b_cblks = (size_t)ncblk * sizeof(int4);
        if ((rc = reserve(c, S.tab, b_frames + b_segs + b_blks + b_cblks))) return rc;
        c->h_tab.resize(b_frames + b_segs + b_blks + b_cblks);
        memcpy(c->h_tab.data(), c->h_frames.data(), b_frames);
        LpfFrame *hs = reinterpret_cast<LpfFrame *>(c->h_tab.data() + b_frames);
        int2 *hb = reinterpret_cast<int2 *>(c->h_tab.data() + b_frames + b_segs);
        int4 *hc = reinterpret_cast<int4 *>(c->h_tab.data() + b_frames + b_segs + b_blks);
        for (int f = 0; f < F; ++f) {
            const LpfFrame &fr = c->h_frames[f];
            const int wpg = std::max(1, fr.cand_words);
            for (int sg = 0; sg < fr.nseg; ++sg) hs[(size_t)fr.seg_off + sg] = fr;
            if (fr.nseg == 0) {
                *hb++ = make_int2(fr.seg_off, f << 3);
                for (int w = 0; w < wpg; ++w) for (int r = 0; r < csplit; ++r) *hc++ = make_int4(fr.seg_off, f, w, r << 3);
            }
            for (int sg = 0; sg < fr.nseg; sg += LPF_LISTS_WAVES) {
                const int nw = std::min(LPF_LISTS_WAVES, fr.nseg - sg);
                *hb++ = make_int2(fr.seg_off + sg, (f << 3) | nw);
                for (int w = 0; w < wpg; ++w) for (int r = 0; r < csplit; ++r) *hc++ = make_int4(fr.seg_off + sg, f, w, (r << 3) | nw);
            }
        }
        S.tab_frames.clear();                              // (nothing valid if the upload fails half way)
        if ((rc = upload(c, S.tab.p, c->h_tab.data(), c->h_tab.size()))) return rc;
        S.tab_frames = c->h_frames;
        S.o_segs = b_frames; S.o_blks = b_frames + b_segs; S.o_cblks = b_frames + b_segs + b_blks;
        ++c->generation;                  // graphs captured for another geometry read these tables
    }

    LpfParams P;
    memset(&P, 0, sizeof P);
    memcpy(P.T, c->T, sizeof P.T);
    memcpy(P.K, c->K, sizeof P.K);
    P.dmin = c->dmin; P.dmax = c->dmax; P.W = c->W; P.H = c->H;
    P.F = F; P.M = M; P.seg_pts = (int)seg_pts; P.nseg_total = nseg_total; P.nseg_cap = nseg_cap; P.ngrp_cap = ngrp_cap;
    P.oriented = BX.oriented; P.inst_cap = out->inst_cap;
    P.frame0 = c->h_frames[0];
    if (F > 1) {
        P.frames = (const LpfFrame *)S.tab.p;
        P.segs = (const LpfFrame *)((const char *)S.tab.p + S.o_segs);
        P.blks = (const int2 *)((const char *)S.tab.p + S.o_blks);
        P.cblks = (const int4 *)((const char *)S.tab.p + S.o_cblks);
    }
    // lent masks of a pipelined context (lpf_ctx::Ride): a small fused launch reads them directly, a large one in mode 4 carries
    // their pack, anything else (mode 2, float masks, a host-memory run) packs them now
    // (directly: M gathers per valid point against M reads per pixel for the pack -- it pays while a frame has fewer points than
    //  half its image has pixels: a real scan, 110 k points on 530 k pixels, 29.9 -> 24.9 us per 20-frame batch in a pipelined
    //  stream; a synthetic 2 M-point cloud is better off with the pack riding, 21.8 vs 24.2 us)
    const bool sparse_frames = Ntot * 2 <= (int64_t)F * c->W * c->H;
    // ... and with the masks' rectangles (lpf_set_mask_rects) a launch of sparse frames of ANY size reads the masks inside them: no
    // pack, no label image (146 real frames per step: 176 -> 134 us).  Dense frames keep the pack: with 2 M points on 530 k pixels and
    // rectangles that cover a good part of the image every row meets a rectangle, and the exact test of such a row is a dependent round
    // trip (one 2 M-point cloud with 8 disk masks, tiles 18.5 -> 21 us, pipelined stream 21.4 -> 25.4 us per cloud).
    // Small launches stay with the tiles that read all M mask bytes of a valid point (LpfDirect) and gate them by the rectangles: the
    // candidate grid would be one more kernel in front of a launch that is as long as its chain (a single real frame in order 18.8 vs
    // 23.0 us; pipelined 11.1 vs 10.2, 4 frames 21.1 vs 21.4, 20 frames 35.0 vs 36.6).
    const bool direct_rect_fused = c->ride.valid && fused && M > 0 && sparse_frames && !small && c->ride.rects &&
                                   ((!c->ride.f32 && c->ride.mode == 0) || (c->ride.f32 && c->ride.mode == 1));
    const bool direct_fused = direct_rect_fused || (c->ride.valid && fused && small && sparse_frames && M > 0);
    const bool ride_pack = c->ride.valid && fused && !direct_fused && c->ride.can_ride && M > 0;
    if (c->ride.valid && !direct_fused && !ride_pack && (rc = pack_ride_now(c))) return rc;
    // masks left unpacked: a small serial launch reads them directly, anything else packs them now (same stream, ahead of K1)
    const bool direct_rect = M > 0 && c->lazy.valid && !fused && sparse_frames && !small && c->lazy.rects &&
                             ((!c->lazy.f32 && c->lazy.mode == 0) || (c->lazy.f32 && c->lazy.mode == 1));
    const bool direct = direct_rect || (M > 0 && c->lazy.valid && small && sparse_frames && !fused);
    if (M > 0 && c->lazy.valid && !direct && (rc = ensure_packed(c))) return rc;
    // The label image lives in the scratch set that was current when the masks were set.  A pipelined run must find it in its
    // own set (the sets rotate: masks are set before every run); any other run has nothing owed and reads it where it is.
    const lpf_ctx::Scratch &SM = c->sc[c->mask_set];
    if (M > 0 && fused && !direct && c->mask_set != c->parity)
        return fail(c, LPF_ERR_STATE, "the masks were set for another scratch set: in the pipelined modes the label images rotate with the scratch sets -- call lpf_set_masks_* before every lpf_run* (and after switching modes)");
    P.label_img = (M > 0) ? (direct ? c->lazy.p : direct_fused ? c->ride.masks : SM.label_cur) : nullptr;
    // (the rectangles also gate the tiles that read all M mask bytes: small sparse launches)
    P.rects = (direct && c->lazy.valid) ? c->lazy.rects : (direct_fused && c->ride.valid) ? c->ride.rects : nullptr;
    // ... whose tiles look a point's candidates up in a coarse grid of the rectangles (a few KB per frame), built once per run: by
    // blocks of this run's own launch where the tiles come a launch later (mode 4), else by a small kernel ahead of the tiles
    LpfRectJob RG;
    memset(&RG, 0, sizeof RG);
    const bool rect_tiles = direct_rect || direct_rect_fused;
    if (rect_tiles) {
        RG.rects = P.rects; RG.F = F; RG.M = M;
        RG.cw = (c->W + LPF_RG_CELL - 1) / LPF_RG_CELL; RG.ch = (c->H + LPF_RG_CELL - 1) / LPF_RG_CELL;
        RG.cells = RG.cw * RG.ch; RG.bpf = (RG.cells + LPF_BLOCK - 1) / LPF_BLOCK;
        if ((rc = reserve(c, S.rgrid, (size_t)F * RG.cells * sizeof(uint32_t)))) return rc;
        RG.grid = (uint32_t *)S.rgrid.p;
        P.rect_grid = RG.grid; P.rg_cw = RG.cw; P.rg_cells = RG.cells;
    }
    if (M > 0 && !P.label_img) return fail(c, LPF_ERR_STATE, "no masks for this run's scratch set: in the pipelined modes the label images rotate with the scratch sets -- call lpf_set_masks_* before every lpf_run* (and after switching modes)");
    P.boxp = (const double *)BX.boxp.p; P.boxq = (const float *)BX.boxq.p;
    P.cand = (const unsigned long long *)BX.cand.p;
    P.vbal = (unsigned long long *)S.vbal.p; P.mbal = (unsigned long long *)S.mbal.p;
    P.seg_tab = (uint4 *)S.seg_tab.p; P.grp_tab = (uint4 *)S.grp_tab.p; P.frm_tab = (uint4 *)S.frm_tab.p;
    P.seg_pre = pre_scan ? (uint4 *)S.seg_pre.p : nullptr;
    P.cnt = (unsigned *)S.cnt.p;
    P.nblk = nblk; P.ncblk = ncblk; P.csplit = csplit; P.lists_small = few ? 1 : 0; P.count_boxes = count_boxes ? 1 : 0;
    P.count_lazy = small ? 0 : 1;

    // ---- buffers: caller's HBM pointers, or internal staging for host callers -----------
    const size_t n = (size_t)Ntot;
    if (pts_on_device) {
        P.pts = (const float4 *)pts;
    } else {
        if ((rc = reserve(c, c->st_pts, n * 16))) return rc;
        if (n) LPF_HIP(c, hipMemcpyAsync(c->st_pts.p, pts, n * 16, hipMemcpyHostToDevice, c->stream));
        P.pts = (const float4 *)c->st_pts.p;
    }
#define LPF_OUTBUF(field, member, stage, bytes)                                   \
    if (out->member) {                                                            \
        if (host_io) {                                                            \
            if ((rc = reserve(c, c->stage, (bytes)))) return rc;                  \
            P.field = (decltype(P.field))c->stage.p;                              \
        } else {                                                                  \
            P.field = (decltype(P.field))out->member;                             \
        }                                                                         \
    }
    LPF_OUTBUF(uv, uv, st_uv, n * 8)
    LPF_OUTBUF(label_bits, label_bits, st_label, n * 4)
    // the compact copies are gathered from the dense arrays: keep those in internal buffers when the caller skips them
    if (out->uv_valid && !P.uv) { if ((rc = reserve(c, c->st_uv, n * 8))) return rc; P.uv = (decltype(P.uv))c->st_uv.p; }
    if (out->label_valid && !P.label_bits) { if ((rc = reserve(c, c->st_label, n * 4))) return rc; P.label_bits = (decltype(P.label_bits))c->st_label.p; }
    LPF_OUTBUF(uv_valid, uv_valid, st_uvv, n * 8)
    LPF_OUTBUF(label_valid, label_valid, st_labv, n * 4)
    LPF_OUTBUF(depth, depth, st_depth, n * 8)
    LPF_OUTBUF(uf, u_f, st_uf, n * 8)
    LPF_OUTBUF(vf, v_f, st_vf, n * 8)
    LPF_OUTBUF(valid_idx, valid_idx, st_valid, n * 8)
    LPF_OUTBUF(inst_idx, inst_idx, st_inst, (size_t)F * (size_t)out->inst_cap * 8)
    LPF_OUTBUF(count_out, count_mb, st_count, (size_t)(M > 0 ? M : 1) * (Btot > 0 ? Btot : 1) * 4)
    // the summary is always produced: host callers need it to size the list copies
    if (host_io || !out->summary) {
        if ((rc = reserve(c, c->st_summary, (size_t)F * sizeof(lpf_frame_summary)))) return rc;
        P.summary = c->st_summary.p;
    } else {
        P.summary = out->summary;
    }
#undef LPF_OUTBUF
    if (M > 0) {                         // K1 -> tail hand-off of the masked points (sparse writes into N slots)
        if ((rc = reserve(c, S.mlist, n * 16))) return rc;
        P.mlist = (float4 *)S.mlist.p;
    }

    // the box tables this run's tail counts into: rebuilt if the camera changed since they were made
    if (BX.F && c->cand_dirty && !BX.job_valid) box_rebuild_job(BX);
    c->cand_dirty = false;
    BX.used = true;
    BX.last_ref = c->run_seq++;

    // small clouds: 512-point tiles (more, shorter blocks); large batches: 1024-point tiles
    // Small geometry: 512-point tiles while they fit the chip's block slots in one round (256 CUs x 7 blocks: a real frame is 214
    // tiles, a launch of a few frames a few hundred), one 1024-point tile per segment beyond -- a single mid-size cloud in 512-point
    // tiles pays for a third, nearly empty round of blocks (one cloud per launch set, in order / pipelined, us: 1 M points 43.6 / 17.4
    // -> 38.7 / 14.7, 2 M 39.2 / 21.4 -> 37.6 / 20.5, 3 M 63.9 / 28.4 -> 45.5 / 26.8).  Not for tiles that read M mask values per
    // valid point (LpfDirect: 512-point form only; small sparse launches).  Lab geometry 5 forces the 1024-point form, 1 / 4 the other.
    const bool plain_direct = (direct && !direct_rect) || (direct_fused && !direct_rect_fused);
    const bool small_1024 = small && !plain_direct && (c->geometry == 5 || (c->geometry == 0 && Ntot > 1792ll * 512));
    P.tile_pts = small ? (small_1024 ? 1024 : 512) : 1024;
    // the fused launch shares the chip with the previous run's tail blocks: 2048-point tiles keep twice the loads in flight
    // per wave, so the streaming work holds its bandwidth on fewer resident blocks (measured: 104.9 vs 108.8 us per step)
    if (fused && !small) P.tile_pts = direct_rect_fused ? LPF_RECT_FUSED_TILE : 2048;
    const int nk1 = nseg_total * (int)(seg_pts / P.tile_pts);
    const int lb = (M > 0) ? SM.label_bytes : 4;
    const bool want_lists = out->valid_idx || out->inst_idx;
    const int ntail = (count_boxes ? ncblk : 0) + (want_lists ? nblk : 0);        // no lists wanted and no boxes: no tail blocks at all
    hipEvent_t e0 = nullptr, e1 = nullptr;
    // mode 4: the mask pack and the tiles of one launch share the label element type -- else the pipeline is drained first
    if (fused && c->defer && c->pend_k1.valid && ride_pack && (c->pend_k1.direct || c->pend_k1.lb != lb) && (rc = flush_pending(c))) return rc;
    // the candidate grid of the masks' rectangles: where this run's tiles are in this run's own launch(es) -- in order, mode 2 -- a
    // small kernel ahead of them; in mode 4 it rides in the launch below (the tiles come a launch later)
    if (rect_tiles && nk1 > 0 && !(fused && c->defer)) {
        hipLaunchKernelGGL(lpf_rect_grid_kernel, dim3((unsigned)(RG.F * RG.bpf)), dim3(LPF_BLOCK), 0, c->stream, RG);
        LPF_HIP(c, hipGetLastError());
    }
    // (profiling brackets the launches that carry streaming tiles: in mode 4 the first launch after a drain carries none)
    const bool carries_k1 = (fused && c->defer) ? (c->pend_k1.valid && c->pend_k1.nk1 > 0) : (nk1 > 0 || fused);
    if (carries_k1 && c->profiling && c->ev_used < (1u << 16)) {
        if (c->ev.size() < 2 * (c->ev_used + 1)) {
            hipEvent_t a, b;
            LPF_HIP(c, hipEventCreate(&a));
            LPF_HIP(c, hipEventCreate(&b));
            c->ev.push_back(a); c->ev.push_back(b);
        }
        e0 = c->ev[2 * c->ev_used]; e1 = c->ev[2 * c->ev_used + 1];
        ++c->ev_used;
        LPF_HIP(c, hipEventRecord(e0, c->stream));
    }
    if (fused) {
        // ---- one launch (lpf_step_t): K1 tiles, the tail blocks of the run before dealt out among them, the summaries of the
        //      run before that, and the box job of THIS run (its tables are read by this run's tail, one or two launches on).
        //      Mode 2: the tiles are this run's.  Mode 4: this run's MASK PACK rides as well (lent uint8 masks), the tiles are
        //      the previous run's -- everything one launch later, nothing left on the stream between two steps.
        lpf_ctx::Pending cur;
        cur.valid = true; cur.P = P; cur.pre = pre_scan; cur.ntail = ntail; cur.nk1 = nk1; cur.lb = lb; cur.small = small;
        cur.direct = direct_fused; cur.dsel = direct_rect_fused ? (c->ride.f32 ? 5 : 4) : c->ride.f32 ? c->ride.mode : 0;
        const lpf_ctx::Pending KK = c->defer ? c->pend_k1 : cur, Q = c->pend_tail, R = c->pend_fin;
        if ((rc = launch_step(c, KK, Q, R, ride_pack, lb, &BX, e1, (rect_tiles && c->defer) ? &RG : nullptr))) return rc;
        c->pend_fin = Q;                                   // its tail has just been launched: summaries in a later launch
        c->pend_tail = KK;
        if (c->defer) c->pend_k1 = cur;
        c->ride.valid = false;
        c->parity = (c->parity + 1) % (c->defer ? 4 : 3);
        return LPF_OK;
    }
    if (BX.job_valid && nk1 > 0 && BX.F > 0 && BX.box_off[BX.F] > 0) {
        // in order, with a box job waiting: the tiles and the job share one launch (lpf_step_t with those two roles), the tail
        // that reads the tables follows in the next -- the same number of launches as with boxes that never change
        lpf_ctx::Pending cur, none;
        cur.valid = true; cur.P = P; cur.pre = false; cur.ntail = 0; cur.nk1 = nk1; cur.lb = lb; cur.small = small;
        cur.direct = direct; cur.dsel = direct_rect ? (c->lazy.f32 ? 5 : 4) : c->lazy.f32 ? c->lazy.mode : 0;
        if ((rc = launch_step(c, cur, none, none, false, lb, &BX, e1))) return rc;
    } else {
    if ((rc = launch_box_job(c, BX))) return rc;           // (no tiles to ride with)
    if (nk1 > 0) {
        const dim3 g1((unsigned)nk1);
#define LPF_K1_LAUNCH(R, LT) hipLaunchKernelGGL((lpf_k1_project_t<R, LPF_K1_FLAGS, LT>), g1, dim3(LPF_BLOCK), 0, c->stream, P)
        if (direct_rect) {
            typedef LpfDirectRect<uint8_t, 0> R0; typedef LpfDirectRect<float, 1> R1;
            if (P.tile_pts == 512) { if (!c->lazy.f32) LPF_K1_LAUNCH(2, R0); else LPF_K1_LAUNCH(2, R1); }
            else       { if (!c->lazy.f32) LPF_K1_LAUNCH(4, R0); else LPF_K1_LAUNCH(4, R1); }
        } else if (direct) {
            typedef LpfDirect<uint8_t, 0> D0; typedef LpfDirect<float, 1> D1; typedef LpfDirect<float, 2> D2; typedef LpfDirect<float, 3> D3;
            if (!c->lazy.f32) LPF_K1_LAUNCH(2, D0);
            else if (c->lazy.mode == 1) LPF_K1_LAUNCH(2, D1);
            else if (c->lazy.mode == 2) LPF_K1_LAUNCH(2, D2);
            else LPF_K1_LAUNCH(2, D3);
        } else if (P.tile_pts == 512) { if (lb == 1) LPF_K1_LAUNCH(2, uint8_t); else if (lb == 2) LPF_K1_LAUNCH(2, uint16_t); else LPF_K1_LAUNCH(2, uint32_t); }
        else       { if (lb == 1) LPF_K1_LAUNCH(4, uint8_t); else if (lb == 2) LPF_K1_LAUNCH(4, uint16_t); else LPF_K1_LAUNCH(4, uint32_t); }
#undef LPF_K1_LAUNCH
        LPF_HIP(c, hipGetLastError());
        if (e1) LPF_HIP(c, hipEventRecord(e1, c->stream));
    }
    }
    // ---- the tail: lists and box counts in one launch (a wave per segment each, side by side), then the per-frame summaries ----
    if (pre_scan && nseg_total > 0) {
        hipLaunchKernelGGL(lpf_scan_segments, dim3(F), dim3(LPF_BLOCK), 0, c->stream, P);
        LPF_HIP(c, hipGetLastError());
    }
    if (ntail > 0) {
        if (small && count_boxes && c->geometry != 4 && (nblk <= LPF_WIDE_BELOW || c->geometry == 1)) {   // a frame or a few, dense real segments: the box-count blocks share their chunks over 16 waves
            if (pre_scan) hipLaunchKernelGGL((lpf_tail_wide_t<true>), dim3((unsigned)ntail), dim3(64 * LPF_WIDE_WAVES), 0, c->stream, P);
            else hipLaunchKernelGGL((lpf_tail_wide_t<false>), dim3((unsigned)ntail), dim3(64 * LPF_WIDE_WAVES), 0, c->stream, P);
        } else {
            launch_tail(c->stream, P, ntail, pre_scan);
        }
        LPF_HIP(c, hipGetLastError());
    }
    hipLaunchKernelGGL(lpf_finalize, dim3(F), dim3(LPF_BLOCK), 0, c->stream, P);
    LPF_HIP(c, hipGetLastError());

    if (host_io) {
#define LPF_D2H(member, field, bytes) \
    if (out->member && (bytes)) LPF_HIP(c, hipMemcpyAsync(out->member, P.field, (bytes), hipMemcpyDeviceToHost, c->stream));
        LPF_D2H(uv, uv, n * 8)
        LPF_D2H(label_bits, label_bits, n * 4)
        LPF_D2H(depth, depth, n * 8)
        LPF_D2H(u_f, uf, n * 8)
        LPF_D2H(v_f, vf, n * 8)
        // Result buffers in page-locked memory (lpf_host_alloc, hipHostMalloc): the filled parts of the compact results are written
        // there by a kernel that reads the lengths from the summaries on the device -- one launch and ONE host wait, where the copy
        // engine needs the summaries on the host first (a wait), four copies per frame and a second wait.
        {
            LpfToHost D;
            memset(&D, 0, sizeof D);
            bool r2h = out->summary != nullptr && (out->valid_idx || out->inst_idx);
            auto alias = [&](void *host, void **dev) {
                *dev = host ? device_alias_of_pinned(host) : nullptr;
                if (host && !*dev) r2h = false;
            };
            if (r2h) {
                alias(out->summary, &D.summary);
                alias(out->valid_idx, (void **)&D.valid_idx);
                alias(out->uv_valid, (void **)&D.uv_valid);
                alias(out->label_valid, (void **)&D.label_valid);
                alias(out->inst_idx, (void **)&D.inst_idx);
                alias(out->count_mb, (void **)&D.count_mb);
            }
            if (r2h) {
                D.inst_cap = out->inst_cap;
                D.n_count = out->count_mb ? M * Btot : 0;
                hipLaunchKernelGGL(lpf_results_to_host, dim3((unsigned)F * LPF_R2H_BLOCKS), dim3(LPF_BLOCK), 0, c->stream, P, D);
                LPF_HIP(c, hipGetLastError());
                LPF_HIP(c, host_wait(c));
                return LPF_OK;
            }
        }
        LPF_D2H(count_mb, count_out, (size_t)M * Btot * 4)
        // lists: fetch the summary first, then only the filled part of each list
        std::vector<lpf_frame_summary> hs((size_t)F);
        LPF_HIP(c, hipMemcpyAsync(hs.data(), P.summary, (size_t)F * sizeof(lpf_frame_summary), hipMemcpyDeviceToHost, c->stream));
        LPF_HIP(c, host_wait(c));
        for (int f = 0; f < F; ++f) {
            const size_t nv = (size_t)hs[f].n_valid;
            if (out->valid_idx && nv)
                LPF_HIP(c, hipMemcpyAsync(out->valid_idx + frame_off[f], P.valid_idx + frame_off[f], nv * 8,
                                          hipMemcpyDeviceToHost, c->stream));
            if (out->uv_valid && nv)
                LPF_HIP(c, hipMemcpyAsync(out->uv_valid + 2 * frame_off[f], P.uv_valid + frame_off[f], nv * 8,
                                          hipMemcpyDeviceToHost, c->stream));
            if (out->label_valid && nv)
                LPF_HIP(c, hipMemcpyAsync(out->label_valid + frame_off[f], P.label_valid + frame_off[f], nv * 4,
                                          hipMemcpyDeviceToHost, c->stream));
            int64_t tot = hs[f].inst_off[LPF_MAX_MASKS];
            if (tot > out->inst_cap) tot = out->inst_cap;
            if (out->inst_idx && tot > 0)
                LPF_HIP(c, hipMemcpyAsync(out->inst_idx + (size_t)f * out->inst_cap, P.inst_idx + (size_t)f * out->inst_cap,
                                          (size_t)tot * 8, hipMemcpyDeviceToHost, c->stream));
        }
        if (out->summary) memcpy(out->summary, hs.data(), (size_t)F * sizeof(lpf_frame_summary));
#undef LPF_D2H
        LPF_HIP(c, host_wait(c));
    }
    return LPF_OK;
}

int lpf_points_in_boxes(lpf_ctx *c, const float *pts, int64_t k, int stride, const double *corners, int B, int oriented,
                        uint8_t *inside, int on_device)
{
    if (!c) return LPF_ERR_ARG;
    if (use_device(c)) return LPF_ERR_HIP;
    if (k < 0 || B < 0 || (stride != 3 && stride != 4) || (k > 0 && !pts) || (B > 0 && !corners) || (k > 0 && B > 0 && !inside))
        return fail(c, LPF_ERR_ARG, "points_in_boxes: k=%lld B=%d stride=%d", (long long)k, B, stride);
    if (k == 0 || B == 0) return LPF_OK;
    std::vector<double> bp((size_t)B * 16);
    for (int b = 0; b < B; ++b) box_params(corners + (size_t)b * 24, oriented, bp.data() + (size_t)b * 16);
    int rc;
    if ((rc = reserve(c, c->pib_box, bp.size() * 8))) return rc;
    LPF_HIP(c, hipMemcpyAsync(c->pib_box.p, bp.data(), bp.size() * 8, hipMemcpyHostToDevice, c->stream));
    const float *d_pts = pts;
    uint8_t *d_out = inside;
    if (!on_device) {
        if ((rc = reserve(c, c->pib_pts, (size_t)k * stride * 4))) return rc;
        if ((rc = reserve(c, c->pib_out, (size_t)k * B))) return rc;
        LPF_HIP(c, hipMemcpyAsync(c->pib_pts.p, pts, (size_t)k * stride * 4, hipMemcpyHostToDevice, c->stream));
        d_pts = (const float *)c->pib_pts.p; d_out = (uint8_t *)c->pib_out.p;
    }
    hipLaunchKernelGGL(lpf_points_in_boxes_kernel, dim3((unsigned)((k + LPF_BLOCK - 1) / LPF_BLOCK)), dim3(LPF_BLOCK), 0, c->stream,
                       d_pts, (long long)k, stride, (const double *)c->pib_box.p, B, oriented ? 1 : 0, d_out);
    LPF_HIP(c, hipGetLastError());
    if (!on_device) LPF_HIP(c, hipMemcpyAsync(inside, d_out, (size_t)k * B, hipMemcpyDeviceToHost, c->stream));
    LPF_HIP(c, host_wait(c));           // bp is a local
    return LPF_OK;
}

// OpenCV's weight table of one axis (resize.cpp: the xofs / alpha loop of resizeGeneric_, 8-bit INTER_LINEAR): per destination index
// {source index, second source index, w0, w1}.  Every step is a separate IEEE operation (volatile: no contraction, no excess precision).
// `clamp`: the x axis -- resize() sets {index, fraction} to {0, 0} / {n_src - 1, 0} beyond the ends.  The y axis keeps the fraction
// and only clips the two row indices (resizeGeneric_Invoker: clip(sy + k, 0, h)), so an edge row is blended with itself under both
// weights, whose two truncating shifts are not those of a single weight of 2048.
static void resize_table(int n_dst, int n_src, bool clamp, std::vector<int4> &tab)
{
    tab.resize((size_t)n_dst);
    volatile double ratio = (double)n_dst / (double)n_src;
    volatile double scale = 1.0 / ratio;
    for (int d = 0; d < n_dst; ++d) {
        volatile double a = ((double)d + 0.5) * scale;
        volatile double b = a - 0.5;
        volatile float f = (float)b;
        int s = (int)floorf(f);
        volatile float fr = f - (float)s;
        if (clamp && s < 0) { s = 0; fr = 0.f; }
        if (clamp && s >= n_src - 1) { s = n_src - 1; fr = 0.f; }
        volatile float one_minus = 1.0f - fr;
        volatile float p0 = one_minus * 2048.0f, p1 = fr * 2048.0f;
        long w0 = lrintf(p0), w1 = lrintf(p1);              // round half to even (the default rounding mode), as cvRound
        w0 = w0 < -32768 ? -32768 : w0 > 32767 ? 32767 : w0;
        w1 = w1 < -32768 ? -32768 : w1 > 32767 ? 32767 : w1;
        const int s0 = s < 0 ? 0 : s > n_src - 1 ? n_src - 1 : s, s1 = s + 1 < 0 ? 0 : s + 1 > n_src - 1 ? n_src - 1 : s + 1;
        tab[(size_t)d] = make_int4(s0, s1, (int)w0, (int)w1);
    }
}

// n planes [h][w] of uint8 -> n planes at the camera's size [H][W], as cv2.resize(plane, (W, H)) does (INTER_LINEAR; V3:222).
int lpf_resize_masks_u8(lpf_ctx *c, const uint8_t *src, int n, int h, int w, uint8_t *dst, int on_device)
{
    if (!c) return LPF_ERR_ARG;
    if (use_device(c)) return LPF_ERR_HIP;
    if (!c->have_camera) return fail(c, LPF_ERR_STATE, "lpf_set_camera has not been called (the target size is the camera's)");
    if (n < 0 || h <= 0 || w <= 0 || (n > 0 && (!src || !dst)) || (long long)h * w > 0x7fffffffll || (long long)n * h * w > (1ll << 36) ||
        (long long)n * c->W * c->H > (1ll << 36))
        return fail(c, LPF_ERR_ARG, "resize_masks: n=%d h=%d w=%d src=%p dst=%p (a plane of at most 2^31 - 1 pixels -- the kernel indexes it with "
                                    "32-bit arithmetic -- and at most 2^36 pixels in all, in and out)", n, h, w, (const void *)src, (void *)dst);
    if (n == 0) return LPF_OK;
    const int W = c->W, H = c->H;
    const bool area2 = w == 2 * W && h == 2 * H;              // cv2.resize hands this one to INTER_AREA: the rounded mean of 2 x 2 pixels
    const size_t in_bytes = (size_t)n * h * w, out_bytes = (size_t)n * H * W;
    int rc;
    if (c->capturing) return fail(c, LPF_ERR_STATE, "resize_masks inside a graph capture");
    if (anything_owed(c) && (rc = sync_all(c))) return rc;                // (the staging buffers below may be in use by owed runs)
    const uint8_t *dS = src;
    uint8_t *dD = dst;
    const bool linear = !area2 && !(h == H && w == W);
    std::vector<int4> xt, yt;
    const size_t tab_bytes = ((size_t)W + (size_t)H) * sizeof(int4);
    if ((rc = reserve(c, c->resize_buf, tab_bytes + (on_device ? 0 : in_bytes + out_bytes)))) return rc;
    if (linear) {
        resize_table(W, w, true, xt);
        resize_table(H, h, false, yt);
        // the tables are host vectors of this call: they travel through the pinned ring (copied now, queued in stream order), so a
        // device-mode call returns without waiting for the GPU
        if ((rc = upload(c, c->resize_buf.p, xt.data(), (size_t)W * sizeof(int4)))) return rc;
        if ((rc = upload(c, (char *)c->resize_buf.p + (size_t)W * sizeof(int4), yt.data(), (size_t)H * sizeof(int4)))) return rc;
    }
    if (!on_device) {
        dS = (const uint8_t *)c->resize_buf.p + tab_bytes;
        dD = (uint8_t *)c->resize_buf.p + tab_bytes + in_bytes;
        LPF_HIP(c, hipMemcpyAsync((void *)dS, src, in_bytes, hipMemcpyHostToDevice, c->stream));
    }
    if (!linear && !area2) {
        LPF_HIP(c, hipMemcpyAsync(dD, dS, out_bytes, hipMemcpyDeviceToDevice, c->stream));       // cv2.resize to the same size copies
    } else if (area2) {
        const long long quads = (long long)n * H * ((W + 3) / 4);
        hipLaunchKernelGGL(lpf_resize_area2_u8_kernel, dim3((unsigned)((quads + LPF_BLOCK - 1) / LPF_BLOCK)), dim3(LPF_BLOCK), 0, c->stream,
                           dS, dD, W, H, quads);
        LPF_HIP(c, hipGetLastError());
    } else {
        const long long total = (long long)out_bytes;
        hipLaunchKernelGGL(lpf_resize_linear_u8_kernel, dim3((unsigned)((total + LPF_BLOCK - 1) / LPF_BLOCK)), dim3(LPF_BLOCK), 0, c->stream,
                           dS, dD, (const int4 *)c->resize_buf.p, (const int4 *)c->resize_buf.p + W, w, h, W, H, total);
        LPF_HIP(c, hipGetLastError());
    }
    if (!on_device) {
        LPF_HIP(c, hipMemcpyAsync(dst, dD, out_bytes, hipMemcpyDeviceToHost, c->stream));
        LPF_HIP(c, host_wait(c));
    }
    return LPF_OK;
}

// n planes [h][w] of uint8 values, eroded `iters` times with the 3x3 cross at their own size (V3:83-90): see include/lpf.h
int lpf_erode_masks_u8(lpf_ctx *c, const uint8_t *src, int n, int h, int w, int iters, uint8_t *dst, int on_device)
{
    if (!c) return LPF_ERR_ARG;
    if (use_device(c)) return LPF_ERR_HIP;
    if (n < 0 || h <= 0 || w <= 0 || iters < 0 || (n > 0 && (!src || !dst || src == dst)) || (long long)h * w > 0x7fffffffll || (long long)n * h * w > (1ll << 36))
        return fail(c, LPF_ERR_ARG, "erode_masks: n=%d h=%d w=%d iters=%d src=%p dst=%p (src != dst)", n, h, w, iters, (const void *)src, (void *)dst);
    if (n == 0) return LPF_OK;
    if (c->capturing) return fail(c, LPF_ERR_STATE, "erode_masks inside a graph capture");
    int rc;
    if (anything_owed(c) && (rc = sync_all(c))) return rc;                // (the staging buffer below may be in use by owed runs)
    const size_t bytes = (size_t)n * h * w;
    // device callers: dst and one scratch plane set ping-pong; host callers: three staging copies
    if ((rc = reserve(c, c->resize_buf, on_device ? bytes : 3 * bytes))) return rc;
    uint8_t *a = on_device ? dst : (uint8_t *)c->resize_buf.p, *b = on_device ? (uint8_t *)c->resize_buf.p : (uint8_t *)c->resize_buf.p + bytes;
    const uint8_t *cur = src;
    if (!on_device) {
        uint8_t *in = (uint8_t *)c->resize_buf.p + 2 * bytes;
        LPF_HIP(c, hipMemcpyAsync(in, src, bytes, hipMemcpyHostToDevice, c->stream));
        cur = in;
    }
    const long long total = (long long)bytes;
    const dim3 g((unsigned)((total + LPF_BLOCK - 1) / LPF_BLOCK));
    // the last iteration must land in `a` (dst for device callers): with an even count the first one goes to b
    uint8_t *out = (iters % 2 == 1) ? a : b;
    for (int it = 0; it < iters; ++it) {
        hipLaunchKernelGGL(lpf_erode_u8_kernel, g, dim3(LPF_BLOCK), 0, c->stream, cur, out, w, h, total);
        LPF_HIP(c, hipGetLastError());
        cur = out;
        out = (out == a) ? b : a;
    }
    if (iters == 0) LPF_HIP(c, hipMemcpyAsync(a, cur, bytes, hipMemcpyDeviceToDevice, c->stream));
    if (!on_device) {
        LPF_HIP(c, hipMemcpyAsync(dst, a, bytes, hipMemcpyDeviceToHost, c->stream));
        LPF_HIP(c, host_wait(c));
    }
    return LPF_OK;
}

int lpf_depth_image(lpf_ctx *c, const float *pts, int64_t N, int on_device, double *depth_img, int32_t *winner)
{
    if (!c) return LPF_ERR_ARG;
    if (use_device(c)) return LPF_ERR_HIP;
    if (!c->have_camera) return fail(c, LPF_ERR_STATE, "lpf_set_camera has not been called");
    if (N < 0 || N > 0x7ffffff0ll || (N > 0 && !pts) || !depth_img) return fail(c, LPF_ERR_ARG, "depth_image: N=%lld", (long long)N);
    const size_t hw = (size_t)c->W * c->H;
    int rc;
    if ((rc = reserve(c, c->dimg, hw * 12))) return rc;                 // f64 image + u32 winners
    double *dD = on_device ? depth_img : (double *)c->dimg.p;
    unsigned *dW = (unsigned *)((char *)c->dimg.p + hw * 8);
    const float *dP = pts;
    if (!on_device && N > 0) {
        if ((rc = reserve(c, c->st_pts, (size_t)N * 16))) return rc;
        LPF_HIP(c, hipMemcpyAsync(c->st_pts.p, pts, (size_t)N * 16, hipMemcpyHostToDevice, c->stream));
        dP = (const float *)c->st_pts.p;
    }
    LPF_HIP(c, hipMemsetAsync(dD, 0, hw * 8, c->stream));
    LPF_HIP(c, hipMemsetAsync(dW, 0, hw * 4, c->stream));
    if (N > 0) {
        LpfParams P;
        memset(&P, 0, sizeof P);
        memcpy(P.T, c->T, sizeof P.T); memcpy(P.K, c->K, sizeof P.K);
        P.dmin = c->dmin; P.dmax = c->dmax; P.W = c->W; P.H = c->H; P.pts = (const float4 *)dP;
        const dim3 g((unsigned)((N + LPF_BLOCK - 1) / LPF_BLOCK));
        hipLaunchKernelGGL((lpf_depth_image_kernel<0>), g, dim3(LPF_BLOCK), 0, c->stream, P, (int)N, dW, dD);
        hipLaunchKernelGGL((lpf_depth_image_kernel<1>), g, dim3(LPF_BLOCK), 0, c->stream, P, (int)N, dW, dD);
        LPF_HIP(c, hipGetLastError());
    }
    if (!on_device) {
        LPF_HIP(c, hipMemcpyAsync(depth_img, dD, hw * 8, hipMemcpyDeviceToHost, c->stream));
        if (winner) {
            LPF_HIP(c, hipMemcpyAsync(winner, dW, hw * 4, hipMemcpyDeviceToHost, c->stream));
            LPF_HIP(c, host_wait(c));
            for (size_t i = 0; i < hw; ++i) winner[i] -= 1;          // stored as index + 1, 0 = none
            return LPF_OK;
        }
        LPF_HIP(c, host_wait(c));
    } else if (winner) {
        return fail(c, LPF_ERR_ARG, "depth_image: the winner image is only returned to host callers");
    }
    return LPF_OK;
}

int lpf_prepare_boxes(lpf_ctx *c, const double *corners_cam0, int nbox, const double Tcv[16], uint8_t *visible, double *corners_velo,
                      double *bbox2d, int32_t *front)
{
    if (!c) return LPF_ERR_ARG;
    if (use_device(c)) return LPF_ERR_HIP;
    if (!c->have_camera) return fail(c, LPF_ERR_STATE, "lpf_set_camera has not been called");
    if (nbox < 0 || (nbox > 0 && (!corners_cam0 || !Tcv))) return fail(c, LPF_ERR_ARG, "prepare_boxes: nbox=%d", nbox);
    if (nbox == 0) return LPF_OK;
    const size_t nb = (size_t)nbox;
    const size_t o_in = 0, o_cv = o_in + nb * 192, o_bb = o_cv + nb * 192, o_fr = o_bb + nb * 32, o_vis = o_fr + nb * 4, total = o_vis + nb;
    int rc;
    if (c->capturing) return fail(c, LPF_ERR_STATE, "prepare_boxes inside a graph capture");
    if ((rc = reserve(c, c->boxprep, total))) return rc;
    char *base = (char *)c->boxprep.p;
    // in through the pinned ring, back through a page-locked landing area in ONE copy (the four results lie behind each other): a
    // pageable pointer on either side makes every hipMemcpyAsync a blocking staged copy -- five of them were 0.9 ms per frame of
    // process_frames, fifty times the kernel
    if ((rc = upload(c, base + o_in, corners_cam0, nb * 192))) return rc;
    const size_t back = total - o_cv;
    if (c->pin_back.cap < back) {
        if (c->pin_back.p) { LPF_HIP(c, hipStreamSynchronize(c->stream)); (void)hipHostFree(c->pin_back.p); c->pin_back.p = nullptr; c->pin_back.cap = 0; }
        const size_t cap = back + back / 2 + 4096;
        LPF_HIP(c, hipHostMalloc((void **)&c->pin_back.p, cap, hipHostMallocDefault));
        c->pin_back.cap = cap;
    }
    LpfBoxPrep A;
    memcpy(A.Tcv, Tcv, sizeof A.Tcv);
    memcpy(A.K, c->K, sizeof A.K);
    A.W = c->W; A.H = c->H;
    hipLaunchKernelGGL(lpf_box_prep_kernel, dim3((unsigned)((nb * 8 + LPF_BLOCK - 1) / LPF_BLOCK)), dim3(LPF_BLOCK), 0, c->stream, A,
                       (const double *)(base + o_in), nbox, (uint8_t *)(base + o_vis), (double *)(base + o_cv), (double *)(base + o_bb),
                       (int *)(base + o_fr));
    LPF_HIP(c, hipGetLastError());
    LPF_HIP(c, hipMemcpyAsync(c->pin_back.p, base + o_cv, back, hipMemcpyDeviceToHost, c->stream));
    LPF_HIP(c, host_wait(c));
    const char *h = c->pin_back.p - o_cv;                         // (the landing area starts at the corners)
    if (visible) memcpy(visible, h + o_vis, nb);
    if (corners_velo) memcpy(corners_velo, h + o_cv, nb * 192);
    if (bbox2d) memcpy(bbox2d, h + o_bb, nb * 32);
    if (front) memcpy(front, h + o_fr, nb * 4);
    return LPF_OK;
}

int lpf_graph_begin(lpf_ctx *c)
{
    if (!c) return LPF_ERR_ARG;
    if (use_device(c)) return LPF_ERR_HIP;
    if (c->fused) return fail(c, LPF_ERR_STATE, "graph capture needs pipelining off");
    if (c->capturing) return fail(c, LPF_ERR_STATE, "already capturing");
    int rc = sync_all(c);
    if (rc) return rc;
    LPF_HIP(c, hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
    c->capturing = true;
    return LPF_OK;
}

int lpf_graph_end(lpf_ctx *c, lpf_graph **out)
{
    if (!c || !out) return LPF_ERR_ARG;
    if (use_device(c)) return LPF_ERR_HIP;
    if (!c->capturing) return fail(c, LPF_ERR_STATE, "lpf_graph_end without lpf_graph_begin");
    { int rc_ = launch_box_job(c, c->bx[c->box_cur]); if (rc_) return rc_; }      // boxes set last in the capture, with no run behind them
    c->capturing = false;
    lpf_graph *g = new (std::nothrow) lpf_graph();
    if (!g) return fail(c, LPF_ERR_NOMEM, "out of host memory");
    hipError_t e = hipStreamEndCapture(c->stream, &g->graph);
    if (e == hipSuccess) e = hipGraphInstantiate(&g->exec, g->graph, nullptr, nullptr, 0);
    if (e != hipSuccess) {
        if (g->graph) (void)hipGraphDestroy(g->graph);
        delete g;
        return fail(c, LPF_ERR_HIP, "graph capture failed: %s (a call inside the capture allocated, copied from pageable memory or synchronised?)",
                    hipGetErrorString(e));
    }
    g->generation = c->generation;
    *out = g;
    return LPF_OK;
}

int lpf_graph_launch(lpf_ctx *c, lpf_graph *g)
{
    if (!c || !g || !g->exec) return LPF_ERR_ARG;
    if (use_device(c)) return LPF_ERR_HIP;
    if (g->generation != c->generation)
        return fail(c, LPF_ERR_STATE, "lpf_graph_launch: the graph is stale -- since its capture the context changed geometry, boxes, camera, "
                                      "stream or mode, or regrew a buffer the graph points into; capture it again");
    LPF_HIP(c, hipGraphLaunch(g->exec, c->stream));
    return LPF_OK;
}

void lpf_graph_destroy(lpf_graph *g)
{
    if (!g) return;
    if (g->exec) (void)hipGraphExecDestroy(g->exec);
    if (g->graph) (void)hipGraphDestroy(g->graph);
    delete g;
}

int lpf_get_stats(lpf_ctx *c, int64_t *out, int n, int reset)
{
    if (!c || (n > 0 && !out) || n < 0) return LPF_ERR_ARG;
    for (int i = 0; i < n; ++i) out[i] = i < 8 ? (int64_t)c->stats[i] : 0;
    if (reset) for (long long &v : c->stats) v = 0;
    return LPF_OK;
}

int lpf_profile_enable(lpf_ctx *c, int on)
{
    if (!c) return LPF_ERR_ARG;
    c->profiling = on != 0;
    return LPF_OK;
}

// ---- the one exchange step of the sharded path: a small int64 all-reduce over RCCL -----------------------
// librccl is resolved at the first call (dlopen), so liblpf.so carries no link-time dependency on it and a
// single-GPU user never loads it.  The communicator is the caller's (ncclCommInitRank / ncclCommInitAll).
int lpf_allreduce_metrics(lpf_ctx *c, int64_t *vec, int n, int op, void *rccl_comm)
{
    if (!c) return LPF_ERR_ARG;
    if (use_device(c)) return LPF_ERR_HIP;
    if (!vec || n <= 0 || n > (1 << 20) || op < 0 || op > 2 || !rccl_comm)
        return fail(c, LPF_ERR_ARG, "allreduce_metrics: vec=%p n=%d op=%d (0 sum, 1 min, 2 max) comm=%p", (void *)vec, n, op, rccl_comm);
    if (c->capturing) return fail(c, LPF_ERR_STATE, "lpf_allreduce_metrics inside graph capture");
    typedef int (*allreduce_fn)(const void *, void *, size_t, int, int, void *, hipStream_t);
    typedef const char *(*errstr_fn)(int);
    // The communicator belongs to whichever RCCL created it, and a process may hold more than one copy (torch wheels
    // bundle their own): take ncclAllReduce from a library that is ALREADY loaded -- the global scope first, then the
    // usual names without loading anything -- and only then load the system's librccl.  Resolved once (thread-safe).
    static allreduce_fn p_allreduce = nullptr;
    static errstr_fn p_errstr = nullptr;
    static std::once_flag once;
    std::call_once(once, [] {
        void *h = nullptr;
        void *sym = dlsym(RTLD_DEFAULT, "ncclAllReduce");
        if (!sym) {
            for (const char *name : {"librccl.so", "librccl.so.1"}) {
                if ((h = dlopen(name, RTLD_NOW | RTLD_NOLOAD))) break;
            }
            if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
            if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
            if (h) sym = dlsym(h, "ncclAllReduce");
        }
        p_allreduce = (allreduce_fn)sym;
        p_errstr = (errstr_fn)(h ? dlsym(h, "ncclGetErrorString") : dlsym(RTLD_DEFAULT, "ncclGetErrorString"));
    });
    if (!p_allreduce) return fail(c, LPF_ERR_STATE, "allreduce_metrics: no ncclAllReduce in this process and librccl.so cannot be loaded (%s)", dlerror());
    int rc;
    if ((rc = reserve(c, c->coll, (size_t)n * 8))) return rc;                 // a buffer of its own (lpf_points_in_boxes uses pib_out)
    LPF_HIP(c, hipMemcpyAsync(c->coll.p, vec, (size_t)n * 8, hipMemcpyHostToDevice, c->stream));
    static const int red[3] = {0 /* ncclSum */, 3 /* ncclMin */, 2 /* ncclMax */};
    const int nrc = p_allreduce(c->coll.p, c->coll.p, (size_t)n, 4 /* ncclInt64 */, red[op], rccl_comm, c->stream);
    if (nrc != 0) return fail(c, LPF_ERR_HIP, "ncclAllReduce failed: %s", p_errstr ? p_errstr(nrc) : "?");
    LPF_HIP(c, hipMemcpyAsync(vec, c->coll.p, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream));
    LPF_HIP(c, host_wait(c));          // blocks, with no timeout, until every rank has joined the collective
    return LPF_OK;
}

int lpf_profile_overhead(lpf_ctx *c, double *empty_bracket_ms)
{
    if (!c) return LPF_ERR_ARG;
    if (use_device(c)) return LPF_ERR_HIP;
    if (c->capturing) return fail(c, LPF_ERR_STATE, "lpf_profile_overhead inside graph capture");
    // What a pair of event records costs with nothing between them, on this stream, right now: the bracket
    // around a kernel contains this much that is not the kernel (median of 33 pairs, each after a sync).
    hipEvent_t e0, e1;
    LPF_HIP(c, hipEventCreate(&e0));
    LPF_HIP(c, hipEventCreate(&e1));
    float v[33];
    for (int i = 0; i < 33; ++i) {
        LPF_HIP(c, hipEventRecord(e0, c->stream));
        LPF_HIP(c, hipEventRecord(e1, c->stream));
        LPF_HIP(c, host_wait(c));
        LPF_HIP(c, hipEventElapsedTime(&v[i], e0, e1));
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    std::sort(v, v + 33);
    if (empty_bracket_ms) *empty_bracket_ms = (double)v[16];
    return LPF_OK;
}

int lpf_profile_read(lpf_ctx *c, double *k1_ms_sum, int64_t *k1_launches, int reset)
{
    if (!c) return LPF_ERR_ARG;
    if (use_device(c)) return LPF_ERR_HIP;
    LPF_HIP(c, host_wait(c));
    double sum = 0.0;
    for (size_t i = 0; i < c->ev_used; ++i) {
        float ms = 0.f;
        LPF_HIP(c, hipEventElapsedTime(&ms, c->ev[2 * i], c->ev[2 * i + 1]));
        sum += ms;
    }
    if (k1_ms_sum) *k1_ms_sum = sum;
    if (k1_launches) *k1_launches = (int64_t)c->ev_used;
    if (reset) c->ev_used = 0;
    return LPF_OK;
}

int lpf_run(lpf_ctx *c, const float *pts, int64_t N, int pts_on_device, const lpf_outputs *out)
{
    const int64_t off[2] = {0, N};
    return lpf_run_batch(c, pts, off, 1, pts_on_device, out);
}

// one frame of a stream in one call: masks (+ rectangles), boxes, run (include/lpf.h)
int lpf_run_frame(lpf_ctx *c, const lpf_frame_job *j)
{
    if (!c) return LPF_ERR_ARG;
    if (!j) return fail(c, LPF_ERR_ARG, "lpf_run_frame: job is NULL");
    if (!j->out.on_device) return fail(c, LPF_ERR_ARG, "lpf_run_frame: device-mode outputs only (out.on_device = 1)");
    if (j->n_masks < 0 || j->n_boxes < 0) return fail(c, LPF_ERR_ARG, "lpf_run_frame: n_masks=%d n_boxes=%d", j->n_masks, j->n_boxes);
    int rc;
    if (j->masks) {
        if (j->mask_rects && (rc = lpf_set_mask_rects(c, j->mask_rects, 1, 1, j->n_masks))) return rc;
        if ((rc = lpf_set_masks_u8(c, j->masks, 1, j->n_masks, 0, 2))) return rc;
    }
    if (j->corners_cam0) {
        const int32_t boff[2] = {0, j->n_boxes};
        if ((rc = lpf_set_boxes_cam0(c, j->corners_cam0, 2, boff, 1, j->T_cam_to_velo, j->filter_visible, j->oriented, nullptr, nullptr, nullptr, nullptr)))
            return rc;
    }
    return lpf_run(c, j->pts, j->n_points, 1, &j->out);
}

}  // extern "C"

#include "lpf_reader.hip.h"
