/*
 * lpf.h -- C ABI of the MI355X LiDAR projection + instance point-filter library
 * (liblpf.so, built from lidar_object_detection_amd/csrc/).
 *
 * The reference (KaranSankla/Lidar_Object_Detection) has no FFI: its hot path is a
 * run of inline NumPy statements and small Python functions inside each script's
 * frame loop.  This header is the narrowest data cut that contains that path; each
 * entry point names the reference statements it replaces (paths relative to
 * /root/reference/Coding_testes).  INTEGRATION.md shows the ctypes stub a reference
 * maintainer would add.
 *
 * Conventions
 *   - plain C, caller-owned buffers, no exceptions: every call returns LPF_OK (0) or
 *     a negative lpf_status; lpf_last_error() gives the text.
 *   - a context is bound to one GPU and one HIP stream; it is not thread-safe.
 *     One context per GPU / rank.
 *   - "on_device" flags say whether the pointers of that call are device (HBM)
 *     pointers.  With device outputs lpf_run* only enqueues work on the context's
 *     stream and returns; call lpf_sync() (or synchronise the stream you attached
 *     with lpf_set_stream) before reading.  With host pointers the call stages
 *     through internal HBM buffers and is synchronous.
 *   - a batch is F frames that share the camera; points of all frames are
 *     concatenated, frame f owns points [frame_off[f], frame_off[f+1]).
 */
#ifndef LPF_H
#define LPF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LPF_MAX_MASKS 32          /* instances per frame: one bit each in label_bits */
#define LPF_ABI_VERSION 8          /* 2: lpf_outputs gained uv_valid / label_valid
                                      3: lpf_set_stream(ctx, NULL) is the null stream (was: an internal stream -> lpf_use_own_stream);
                                         lpf_set_pipelined modes; stale graphs are refused
                                      4: on_device = 2 (lent masks) in lpf_set_masks_*; lpf_set_pipelined(4)
                                      5: boxes per run in the software-pipelined modes (a ring of box sets, their preparation rides in
                                         the run's launch; on_device = 2 = lent corners in lpf_set_boxes_ex / lpf_set_boxes_cam0);
                                         geometry tables per run (a new batch shape no longer drains the pipeline);
                                         lpf_set_pipelined modes 1 / 3 and lpf_set_cu_partition removed (measured slower, DESIGN.md
                                         section 8); lpf_set_geometry only in lab builds (-DLPF_LAB)
                                      6: lpf_set_mask_rects, lpf_resize_masks_u8 (added; nothing else changed)
                                      7: lpf_build_id, lpf_host_alloc / lpf_host_free, lpf_run_frame, lpf_erode_masks_u8 (added; nothing else changed)
                                      8: lpf_reader_submit_frame, lpf_reader_boxes, lpf_parse_boxes_json (added); lpf_resize_masks_u8 no longer
                                         refuses an exact halving (it is cv2.resize's INTER_AREA case) */

typedef enum lpf_status {
    LPF_OK = 0,
    LPF_ERR_ARG = -1,             /* bad argument (null, negative, M > 32, ...) */
    LPF_ERR_HIP = -2,             /* a HIP runtime call failed */
    LPF_ERR_STATE = -3,           /* camera / masks / boxes not set for this call */
    LPF_ERR_NOMEM = -4,           /* device or host allocation failed */
    LPF_ERR_IO = -5               /* a scan file is missing, truncated or not [N][4] float32 (lpf_reader_*) */
} lpf_status;

typedef struct lpf_ctx lpf_ctx;

/* Per-frame scalar results (host or device memory, see lpf_outputs.on_device). */
typedef struct lpf_frame_summary {
    int64_t n_valid;                        /* len(valid_indices), V3:585 */
    int64_t n_labelled;                     /* points in >=1 mask == bg_assigned.sum(), V4:290-298 */
    int64_t inst_count[LPF_MAX_MASKS];      /* len(car_point_sets[m]), V3:227-231 */
    int64_t inst_off[LPF_MAX_MASKS + 1];    /* list m = inst_idx[inst_off[m] .. inst_off[m+1]) */
    int64_t best_cnt[LPF_MAX_MASKS];        /* best_match_count, V3:353-376 (0 if none) */
    int32_t best_box[LPF_MAX_MASKS];        /* best_bbox_idx into the frame's box list, -1 if none */
    int32_t inst_overflow;                  /* 1 if sum(inst_count) > inst_cap: lists truncated */
    int32_t reserved;
} lpf_frame_summary;

/* Output buffers of lpf_run / lpf_run_batch.  Any pointer may be NULL (not wanted).
 * Ntot = frame_off[F]; Btot = total boxes over the batch. */
typedef struct lpf_outputs {
    int32_t  *uv;          /* [Ntot][2]  (u, v) = np.round(x/|z|), np.round(y/|z|)  (V3:568-569),
                              saturated to int32 (only ever differs from the reference's int64
                              for |pixel| >= 2^31, which is never a valid point) */
    uint32_t *label_bits;  /* [Ntot]     bit m set <=> point valid and inside mask m (V3:225); 0 if !valid */
    double   *depth;       /* [Ntot]     signed depth incl. the 0 -> -1e-6 patch (cam2image) */
    double   *u_f;         /* [Ntot]     x/|z| before rounding */
    double   *v_f;         /* [Ntot]     y/|z| before rounding */
    int64_t  *valid_idx;   /* [Ntot]     frame f's np.where(valid)[0] at valid_idx[frame_off[f] ...],
                                          indices relative to the frame, ascending (V3:585) */
    int64_t  *inst_idx;    /* [F][inst_cap]  per frame: instance lists, concatenated in mask order,
                                          each ascending (== valid_indices[mask_indices], V3:225-228) */
    int64_t   inst_cap;    /*            capacity per frame of inst_idx (entries) */
    int32_t  *count_mb;    /* [M * Btot] frame f's [M][B_f] block at M*box_off[f]:
                                          np.sum(oriented_point_in_bbox(car_points_m, box_b)), V3:366-370 */
    lpf_frame_summary *summary;  /* [F] */
    int32_t   on_device;   /* 1: all pointers in this struct are device pointers, call is asynchronous */
    int32_t   reserved;
    int32_t  *uv_valid;    /* [Ntot][2]  compact: u_valid, v_valid = u[valid], v[valid] (V3:590-591) -- frame f's at
                              uv_valid[frame_off[f] ...], in valid_idx order, n_valid entries.  Needs valid_idx too. */
    uint32_t *label_valid; /* [Ntot]     compact: label_bits[valid_indices], same order */
} lpf_outputs;

/* ---- lifetime ---------------------------------------------------------------- */
int  lpf_abi_version(void);
/* Which sources this binary was compiled from: the first 16 hex digits of the SHA-256 over csrc/ + this header + the compiler flags,
 * baked in at build time (lidar_object_detection_amd/_build.py: source_id).  The loader compares it with the sources beside it and
 * rebuilds or refuses a stale library; bench.py prints it with every number.  "unknown" for a build made without the build script. */
const char *lpf_build_id(void);
/* Page-locked host memory (hipHostMalloc) for callers that are not torch programs: results copied into it are DMA transfers, and a
 * frame loop that reuses it does not allocate per call.  NULL on failure (lpf_last_error(NULL)). */
void *lpf_host_alloc(size_t bytes);
void  lpf_host_free(void *p);
int  lpf_create(lpf_ctx **out, int device_id);
void lpf_destroy(lpf_ctx *ctx);
const char *lpf_last_error(const lpf_ctx *ctx);      /* ctx may be NULL: error of the last failed lpf_create */
/* Run on a stream the caller owns, e.g. torch.cuda.current_stream().cuda_stream.  The handle is used as it is:
 * NULL (0) is the null stream -- which is what torch's default stream is -- so device-mode calls are ordered
 * with the caller's other work on that stream and need no device-wide synchronisation.  A new context runs on an
 * internal non-blocking stream; lpf_use_own_stream() goes back to one. */
int  lpf_set_stream(lpf_ctx *ctx, void *hip_stream);
int  lpf_use_own_stream(lpf_ctx *ctx);
/* Ordering contract of device mode.  Every device pointer handed to lpf_set_masks_* / lpf_run* is read or written by
 * kernels on the context's stream(s), in the order of the calls, and by nothing else.  A context that shares the
 * caller's stream (lpf_set_stream) is ordered with the caller's work by the stream itself.  A context on its own
 * stream is NOT: inputs produced on another stream, and -- with a stream-ordered caching allocator such as torch's
 * -- even output buffers, whose memory may still be in use by kernels queued earlier on the allocating stream, need an
 * edge first.  lpf_wait_for_stream makes the context's stream wait, on the device, for everything queued so far
 * on producer; lpf_release_to_stream makes consumer wait for everything the context has queued (what the pipelined
 * modes still owe is launched first).  Neither blocks the host.  (A missing edge is how round 1's bench once
 * died inside torch's set-up gather: DESIGN.md, "The bench_s1 fault".) */
int  lpf_wait_for_stream(lpf_ctx *ctx, void *producer_stream);
int  lpf_release_to_stream(lpf_ctx *ctx, void *consumer_stream);
int  lpf_sync(lpf_ctx *ctx);
/* Software-pipelined device-mode runs.  on = 2: ONE launch per run -- the launch of run i carries its own streaming
 * kernel, the tail (index lists, box counts) of run i-1 dealt out among the streaming tiles, and the per-frame summaries of
 * run i-2; three scratch sets rotate, nothing in a launch depends on anything else in it, no second stream and no event is
 * involved.  What is still owed is launched by lpf_sync(), lpf_release_to_stream() or a call that needs the pipeline empty.
 * on = 4: as 2, and the MASK PACK rides as well, so that nothing is left on the stream between two launches: the launch
 * made by run i carries the pack of run i's masks (uint8 masks lent with on_device = 2 and no erosion; other masks are
 * packed by their own launch as before), the streaming kernel of run i-1 -- its label images were packed one launch
 * earlier -- the tail of run i-2 and the summaries of run i-3; the pack blocks come behind the streaming tiles and fill
 * their ramp-down (16 M-point step: 92 us instead of 98).  Four scratch sets rotate.  A run's POINTS are read, and its
 * outputs written, by the launch of the NEXT run (or by lpf_sync / lpf_release_to_stream): keep them untouched until then.
 * Per-run state travels with the run: the label images rotate with the scratch sets (call lpf_set_masks_* before every
 * lpf_run* while a pipelined mode is on); lpf_set_boxes* for the next run writes the next of four box sets and its table
 * set-up rides in that run's launch (boxes that are not set again stay in force); a run whose batch shape differs from the
 * previous one's brings its own frame / segment / block tables -- none of these drains the pipeline or waits for the GPU.
 * Lent inputs (on_device = 2: masks, box corners) of run i must stay unchanged until the launch after the next has been
 * queued AND has executed, i.e. until the results of run i are complete.  With a pipelined mode on, the outputs of a run are
 * complete after lpf_sync() (or lpf_release_to_stream()), not after the caller's stream alone.  0 = off (default).
 * (Modes 1 and 3 -- tail kernels / mask pack on internal side streams -- and lpf_set_cu_partition existed up to ABI 4; they
 *  measured slower than mode 2 on every workload and were removed.) */
int  lpf_set_pipelined(lpf_ctx *ctx, int on);

#ifdef LPF_LAB
/* Lab builds only (liblpf_lab.so; tools/ and the forced-geometry tests).  Launch geometry: a run cuts every frame into
 * segments -- one list wave each -- and K1 tiles: 1024-point segments of 512-point tiles for launches of up to 3.5 Mi points
 * (a real frame is then ~107 waves instead of 27), 4096-point segments of 1024-point tiles beyond; a launch of a few frames
 * runs the tail in its wide form (16 waves share the masked points of four segments).  Results do not depend on any of it.
 * 0 = by launch size (what the product always does), 1 = small with the wide tail, 2 = large, 3 = large with the segment
 * prefixes taken from the scan kernel (what frames of more than 64 x 64 segments get by themselves), 4 = small with the
 * narrow tail. */
int  lpf_set_geometry(lpf_ctx *ctx, int mode);
/* Role clock of the step launches of the software-pipelined modes: out[6][5] = per role (0 summaries, 1 box job, 2 lists, 3 box
 * counts, 4 mask pack, 5 project+label tiles) {first block start, last block end, sum of block durations, blocks, longest block}
 * in ticks of the 100 MHz wall clock, accumulated since the last reset.  The first call switches it on.  Synchronises. */
int  lpf_lab_role_clock(lpf_ctx *ctx, unsigned long long *out, int reset);
#endif

/* ---- per-sequence state --------------------------------------------------------
 * Replaces V3:565-569 + V3:584 constants.  T = TrVeloToRect (row-major 4x4, V3:535),
 * K = camera.K[:3,:3] (row-major 3x3), W,H = camera.width/height,
 * valid <=> 0<=u<W && 0<=v<H && depth > depth_min_excl && depth < depth_max_excl
 * (reference: 0 and 50, or 0 and 30 in V4/V5). */
int lpf_set_camera(lpf_ctx *ctx, const double T_velo_to_rect[16], const double K[9],
                   int W, int H, double depth_min_excl, double depth_max_excl);

/* ---- per-frame (or per-batch) state --------------------------------------------
 * Masks of F frames, M <= 32 per frame (pad with all-zero masks), each H x W.
 * u8: nonzero = member.  f32: the reference's float masks;
 *   binarize = 0 : member <=> mask.astype(np.uint8) != 0                  (V3:222-225 on raw masks, V2/V4)
 *   binarize = 1 : (mask*255).astype(uint8) -> erode -> /255.0 -> astype(uint8) != 0   (V3:82-97 then V3:222)
 *   binarize = 2 : member <=> mask > 0.5 on the raw float mask   (Same_color.py:125, vis.py:185,
 *                  seg_with_pointcloud.py:167: the scripts that index the YOLO mask without astype)
 * erode_iters: iterations of cv2.erode with the 3x3 MORPH_ELLIPSE (cross) element (V3:83-90).
 * The packed result is a label image [F][H][W], bit m = mask m, kept in HBM.
 * on_device: 0 = host memory (copied before the call returns); 1 = device memory, packed in stream order by this call (the
 *   buffer may be rewritten, in stream order, as soon as the call has returned); 2 = device memory LENT to the context: it
 *   stays unchanged until every run that uses these masks has completed.  Lent masks (and host masks, which sit in the
 *   context's own staging buffer) that need no erosion are not packed at the call in serial mode: a small launch (a real
 *   frame or a few) then looks a valid point's M mask values up directly -- ~20 k points x M bytes instead of a separate
 *   4.7 us launch over 530 k pixels x M -- and a large one packs them first, on the same stream.  The software-pipelined
 *   modes (lpf_set_pipelined 2 / 4) leave lent masks to the next lpf_run* in the same way: a small launch's tiles read
 *   them directly, a large one packs them (mode 4: by blocks of its own launch, see there).  Same results. */
int lpf_set_masks_u8(lpf_ctx *ctx, const uint8_t *masks, int F, int M, int erode_iters, int on_device);
/* Optional hint for the NEXT lpf_set_masks_* call with the same F and M: rects[F][M][4] = {x0, y0, x1, y1} (int32, pixels, half
 * open) -- the caller's word that mask m of frame f is zero outside its rectangle.  A detector hands out every mask with its 2D box
 * and crops the mask to it (the reference's segmenter returns them side by side, cvs_erosion.py:86-87, 110: `boxes`, `masks`); a real
 * frame's masks are a few per cent non-zero.  With the hint a mask is READ AS ZERO OUTSIDE ITS RECTANGLE, pixel for pixel, by every
 * form that takes it -- uint8 masks and float masks under binarize = 0, without erosion, image at least 16 pixels wide:
 *   - launches of sparse frames (fewer points per frame than half the image has pixels) read the lent / staged masks THEMSELVES:
 *     large launches through a per-frame candidate grid of the rectangles (16 x 16-pixel cells) and an exact test for the rows that
 *     have a candidate -- no pack, no label image, whatever the launch size; small launches (a frame or a few) read a valid point's M
 *     mask bytes and gate them by the rectangles;
 *   - dense frames' masks are packed, and the pack reads a 16-pixel group of mask m only where it meets m's rectangle.
 * A rectangle may reach beyond the image (it is read as clipped to it; INT32_MIN / INT32_MAX as "no limit" are fine), x1 <= x0 or
 * y1 <= y0 is an empty one.
 * The same results as without the hint as long as the caller's word holds; erosion and the other float rules ignore it.  on_device:
 * 0 = host memory, copied now without a wait; otherwise device memory (16-byte aligned) that stays unchanged until the runs that use
 * these masks have completed, like lent masks.  rects = NULL clears a pending hint.  The hint is consumed by the next lpf_set_masks_*. */
int lpf_set_mask_rects(lpf_ctx *ctx, const int32_t *rects, int on_device, int F, int M);
int lpf_set_masks_f32(lpf_ctx *ctx, const float *masks, int F, int M, int binarize,
                      int erode_iters, int on_device);
/* Pre-packed label images [F][H][W] (bit m = mask m). */
int lpf_set_label_image(lpf_ctx *ctx, const uint32_t *label, int F, int M, int on_device);
/* Read back the label images currently held ([F][H][W]); for tests of the pack/erode kernels. */
int lpf_get_label_image(lpf_ctx *ctx, uint32_t *out, int on_device);

/* Box corners in the velodyne frame, f64 [Btot][8][3] in the dataset's corner order
 * (output of transform_bboxes_to_velodyne, V3:41-52); frame f owns boxes
 * [box_off[f], box_off[f+1]).  oriented = 1: oriented_point_in_bbox (V3:167-204, the
 * three skewed slabs c1-c0, c3-c0, c4-c0); 0: point_in_bbox (V3:143-164).  box_off: host memory.
 * The box tables (slab parameters in the reference's arithmetic, float bounds, per-cell candidate lists) are built on the
 * device by one block per frame -- a kernel on the context's stream in serial mode (capturable: with unchanged box counts a
 * per-frame box change sits inside a captured graph), blocks of the next lpf_run*'s own launch in the software-pipelined
 * modes.  No call waits for the GPU unless it returns results to host memory or has to grow a buffer.
 * on_device: 0 = host memory (copied before the call returns); 1 = device memory, copied in stream order by this call (the
 * buffer may be rewritten, in stream order, as soon as the call has returned); 2 = device memory LENT to the context: read
 * when the tables are built (by the next lpf_run* in the pipelined modes), it stays unchanged until that run has completed.
 * lpf_set_camera must come first; changing W or H afterwards drops the boxes. */
int lpf_set_boxes(lpf_ctx *ctx, const double *corners_velo, const int32_t *box_off, int F, int oriented);
int lpf_set_boxes_ex(lpf_ctx *ctx, const double *corners_velo, int on_device, const int32_t *box_off, int F, int oriented);
/* The reference's per-frame box preparation and lpf_set_boxes in one device-side step (V3:556-562):
 *   corners_cam0  f64 [Btot][8][3], the 'corners_cam0' of BBoxes_<frame>.json (host or device per on_device, as above)
 *   T_cam_to_velo inv(TrVeloToCam), row-major 4x4 (host)
 *   filter_visible = 1: boxes that filter_visible_bboxes (V3:121-140) drops stay in the tables at their position but can
 *                  never be hit: count_mb keeps one column per GIVEN box (zero for a dropped one) and best_box indexes the
 *                  given list; the position in the reference's filtered list is the number of kept boxes before it.
 * Optional outputs (NULL = not wanted; host or device like the input): visible[Btot] (1 = kept), corners_velo
 * [Btot][8][3] (transform_bboxes_to_velodyne, V3:41-52), bbox2d [Btot][4] and front [Btot] as lpf_prepare_boxes.  Host outputs
 * are filled when the call returns (it waits for them); device outputs are written when the tables are built -- in the
 * pipelined modes by the launch of the next lpf_run*, or by lpf_sync / lpf_release_to_stream if that comes first. */
int lpf_set_boxes_cam0(lpf_ctx *ctx, const double *corners_cam0, int on_device, const int32_t *box_off, int F,
                       const double T_cam_to_velo[16], int filter_visible, int oriented,
                       uint8_t *visible, double *corners_velo, double *bbox2d, int32_t *front);

/* ---- the hot path ----------------------------------------------------------------
 * Replaces, per frame: V3:565-569 (transform + cam2image), V3:584-592 (clip, np.where),
 * extract_car_points_by_mask (V3:211-233), the oriented_point_in_bbox counting loop and
 * best-box scan of calculate_car_point_statistics (V3:344-379).
 * pts: f32 [Ntot][4] (x, y, z, reflectance) exactly as read from the .bin (V3:28). */
int lpf_run(lpf_ctx *ctx, const float *pts, int64_t N, int pts_on_device, const lpf_outputs *out);
int lpf_run_batch(lpf_ctx *ctx, const float *pts, const int64_t *frame_off, int F,
                  int pts_on_device, const lpf_outputs *out);

/* One frame of a stream in ONE call (one FFI crossing instead of four): what a frame of the reference's loop brings -- its scan, its
 * detection masks with their 2D boxes, its annotated 3D boxes (V3:545-562) -- and where its results go.  Exactly the sequence
 *   lpf_set_mask_rects(ctx, mask_rects, 1, 1, n_masks)                  if mask_rects
 *   lpf_set_masks_u8(ctx, masks, 1, n_masks, 0, 2)                      if masks        (lent: on_device = 2)
 *   lpf_set_boxes_cam0(ctx, corners_cam0, 2, {0, n_boxes}, 1, T_cam_to_velo, filter_visible, oriented, 0, 0, 0, 0)   if corners_cam0
 *   lpf_run(ctx, pts, n_points, 1, &out)
 * with the same meaning, ownership and error behaviour; a NULL masks / corners_cam0 leaves the masks / boxes in force as they are.
 * Every pointer except T_cam_to_velo is device memory; out.on_device must be 1.  For launch-bound frame loops: in a software-
 * pipelined stream of single real frames the three calls took the host 7.9 us per frame against 9.2 us on the GPU. */
typedef struct lpf_frame_job {
    const float   *pts;            /* [n_points][4] */
    int64_t        n_points;
    const uint8_t *masks;          /* [n_masks][H][W], lent; or NULL */
    const int32_t *mask_rects;     /* [n_masks][4] {x0, y0, x1, y1}, lent like the masks; or NULL */
    const double  *corners_cam0;   /* [n_boxes][8][3], lent; or NULL */
    const double  *T_cam_to_velo;  /* host memory, row-major 4x4 (with corners_cam0) */
    int32_t        n_masks, n_boxes;
    int32_t        filter_visible, oriented;
    lpf_outputs    out;
} lpf_frame_job;
int lpf_run_frame(lpf_ctx *ctx, const lpf_frame_job *job);

/* ---- box membership as a stand-alone operator -------------------------------------------
 * inside[b*k + i] = 1 if point i lies in box b, else 0: the boolean arrays the reference's
 * oriented_point_in_bbox (V3:167-208, oriented = 1) and point_in_bbox (V3:143-164, oriented = 0)
 * return, for B boxes at once.  pts: f32 [k][stride] with stride 3 or 4 (x, y, z first);
 * corners_velo: f64 [B][8][3] host memory; pts / inside are host or device per on_device. */
int lpf_points_in_boxes(lpf_ctx *ctx, const float *pts, int64_t k, int stride, const double *corners_velo,
                        int B, int oriented, uint8_t *inside, int on_device);

/* ---- last-writer depth image -------------------------------------------------------------------
 * seg_with_pointcloud.py:160-170 fills, per mask, depthMap[v,u] = depth[idx] for idx ascending over
 * the valid points inside the mask; the last valid point of a pixel wins whatever the mask, so one
 * image describes all of them: depthMap_i = where(mask_i > 0.5, D, 0).  depth_img: f64 [H][W]
 * (0 where no valid point projects), winner: int32 [H][W] index of that point or -1 (may be NULL).
 * Uses lpf_set_camera's transform and depth window.  pts as in lpf_run; outputs follow on_device. */
int lpf_depth_image(lpf_ctx *ctx, const float *pts, int64_t N, int on_device, double *depth_img, int32_t *winner);

/* cv2.resize(mask.astype(np.uint8), (camera.width, camera.height)) (V3:222; INTER_LINEAR, the default) for masks that do not arrive at
 * the camera's size (the reference's scripts all pass retina_masks=True, so theirs do): n planes [h][w] of uint8 -> n planes [H][W]
 * (the size of lpf_set_camera), which then go to lpf_set_masks_u8 (nonzero = member = the reference's `> 0.5`).  Restated from
 * OpenCV 4.x resize.cpp (HResizeLinear / VResizeLinear, 11-bit weights; the x axis clamps index and fraction at the ends, the y axis
 * keeps the fraction and clips the two row indices) and pinned by construction only -- OpenCV is not part of this image and the
 * reference holds no resized fixture (oracle/numpy_path.py: cv2_resize_linear_u8 states the formula).  An exact 2 x 2
 * decimation (h == 2 H and w == 2 W) is what OpenCV hands to INTER_AREA: the rounded mean (a + b + c + d + 2) >> 2.  on_device: both pointers in host (0) or device (1) memory; device
 * callers: in stream order.  Not capturable. */
int lpf_resize_masks_u8(lpf_ctx *ctx, const uint8_t *src, int n, int h, int w, uint8_t *dst, int on_device);

/* cv2.erode(plane, cv2.getStructuringElement(cv2.MORPH_ELLIPSE, (3, 3)), iterations=iters) on n planes [h][w] of 8-bit VALUES, at the
 * planes' own size (V3:83-90): the minimum over the plus-shaped neighbourhood, the image border left out.  For masks that do not
 * arrive at camera size the reference erodes first and resizes afterwards (V3:82-97, then V3:222): this call, then
 * lpf_resize_masks_u8.  (Masks at camera size are eroded inside lpf_set_masks_*, on the packed label image.)  src != dst; host or
 * device pointers per on_device, device callers in stream order.  Pinned by construction only, like lpf_set_masks_*'s erosion. */
int lpf_erode_masks_u8(lpf_ctx *ctx, const uint8_t *src, int n, int h, int w, int iters, uint8_t *dst, int on_device);

/* ---- box preparation on the GPU --------------------------------------------------------------
 * For nbox annotated boxes given by their 8 corners in the cam-0 frame (f64 [nbox][8][3], the
 * 'corners_cam0' of BBoxes_<frame>.json):
 *   visible[b]      = 1 if filter_visible_bboxes keeps the box (V3:121-140; needs lpf_set_camera's K, W, H)
 *   corners_velo    = transform_bboxes_to_velodyne's output (V3:41-52), f64 [nbox][8][3];
 *                     T_cam_to_velo = inv(TrVeloToCam), row-major 4x4
 *   bbox2d[b][4]    = {min u, min v, max u, max v} of the projected corners with depth > 0 and
 *   front[b]        = how many corners those are (V4:157-168; 0 -> the box is skipped by the IoU match)
 * Any output may be NULL.  Host pointers. */
int lpf_prepare_boxes(lpf_ctx *ctx, const double *corners_cam0, int nbox, const double T_cam_to_velo[16],
                      uint8_t *visible, double *corners_velo, double *bbox2d, int32_t *front);

/* ---- hipGraph capture of a launch set ------------------------------------------------------
 * lpf_graph_begin puts the context's stream into capture mode; every device-mode lpf_set_masks_* /
 * lpf_run* call issued until lpf_graph_end is recorded instead of executed (pointers and sizes are
 * baked in; the caller may add its own async copies on the same stream in between).  lpf_graph_end
 * instantiates the graph; lpf_graph_launch replays it on the context's stream.  The context must
 * have run the same shapes once before capture (so no allocation or table upload happens inside
 * it), and pipelining must be off.  For launch-bound per-frame loops (10 Hz streaming).
 * A graph points into buffers and tables the context owns.  Anything that moves or rewrites them after the
 * capture -- a run with another batch geometry, lpf_set_boxes with other box counts, lpf_set_camera, lpf_set_stream, a mode
 * switch, a call that had to grow a scratch buffer -- makes the graph stale: lpf_graph_launch then returns LPF_ERR_STATE
 * instead of replaying it.  An error returned by a call made inside a capture abandons the capture. */
typedef struct lpf_graph lpf_graph;
int  lpf_graph_begin(lpf_ctx *ctx);
int  lpf_graph_end(lpf_ctx *ctx, lpf_graph **out);
int  lpf_graph_launch(lpf_ctx *ctx, lpf_graph *g);
void lpf_graph_destroy(lpf_graph *g);

/* ---- measurement -------------------------------------------------------------------
 * With profiling on, every lpf_run* brackets its project+label kernel (lpf_k1_project, the
 * dominant kernel) with HIP events on the context's stream.  lpf_profile_read waits for the
 * stream, returns the summed elapsed milliseconds and the number of bracketed launches
 * since the last reset.  (No reference counterpart: the reference has no timers.) */
int lpf_profile_enable(lpf_ctx *ctx, int on);
int lpf_profile_read(lpf_ctx *ctx, double *k1_ms_sum, int64_t *k1_launches, int reset);
/* Duration between two event records with nothing between them on the context's stream (median of 33):
 * the part of an lpf_profile_* bracket that is not the kernel (4.6 us on MI355X / ROCm 7.2). */
int lpf_profile_overhead(lpf_ctx *ctx, double *empty_bracket_ms);

/* What the context has done so far (for tests and tuning: a software-pipelined stream must show no host wait and no drain):
 * out[0] host waits (the calling thread blocked on the GPU), [1] drains (owed work of the pipelined modes launched outside
 * a run), [2] table / corner uploads through the pinned ring (no wait), [3] step launches, [4] box jobs launched as a kernel
 * of their own, [5] box jobs that rode in a step launch, [6] uploads too large for the ring (they wait); n <= 8. */
int lpf_get_stats(lpf_ctx *ctx, int64_t *out, int n, int reset);

/* ---- multi-GPU: the one exchange step ------------------------------------------------------------
 * Frames shard across ranks with no data-path collective (V3:541: the frame loop has no cross-frame state);
 * what the ranks exchange at the end is a short int64 vector -- the aggregates of analyze_master_csv
 * (cvs:268-295): frames with rows, cars, matched cars, sums of total / inside / outside points, the inside
 * percentages as integer hundredths -- summed (op 0), or reduced by MIN (1) / MAX (2).  vec: host memory,
 * reduced in place over the caller's RCCL communicator (ncclComm_t, from ncclCommInitRank), one rank per GPU,
 * on the context's stream; returns when vec holds the result -- it blocks, with no timeout, until every rank of the
 * communicator has made the call.  ncclAllReduce is taken from the RCCL already loaded in the process (so that it is the
 * library the communicator came from); only a process with none gets the system's librccl loaded at the first call. */
int lpf_allreduce_metrics(lpf_ctx *ctx, int64_t *vec, int n, int op, void *rccl_comm);

/* ---- scan reader ------------------------------------------------------------------------------
 * Double-buffered velodyne .bin reader for frame loops and the 10 Hz stream.  Stands where the
 * reference calls Kitti360Viewer3DRaw.loadVelodyneData once per frame (V3:24-28, V3:545:
 * np.fromfile(path, float32).reshape(-1, 4); RuntimeError('<path> does not exist!') when missing):
 * a worker thread reads the submitted files, in order, into pinned host memory and an internal
 * copy stream moves them to HBM while earlier scans are being processed.
 *   lpf_reader_create  n_buffers in [2,16] scans in flight, each up to max_points points.
 *   lpf_reader_submit  enqueue a file (returns at once; any number may be outstanding).
 *   lpf_reader_next    the oldest submitted scan: *d_pts = its HBM copy (pass to lpf_run* with
 *                      pts_on_device = 1), *h_pts = the pinned host copy (the array the reference
 *                      would hold; for host-side gathers), *n_points = N.  The context's stream
 *                      waits on the device for the copy; the host does not block on PCIe.  Pointers
 *                      stay valid until the next lpf_reader_next on this reader (work already enqueued
 *                      on the context's stream may still use them after that).  A missing / malformed
 *                      file gives LPF_ERR_IO for that scan (lpf_last_error names it) and the reader
 *                      carries on with the following one.
 *   lpf_reader_destroy before lpf_destroy of its context (the reader uses the context's stream).
 *   lpf_reader_wait    host-side wait for the copy of the scan handed out last (only for callers that
 *                      touch *d_pts outside the context's stream). */
typedef struct lpf_reader lpf_reader;
int  lpf_reader_create(lpf_ctx *ctx, lpf_reader **out, int n_buffers, int64_t max_points);
int  lpf_reader_submit(lpf_reader *rd, const char *path);
int  lpf_reader_next(lpf_reader *rd, const float **d_pts, const float **h_pts, int64_t *n_points);
int  lpf_reader_wait(lpf_reader *rd);
void lpf_reader_destroy(lpf_reader *rd);

/* A frame's second file: bboxes_3D_cam0/BBoxes_<frame>.json, which the reference reads with json.load (load_bounding_boxes,
 * V3:31-38; cvs_erosion.py:333): a list of {"index": int, "corners_cam0": [[x, y, z] x 8]}.  Parsing 270 KB of 17-digit numbers
 * (frame 2449, 314 boxes) costs a Python frame loop more than all of its GPU work, so the reader's worker can do it beside the scan:
 *   lpf_reader_submit_frame  as lpf_reader_submit, plus the frame's box file (NULL: none).
 *   lpf_reader_boxes         the box file of the scan handed out by the last lpf_reader_next: *state says what became of it
 *                            (LPF_BOXES_*); when PARSED, *corners_cam0 = float64 [*nbox][8][3] -- what lpf_prepare_boxes /
 *                            lpf_set_boxes_cam0 take -- and *index = int32 [*nbox], host memory of the reader, valid until the next
 *                            lpf_reader_next.  A missing or unexpected box file never fails the scan.
 *   lpf_parse_boxes_json     the same parser on its own (no context, no GPU): fills the caller's arrays of capacity cap boxes;
 *                            *nbox = boxes in the file; more than cap: LPF_ERR_ARG with *nbox set (file size / 64 boxes always do).
 * Numbers are converted with strtod in the C locale -- correctly rounded, as Python's float() is: the same doubles.  OTHER means the
 * file is not that plain schema (other keys, a float index, NaN / Infinity, escapes in keys, malformed or trailing text) and has NOT
 * been interpreted: hand it to a JSON library, whose result or error is then the reference's. */
enum lpf_boxes_state {
    LPF_BOXES_PARSED = 0,          /* corners and indices are there (possibly zero boxes: the reference skips such a frame) */
    LPF_BOXES_ABSENT = 1,          /* no such file: the reference prints "No bounding boxes found" and skips the frame */
    LPF_BOXES_OTHER = 2,           /* not the plain schema / unreadable: not interpreted */
    LPF_BOXES_NONE = 3             /* no box file was submitted with that scan */
};
int  lpf_reader_submit_frame(lpf_reader *rd, const char *scan_path, const char *boxes_path);
int  lpf_reader_boxes(lpf_reader *rd, const double **corners_cam0, const int32_t **index, int *nbox, int *state);
int  lpf_parse_boxes_json(const char *path, double *corners_cam0, int32_t *index, int cap, int *nbox, int *state);

#ifdef __cplusplus
}
#endif
#endif /* LPF_H */
