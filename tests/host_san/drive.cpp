// drive.cpp -- drives the HOST side of liblpf (lpf_api.hip compiled --offload-host-only against fake_hip.cpp) under a sanitizer:
// argument validation, host-memory runs, the software-pipelined modes with new masks / rectangles / boxes / batch shapes every
// run (the pinned upload ring is lapped several times: > 40 000 small uploads and sizes that fill a quarter exactly), the box-set
// and scratch-set rotation, the graph state machine, mask resize / erosion staging and the reader's worker threads.  Kernel
// launches do nothing (fake_hip.cpp), so results are not checked here -- the GPU tests do that; what is checked is that every
// call returns what it should and that the sanitizer stays silent.
#include "../../include/lpf.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

extern "C" long long fake_hip_launches(void);

static int g_fail = 0;
#define CHECK(cond) do { if (!(cond)) { fprintf(stderr, "drive.cpp:%d: CHECK failed: %s  [%s]\n", __LINE__, #cond, ctx_err()); ++g_fail; } } while (0)
static lpf_ctx *g_ctx = nullptr;
static const char *ctx_err() { return lpf_last_error(g_ctx); }

static const double T16[16] = {0, -1, 0, 0.1, 0, 0, -1, 0.2, 1, 0, 0, 0.3, 0, 0, 0, 1};
static const double K9[9] = {552.5, 0, 682.0, 0, 552.5, 238.7, 0, 0, 1};
static const int W = 128, H = 48;                             // (a small image; pipelined_streams also runs a 1408 x 376 one)

struct Dev {                      // "device" buffers: heap blocks, so that the sanitizer knows their bounds
    std::vector<void *> all;
    template <typename T> T *get(size_t n) { void *p = calloc(n ? n : 1, sizeof(T)); all.push_back(p); return (T *)p; }
    ~Dev() { for (void *p : all) free(p); }
};

static void fill_outputs(Dev &D, lpf_outputs &o, int64_t n, int F, int M, int Btot, int64_t cap, int on_device)
{
    memset(&o, 0, sizeof o);
    o.uv = D.get<int32_t>(2 * n); o.label_bits = D.get<uint32_t>(n); o.valid_idx = D.get<int64_t>(n);
    o.inst_idx = D.get<int64_t>((size_t)F * cap); o.inst_cap = cap; o.count_mb = D.get<int32_t>((size_t)(M ? M : 1) * (Btot ? Btot : 1));
    o.summary = D.get<lpf_frame_summary>(F); o.uv_valid = D.get<int32_t>(2 * n); o.label_valid = D.get<uint32_t>(n);
    o.on_device = on_device;
}

static void argument_errors()
{
    lpf_ctx *c = nullptr;
    CHECK(lpf_create(nullptr, 0) == LPF_ERR_ARG);
    CHECK(lpf_create(&c, 5) == LPF_ERR_ARG && c == nullptr);
    CHECK(lpf_create(&c, 0) == LPF_OK && c);
    g_ctx = c;
    lpf_outputs o; memset(&o, 0, sizeof o);
    float p4[4] = {1, 2, 3, 0};
    CHECK(lpf_run(c, p4, 1, 0, &o) == LPF_ERR_STATE);                       // no camera yet
    CHECK(lpf_set_masks_u8(c, nullptr, 1, 1, 0, 0) == LPF_ERR_STATE);
    CHECK(lpf_set_camera(c, nullptr, K9, W, H, 0, 50) == LPF_ERR_ARG);
    CHECK(lpf_set_camera(c, T16, K9, 0, H, 0, 50) == LPF_ERR_ARG);
    CHECK(lpf_set_camera(c, T16, K9, W, H, 0, 50) == LPF_OK);
    CHECK(lpf_set_masks_u8(c, nullptr, 1, 33, 0, 0) == LPF_ERR_ARG);
    CHECK(lpf_set_masks_u8(c, nullptr, 1, 2, 0, 0) == LPF_ERR_ARG);
    CHECK(lpf_set_masks_f32(c, nullptr, 0, 0, 7, 0, 0) == LPF_ERR_ARG);
    CHECK(lpf_set_pipelined(c, 1) == LPF_ERR_ARG && lpf_set_pipelined(c, 3) == LPF_ERR_ARG && lpf_set_pipelined(c, 9) == LPF_ERR_ARG);
    int32_t bad_off[2] = {1, 3};
    double corners[24 * 3] = {0};
    CHECK(lpf_set_boxes(c, corners, bad_off, 1, 1) == LPF_ERR_ARG);
    int32_t desc[3] = {0, 2, 1};
    CHECK(lpf_set_boxes(c, corners, desc, 2, 1) == LPF_ERR_ARG);
    CHECK(lpf_run(c, nullptr, 5, 0, &o) == LPF_ERR_ARG);
    CHECK(lpf_run(c, p4, -1, 0, &o) == LPF_ERR_ARG);
    o.inst_idx = (int64_t *)p4; o.inst_cap = 0;
    CHECK(lpf_run(c, p4, 1, 0, &o) == LPF_ERR_ARG);
    memset(&o, 0, sizeof o);
    o.uv_valid = (int32_t *)p4;
    CHECK(lpf_run(c, p4, 1, 0, &o) == LPF_ERR_ARG);                         // uv_valid needs valid_idx
    CHECK(lpf_graph_end(c, nullptr) == LPF_ERR_ARG);
    lpf_graph *g = nullptr;
    CHECK(lpf_graph_end(c, &g) == LPF_ERR_STATE);
    CHECK(lpf_run_frame(c, nullptr) == LPF_ERR_ARG);
    CHECK(lpf_allreduce_metrics(c, nullptr, 1, 0, nullptr) == LPF_ERR_ARG);
    CHECK(lpf_resize_masks_u8(c, nullptr, 1, 4, 4, nullptr, 0) == LPF_ERR_ARG);
    uint8_t px[64] = {0};
    {
        std::vector<uint8_t> big((size_t)4 * W * H, 1), out((size_t)W * H);
        CHECK(lpf_resize_masks_u8(c, big.data(), 1, 2 * H, 2 * W, out.data(), 0) == LPF_OK);   // the INTER_AREA case: no weight tables
        CHECK(lpf_resize_masks_u8(c, big.data(), 1, 2 * H, W, out.data(), 0) == LPF_OK);       // twice in one axis: linear, with tables
    }
    CHECK(lpf_erode_masks_u8(c, px, 1, 8, 8, 1, px, 0) == LPF_ERR_ARG);          // src == dst
    CHECK(strlen(lpf_build_id()) > 0 && lpf_abi_version() == LPF_ABI_VERSION);
    void *pin = lpf_host_alloc(1000);
    CHECK(pin != nullptr);
    memset(pin, 1, 1000);
    lpf_host_free(pin);
    lpf_destroy(c);
    lpf_destroy(nullptr);
    g_ctx = nullptr;
}

static void host_memory_runs()
{
    lpf_ctx *c = nullptr;
    CHECK(lpf_create(&c, 0) == LPF_OK);
    g_ctx = c;
    CHECK(lpf_set_camera(c, T16, K9, W, H, 0, 50) == LPF_OK);
    Dev D;
    const int F = 3, M = 5;
    const int64_t off[F + 1] = {0, 1000, 1000, 4321};                      // an empty frame in the middle
    std::vector<float> pts(4 * off[F], 1.0f);
    std::vector<uint8_t> masks((size_t)F * M * W * H, 1);
    std::vector<float> fmasks((size_t)F * M * W * H, 1.0f);
    std::vector<double> corners(24 * 7, 0.5);
    const int32_t boff[F + 1] = {0, 3, 3, 7};
    lpf_outputs o;
    fill_outputs(D, o, off[F], F, M, 7, 5000, 0);
    for (int it = 0; it < 3; ++it) {
        CHECK(lpf_set_masks_u8(c, masks.data(), F, M, it, 0) == LPF_OK);
        CHECK(lpf_set_boxes(c, corners.data(), boff, F, it & 1) == LPF_OK);
        CHECK(lpf_run_batch(c, pts.data(), off, F, 0, &o) == LPF_OK);
        CHECK(lpf_set_masks_f32(c, fmasks.data(), F, M, it % 3, it, 0) == LPF_OK);
        CHECK(lpf_run_batch(c, pts.data(), off, F, 0, &o) == LPF_OK);
    }
    std::vector<uint32_t> img((size_t)F * W * H);
    CHECK(lpf_get_label_image(c, img.data(), 0) == LPF_OK);
    CHECK(lpf_set_label_image(c, img.data(), F, M, 0) == LPF_OK);
    CHECK(lpf_run_batch(c, pts.data(), off, F, 0, &o) == LPF_OK);
    CHECK(lpf_run_batch(c, pts.data(), off, 2, 0, &o) == LPF_ERR_STATE);    // masks were set for 3 frames
    // box preparation, stand-alone operators
    std::vector<uint8_t> vis(7), inside(7 * 100);
    std::vector<double> cv(24 * 7), bb(4 * 7);
    std::vector<int32_t> fr(7);
    double Tcv[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    CHECK(lpf_prepare_boxes(c, corners.data(), 7, Tcv, vis.data(), cv.data(), bb.data(), fr.data()) == LPF_OK);
    CHECK(lpf_set_boxes_cam0(c, corners.data(), 0, boff, F, Tcv, 1, 1, vis.data(), cv.data(), bb.data(), fr.data()) == LPF_OK);
    CHECK(lpf_points_in_boxes(c, pts.data(), 100, 4, corners.data(), 7, 1, inside.data(), 0) == LPF_OK);
    std::vector<double> dimg((size_t)W * H);
    std::vector<int32_t> win((size_t)W * H);
    CHECK(lpf_depth_image(c, pts.data(), 1000, 0, dimg.data(), win.data()) == LPF_OK);
    // masks of another size: erosion at their own size, then the resize
    std::vector<uint8_t> small((size_t)2 * 20 * 50, 255), er(small.size()), big((size_t)2 * W * H);
    for (int iters = 0; iters < 4; ++iters) CHECK(lpf_erode_masks_u8(c, small.data(), 2, 20, 50, iters, er.data(), 0) == LPF_OK);
    CHECK(lpf_resize_masks_u8(c, er.data(), 2, 20, 50, big.data(), 0) == LPF_OK);
    CHECK(lpf_resize_masks_u8(c, big.data(), 2, H, W, big.data() /* same size: a copy */, 0) == LPF_OK);
    int64_t st[8];
    CHECK(lpf_get_stats(c, st, 8, 1) == LPF_OK && st[0] > 0);
    double ms = 0; int64_t nl = 0;
    CHECK(lpf_profile_enable(c, 1) == LPF_OK && lpf_run_batch(c, pts.data(), off, F, 0, &o) == LPF_OK);
    CHECK(lpf_profile_read(c, &ms, &nl, 1) == LPF_OK && nl == 1 && lpf_profile_overhead(c, &ms) == LPF_OK);
    lpf_destroy(c);
    g_ctx = nullptr;
}

// the software-pipelined modes with everything new every run, long enough to lap the 8 MiB pinned ring several times
static void pipelined_streams(int mode, int runs, const int W, const int H)
{
    // (W x H = 128 x 48: the frames are dense -- masks are packed, the pack rides in mode 4; 1408 x 376: sparse -- with rectangles the
    //  tiles read the masks themselves, small launches gated, and large ones through the candidate grid, which rides in mode 4)
    lpf_ctx *c = nullptr;
    CHECK(lpf_create(&c, 0) == LPF_OK);
    g_ctx = c;
    CHECK(lpf_set_camera(c, T16, K9, W, H, 0, 50) == LPF_OK);
    CHECK(lpf_set_pipelined(c, mode) == LPF_OK);
    Dev D;
    const int M = 4, FMAX = 6;
    const int64_t NMAX = (W > 1000) ? 6000000 : 40000;        // (the large image: now and then a launch beyond 3.5 Mi points, the large geometry)
    float *pts = D.get<float>(4 * NMAX);
    uint8_t *masks = D.get<uint8_t>((size_t)FMAX * M * W * H);
    double *dcorners = D.get<double>(24 * 64);
    int32_t *drects = D.get<int32_t>((size_t)FMAX * M * 4);
    std::vector<double> hcorners(24 * 11000, 0.25);
    std::vector<int32_t> hrects((size_t)FMAX * M * 4, 3);
    lpf_outputs o;
    fill_outputs(D, o, NMAX, FMAX, M, 11000, NMAX, 1);
    double Tcv[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    unsigned rng = 12345u + (unsigned)mode;
    auto rnd = [&](unsigned n) { rng = rng * 1664525u + 1013904223u; return (rng >> 8) % n; };
    for (int it = 0; it < runs; ++it) {
        const int F = 1 + (int)rnd(FMAX);
        int64_t off[FMAX + 1];
        off[0] = 0;
        const unsigned per_frame = (W > 1000 && it % 7 != 0) ? 200000u : (unsigned)(NMAX / FMAX);     // (mostly sparse frames on the large image)
        for (int f = 0; f < F; ++f) off[f + 1] = off[f] + (int64_t)rnd(per_frame);
        // rectangles from host memory (F * M * 16 bytes <= 384: the 256-byte pieces that walked the ring's head to its end) or lent
        if (it % 3 != 2) CHECK(lpf_set_mask_rects(c, it % 3 ? hrects.data() : drects, it % 3 ? 0 : 1, F, M) == LPF_OK);
        CHECK(lpf_set_masks_u8(c, masks, F, M, 0, 2) == LPF_OK);
        int32_t boff[FMAX + 1];
        boff[0] = 0;
        // mostly a few boxes; now and then sizes that fill a quarter of the ring exactly (192 B per box: 10922 boxes = 2 MiB - 128 B)
        const int per = (it % 97 == 0) ? 10922 / F : (int)rnd(9);
        for (int f = 0; f < F; ++f) boff[f + 1] = boff[f] + per;
        if (it % 5 == 0) CHECK(lpf_set_boxes_cam0(c, dcorners, 2, boff, F, Tcv, 1, 1, nullptr, nullptr, nullptr, nullptr) == (boff[F] <= 64 ? LPF_OK : LPF_OK));
        else CHECK(lpf_set_boxes_ex(c, hcorners.data(), 0, boff, F, it & 1) == LPF_OK);
        if (boff[F] > 64 && it % 5 == 0) CHECK(lpf_set_boxes_ex(c, hcorners.data(), 0, boff, F, 1) == LPF_OK);   // (the lent array holds 64 boxes)
        if (F == 1 && it % 4 == 0) {                       // the one-call form: rectangles, lent masks, lent cam-0 corners, the run
            lpf_frame_job j;
            memset(&j, 0, sizeof j);
            j.pts = pts; j.n_points = off[1]; j.masks = masks; j.mask_rects = drects; j.n_masks = M;
            j.corners_cam0 = dcorners; j.n_boxes = 17; j.T_cam_to_velo = Tcv; j.filter_visible = 1; j.oriented = 1; j.out = o;
            CHECK(lpf_run_frame(c, &j) == LPF_OK);
        } else {
            CHECK(lpf_run_batch(c, pts, off, F, 1, &o) == LPF_OK);
        }
        if (it % 1000 == 999) CHECK(lpf_release_to_stream(c, nullptr) == LPF_OK);
    }
    int64_t st[8];
    CHECK(lpf_get_stats(c, st, 8, 0) == LPF_OK);
    fprintf(stderr, "  mode %d: %d runs, uploads %lld, blocking uploads %lld, host waits %lld, drains %lld, step launches %lld\n", mode, runs,
            (long long)st[2], (long long)st[6], (long long)st[0], (long long)st[1], (long long)st[3]);
    CHECK(st[2] > runs);                                                     // the ring was really used
    CHECK(lpf_sync(c) == LPF_OK);
    CHECK(lpf_set_pipelined(c, 0) == LPF_OK);
    lpf_destroy(c);
    g_ctx = nullptr;
}

// the ring alone: 256-byte uploads by the ten thousand (the head reached the end of the buffer after 32768 of them)
static void ring_of_small_uploads()
{
    lpf_ctx *c = nullptr;
    CHECK(lpf_create(&c, 0) == LPF_OK);
    g_ctx = c;
    CHECK(lpf_set_camera(c, T16, K9, W, H, 0, 50) == LPF_OK);
    std::vector<int32_t> rects(4 * 16, 1);
    for (int it = 0; it < 70000; ++it) CHECK(lpf_set_mask_rects(c, rects.data(), 0, 1 + it % 4, 4) == LPF_OK);
    int64_t st[8];
    CHECK(lpf_get_stats(c, st, 8, 0) == LPF_OK && st[2] == 70000);
    lpf_destroy(c);
    g_ctx = nullptr;
}

static void graphs()
{
    lpf_ctx *c = nullptr;
    CHECK(lpf_create(&c, 0) == LPF_OK);
    g_ctx = c;
    CHECK(lpf_set_camera(c, T16, K9, W, H, 0, 50) == LPF_OK);
    Dev D;
    const int M = 3;
    const int64_t N = 5000;
    float *pts = D.get<float>(4 * N);
    uint8_t *masks = D.get<uint8_t>((size_t)M * W * H);
    double *corners = D.get<double>(24 * 5);
    const int32_t boff[2] = {0, 5};
    lpf_outputs o;
    fill_outputs(D, o, N, 1, M, 5, N, 1);
    auto frame = [&]() {
        CHECK(lpf_set_masks_u8(c, masks, 1, M, 1, 1) == LPF_OK);
        CHECK(lpf_set_boxes_ex(c, corners, 1, boff, 1, 1) == LPF_OK);
        CHECK(lpf_run(c, pts, N, 1, &o) == LPF_OK);
    };
    frame();                                                                // the shapes once before the capture
    CHECK(lpf_set_pipelined(c, 2) == LPF_OK && lpf_graph_begin(c) == LPF_ERR_STATE && lpf_set_pipelined(c, 0) == LPF_OK);
    CHECK(lpf_graph_begin(c) == LPF_OK && lpf_graph_begin(c) == LPF_ERR_STATE);     // (an error inside a capture abandons it)
    lpf_graph *g0 = nullptr;
    CHECK(lpf_graph_end(c, &g0) == LPF_ERR_STATE);
    CHECK(lpf_graph_begin(c) == LPF_OK);
    frame();
    lpf_graph *g = nullptr;
    CHECK(lpf_graph_end(c, &g) == LPF_OK && g);
    for (int i = 0; i < 5; ++i) CHECK(lpf_graph_launch(c, g) == LPF_OK);
    CHECK(lpf_set_camera(c, T16, K9, W, H, 0, 30) == LPF_OK);               // anything the graph baked in: stale
    CHECK(lpf_graph_launch(c, g) == LPF_ERR_STATE);
    lpf_graph_destroy(g);
    // an error inside a capture abandons it; the context stays usable
    frame();
    CHECK(lpf_graph_begin(c) == LPF_OK);
    uint8_t host_masks[1] = {0};
    CHECK(lpf_set_masks_u8(c, host_masks, 1, M + 40, 0, 1) == LPF_ERR_ARG);
    CHECK(lpf_graph_end(c, &g) == LPF_ERR_STATE);
    frame();
    CHECK(lpf_graph_begin(c) == LPF_OK);
    std::vector<float> big(4 * 4 * N);
    lpf_outputs ob;
    fill_outputs(D, ob, 4 * N, 1, M, 5, 4 * N, 1);
    float *bigpts = D.get<float>(4 * 4 * N);
    CHECK(lpf_run(c, bigpts, 4 * N, 1, &ob) == LPF_ERR_STATE);              // a buffer would have to grow inside the capture
    frame();
    lpf_graph_destroy(nullptr);
    lpf_destroy(c);
    g_ctx = nullptr;
}

static void reader(const char *tmpdir)
{
    lpf_ctx *c = nullptr;
    CHECK(lpf_create(&c, 0) == LPF_OK);
    g_ctx = c;
    std::vector<std::string> paths;
    for (int i = 0; i < 12; ++i) {
        std::string p = std::string(tmpdir) + "/scan" + std::to_string(i) + ".bin";
        if (i != 4) {                                                       // scan 4 is missing
            FILE *f = fopen(p.c_str(), "wb");
            std::vector<float> v((size_t)4 * (100 + 37 * i) + (i == 7 ? 3 : 0), (float)i);      // scan 7 is not [N][4]
            fwrite(v.data(), 4, v.size(), f);
            fclose(f);
        }
        paths.push_back(p);
    }
    // the frames' box files ride along: scan i has i boxes (file 3 is absent, file 5 has a key the schema lacks, file 6 is cut short)
    std::vector<std::string> bpaths;
    for (int i = 0; i < 12; ++i) {
        std::string p = std::string(tmpdir) + "/BBoxes_" + std::to_string(i) + ".json";
        if (i != 3) {
            std::string t = "[";
            for (int b = 0; b < i; ++b) {
                t += std::string(b ? ",\n " : "") + "{\"index\": " + std::to_string(100 * i + b) + ", \"corners_cam0\": [";
                for (int k = 0; k < 8; ++k) t += std::string(k ? ", " : "") + "[" + std::to_string(i) + ".5, -" + std::to_string(b) + "e-1, " + std::to_string(k) + "]";
                t += i == 5 ? "], \"label\": 1}" : "]}";
            }
            t += "]";
            if (i == 6) t.resize(t.size() / 2);
            FILE *f = fopen(p.c_str(), "wb");
            fwrite(t.data(), 1, t.size(), f);
            fclose(f);
        }
        bpaths.push_back(p);
    }
    lpf_reader *rd = nullptr;
    CHECK(lpf_reader_create(c, &rd, 1, 1000) == LPF_ERR_ARG);
    CHECK(lpf_reader_create(c, &rd, 3, 600) == LPF_OK && rd);
    {
        int nb = -1, st = -1;
        CHECK(lpf_reader_boxes(rd, nullptr, nullptr, &nb, &st) == LPF_ERR_STATE);                // nothing handed out yet
    }
    for (int i = 0; i < 12; ++i) CHECK(lpf_reader_submit_frame(rd, paths[(size_t)i].c_str(), i == 11 ? nullptr : bpaths[(size_t)i].c_str()) == LPF_OK);
    for (int i = 0; i < 12; ++i) {
        const float *d = nullptr, *h = nullptr;
        int64_t n = 0;
        const int rc = lpf_reader_next(rd, &d, &h, &n);
        if (i == 4 || i == 7 || 100 + 37 * i > 600) CHECK(rc == LPF_ERR_IO);
        else { CHECK(rc == LPF_OK && n == 100 + 37 * i && h && d && h[0] == (float)i && d[4 * n - 1] == (float)i); CHECK(lpf_reader_wait(rd) == LPF_OK); }
        const double *bc = nullptr; const int32_t *bi = nullptr;
        int nb = -1, st = -1;
        CHECK(lpf_reader_boxes(rd, &bc, &bi, &nb, &st) == LPF_OK);                                 // (a failed scan still has its box file)
        if (i == 11) CHECK(st == LPF_BOXES_NONE && nb == 0 && !bc);
        else if (i == 3) CHECK(st == LPF_BOXES_ABSENT && nb == 0);
        else if (i == 5 || i == 6) CHECK(st == LPF_BOXES_OTHER && nb == 0 && !bc && !bi);
        else {
            CHECK(st == LPF_BOXES_PARSED && nb == i);
            for (int b = 0; b < nb; ++b)
                CHECK(bi[b] == 100 * i + b && bc[24 * b] == i + 0.5 && bc[24 * b + 1] == -(double)b / 10.0 && bc[24 * b + 23] == 7.0);
        }
    }
    for (int i = 0; i < 3; ++i) CHECK(lpf_reader_submit(rd, paths[i].c_str()) == LPF_OK);      // destroyed with scans still queued
    lpf_reader_destroy(rd);
    lpf_destroy(c);
    g_ctx = nullptr;
}

// lpf_parse_boxes_json on every prefix of a file and on a few thousand damaged copies of it: whatever the bytes, the parser stays inside
// its buffer (ASan) and either parses or says OTHER
static void box_files(const char *tmpdir)
{
    std::string good = "[";
    for (int b = 0; b < 3; ++b) {
        good += std::string(b ? ", " : "") + "{\"index\": " + std::to_string(b - 1) + ", \"corners_cam0\": [";
        for (int k = 0; k < 8; ++k) good += std::string(k ? "," : "") + "[-1.25e1, 2, 0.30000000000000004]";
        good += "]}";
    }
    good += "]\n";
    const std::string p = std::string(tmpdir) + "/BBoxes_fuzz.json";
    auto run = [&](const std::string &text, int &nb, int &st, double *cs, int32_t *ix, int cap) {
        FILE *f = fopen(p.c_str(), "wb");
        fwrite(text.data(), 1, text.size(), f);
        fclose(f);
        return lpf_parse_boxes_json(p.c_str(), cs, ix, cap, &nb, &st);
    };
    double cs[4 * 24]; int32_t ix[4];
    int nb = -1, st = -1;
    CHECK(run(good, nb, st, cs, ix, 4) == LPF_OK && st == LPF_BOXES_PARSED && nb == 3 && ix[0] == -1 && ix[2] == 1 && cs[0] == -12.5 && cs[71] == 0.30000000000000004);
    CHECK(run(good, nb, st, cs, ix, 2) == LPF_ERR_ARG && nb == 3);
    CHECK(run(good, nb, st, nullptr, nullptr, 0) == LPF_ERR_ARG && nb == 3);
    for (size_t k = 0; k + 2 < good.size(); ++k) CHECK(run(good.substr(0, k), nb, st, cs, ix, 4) == LPF_OK && st == LPF_BOXES_OTHER && nb == 0);
    uint32_t x = 12345;
    auto rnd = [&x] { x = x * 1664525u + 1013904223u; return x >> 8; };
    int parsed = 0;
    for (int it = 0; it < 4000; ++it) {
        std::string t = good;
        const int edits = 1 + (int)(rnd() % 3);
        for (int e = 0; e < edits; ++e) {
            const size_t at = rnd() % t.size();
            switch (rnd() % 4) {
            case 0: t[at] = (char)(rnd() & 0xff); break;
            case 1: t.erase(at, 1 + rnd() % 4); break;
            case 2: t.insert(at, 1, "0123456789.eE+-[]{},:\" \n"[rnd() % 24]); break;
            default: t[at] = "0123456789.eE+-[]{},:\" \n"[rnd() % 24]; break;
            }
            if (t.empty()) t = " ";
        }
        const int rc = run(t, nb, st, cs, ix, 4);
        CHECK((rc == LPF_OK && (st == LPF_BOXES_OTHER || (st == LPF_BOXES_PARSED && nb <= 4))) || (rc == LPF_ERR_ARG && nb > 4));
        parsed += st == LPF_BOXES_PARSED;
    }
    CHECK(parsed > 0 && parsed < 4000);                                   // (a changed digit still parses; a lost bracket does not)
    remove(p.c_str());
    CHECK(run("", nb, st, cs, ix, 4) == LPF_OK && st == LPF_BOXES_OTHER);
    remove(p.c_str());
    CHECK(lpf_parse_boxes_json(p.c_str(), cs, ix, 4, &nb, &st) == LPF_OK && st == LPF_BOXES_ABSENT);
}

int main(int argc, char **argv)
{
    const char *tmp = argc > 1 ? argv[1] : "/tmp";
    const int runs = argc > 2 ? atoi(argv[2]) : 12000;
    argument_errors();
    host_memory_runs();
    ring_of_small_uploads();
    pipelined_streams(2, runs, W, H);
    pipelined_streams(4, runs, W, H);
    pipelined_streams(2, runs / 8, 1408, 376);
    pipelined_streams(4, runs / 8, 1408, 376);
    pipelined_streams(0, runs / 8, 1408, 376);               // in order: the candidate grid goes ahead of the tiles as a kernel
    graphs();
    reader(tmp);
    box_files(tmp);
    fprintf(stderr, "drive: %d failed checks, %lld fake launches\n", g_fail, fake_hip_launches());
    return g_fail ? 1 : 0;
}
