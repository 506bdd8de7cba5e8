// lpf_reader.hip.h -- double-buffered velodyne scan reader (host code, part of lpf_api.hip's
// translation unit).  Stands where the reference calls Kitti360Viewer3DRaw.loadVelodyneData
// (V3:24-28: np.fromfile(path, float32).reshape(-1, 4), RuntimeError when the file is missing)
// once per frame and then hands the array to NumPy; here worker threads read scans k+1, k+2, ...
// from disk into pinned host memory and an internal copy stream moves them to HBM while the
// kernels of scan k run, so the frame loop never waits for the file system or for PCIe.
//
//   submit(path) ... submit(path)        any time, any number (FIFO)
//   next() -> device pointer, host pointer, N of the oldest submitted scan; the caller's stream
//             waits (on the device) for that scan's copy.  The pointers stay valid until the next
//             call of next(); work already enqueued on the context's stream may keep reading them.
//
// A frame's second file, bboxes_3D_cam0/BBoxes_<frame>.json (load_bounding_boxes, V3:31-38: json.load of a list of {"index": int,
// "corners_cam0": 8 x 3 numbers}), can ride along: submit_frame(scan, boxes) has the worker parse it too -- 270 KB of 17-digit
// numbers for frame 2449, 2.4 ms of json.load per frame on the GPU box, more than everything else process_frames does per frame --
// and boxes() hands out the corners as the float64 [B][8][3] array the GPU wants.  Numbers are converted by strtod in the C locale:
// correctly rounded, like Python's float() -- the same doubles.  Anything that is not that plain schema (other keys, NaN, a
// malformed file) is reported as such and left to the caller's JSON library, so its errors stay what they were.
#pragma once

#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>

#include <cerrno>
#include <locale.h>
#include <sys/stat.h>

struct lpf_reader {
    lpf_ctx *ctx = nullptr;
    int device = 0;
    int64_t max_pts = 0;
    struct Slot {
        float *h = nullptr;               // pinned host copy (what np.fromfile would have returned)
        float *d = nullptr;               // HBM copy
        hipEvent_t copied = nullptr;      // H2D of this slot finished (copy stream)
        hipEvent_t consumed = nullptr;    // the consumer's stream is past its last use of the slot
        bool copied_rec = false, consumed_rec = false;
        int64_t n = 0;
        int status = LPF_OK;
        std::string err;
        int box_state = LPF_BOXES_NONE;   // the frame's box file, when one was submitted with the scan
        std::vector<double> box_corners;  // [nbox][8][3]
        std::vector<int32_t> box_index;   // [nbox]
    };
    std::vector<Slot> slots;
    hipStream_t copy_stream = nullptr;
    std::vector<std::thread> workers;     // file reads of consecutive scans overlap (page cache -> pinned copy is
                                          // one core's memcpy rate per thread)
    std::mutex mu;
    std::condition_variable cv;
    std::deque<std::pair<std::string, std::string>> pending;      // {scan, box file or ""} submitted, not yet picked up
    int64_t picked = 0;                   // sequence number of the next scan a worker will take
    std::vector<int> ready;               // ready[seq % slots] = filled slot of scan seq, or -1
    std::deque<int> free_slots;
    int handed = -1;                      // slot given out by the last next()
    int64_t submitted = 0, delivered = 0;
    bool stop = false;
};

namespace {

// ---- BBoxes_<frame>.json --------------------------------------------------------------------------------------------------------
struct BoxCur { const char *p, *e; };

inline void box_ws(BoxCur &c) { while (c.p < c.e && (*c.p == ' ' || *c.p == '\t' || *c.p == '\n' || *c.p == '\r')) ++c.p; }
inline bool box_lit(BoxCur &c, char ch) { box_ws(c); if (c.p < c.e && *c.p == ch) { ++c.p; return true; } return false; }
inline bool box_digit(char ch) { return ch >= '0' && ch <= '9'; }

// one JSON number (RFC 8259 grammar, nothing strtod would accept beyond it): its span, and whether it is an integer literal
bool box_number_span(BoxCur &c, const char *&b, bool &integral)
{
    box_ws(c);
    const char *p = c.p, *e = c.e;
    b = p;
    if (p < e && *p == '-') ++p;
    if (p >= e) return false;
    if (*p == '0') ++p;
    else if (*p >= '1' && *p <= '9') { while (p < e && box_digit(*p)) ++p; }
    else return false;
    integral = true;
    if (p < e && *p == '.') {
        ++p;
        if (p >= e || !box_digit(*p)) return false;
        while (p < e && box_digit(*p)) ++p;
        integral = false;
    }
    if (p < e && (*p == 'e' || *p == 'E')) {
        ++p;
        if (p < e && (*p == '+' || *p == '-')) ++p;
        if (p >= e || !box_digit(*p)) return false;
        while (p < e && box_digit(*p)) ++p;
        integral = false;
    }
    c.p = p;
    return true;
}

locale_t box_c_locale()
{
    static locale_t loc = newlocale(LC_ALL_MASK, "C", (locale_t)0);      // (never freed: one per process)
    return loc;
}

// float(text) as Python computes it: the nearest double (ties to even), +-inf beyond the range
bool box_number(BoxCur &c, double &v)
{
    const char *b; bool integral;
    if (!box_number_span(c, b, integral)) return false;
    const size_t n = (size_t)(c.p - b);
    char small[64];
    std::string big;
    const char *z;
    if (n < sizeof small) { memcpy(small, b, n); small[n] = 0; z = small; }
    else { big.assign(b, n); z = big.c_str(); }
    char *end = nullptr;
    const locale_t loc = box_c_locale();
    if (!loc) return false;                 // (no "C" locale object to be had: nothing is interpreted, the caller's json.load does it)
    v = strtod_l(z, &end, loc);
    if (end != z + n) return false;
    if (integral) {                         // json.load makes an int of it, and the float64 array a double of that int: "-0" is +0.0 there,
        if (v == 0.0) v = 0.0;              // and an integer beyond the doubles' range raises OverflowError instead of becoming inf
        else if (std::isinf(v)) return false;
    }
    return true;
}

bool box_int32(BoxCur &c, int32_t &v)
{
    const char *b; bool integral;
    if (!box_number_span(c, b, integral) || !integral) return false;   // 1.0 or 1e2 is a float to json.load: not this schema
    const size_t n = (size_t)(c.p - b);
    if (n > 11) return false;
    char z[16];
    memcpy(z, b, n); z[n] = 0;
    errno = 0;
    const long long w = strtoll(z, nullptr, 10);
    if (errno || w < INT32_MIN || w > INT32_MAX) return false;
    v = (int32_t)w;
    return true;
}

// a key: "..." without escapes (the two keys of the schema have none)
bool box_key(BoxCur &c, const char *&b, size_t &n)
{
    if (!box_lit(c, '"')) return false;
    b = c.p;
    while (c.p < c.e && *c.p != '"') { if (*c.p == '\\' || (unsigned char)*c.p < 0x20) return false; ++c.p; }
    if (c.p >= c.e) return false;
    n = (size_t)(c.p - b);
    ++c.p;
    return true;
}

// the whole text: [ {"index": i, "corners_cam0": [[x, y, z] x 8]}, ... ] (keys in either order, each exactly once)
bool box_parse_text(const char *text, size_t bytes, std::vector<double> &corners, std::vector<int32_t> &index)
{
    BoxCur c{text, text + bytes};
    corners.clear(); index.clear();
    if (!box_lit(c, '[')) return false;
    if (!box_lit(c, ']')) {
        for (;;) {
            if (!box_lit(c, '{')) return false;
            bool have_i = false, have_c = false;
            int32_t idx = 0;
            const size_t at = corners.size();
            corners.resize(at + 24);
            for (;;) {
                const char *k; size_t kn;
                if (!box_key(c, k, kn) || !box_lit(c, ':')) return false;
                if (kn == 5 && !memcmp(k, "index", 5)) {
                    if (have_i || !box_int32(c, idx)) return false;
                    have_i = true;
                } else if (kn == 12 && !memcmp(k, "corners_cam0", 12)) {
                    if (have_c || !box_lit(c, '[')) return false;
                    for (int r = 0; r < 8; ++r) {
                        if ((r && !box_lit(c, ',')) || !box_lit(c, '[')) return false;
                        for (int a = 0; a < 3; ++a)
                            if ((a && !box_lit(c, ',')) || !box_number(c, corners[at + (size_t)(r * 3 + a)])) return false;
                        if (!box_lit(c, ']')) return false;
                    }
                    if (!box_lit(c, ']')) return false;
                    have_c = true;
                } else {
                    return false;                                   // a key the schema does not have
                }
                if (box_lit(c, ',')) continue;
                if (box_lit(c, '}')) break;
                return false;
            }
            if (!have_i || !have_c) return false;
            index.push_back(idx);
            if (box_lit(c, ',')) continue;
            if (box_lit(c, ']')) break;
            return false;
        }
    }
    box_ws(c);
    return c.p == c.e;                                              // (json.load: "Extra data" otherwise)
}

int box_parse_file(const char *path, std::vector<double> &corners, std::vector<int32_t> &index)
{
    corners.clear(); index.clear();
    FILE *f = fopen(path, "rb");
    if (!f) return errno == ENOENT ? LPF_BOXES_ABSENT : LPF_BOXES_OTHER;     // (FileNotFoundError is the one error V3:31-38 catches)
    struct stat st;
    if (fstat(fileno(f), &st) != 0 || !S_ISREG(st.st_mode) || st.st_size > (1ll << 30)) { fclose(f); return LPF_BOXES_OTHER; }
    std::vector<char> text((size_t)st.st_size);
    size_t got = 0;
    while (got < text.size()) {
        const size_t k = fread(text.data() + got, 1, text.size() - got, f);
        if (k == 0) break;
        got += k;
    }
    fclose(f);
    if (got != text.size() || !box_parse_text(text.data(), text.size(), corners, index)) { corners.clear(); index.clear(); return LPF_BOXES_OTHER; }
    return LPF_BOXES_PARSED;
}

void reader_fill(lpf_reader *r, lpf_reader::Slot &S, const std::string &path, const std::string &boxes)
{
    S.status = LPF_OK; S.err.clear(); S.n = 0;
    S.box_state = boxes.empty() ? LPF_BOXES_NONE : box_parse_file(boxes.c_str(), S.box_corners, S.box_index);   // (host memory only: the
                                                                                                             //  slot is the worker's until it is ready)
    if (S.copied_rec) (void)hipEventSynchronize(S.copied);       // pinned buffer still feeding the previous copy?
    struct stat st;
    FILE *f = fopen(path.c_str(), "rb");
    if (!f || fstat(fileno(f), &st) != 0 || !S_ISREG(st.st_mode)) {
        if (f) fclose(f);
        S.status = LPF_ERR_IO; S.err = path + " does not exist!";                     // V3:26-27
        return;
    }
    const long long bytes = (long long)st.st_size;
    if (bytes % 16 != 0) {                                                            // np.reshape(pcd, [-1, 4]) would raise
        fclose(f);
        char b[160]; snprintf(b, sizeof b, ": %lld bytes is not a whole number of float32 x 4 points", bytes);
        S.status = LPF_ERR_IO; S.err = path + b;
        return;
    }
    if (bytes / 16 > r->max_pts) {
        fclose(f);
        char b[160]; snprintf(b, sizeof b, ": %lld points, reader was created for at most %lld", bytes / 16, (long long)r->max_pts);
        S.status = LPF_ERR_ARG; S.err = path + b;
        return;
    }
    size_t got = 0;
    while (got < (size_t)bytes) {
        const size_t k = fread((char *)S.h + got, 1, (size_t)bytes - got, f);
        if (k == 0) break;
        got += k;
    }
    fclose(f);
    if (got != (size_t)bytes) { S.status = LPF_ERR_IO; S.err = path + ": short read"; return; }
    S.n = bytes / 16;
    hipError_t e = hipSuccess;
    if (S.consumed_rec) e = hipStreamWaitEvent(r->copy_stream, S.consumed, 0);       // HBM buffer still being read?
    if (e == hipSuccess && bytes) e = hipMemcpyAsync(S.d, S.h, (size_t)bytes, hipMemcpyHostToDevice, r->copy_stream);
    if (e == hipSuccess) e = hipEventRecord(S.copied, r->copy_stream);
    if (e != hipSuccess) { S.status = LPF_ERR_HIP; S.err = path + ": " + hipGetErrorString(e); return; }
    S.copied_rec = true;
}

void reader_main(lpf_reader *r)
{
    (void)hipSetDevice(r->device);
    for (;;) {
        std::string path, boxes;
        int s;
        int64_t seq;
        {
            // scans take their slots in submission order, so the oldest outstanding scan always owns one
            std::unique_lock<std::mutex> lk(r->mu);
            r->cv.wait(lk, [r] { return r->stop || (!r->pending.empty() && !r->free_slots.empty()); });
            if (r->stop) return;
            path.swap(r->pending.front().first); boxes.swap(r->pending.front().second); r->pending.pop_front();
            s = r->free_slots.front(); r->free_slots.pop_front();
            seq = r->picked++;
        }
        reader_fill(r, r->slots[(size_t)s], path, boxes);
        {
            std::lock_guard<std::mutex> lk(r->mu);
            r->ready[(size_t)(seq % (int64_t)r->slots.size())] = s;
        }
        r->cv.notify_all();
    }
}

}  // namespace

extern "C" {

void lpf_reader_destroy(lpf_reader *r)
{
    if (!r) return;
    {
        std::lock_guard<std::mutex> lk(r->mu);
        r->stop = true;
    }
    r->cv.notify_all();
    for (auto &w : r->workers) if (w.joinable()) w.join();
    (void)hipSetDevice(r->device);
    if (r->copy_stream) (void)hipStreamSynchronize(r->copy_stream);
    if (r->ctx) (void)hipStreamSynchronize(r->ctx->stream);      // kernels may still read a slot's HBM copy
    for (auto &S : r->slots) {
        if (S.h) (void)hipHostFree(S.h);
        if (S.d) (void)hipFree(S.d);
        if (S.copied) (void)hipEventDestroy(S.copied);
        if (S.consumed) (void)hipEventDestroy(S.consumed);
    }
    if (r->copy_stream) (void)hipStreamDestroy(r->copy_stream);
    delete r;
}

int lpf_reader_create(lpf_ctx *c, lpf_reader **out, int n_buffers, int64_t max_points)
{
    if (!c) return LPF_ERR_ARG;
    if (!out || n_buffers < 2 || n_buffers > 16 || max_points <= 0 || max_points > 0x7fffffffll - LPF_SEG_QUANTUM)
        return fail(c, LPF_ERR_ARG, "reader_create: out=%p n_buffers=%d (2..16) max_points=%lld", (void *)out, n_buffers, (long long)max_points);
    if (use_device(c)) return LPF_ERR_HIP;
    lpf_reader *r = new lpf_reader;
    r->ctx = c; r->device = c->device; r->max_pts = max_points;
    r->slots.resize((size_t)n_buffers);
    hipError_t e = hipStreamCreateWithFlags(&r->copy_stream, hipStreamNonBlocking);
    for (int i = 0; i < n_buffers && e == hipSuccess; ++i) {
        auto &S = r->slots[(size_t)i];
        e = hipHostMalloc((void **)&S.h, (size_t)max_points * 16, hipHostMallocDefault);
        if (e == hipSuccess) e = hipMalloc((void **)&S.d, (size_t)max_points * 16);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&S.copied, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&S.consumed, hipEventDisableTiming);
        r->free_slots.push_back(i);
    }
    if (e != hipSuccess) {
        const int code = (e == hipErrorOutOfMemory) ? LPF_ERR_NOMEM : LPF_ERR_HIP;
        fail(c, code, "reader_create: %s (%d buffers of %lld points)", hipGetErrorString(e), n_buffers, (long long)max_points);
        lpf_reader_destroy(r);
        return code;
    }
    r->ready.assign((size_t)n_buffers, -1);
    const int n_workers = n_buffers > 2 ? 2 : 1;
    for (int i = 0; i < n_workers; ++i) r->workers.emplace_back(reader_main, r);
    *out = r;
    return LPF_OK;
}

int lpf_reader_submit_frame(lpf_reader *r, const char *path, const char *boxes_path)
{
    if (!r) return LPF_ERR_ARG;
    if (!path || !*path) return fail(r->ctx, LPF_ERR_ARG, "reader_submit: empty path");
    {
        std::lock_guard<std::mutex> lk(r->mu);
        r->pending.emplace_back(path, boxes_path ? boxes_path : "");
        ++r->submitted;
    }
    r->cv.notify_all();
    return LPF_OK;
}

int lpf_reader_submit(lpf_reader *r, const char *path) { return lpf_reader_submit_frame(r, path, nullptr); }

int lpf_reader_boxes(lpf_reader *r, const double **corners_cam0, const int32_t **index, int *nbox, int *state)
{
    if (!r) return LPF_ERR_ARG;
    if (corners_cam0) *corners_cam0 = nullptr;
    if (index) *index = nullptr;
    if (nbox) *nbox = 0;
    if (state) *state = LPF_BOXES_NONE;
    if (r->handed < 0) return fail(r->ctx, LPF_ERR_STATE, "reader_boxes: no scan has been handed out");
    const auto &S = r->slots[(size_t)r->handed];                  // (the consumer's until the next lpf_reader_next)
    if (state) *state = S.box_state;
    if (S.box_state == LPF_BOXES_PARSED) {
        if (corners_cam0) *corners_cam0 = S.box_corners.data();
        if (index) *index = S.box_index.data();
        if (nbox) *nbox = (int)S.box_index.size();
    }
    return LPF_OK;
}

int lpf_parse_boxes_json(const char *path, double *corners_cam0, int32_t *index, int cap, int *nbox, int *state)
{
    if (!path || !*path || !nbox || !state || cap < 0 || (cap > 0 && (!corners_cam0 || !index))) return LPF_ERR_ARG;
    std::vector<double> cs;
    std::vector<int32_t> ix;
    *state = box_parse_file(path, cs, ix);
    *nbox = (int)ix.size();
    if (*state != LPF_BOXES_PARSED) return LPF_OK;
    if ((int)ix.size() > cap) return LPF_ERR_ARG;                 // *nbox says what is needed
    if (!ix.empty()) {
        memcpy(corners_cam0, cs.data(), cs.size() * sizeof(double));
        memcpy(index, ix.data(), ix.size() * sizeof(int32_t));
    }
    return LPF_OK;
}

int lpf_reader_next(lpf_reader *r, const float **d_pts, const float **h_pts, int64_t *n_points)
{
    if (!r) return LPF_ERR_ARG;
    lpf_ctx *c = r->ctx;
    if (use_device(c)) return LPF_ERR_HIP;
    if (c->capturing) return fail(c, LPF_ERR_STATE, "reader_next inside graph capture");
    int s;
    {
        std::unique_lock<std::mutex> lk(r->mu);
        if (r->delivered == r->submitted) return fail(c, LPF_ERR_STATE, "reader_next: no scan outstanding (submit first)");
        // the slot handed out last time goes back to the worker once the stream is past the work enqueued so far
        if (r->handed >= 0) {
            auto &Pv = r->slots[(size_t)r->handed];
            hipError_t e = hipEventRecord(Pv.consumed, c->stream);
            if (e != hipSuccess) return fail(c, LPF_ERR_HIP, "reader_next: hipEventRecord: %s", hipGetErrorString(e));
            Pv.consumed_rec = true;
            r->free_slots.push_back(r->handed);
            r->handed = -1;
            r->cv.notify_all();
        }
        const size_t want = (size_t)(r->delivered % (int64_t)r->slots.size());
        r->cv.wait(lk, [r, want] { return r->ready[want] >= 0; });
        s = r->ready[want]; r->ready[want] = -1;
        ++r->delivered;
        r->handed = s;
    }
    auto &S = r->slots[(size_t)s];
    if (d_pts) *d_pts = nullptr;
    if (h_pts) *h_pts = nullptr;
    if (n_points) *n_points = 0;
    if (S.status != LPF_OK) return fail(c, S.status, "%s", S.err.c_str());
    LPF_HIP(c, hipStreamWaitEvent(c->stream, S.copied, 0));
    if (d_pts) *d_pts = S.d;
    if (h_pts) *h_pts = S.h;
    if (n_points) *n_points = S.n;
    return LPF_OK;
}

/* Host-side wait for the H2D copy of the scan returned by the last lpf_reader_next (only needed by
 * callers that read the HBM copy from a stream other than the context's). */
int lpf_reader_wait(lpf_reader *r)
{
    if (!r) return LPF_ERR_ARG;
    if (r->handed < 0) return LPF_OK;
    auto &S = r->slots[(size_t)r->handed];
    if (S.status == LPF_OK && S.copied_rec) LPF_HIP(r->ctx, hipEventSynchronize(S.copied));
    return LPF_OK;
}

}  // extern "C"
