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
#pragma once

#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>

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
    };
    std::vector<Slot> slots;
    hipStream_t copy_stream = nullptr;
    std::vector<std::thread> workers;     // file reads of consecutive scans overlap (page cache -> pinned copy is
                                          // one core's memcpy rate per thread)
    std::mutex mu;
    std::condition_variable cv;
    std::deque<std::string> pending;      // submitted, not yet picked up
    int64_t picked = 0;                   // sequence number of the next scan a worker will take
    std::vector<int> ready;               // ready[seq % slots] = filled slot of scan seq, or -1
    std::deque<int> free_slots;
    int handed = -1;                      // slot given out by the last next()
    int64_t submitted = 0, delivered = 0;
    bool stop = false;
};

namespace {

void reader_fill(lpf_reader *r, lpf_reader::Slot &S, const std::string &path)
{
    S.status = LPF_OK; S.err.clear(); S.n = 0;
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
        std::string path;
        int s;
        int64_t seq;
        {
            // scans take their slots in submission order, so the oldest outstanding scan always owns one
            std::unique_lock<std::mutex> lk(r->mu);
            r->cv.wait(lk, [r] { return r->stop || (!r->pending.empty() && !r->free_slots.empty()); });
            if (r->stop) return;
            path.swap(r->pending.front()); r->pending.pop_front();
            s = r->free_slots.front(); r->free_slots.pop_front();
            seq = r->picked++;
        }
        reader_fill(r, r->slots[(size_t)s], path);
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

int lpf_reader_submit(lpf_reader *r, const char *path)
{
    if (!r) return LPF_ERR_ARG;
    if (!path || !*path) return fail(r->ctx, LPF_ERR_ARG, "reader_submit: empty path");
    {
        std::lock_guard<std::mutex> lk(r->mu);
        r->pending.emplace_back(path);
        ++r->submitted;
    }
    r->cv.notify_all();
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
