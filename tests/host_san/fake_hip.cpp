// fake_hip.cpp -- a FUNCTIONAL stand-in for the HIP runtime, for the sanitizer builds of liblpf's HOST side only
// (tests/host_san/Makefile; never part of the product, never used on a GPU box).  lpf_api.hip is compiled with
// `hipcc --offload-host-only -fsanitize=...` and linked against this file instead of libamdhip64: device memory is host memory
// (calloc), copies are memcpy at the time of the call, kernel launches do nothing, streams / events / graphs are bookkeeping
// objects.  What runs under ASan / UBSan / TSan is therefore everything the host code does around the launches: argument
// validation, the pinned upload ring, the rotation of scratch sets and box sets, the table builders, the graph state machine and
// the reader's worker threads -- with every "device" and "pinned" buffer a heap block whose bounds the sanitizer knows.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>

extern "C" {

typedef int hipError_t;
typedef struct fake_stream { int capturing; } *hipStream_t;
typedef struct fake_event { std::atomic<int> recorded; } *hipEvent_t;
typedef struct fake_graph { int n; } *hipGraph_t;
typedef struct fake_exec { int n; } *hipGraphExec_t;
struct dim3_ { unsigned x, y, z; };

static std::atomic<long long> g_launches{0}, g_copies{0};
long long fake_hip_launches(void) { return g_launches.load(); }
long long fake_hip_copies(void) { return g_copies.load(); }

hipError_t hipGetDeviceCount(int *n) { *n = 1; return 0; }
hipError_t hipSetDevice(int) { return 0; }
hipError_t hipGetLastError(void) { return 0; }
const char *hipGetErrorString(hipError_t) { return "fake HIP error"; }

hipError_t hipMalloc(void **p, size_t n) { *p = calloc(n ? n : 1, 1); return *p ? 0 : 2; }
hipError_t hipFree(void *p) { free(p); return 0; }
hipError_t hipHostMalloc(void **p, size_t n, unsigned) { *p = malloc(n ? n : 1); return *p ? 0 : 2; }
hipError_t hipHostFree(void *p) { free(p); return 0; }
hipError_t hipPointerGetAttributes(void *, const void *) { return 1; }      // (nothing is GPU-mapped here: host results take the copy path)
hipError_t hipMemcpy(void *d, const void *s, size_t n, int) { ++g_copies; memmove(d, s, n); return 0; }
hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, int, hipStream_t) { ++g_copies; memmove(d, s, n); return 0; }
hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) { memset(d, v, n); return 0; }

hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = new fake_stream{0}; return 0; }
hipError_t hipStreamDestroy(hipStream_t s) { delete s; return 0; }
hipError_t hipStreamSynchronize(hipStream_t) { return 0; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return 0; }
hipError_t hipStreamBeginCapture(hipStream_t s, int) { if (s) s->capturing = 1; return 0; }
hipError_t hipStreamEndCapture(hipStream_t s, hipGraph_t *g) { if (s) s->capturing = 0; *g = new fake_graph{1}; return 0; }

hipError_t hipEventCreate(hipEvent_t *e) { *e = new fake_event; (*e)->recorded = 0; return 0; }
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { return hipEventCreate(e); }
hipError_t hipEventDestroy(hipEvent_t e) { delete e; return 0; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->recorded = 1; return 0; }
hipError_t hipEventQuery(hipEvent_t) { return 0; }
hipError_t hipEventSynchronize(hipEvent_t) { return 0; }
hipError_t hipEventElapsedTime(float *ms, hipEvent_t, hipEvent_t) { *ms = 0.001f; return 0; }

hipError_t hipGraphInstantiate(hipGraphExec_t *x, hipGraph_t, void *, void *, size_t) { *x = new fake_exec{1}; return 0; }
hipError_t hipGraphLaunch(hipGraphExec_t, hipStream_t) { ++g_launches; return 0; }
hipError_t hipGraphDestroy(hipGraph_t g) { delete g; return 0; }
hipError_t hipGraphExecDestroy(hipGraphExec_t x) { delete x; return 0; }

// kernel launches: the host stubs hipcc generates push a launch configuration, pop it again and call hipLaunchKernel
static thread_local struct { dim3_ g, b; size_t shm; hipStream_t s; } t_cfg;
hipError_t __hipPushCallConfiguration(dim3_ g, dim3_ b, size_t shm, hipStream_t s) { t_cfg.g = g; t_cfg.b = b; t_cfg.shm = shm; t_cfg.s = s; return 0; }
hipError_t __hipPopCallConfiguration(dim3_ *g, dim3_ *b, size_t *shm, hipStream_t *s) { *g = t_cfg.g; *b = t_cfg.b; *shm = t_cfg.shm; *s = t_cfg.s; return 0; }
hipError_t hipLaunchKernel(const void *, dim3_ g, dim3_ b, void **, size_t, hipStream_t)
{
    if (g.x == 0 || b.x == 0 || b.x > 1024) { fprintf(stderr, "fake_hip: launch with grid %u block %u\n", g.x, b.x); abort(); }
    ++g_launches;
    return 0;
}
void **__hipRegisterFatBinary(const void *) { static void *h; return &h; }
void __hipRegisterFunction(void **, const void *, char *, const char *, unsigned, void *, void *, void *, void *, int *) {}
void __hipRegisterVar(void **, void *, char *, const char *, int, size_t, int, int) {}
void __hipUnregisterFatBinary(void **) {}

}  // extern "C"
