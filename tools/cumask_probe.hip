// tools/cumask_probe.hip -- development probe (not part of the product): which CUs does bit i of a
// hipExtStreamCreateWithCUMask mask select on this part?  Launches a census kernel on streams whose masks
// have the low n bits set (n = 8, 16, 32, 64, 128) and on the complement, and prints the set of
// (XCC, SE, CU) the blocks ran on.
//   hipcc --offload-arch=gfx950 -O2 -o tools/cumask_probe tools/cumask_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <set>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void census(unsigned *out, int spin)
{
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    // keep the block alive for a while so that every admitted CU gets blocks
    unsigned long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < (unsigned long long)spin) { }
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
}

static void run(const char *name, const std::vector<uint32_t> &mask, unsigned *d_out, int nblk)
{
    hipStream_t s;
    if (mask.empty()) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    else CK(hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data()));
    CK(hipMemsetAsync(d_out, 0xff, (size_t)nblk * 8, s));
    hipLaunchKernelGGL(census, dim3(nblk), dim3(256), 0, s, d_out, 20000);
    CK(hipStreamSynchronize(s));
    std::vector<unsigned> h((size_t)nblk * 2);
    CK(hipMemcpy(h.data(), d_out, h.size() * 4, hipMemcpyDeviceToHost));
    std::map<unsigned, std::set<unsigned>> per_xcc;      // xcc -> {se*100 + sh*16... + cu}
    for (int b = 0; b < nblk; ++b) {
        const unsigned hw = h[2 * b], xcc = h[2 * b + 1] & 0xf;
        const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
        per_xcc[xcc].insert(se * 1000 + sh * 100 + cu);
    }
    size_t total = 0;
    for (auto &kv : per_xcc) total += kv.second.size();
    printf("%-28s distinct CUs %3zu :", name, total);
    for (auto &kv : per_xcc) {
        printf("  xcc%u[", kv.first);
        for (unsigned v : kv.second) printf(" %u.%u.%u", v / 1000, (v / 100) % 10, v % 100);
        printf(" ]");
    }
    printf("\n");
    CK(hipStreamDestroy(s));
}

int main()
{
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    printf("device %s, %d CUs\n", p.name, p.multiProcessorCount);
    const int nblk = 8192;
    unsigned *d_out;
    CK(hipMalloc(&d_out, (size_t)nblk * 8));
    run("no mask", {}, d_out, nblk);
    const int words = (p.multiProcessorCount + 31) / 32;
    for (int n : {8, 16, 32, 64, 128}) {
        std::vector<uint32_t> m(words, 0u), c(words, 0u);
        for (int i = 0; i < p.multiProcessorCount; ++i) {
            if (i < n) m[i / 32] |= 1u << (i % 32); else c[i / 32] |= 1u << (i % 32);
        }
        char nm[64];
        snprintf(nm, sizeof nm, "low %d bits", n);
        run(nm, m, d_out, nblk);
        snprintf(nm, sizeof nm, "all but low %d bits", n);
        run(nm, c, d_out, nblk);
    }
    return 0;
}
