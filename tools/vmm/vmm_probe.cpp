// exploration: does a destination mapped through the HIP virtual-memory API (1 GiB-aligned address, one physical handle) put the
// strided NTT pass into its fast placement reliably?  Compared with plain hipMalloc'd destinations in the same process.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../include/zkhip.h"
#include "../../include/zkhip_chips.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s -> %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)
static float time_pass(zkhip_ctx* ctx, const uint32_t* src, uint32_t* dst, int which) {
    hipStream_t st = (hipStream_t)zkhip_ctx_stream(ctx);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 5; i++) zkhip_ntt_pass(ctx, src, dst, 256, 20, 256, which);
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < 200; i++) zkhip_ntt_pass(ctx, src, dst, 256, 20, 256, which);
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / 200;
}
int main(int argc, char** argv) {
    const size_t GiB = (size_t)1 << 30;
    const int churn = argc > 1 ? std::atoi(argv[1]) : 0;
    zkhip_ctx* ctx = nullptr;
    if (zkhip_ctx_create(0, nullptr, &ctx) != 0) { std::printf("ctx: %s\n", zkhip_last_error()); return 1; }
    // optional churn: allocate and free odd-sized blocks first so that the free lists are not pristine
    std::vector<void*> junk;
    for (int i = 0; i < churn; i++) { void* p; CK(hipMalloc(&p, ((size_t)37 + 61 * i) << 20)); junk.push_back(p); }
    for (size_t i = 0; i < junk.size(); i += 2) CK(hipFree(junk[i]));
    void* src; if (zkhip_malloc(ctx, GiB, &src) != 0) return 1;
    zkhip_fill_uniform(ctx, 1, 20, 256, (uint32_t*)src, 256);
    hipMemAllocationProp prop{};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t gran = 0;
    CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    std::printf("recommended granularity %zu KiB, src %p\n", gran >> 10, src);
    for (int trial = 0; trial < 4; trial++) {
        void* va = nullptr;
        CK(hipMemAddressReserve(&va, GiB, GiB, nullptr, 0));
        hipMemGenericAllocationHandle_t h;
        CK(hipMemCreate(&h, GiB, &prop, 0));
        CK(hipMemMap(va, GiB, 0, h, 0));
        hipMemAccessDesc acc{};
        acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
        CK(hipMemSetAccess(va, GiB, &acc, 1));
        void* plain; CK(hipMalloc(&plain, GiB));
        std::printf("trial %d: vmm dst %p pass0 %.4f pass1 %.4f | hipMalloc dst %p pass0 %.4f pass1 %.4f\n", trial, va,
                    time_pass(ctx, (uint32_t*)src, (uint32_t*)va, 0), time_pass(ctx, (uint32_t*)src, (uint32_t*)va, 1), plain,
                    time_pass(ctx, (uint32_t*)src, (uint32_t*)plain, 0), time_pass(ctx, (uint32_t*)src, (uint32_t*)plain, 1));
        // keep both alive so that the next trial lands elsewhere
    }
    // and the other way round: source in VMM memory
    return 0;
}
