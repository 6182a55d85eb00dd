// clockprobe -- the shader clock as a time line, sampled beside whatever else the process runs on the GPU ("the in-proof
// clock at the measured operating point is a number, not an inference").  A host thread launches a one-wave kernel every `period_us` on a
// high-priority stream of its own; the wave spins for ~20 us of the 100 MHz real-time counter and records
//     (s_memrealtime at its start, delta s_memtime, delta s_memrealtime)      -> clock = d_memtime / d_realtime x 100 MHz
// (MI355X_MICROARCH.md, "DVFS give-back" item 6).  Not part of the product: tools/ only, nothing in libzkhip.so links it.
//   hipcc --offload-arch=gfx950 -O2 -shared -fPIC tools/clockprobe/clockprobe.hip -o tools/clockprobe/libclockprobe.so
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <thread>

namespace {
__global__ void probe_kernel(uint64_t* out, uint32_t slot, uint64_t spin_ticks) {
    if (threadIdx.x != 0) return;
    const uint64_t r0 = __builtin_amdgcn_s_memrealtime();
    const uint64_t m0 = __builtin_amdgcn_s_memtime();
    uint64_t r1 = r0, m1 = m0;
    while (r1 - r0 < spin_ticks) {
        __builtin_amdgcn_s_sleep(8);
        r1 = __builtin_amdgcn_s_memrealtime();
        m1 = __builtin_amdgcn_s_memtime();
    }
    out[3 * (size_t)slot + 0] = r0;
    out[3 * (size_t)slot + 1] = m1 - m0;
    out[3 * (size_t)slot + 2] = r1 - r0;
}
uint64_t* g_buf = nullptr;      // pinned host memory, device-visible
uint32_t g_cap = 0;
std::atomic<uint32_t> g_n{0};
std::atomic<bool> g_stop{false};
std::thread g_thread;
hipStream_t g_stream = nullptr;
}  // namespace

extern "C" {
// starts sampling: at most `capacity` samples, one every period_us microseconds.  0 on success
int clockprobe_start(int device, uint32_t capacity, uint32_t period_us) {
    if (g_buf) return -1;
    if (hipSetDevice(device) != hipSuccess) return -2;
    if (hipHostMalloc((void**)&g_buf, (size_t)capacity * 24, hipHostMallocDefault) != hipSuccess) return -3;
    for (size_t i = 0; i < (size_t)capacity * 3; i++) g_buf[i] = 0;
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    if (hipStreamCreateWithPriority(&g_stream, hipStreamNonBlocking, hi) != hipSuccess) return -4;
    g_cap = capacity;
    g_n.store(0);
    g_stop.store(false);
    g_thread = std::thread([device, period_us] {
        (void)hipSetDevice(device);
        auto next = std::chrono::steady_clock::now();
        while (!g_stop.load()) {
            const uint32_t i = g_n.load();
            if (i >= g_cap) break;
            hipLaunchKernelGGL(probe_kernel, dim3(1), dim3(64), 0, g_stream, g_buf, i, (uint64_t)2000);
            g_n.store(i + 1);
            next += std::chrono::microseconds(period_us);
            std::this_thread::sleep_until(next);
        }
        (void)hipStreamSynchronize(g_stream);
    });
    return 0;
}
// stops; copies up to `cap` samples (r0, d_memtime, d_realtime) into out; returns how many were taken
uint32_t clockprobe_stop(uint64_t* out, uint32_t cap) {
    if (!g_buf) return 0;
    g_stop.store(true);
    g_thread.join();
    uint32_t n = g_n.load();
    if (n > cap) n = cap;
    for (size_t i = 0; i < (size_t)n * 3; i++) out[i] = g_buf[i];
    (void)hipStreamDestroy(g_stream);
    (void)hipHostFree(g_buf);
    g_buf = nullptr;
    g_stream = nullptr;
    return n;
}
// the device's real-time counter now (for aligning the samples with host-side phase marks): one tiny launch
}
