// How much CPU does a host thread burn while it waits for the GPU?  hipStreamSynchronize against hipEventSynchronize on an event with
// and without hipEventBlockingSync, and after hipSetDeviceFlags(hipDeviceScheduleBlockingSync).  build: hipcc --offload-arch=gfx950 -O2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <ctime>
__global__ void spin(uint64_t cycles, uint32_t* out) {
    const uint64_t t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) {}
    if (out) out[0] = 1;
}
static double thread_cpu() { timespec ts; clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts); return ts.tv_sec + ts.tv_nsec * 1e-9; }
static double wall() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + ts.tv_nsec * 1e-9; }
int main(int argc, char** argv) {
    if (argc > 1 && atoi(argv[1])) { hipError_t e = hipSetDeviceFlags(hipDeviceScheduleBlockingSync); printf("hipSetDeviceFlags(BlockingSync): %s\n", hipGetErrorString(e)); }
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    hipEvent_t eb, ed; hipEventCreateWithFlags(&eb, hipEventBlockingSync | hipEventDisableTiming); hipEventCreateWithFlags(&ed, hipEventDisableTiming);
    uint32_t* d; hipMalloc(&d, 4);
    const uint64_t cyc = 100000000ull * 2 / 10;     // wall_clock64 runs at 100 MHz: 0.2 s
    for (int mode = 0; mode < 4; mode++) {
        hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, cyc, d);
        const double c0 = thread_cpu(), w0 = wall();
        uint32_t h = 0;
        if (mode == 0) hipStreamSynchronize(s);
        else if (mode == 1) { hipEventRecord(eb, s); hipEventSynchronize(eb); }
        else if (mode == 2) { hipEventRecord(ed, s); hipEventSynchronize(ed); }
        else { hipMemcpyAsync(&h, d, 4, hipMemcpyDeviceToHost, s); hipStreamSynchronize(s); }
        const double c1 = thread_cpu(), w1 = wall();
        const char* names[4] = {"hipStreamSynchronize", "hipEventSynchronize (BlockingSync event)", "hipEventSynchronize (default event)", "hipMemcpyAsync to pageable + hipStreamSynchronize"};
        printf("%-52s wall %.3f s, this thread's CPU %.3f s\n", names[mode], w1 - w0, c1 - c0);
    }
    return 0;
}
