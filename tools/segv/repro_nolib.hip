// tools/segv/repro_nolib.hip -- the thread / launch pattern of libzkhip's batch entries with NO library code: does
// `rocprofv3 --kernel-trace` crash inside hipLaunchKernel on its own?
//   repro_nolib <mode> <rounds>
//     mode t: per round, 16 SHORT-LIVED threads (std::thread, joined at the end of the round -- what deal_jobs does per call), each with
//             its own non-blocking stream, 200 trivial launches + a stream synchronise every 20, while 8 pooled threads do host work
//     mode f: per round, 6 threads ("lanes"), each running 16 FIBERS (ucontext, mmap'ed 1 MiB stacks) that take turns launching from
//             their fiber stacks with arguments read from a pinned ring (what batch.cpp does)
//     mode p: like t, but the 16 threads are created ONCE and reused for every round (is thread churn the trigger?)
//     mode g: per round, 4 short-lived threads; each captures 200 trivial launches into a HIP GRAPH (thread-local capture), instantiates it
//             and launches it 20 times -- what the library's FRI commit phase does outside lock-step batches (csrc/prover.cpp)
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <ucontext.h>

#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <thread>
#include <vector>

__global__ void tiny(uint32_t* p, uint32_t v) { p[blockIdx.x * blockDim.x + threadIdx.x] += v; }
struct BArgs { uint32_t* p; uint32_t v; };
__global__ void tiny_batch(const BArgs* a) { const BArgs& m = a[blockIdx.z]; m.p[blockIdx.x * blockDim.x + threadIdx.x] += m.v; }

static void worker_plain(int launches) {
    hipStream_t s;
    if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) std::abort();
    uint32_t* d;
    if (hipMalloc((void**)&d, 64 * 256 * 4) != hipSuccess) std::abort();
    for (int i = 0; i < launches; i++) {
        hipLaunchKernelGGL(tiny, dim3(64), dim3(256), 0, s, d, (uint32_t)i);
        if (i % 20 == 19) (void)hipStreamSynchronize(s);
    }
    (void)hipStreamSynchronize(s);
    (void)hipFree(d);
    (void)hipStreamDestroy(s);
}

static void worker_graph(int launches) {
    hipStream_t s;
    if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) std::abort();
    uint32_t* d;
    if (hipMalloc((void**)&d, 64 * 256 * 4) != hipSuccess) std::abort();
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    if (hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) != hipSuccess) std::abort();
    for (int i = 0; i < launches; i++) hipLaunchKernelGGL(tiny, dim3(64), dim3(256), 0, s, d, (uint32_t)i);
    if (hipStreamEndCapture(s, &graph) != hipSuccess) std::abort();
    if (hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess) std::abort();
    (void)hipGraphDestroy(graph);
    for (int r = 0; r < 20; r++) {
        if (hipGraphLaunch(exec, s) != hipSuccess) std::abort();
        (void)hipStreamSynchronize(s);
    }
    (void)hipGraphExecDestroy(exec);
    (void)hipFree(d);
    (void)hipStreamDestroy(s);
}

// ---- fibers
struct Lane {
    static constexpr int M = 16;
    ucontext_t lane, fib[M];
    void* stacks[M];
    int current = -1, left[M];
    hipStream_t s;
    uint32_t* d[M];
    BArgs* ring;
    int launches;
};
static thread_local Lane* t_lane;
static void fiber_main() {
    Lane* L = t_lane;
    const int b = L->current;
    for (int i = 0; i < L->launches; i++) {
        L->ring[b] = BArgs{L->d[b], (uint32_t)i};          // a member's request; the lane launches when all are parked
        swapcontext(&L->fib[b], &L->lane);
    }
    L->left[b] = 1;
    swapcontext(&L->fib[b], &L->lane);
}
static void worker_lane(int launches) {
    Lane L;
    t_lane = &L;
    L.launches = launches;
    if (hipStreamCreateWithFlags(&L.s, hipStreamNonBlocking) != hipSuccess) std::abort();
    if (hipHostMalloc((void**)&L.ring, sizeof(BArgs) * Lane::M * 4096, hipHostMallocPortable | hipHostMallocMapped) != hipSuccess) std::abort();
    for (int b = 0; b < Lane::M; b++) {
        if (hipMalloc((void**)&L.d[b], 64 * 256 * 4) != hipSuccess) std::abort();
        L.stacks[b] = mmap(nullptr, 1 << 20, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_STACK, -1, 0);
        getcontext(&L.fib[b]);
        L.fib[b].uc_stack.ss_sp = L.stacks[b]; L.fib[b].uc_stack.ss_size = 1 << 20; L.fib[b].uc_link = nullptr;
        makecontext(&L.fib[b], (void (*)())fiber_main, 0);
        L.left[b] = 0;
    }
    BArgs* slot = L.ring;
    for (int round = 0;; round++) {
        int live = 0;
        for (int b = 0; b < Lane::M; b++) if (!L.left[b]) { L.current = b; swapcontext(&L.lane, &L.fib[b]); if (!L.left[b]) live++; }
        if (!live) break;
        // merged launch: arguments of all members from the pinned ring (copied to the next ring slot, as batch.cpp does)
        BArgs* dst = L.ring + Lane::M * (1 + round % 4095);
        for (int b = 0; b < Lane::M; b++) dst[b] = slot[b];
        hipLaunchKernelGGL(tiny_batch, dim3(64, 1, Lane::M), dim3(256), 0, L.s, (const BArgs*)dst);
        if (round % 20 == 19) (void)hipStreamSynchronize(L.s);
    }
    (void)hipStreamSynchronize(L.s);
    for (int b = 0; b < Lane::M; b++) { (void)hipFree(L.d[b]); munmap(L.stacks[b], 1 << 20); }
    (void)hipHostFree(L.ring);
    (void)hipStreamDestroy(L.s);
}

int main(int argc, char** argv) {
    if (argc < 3) { std::fprintf(stderr, "usage: %s t|f|p|g <rounds>\n", argv[0]); return 1; }
    const char mode = argv[1][0];
    const int rounds = std::atoi(argv[2]);
    std::atomic<bool> stop{false};
    std::atomic<uint64_t> sink{0};
    std::vector<std::thread> host;
    for (int t = 0; t < 8; t++) host.emplace_back([&] { uint64_t x = 1; while (!stop.load()) { for (int i = 0; i < 100000; i++) x = x * 6364136223846793005ull + 1; sink += x; std::this_thread::yield(); } });
    if (mode == 'p') {
        std::mutex mu; std::condition_variable cv; int gen = 0, done = 0; bool quit = false;
        std::vector<std::thread> pool;
        for (int t = 0; t < 16; t++) pool.emplace_back([&] {
            int seen = 0;
            for (;;) {
                { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return quit || gen != seen; }); if (quit) return; seen = gen; }
                worker_plain(200);
                { std::lock_guard<std::mutex> lk(mu); done++; } cv.notify_all();
            }
        });
        for (int r = 0; r < rounds; r++) {
            { std::lock_guard<std::mutex> lk(mu); gen++; done = 0; } cv.notify_all();
            std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return done == 16; });
        }
        { std::lock_guard<std::mutex> lk(mu); quit = true; } cv.notify_all();
        for (auto& t : pool) t.join();
    } else {
        for (int r = 0; r < rounds; r++) {
            std::vector<std::thread> ts;
            const int nt = mode == 'f' ? 6 : mode == 'g' ? 4 : 16;
            for (int t = 0; t < nt; t++) ts.emplace_back(mode == 'f' ? worker_lane : mode == 'g' ? worker_graph : worker_plain, 200);
            for (auto& t : ts) t.join();
        }
    }
    stop = true;
    for (auto& t : host) t.join();
    std::printf("mode %c: %d rounds done\n", mode, rounds);
    return 0;
}
